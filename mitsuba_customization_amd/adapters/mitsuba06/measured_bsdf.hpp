// measured_bsdf.hpp — Mitsuba 0.6 BSDF plugin classes "merl" and "customized_measurement" over
// libmerl_hip (reference: /root/reference/README.md:1 names both plugins; their sources are absent,
// so the class follows the public Mitsuba 0.6 BSDF interface).
//
// Scene usage (same as any 0.6 bsdf):
//   <bsdf type="merl"> <string name="filename" value="gold-metallic-paint.binary"/> </bsdf>
//   <bsdf type="customized_measurement"> <string name="filename" value="mine.binary"/>
//        <float name="scaleR" value="1"/> ... </bsdf>
// Optional: interpolation = "trilinear" (default) | "nearest";  node = "integer" (default) | "center";
//           sampling = "cosine" (default, the upstream convention) | "table" (importance sampling off the table's row
//                      marginal) | "table2d" (off its conditional rows P(theta_h | theta_i): a third less variance again);
//           scalar = "cpu" (default: the virtual per-ray eval / sample / pdf evaluate on the calling render thread, like
//                    the CPU plugin this replaces) | "gpu" (through the device's one-unit call service);
//           device = GPU ordinal (default 0).  Whole-array and wavefront calls (BatchedBSDF) always run on the GPU.
#pragma once
#ifdef MERL_USE_REAL_MITSUBA
#include <mitsuba/render/bsdf.h>
#include <mitsuba/core/properties.h>
#include <mitsuba/core/fresolver.h>
#else
#include <mitsuba/mitsuba.h>
#endif

#include "../common/batched_bsdf.hpp"
#include "../common/merl_gpu_material.hpp"

MTS_NAMESPACE_BEGIN

class MeasuredBSDFBase : public BSDF, public BatchedBSDF {
public:
    explicit MeasuredBSDFBase(const Properties &props) : BSDF(props)
    {
        // scene-relative names resolve through the host's FileResolver, like every 0.6 plugin that reads a file; the name
        // as the scene gave it is what serialize() sends to a network-render worker, whose own resolver finds ITS copy
        m_scene_filename = props.getString("filename");
        m_filename = Thread::getThread()->getFileResolver()->resolve(m_scene_filename).string();
        m_cpu_scalar = merl_gpu::parse_scalar_cpu(props.getString("scalar", "cpu"));
        m_key.device = props.getInteger("device", 0);
        m_key.lookup = merl_gpu::parse_lookup(props.getString("interpolation", "trilinear"));
        m_key.node = merl_gpu::parse_node(props.getString("node", "integer"));
        m_key.disk_map = 0;                       // Mitsuba 0.6's squareToUniformDiskConcentric flavour
        m_key.sampling = merl_gpu::parse_sampling(props.getString("sampling", "cosine"));
        // the conventions only the missing reference source could settle, as properties
        m_key.cosine = merl_gpu::parse_cosine_factor(props.getString("cosine_factor", "included"));
        m_key.negative = merl_gpu::parse_negative_values(props.getString("negative_values", "clamp"));
    }

    // Unserialising constructor (network rendering: the scene object arrives at a worker as a stream).  The
    // worker reloads the table from the same path onto ITS GPU, so the path must resolve there too.
    MeasuredBSDFBase(Stream *stream, InstanceManager *manager) : BSDF(stream, manager)
    {
        m_scene_filename = stream->readString();
        m_filename = Thread::getThread()->getFileResolver()->resolve(m_scene_filename).string();
        m_cpu_scalar = stream->readInt() != 0;
        m_key.device = stream->readInt();
        m_key.lookup = stream->readInt();
        m_key.node = stream->readInt();
        m_key.disk_map = 0;
        m_key.sampling = stream->readInt();
        m_key.cosine = stream->readInt();
        m_key.negative = stream->readInt();
        if (m_key.lookup < 0 || m_key.lookup > 1 || m_key.node < 0 || m_key.node > 1 || m_key.sampling < 0 || m_key.sampling > 2 ||
            m_key.cosine < 0 || m_key.cosine > 1 || m_key.negative < 0 || m_key.negative > 2)
            throw merl_gpu::Error(MRL_ERR_INVALID, "corrupt serialised BSDF");
    }

    void serialize(Stream *stream, InstanceManager *manager) const override
    {
        BSDF::serialize(stream, manager);
        stream->writeString(m_scene_filename);
        stream->writeInt(m_cpu_scalar ? 1 : 0);
        stream->writeInt(m_key.device);
        stream->writeInt(m_key.lookup);
        stream->writeInt(m_key.node);
        stream->writeInt(m_key.sampling);
        stream->writeInt(m_key.cosine);
        stream->writeInt(m_key.negative);
    }

    void configure() override
    {
        m_components.clear();
        m_components.push_back(EGlossyReflection | EFrontSide);
        m_usesRayDifferentials = false;
        BSDF::configure();
    }

    // eval: f * cos(theta_o); zero unless both directions are on the front side, the measure is the
    // solid angle and the query asks for the (only) reflection lobe
    Spectrum eval(const BSDFSamplingRecord &bRec, EMeasure measure) const override
    {
        if (!(bRec.typeMask & EGlossyReflection) || measure != ESolidAngle || (bRec.component != -1 && bRec.component != 0))
            return Spectrum(0.0f);
        const float wi[3] = { bRec.wi.x, bRec.wi.y, bRec.wi.z }, wo[3] = { bRec.wo.x, bRec.wo.y, bRec.wo.z };
        float rgb[3];
        m_material.eval1(wi, wo, rgb);
        Spectrum s;
        s.fromLinearRGB(rgb[0], rgb[1], rgb[2]);
        return s;
    }

    Float pdf(const BSDFSamplingRecord &bRec, EMeasure measure) const override
    {
        if (!(bRec.typeMask & EGlossyReflection) || measure != ESolidAngle || (bRec.component != -1 && bRec.component != 0))
            return 0.0f;
        const float wi[3] = { bRec.wi.x, bRec.wi.y, bRec.wi.z }, wo[3] = { bRec.wo.x, bRec.wo.y, bRec.wo.z };
        return m_material.pdf1(wi, wo);
    }

    Spectrum sample(BSDFSamplingRecord &bRec, Float &pdf, const Point2 &sample) const override
    {
        pdf = 0.0f;
        if (!(bRec.typeMask & EGlossyReflection) || (bRec.component != -1 && bRec.component != 0))
            return Spectrum(0.0f);
        const float wi[3] = { bRec.wi.x, bRec.wi.y, bRec.wi.z }, u[2] = { sample.x, sample.y };
        float wo[3], w[3];
        m_material.sample1(wi, u, wo, pdf, w);
        if (!(pdf > 0.0f)) return Spectrum(0.0f);
        bRec.wo = Vector(wo[0], wo[1], wo[2]);
        bRec.eta = 1.0f;
        bRec.sampledComponent = 0;
        bRec.sampledType = EGlossyReflection;
        Spectrum s;
        s.fromLinearRGB(w[0], w[1], w[2]);
        return s;
    }

    Spectrum sample(BSDFSamplingRecord &bRec, const Point2 &sample_) const override
    {
        Float pdf;
        return sample(bRec, pdf, sample_);
    }

    // ---- BatchedBSDF ----
    void evalBatch(const float *wi, const float *wo, size_t n, float *rgb) const override { m_material.eval_batch(wi, wo, n, rgb); }
    void pdfBatch(const float *wi, const float *wo, size_t n, float *pdf) const override { m_material.pdf_batch(wi, wo, n, pdf); }
    void sampleBatch(const float *wi, const float *u, size_t n, float *wo, float *pdf, float *weight) const override
    {
        m_material.sample_batch(wi, u, n, wo, pdf, weight);
    }
    void evalSampleBatch(const float *wi, const float *wo, const float *u, size_t n,
                         float *rgb, float *pdf, float *wo2, float *pdf2, float *weight) const override
    {
        m_material.eval_sample_batch(wi, wo, u, n, rgb, pdf, wo2, pdf2, weight);
    }
    void evalPdfBatch(const float *wi, const float *wo, size_t n, float *rgb, float *pdf) const override
    {
        m_material.eval_pdf_batch(wi, wo, n, rgb, pdf);
    }
    void evalSampleQueue(const float *wi, const float *wo, const float *u,
                         const uint32_t *queue, const uint32_t *count, size_t capacity,
                         float *rgb, float *pdf, float *wo2, float *pdf2, float *weight) const override
    {
        m_material.eval_sample_queue(wi, wo, u, queue, count, capacity, rgb, pdf, wo2, pdf2, weight);
    }
    void synchronize() const override { m_material.synchronize(); }

    std::string toString() const override
    {
        std::ostringstream oss;
        oss << pluginName() << "[filename=\"" << m_filename << "\", device=" << m_key.device
            << ", interpolation=" << (m_key.lookup ? "trilinear" : "nearest") << ", scalar=" << (m_material.cpu_scalar() ? "cpu" : "gpu")
            << ", material=" << m_material.id() << "]";
        return oss.str();
    }

protected:
    virtual const char *pluginName() const = 0;
    // after the subclass has loaded m_material
    void finish_load() { if (m_cpu_scalar) m_material.use_cpu_scalar(); }
    std::string m_filename;          // resolved on this host
    std::string m_scene_filename;    // as the scene (or the master) named it
    bool m_cpu_scalar = true;
    merl_gpu::ContextKey m_key;
    merl_gpu::Material m_material;
};

// type="merl": MERL .binary (90 x 90 x 180, fixed channel scales)
class MerlBSDF : public MeasuredBSDFBase {
public:
    explicit MerlBSDF(const Properties &props) : MeasuredBSDFBase(props)
    {
        m_material = merl_gpu::Material::load_merl(m_key, m_filename);
        finish_load();
    }
    MerlBSDF(Stream *stream, InstanceManager *manager) : MeasuredBSDFBase(stream, manager)
    {
        m_material = merl_gpu::Material::load_merl(m_key, m_filename);
        finish_load();
        configure();
    }
    MTS_DECLARE_CLASS()
protected:
    const char *pluginName() const override { return "MerlBSDF"; }
};

// type="customized_measurement": same parameterisation, table dims from the file header, channel
// scales from the scene (the reference's own format is unknown — SURVEY.md Appendix B item 7)
class CustomizedMeasurement : public MeasuredBSDFBase {
public:
    explicit CustomizedMeasurement(const Properties &props) : MeasuredBSDFBase(props)
    {
        m_scale[0] = props.getFloat("scaleR", 1.0f); m_scale[1] = props.getFloat("scaleG", 1.0f); m_scale[2] = props.getFloat("scaleB", 1.0f);
        // which three angles index the table: "half_diff" (MERL's, default), "standard" (theta_i, theta_o, |dphi|),
        // "standard_full" (theta_i, theta_o, dphi mod 2 pi) — include/merl_hip.h enum mrl_param
        m_param = merl_gpu::parse_parameterization(props.getString("parameterization", "half_diff"));
        if (merl_gpu::Material::is_tensor_file(m_filename) &&
            (props.hasProperty("scaleR") || props.hasProperty("scaleG") || props.hasProperty("scaleB")))
            throw merl_gpu::Error(MRL_ERR_INVALID, m_scene_filename + ": a tensor_file table brings its own channel scales (field \"scale\"); "
                                                   "scaleR / scaleG / scaleB do not apply to it");
        load();
    }
    CustomizedMeasurement(Stream *stream, InstanceManager *manager) : MeasuredBSDFBase(stream, manager)
    {
        for (int c = 0; c < 3; ++c) m_scale[c] = stream->readFloat();
        m_param = stream->readInt();
        load();
        configure();
    }
    void serialize(Stream *stream, InstanceManager *manager) const override
    {
        MeasuredBSDFBase::serialize(stream, manager);
        for (int c = 0; c < 3; ++c) stream->writeFloat(m_scale[c]);
        stream->writeInt(m_param);
    }
    MTS_DECLARE_CLASS()
protected:
    const char *pluginName() const override { return "CustomizedMeasurement"; }
private:
    void load()
    {
        const double scale[3] = { m_scale[0], m_scale[1], m_scale[2] };
        // *.bsdf: the table sits in a tensor_file container and brings its own channel scales
        m_material = merl_gpu::Material::is_tensor_file(m_filename) ? merl_gpu::Material::load_tensor_table(m_key, m_filename, m_param)
                                                                    : merl_gpu::Material::load_table(m_key, m_filename, scale, m_param);
        finish_load();
    }
    Float m_scale[3];
    int m_param = 0;
};

MTS_NAMESPACE_END
