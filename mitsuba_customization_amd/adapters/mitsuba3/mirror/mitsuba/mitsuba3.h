// ABI-MIRROR of the Mitsuba 3 declarations a BSDF plugin touches.
//
// NOT Mitsuba source: this repo's own minimal re-declaration of the public plugin-facing
// interface — template <Float, Spectrum> class BSDF with sample / eval / pdf / eval_pdf /
// to_string, BSDFContext, BSDFSample3, BSDFFlags, Properties, MI_EXPORT_PLUGIN — so that
// adapters/mitsuba3/*.cpp compile and run in a container without a Mitsuba 3 tree (the reference's
// mitsuba3/ gitlink is empty; its pinned version, hence MI_* vs MTS_* spelling, is unknown —
// SURVEY.md Appendix B item 8).  Only the scalar_rgb variant is mirrored (Float = float,
// Spectrum = Color3f): the array variants of upstream come from Dr.Jit, which this build replaces
// by the explicit wavefront entry points of BatchedBSDF (north_star: no Dr.Jit / LLVM / OptiX).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#define MI_EXPORT __attribute__((visibility("default")))
#define NAMESPACE_BEGIN(name) namespace name {
#define NAMESPACE_END(name) }

NAMESPACE_BEGIN(mitsuba)

struct Vector3f {
    float x_, y_, z_;
    Vector3f() : x_(0), y_(0), z_(0) {}
    Vector3f(float x, float y, float z) : x_(x), y_(y), z_(z) {}
    float x() const { return x_; }
    float y() const { return y_; }
    float z() const { return z_; }
};
struct Point2f {
    float x_, y_;
    Point2f() : x_(0), y_(0) {}
    Point2f(float x, float y) : x_(x), y_(y) {}
    float x() const { return x_; }
    float y() const { return y_; }
};
struct Color3f {
    float v[3];
    Color3f() { v[0] = v[1] = v[2] = 0; }
    explicit Color3f(float a) { v[0] = v[1] = v[2] = a; }
    Color3f(float r, float g, float b) { v[0] = r; v[1] = g; v[2] = b; }
    float operator[](int i) const { return v[i]; }
    float &operator[](int i) { return v[i]; }
};

// the spectral variants' Spectrum: four wavelength samples per ray (upstream: Spectrum<Float, 4>), and the traits a plugin tells
// the variants apart with
struct Spectrum4f {
    float v[4];
    Spectrum4f() { v[0] = v[1] = v[2] = v[3] = 0; }
    explicit Spectrum4f(float a) { v[0] = v[1] = v[2] = v[3] = a; }
    Spectrum4f(float a, float b, float c, float d) { v[0] = a; v[1] = b; v[2] = c; v[3] = d; }
    float operator[](int i) const { return v[i]; }
    float &operator[](int i) { return v[i]; }
    static constexpr int Size = 4;
};
template <typename T> constexpr bool is_spectral_v = false;
template <> inline constexpr bool is_spectral_v<Spectrum4f> = true;
template <typename T> constexpr bool is_rgb_v = !is_spectral_v<T>;

struct Frame3f {
    static float cos_theta(const Vector3f &v) { return v.z(); }
};

enum class TransportMode : uint32_t { Radiance = 0, Importance = 1 };

enum class BSDFFlags : uint32_t {
    Empty = 0x00000, Null = 0x00001, DiffuseReflection = 0x00002, DiffuseTransmission = 0x00004,
    GlossyReflection = 0x00008, GlossyTransmission = 0x00010, DeltaReflection = 0x00020, DeltaTransmission = 0x00040,
    Anisotropic = 0x01000, SpatiallyVarying = 0x02000, NonSymmetric = 0x04000, FrontSide = 0x08000, BackSide = 0x10000
};
constexpr uint32_t operator|(BSDFFlags a, BSDFFlags b) { return (uint32_t)a | (uint32_t)b; }
constexpr uint32_t operator+(BSDFFlags a) { return (uint32_t)a; }

struct BSDFContext {
    TransportMode mode = TransportMode::Radiance;
    uint32_t type_mask = 0x1FFu;
    uint32_t component = (uint32_t)-1;
    bool is_enabled(BSDFFlags type, uint32_t comp = 0) const
    {
        return (type_mask & (uint32_t)type) != 0 && (component == (uint32_t)-1 || component == comp);
    }
};

template <typename Float, typename Spectrum> struct SurfaceInteraction {
    Vector3f wi;     // incident direction, local shading frame
    Spectrum wavelengths;   // the ray's wavelengths in nm (spectral variants; unused in the RGB ones — upstream: an empty Color0f there)
};

template <typename Float, typename Spectrum> struct BSDFSample3 {
    Vector3f wo;
    Float pdf = 0;
    Float eta = 1;
    uint32_t sampled_type = 0;
    uint32_t sampled_component = (uint32_t)-1;
};

class Properties {
public:
    Properties() {}
    explicit Properties(const std::string &plugin_name) : m_plugin(plugin_name) {}
    const std::string &plugin_name() const { return m_plugin; }
    bool has_property(const std::string &n) const { return m_str.count(n) || m_num.count(n); }
    void set_string(const std::string &n, const std::string &v) { m_str[n] = v; }
    void set_float(const std::string &n, double v) { m_num[n] = v; }
    void set_int(const std::string &n, int64_t v) { m_num[n] = (double)v; }
    std::string string(const std::string &n) const
    {
        auto it = m_str.find(n);
        if (it == m_str.end()) throw std::runtime_error("Property \"" + n + "\" has not been specified!");
        return it->second;
    }
    std::string string(const std::string &n, const std::string &def) const { auto it = m_str.find(n); return it == m_str.end() ? def : it->second; }
    template <typename T> T get(const std::string &n, T def) const { auto it = m_num.find(n); return it == m_num.end() ? def : (T)it->second; }
private:
    std::string m_plugin;
    std::map<std::string, std::string> m_str;
    std::map<std::string, double> m_num;
};

// ---- FileResolver / Thread: Thread::thread()->file_resolver()->resolve(name), as Mitsuba 3 plugins do ----
struct ResolvedPath {
    std::string m_path;
    const std::string &string() const { return m_path; }
};
class FileResolver {
public:
    // The real resolver is filled by the scene loader (the scene file's directory) and lives in the host's core
    // library.  The mirror has no such library — every plugin .so carries its own copy — so its search path comes
    // from the environment: MITSUBA_MIRROR_DATA_PATH = dir[:dir...] (what the test driver sets for "the scene's directory").
    FileResolver()
    {
        if (const char *env = std::getenv("MITSUBA_MIRROR_DATA_PATH")) {
            std::string all(env);
            size_t a = 0;
            while (a <= all.size()) {
                const size_t b = all.find(':', a);
                const std::string dir = all.substr(a, b == std::string::npos ? std::string::npos : b - a);
                if (!dir.empty()) m_dirs.push_back(dir);
                if (b == std::string::npos) break;
                a = b + 1;
            }
        }
    }
    void append(const std::string &dir) { m_dirs.push_back(dir); }
    void prepend(const std::string &dir) { m_dirs.insert(m_dirs.begin(), dir); }
    ResolvedPath resolve(const std::string &name) const
    {
        if (!name.empty() && name[0] != '/')
            for (const std::string &d : m_dirs) {
                const std::string candidate = d + "/" + name;
                if (FILE *f = std::fopen(candidate.c_str(), "rb")) { std::fclose(f); return { candidate }; }
            }
        return { name };
    }
private:
    std::vector<std::string> m_dirs;
};
class Thread {
public:
    static Thread *thread() { static Thread t; return &t; }
    FileResolver *file_resolver() { return &m_resolver; }
private:
    FileResolver m_resolver;
};

class Object {
public:
    virtual ~Object() {}
    virtual std::string to_string() const = 0;
};

class TraversalCallback;

template <typename Float_, typename Spectrum_> class BSDF : public Object {
public:
    using Float = Float_;
    using Spectrum = Spectrum_;
    using Mask = bool;
    using SurfaceInteraction3f = SurfaceInteraction<Float, Spectrum>;
    using BSDFSample3f = BSDFSample3<Float, Spectrum>;

    explicit BSDF(const Properties &props) : m_flags(0), m_id(props.string("id", "")) {}

    virtual std::pair<BSDFSample3f, Spectrum> sample(const BSDFContext &ctx, const SurfaceInteraction3f &si, Float sample1,
                                                     const Point2f &sample2, Mask active = true) const = 0;
    virtual Spectrum eval(const BSDFContext &ctx, const SurfaceInteraction3f &si, const Vector3f &wo, Mask active = true) const = 0;
    virtual Float pdf(const BSDFContext &ctx, const SurfaceInteraction3f &si, const Vector3f &wo, Mask active = true) const = 0;
    virtual std::pair<Spectrum, Float> eval_pdf(const BSDFContext &ctx, const SurfaceInteraction3f &si, const Vector3f &wo,
                                                Mask active = true) const
    {
        return { eval(ctx, si, wo, active), pdf(ctx, si, wo, active) };
    }
    virtual void traverse(TraversalCallback *) {}

    uint32_t flags() const { return m_flags; }
    uint32_t flags(size_t i) const { return m_components[i]; }
    size_t component_count() const { return m_components.size(); }
    const std::string &id() const { return m_id; }

protected:
    uint32_t m_flags;
    std::vector<uint32_t> m_components;
    std::string m_id;
};

NAMESPACE_END(mitsuba)

#define MI_IMPORT_BASE(Name, ...) using Base = mitsuba::Name<Float, Spectrum>;
#define MI_IMPORT_TYPES(...)
#define MI_DECLARE_CLASS()
#define MI_IMPLEMENT_CLASS_VARIANT(Name, Parent)
// what the plugin manager resolves after dlopen(): the plugin's name/description and, in this
// mirror, the constructor of its scalar_rgb instantiation
#define MI_EXPORT_PLUGIN(Name, Descr)                                                                     \
    extern "C" {                                                                                          \
    MI_EXPORT const char *plugin_name() { return #Name; }                                                 \
    MI_EXPORT const char *plugin_descr() { return Descr; }                                                \
    MI_EXPORT void *plugin_create_scalar_rgb(const mitsuba::Properties &props)                            \
    {                                                                                                     \
        return new mitsuba::Name<float, mitsuba::Color3f>(props);                                         \
    }                                                                                                     \
    MI_EXPORT void *plugin_create_scalar_spectral(const mitsuba::Properties &props)                       \
    {                                                                                                     \
        return new mitsuba::Name<float, mitsuba::Spectrum4f>(props);                                      \
    }                                                                                                     \
    }
