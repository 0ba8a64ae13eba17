// merl_gpu_material.hpp — what both plugin adapters (Mitsuba 0.6 and Mitsuba 3) share: one
// libmerl_hip context per GPU and process, a material handle, and the 1-unit "scalar call"
// plumbing.  Host C++ only; everything that computes goes through the C ABI (include/merl_hip.h).
//
// Threading: the renderers call eval()/sample()/pdf() on a const BSDF from all render threads
// (SURVEY.md §8b).  Where a ONE-unit (scalar, virtual) call is evaluated is the plugin's `scalar` property:
//   "cpu" (default)  on the calling render thread, over a host image of the resident table, with the kernels' own
//                    per-unit functions compiled for the host (mrl_host_*; SURVEY.md §8b "what calls it (2)"): lock-free,
//                    ~0.2-0.4 us per call — what the CPU plugin this replaces costs;
//   "gpu"            through the library's one-unit call service (mrl_scalar_*): the thread writes its request into a
//                    mailbox slot in pinned memory and a resident wave answers it — a PCIe round trip per call
//                    (4.8-6.7 us alone, ~0.6 us amortised over 16 threads).
// Either way the table is resident on the GPU and the batch / wavefront entry points — the fast path, whole arrays —
// run there; without a gfx950 device the constructor throws.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>

#include <sys/stat.h>

#include "../../../include/merl_hip.h"

namespace merl_gpu {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string &what) : std::runtime_error(what), status(st) {}
};

inline void check(mrl_ctx *ctx, int rc, const char *what)
{
    if (rc == MRL_OK) return;
    std::string msg = std::string(what) + ": " + mrl_strerror(rc);
    if (ctx) { const char *d = mrl_last_error(ctx); if (d && *d) msg += std::string(" (") + d + ")"; }
    throw Error(rc, msg);
}

// One context per (process, device).  Lookup mode / node convention / disk map are context-wide
// options in the C ABI, so plugins that ask for different ones on one device get separate contexts.
struct ContextKey {
    int device, lookup, node, disk_map;
    int sampling = 0;              // 0 cosine hemisphere, 1 table importance sampling (MRL_OPT_SAMPLING)
    int cosine = 0;                // MRL_OPT_COSINE_FACTOR: 0 eval() = f cos(theta_o), 1 eval() = f      
    int negative = 0;              // MRL_OPT_NEGATIVE: 0 clamp, 1 keep, 2 skip and renormalise             
    bool operator<(const ContextKey &o) const
    {
        if (device != o.device) return device < o.device;
        if (lookup != o.lookup) return lookup < o.lookup;
        if (node != o.node) return node < o.node;
        if (disk_map != o.disk_map) return disk_map < o.disk_map;
        if (sampling != o.sampling) return sampling < o.sampling;
        if (cosine != o.cosine) return cosine < o.cosine;
        return negative < o.negative;
    }
};

class Context {
public:
    explicit Context(const ContextKey &key) : m_key(key)
    {
        int rc = mrl_init(key.device, &m_ctx);
        if (rc != MRL_OK)
            throw Error(rc, std::string("mrl_init: ") + mrl_strerror(rc));   // no GPU: there is no CPU fallback
        check(m_ctx, mrl_set_option(m_ctx, MRL_OPT_LOOKUP, key.lookup), "mrl_set_option(lookup)");
        // nearest lookups read one texel: the compact rows layout keeps half of them in L2 (32 vs 22 G units/s);
        // trilinear lookups want the whole neighbourhood in one line: bricks
        check(m_ctx, mrl_set_option(m_ctx, MRL_OPT_TABLE_LAYOUT, key.lookup == 0 ? 0 : 1), "mrl_set_option(layout)");
        check(m_ctx, mrl_set_option(m_ctx, MRL_OPT_NODE, key.node), "mrl_set_option(node)");
        check(m_ctx, mrl_set_option(m_ctx, MRL_OPT_DISK_MAP, key.disk_map), "mrl_set_option(disk_map)");
        check(m_ctx, mrl_set_option(m_ctx, MRL_OPT_SAMPLING, key.sampling), "mrl_set_option(sampling)");
        check(m_ctx, mrl_set_option(m_ctx, MRL_OPT_COSINE_FACTOR, key.cosine), "mrl_set_option(cosine factor)");
        check(m_ctx, mrl_set_option(m_ctx, MRL_OPT_NEGATIVE, key.negative), "mrl_set_option(negative values)");     // before the first table
        // scenes with many measured materials: MERL_TABLE_ARENA_MB=<MB> places their tables back to back in one device
        // allocation (multi-table launches are bound by address translation)
        if (const char *mb = std::getenv("MERL_TABLE_ARENA_MB")) {
            const long v = std::atol(mb);
            if (v > 0) check(m_ctx, mrl_set_option(m_ctx, MRL_OPT_TABLE_ARENA_MB, (int)v), "mrl_set_option(table arena)");
        }
    }
    ~Context()
    {
        if (m_ctx) mrl_destroy(m_ctx);
    }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;

    mrl_ctx *raw() const { return m_ctx; }
    std::mutex &mutex() { return m_mutex; }

    // ---- table residency: one upload per (file, channel scales) and context, shared by every plugin instance that
    // names it (two <bsdf> elements with the same .binary, unserialised copies on a worker, scene reloads); the
    // table leaves HBM when the last instance dies (mrl_material_release).
    struct Resident {
        Context *owner;                                 // kept alive by the Material's shared_ptr<Context>
        std::string key;
        int id;
        unsigned long long serial = next_serial();     // never reused (a freed Resident's address and material id can be)
        static unsigned long long next_serial()
        {
            static std::mutex mu;
            static unsigned long long n = 0;
            std::lock_guard<std::mutex> lk(mu);
            return ++n;
        }
        // the host image for one-unit calls on the CPU: taken once per resident table, on the first instance that wants it
        const mrl_host_table *host_table()
        {
            std::lock_guard<std::mutex> call(owner->m_mutex);
            if (!host) check(owner->m_ctx, mrl_material_host_table(owner->m_ctx, id, &host), "mrl_material_host_table");
            return host;
        }
        mrl_host_table *host = nullptr;
        ~Resident()
        {
            if (host) mrl_host_table_release(host);
            std::lock_guard<std::mutex> call(owner->m_mutex);
            auto it = owner->m_resident.find(key);      // a newer upload of the same key may already sit there: keep it
            if (it != owner->m_resident.end() && it->second.expired()) owner->m_resident.erase(it);
            mrl_material_release(owner->m_ctx, id);     // best effort in a destructor
        }
    };
    // loader(ctx, &id) performs the upload when the key is not resident yet (called with the context mutex held)
    template <typename Loader>
    std::shared_ptr<Resident> acquire(const std::string &key, Loader &&loader, const char *what)
    {
        std::unique_lock<std::mutex> lock(m_mutex);
        auto it = m_resident.find(key);
        if (it != m_resident.end())
            if (auto sp = it->second.lock()) return sp;
        int id = -1;
        check(m_ctx, loader(m_ctx, &id), what);
        auto sp = std::shared_ptr<Resident>(new Resident{ this, key, id });     // serial: its default member initialiser
        m_resident[key] = sp;
        return sp;
    }
    size_t resident_tables()
    {
        std::lock_guard<std::mutex> lock(m_mutex);
        return m_resident.size();
    }

    // One virtual BSDF call = one request to the library's scalar service (include/merl_hip.h, "one-unit calls"):
    // the request goes into a mailbox in pinned memory, a resident wave answers it — no launch, no synchronisation, no lock
    // shared with the other render threads, whose calls are answered side by side by the wave's other lanes.
    void scalar_eval_pdf(int material, const float wi[3], const float wo[3], float rgb[3], float &pdf)
    {
        scalar_check(mrl_scalar_eval_pdf(m_ctx, material, wi, wo, rgb, &pdf));
    }
    void scalar_sample(int material, const float wi[3], const float u[2], float wo[3], float &pdf, float weight[3])
    {
        scalar_check(mrl_scalar_sample(m_ctx, material, wi, u, wo, &pdf, weight));
    }

    static std::shared_ptr<Context> get(const ContextKey &key)
    {
        static std::mutex reg_mutex;
        static std::map<ContextKey, std::weak_ptr<Context>> registry;
        std::lock_guard<std::mutex> lock(reg_mutex);
        auto it = registry.find(key);
        if (it != registry.end())
            if (auto sp = it->second.lock()) return sp;
        auto sp = std::make_shared<Context>(key);
        registry[key] = sp;
        return sp;
    }

private:
    void scalar_check(int rc)
    {
        if (rc == MRL_OK) return;
        std::string what = std::string("scalar call: ") + mrl_strerror(rc);
        const char *d = mrl_last_error(m_ctx);
        if (d && *d) what += std::string(" (") + d + ")";
        throw Error(rc, what);
    }

    ContextKey m_key;
    mrl_ctx *m_ctx = nullptr;
    std::mutex m_mutex;                    // keeps a launch and its synchronize together (the C context only locks per call)
    std::map<std::string, std::weak_ptr<Resident>> m_resident;     // guarded by m_mutex
};

// MERL_IMAGE_CACHE_DIR=<dir>: table files are made resident through the library's on-disk image cache (mrl_material_save_image /
// _load_image: the device image, no parse / re-layout / sampling-table kernels — 5 ms instead of 12 for a MERL file).  An image is
// named after everything it depends on: the source's resolved path, size and modification time, the upload's parameters and the
// context's lookup options; a stale, foreign or damaged image is refused by the library and simply rewritten.
inline std::string image_cache_name(const std::string &source, const std::string &what, const ContextKey &key)
{
    const char *dir = std::getenv("MERL_IMAGE_CACHE_DIR");
    if (!dir || !*dir) return std::string();
    struct stat st;
    if (::stat(source.c_str(), &st) != 0) return std::string();
    char tag[512];
    std::snprintf(tag, sizeof tag, "|%lld|%lld.%09ld|%s|l%d|n%d|v%d", (long long)st.st_size, (long long)st.st_mtim.tv_sec, (long)st.st_mtim.tv_nsec,
                  what.c_str(), key.lookup, key.node, key.negative == 0 ? 0 : 1);        // (an image holds clamped or raw values)
    unsigned long long h = 0xCBF29CE484222325ull;
    for (const std::string &part : { source, std::string(tag) })
        for (unsigned char c : part) h = (h ^ c) * 0x100000001B3ull;
    char name[32];
    std::snprintf(name, sizeof name, "%016llx.mrlimg", h);
    return std::string(dir) + "/" + name;
}

// loader through the cache: the image if there is a good one, else the source — and its image for the next process
template <typename Loader>
inline int load_through_image_cache(mrl_ctx *c, const std::string &image, Loader &&from_source, int *id)
{
    if (image.empty()) return from_source(c, id);
    if (mrl_material_load_image(c, image.c_str(), id) == MRL_OK) return MRL_OK;
    const int rc = from_source(c, id);
    if (rc == MRL_OK) (void)mrl_material_save_image(c, *id, image.c_str());      // best effort: a read-only cache directory is not an error
    return rc;
}

// Canonical name of a table file for the residency map: the resolved absolute path when the file exists.
inline std::string canonical_path(const std::string &path)
{
    char buf[4096];
    if (::realpath(path.c_str(), buf)) return std::string(buf);
    return path;
}

// A material living on the GPU + the calls the plugin classes forward to.  Copies share the resident table.
class Material {
public:
    Material() = default;

    static Material load_merl(const ContextKey &key, const std::string &path)
    {
        auto ctx = Context::get(key);
        const std::string image = image_cache_name(canonical_path(path), "merl", key);
        auto res = ctx->acquire("merl|" + canonical_path(path),
                                [&](mrl_ctx *c, int *id) {
                                    return load_through_image_cache(c, image, [&](mrl_ctx *cc, int *i) { return mrl_material_load_merl(cc, path.c_str(), i); }, id);
                                }, "mrl_material_load_merl");
        return Material(ctx, res);
    }
    // param: enum mrl_param — which three angles index the table.  The option is per upload; the loader runs under the
    // context's lock, so setting it around the load cannot leak into another instance's upload.
    static Material load_table(const ContextKey &key, const std::string &path, const double scale[3], int param = MRL_PARAM_HALF_DIFF)
    {
        auto ctx = Context::get(key);
        char sc[160];
        std::snprintf(sc, sizeof sc, "|%.17g|%.17g|%.17g|p%d", scale[0], scale[1], scale[2], param);
        const std::string image = image_cache_name(canonical_path(path), std::string("table") + sc, key);
        auto res = ctx->acquire("table|" + canonical_path(path) + sc,
                                [&](mrl_ctx *c, int *id) {
                                    return load_through_image_cache(c, image, [&](mrl_ctx *cc, int *i) {
                                        int rc = mrl_set_option(cc, MRL_OPT_TABLE_PARAM, param);
                                        if (rc == MRL_OK) rc = mrl_material_load_table(cc, path.c_str(), scale, i);
                                        (void)mrl_set_option(cc, MRL_OPT_TABLE_PARAM, MRL_PARAM_HALF_DIFF);
                                        return rc;
                                    }, id);
                                }, "mrl_material_load_table");
        return Material(ctx, res);
    }

    // a customized_measurement table stored in a tensor_file container (the RGL *.bsdf container): field "table"
    // [3, n_theta_h, n_theta_d, n_phi_d] (+ optional "scale").  The renderers' RGB builds take three channels; wider
    // tables are reachable through the C ABI's *_nch calls, not through a Spectrum-returning plugin.
    static Material load_tensor_table(const ContextKey &key, const std::string &path, int param = MRL_PARAM_HALF_DIFF)
    {
        auto ctx = Context::get(key);
        auto res = ctx->acquire("tensor|" + canonical_path(path) + "|p" + std::to_string(param),
                                [&](mrl_ctx *c, int *id) {
                                    int channels = 0;
                                    int rc = mrl_set_option(c, MRL_OPT_TABLE_PARAM, param);
                                    if (rc == MRL_OK) rc = mrl_material_load_tensor_table(c, path.c_str(), nullptr, id, &channels);
                                    (void)mrl_set_option(c, MRL_OPT_TABLE_PARAM, MRL_PARAM_HALF_DIFF);
                                    if (rc == MRL_OK && channels != 3) {
                                        mrl_material_release(c, *id);
                                        throw Error(MRL_ERR_FORMAT, path + ": the table has " + std::to_string(channels) +
                                                                    " channels; this RGB build of the plugin evaluates three (use the *_nch entry points of the C ABI)");
                                    }
                                    if (rc != MRL_OK && rc != MRL_ERR_HIP && rc != MRL_ERR_OOM)
                                        throw Error(rc, std::string("mrl_material_load_tensor_table: ") + mrl_tensor_file_last_error(nullptr));
                                    return rc;
                                }, "mrl_material_load_tensor_table");
        return Material(ctx, res);
    }
    // the adaptive-parameterisation measured BSDF of an RGL *.bsdf file (what upstream Mitsuba 3's `measured` evaluates)
    static Material load_rgl(const ContextKey &key, const std::string &path)
    {
        auto ctx = Context::get(key);
        auto res = ctx->acquire("rgl|" + canonical_path(path),
                                [&](mrl_ctx *c, int *id) {
                                    const int rc = mrl_material_load_rgl(c, path.c_str(), id);
                                    if (rc != MRL_OK && rc != MRL_ERR_HIP && rc != MRL_ERR_OOM) {
                                        const char *why = mrl_tensor_file_last_error(nullptr);
                                        if (why && *why) throw Error(rc, std::string("mrl_material_load_rgl: ") + why);
                                    }
                                    return rc;
                                }, "mrl_material_load_rgl");
        return Material(ctx, res);
    }
    // is the resident material a spectral RGL file (MRL_KIND_RGL_SPECTRAL)?
    bool spectral() const
    {
        int kind = -1, dims[3];
        std::lock_guard<std::mutex> lock(m_ctx->mutex());
        check(ctx(), mrl_material_info(ctx(), m_id, &kind, dims), "mrl_material_info");
        return kind == MRL_KIND_RGL_SPECTRAL;
    }
    static bool is_tensor_file(const std::string &path)
    {
        return path.size() > 5 && path.compare(path.size() - 5, 5, ".bsdf") == 0;
    }

    // scalar = "cpu": one-unit calls evaluate on the calling thread from now on (idempotent)
    void use_cpu_scalar() { m_host = m_res->host_table(); }
    bool cpu_scalar() const { return m_host != nullptr; }
    bool valid() const { return m_ctx && m_id >= 0; }
    int id() const { return m_id; }
    mrl_ctx *ctx() const { return m_ctx->raw(); }

    // ---- scalar calls (the virtual per-ray BSDF::eval / sample / pdf): one half of the unit each, through the scalar service ----
    // An integrator that weighs light samples asks eval(wi, wo) and then pdf(wi, wo) for the SAME pair (MIS): the device
    // returns both at once, so the second question is answered from a one-entry, per-thread memo instead of a second trip.
    void eval1(const float wi[3], const float wo[3], float rgb[3]) const
    {
        float pdf;
        eval_pdf1(wi, wo, rgb, pdf);
    }
    float pdf1(const float wi[3], const float wo[3]) const
    {
        float rgb[3], pdf;
        eval_pdf1(wi, wo, rgb, pdf);
        return pdf;
    }
    // eval + pdf of the same pair (Mitsuba 3's eval_pdf)
    void eval_pdf1(const float wi[3], const float wo[3], float rgb[3], float &pdf) const
    {
        Memo &m = memo();
        if (m.table != m_res->serial || std::memcmp(m.wi, wi, 12) != 0 || std::memcmp(m.wo, wo, 12) != 0) {
            m.table = 0;                                             // not valid while it is being refilled (an exception leaves it so)
            if (m_host) check(nullptr, mrl_host_eval_pdf(m_host, wi, wo, m.rgb, &m.pdf), "mrl_host_eval_pdf");
            else m_ctx->scalar_eval_pdf(m_id, wi, wo, m.rgb, m.pdf);
            std::memcpy(m.wi, wi, 12); std::memcpy(m.wo, wo, 12);
            m.table = m_res->serial;
        }
        std::memcpy(rgb, m.rgb, 12);
        pdf = m.pdf;
    }
    void sample1(const float wi[3], const float u[2], float wo[3], float &pdf, float weight[3]) const
    {
        if (m_host) check(nullptr, mrl_host_sample(m_host, wi, u, wo, &pdf, weight), "mrl_host_sample");
        else m_ctx->scalar_sample(m_id, wi, u, wo, pdf, weight);
    }
    // ---- spectral materials: W values at the ray's wavelengths; one-unit calls on the calling thread over the host image ----
    void eval_pdf_spectral1(const float wi[3], const float wo[3], const float *wl, int W, float *values, float &pdf) const
    {
        if (!m_host) throw Error(MRL_ERR_INVALID, "spectral one-unit calls evaluate on the CPU (scalar = \"cpu\")");
        check(nullptr, mrl_host_eval_pdf_spectral(m_host, wi, wo, wl, W, values, &pdf), "mrl_host_eval_pdf_spectral");
    }
    void sample_spectral1(const float wi[3], const float u[2], const float *wl, int W, float wo[3], float &pdf, float *weight) const
    {
        if (!m_host) throw Error(MRL_ERR_INVALID, "spectral one-unit calls evaluate on the CPU (scalar = \"cpu\")");
        check(nullptr, mrl_host_sample_spectral(m_host, wi, u, wl, W, wo, &pdf, weight), "mrl_host_sample_spectral");
    }
    void eval_sample_spectral_batch(const float *wi, const float *wo, const float *u, const float *wl, int W, size_t n,
                                    float *values, float *pdf, float *wo2, float *pdf2, float *weight) const
    {
        std::lock_guard<std::mutex> lock(m_ctx->mutex());
        check(ctx(), mrl_eval_sample_spectral_batch(ctx(), wi, wo, u, wl, W, m_id, n, values, pdf, wo2, pdf2, weight), "mrl_eval_sample_spectral_batch");
    }
    // ---- batch / wavefront calls: host or device arrays, n units (see include/merl_hip.h) ----
    void eval_batch(const float *wi, const float *wo, size_t n, float *rgb) const
    {
        std::lock_guard<std::mutex> lock(m_ctx->mutex());
        check(ctx(), mrl_eval_batch(ctx(), wi, wo, nullptr, m_id, n, rgb), "mrl_eval_batch");
    }
    void pdf_batch(const float *wi, const float *wo, size_t n, float *pdf) const
    {
        std::lock_guard<std::mutex> lock(m_ctx->mutex());
        check(ctx(), mrl_pdf_batch(ctx(), wi, wo, nullptr, m_id, n, pdf), "mrl_pdf_batch");
    }
    void eval_pdf_batch(const float *wi, const float *wo, size_t n, float *rgb, float *pdf) const
    {
        std::lock_guard<std::mutex> lock(m_ctx->mutex());
        check(ctx(), mrl_eval_pdf_batch(ctx(), wi, wo, nullptr, m_id, n, rgb, pdf), "mrl_eval_pdf_batch");
    }
    void sample_batch(const float *wi, const float *u, size_t n, float *wo, float *pdf, float *weight) const
    {
        std::lock_guard<std::mutex> lock(m_ctx->mutex());
        check(ctx(), mrl_sample_batch(ctx(), wi, u, nullptr, m_id, n, wo, pdf, weight), "mrl_sample_batch");
    }
    void eval_sample_batch(const float *wi, const float *wo, const float *u, size_t n,
                           float *rgb, float *pdf, float *wo2, float *pdf2, float *weight) const
    {
        std::lock_guard<std::mutex> lock(m_ctx->mutex());
        check(ctx(), mrl_eval_sample_batch(ctx(), wi, wo, u, nullptr, m_id, n, rgb, pdf, wo2, pdf2, weight), "mrl_eval_sample_batch");
    }
    void eval_sample_queue(const float *wi, const float *wo, const float *u,
                           const uint32_t *queue, const uint32_t *count, size_t capacity,
                           float *rgb, float *pdf, float *wo2, float *pdf2, float *weight) const
    {
        std::lock_guard<std::mutex> lock(m_ctx->mutex());
        check(ctx(), mrl_eval_sample_queue(ctx(), wi, wo, u, nullptr, m_id, queue, count, capacity, rgb, pdf, wo2, pdf2, weight),
              "mrl_eval_sample_queue");
    }
    void synchronize() const
    {
        std::lock_guard<std::mutex> lock(m_ctx->mutex());
        check(ctx(), mrl_synchronize(ctx()), "mrl_synchronize");
    }

private:
    Material(std::shared_ptr<Context> ctx, std::shared_ptr<Context::Resident> res)
        : m_ctx(std::move(ctx)), m_res(std::move(res)), m_id(m_res->id) {}

    // the last (table, wi, wo) -> (rgb, pdf) this thread asked for; the values are a pure function of the key
    struct Memo {
        unsigned long long table = 0;                   // Resident::serial, 0 = empty
        float wi[3], wo[3], rgb[3], pdf;
    };
    static Memo &memo()
    {
        thread_local Memo m;
        return m;
    }

    // declaration order = reverse destruction order: the resident handle (which calls into the context) dies first
    std::shared_ptr<Context> m_ctx;
    std::shared_ptr<Context::Resident> m_res;
    int m_id = -1;
    const mrl_host_table *m_host = nullptr;            // owned by m_res; non-null = one-unit calls run on the CPU
};

inline int parse_lookup(const std::string &s)
{
    if (s == "nearest") return 0;
    if (s == "trilinear") return 1;
    throw Error(MRL_ERR_INVALID, "interpolation must be \"nearest\" or \"trilinear\", got \"" + s + "\"");
}
inline int parse_sampling(const std::string &s)
{
    if (s == "cosine") return 0;
    if (s == "table") return 1;
    if (s == "table2d") return 2;          // the conditional table P(theta_h | theta_i): include/merl_hip.h, mrl_material_sampling2d
    throw Error(MRL_ERR_INVALID, "sampling must be \"cosine\", \"table\" or \"table2d\", got \"" + s + "\"");
}
inline int parse_parameterization(const std::string &s)
{
    if (s == "half_diff" || s == "merl") return MRL_PARAM_HALF_DIFF;
    if (s == "standard") return MRL_PARAM_STANDARD;
    if (s == "standard_full") return MRL_PARAM_STANDARD_FULL;
    throw Error(MRL_ERR_INVALID, "parameterization must be \"half_diff\", \"standard\" or \"standard_full\", got \"" + s + "\"");
}
inline bool parse_scalar_cpu(const std::string &s)
{
    if (s == "cpu") return true;
    if (s == "gpu") return false;
    throw Error(MRL_ERR_INVALID, "scalar must be \"cpu\" or \"gpu\", got \"" + s + "\"");
}
// `cosine_factor` — does eval() include cos(theta_o)?
inline int parse_cosine_factor(const std::string &s)
{
    if (s == "included" || s == "true") return 0;
    if (s == "omitted" || s == "false") return 1;
    throw Error(MRL_ERR_INVALID, "cosine_factor must be \"included\" or \"omitted\", got \"" + s + "\"");
}
// `negative_values` — what a negative stored value (a sample that was not measured) does to a lookup
inline int parse_negative_values(const std::string &s)
{
    if (s == "clamp") return MRL_NEGATIVE_CLAMP;
    if (s == "keep") return MRL_NEGATIVE_KEEP;
    if (s == "renormalize" || s == "renormalise") return MRL_NEGATIVE_RENORMALISE;
    throw Error(MRL_ERR_INVALID, "negative_values must be \"clamp\", \"keep\" or \"renormalize\", got \"" + s + "\"");
}
inline int parse_node(const std::string &s)
{
    if (s == "integer") return 0;
    if (s == "center" || s == "centre") return 1;
    throw Error(MRL_ERR_INVALID, "node must be \"integer\" or \"center\", got \"" + s + "\"");
}

} // namespace merl_gpu
