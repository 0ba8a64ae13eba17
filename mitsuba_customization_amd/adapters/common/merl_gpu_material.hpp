// merl_gpu_material.hpp — what both plugin adapters (Mitsuba 0.6 and Mitsuba 3) share: one
// libmerl_hip context per GPU and process, a material handle, and the 1-unit "scalar call"
// plumbing.  Host C++ only; everything that computes goes through the C ABI (include/merl_hip.h).
//
// Threading: the renderers call eval()/sample()/pdf() on a const BSDF from all render threads
// (SURVEY.md §8b).  A libmerl_hip context is thread-compatible, so scalar calls serialise on the
// context's mutex.  A scalar call is a 1-unit GPU batch on pinned, device-mapped memory (zero
// copy, ~tens of microseconds): it is plumbing for existing integrators, not the fast path.  The
// fast path is the batch / wavefront entry points, which take whole arrays.
#pragma once
#include <cstddef>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>

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
    bool operator<(const ContextKey &o) const
    {
        if (device != o.device) return device < o.device;
        if (lookup != o.lookup) return lookup < o.lookup;
        if (node != o.node) return node < o.node;
        if (disk_map != o.disk_map) return disk_map < o.disk_map;
        return sampling < o.sampling;
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
        // scalar-call staging: wi[3] wo[3] u[2] | rgb[3] pdf wo2[3] pdf2 weight[3]
        void *p = nullptr;
        check(m_ctx, mrl_host_alloc(m_ctx, 32 * sizeof(float), &p), "mrl_host_alloc");
        m_pin = static_cast<float *>(p);
    }
    ~Context()
    {
        if (m_ctx) { mrl_host_free(m_ctx, m_pin); mrl_destroy(m_ctx); }
    }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;

    mrl_ctx *raw() const { return m_ctx; }
    std::mutex &mutex() { return m_mutex; }
    float *pinned() { return m_pin; }

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
    ContextKey m_key;
    mrl_ctx *m_ctx = nullptr;
    std::mutex m_mutex;
    float *m_pin = nullptr;
};

// A material living on the GPU + the calls the plugin classes forward to.
class Material {
public:
    Material() = default;
    Material(std::shared_ptr<Context> ctx, int id) : m_ctx(std::move(ctx)), m_id(id) {}

    static Material load_merl(const ContextKey &key, const std::string &path)
    {
        auto ctx = Context::get(key);
        std::lock_guard<std::mutex> lock(ctx->mutex());
        int id = -1;
        check(ctx->raw(), mrl_material_load_merl(ctx->raw(), path.c_str(), &id), "mrl_material_load_merl");
        return Material(ctx, id);
    }
    static Material load_table(const ContextKey &key, const std::string &path, const double scale[3])
    {
        auto ctx = Context::get(key);
        std::lock_guard<std::mutex> lock(ctx->mutex());
        int id = -1;
        check(ctx->raw(), mrl_material_load_table(ctx->raw(), path.c_str(), scale, &id), "mrl_material_load_table");
        return Material(ctx, id);
    }

    bool valid() const { return m_ctx && m_id >= 0; }
    int id() const { return m_id; }
    mrl_ctx *ctx() const { return m_ctx->raw(); }

    // ---- scalar calls: one unit through the GPU ----
    void eval1(const float wi[3], const float wo[3], float rgb[3]) const
    {
        std::lock_guard<std::mutex> lock(m_ctx->mutex());
        float *p = m_ctx->pinned();
        for (int k = 0; k < 3; ++k) { p[k] = wi[k]; p[3 + k] = wo[k]; }
        check(ctx(), mrl_eval_batch(ctx(), p, p + 3, nullptr, m_id, 1, p + 8), "mrl_eval_batch");
        check(ctx(), mrl_synchronize(ctx()), "mrl_synchronize");
        for (int k = 0; k < 3; ++k) rgb[k] = p[8 + k];
    }
    float pdf1(const float wi[3], const float wo[3]) const
    {
        std::lock_guard<std::mutex> lock(m_ctx->mutex());
        float *p = m_ctx->pinned();
        for (int k = 0; k < 3; ++k) { p[k] = wi[k]; p[3 + k] = wo[k]; }
        check(ctx(), mrl_pdf_batch(ctx(), p, p + 3, nullptr, m_id, 1, p + 11), "mrl_pdf_batch");
        check(ctx(), mrl_synchronize(ctx()), "mrl_synchronize");
        return p[11];
    }
    void sample1(const float wi[3], const float u[2], float wo[3], float &pdf, float weight[3]) const
    {
        std::lock_guard<std::mutex> lock(m_ctx->mutex());
        float *p = m_ctx->pinned();
        for (int k = 0; k < 3; ++k) p[k] = wi[k];
        p[6] = u[0]; p[7] = u[1];
        check(ctx(), mrl_sample_batch(ctx(), p, p + 6, nullptr, m_id, 1, p + 12, p + 15, p + 16), "mrl_sample_batch");
        check(ctx(), mrl_synchronize(ctx()), "mrl_synchronize");
        for (int k = 0; k < 3; ++k) { wo[k] = p[12 + k]; weight[k] = p[16 + k]; }
        pdf = p[15];
    }
    // eval + pdf of the same pair in one launch (Mitsuba 3's eval_pdf)
    void eval_pdf1(const float wi[3], const float wo[3], float rgb[3], float &pdf) const
    {
        std::lock_guard<std::mutex> lock(m_ctx->mutex());
        float *p = m_ctx->pinned();
        for (int k = 0; k < 3; ++k) { p[k] = wi[k]; p[3 + k] = wo[k]; }
        check(ctx(), mrl_eval_pdf_batch(ctx(), p, p + 3, nullptr, m_id, 1, p + 8, p + 11), "mrl_eval_pdf_batch");
        check(ctx(), mrl_synchronize(ctx()), "mrl_synchronize");
        for (int k = 0; k < 3; ++k) rgb[k] = p[8 + k];
        pdf = p[11];
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
    std::shared_ptr<Context> m_ctx;
    int m_id = -1;
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
    throw Error(MRL_ERR_INVALID, "sampling must be \"cosine\" or \"table\", got \"" + s + "\"");
}
inline int parse_node(const std::string &s)
{
    if (s == "integer") return 0;
    if (s == "center" || s == "centre") return 1;
    throw Error(MRL_ERR_INVALID, "node must be \"integer\" or \"center\", got \"" + s + "\"");
}

} // namespace merl_gpu
