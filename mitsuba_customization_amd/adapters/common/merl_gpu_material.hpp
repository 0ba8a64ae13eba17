// merl_gpu_material.hpp — what both plugin adapters (Mitsuba 0.6 and Mitsuba 3) share: one
// libmerl_hip context per GPU and process, a material handle, and the 1-unit "scalar call"
// plumbing.  Host C++ only; everything that computes goes through the C ABI (include/merl_hip.h).
//
// Threading: the renderers call eval()/sample()/pdf() on a const BSDF from all render threads
// (SURVEY.md §8b).  A libmerl_hip context serialises its callers (one GPU round trip each), so scalar calls are COMBINED:
// a calling thread posts its request; whichever thread finds no round in flight becomes the
// leader, takes every request posted so far (its own and other threads', up to 256), runs them
// as ONE fused eval+sample batch on pinned, device-mapped memory (zero copy), and hands the
// results back.  One thread alone pays the launch + sync latency (~16 us) per call; T render
// threads share it, so throughput grows with T instead of serialising on a mutex.  Scalar calls
// remain plumbing for existing integrators; the fast path is the batch / wavefront entry
// points, which take whole arrays.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

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

// One scalar plugin call: inputs, the material it addresses, and every output of the fused unit.
struct ScalarRequest {
    float wi[3] = { 0.0f, 0.0f, 1.0f }, wo[3] = { 0.0f, 0.0f, 1.0f }, u[2] = { 0.5f, 0.5f };
    int material = 0;
    float rgb[3], pdf, wo2[3], pdf2, weight[3];
    std::atomic<bool> done{ false };        // set by the round's leader, last thing it does with the request
    int status = MRL_OK;
    std::string error;
};

class Context {
public:
    static constexpr size_t kSlots = 256;          // requests per combined round
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
        // scalar-call staging for kSlots requests: wi[3K] wo[3K] u[2K] mat[K] | rgb[3K] pdf[K] wo2[3K] pdf2[K] weight[3K]
        void *p = nullptr;
        check(m_ctx, mrl_host_alloc(m_ctx, 20 * kSlots * sizeof(float), &p), "mrl_host_alloc");
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

    // ---- table residency: one upload per (file, channel scales) and context, shared by every plugin instance that
    // names it (two <bsdf> elements with the same .binary, unserialised copies on a worker, scene reloads); the
    // table leaves HBM when the last instance dies (mrl_material_release).
    struct Resident {
        Context *owner;                                 // kept alive by the Material's shared_ptr<Context>
        std::string key;
        int id;
        ~Resident()
        {
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
        auto sp = std::shared_ptr<Resident>(new Resident{ this, key, id });
        m_resident[key] = sp;
        return sp;
    }
    size_t resident_tables()
    {
        std::lock_guard<std::mutex> lock(m_mutex);
        return m_resident.size();
    }

    // Post one scalar request and return when its outputs are filled in (throws what the round's call reported).
    // A GPU round lasts ~16 us, so a waiting thread spins for about that long before it sleeps on the condition
    // variable (a render thread has nothing else to do until its BSDF value arrives).
    void submit(ScalarRequest &r)
    {
        {
            std::lock_guard<std::mutex> lk(m_post_mutex);
            m_posted.push_back(&r);
        }
        for (unsigned spins = 0; !r.done.load(std::memory_order_acquire); ++spins) {
            if (!m_round_in_flight.load(std::memory_order_relaxed)) {
                std::unique_lock<std::mutex> lk(m_post_mutex);
                if (m_round_in_flight.load(std::memory_order_relaxed) || r.done.load(std::memory_order_acquire)) continue;
                // lead a round over everything posted so far (this thread's request is among it or already served)
                m_round_in_flight.store(true, std::memory_order_relaxed);
                const size_t k = m_posted.size() < kSlots ? m_posted.size() : kSlots;
                std::vector<ScalarRequest *> round(m_posted.begin(), m_posted.begin() + (std::ptrdiff_t)k);
                m_posted.erase(m_posted.begin(), m_posted.begin() + (std::ptrdiff_t)k);
                lk.unlock();
                run_round(round);                       // ends by publishing done on every request: they are gone after that
                lk.lock();
                m_round_in_flight.store(false, std::memory_order_relaxed);
                m_round_done.notify_all();
                continue;
            }
            if (spins < kSpinsBeforeSleep) { relax(); continue; }
            std::unique_lock<std::mutex> lk(m_post_mutex);
            if (m_round_in_flight.load(std::memory_order_relaxed) && !r.done.load(std::memory_order_acquire)) m_round_done.wait(lk);
        }
        if (r.status != MRL_OK) throw Error(r.status, r.error);
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
    // one fused eval+sample launch over the round's requests, per-request material ids
    void run_round(const std::vector<ScalarRequest *> &round) noexcept
    {
        const size_t k = round.size(), K = kSlots;
        float *wi = m_pin, *wo = wi + 3 * K, *u = wo + 3 * K;
        int32_t *mat = reinterpret_cast<int32_t *>(u + 2 * K);
        float *rgb = u + 2 * K + K, *pdf = rgb + 3 * K, *wo2 = pdf + K, *pdf2 = wo2 + 3 * K, *weight = pdf2 + K;
        for (size_t i = 0; i < k; ++i) {
            const ScalarRequest &q = *round[i];
            for (int c = 0; c < 3; ++c) { wi[3 * i + c] = q.wi[c]; wo[3 * i + c] = q.wo[c]; }
            u[2 * i] = q.u[0]; u[2 * i + 1] = q.u[1];
            mat[i] = q.material;
        }
        int rc;
        std::string what;
        {
            std::lock_guard<std::mutex> call(m_mutex);      // one call sequence (launch + sync) at a time
            rc = mrl_eval_sample_batch(m_ctx, wi, wo, u, mat, 0, k, rgb, pdf, wo2, pdf2, weight);
            if (rc == MRL_OK) rc = mrl_synchronize(m_ctx);
            if (rc != MRL_OK) {
                what = std::string("scalar call: ") + mrl_strerror(rc);
                const char *d = mrl_last_error(m_ctx);
                if (d && *d) what += std::string(" (") + d + ")";
            }
        }
        for (size_t i = 0; i < k; ++i) {
            ScalarRequest &q = *round[i];
            q.status = rc;
            if (rc != MRL_OK) {
                q.error = what;
            } else {
                for (int c = 0; c < 3; ++c) { q.rgb[c] = rgb[3 * i + c]; q.wo2[c] = wo2[3 * i + c]; q.weight[c] = weight[3 * i + c]; }
                q.pdf = pdf[i]; q.pdf2 = pdf2[i];
            }
            q.done.store(true, std::memory_order_release);   // the owner may return and destroy q from here on
        }
    }

    static void relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    static constexpr unsigned kSpinsBeforeSleep = 4000;      // ~40 ns per pause: a few GPU rounds

    ContextKey m_key;
    mrl_ctx *m_ctx = nullptr;
    std::mutex m_mutex;                    // keeps a launch and its synchronize together (the C context only locks per call)
    std::map<std::string, std::weak_ptr<Resident>> m_resident;     // guarded by m_mutex
    float *m_pin = nullptr;
    std::mutex m_post_mutex;               // guards the three members below
    std::condition_variable m_round_done;
    std::vector<ScalarRequest *> m_posted;
    std::atomic<bool> m_round_in_flight{ false };
};

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
        auto res = ctx->acquire("merl|" + canonical_path(path),
                                [&](mrl_ctx *c, int *id) { return mrl_material_load_merl(c, path.c_str(), id); }, "mrl_material_load_merl");
        return Material(ctx, res);
    }
    // param: enum mrl_param — which three angles index the table.  The option is per upload; the loader runs under the
    // context's lock, so setting it around the load cannot leak into another instance's upload.
    static Material load_table(const ContextKey &key, const std::string &path, const double scale[3], int param = MRL_PARAM_HALF_DIFF)
    {
        auto ctx = Context::get(key);
        char sc[160];
        std::snprintf(sc, sizeof sc, "|%.17g|%.17g|%.17g|p%d", scale[0], scale[1], scale[2], param);
        auto res = ctx->acquire("table|" + canonical_path(path) + sc,
                                [&](mrl_ctx *c, int *id) {
                                    int rc = mrl_set_option(c, MRL_OPT_TABLE_PARAM, param);
                                    if (rc == MRL_OK) rc = mrl_material_load_table(c, path.c_str(), scale, id);
                                    (void)mrl_set_option(c, MRL_OPT_TABLE_PARAM, MRL_PARAM_HALF_DIFF);
                                    return rc;
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
    static bool is_tensor_file(const std::string &path)
    {
        return path.size() > 5 && path.compare(path.size() - 5, 5, ".bsdf") == 0;
    }

    bool valid() const { return m_ctx && m_id >= 0; }
    int id() const { return m_id; }
    mrl_ctx *ctx() const { return m_ctx->raw(); }

    // ---- scalar calls: combined with the other render threads' calls into one GPU round ----
    void eval1(const float wi[3], const float wo[3], float rgb[3]) const
    {
        ScalarRequest r;
        fill(r, wi, wo, nullptr);
        m_ctx->submit(r);
        for (int k = 0; k < 3; ++k) rgb[k] = r.rgb[k];
    }
    float pdf1(const float wi[3], const float wo[3]) const
    {
        ScalarRequest r;
        fill(r, wi, wo, nullptr);
        m_ctx->submit(r);
        return r.pdf;
    }
    void sample1(const float wi[3], const float u[2], float wo[3], float &pdf, float weight[3]) const
    {
        ScalarRequest r;
        fill(r, wi, nullptr, u);
        m_ctx->submit(r);
        for (int k = 0; k < 3; ++k) { wo[k] = r.wo2[k]; weight[k] = r.weight[k]; }
        pdf = r.pdf2;
    }
    // eval + pdf of the same pair (Mitsuba 3's eval_pdf)
    void eval_pdf1(const float wi[3], const float wo[3], float rgb[3], float &pdf) const
    {
        ScalarRequest r;
        fill(r, wi, wo, nullptr);
        m_ctx->submit(r);
        for (int k = 0; k < 3; ++k) rgb[k] = r.rgb[k];
        pdf = r.pdf;
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
    void fill(ScalarRequest &r, const float wi[3], const float *wo, const float *u) const
    {
        r.material = m_id;
        for (int k = 0; k < 3; ++k) { r.wi[k] = wi[k]; if (wo) r.wo[k] = wo[k]; }
        if (u) { r.u[0] = u[0]; r.u[1] = u[1]; }
    }

    Material(std::shared_ptr<Context> ctx, std::shared_ptr<Context::Resident> res)
        : m_ctx(std::move(ctx)), m_res(std::move(res)), m_id(m_res->id) {}

    // declaration order = reverse destruction order: the resident handle (which calls into the context) dies first
    std::shared_ptr<Context> m_ctx;
    std::shared_ptr<Context::Resident> m_res;
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
inline int parse_parameterization(const std::string &s)
{
    if (s == "half_diff" || s == "merl") return MRL_PARAM_HALF_DIFF;
    if (s == "standard") return MRL_PARAM_STANDARD;
    if (s == "standard_full") return MRL_PARAM_STANDARD_FULL;
    throw Error(MRL_ERR_INVALID, "parameterization must be \"half_diff\", \"standard\" or \"standard_full\", got \"" + s + "\"");
}
inline int parse_node(const std::string &s)
{
    if (s == "integer") return 0;
    if (s == "center" || s == "centre") return 1;
    throw Error(MRL_ERR_INVALID, "node must be \"integer\" or \"center\", got \"" + s + "\"");
}

} // namespace merl_gpu
