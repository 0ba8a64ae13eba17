// combiner_tsan.cpp — ThreadSanitizer run of the adapters' scalar-call combiner (Context::submit) on the CPU.
// The C ABI is replaced by a test double defined in this file (a deterministic arithmetic function of the
// inputs, with a short sleep standing in for the GPU round trip), so only the host-side protocol is under
// test: no request lost, none served twice, every thread gets ITS result, errors reach every waiter.
// Built and run by tests/test_sanitize_cpu.py with -fsanitize=thread.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../mitsuba_customization_amd/adapters/common/merl_gpu_material.hpp"

struct mrl_ctx { int materials = 0; std::atomic<int> in_call{ 0 }; std::atomic<long> rounds{ 0 }, units{ 0 }; bool fail_next = false; };
static mrl_ctx g_ctx;

extern "C" {
const char *mrl_strerror(int) { return "test double"; }
const char *mrl_last_error(const mrl_ctx *) { return "injected failure"; }
int mrl_init(int, mrl_ctx **out) { *out = &g_ctx; return MRL_OK; }
int mrl_destroy(mrl_ctx *) { return MRL_OK; }
int mrl_set_option(mrl_ctx *, int, int) { return MRL_OK; }
int mrl_host_alloc(mrl_ctx *, size_t bytes, void **out) { *out = std::malloc(bytes); return *out ? MRL_OK : MRL_ERR_OOM; }
int mrl_host_free(mrl_ctx *, void *p) { std::free(p); return MRL_OK; }
int mrl_material_load_merl(mrl_ctx *c, const char *, int *id) { *id = c->materials++; return MRL_OK; }
int mrl_material_load_table(mrl_ctx *c, const char *, const double *, int *id) { *id = c->materials++; return MRL_OK; }
int mrl_material_release(mrl_ctx *, int) { return MRL_OK; }
int mrl_material_load_tensor_table(mrl_ctx *c, const char *, const char *, int *id, int *ch) { *id = c->materials++; *ch = 3; return MRL_OK; }
const char *mrl_tensor_file_last_error(const mrl_tensor_file *) { return ""; }
int mrl_synchronize(mrl_ctx *) { return MRL_OK; }
int mrl_eval_sample_batch(mrl_ctx *c, const float *wi, const float *wo, const float *u, const int32_t *mat, int32_t, size_t n,
                          float *rgb, float *pdf, float *wo2, float *pdf2, float *w)
{
    if (c->in_call.fetch_add(1) != 0) { std::fprintf(stderr, "two callers inside the thread-compatible context\n"); std::abort(); }
    std::this_thread::sleep_for(std::chrono::microseconds(15));
    int rc = MRL_OK;
    if (c->fail_next) { c->fail_next = false; rc = MRL_ERR_HIP; }
    else
        for (size_t i = 0; i < n; ++i) {
            for (int k = 0; k < 3; ++k) { rgb[3 * i + k] = wi[3 * i + k] + 2.0f * wo[3 * i + k] + (float)mat[i]; wo2[3 * i + k] = wi[3 * i + k] - u[2 * i]; w[3 * i + k] = u[2 * i + 1]; }
            pdf[i] = wo[3 * i + 2]; pdf2[i] = u[2 * i] + u[2 * i + 1];
        }
    c->rounds++; c->units += (long)n;
    c->in_call.fetch_sub(1);
    return rc;
}
int mrl_eval_batch(mrl_ctx *, const float *, const float *, const int32_t *, int32_t, size_t, float *) { return MRL_OK; }
int mrl_pdf_batch(mrl_ctx *, const float *, const float *, const int32_t *, int32_t, size_t, float *) { return MRL_OK; }
int mrl_eval_pdf_batch(mrl_ctx *, const float *, const float *, const int32_t *, int32_t, size_t, float *, float *) { return MRL_OK; }
int mrl_sample_batch(mrl_ctx *, const float *, const float *, const int32_t *, int32_t, size_t, float *, float *, float *) { return MRL_OK; }
int mrl_eval_sample_queue(mrl_ctx *, const float *, const float *, const float *, const int32_t *, int32_t, const uint32_t *, const uint32_t *,
                          size_t, float *, float *, float *, float *, float *) { return MRL_OK; }
}

int main()
{
    using namespace merl_gpu;
    const ContextKey key{ 0, 1, 0, 0, 0 };
    Material a = Material::load_merl(key, "a"), b = Material::load_merl(key, "b");
    const unsigned T = 12, per_thread = 1500;
    std::atomic<long> bad{ 0 };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < T; ++t)
        pool.emplace_back([&, t]() {
            const Material &m = (t & 1) ? b : a;
            for (unsigned i = 0; i < per_thread; ++i) {
                const float wi[3] = { (float)t, (float)i, 1.0f }, wo[3] = { 0.5f * t, 0.25f * i, 2.0f }, u[2] = { 0.001f * i, 0.01f * t };
                float rgb[3], wo2[3], w[3], pdf;
                m.eval1(wi, wo, rgb);
                for (int k = 0; k < 3; ++k) if (rgb[k] != wi[k] + 2.0f * wo[k] + (float)m.id()) bad++;
                if (m.pdf1(wi, wo) != wo[2]) bad++;
                m.sample1(wi, u, wo2, pdf, w);
                if (wo2[0] != wi[0] - u[0] || pdf != u[0] + u[1] || w[2] != u[1]) bad++;
            }
        });
    for (auto &th : pool) th.join();
    const long calls = 3L * T * per_thread;
    std::printf("calls %ld rounds %ld units %ld wrong %ld\n", calls, g_ctx.rounds.load(), g_ctx.units.load(), bad.load());
    if (bad.load() != 0 || g_ctx.units.load() != calls || g_ctx.rounds.load() >= calls) return 1;   // every call served once, and rounds were shared

    // an error inside a round reaches the caller as an exception, and the combiner keeps working afterwards
    g_ctx.fail_next = true;
    const float wi[3] = { 0, 0, 1 }, wo[3] = { 0, 0, 1 };
    float rgb[3];
    bool threw = false;
    try { a.eval1(wi, wo, rgb); } catch (const Error &e) { threw = e.status == MRL_ERR_HIP && std::strstr(e.what(), "injected failure"); }
    a.eval1(wi, wo, rgb);
    if (!threw || rgb[2] != 3.0f) return 2;
    std::puts("combiner ok");
    return 0;
}
