// scalar_host.cpp — a C++ host that makes ONE-unit calls from many threads, the way the render threads of a stock per-ray
// integrator call a BSDF plugin:  mrl_scalar_eval_sample(ctx, material, wi, wo, u, out[11]).
//
//   scalar_host [--threads T] [--calls K] [--table file.binary] [--rgl file_rgb.bsdf] [--churn]
//
// It (1) checks every answer bit for bit against mrl_eval_sample_batch on the same inputs, (2) times the calls — one
// thread alone (the latency of a call) and T threads together (what a render sees) —, (3) with --churn lets another
// thread upload and release tables and flip an option meanwhile (the calls must keep returning the right numbers: the
// service pauses around every change).  Prints one JSON line; exit code 0 = all answers right.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "merl_hip.h"

static uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static float uni(uint64_t &s) { s = mix(s); return (float)(s >> 40) * (1.0f / 16777216.0f); }
static void hemi(uint64_t &s, float v[3])
{
    const float z = 0.02f + 0.98f * uni(s), ph = 6.2831853f * uni(s), r = std::sqrt(1.0f - z * z);
    v[0] = r * std::cos(ph); v[1] = r * std::sin(ph); v[2] = z;
}

#define CHECK(expr)                                                                                          \
    do { const int rc_ = (expr); if (rc_ != MRL_OK) { std::fprintf(stderr, "%s: %s (%s)\n", #expr, mrl_strerror(rc_), mrl_last_error(ctx)); return 2; } } while (0)

int main(int argc, char **argv)
{
    int threads = 16; long calls = 20000; bool churn = false; const char *table = nullptr, *rgl_file = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--threads") && i + 1 < argc) threads = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--calls") && i + 1 < argc) calls = std::atol(argv[++i]);
        else if (!std::strcmp(argv[i], "--table") && i + 1 < argc) table = argv[++i];
        else if (!std::strcmp(argv[i], "--rgl") && i + 1 < argc) rgl_file = argv[++i];
        else if (!std::strcmp(argv[i], "--churn")) churn = true;
        else { std::fprintf(stderr, "usage: scalar_host [--threads T] [--calls K] [--table file.binary] [--rgl file_rgb.bsdf] [--churn]\n"); return 64; }
    }
    mrl_ctx *ctx = nullptr;
    { const int rc = mrl_init(0, &ctx); if (rc != MRL_OK) { std::fprintf(stderr, "mrl_init: %s\n", mrl_strerror(rc)); return 2; } }
    // materials: a table (from a file, or a small smooth synthetic one) and a GGX conductor
    const int dims[3] = { 24, 20, 36 };
    std::vector<double> planar((size_t)3 * dims[0] * dims[1] * dims[2]);
    for (size_t i = 0; i < planar.size(); ++i) planar[i] = 40.0 + 30.0 * std::sin(0.37 * (double)(i % 9973)) + 0.001 * (double)(i % 4099);
    const double scale[3] = { 0.01, 0.012, 0.016 };
    int tab = -1, ggx = -1;
    if (table) CHECK(mrl_material_load_merl(ctx, table, &tab));
    else CHECK(mrl_material_upload_table(ctx, planar.data(), dims, scale, &tab));
    const float eta[3] = { 0.143f, 0.375f, 1.442f }, kk[3] = { 3.983f, 2.386f, 1.603f };
    CHECK(mrl_material_ggx(ctx, 0.2f, eta, kk, &ggx));

    // the reference answers: one batch call over every request of every thread
    const size_t n = (size_t)threads * (size_t)calls;
    std::vector<float> wi(3 * n), wo(3 * n), u(2 * n), want(11 * n);
    std::vector<int32_t> mat(n);
    for (size_t i = 0; i < n; ++i) {
        uint64_t s = 0x5EEDull * (i + 1);
        hemi(s, &wi[3 * i]); hemi(s, &wo[3 * i]); u[2 * i] = uni(s); u[2 * i + 1] = uni(s);
        mat[i] = (i % 5 == 4) ? ggx : tab;
        if (i % 97 == 0) wi[3 * i + 2] = -wi[3 * i + 2];                // below the horizon: zeros
    }
    {
        std::vector<float> rgb(3 * n), pdf(n), wo2(3 * n), pdf2(n), w(3 * n);
        CHECK(mrl_eval_sample_batch(ctx, wi.data(), wo.data(), u.data(), mat.data(), 0, n, rgb.data(), pdf.data(), wo2.data(), pdf2.data(), w.data()));
        CHECK(mrl_synchronize(ctx));
        for (size_t i = 0; i < n; ++i) {
            float *o = &want[11 * i];
            std::memcpy(o, &rgb[3 * i], 12); o[3] = pdf[i]; std::memcpy(o + 4, &wo2[3 * i], 12); o[7] = pdf2[i]; std::memcpy(o + 8, &w[3 * i], 12);
        }
    }
    using Clock = std::chrono::steady_clock;
    std::atomic<long> wrong{ 0 }, failed{ 0 };
    auto worker = [&](int t, long k0, long k1) {
        for (long k = k0; k < k1; ++k) {
            const size_t i = (size_t)t * (size_t)calls + (size_t)k;
            float out[11];
            const int rc = mrl_scalar_eval_sample(ctx, mat[i], &wi[3 * i], &wo[3 * i], &u[2 * i], out);
            if (rc != MRL_OK) { ++failed; continue; }
            if (std::memcmp(out, &want[11 * i], sizeof out) != 0) ++wrong;
        }
    };
    // 1. one thread alone: the latency of a call (after a short warm-up that starts the service)
    worker(0, 0, std::min<long>(calls, 200));
    const long solo = std::min<long>(calls, 5000);
    auto t0 = Clock::now();
    worker(0, 0, solo);
    const double solo_us = std::chrono::duration<double, std::micro>(Clock::now() - t0).count() / (double)solo;
    // ... and the two halves on their own: what one virtual eval() / pdf(), resp. one sample(), asks for
    double half_us[2] = { 0.0, 0.0 };
    for (int half = 0; half < 2; ++half) {
        t0 = Clock::now();
        for (long k = 0; k < solo; ++k) {
            const size_t i = (size_t)k;
            float rgb[3], pdf, wo2[3], pdf2, w[3];
            const int rc = half == 0 ? mrl_scalar_eval_pdf(ctx, mat[i], &wi[3 * i], &wo[3 * i], rgb, &pdf)
                                     : mrl_scalar_sample(ctx, mat[i], &wi[3 * i], &u[2 * i], wo2, &pdf2, w);
            if (rc != MRL_OK) { ++failed; continue; }
            const float *ref = &want[11 * i];
            const bool ok = half == 0 ? (!std::memcmp(rgb, ref, 12) && !std::memcmp(&pdf, ref + 3, 4))
                                      : (!std::memcmp(wo2, ref + 4, 12) && !std::memcmp(&pdf2, ref + 7, 4) && !std::memcmp(w, ref + 8, 12));
            if (!ok) ++wrong;
        }
        half_us[half] = std::chrono::duration<double, std::micro>(Clock::now() - t0).count() / (double)solo;
    }
    // 2. all threads together (+ the churn thread)
    std::atomic<bool> stop{ false };
    long churn_rounds = 0;
    std::thread churner;
    if (churn) churner = std::thread([&] {
        while (!stop.load()) {
            int extra = -1;
            if (mrl_material_upload_table(ctx, planar.data(), dims, scale, &extra) != MRL_OK) { ++failed; break; }
            (void)mrl_set_option(ctx, MRL_OPT_BLOCK_MAP, (int)(churn_rounds & 1));      // any option change pauses the service
            if (mrl_material_release(ctx, extra) != MRL_OK) { ++failed; break; }
            ++churn_rounds;
            std::this_thread::sleep_for(std::chrono::microseconds(300));
        }
    });
    t0 = Clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(worker, t, 0L, calls);
    for (auto &th : pool) th.join();
    const double all_s = std::chrono::duration<double>(Clock::now() - t0).count();
    stop.store(true);
    if (churner.joinable()) churner.join();
    // 3. the same units on the CPU (mrl_host_*: what a plugin with scalar="cpu" calls): one thread, then all threads; answers
    //    compared with the batch call — one Float ulp at most, and how many units are the same bits
    double host_solo_us = 0.0, host_all_us = 0.0, host_same = 0.0, host_worst = 0.0;
    {
        mrl_host_table *ht = nullptr;
        CHECK(mrl_material_host_table(ctx, tab, &ht));
        std::atomic<long> same{ 0 }, units{ 0 };
        std::vector<double> worst((size_t)threads, 0.0);
        auto host_worker = [&](int t, long k0, long k1) {
            for (long k = k0; k < k1; ++k) {
                const size_t i = (size_t)t * (size_t)calls + (size_t)k;
                if (mat[i] != tab) continue;
                float o[11];
                if (mrl_host_eval_sample(ht, &wi[3 * i], &wo[3 * i], &u[2 * i], o) != MRL_OK) { ++failed; continue; }
                ++units;
                if (!std::memcmp(o, &want[11 * i], sizeof o)) { ++same; continue; }
                for (int c = 0; c < 11; ++c) {
                    const double a = o[c], b = want[11 * i + c];
                    if (a != b) worst[(size_t)t] = std::max(worst[(size_t)t], std::fabs(a - b) / std::max(std::fabs(b), 1e-30));
                }
            }
        };
        host_worker(0, 0, std::min<long>(calls, 200));
        same = 0; units = 0;
        t0 = Clock::now();
        host_worker(0, 0, calls);
        host_solo_us = std::chrono::duration<double, std::micro>(Clock::now() - t0).count() / (double)std::max<long>(units.load(), 1);
        same = 0; units = 0;
        t0 = Clock::now();
        std::vector<std::thread> hp;
        for (int t = 0; t < threads; ++t) hp.emplace_back(host_worker, t, 0L, calls);
        for (auto &th : hp) th.join();
        host_all_us = std::chrono::duration<double, std::micro>(Clock::now() - t0).count() / (double)std::max<long>(units.load(), 1);
        host_same = (double)same.load() / (double)std::max<long>(units.load(), 1);
        for (double w : worst) host_worst = std::max(host_worst, w);
        if (host_worst > 1.3e-7) ++wrong;                        // more than one Float ulp from the batch call
        mrl_host_table_release(ht);
    }
    // 4. --rgl: an RGL adaptive-parameterisation material (upstream Mitsuba 3 `measured`): one-unit calls on the CPU over its host
    //    image against the GPU batch call on the same units — the same functions on two targets, 1e-6 relative at most
    double rgl_solo_us = 0.0, rgl_all_us = 0.0, rgl_same = 0.0, rgl_worst = 0.0;
    if (rgl_file) {
        int rid = -1;
        if (mrl_material_load_rgl(ctx, rgl_file, &rid) != MRL_OK) {
            std::fprintf(stderr, "mrl_material_load_rgl: %s %s\n", mrl_tensor_file_last_error(nullptr), mrl_last_error(ctx));
            return 3;
        }
        std::vector<float> rgb(3 * n), pdf(n), wo2(3 * n), pdf2(n), w(3 * n), ref(11 * n);
        CHECK(mrl_eval_sample_batch(ctx, wi.data(), wo.data(), u.data(), nullptr, rid, n, rgb.data(), pdf.data(), wo2.data(), pdf2.data(), w.data()));
        CHECK(mrl_synchronize(ctx));
        for (size_t i = 0; i < n; ++i) {
            float *o = &ref[11 * i];
            std::memcpy(o, &rgb[3 * i], 12); o[3] = pdf[i]; std::memcpy(o + 4, &wo2[3 * i], 12); o[7] = pdf2[i]; std::memcpy(o + 8, &w[3 * i], 12);
        }
        mrl_host_table *ht = nullptr;
        CHECK(mrl_material_host_table(ctx, rid, &ht));
        std::atomic<long> same{ 0 };
        std::vector<double> worst((size_t)threads, 0.0);
        auto rgl_worker = [&](int t, long k0, long k1) {
            for (long k = k0; k < k1; ++k) {
                const size_t i = (size_t)t * (size_t)calls + (size_t)k;
                float o[11];
                if (mrl_host_eval_sample(ht, &wi[3 * i], &wo[3 * i], &u[2 * i], o) != MRL_OK) { ++failed; continue; }
                if (!std::memcmp(o, &ref[11 * i], sizeof o)) { ++same; continue; }
                // a sampled direction one ulp away moves its pdf / weight: compare eval and pdf, which are functions of the inputs alone
                for (int c = 0; c < 4; ++c) {
                    const double a = o[c], b = ref[11 * i + c];
                    if (a != b) worst[(size_t)t] = std::max(worst[(size_t)t], std::fabs(a - b) / std::max(std::fabs(b), 1e-30));
                }
            }
        };
        rgl_worker(0, 0, std::min<long>(calls, 200));
        same = 0;
        t0 = Clock::now();
        rgl_worker(0, 0, calls);
        rgl_solo_us = std::chrono::duration<double, std::micro>(Clock::now() - t0).count() / (double)calls;
        same = 0;
        t0 = Clock::now();
        std::vector<std::thread> hp;
        for (int t = 0; t < threads; ++t) hp.emplace_back(rgl_worker, t, 0L, calls);
        for (auto &th : hp) th.join();
        rgl_all_us = std::chrono::duration<double, std::micro>(Clock::now() - t0).count() / (double)n;
        rgl_same = (double)same.load() / (double)n;
        for (double x : worst) rgl_worst = std::max(rgl_worst, x);
        if (rgl_worst > 1e-6) ++wrong;
        mrl_host_table_release(ht);
        CHECK(mrl_material_release(ctx, rid));
    }
    // an id the scalar path refuses
    float out[11];
    const int bad = mrl_scalar_eval_sample(ctx, 99, &wi[0], &wo[0], &u[0], out);
    std::printf("{\"threads\": %d, \"calls_per_thread\": %ld, \"solo_us_per_call\": %.3f, \"solo_eval_pdf_us\": %.3f, \"solo_sample_us\": %.3f, \"all_threads_Mcalls_per_s\": %.4f, "
                "\"all_threads_us_per_call_amortised\": %.4f, \"cpu_path_us_per_call\": %.4f, \"cpu_path_us_per_call_amortised\": %.4f, "
                "\"cpu_path_units_bit_identical_to_batch\": %.6f, \"cpu_path_worst_rel_diff_to_batch\": %.3g, "
                "\"rgl_cpu_path_us_per_call\": %.4f, \"rgl_cpu_path_us_per_call_amortised\": %.4f, \"rgl_units_bit_identical_to_batch\": %.6f, "
                "\"rgl_worst_rel_diff_to_batch\": %.3g, \"wrong\": %ld, \"failed\": %ld, \"churn_rounds\": %ld, \"unknown_id_status\": %d}\n",
                threads, calls, solo_us, half_us[0], half_us[1], (double)n / all_s / 1e6, all_s * 1e6 / (double)n, host_solo_us, host_all_us, host_same, host_worst,
                rgl_solo_us, rgl_all_us, rgl_same, rgl_worst, wrong.load(), failed.load(), churn_rounds, bad);
    mrl_destroy(ctx);
    return (wrong.load() == 0 && failed.load() == 0 && bad == MRL_ERR_MATERIAL) ? 0 : 1;
}
