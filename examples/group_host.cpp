// group_host.cpp — a native C++ host that drives several GPUs through the C ABI's device groups
// (include/merl_hip.h): no Python, no torch, one process, one mrl_group.  It is what a C++ renderer's
// multi-GPU outer loop looks like, reduced to the BSDF path: replicate the tables, generate each member's
// unit tile in place, run the sharded fused eval+sample with the results gathered to a root device
// (RCCL point-to-point over xGMI, or device copies), time it, and check the gathered arrays bit for bit
// against a single-device run over the same unit range.
//
//   g++ -std=c++17 -O2 -I include examples/group_host.cpp -L mitsuba_customization_amd/lib -lmerl_hip -o group_host
//   group_host --devices 0,1,2,3 [--transport auto|rccl|copy] [--units-per-device N] [--chunk C] [--tables T]
//              [--steps K] [--warmup W] [--table file.binary] [--check] [--root R] [--selftest] [--no-fallback] [--reserve-cus K]
//   (a device may repeat, e.g. --devices 0,0,0: rehearsal on a 1-GPU box, transport = device copies)
// --selftest: before the pipeline, every peer -> root link on its own (mrl_group_link_test: 1 / 4 / 16 / 64 MB, timed and
//   bit-checked, RCCL and device copies), reported per link.
// With more than one device and a transport that may be RCCL, the work runs in a CHILD process (this program again,
//   started before anything here touches the GPU); if that child fails — RCCL cannot initialise, a send errors, the
//   process dies — a fresh child repeats the run with device copies, and the line it prints says what failed.  Exit code
//   non-zero only if both fail (--no-fallback: no second attempt).
// Prints ONE JSON line; three rates side by side: compute only, with the rgb-only gather (12 B per unit across the links)
//   and with the full gather (44 B per unit); exit code 0 only if every call succeeded and --check found no difference.
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "merl_hip.h"

namespace {

#define GCHECK(call)                                                                                   \
    do {                                                                                               \
        int rc_ = (call);                                                                              \
        if (rc_ != MRL_OK) {                                                                           \
            std::fprintf(stderr, "%s -> %s (%s)\n", #call, mrl_strerror(rc_), mrl_group_last_error(group)); \
            return 1;                                                                                  \
        }                                                                                              \
    } while (0)
#define CCHECK(ctx, call)                                                                              \
    do {                                                                                               \
        int rc_ = (call);                                                                              \
        if (rc_ != MRL_OK) {                                                                           \
            std::fprintf(stderr, "%s -> %s (%s)\n", #call, mrl_strerror(rc_), mrl_last_error(ctx));    \
            return 1;                                                                                  \
        }                                                                                              \
    } while (0)

// A smooth, strictly positive MERL-shaped table (a glossy lobe in theta_h over a diffuse floor), raw file units:
// stand-in for a measured material when no .binary is given.  Any table exercises the same code; this one
// spans several decades like measured data does.
std::vector<double> synthetic_table(int seed)
{
    const int H = 90, D = 90, P = 180;
    const size_t plane = (size_t)H * D * P;
    std::vector<double> t(3 * plane);
    const double alpha = 0.05 + 0.03 * (seed % 7), albedo[3] = { 0.3 + 0.05 * (seed % 5), 0.25, 0.2 + 0.04 * (seed % 3) };
    for (int h = 0; h < H; ++h) {
        const double th = (double)h * h / (H * (double)H) * 1.5707963267948966;
        const double c = std::cos(th), tn = std::tan(th);
        const double lobe = 1.0 / (3.141592653589793 * alpha * alpha * c * c * c * c * std::pow(1.0 + tn * tn / (alpha * alpha), 2.0));
        for (int d = 0; d < D; ++d) {
            const double td = (d + 0.5) / D * 1.5707963267948966, fres = 0.04 + 0.96 * std::pow(1.0 - std::cos(td), 5.0);
            for (int p = 0; p < P; ++p) {
                const size_t i = ((size_t)h * D + d) * P + p;
                const double wob = 1.0 + 0.05 * std::sin(0.1 * p + seed);
                t[i] = 1500.0 * (albedo[0] / 3.141592653589793 + lobe * fres * 0.25) * wob;
                t[i + plane] = 1500.0 / 1.15 * (albedo[1] / 3.141592653589793 + lobe * fres * 0.25) * wob;
                t[i + 2 * plane] = 1500.0 / 1.66 * (albedo[2] / 3.141592653589793 + lobe * fres * 0.25) * wob;
            }
        }
    }
    return t;
}

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

extern char **environ;

// this program again, as a child process, with extra arguments; returns its exit code, or 128 + signal
static int run_child(char **argv, const std::vector<std::string> &extra)
{
    std::vector<std::string> args;
    for (char **a = argv; *a; ++a) args.push_back(*a);
    for (const std::string &e : extra) args.push_back(e);
    std::vector<char *> cargs;
    for (std::string &a : args) cargs.push_back(&a[0]);
    cargs.push_back(nullptr);
    pid_t pid = 0;
    if (posix_spawn(&pid, "/proc/self/exe", nullptr, nullptr, cargs.data(), environ) != 0) return 127;
    int status = 0;
    if (waitpid(pid, &status, 0) < 0) return 127;
    return WIFEXITED(status) ? WEXITSTATUS(status) : 128 + (WIFSIGNALED(status) ? WTERMSIG(status) : 0);
}

int main(int argc, char **argv)
{
    std::vector<int> devices = { 0 };
    int transport = MRL_TRANSPORT_AUTO, tables = 1, steps = 5, warmup = 2, root = 0;
    size_t units_per_device = (size_t)8 << 20, chunk = (size_t)2 << 20;
    bool check = false, selftest = false, child = false, fallback = true;
    int reserve_cus = 0;
    std::string table_file, fallback_from;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--devices") {
            devices.clear();
            std::string s = next();
            size_t p = 0;
            while (p <= s.size()) {
                const size_t q = s.find(',', p);
                devices.push_back(std::atoi(s.substr(p, q == std::string::npos ? std::string::npos : q - p).c_str()));
                if (q == std::string::npos) break;
                p = q + 1;
            }
        } else if (a == "--transport") {
            const std::string t = next();
            transport = t == "rccl" ? MRL_TRANSPORT_RCCL : t == "copy" ? MRL_TRANSPORT_PEER_COPY : MRL_TRANSPORT_AUTO;
        } else if (a == "--units-per-device") units_per_device = (size_t)std::atoll(next());
        else if (a == "--chunk") chunk = (size_t)std::atoll(next());
        else if (a == "--tables") tables = std::atoi(next());
        else if (a == "--steps") steps = std::atoi(next());
        else if (a == "--warmup") warmup = std::atoi(next());
        else if (a == "--root") root = std::atoi(next());
        else if (a == "--table") table_file = next();
        else if (a == "--check") check = true;
        else if (a == "--selftest") selftest = true;
        else if (a == "--no-fallback") fallback = false;
        else if (a == "--reserve-cus") reserve_cus = std::atoi(next());
        else if (a == "--child") child = true;
        else if (a == "--fallback-from") fallback_from = next();
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    const int G = (int)devices.size();
    if (G < 1 || tables < 1 || steps < 1 || root < 0 || root >= G || chunk < 1) { std::fprintf(stderr, "bad arguments\n"); return 2; }

    // ---- parent: with an RCCL leg possible, the run happens in a child; a failed child is followed by a fresh one on copies.
    //      Nothing before this point has touched the GPU (a process that has must not start another program). ----
    if (!child && G > 1 && transport != MRL_TRANSPORT_PEER_COPY) {
        const int rc = run_child(argv, { "--child" });
        if (rc == 0 || !fallback) return rc;
        std::fprintf(stderr, "group_host: the run with transport %s failed (%s %d); repeating it with device copies in a fresh process\n",
                     transport == MRL_TRANSPORT_RCCL ? "rccl" : "auto", rc >= 128 ? "signal" : "exit code", rc >= 128 ? rc - 128 : rc);
        const int rc2 = run_child(argv, { "--child", "--transport", "copy", "--fallback-from",
                                          std::string(transport == MRL_TRANSPORT_RCCL ? "rccl" : "auto") + (rc >= 128 ? " signal " : " exit code ") +
                                              std::to_string(rc >= 128 ? rc - 128 : rc) });
        return rc2;
    }

    mrl_group *group = nullptr;
    {
        const int rc = mrl_group_init(G, devices.data(), transport, &group);
        if (rc != MRL_OK) { std::fprintf(stderr, "mrl_group_init -> %s (%s)\n", mrl_strerror(rc), mrl_group_last_error(nullptr)); return 1; }
    }
    const int used_transport = mrl_group_transport(group);
    // CUs the compute grids leave to the transfer kernels (RCCL's send / receive are kernels): MRL_OPT_RESERVED_CUS on every member
    if (reserve_cus > 0) GCHECK(mrl_group_set_option(group, MRL_OPT_RESERVED_CUS, reserve_cus));
    std::vector<int> ids;
    std::vector<std::vector<double>> synthetic(16);                    // 100 resident tables cycle through 16 distinct ones
    for (int t = 0; t < tables; ++t) {
        int id = -1;
        if (!table_file.empty()) GCHECK(mrl_group_material_load_merl(group, table_file.c_str(), &id));
        else {
            std::vector<double> &tab = synthetic[(size_t)(t % 16)];
            if (tab.empty()) tab = synthetic_table(t % 16);
            GCHECK(mrl_group_material_upload_f64(group, tab.data(), &id));
        }
        ids.push_back(id);
    }
    const size_t n_total = units_per_device * (size_t)G;
    std::vector<mrl_tile_inputs> tiles((size_t)G);
    GCHECK(mrl_group_generate_tiles(group, 0x5EEDu, 0, n_total, tables > 1 ? tables : 0, tiles.data()));

    mrl_ctx *rctx = nullptr;
    GCHECK(mrl_group_context(group, root, &rctx));
    float *out = nullptr;                                        // rgb[3n] pdf[n] wo[3n] pdf2[n] weight[3n] on the root
    CCHECK(rctx, mrl_device_alloc(rctx, n_total * 11 * sizeof(float), (void **)&out));
    float *o_rgb = out, *o_pdf = out + 3 * n_total, *o_wo = out + 4 * n_total, *o_pdf2 = out + 7 * n_total, *o_w = out + 8 * n_total;

    // ---- link selftest: every peer -> root link on its own, before the pipeline ----
    std::string selftest_json = "null";
    if (selftest && G > 1) {
        selftest_json = "{";
        const int modes[2] = { MRL_TRANSPORT_RCCL, MRL_TRANSPORT_PEER_COPY };
        bool first_mode = true;
        for (int mode : modes) {
            if (mode == MRL_TRANSPORT_RCCL && used_transport != MRL_TRANSPORT_RCCL) continue;
            selftest_json += std::string(first_mode ? "" : ", ") + "\"" + (mode == MRL_TRANSPORT_RCCL ? "rccl" : "peer_copy") + "\": [";
            first_mode = false;
            const size_t sizes[4] = { (size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20 };
            for (int k = 0; k < 4; ++k) {
                std::vector<mrl_link_report> rep((size_t)G);
                // a failed link test is REPORTED, it does not end the run: the pipeline below is the measurement, and its --check
                // compares every gathered value (a selftest bug must not hide the transport behind the fallback)
                int rc_link = mrl_group_link_test(group, sizes[k], mode, root, rep.data());
                if (rc_link == MRL_OK) rc_link = mrl_group_link_test(group, sizes[k], mode, root, rep.data());       // second pass: timed warm
                if (rc_link != MRL_OK) {
                    std::string why = mrl_group_last_error(group) ? mrl_group_last_error(group) : "";
                    for (char &c : why) if (c == '"' || c == '\\' || c == '\n') c = ' ';
                    selftest_json += std::string(k ? ", " : "") + "{\"bytes\": " + std::to_string(sizes[k]) + ", \"failed\": \"" + mrl_strerror(rc_link) + ": " + why + "\"}";
                    std::fprintf(stderr, "group_host: link selftest (%s, %zu bytes) failed: %s (%s)\n", mode == MRL_TRANSPORT_RCCL ? "rccl" : "peer_copy", sizes[k],
                                 mrl_strerror(rc_link), why.c_str());
                    break;
                }
                selftest_json += std::string(k ? ", " : "") + "{\"bytes\": " + std::to_string(sizes[k]) + ", \"GBps_per_peer\": [";
                bool first_peer = true;
                for (int r = 0; r < G; ++r) {
                    if (r == root) continue;
                    char num[32];
                    std::snprintf(num, sizeof num, "%.2f", rep[(size_t)r].GBps);
                    selftest_json += std::string(first_peer ? "" : ", ") + num;
                    first_peer = false;
                }
                selftest_json += "], \"mismatches\": 0}";
            }
            selftest_json += "]";
        }
        selftest_json += "}";
    }

    // ---- compute only: every member runs its tile into member-local arrays, nothing moves between devices ----
    std::vector<float *> local((size_t)G, nullptr);
    std::vector<mrl_ctx *> ctxs((size_t)G, nullptr);
    for (int r = 0; r < G; ++r) {
        GCHECK(mrl_group_context(group, r, &ctxs[(size_t)r]));
        size_t lo, hi;
        mrl_tile_bounds(n_total, G, r, &lo, &hi);
        CCHECK(ctxs[(size_t)r], mrl_device_alloc(ctxs[(size_t)r], (hi - lo) * 11 * sizeof(float), (void **)&local[(size_t)r]));
    }
    auto compute_only = [&]() -> int {
        for (int r = 0; r < G; ++r) {
            size_t lo, hi;
            mrl_tile_bounds(n_total, G, r, &lo, &hi);
            const size_t n = hi - lo;
            float *b = local[(size_t)r];
            const mrl_tile_inputs &in = tiles[(size_t)r];
            const int rc = mrl_eval_sample_batch(ctxs[(size_t)r], in.wi, in.wo, in.u, in.mat, ids[0], n, b, b + 3 * n, b + 4 * n, b + 7 * n, b + 8 * n);
            if (rc != MRL_OK) return rc;
        }
        return mrl_group_synchronize(group);
    };
    for (int w = 0; w < warmup; ++w) GCHECK(compute_only());
    double t0 = now_ms();
    for (int s = 0; s < steps; ++s) GCHECK(compute_only());
    const double compute_ms = (now_ms() - t0) / steps;

    // ---- end to end: sharded run with the chunk-pipelined gather into the root's arrays ----
    auto sharded = [&]() -> int {
        const int rc = mrl_group_eval_sample_sharded(group, tiles.data(), ids[0], n_total, chunk, root, o_rgb, o_pdf, o_wo, o_pdf2, o_w);
        return rc != MRL_OK ? rc : mrl_group_synchronize(group);
    };
    for (int w = 0; w < warmup; ++w) GCHECK(sharded());
    t0 = now_ms();
    for (int s = 0; s < steps; ++s) GCHECK(sharded());
    const double gathered_ms = (now_ms() - t0) / steps;
    std::vector<float> member_ms((size_t)G, 0.0f);
    GCHECK(mrl_group_last_timing(group, member_ms.data()));

    // ---- the same with the rgb-only gather: eval alone, 12 B per unit across the links ----
    auto sharded_rgb = [&]() -> int {
        const int rc = mrl_group_eval_sharded(group, tiles.data(), ids[0], n_total, chunk, root, o_rgb);
        return rc != MRL_OK ? rc : mrl_group_synchronize(group);
    };
    for (int w = 0; w < warmup; ++w) GCHECK(sharded_rgb());
    t0 = now_ms();
    for (int s = 0; s < steps; ++s) GCHECK(sharded_rgb());
    const double rgb_gathered_ms = (now_ms() - t0) / steps;
    GCHECK(sharded());                                           // the arrays --check compares are the fused call's

    // ---- check: the gathered arrays == one device evaluating the whole unit range, bit for bit ----
    long long mismatches = -1;
    if (check) {
        mismatches = 0;
        const size_t piece = (size_t)4 << 20;                    // compare in pieces: bounded host and device memory
        float *d_in = nullptr, *d_ref = nullptr;
        int32_t *d_mat = nullptr;
        CCHECK(rctx, mrl_device_alloc(rctx, piece * 8 * sizeof(float), (void **)&d_in));
        CCHECK(rctx, mrl_device_alloc(rctx, piece * 11 * sizeof(float), (void **)&d_ref));
        CCHECK(rctx, mrl_device_alloc(rctx, piece * sizeof(int32_t), (void **)&d_mat));
        std::vector<float> h_ref(piece * 11), h_got(piece * 11);
        for (size_t a = 0; a < n_total; a += piece) {
            const size_t n = std::min(piece, n_total - a);
            CCHECK(rctx, mrl_generate_pairs(rctx, 0x5EEDu, a, n, d_in, d_in + 3 * n, d_in + 6 * n));
            if (tables > 1) CCHECK(rctx, mrl_generate_materials(rctx, 0x5EEDu, a, n, tables, d_mat));
            CCHECK(rctx, mrl_eval_sample_batch(rctx, d_in, d_in + 3 * n, d_in + 6 * n, tables > 1 ? d_mat : nullptr, ids[0], n,
                                               d_ref, d_ref + 3 * n, d_ref + 4 * n, d_ref + 7 * n, d_ref + 8 * n));
            CCHECK(rctx, mrl_copy_to_host(rctx, h_ref.data(), d_ref, n * 11 * sizeof(float)));
            const float *src[5] = { o_rgb + 3 * a, o_pdf + a, o_wo + 3 * a, o_pdf2 + a, o_w + 3 * a };
            const size_t width[5] = { 3, 1, 3, 1, 3 }, at[5] = { 0, 3, 4, 7, 8 };
            for (int k = 0; k < 5; ++k) {
                CCHECK(rctx, mrl_copy_to_host(rctx, h_got.data(), src[k], n * width[k] * sizeof(float)));
                const float *want = h_ref.data() + at[k] * n;
                for (size_t j = 0; j < n * width[k]; ++j)
                    if (std::memcmp(&h_got[j], &want[j], 4) != 0) ++mismatches;
            }
        }
        CCHECK(rctx, mrl_device_free(rctx, d_in));
        CCHECK(rctx, mrl_device_free(rctx, d_ref));
        CCHECK(rctx, mrl_device_free(rctx, d_mat));
    }

    float slowest = 0.0f;
    for (float m : member_ms) slowest = std::max(slowest, m);
    const double bytes_into_root = (double)(n_total - units_per_device) * 44.0;
    std::printf("{\"what\": \"native C++ host, one process, mrl_group over %d device(s)\", \"devices\": [", G);
    for (int r = 0; r < G; ++r) std::printf("%s%d", r ? ", " : "", devices[(size_t)r]);
    std::printf("], \"transport\": \"%s\", \"units_per_device\": %zu, \"chunk_units\": %zu, \"tables_resident\": %d, \"steps\": %d, "
                "\"compute_only_ms\": %.4f, \"compute_only_Meval_s\": %.1f, \"rgb_gathered_ms\": %.4f, \"rgb_gathered_Meval_s\": %.1f, "
                "\"gathered_ms\": %.4f, \"gathered_Meval_s\": %.1f, "
                "\"slowest_member_device_ms\": %.4f, \"bytes_into_root\": %.0f, \"root_ingress_GBps\": %.2f, \"rgb_bytes_into_root\": %.0f, "
                "\"rgb_root_ingress_GBps\": %.2f, \"check_mismatches\": %lld, \"fallback_from\": %s%s%s, \"reserved_cus\": %d, \"selftest\": %s}\n",
                used_transport == MRL_TRANSPORT_RCCL ? "rccl" : "peer_copy", units_per_device, chunk, tables, steps,
                compute_ms, (double)n_total / compute_ms / 1e3, rgb_gathered_ms, (double)n_total / rgb_gathered_ms / 1e3,
                gathered_ms, (double)n_total / gathered_ms / 1e3, slowest,
                bytes_into_root, gathered_ms > 0 ? bytes_into_root / gathered_ms / 1e6 : 0.0, bytes_into_root * 12.0 / 44.0,
                rgb_gathered_ms > 0 ? bytes_into_root * 12.0 / 44.0 / rgb_gathered_ms / 1e6 : 0.0, mismatches,
                fallback_from.empty() ? "" : "\"", fallback_from.empty() ? "null" : fallback_from.c_str(), fallback_from.empty() ? "" : "\"",
                reserve_cus, selftest_json.c_str());
    for (int r = 0; r < G; ++r) (void)mrl_device_free(ctxs[(size_t)r], local[(size_t)r]);
    (void)mrl_device_free(rctx, out);
    GCHECK(mrl_group_destroy(group));
    return (check && mismatches != 0) ? 3 : 0;
}
