// merl_ctx.hpp — what the translation units behind the C ABI (include/merl_hip.h) share: the context, its material records, the
// error / lock macros and the internal functions one file defines and another calls.  Not installed; nothing here is part of the ABI.
//   merl_abi.hip           contexts, options, streams, memory helpers, one-unit call service, errors
//   merl_materials.hip     material constructors (MERL / customized_measurement / n-channel / GGX / RGL), release, host images
//   merl_calls.hip         batch, queue and n-channel calls: pointer plumbing, pipelined host arrays, launches
//   merl_image_cache.hip   on-disk cache of a material's device image
//   merl_rgl_spectral.hip  spectral RGL materials: constructor + calls
#pragma once
#include "../../include/merl_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>
#include <unistd.h>

#include "merl_kernels.hpp"
#include "merl_rgl.hpp"
#include "merl_image_file.hpp"
#include "merl_scalar_host.hpp"
#include "merl_host_table.hpp"

namespace mrlabi {


inline constexpr int kMerlDims[3] = { 90, 90, 180 };
inline constexpr double kMerlScale[3] = { 1.0 / 1500.0, 1.15 / 1500.0, 1.66 / 1500.0 };

struct MaterialHost {
    mrl::MaterialDev dev;
    float4 *d_texels = nullptr;
    bool in_arena = false;           // d_texels is a slice of the context's table arena (MRL_OPT_TABLE_ARENA_MB): not freed on its own
    double *d_sampling = nullptr;
    double *d_sampling2d = nullptr;  // P(theta_h | theta_i) rows (RGB tables), built on the device at upload
    size_t bytes = 0;                // device bytes this material holds (table + sampling marginal)
    mrl::RglDev rgl{};               // KIND_RGL: the five functions' descriptor (pointers into d_texels)
    bool released = false;           // tombstone left by mrl_material_release; the slot may be reused
    int rows_lookup = 1, rows_node = 0;      // the lookup / node options the conditional sampling rows were integrated under (at upload)
};

} // namespace mrlabi
using mrlabi::MaterialHost;


// ---- pipelined host-array path -------------------------------------------------------------------------------------
// A host that holds plain (pageable) arrays — what a CPU renderer hands over — used to be staged with hipMemcpyAsync,
// which the runtime serialises through one bounce buffer at ~11 GB/s (140-150 M units/s).  Instead: a few copy threads
// move chunk c+1 of the caller's arrays into pinned, device-mapped buffers and chunk c-1 of the results out of them,
// while the kernel of chunk c reads and writes the pinned buffers over PCIe itself (zero copy, no staging in HBM).
struct CopyPool {
    struct Seg { void *dst; const void *src; size_t bytes; };
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable wake, done;
    std::vector<Seg> segs;
    size_t next = 0, finished = 0;
    uint64_t generation = 0;
    bool quit = false;

    void start(int n)
    {
        for (int t = 0; t < n; ++t)
            workers.emplace_back([this]() {
                uint64_t seen = 0;
                for (;;) {
                    std::unique_lock<std::mutex> lk(mu);
                    wake.wait(lk, [&]() { return quit || (generation != seen && next < segs.size()) || (generation != seen && segs.empty()); });
                    if (quit) return;
                    if (next >= segs.size()) { seen = generation; continue; }
                    while (next < segs.size()) {
                        const Seg sg = segs[next++];
                        lk.unlock();
                        std::memcpy(sg.dst, sg.src, sg.bytes);
                        lk.lock();
                        if (++finished == segs.size()) done.notify_all();
                    }
                    seen = generation;
                }
            });
    }
    // copies every segment, split into slices so that all workers (and the caller) share the work; returns when done
    void run(const std::vector<Seg> &whole)
    {
        constexpr size_t kSlice = (size_t)2 << 20;
        std::vector<Seg> sliced;
        for (const Seg &w : whole)
            for (size_t off = 0; off < w.bytes; off += kSlice)
                sliced.push_back({ (char *)w.dst + off, (const char *)w.src + off, std::min(kSlice, w.bytes - off) });
        if (sliced.empty()) return;
        if (workers.empty()) { for (const Seg &sg : sliced) std::memcpy(sg.dst, sg.src, sg.bytes); return; }
        std::unique_lock<std::mutex> lk(mu);
        segs = std::move(sliced); next = 0; finished = 0; ++generation;
        wake.notify_all();
        while (next < segs.size()) {                              // the caller copies too
            const Seg sg = segs[next++];
            lk.unlock();
            std::memcpy(sg.dst, sg.src, sg.bytes);
            lk.lock();
            ++finished;
        }
        done.wait(lk, [&]() { return finished == segs.size(); });
        segs.clear();
    }
    void stop()
    {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        wake.notify_all();
        for (auto &t : workers) t.join();
        workers.clear();
    }
};

struct HostPipe {
    char *pin[2] = { nullptr, nullptr };             // per slot: inputs then outputs of one chunk
    size_t slot_bytes = 0;
    hipEvent_t done[2] = { nullptr, nullptr };
    CopyPool pool;
    int threads = -1;                                // workers the pool was started with
};

// the device side of the one-unit call service (merl_scalar_host.hpp): where the mailbox lives and how an instance of
// the service kernel is put on its own stream
struct ScalarDevice {
    mrl_ctx *ctx = nullptr;
    mrl::ScalarBoard *b = nullptr;           // pinned, coherent host memory; nullptr until the first scalar call
    mrl::ScalarBoard *b_dev = nullptr;       // the same memory as the device addresses it
    hipStream_t stream = nullptr;            // non-blocking: batch launches on the context's stream never queue behind an instance
    uint64_t lifetime_ticks = 50000;         // 500 us of the 100 MHz wall clock
    std::atomic<bool> ok{ true };
    mrl::ScalarBoard *board() { return b; }
    bool launch(uint32_t gen);
    bool healthy() { return ok.load(std::memory_order_relaxed); }
};
using ScalarSvc = mrl::ScalarService<ScalarDevice>;

struct mrl_ctx {
    // every entry point that touches the context takes this lock: calls from several host threads are safe and serialise
    // (device-pointer calls only enqueue, so the lock is held for microseconds; host-array calls hold it for their duration)
    mutable std::recursive_mutex mu;
    int device = 0;
    int compute_units = 256;
    std::string device_name;
    size_t total_mem = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // kind-partitioned mixed batches: [2][queue_cap] unit indices + partition work area behind them
    uint32_t *d_queues = nullptr;
    size_t queue_cap = 0;
    // mrl_partition_by_material: per-chunk count table + totals
    uint32_t *d_part_work = nullptr;
    size_t part_work_cap = 0;
    std::vector<MaterialHost> materials;
    mrl::MaterialDev *d_materials = nullptr;
    size_t d_materials_cap = 0;
    size_t material_bytes = 0;       // sum of MaterialHost::bytes over live materials
    size_t memory_limit = 0;         // MRL_OPT_MEMORY_LIMIT_MB in bytes; 0 = none
    // what a tombstone points at: one all-zero cell (valid in both layouts) + a 1-row sampling marginal
    void *d_dummy = nullptr;
    mrl::Options opts{ 1, 0, 0, 0, 0, 0 };
    int kernel_variant = 3;          // MRL_OPT_KERNEL default: cooperative LDS-DMA brick fetch
    int table_layout = 1;            // layout of tables uploaded from now on (mrl::Layout)
    int table_param = 0;             // parameterisation of customized_measurement tables uploaded from now on (mrl::Param)
    size_t host_chunk = (size_t)1 << 22;
    int block_map = 0;               // MRL_OPT_BLOCK_MAP
    int rgl_search = 0;              // MRL_OPT_RGL_SEARCH
    int device_cus = 256;            // the device's compute units (compute_units = device_cus - reserved_cus sizes the persistent grids)
    int reserved_cus = 0;            // MRL_OPT_RESERVED_CUS
    hipStream_t masked_stream = nullptr;     // own stream restricted to the unreserved CUs (created by MRL_OPT_RESERVED_CUS > 0)
    int host_threads = 4;            // MRL_OPT_HOST_THREADS: copy threads of the pipelined host-array path; 0 = staged hipMemcpy path
    HostPipe pipe;
    void *d_stage = nullptr;
    size_t d_stage_bytes = 0;
    ScalarDevice scalar_dev;
    std::atomic<ScalarSvc *> scalar{ nullptr };      // created by the first mrl_scalar_eval_sample
    // MRL_OPT_TABLE_ARENA_MB: one device allocation that RGB tables are placed in back to back (2 MiB aligned)
    char *arena = nullptr;
    size_t arena_bytes = 0, arena_used = 0;
    int arena_live = 0;              // tables currently placed in it; the bump pointer rewinds when the last one leaves
    std::string last_error;
};


#define MRL_GUARD(ctx) std::lock_guard<std::recursive_mutex> mrl_guard_((ctx)->mu)

#define MRL_HIP(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            (void)hipGetLastError();                                                         \
            return fail((ctx), MRL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
        }                                                                                    \
    } while (0)

// an allocation: out-of-memory is its own status (MRL_ERR_OOM), everything else MRL_ERR_HIP
#define MRL_ALLOC(ctx, expr)                                                                 \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            (void)hipGetLastError();                                                         \
            return fail((ctx), _e == hipErrorOutOfMemory ? MRL_ERR_OOM : MRL_ERR_HIP,        \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                  \
        }                                                                                    \
    } while (0)

namespace mrlabi {

inline constexpr size_t kMaxSegments = 256 * 8 + 64;     // partition_geometry caps segments at 8 per CU

// ---- merl_abi.hip ----
// a non-blocking stream whose kernels may use all but `reserved` of the device's `device_cus` compute units (reserved = 0: a plain
// stream).  The reserved CUs are spread evenly over the mask (one per XCD for 8): what a communication library's kernels — RCCL's
// send / receive — run on while a persistent compute grid owns the rest (include/merl_hip.h, MRL_OPT_RESERVED_CUS).
hipError_t create_compute_stream(int device_cus, int reserved, hipStream_t *out);
int fail(mrl_ctx *ctx, int status, const std::string &msg);
int pointer_kind(const void *p);                                  // 1 = the device can dereference it, 0 = plain host
int common_kind(std::initializer_list<const void *> ptrs);        // 0 / 1, or -1 on a mix
int sync_material_array(mrl_ctx *ctx);
int ensure_dummy(mrl_ctx *ctx);
mrl::MaterialDev tombstone_dev(const mrl_ctx *ctx);
int budget_check(mrl_ctx *ctx, size_t need, size_t in_arena = 0);
bool arena_has_room(const mrl_ctx *ctx, size_t bytes);           // would table_alloc place `bytes` in the context's arena?
int place_material(mrl_ctx *ctx, const MaterialHost &m, int *out_id);
hipError_t table_alloc(mrl_ctx *ctx, size_t bytes, float4 **out, bool *in_arena);
void table_free(mrl_ctx *ctx, float4 *p, bool in_arena);

// Whoever changes what a running service instance reads (the material array, the tables behind it, the options) holds one
// of these: no scalar call is in flight and no instance is running while it lives (merl_scalar_host.hpp, "writer").
struct ScalarPause {
    ScalarSvc *svc;
    explicit ScalarPause(mrl_ctx *ctx) : svc(ctx->scalar.load(std::memory_order_acquire))
    {
        if (svc && !svc->pause()) ctx->scalar_dev.ok.store(false, std::memory_order_relaxed);
    }
    ~ScalarPause() { if (svc) svc->resume(); }
    ScalarPause(const ScalarPause &) = delete;
    ScalarPause &operator=(const ScalarPause &) = delete;
};

} // namespace mrlabi
