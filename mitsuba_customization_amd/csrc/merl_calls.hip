// merl_calls.hip — the batch, queue and n-channel calls of the C ABI (include/merl_hip.h): argument checks, host-or-device pointer
// plumbing, the pipelined host-array path, the launches.  No CPU evaluation path exists here: every entry point ends in a gfx950
// kernel launch or an error.
#include "merl_ctx.hpp"

namespace mrlabi {

int ensure_stage(mrl_ctx *ctx, size_t units, size_t unit_bytes)
{
    if (units * unit_bytes <= ctx->d_stage_bytes) return MRL_OK;
    if (ctx->d_stage) { MRL_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_stage); ctx->d_stage = nullptr; ctx->d_stage_bytes = 0; }
    MRL_ALLOC(ctx, hipMalloc(&ctx->d_stage, units * unit_bytes));
    ctx->d_stage_bytes = units * unit_bytes;
    return MRL_OK;
}

struct BatchCall {
    int mode;                                    // 0 eval, 1 pdf, 2 sample, 3 eval+sample, 4 eval+pdf
    const float *wi, *wo, *u;
    const int32_t *mat;
    int32_t single_id;
    size_t n;
    float *out_rgb, *out_pdf, *out_wo, *out_pdf2, *out_weight;
    int n_ch = 0;                                // 0: the RGB entry points; > 0: *_nch calls, out_rgb / out_weight are n x n_ch
};

inline bool call_has_eval(int mode) { return mode == 0 || mode == 3 || mode == 4; }
inline bool call_has_pdf(int mode) { return mode == 1 || mode == 3 || mode == 4; }
inline bool call_has_sample(int mode) { return mode == 2 || mode == 3; }


int ensure_queues(mrl_ctx *ctx, size_t units)
{
    if (units <= ctx->queue_cap) return MRL_OK;
    if (ctx->d_queues) {
        MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_queues);
        ctx->d_queues = nullptr; ctx->queue_cap = 0;
    }
    MRL_ALLOC(ctx, hipMalloc((void **)&ctx->d_queues, (2 * units + 4 * kMaxSegments + 2) * sizeof(uint32_t)));
    ctx->queue_cap = units;
    return MRL_OK;
}

// kernel arguments of a call whose pointers are all device-accessible
struct DeviceCall {
    mrl::BatchArgs args;
    bool multi, has_ggx, has_table, has_rgl;
};

DeviceCall device_call(const mrl_ctx *ctx, const BatchCall &c)
{
    DeviceCall d;
    mrl::BatchArgs &a = d.args;
    std::memset(&a, 0, sizeof a);
    a.wi = c.wi; a.wo = c.wo; a.u = c.u; a.mat = c.mat; a.n = c.n;
    a.out_rgb = c.out_rgb; a.out_pdf = c.out_pdf; a.out_wo = c.out_wo; a.out_pdf2 = c.out_pdf2; a.out_weight = c.out_weight;
    a.materials = ctx->d_materials;
    a.n_materials = (int)ctx->materials.size();
    a.opts = ctx->opts;
    a.safe = tombstone_dev(ctx);
    a.block_map = ctx->block_map;
    d.multi = c.mat != nullptr;
    if (!d.multi) a.single = ctx->materials[(size_t)c.single_id].dev;
    d.has_ggx = d.has_table = d.has_rgl = false;
    a.any_standard = 0;
    for (const auto &m : ctx->materials) {
        if (m.released) continue;
        d.has_rgl = d.has_rgl || m.dev.kind == mrl::KIND_RGL;
        if (d.multi && m.dev.kind != mrl::KIND_GGX && m.dev.param != mrl::PARAM_HALF_DIFF) a.any_standard = 1;
        d.has_ggx = d.has_ggx || m.dev.kind == mrl::KIND_GGX;
        d.has_table = d.has_table || m.dev.kind == mrl::KIND_MERL || m.dev.kind == mrl::KIND_TABLE ||
                      (c.mode == 1 && m.dev.kind == mrl::KIND_TABLE_NCH);       // pdf serves n-channel tables too
    }
    if (!d.has_ggx && !d.has_table) d.has_table = true;        // only tombstones left: the table path renders them as zeros
    if (!d.multi && a.single.kind != mrl::KIND_GGX && a.single.param != mrl::PARAM_HALF_DIFF) a.any_standard = 1;
    return d;
}

// null-pointer and material checks shared by the whole-array and the queue entry points
int check_call(mrl_ctx *ctx, const BatchCall &c)
{
    const bool has_eval = call_has_eval(c.mode), has_pdf = call_has_pdf(c.mode), has_sample = call_has_sample(c.mode);
    const bool needs_wo = has_eval || has_pdf, needs_u = has_sample;
    if (!c.wi || (needs_wo && !c.wo) || (needs_u && !c.u) || (has_eval && !c.out_rgb) || (has_pdf && !c.out_pdf) ||
        (has_sample && (!c.out_wo || !c.out_pdf2 || !c.out_weight)))
        return fail(ctx, MRL_ERR_INVALID, "null array argument");
    if (ctx->materials.empty()) return fail(ctx, MRL_ERR_MATERIAL, "no material loaded");
    if (!c.mat && (c.single_id < 0 || (size_t)c.single_id >= ctx->materials.size() || ctx->materials[(size_t)c.single_id].released))
        return fail(ctx, MRL_ERR_MATERIAL, "unknown material id");
    if (!c.mat) {
        const mrl::MaterialDev &d = ctx->materials[(size_t)c.single_id].dev;
        if (d.kind == mrl::KIND_RGL) {
            if (c.n_ch > 0) return fail(ctx, MRL_ERR_MATERIAL, "an RGL material has three channels: use the RGB entry points");
            return MRL_OK;
        }
        if (d.kind == mrl::KIND_RGL_SPECTRAL) {                // the pdf is wavelength-free: the RGB pdf call serves it
            if (c.mode == 1 && c.n_ch == 0) return MRL_OK;
            return fail(ctx, MRL_ERR_MATERIAL, "a spectral RGL material: use the mrl_*_spectral_batch entry points");
        }
        if (c.n_ch == 0 && c.mode != 1 && !mrl::kind_is_rgb_path(d.kind))               // pdf is channel-free
            return fail(ctx, MRL_ERR_MATERIAL, "material has " + std::to_string(d.n_ch) + " channels: use the *_nch entry points");
        if (c.n_ch > 0 && c.mode != 1 && (d.kind != mrl::KIND_TABLE_NCH || d.n_ch != c.n_ch))
            return fail(ctx, MRL_ERR_MATERIAL, "material does not have " + std::to_string(c.n_ch) + " channels");
    }
    return MRL_OK;
}

// host-or-device kind of the arrays a call of this mode touches (-1: mixed)
int call_pointer_kind(const BatchCall &c, const void *extra0 = nullptr, const void *extra1 = nullptr)
{
    const bool has_eval = call_has_eval(c.mode), has_pdf = call_has_pdf(c.mode), has_sample = call_has_sample(c.mode);
    const bool needs_wo = has_eval || has_pdf, needs_u = has_sample;
    return common_kind({ c.wi, needs_wo ? c.wo : nullptr, needs_u ? c.u : nullptr, c.mat, extra0, extra1,
                         has_eval ? c.out_rgb : nullptr, has_pdf ? c.out_pdf : nullptr,
                         has_sample ? c.out_wo : nullptr, has_sample ? c.out_pdf2 : nullptr,
                         has_sample ? c.out_weight : nullptr });
}

int launch_device(mrl_ctx *ctx, const BatchCall &c)
{
    const DeviceCall d = device_call(ctx, c);
    const mrl::BatchArgs &a = d.args;
    if (!d.multi && (a.single.kind == mrl::KIND_RGL || a.single.kind == mrl::KIND_RGL_SPECTRAL)) {     // adaptive-parameterisation material: its own kernel (spectral: pdf only)
        MRL_HIP(ctx, mrl::launch_rgl(c.mode, a, &ctx->materials[(size_t)c.single_id].rgl, false, ctx->rgl_search, ctx->compute_units, ctx->stream));
        return MRL_OK;
    }
    if (c.n_ch > 0 && c.mode != 1) {                          // n-channel tables: their own kernels (pdf is channel-free)
        MRL_HIP(ctx, mrl::launch_batch_nch(c.mode, a, d.multi, c.n_ch, ctx->compute_units, ctx->stream));
        return MRL_OK;
    }
    const bool multi = d.multi, has_ggx = d.has_ggx, has_table = d.has_table;
    // MRL_OPT_KERNEL >= 4: a batch that may mix table and analytic materials is split into one dense queue
    // per kind (count / scan / partition, no atomics); each queue then runs through its dedicated kernel
    if (multi && has_ggx && has_table && ctx->kernel_variant >= 4 && c.mode != 1 && ctx->table_layout == mrl::LAYOUT_BRICK &&
        ctx->opts.lookup == 1 && c.n < ((size_t)1 << 32)) {
        uint32_t segments = 0, seg_len = 0;
        mrl::partition_geometry(c.n, ctx->compute_units, &segments, &seg_len);
        if (segments > kMaxSegments) return fail(ctx, MRL_ERR_INVALID, "partition geometry");
        int rc = ensure_queues(ctx, c.n);
        if (rc != MRL_OK) return rc;
        uint32_t *q_table = ctx->d_queues, *q_ggx = ctx->d_queues + ctx->queue_cap, *work = ctx->d_queues + 2 * ctx->queue_cap;
        MRL_HIP(ctx, mrl::launch_partition_kinds(c.mat, c.n, ctx->d_materials, a.n_materials, q_table, q_ggx, work,
                                                 segments, seg_len, ctx->stream));
        const uint32_t *totals = work + 4 * (size_t)segments;
        mrl::BatchArgs qa = a;
        qa.idx = q_table; qa.idx_count = totals;
        MRL_HIP(ctx, mrl::launch_batch_queue(c.mode, qa, false, ctx->compute_units, ctx->stream));
        qa.idx = q_ggx; qa.idx_count = totals + 1;
        MRL_HIP(ctx, mrl::launch_batch_queue(c.mode, qa, true, ctx->compute_units, ctx->stream));
        if (d.has_rgl) MRL_HIP(ctx, mrl::launch_rgl(c.mode, a, nullptr, false, ctx->rgl_search, ctx->compute_units, ctx->stream));
        return MRL_OK;
    }
    MRL_HIP(ctx, mrl::launch_batch(c.mode, a, multi, ctx->kernel_variant, ctx->table_layout, has_ggx, has_table, ctx->compute_units, ctx->stream));
    // the context holds RGL materials: their units (zeros so far) are evaluated by a second launch on the same stream
    if (multi && d.has_rgl) MRL_HIP(ctx, mrl::launch_rgl(c.mode, a, nullptr, false, ctx->rgl_search, ctx->compute_units, ctx->stream));
    return MRL_OK;
}

// Host arrays, pipelined (see HostPipe): per chunk  copy-in (threads) -> kernel on the pinned slot (zero copy over PCIe)
// -> copy-out (threads), double buffered so that the copies of chunks c+1 / c-1 overlap the kernel of chunk c.
int run_host_pipelined(mrl_ctx *ctx, const BatchCall &c)
{
    const bool has_eval = call_has_eval(c.mode), has_pdf = call_has_pdf(c.mode), has_sample = call_has_sample(c.mode);
    const bool needs_wo = has_eval || has_pdf, needs_u = has_sample;
    const size_t C = c.n_ch > 0 ? (size_t)c.n_ch : 3;
    const size_t unit_bytes = 56 + 8 * C;
    const size_t chunk = std::min(std::min(c.n, ctx->host_chunk), (size_t)1 << 20);
    HostPipe &hp = ctx->pipe;
    if (chunk * unit_bytes > hp.slot_bytes) {
        MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int s = 0; s < 2; ++s) { if (hp.pin[s]) (void)hipHostFree(hp.pin[s]); hp.pin[s] = nullptr; }
        hp.slot_bytes = 0;
        for (int s = 0; s < 2; ++s) {
            const hipError_t e = hipHostMalloc((void **)&hp.pin[s], chunk * unit_bytes, hipHostMallocMapped | hipHostMallocPortable);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                for (int k = 0; k < 2; ++k) { if (hp.pin[k]) (void)hipHostFree(hp.pin[k]); hp.pin[k] = nullptr; }
                return fail(ctx, MRL_ERR_OOM, std::string("pinned staging: ") + hipGetErrorString(e));
            }
        }
        hp.slot_bytes = chunk * unit_bytes;
    }
    for (int s = 0; s < 2; ++s)
        if (!hp.done[s]) MRL_HIP(ctx, hipEventCreateWithFlags(&hp.done[s], hipEventDisableTiming));
    if (hp.threads != ctx->host_threads) {                   // the caller copies as well: n - 1 helpers
        hp.pool.stop();
        hp.pool.quit = false;
        hp.pool.start(std::max(0, ctx->host_threads - 1));
        hp.threads = ctx->host_threads;
    }
    struct Slot { float *wi, *wo, *u; int32_t *mat; float *pdf, *wo2, *pdf2, *rgb, *w; };
    auto slot = [&](int s) {
        char *b = hp.pin[s];
        return Slot{ (float *)b, (float *)(b + 12 * chunk), (float *)(b + 24 * chunk), (int32_t *)(b + 32 * chunk), (float *)(b + 36 * chunk),
                     (float *)(b + 40 * chunk), (float *)(b + 52 * chunk), (float *)(b + 56 * chunk), (float *)(b + (56 + 4 * C) * chunk) };
    };
    const size_t steps = (c.n + chunk - 1) / chunk;
    for (size_t k = 0; k <= steps; ++k) {
        if (k < steps) {                                      // copy-in + launch of chunk k
            const int s = (int)(k & 1);
            const size_t off = k * chunk, m = std::min(chunk, c.n - off);
            const Slot sl = slot(s);
            std::vector<CopyPool::Seg> in = { { sl.wi, c.wi + 3 * off, 12 * m } };
            if (needs_wo) in.push_back({ sl.wo, c.wo + 3 * off, 12 * m });
            if (needs_u) in.push_back({ sl.u, c.u + 2 * off, 8 * m });
            if (c.mat) in.push_back({ sl.mat, c.mat + off, 4 * m });
            hp.pool.run(in);                                  // slot s was last read by kernel k-2, whose event was waited for below
            BatchCall d = c;
            d.wi = sl.wi; d.wo = sl.wo; d.u = sl.u; d.mat = c.mat ? sl.mat : nullptr; d.n = m;
            d.out_rgb = sl.rgb; d.out_pdf = sl.pdf; d.out_wo = sl.wo2; d.out_pdf2 = sl.pdf2; d.out_weight = sl.w;
            const int rc = launch_device(ctx, d);
            if (rc != MRL_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
            MRL_HIP(ctx, hipEventRecord(hp.done[s], ctx->stream));
        }
        if (k > 0) {                                          // copy-out of chunk k-1, while the kernel of chunk k runs
            const int s = (int)((k - 1) & 1);
            const size_t off = (k - 1) * chunk, m = std::min(chunk, c.n - off);
            const Slot sl = slot(s);
            MRL_HIP(ctx, hipEventSynchronize(hp.done[s]));
            std::vector<CopyPool::Seg> out;
            if (has_eval) out.push_back({ c.out_rgb + C * off, sl.rgb, 4 * C * m });
            if (has_pdf) out.push_back({ c.out_pdf + off, sl.pdf, 4 * m });
            if (has_sample) {
                out.push_back({ c.out_wo + 3 * off, sl.wo2, 12 * m });
                out.push_back({ c.out_pdf2 + off, sl.pdf2, 4 * m });
                out.push_back({ c.out_weight + C * off, sl.w, 4 * C * m });
            }
            hp.pool.run(out);
        }
    }
    return MRL_OK;
}

int run_batch(mrl_ctx *ctx, const BatchCall &c)
{
    if (!ctx) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (c.n == 0) return MRL_OK;
    const bool has_eval = call_has_eval(c.mode), has_pdf = call_has_pdf(c.mode), has_sample = call_has_sample(c.mode);
    const bool needs_wo = has_eval || has_pdf, needs_u = has_sample;
    int rc = check_call(ctx, c);
    if (rc != MRL_OK) return rc;
    MRL_HIP(ctx, hipSetDevice(ctx->device));

    const int kind = call_pointer_kind(c);
    if (kind < 0) return fail(ctx, MRL_ERR_POINTER_MIX, "host and device pointers mixed in one call");
    if (kind == 1) return launch_device(ctx, c);

    if (ctx->host_threads > 0) {
        rc = run_host_pipelined(ctx, c);
        if (rc != MRL_ERR_OOM) return rc;                     // no pinned memory to be had: fall back to the staged path
        (void)hipGetLastError();
    }
    // host pointers: stage through HBM in chunks; returns when the outputs are on the host
    const size_t C = c.n_ch > 0 ? (size_t)c.n_ch : 3;          // values per unit in out_rgb / out_weight
    const size_t unit_bytes = 56 + 8 * C;
    const size_t chunk = std::min(c.n, ctx->host_chunk);
    rc = ensure_stage(ctx, chunk, unit_bytes);
    if (rc != MRL_OK) return rc;
    char *base = (char *)ctx->d_stage;
    const size_t cu = chunk;
    float *d_wi = (float *)base;               float *d_wo = (float *)(base + 12 * cu);
    float *d_u = (float *)(base + 24 * cu);    int32_t *d_mat = (int32_t *)(base + 32 * cu);
    float *d_pdf = (float *)(base + 36 * cu);  float *d_wo2 = (float *)(base + 40 * cu);
    float *d_pdf2 = (float *)(base + 52 * cu); float *d_rgb = (float *)(base + 56 * cu);
    float *d_w = (float *)(base + (56 + 4 * C) * cu);
    for (size_t off = 0; off < c.n; off += chunk) {
        const size_t m = std::min(chunk, c.n - off);
        MRL_HIP(ctx, hipMemcpyAsync(d_wi, c.wi + 3 * off, 12 * m, hipMemcpyHostToDevice, ctx->stream));
        if (needs_wo) MRL_HIP(ctx, hipMemcpyAsync(d_wo, c.wo + 3 * off, 12 * m, hipMemcpyHostToDevice, ctx->stream));
        if (needs_u) MRL_HIP(ctx, hipMemcpyAsync(d_u, c.u + 2 * off, 8 * m, hipMemcpyHostToDevice, ctx->stream));
        if (c.mat) MRL_HIP(ctx, hipMemcpyAsync(d_mat, c.mat + off, 4 * m, hipMemcpyHostToDevice, ctx->stream));
        BatchCall d = c;
        d.wi = d_wi; d.wo = d_wo; d.u = d_u; d.mat = c.mat ? d_mat : nullptr; d.n = m;
        d.out_rgb = d_rgb; d.out_pdf = d_pdf; d.out_wo = d_wo2; d.out_pdf2 = d_pdf2; d.out_weight = d_w;
        rc = launch_device(ctx, d);
        if (rc != MRL_OK) return rc;
        if (has_eval) MRL_HIP(ctx, hipMemcpyAsync(c.out_rgb + C * off, d_rgb, 4 * C * m, hipMemcpyDeviceToHost, ctx->stream));
        if (has_pdf) MRL_HIP(ctx, hipMemcpyAsync(c.out_pdf + off, d_pdf, 4 * m, hipMemcpyDeviceToHost, ctx->stream));
        if (has_sample) {
            MRL_HIP(ctx, hipMemcpyAsync(c.out_wo + 3 * off, d_wo2, 12 * m, hipMemcpyDeviceToHost, ctx->stream));
            MRL_HIP(ctx, hipMemcpyAsync(c.out_pdf2 + off, d_pdf2, 4 * m, hipMemcpyDeviceToHost, ctx->stream));
            MRL_HIP(ctx, hipMemcpyAsync(c.out_weight + C * off, d_w, 4 * C * m, hipMemcpyDeviceToHost, ctx->stream));
        }
        MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MRL_OK;
}

// mrl_*_queue: a caller-built queue of unit indices with a device-side length
int run_queue(mrl_ctx *ctx, const BatchCall &c, const uint32_t *queue, const uint32_t *queue_count)
{
    if (!ctx) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (c.n == 0) return MRL_OK;
    if (!queue || !queue_count) return fail(ctx, MRL_ERR_INVALID, "null array argument");
    int rc = check_call(ctx, c);
    if (rc != MRL_OK) return rc;
    if (c.n > ((size_t)1 << 32)) return fail(ctx, MRL_ERR_INVALID, "queue capacity exceeds 2^32 (indices are uint32)");
    MRL_HIP(ctx, hipSetDevice(ctx->device));
    if (call_pointer_kind(c, queue, queue_count) != 1) return fail(ctx, MRL_ERR_POINTER_MIX, "queue calls take device pointers only");
    DeviceCall d = device_call(ctx, c);
    d.args.idx = queue; d.args.idx_count = queue_count;
    if (!d.multi && (d.args.single.kind == mrl::KIND_RGL || d.args.single.kind == mrl::KIND_RGL_SPECTRAL)) {   // (spectral: pdf only, see check_call)
        MRL_HIP(ctx, mrl::launch_rgl(c.mode, d.args, &ctx->materials[(size_t)c.single_id].rgl, true, ctx->rgl_search, ctx->compute_units, ctx->stream));
        return MRL_OK;
    }
    if (c.n_ch > 0 && c.mode != 1) {                          // n-channel tables: the same kernels walk the queue
        MRL_HIP(ctx, mrl::launch_batch_nch(c.mode, d.args, d.multi, c.n_ch, ctx->compute_units, ctx->stream));
        return MRL_OK;
    }
    MRL_HIP(ctx, mrl::launch_batch_indexed(c.mode, d.args, d.multi, ctx->table_layout, d.has_ggx, d.has_table, ctx->compute_units, ctx->stream));
    if (d.multi && d.has_rgl) MRL_HIP(ctx, mrl::launch_rgl(c.mode, d.args, nullptr, true, ctx->rgl_search, ctx->compute_units, ctx->stream));
    return MRL_OK;
}

} // namespace mrlabi
using namespace mrlabi;

extern "C" {

int mrl_eval_batch(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id, size_t n, float *out_rgb)
{
    BatchCall c{ 0, wi, wo, nullptr, mat, single_id, n, out_rgb, nullptr, nullptr, nullptr, nullptr };
    return run_batch(ctx, c);
}

int mrl_pdf_batch(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id, size_t n, float *out_pdf)
{
    BatchCall c{ 1, wi, wo, nullptr, mat, single_id, n, nullptr, out_pdf, nullptr, nullptr, nullptr };
    return run_batch(ctx, c);
}

int mrl_sample_batch(mrl_ctx *ctx, const float *wi, const float *u, const int32_t *mat, int32_t single_id, size_t n,
                     float *out_wo, float *out_pdf, float *out_weight)
{
    BatchCall c{ 2, wi, nullptr, u, mat, single_id, n, nullptr, nullptr, out_wo, out_pdf, out_weight };
    return run_batch(ctx, c);
}

int mrl_eval_sample_batch(mrl_ctx *ctx, const float *wi, const float *wo, const float *u, const int32_t *mat, int32_t single_id,
                          size_t n, float *out_rgb, float *out_pdf, float *out_wo, float *out_pdf2, float *out_weight)
{
    BatchCall c{ 3, wi, wo, u, mat, single_id, n, out_rgb, out_pdf, out_wo, out_pdf2, out_weight };
    return run_batch(ctx, c);
}

int mrl_eval_pdf_batch(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id,
                       size_t n, float *out_rgb, float *out_pdf)
{
    BatchCall c{ 4, wi, wo, nullptr, mat, single_id, n, out_rgb, out_pdf, nullptr, nullptr, nullptr };
    return run_batch(ctx, c);
}

int mrl_partition_by_material(mrl_ctx *ctx, const int32_t *mat, size_t n, uint32_t *queue_out, uint32_t *offsets_out, uint32_t *counts_out)
{
    if (!ctx) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (!offsets_out || !counts_out || (n > 0 && (!mat || !queue_out))) return fail(ctx, MRL_ERR_INVALID, "null array argument");
    if (n > ((size_t)1 << 32)) return fail(ctx, MRL_ERR_INVALID, "more than 2^32 slots (queue entries are uint32)");
    const int K = (int)ctx->materials.size();
    if (K == 0) return fail(ctx, MRL_ERR_MATERIAL, "no material loaded");
    if (K > mrl::kMaxPartitionMaterials) return fail(ctx, MRL_ERR_INVALID, "too many materials for the partition kernel");
    MRL_HIP(ctx, hipSetDevice(ctx->device));
    if (common_kind({ mat, queue_out, offsets_out, counts_out }) != 1) return fail(ctx, MRL_ERR_POINTER_MIX, "partition takes device pointers only");
    if (n == 0) {                                     // nothing to partition: every group is empty
        MRL_HIP(ctx, hipMemsetAsync(offsets_out, 0, ((size_t)K + 1) * sizeof(uint32_t), ctx->stream));
        MRL_HIP(ctx, hipMemsetAsync(counts_out, 0, (size_t)K * sizeof(uint32_t), ctx->stream));
        return MRL_OK;
    }
    uint32_t chunks = 0, chunk_len = 0;
    mrl::material_partition_geometry(n, ctx->compute_units, &chunks, &chunk_len);
    const size_t need = (size_t)chunks * K + K;
    if (need > ctx->part_work_cap) {
        if (ctx->d_part_work) { MRL_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_part_work); ctx->d_part_work = nullptr; ctx->part_work_cap = 0; }
        MRL_ALLOC(ctx, hipMalloc((void **)&ctx->d_part_work, need * sizeof(uint32_t)));
        ctx->part_work_cap = need;
    }
    MRL_HIP(ctx, mrl::launch_partition_materials(mat, n, K, queue_out, offsets_out, counts_out, ctx->d_part_work, chunks, chunk_len,
                                                 ctx->compute_units, ctx->stream));
    return MRL_OK;
}

int mrl_eval_pdf_queue(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id,
                       const uint32_t *queue, const uint32_t *queue_count, size_t capacity, float *out_rgb, float *out_pdf)
{
    BatchCall c = { 4, wi, wo, nullptr, mat, single_id, capacity, out_rgb, out_pdf, nullptr, nullptr, nullptr };
    return run_queue(ctx, c, queue, queue_count);
}

int mrl_eval_queue(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id,
                   const uint32_t *queue, const uint32_t *queue_count, size_t capacity, float *out_rgb)
{
    BatchCall c = { 0, wi, wo, nullptr, mat, single_id, capacity, out_rgb, nullptr, nullptr, nullptr, nullptr };
    return run_queue(ctx, c, queue, queue_count);
}

int mrl_pdf_queue(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id,
                  const uint32_t *queue, const uint32_t *queue_count, size_t capacity, float *out_pdf)
{
    BatchCall c = { 1, wi, wo, nullptr, mat, single_id, capacity, nullptr, out_pdf, nullptr, nullptr, nullptr };
    return run_queue(ctx, c, queue, queue_count);
}

int mrl_sample_queue(mrl_ctx *ctx, const float *wi, const float *u, const int32_t *mat, int32_t single_id,
                     const uint32_t *queue, const uint32_t *queue_count, size_t capacity,
                     float *out_wo, float *out_pdf, float *out_weight)
{
    BatchCall c = { 2, wi, nullptr, u, mat, single_id, capacity, nullptr, nullptr, out_wo, out_pdf, out_weight };
    return run_queue(ctx, c, queue, queue_count);
}

int mrl_eval_sample_queue(mrl_ctx *ctx, const float *wi, const float *wo, const float *u,
                          const int32_t *mat, int32_t single_id,
                          const uint32_t *queue, const uint32_t *queue_count, size_t capacity,
                          float *out_rgb, float *out_pdf, float *out_wo, float *out_pdf2, float *out_weight)
{
    BatchCall c = { 3, wi, wo, u, mat, single_id, capacity, out_rgb, out_pdf, out_wo, out_pdf2, out_weight };
    return run_queue(ctx, c, queue, queue_count);
}

int mrl_generate_pairs(mrl_ctx *ctx, uint64_t seed, uint64_t first_index, size_t n, float *wi, float *wo, float *u)
{
    if (!ctx) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (!wi || !wo || !u) return fail(ctx, MRL_ERR_INVALID, "null argument");
    MRL_HIP(ctx, hipSetDevice(ctx->device));
    if (common_kind({ wi, wo, u }) != 1) return fail(ctx, MRL_ERR_INVALID, "generator needs device pointers");
    MRL_HIP(ctx, mrl::launch_generate_pairs(seed, first_index, n, wi, wo, u, ctx->compute_units, ctx->stream));
    return MRL_OK;
}

int mrl_generate_materials(mrl_ctx *ctx, uint64_t seed, uint64_t first_index, size_t n, int n_materials, int32_t *mat)
{
    if (!ctx) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (!mat || n_materials < 1) return fail(ctx, MRL_ERR_INVALID, "bad argument");
    MRL_HIP(ctx, hipSetDevice(ctx->device));
    if (pointer_kind(mat) != 1) return fail(ctx, MRL_ERR_INVALID, "generator needs device pointers");
    MRL_HIP(ctx, mrl::launch_generate_materials(seed, first_index, n, n_materials, mat, ctx->compute_units, ctx->stream));
    return MRL_OK;
}

static int nch_call(mrl_ctx *ctx, BatchCall c, int n_channels)
{
    if (!ctx) return MRL_ERR_INVALID;
    if (n_channels < 1 || n_channels > mrl::kMaxChannels) return fail(ctx, MRL_ERR_INVALID, "channel count must be 1.." + std::to_string(mrl::kMaxChannels));
    c.n_ch = n_channels == 3 ? 0 : n_channels;           // three channels: the RGB path, RGB materials
    return run_batch(ctx, c);
}

static int nch_queue_call(mrl_ctx *ctx, BatchCall c, int n_channels, const uint32_t *queue, const uint32_t *queue_count)
{
    if (!ctx) return MRL_ERR_INVALID;
    if (n_channels < 1 || n_channels > mrl::kMaxChannels) return fail(ctx, MRL_ERR_INVALID, "channel count must be 1.." + std::to_string(mrl::kMaxChannels));
    c.n_ch = n_channels == 3 ? 0 : n_channels;
    return run_queue(ctx, c, queue, queue_count);
}

int mrl_eval_queue_nch(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id, const uint32_t *queue,
                       const uint32_t *queue_count, size_t capacity, int n_channels, float *out_values)
{
    BatchCall c{ 0, wi, wo, nullptr, mat, single_id, capacity, out_values, nullptr, nullptr, nullptr, nullptr };
    return nch_queue_call(ctx, c, n_channels, queue, queue_count);
}

int mrl_sample_queue_nch(mrl_ctx *ctx, const float *wi, const float *u, const int32_t *mat, int32_t single_id, const uint32_t *queue,
                         const uint32_t *queue_count, size_t capacity, int n_channels, float *out_wo, float *out_pdf, float *out_weight)
{
    BatchCall c{ 2, wi, nullptr, u, mat, single_id, capacity, nullptr, nullptr, out_wo, out_pdf, out_weight };
    return nch_queue_call(ctx, c, n_channels, queue, queue_count);
}

int mrl_eval_pdf_queue_nch(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id, const uint32_t *queue,
                           const uint32_t *queue_count, size_t capacity, int n_channels, float *out_values, float *out_pdf)
{
    BatchCall c{ 4, wi, wo, nullptr, mat, single_id, capacity, out_values, out_pdf, nullptr, nullptr, nullptr };
    return nch_queue_call(ctx, c, n_channels, queue, queue_count);
}

int mrl_eval_sample_queue_nch(mrl_ctx *ctx, const float *wi, const float *wo, const float *u, const int32_t *mat, int32_t single_id,
                              const uint32_t *queue, const uint32_t *queue_count, size_t capacity, int n_channels,
                              float *out_values, float *out_pdf, float *out_wo, float *out_pdf2, float *out_weight)
{
    BatchCall c{ 3, wi, wo, u, mat, single_id, capacity, out_values, out_pdf, out_wo, out_pdf2, out_weight };
    return nch_queue_call(ctx, c, n_channels, queue, queue_count);
}

int mrl_eval_batch_nch(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id, size_t n, int n_channels,
                       float *out_values)
{
    BatchCall c{ 0, wi, wo, nullptr, mat, single_id, n, out_values, nullptr, nullptr, nullptr, nullptr };
    return nch_call(ctx, c, n_channels);
}

int mrl_sample_batch_nch(mrl_ctx *ctx, const float *wi, const float *u, const int32_t *mat, int32_t single_id, size_t n, int n_channels,
                         float *out_wo, float *out_pdf, float *out_weight)
{
    BatchCall c{ 2, wi, nullptr, u, mat, single_id, n, nullptr, nullptr, out_wo, out_pdf, out_weight };
    return nch_call(ctx, c, n_channels);
}

int mrl_eval_pdf_batch_nch(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id, size_t n, int n_channels,
                           float *out_values, float *out_pdf)
{
    BatchCall c{ 4, wi, wo, nullptr, mat, single_id, n, out_values, out_pdf, nullptr, nullptr, nullptr };
    return nch_call(ctx, c, n_channels);
}

int mrl_eval_sample_batch_nch(mrl_ctx *ctx, const float *wi, const float *wo, const float *u, const int32_t *mat, int32_t single_id, size_t n,
                              int n_channels, float *out_values, float *out_pdf, float *out_wo, float *out_pdf2, float *out_weight)
{
    BatchCall c{ 3, wi, wo, u, mat, single_id, n, out_values, out_pdf, out_wo, out_pdf2, out_weight };
    return nch_call(ctx, c, n_channels);
}

} // extern "C"
