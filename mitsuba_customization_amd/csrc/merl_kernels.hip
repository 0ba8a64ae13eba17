// merl_kernels.hip — gfx950 kernels of the batched BSDF hot path and their launchers.
//
// One lane = one unit (pair) per step.  Streams (wi, wo, u in; rgb, pdf, wo', pdf', weight out) are
// contiguous per wave: 768 B (xyz) / 512 B (uv) / 256 B (scalar) per wave-instruction.
//   k_batch        generic: every material kind, ocml f64 math (MRL_OPT_KERNEL 0, the A/B baseline)
//   k_table        tuned f64 math, one lane gathers its own texels (variants 1/2; nearest lookups, rows layout)
//   k_table_dma    brick layout + cooperative LDS-DMA fetch of the neighbourhood (variant 3, the default)
//   k_ggx          tuned analytic GGX rough conductor
//   k_count_kinds / k_scan_segments / k_partition_kinds   ballot/prefix partition of kind-mixed batches (variant 4)
//   k_build_bricks / k_build_rows   table re-layout at upload;  k_generate_*   synthetic inputs
// What bounds each of them, with counter evidence: DESIGN.md §5-6.
#include "merl_kernels.hpp"
#include "merl_table_fast.hpp"
#include "merl_ggx_fast.hpp"

namespace mrl {

namespace {

constexpr int kBlock = 256;

enum Mode : int { MODE_EVAL = 0, MODE_PDF = 1, MODE_SAMPLE = 2, MODE_EVAL_SAMPLE = 3, MODE_EVAL_PDF = 4 };
// what a mode computes: eval(wi, wo) -> rgb, pdf(wi, wo), sample(wi, u) -> (wo', pdf', weight')
constexpr bool mode_eval(int m) { return m == MODE_EVAL || m == MODE_EVAL_SAMPLE || m == MODE_EVAL_PDF; }
constexpr bool mode_pdf(int m) { return m == MODE_PDF || m == MODE_EVAL_SAMPLE || m == MODE_EVAL_PDF; }
constexpr bool mode_sample(int m) { return m == MODE_SAMPLE || m == MODE_EVAL_SAMPLE; }

__device__ __forceinline__ void load3(const float *p, size_t i, float &x, float &y, float &z)
{
    const float *q = p + 3 * i;
    x = q[0]; y = q[1]; z = q[2];
}
__device__ __forceinline__ void store3(float *p, size_t i, const float v[3])
{
    float *q = p + 3 * i;
    q[0] = v[0]; q[1] = v[1]; q[2] = v[2];
}
// streaming (read-once / write-once) accesses: the nt hint keeps them from displacing table
// lines in the XCD's L2
template <bool NT> __device__ __forceinline__ float ldf(const float *p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }
template <bool NT> __device__ __forceinline__ void stf(float *p, float v) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT> __device__ __forceinline__ void load3s(const float *p, size_t i, float &x, float &y, float &z)
{
    const float *q = p + 3 * i;
    x = ldf<NT>(q); y = ldf<NT>(q + 1); z = ldf<NT>(q + 2);
}
template <bool NT> __device__ __forceinline__ void store3s(float *p, size_t i, const float v[3])
{
    float *q = p + 3 * i;
    stf<NT>(q, v[0]); stf<NT>(q + 1, v[1]); stf<NT>(q + 2, v[2]);
}

// number of work items of a launch: a.n, or for a queue launch the device-side count clamped to the capacity a.n
template <bool INDEXED> __device__ __forceinline__ size_t item_count(const BatchArgs &a)
{
    if constexpr (INDEXED) { const size_t c = (size_t)*a.idx_count; return c < a.n ? c : a.n; }
    else return a.n;
}

// INDEXED: walk the queue a.idx[0 .. *a.idx_count) of unit indices instead of [0, n)
template <int MODE, bool MULTI, bool INDEXED = false>
__global__ __launch_bounds__(kBlock) void k_batch(BatchArgs a)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    const size_t n_items = item_count<INDEXED>(a);
    for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < n_items; j += stride) {
        const size_t i = INDEXED ? (size_t)a.idx[j] : j;
        MaterialDev m;
        bool valid = true;
        if constexpr (MULTI) {
            int id = a.mat[i];
            valid = id >= 0 && id < a.n_materials;
            m = a.materials[valid ? id : 0];
            valid = valid && (kind_is_rgb_path(m.kind) || (MODE == MODE_PDF && m.kind == KIND_TABLE_NCH));
            if (!valid) m = a.safe;
        } else {
            m = a.single;
        }
        float wix, wiy, wiz;
        load3(a.wi, i, wix, wiy, wiz);
        if (!valid) wiz = 0.0f;                     // unknown material id: every output zero

        if constexpr (mode_eval(MODE) || mode_pdf(MODE)) {
            float wox, woy, woz;
            load3(a.wo, i, wox, woy, woz);
            if constexpr (mode_eval(MODE)) {
                float rgb[3];
                unit_eval(m, a.opts, wix, wiy, wiz, wox, woy, woz, rgb);
                store3(a.out_rgb, i, rgb);
            }
            if constexpr (mode_pdf(MODE)) {
                a.out_pdf[i] = unit_pdf(m, a.opts, wix, wiy, wiz, wox, woy, woz);
            }
        }
        if constexpr (mode_sample(MODE)) {
            float u0 = a.u[2 * i], u1 = a.u[2 * i + 1];
            float wo2[3], pdf2, w[3];
            unit_sample(m, a.opts, wix, wiy, wiz, u0, u1, wo2, pdf2, w);
            store3(a.out_wo, i, wo2);
            a.out_pdf2[i] = pdf2;
            store3(a.out_weight, i, w);
        }
    }
}

// ---- variant 1: tuned table path (merl_table_fast.hpp); GGX lanes of a mixed batch take the generic functions ----
template <int MODE, bool MULTI, bool NT, int LOOKUP, int LAYOUT>
__global__ __launch_bounds__(kBlock) void k_table(BatchArgs a)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += stride) {
        MaterialDev m;
        bool known = true;
        if constexpr (MULTI) {
            int id = a.mat[i];
            known = id >= 0 && id < a.n_materials;
            m = a.materials[known ? id : 0];
            known = known && (kind_is_rgb_path(m.kind) || (MODE == MODE_PDF && m.kind == KIND_TABLE_NCH));
            if (!known) m = a.safe;
        } else {
            m = a.single;
        }
        float wix, wiy, wiz;
        load3s<NT>(a.wi, i, wix, wiy, wiz);
        if (!known) wiz = 0.0f;
        float wox = 0.0f, woy = 0.0f, woz = 1.0f, u0 = 0.0f, u1 = 0.0f;
        if constexpr (mode_eval(MODE) || mode_pdf(MODE)) load3s<NT>(a.wo, i, wox, woy, woz);
        if constexpr (mode_sample(MODE)) { u0 = ldf<NT>(a.u + 2 * i); u1 = ldf<NT>(a.u + 2 * i + 1); }

        float rgb[3], pdf = 0.0f, wo2[3], pdf2, w[3];
        if (MULTI && m.kind == KIND_GGX) {
            if constexpr (mode_eval(MODE)) unit_eval(m, a.opts, wix, wiy, wiz, wox, woy, woz, rgb);
            if constexpr (mode_pdf(MODE)) pdf = unit_pdf(m, a.opts, wix, wiy, wiz, wox, woy, woz);
            if constexpr (mode_sample(MODE)) unit_sample(m, a.opts, wix, wiy, wiz, u0, u1, wo2, pdf2, w);
        } else {
            if constexpr (MODE == MODE_PDF) {
                pdf = (wiz > 0.0f && woz > 0.0f) ? woz * kInvPiF : 0.0f;
                if (a.opts.sampling && pdf > 0.0f)
                    pdf = (float)fast::table_pdf(m, fast::normalize_f32(wix, wiy, wiz), fast::normalize_f32(wox, woy, woz), woz, a.opts.sampling);
            } else {
                const fast::Vec3 in = fast::normalize_f32(wix, wiy, wiz);
                if constexpr (mode_eval(MODE)) fast::unit_eval<LOOKUP, LAYOUT>(m, a.opts, in, wix, wiy, wiz, wox, woy, woz, rgb);
                if constexpr (mode_pdf(MODE)) {
                    pdf = (wiz > 0.0f && woz > 0.0f) ? woz * kInvPiF : 0.0f;
                    if (a.opts.sampling && pdf > 0.0f) pdf = (float)fast::table_pdf(m, in, fast::normalize_f32(wox, woy, woz), woz, a.opts.sampling);
                }
                if constexpr (mode_sample(MODE)) fast::unit_sample<LOOKUP, LAYOUT>(m, a.opts, in, wix, wiy, wiz, u0, u1, wo2, pdf2, w);
            }
        }
        if constexpr (mode_eval(MODE)) store3s<NT>(a.out_rgb, i, rgb);
        if constexpr (mode_pdf(MODE)) stf<NT>(a.out_pdf + i, pdf);
        if constexpr (mode_sample(MODE)) {
            store3s<NT>(a.out_wo, i, wo2);
            stf<NT>(a.out_pdf2 + i, pdf2);
            store3s<NT>(a.out_weight, i, w);
        }
    }
}

// ---- variant 3: brick layout + cooperative LDS-DMA fetch of the interpolation neighbourhood ----------
// With one lane per unit, every lane's six 16-B loads of its brick hit a different 128-B line: the CU's
// texture addresser (TCP tag lookup, one lane-address per clock) becomes the limit (TA_BUSY 84 %,
// profiles/r01_v2_brick_pmc_summary.json).  Here the wave fetches its 64 bricks cooperatively instead:
// in step k lanes 8g..8g+7 copy the 128-B brick of unit 8k+g with ONE global_load_lds_dwordx4 (16 B per
// lane, the eight lanes cover the whole line: 2 tag lookups instead of 6-8), straight into LDS with no
// VGPR staging.  Brick addresses travel between lanes by ds_bpermute.  The LDS image of a DMA is
// lane-linear, so the 16-B piece a lane copies is XOR-swizzled on the SOURCE side
// (piece = (lane&7) ^ ((unit>>1)&7)); unit j then reads its piece p at slot 8j + (p ^ ((j>>1)&7)),
// which is bank-conflict-free for ds_read_b128's 16-lane groups.
#ifndef MRL_DMA_BLOCK
#define MRL_DMA_BLOCK 256
#endif
constexpr int kDmaBlock = MRL_DMA_BLOCK;
// blocks of the LDS-DMA kernel one CU holds: 8 KB of LDS per wave and lookup (+ the exchange pages) out of 160 KB
constexpr int dma_blocks_per_cu(int mode)
{
    const int lookups = (mode == 3 /* eval + sample */) ? 2 : 1;
    const int per_block = (kDmaBlock / 64) * lookups * (8192 + 512);
    return (160 * 1024) / per_block;
}

__device__ __forceinline__ unsigned brick_swz(unsigned unit) { return (unit >> 1) & 7u; }

// Source address of the 16-B piece this lane copies in step k (unit 8k + lane/8).  The 64 cell indices of the wave
// change hands through a wave-private LDS page: lane L = 8k + g stores its index at dword g*8 + k, and lane-group g then
// reads the eight indices it copies for (units g, 8 + g, ..., 56 + g) back as TWO ds_read_b128 of dwords g*8 .. g*8 + 7.
// One ds_write_b32 + two ds_read_b128 per lookup instead of eight ds_bpermute_b32 (24 cycles each on the CU's one LDS
// pipe, which this kernel keeps 60 % busy: tools/microbench/valu_issue.hip, profiles/r03_valu_issue.json).  Mixed batches
// exchange the 64-bit brick address the same way (ds_write_b64, four ds_read_b128).  LDS operations of one wave execute
// in order, so the reads see the writes; the wavefront fence keeps the compiler from moving them.
template <bool MULTI>
struct BrickSources {
    const float4 *src[8];          // MULTI: the 16-B piece this lane copies in step k
    uint32_t cell[8];              // single material: the cell index of step k's unit
    uint32_t piece_bytes[2];       // ... and this lane's piece offset inside a brick, for even / odd k
    __device__ __forceinline__ BrickSources(uint32_t idx, const float4 *lane_base, unsigned lane, uint32_t *page)
    {
        const unsigned g = lane >> 3, k_own = lane >> 3, g_own = lane & 7u;
        const unsigned piece_lane = lane & 7u;
        if constexpr (MULTI) {
            uint64_t *page64 = (uint64_t *)page;
            page64[g_own * 8u + k_own] = (uint64_t)(lane_base + (size_t)idx * 8);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const ulonglong2 *rd = (const ulonglong2 *)(page64 + g * 8u);
            const ulonglong2 a0 = rd[0], a1 = rd[1], a2 = rd[2], a3 = rd[3];
            const uint64_t got[8] = { a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a3.y };
#pragma unroll
            for (int k = 0; k < 8; ++k) src[k] = (const float4 *)got[k] + (piece_lane ^ brick_swz(8u * k + g));
        } else {
            page[g_own * 8u + k_own] = idx;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint4 *rd = (const uint4 *)(page + g * 8u);
            const uint4 a0 = rd[0], a1 = rd[1];
            cell[0] = a0.x; cell[1] = a0.y; cell[2] = a0.z; cell[3] = a0.w; cell[4] = a1.x; cell[5] = a1.y; cell[6] = a1.z; cell[7] = a1.w;
            // brick_swz(8k + g) = (4k + (g >> 1)) & 7: two values, by the parity of k
            piece_bytes[0] = (piece_lane ^ brick_swz(g)) * 16u;
            piece_bytes[1] = (piece_lane ^ brick_swz(8u + g)) * 16u;
        }
    }
    // offsets32: the table is smaller than 4 GB (wave-uniform).  The source address of a copy is then the table's base in
    // SGPRs plus a 32-bit lane offset — ONE vector instruction per copy (cell << 7 | piece) and the scalar-base form of
    // global_load_lds — where the general form spends three on 64-bit shifts and adds.
    __device__ __forceinline__ void copy_to(float4 *lds_slots, const float4 *single_base, bool offsets32) const
    {
        if constexpr (MULTI) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src[k],
                                                 (__attribute__((address_space(3))) void *)(lds_slots + k * 64), 16, 0, 0);
        } else if (offsets32) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t off = (cell[k] << 7) + piece_bytes[k & 1];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)single_base + off),
                                                 (__attribute__((address_space(3))) void *)(lds_slots + k * 64), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint64_t off = ((uint64_t)cell[k] << 7) + piece_bytes[k & 1];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)single_base + off),
                                                 (__attribute__((address_space(3))) void *)(lds_slots + k * 64), 16, 0, 0);
            }
        }
    }
};

struct BrickWeights { double fh, fd, fp; };

// a3 tail for the brick layout: cell index + interpolation fractions
// STD = false: every table of the launch is in MERL's half/diff form (periodic phi_d), the branch is compiled out
template <bool STD>
__device__ __forceinline__ uint32_t brick_cell(const MaterialDev &m, const Coords &c, int node, BrickWeights &w)
{
#pragma clang fp contract(off)
    const double shift = node ? 0.5 : 0.0;
    int h0, d0, p0;
    split_clamped(c.xh - shift, m.n_th, h0, w.fh);
    split_clamped(c.xd - shift, m.n_td, d0, w.fd);
    if constexpr (STD) split_phi(param_phi_periodic(m.param), c.xp - shift, m.n_pd, p0, w.fp);
    else split_periodic(c.xp - shift, m.n_pd, p0, w.fp);
    return (uint32_t)((h0 * m.n_td + d0) * m.n_pd + p0);
}

// POLICY = false: the plain blend, no branch (the headline instantiation); true: MRL_OPT_NEGATIVE's policy (wave-uniform)
template <bool POLICY>
__device__ __forceinline__ Rgbf brick_interp(const float4 *lds_slots, unsigned lane, const BrickWeights &w, int blend)
{
    const unsigned f = brick_swz(lane);
    const float4 *q = lds_slots + 8u * lane;
    const float4 q0 = q[0u ^ f], q1 = q[1u ^ f], q2 = q[2u ^ f], q3 = q[3u ^ f], q4 = q[4u ^ f], q5 = q[5u ^ f];
    if constexpr (POLICY) return blend_brick_as(blend, q0, q1, q2, q3, q4, q5, corner_weights(w.fh, w.fd, w.fp));
    else return blend_brick(q0, q1, q2, q3, q4, q5, corner_weights(w.fh, w.fd, w.fp));
}

// One unit's registers while it travels through the DMA kernels.
struct UnitIO {
    float wix, wiy, wiz, wox, woy, woz, u0, u1;
    float wi_sum;                                             // wix + wiy + wiz (NaN / inf probe shared by the unit's lookups)
    float rgb[3], pdf, wo2[3], pdf2, w[3];
};

// The stream inputs of one tile as requested from memory (k_table_dma's one-tile-ahead pipeline)
struct TileIn {
    float wix, wiy, wiz, wox, woy, woz, u0, u1;
    int id;                                                   // material id (mixed batches)
    uint32_t active;                                          // 0: a lane beyond the end of the batch (computes, stores nothing)
    size_t first, i;                                          // wave-uniform first unit + the lane's offset / queued unit index
};                                                            // no padding: the struct lives in registers (SROA), never in memory

// Table lanes of one wave: transform, cooperative brick copy, blend.  EVERY lane of the wave must call
// it (the copy is wave-wide); lanes whose material is not a table pass is_table = false, take part with
// a harmless source and keep their outputs untouched.
// STD: the launch may meet tables in one of the standard parameterisations (BatchArgs::any_standard) or runs under
// MRL_OPT_NEGATIVE = 2 (the renormalising blend); without it the kernel is the half/diff-only code with the plain blend
// (A/B on one box: the wave-uniform parameterisation branch costs the headline launch 0.5 %).
template <int MODE, bool MULTI, bool GGX, bool STD>
__device__ __forceinline__ void table_lanes(const BatchArgs &a, const MaterialDev &m, bool is_table, UnitIO &io,
                                            const fast::Vec3 &in, float4 *ldsA, float4 *ldsB, uint32_t *pageA, uint32_t *pageB, unsigned lane)
{
    constexpr bool HAS_EVAL = mode_eval(MODE);
    constexpr bool HAS_SAMPLE = mode_sample(MODE);
    // single material: bricks below 4 GB take 32-bit source offsets (wave-uniform; loop-invariant)
    const bool offsets32 = !MULTI && (size_t)m.n_th * m.n_td * m.n_pd <= ((size_t)1 << 25);
    // a valid 128-B source for lanes without a table: the material array itself, cell 0
    const float4 *lane_base = (GGX && !is_table) ? (const float4 *)a.materials : m.texels;
    const fast::TableMaps maps = STD ? fast::TableMaps(m) : fast::TableMaps(m.n_th, m.n_td, m.n_pd, PARAM_HALF_DIFF);   // STD = false never reads m.param
    BrickWeights wA, wB;
    uint32_t cellA = 0, cellB = 0;
    float sx = 0.0f, sy = 0.0f, sz = 1.0f;
    // Per lookup: transform -> address exchange -> the 8 copies.  The eval lookup's copies are in flight while the sample
    // lookup is transformed (its L2 / fabric latency hides behind ~250 VALU instructions of this wave, not only behind the
    // SIMD's other wave): on the bench's random inputs 1.5 % faster than exchanging and copying both lookups at the end.
    if constexpr (HAS_EVAL) {
        const fast::Dir out = fast::dir_f32(io.wox, io.woy, io.woz);
        cellA = brick_cell<STD>(m, STD ? maps(in, out) : fast::coords(in, out, maps.k_th, maps.k_td, maps.k_pd), a.opts.node, wA);
        if (GGX && !is_table) cellA = 0;
        const BrickSources<MULTI> srcA(cellA, lane_base, lane, pageA);
        srcA.copy_to(ldsA, m.texels, offsets32);
        __builtin_amdgcn_sched_barrier(0);
    }
    float sp = 0.0f;                                          // pdf of the sampled direction
    if constexpr (HAS_SAMPLE) {
        if (a.opts.sampling && (!GGX || is_table)) {          // option is wave-uniform
            fast::table_sample_dir(m, a.opts.disk_map, in, io.u0, io.u1, sx, sy, sz, a.opts.sampling);
            const bool up = sz > 0.0f;
            if (!up) { sx = 0.0f; sy = 0.0f; sz = 1.0f; }     // rejected: look up a harmless cell, report zeros
            sp = up ? (float)fast::table_pdf(m, in, fast::normalize_f32(sx, sy, sz), sz, a.opts.sampling) : 0.0f;
        } else {
            square_to_cosine_hemisphere(a.opts.disk_map, io.u0, io.u1, sx, sy, sz);
            sp = sz > 0.0f ? sz * kInvPiF : 0.0f;
        }
        const fast::Dir out = fast::dir_f32(sx, sy, sz);
        cellB = brick_cell<STD>(m, STD ? maps(in, out) : fast::coords(in, out, maps.k_th, maps.k_td, maps.k_pd), a.opts.node, wB);
        if (GGX && !is_table) cellB = 0;
        const BrickSources<MULTI> srcB(cellB, lane_base, lane, pageB);
        srcB.copy_to(ldsB, m.texels, offsets32);
    }
    // No explicit wait: the compiler tracks the LDS-DMA copies per LDS array (the eval and the sample lookup have their
    // own __shared__ arrays), so the blend of the eval lookup waits for ITS eight copies only (s_waitcnt vmcnt(8): the sample
    // lookup's are still in flight) and the sample blend for the rest.  Own wave only: no barrier.

    if constexpr (HAS_EVAL) {
        const Rgbf v = brick_interp<STD>(ldsA, lane, wA, blend_of(a.opts));
        if (!GGX || is_table) {
            fast::eval_tail(v, io.wi_sum, io.wiz, io.wox, io.woy, io.woz, io.rgb, a.opts.cosine != 0);
            if constexpr (mode_pdf(MODE)) {
                const bool valid = (io.wiz > 0.0f) && (io.woz > 0.0f);
                io.pdf = valid ? io.woz * kInvPiF : 0.0f;
                if (a.opts.sampling && valid) io.pdf = (float)fast::table_pdf(m, in, fast::normalize_f32(io.wox, io.woy, io.woz), io.woz, a.opts.sampling);
            }
        }
    }
    if constexpr (HAS_SAMPLE) {
        const Rgbf v = brick_interp<STD>(ldsB, lane, wB, blend_of(a.opts));
        if (!GGX || is_table) fast::sample_tail(v, io.wi_sum, io.wiz, sx, sy, sz, sp, a.opts.sampling != 0, io.wo2, io.pdf2, io.w, a.opts.cosine != 0);
    }
}

// One analytic (GGX) lane: tuned functions of merl_ggx_fast.hpp
template <int MODE>
__device__ __forceinline__ void ggx_lane(const MaterialDev &m, UnitIO &io, const fast::Vec3 &in)
{
    constexpr bool HAS_EVAL = mode_eval(MODE);
    constexpr bool HAS_SAMPLE = mode_sample(MODE);
    const fast::GgxConsts g(m);
    if constexpr (HAS_EVAL) {
        const fast::Vec3 out = fast::normalize_f32(io.wox, io.woy, io.woz);
        double v[3], p;
        fast::ggx_eval_pdf(g, in, out, v, p);
        const bool valid = (io.wiz > 0.0f) && (io.woz > 0.0f);
        const double poison = fast::cos_or_nan(io.wix, io.wiy, io.wiz, io.wox, io.woy, 1.0f);
        io.rgb[0] = valid ? (float)(v[0] * poison) : 0.0f; io.rgb[1] = valid ? (float)(v[1] * poison) : 0.0f;
        io.rgb[2] = valid ? (float)(v[2] * poison) : 0.0f;
        if constexpr (mode_pdf(MODE)) io.pdf = valid ? (float)(p * poison) : 0.0f;
    }
    if constexpr (HAS_SAMPLE) {
        fast::ggx_sample(g, in, io.u0, io.u1, io.wo2, io.pdf2, io.w);
        if (!(io.wiz > 0.0f)) { io.wo2[0] = io.wo2[1] = io.wo2[2] = 0.0f; io.pdf2 = 0.0f; io.w[0] = io.w[1] = io.w[2] = 0.0f; }
    }
}

// `first` is wave-uniform (the tile's first unit; 0 for queue launches), `i` the lane's offset from it: the stores then take a
// scalar base and a 32-bit lane offset
template <int MODE, bool NT>
__device__ __forceinline__ void store_unit(const BatchArgs &a, size_t first, size_t i, const UnitIO &io)
{
    constexpr bool HAS_EVAL = mode_eval(MODE);
    constexpr bool HAS_SAMPLE = mode_sample(MODE);
    if constexpr (HAS_EVAL) store3s<NT>(a.out_rgb + 3 * first, i, io.rgb);
    if constexpr (mode_pdf(MODE)) stf<NT>(a.out_pdf + first + i, io.pdf);
    if constexpr (HAS_SAMPLE) {
        store3s<NT>(a.out_wo + 3 * first, i, io.wo2);
        stf<NT>(a.out_pdf2 + first + i, io.pdf2);
        store3s<NT>(a.out_weight + 3 * first, i, io.w);
    }
}

// GGX: the batch may contain analytic (GGX) materials next to table materials; their lanes run the tuned
// GGX functions after the table lanes, under divergence (a mixed wave pays for both paths; kind-uniform
// waves skip the other path through wave-uniform branches).  GGX = false drops that code from the kernel.
// The alternative — partitioning the batch into per-kind queues first, MRL_OPT_KERNEL 4 — is measured in
// it wins only when most units are analytic, because a sparse queue reads whole 128-B lines
// of the stream arrays for 12 B of payload.
// INDEXED: the kernel walks a list of unit indices (one kind's queue built by k_partition_kinds) instead of [0, n).
template <int MODE, bool MULTI, bool NT, bool GGX, bool INDEXED = false, bool STD = false>
__global__ __launch_bounds__(kDmaBlock) void k_table_dma(BatchArgs a)
{
    static_assert(MODE != MODE_PDF, "pdf needs no table");
    static_assert(MULTI || !GGX, "a single-material GGX launch uses k_ggx");
    constexpr bool HAS_EVAL = mode_eval(MODE);
    constexpr bool HAS_SAMPLE = mode_sample(MODE);
    constexpr int LOOKUPS = (HAS_EVAL ? 1 : 0) + (HAS_SAMPLE ? 1 : 0);
    // 8 KB per wave and lookup; one array per lookup, so that the compiler's LDS-DMA tracking tells them apart
    __shared__ float4 lds_eval[HAS_EVAL ? kDmaBlock / 64 : 1][HAS_EVAL ? 512 : 1];
    __shared__ float4 lds_sample[HAS_SAMPLE ? kDmaBlock / 64 : 1][HAS_SAMPLE ? 512 : 1];
    __shared__ uint32_t pages[kDmaBlock / 64][LOOKUPS][MULTI ? 128 : 64];   // address exchange (BrickSources)

    // the wave index in an SGPR: LDS bases (M0 of the copies) and the tile's stream addresses are then scalar work
    const unsigned lane = threadIdx.x & 63u, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float4 *ldsA = lds_eval[HAS_EVAL ? wave : 0];
    float4 *ldsB = lds_sample[HAS_SAMPLE ? wave : 0];
    uint32_t *pageA = pages[wave][0], *pageB = pages[wave][LOOKUPS - 1];
    const size_t n_items = item_count<INDEXED>(a);
    // Block -> tile map.  Workgroups are dealt round-robin to the 8 XCDs (blocks b and b + 8 share one; observed, used
    // for speed only).  a.block_map = 1: XCD x walks ITS contiguous eighth of the batch, so units that are neighbours in
    // the batch — neighbouring pixels and scanlines of a render — meet in one XCD's L2 instead of being fetched by all
    // eight; 0: the plain interleaved grid-stride walk (block b takes tiles b, b + G, ...).  Results do not depend on it.
    const size_t tiles = (n_items + kDmaBlock - 1) / kDmaBlock;
    size_t tile = blockIdx.x, tile_end = tiles, tile_step = gridDim.x;
    if (a.block_map && (gridDim.x & 7u) == 0) {
        const size_t x = blockIdx.x & 7u, per_xcd = gridDim.x >> 3;
        tile = x * tiles / 8 + (blockIdx.x >> 3);
        tile_end = (x + 1) * tiles / 8;
        tile_step = per_xcd;
    }
    // The loop is software-pipelined by one tile: the stream inputs of tile t + 1 are requested before tile t is
    // transformed, so their HBM latency (the longest single wait of an iteration) runs under a whole iteration of work.
    auto tile_live = [&](size_t t) { return t < tile_end && t * kDmaBlock + wave * 64u < n_items; };      // wave-uniform
    auto request = [&](size_t t) {
        TileIn r;
        const size_t base = t * kDmaBlock + wave * 64u;
        r.active = base + lane < n_items ? 1u : 0u;
        // `first` + `i`: a wave-uniform first unit plus a per-lane offset.  Whole-array launches: the tile's base and the lane
        // number (tail lanes recompute the last unit, store nothing) — stream addresses are then a scalar base and a 32-bit
        // lane offset; queue launches: 0 and the queued unit index.
        r.first = INDEXED ? (size_t)0 : base;
        if constexpr (INDEXED) r.i = (size_t)a.idx[r.active ? base + lane : n_items - 1];
        else r.i = (size_t)(r.active ? lane : (unsigned)(n_items - 1 - base));
        r.id = 0;
        if constexpr (MULTI) r.id = (a.mat + r.first)[r.i];
        r.wox = 0.0f; r.woy = 0.0f; r.woz = 1.0f; r.u0 = 0.0f; r.u1 = 0.0f;
        load3s<NT>(a.wi + 3 * r.first, r.i, r.wix, r.wiy, r.wiz);
        if constexpr (HAS_EVAL) load3s<NT>(a.wo + 3 * r.first, r.i, r.wox, r.woy, r.woz);
        if constexpr (HAS_SAMPLE) { const float *up = a.u + 2 * r.first; r.u0 = ldf<NT>(up + 2 * r.i); r.u1 = ldf<NT>(up + 2 * r.i + 1); }
        return r;
    };
    bool live = tile_live(tile);
    TileIn cur = {};
    if (live) cur = request(tile);
    // The first tile's inputs have landed before the loop is entered (a real S_WAITCNT vmcnt(0)): the compiler's wait-count
    // pass merges the states of both loop entries, and with loads pending on this one it would wait for everything —
    // the copies just issued included — at the first use of any loop-carried input register.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    while (live) {
        const size_t tile_next = tile + tile_step;
        const bool live_next = tile_live(tile_next);
        // requested unconditionally (a wave without a next tile asks for its current one again): a conditional request
        // turns into register copies right behind the loads, i.e. into a wait for them
        const TileIn nxt = request(live_next ? tile_next : tile);

        MaterialDev m;
        bool known = true;
        if constexpr (MULTI) {
            known = cur.id >= 0 && cur.id < a.n_materials;
            m = a.materials[known ? cur.id : 0];
            known = known && kind_is_rgb_path(m.kind);
            if (!known) m = a.safe;
        } else {
            m = a.single;
        }
        const bool is_table = !GGX || m.kind != KIND_GGX;

        UnitIO io = {};
        io.wix = cur.wix; io.wiy = cur.wiy; io.wiz = known ? cur.wiz : 0.0f;
        io.wox = cur.wox; io.woy = cur.woy; io.woz = cur.woz; io.u0 = cur.u0; io.u1 = cur.u1;
        io.wi_sum = io.wix + io.wiy + io.wiz;
        const fast::Vec3 in = fast::normalize_f32(io.wix, io.wiy, io.wiz);

        if (!GGX || __ballot(is_table) != 0ull)               // wave-uniform
            table_lanes<MODE, MULTI, GGX, STD>(a, m, is_table, io, in, ldsA, ldsB, pageA, pageB, lane);
        if constexpr (GGX) {
            if (!is_table) ggx_lane<MODE>(m, io, in);
        }
        if (cur.active) store_unit<MODE, NT>(a, cur.first, cur.i, io);
        cur = nxt; tile = tile_next; live = live_next;
    }
}

// ---- tuned GGX rough conductor (single-material launches of an analytic material) ----------------
// PER_LANE: the material comes from mat[i] (a batch over several analytic materials);
// INDEXED: walks a queue of unit indices (a caller's wavefront queue, or the GGX queue of a kind-partitioned batch).
template <int MODE, bool NT, bool PER_LANE = false, bool INDEXED = false>
__global__ __launch_bounds__(kBlock) void k_ggx(BatchArgs a)
{
    constexpr bool HAS_EVAL = mode_eval(MODE);
    constexpr bool HAS_PDF = mode_pdf(MODE);
    constexpr bool HAS_SAMPLE = mode_sample(MODE);
    const size_t stride = (size_t)gridDim.x * kBlock;
    const size_t n_items = item_count<INDEXED>(a);
    for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < n_items; j += stride) {
        const size_t i = INDEXED ? (size_t)a.idx[j] : j;
        bool known = true;
        int id = 0;
        if constexpr (PER_LANE) {
            id = a.mat[i];
            known = id >= 0 && id < a.n_materials && a.materials[id].kind == KIND_GGX;
        }
        const fast::GgxConsts g(PER_LANE ? a.materials[known ? id : 0] : a.single);
        float wix, wiy, wiz;
        load3s<NT>(a.wi, i, wix, wiy, wiz);
        if (!known) wiz = 0.0f;                               // unknown material id: every output zero
        const fast::Vec3 in = fast::normalize_f32(wix, wiy, wiz);
        if constexpr (HAS_EVAL || HAS_PDF) {
            float wox, woy, woz;
            load3s<NT>(a.wo, i, wox, woy, woz);
            const fast::Vec3 out = fast::normalize_f32(wox, woy, woz);
            double v[3], p;
            fast::ggx_eval_pdf(g, in, out, v, p);
            const bool valid = (wiz > 0.0f) && (woz > 0.0f);
            const double poison = fast::cos_or_nan(wix, wiy, wiz, wox, woy, 1.0f);     // 1.0, or NaN for non-finite input
            if constexpr (HAS_EVAL) {
                const float rgb[3] = { valid ? (float)(v[0] * poison) : 0.0f, valid ? (float)(v[1] * poison) : 0.0f,
                                       valid ? (float)(v[2] * poison) : 0.0f };
                store3s<NT>(a.out_rgb, i, rgb);
            }
            if constexpr (HAS_PDF) stf<NT>(a.out_pdf + i, valid ? (float)(p * poison) : 0.0f);
        }
        if constexpr (HAS_SAMPLE) {
            const float u0 = ldf<NT>(a.u + 2 * i), u1 = ldf<NT>(a.u + 2 * i + 1);
            float wo2[3], pdf2, w[3];
            fast::ggx_sample(g, in, u0, u1, wo2, pdf2, w);
            if (!(wiz > 0.0f)) { wo2[0] = wo2[1] = wo2[2] = 0.0f; pdf2 = 0.0f; w[0] = w[1] = w[2] = 0.0f; }
            store3s<NT>(a.out_wo, i, wo2);
            stf<NT>(a.out_pdf2 + i, pdf2);
            store3s<NT>(a.out_weight, i, w);
        }
    }
}

// ---- variant 4: per-kind queues for batches that mix table and analytic materials -----------------
// A random mix makes every wave of a fused kernel pay for both code paths, at the register budget of
// their union.  Instead the batch is partitioned once into two DENSE queues of unit indices and each
// queue runs through its own dedicated kernel at that kernel's best occupancy (the fabric-bound table
// kernel, the VALU-bound GGX kernel).  The partition uses no atomics (one queue counter would serialise
// ~1M wave atomics into 15+ ms) and is deterministic:
//   k_count_kinds      block b counts the kinds of segment b = units [b*seg_len, (b+1)*seg_len)
//   k_scan_segments    one block: exclusive prefix of the per-segment counts -> queue offsets, totals
//   k_partition_kinds  block b again walks its segment in 256-unit tiles: per wave a ballot of each
//                      kind, popcount for the wave's count, mbcnt for the lane's rank (the wavefront
//                      ballot/prefix), a prefix over the block's 4 waves in LDS, running offsets in
//                      registers; lane writes its unit index to offset[b][kind] + rank
// Queue entries ascend, so the consumers' indexed loads/stores stay line-friendly, and consumers see a
// dense [0, total) range: no idle lanes, no load imbalance between blocks.
struct KindFlags { bool tab, ggx; };
__device__ __forceinline__ KindFlags kind_of(const int32_t *mat, size_t i, bool in_range, const MaterialDev *materials, int n_materials)
{
    const int id = in_range ? mat[i] : -1;
    const bool known = id >= 0 && id < n_materials;
    const bool ggx = known && materials[id].kind == KIND_GGX;
    return { in_range && !ggx, ggx };                         // unknown ids ride the table queue and come out as zeros
}

__global__ __launch_bounds__(kBlock) void k_count_kinds(const int32_t *mat, size_t n, const MaterialDev *materials, int n_materials,
                                                       uint32_t *counts, uint32_t seg_len)
{
    __shared__ unsigned s_sum[2];
    if (threadIdx.x < 2) s_sum[threadIdx.x] = 0;
    __syncthreads();
    const size_t first = (size_t)blockIdx.x * seg_len;
    const size_t last = first + seg_len < n ? first + seg_len : n;
    unsigned t = 0, g = 0;
    for (size_t i = first + threadIdx.x; i < last; i += kBlock) {
        const KindFlags k = kind_of(mat, i, true, materials, n_materials);
        t += k.tab; g += k.ggx;
    }
    // wave reduce by ballot-free shuffle, then one LDS atomic per wave
    for (int off = 32; off > 0; off >>= 1) { t += __shfl_down(t, off); g += __shfl_down(g, off); }
    if ((threadIdx.x & 63u) == 0) { atomicAdd(&s_sum[0], t); atomicAdd(&s_sum[1], g); }
    __syncthreads();
    if (threadIdx.x < 2) counts[2 * blockIdx.x + threadIdx.x] = s_sum[threadIdx.x];
}

// counts[S][2] -> offsets[S][2] (exclusive prefix per kind), totals[2]; S <= a few thousand: one block
__global__ __launch_bounds__(kBlock) void k_scan_segments(const uint32_t *counts, uint32_t segments, uint32_t *offsets, uint32_t *totals)
{
    __shared__ unsigned s_part[kBlock][2];
    const unsigned tid = threadIdx.x;
    const unsigned per = (segments + kBlock - 1) / kBlock;
    const unsigned lo = tid * per, hi = lo + per < segments ? lo + per : segments;
    unsigned t = 0, g = 0;
    for (unsigned s = lo; s < hi; ++s) { t += counts[2 * s]; g += counts[2 * s + 1]; }
    s_part[tid][0] = t; s_part[tid][1] = g;
    __syncthreads();
    unsigned bt = 0, bg = 0;
    for (unsigned k = 0; k < tid; ++k) { bt += s_part[k][0]; bg += s_part[k][1]; }
    for (unsigned s = lo; s < hi; ++s) {
        offsets[2 * s] = bt; offsets[2 * s + 1] = bg;
        bt += counts[2 * s]; bg += counts[2 * s + 1];
    }
    if (tid == kBlock - 1) { totals[0] = bt; totals[1] = bg; }
}

__global__ __launch_bounds__(kBlock) void k_partition_kinds(const int32_t *mat, size_t n, const MaterialDev *materials,
                                                           int n_materials, uint32_t *queue_table, uint32_t *queue_ggx,
                                                           const uint32_t *offsets, uint32_t seg_len)
{
    constexpr int WAVES = kBlock / 64;
    __shared__ unsigned s_cnt[2][WAVES][2];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const size_t first = (size_t)blockIdx.x * seg_len;
    const size_t last = first + seg_len < n ? first + seg_len : n;
    unsigned t_run = offsets[2 * blockIdx.x], g_run = offsets[2 * blockIdx.x + 1], parity = 0;
    for (size_t tile = first; tile < last; tile += kBlock) {          // block-uniform trip count
        const size_t i = tile + tid;
        const KindFlags k = kind_of(mat, i, i < last, materials, n_materials);
        const unsigned long long mg = __ballot(k.ggx), mt = __ballot(k.tab);
        if (lane == 0) { s_cnt[parity][wave][0] = (unsigned)__popcll(mt); s_cnt[parity][wave][1] = (unsigned)__popcll(mg); }
        __syncthreads();
        unsigned t_before = 0, g_before = 0, t_tile = 0, g_tile = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const unsigned ct = s_cnt[parity][w][0], cg = s_cnt[parity][w][1];
            t_tile += ct; g_tile += cg;
            t_before += (unsigned)w < wave ? ct : 0u;
            g_before += (unsigned)w < wave ? cg : 0u;
        }
        if (k.tab) queue_table[t_run + t_before + __builtin_amdgcn_mbcnt_hi((unsigned)(mt >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mt, 0u))] = (uint32_t)i;
        if (k.ggx) queue_ggx[g_run + g_before + __builtin_amdgcn_mbcnt_hi((unsigned)(mg >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mg, 0u))] = (uint32_t)i;
        t_run += t_tile; g_run += g_tile;
        parity ^= 1u;                                         // double-buffered counts: one barrier per tile
    }
}

// ---- per-material compaction for wavefront callers (mrl_partition_by_material) ---------------------
// A stable partition of the slots [0, n) by material id into one ascending queue per material, built with
// the wavefront ballot/prefix idiom and no global atomics.  Every WAVE owns a contiguous chunk of the slots
// and walks it 64 slots at a time; inside a step it peels off one material at a time:
//     id0  = the id of the first lane not yet served          (readlane of the lowest set bit)
//     mask = ballot(id == id0)                                 (who else carries it)
//     rank = mbcnt(mask)                                       (my position among them)
// so a step costs as many iterations as it holds DISTINCT materials.  Pass 1 leaves counts[chunk][k], pass 2
// turns them into start offsets (one block per material scans its column; a last block lays the materials
// end to end), pass 3 repeats the walk and writes slot indices to base[k] + offset[chunk][k] + running + rank.
constexpr int kPartWaves = kBlock / 64;

__device__ __forceinline__ unsigned lane_rank(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// WRITE = false: count; WRITE = true: scatter slot indices.  table: [chunks][K] counts (pass 1 out) or start offsets (pass 3 in)
template <bool WRITE>
__global__ __launch_bounds__(kBlock) void k_partition_materials(const int32_t *mat, size_t n, int K, uint32_t chunk_len,
                                                               uint32_t *table, uint32_t *queue)
{
    extern __shared__ unsigned s_run[];                       // [kPartWaves][K] running counters, wave-private
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const size_t chunk = (size_t)blockIdx.x * kPartWaves + wave;
    volatile unsigned *run = s_run + (size_t)wave * K;
    uint32_t *row = table + chunk * (size_t)K;
    for (int k = (int)lane; k < K; k += 64) run[k] = WRITE ? row[k] : 0u;
    __builtin_amdgcn_wave_barrier();
    const size_t first = chunk * chunk_len;
    const size_t last = first + chunk_len < n ? first + chunk_len : n;
    int id_next = first + lane < last ? mat[first + lane] : -1;
    for (size_t base = first; base < last; base += 64) {      // wave-uniform trip count
        const size_t i = base + lane;
        const int id = id_next;
        id_next = i + 64 < last ? mat[i + 64] : -1;           // next step's ids travel while this step is ranked
        const bool valid = id >= 0 && id < K;                 // ids outside the material list are dropped
        unsigned long long todo = __ballot(valid);
        while (todo) {                                        // wave-uniform
            const int leader = __builtin_ctzll(todo);
            const int id0 = __builtin_amdgcn_readlane(id, leader);
            const unsigned long long mask = __ballot(valid && id == id0);
            const unsigned start = run[id0];
            if constexpr (WRITE) {
                if (valid && id == id0) queue[start + lane_rank(mask)] = (uint32_t)i;
            }
            __builtin_amdgcn_wave_barrier();
            if ((int)lane == leader) run[id0] = start + (unsigned)__popcll(mask);
            __builtin_amdgcn_wave_barrier();
            todo &= ~mask;
        }
    }
    if constexpr (!WRITE) {
        for (int k = (int)lane; k < K; k += 64) row[k] = run[k];
    }
}

// block k: exclusive prefix of column k of counts[chunks][K] in place, column total -> totals[k]
__global__ __launch_bounds__(kBlock) void k_scan_material_columns(uint32_t *table, uint32_t chunks, int K, uint32_t *totals)
{
    __shared__ unsigned s_part[kBlock];
    const unsigned tid = threadIdx.x, k = blockIdx.x;
    const unsigned per = (chunks + kBlock - 1) / kBlock;
    const unsigned lo = tid * per, hi = lo + per < chunks ? lo + per : chunks;
    unsigned sum = 0;
    for (unsigned c = lo; c < hi; ++c) sum += table[(size_t)c * K + k];
    s_part[tid] = sum;
    __syncthreads();
    unsigned before = 0;
    for (unsigned t = 0; t < tid; ++t) before += s_part[t];
    for (unsigned c = lo; c < hi; ++c) {
        const unsigned v = table[(size_t)c * K + k];
        table[(size_t)c * K + k] = before;
        before += v;
    }
    if (tid == kBlock - 1) totals[k] = before;
}

// one block: materials end to end -> offsets[K + 1], counts[K]; then every column gets its material's base added
__global__ __launch_bounds__(kBlock) void k_material_bases(const uint32_t *totals, int K, uint32_t *offsets, uint32_t *counts)
{
    if (threadIdx.x == 0) {
        unsigned at = 0;
        for (int k = 0; k < K; ++k) { offsets[k] = at; counts[k] = totals[k]; at += totals[k]; }
        offsets[K] = at;
    }
}

__global__ __launch_bounds__(kBlock) void k_add_material_bases(uint32_t *table, size_t cells, int K, const uint32_t *offsets)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < cells; j += stride) table[j] += offsets[j % (size_t)K];
}

// ---- a1: table re-layout on the device (planar f64 file order -> HBM layout), one thread per output texel/brick ----
// clamp (MRL_OPT_NEGATIVE = 0, the default): MERL's negative "not measured" markers become 0 in the image; otherwise the value
// stays as the file holds it (kept, or recognised and left out by the renormalising blend)
__device__ __forceinline__ float scaled_texel(const double *planar, size_t plane, size_t index, int ch, double scale, int clamp)
{
    const double v = planar[index + (size_t)ch * plane] * scale;
    return (v > 0.0 || !clamp) ? (float)v : 0.0f;
}

__global__ __launch_bounds__(kBlock) void k_build_bricks(const double *planar, int n_th, int n_td, int n_pd, int phi_periodic,
                                                        double s0, double s1, double s2, int clamp, float4 *bricks)
{
    const size_t cells = (size_t)n_th * n_td * n_pd, plane = cells;
    const size_t stride = (size_t)gridDim.x * kBlock;
    const double scale[3] = { s0, s1, s2 };
    for (size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x; c < cells; c += stride) {
        const int ip = (int)(c % (size_t)n_pd), id = (int)((c / (size_t)n_pd) % (size_t)n_td), ih = (int)(c / ((size_t)n_pd * n_td));
        float v[32];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int sh = min(ih + (k >> 2), n_th - 1), sd = min(id + ((k >> 1) & 1), n_td - 1), sp = phi_periodic ? (ip + (k & 1)) % n_pd : min(ip + (k & 1), n_pd - 1);
            const size_t src = ((size_t)sh * n_td + sd) * n_pd + sp;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) v[3 * k + ch] = scaled_texel(planar, plane, src, ch, scale[ch], clamp);
        }
#pragma unroll
        for (int pad = 24; pad < 32; ++pad) v[pad] = 0.0f;
        float4 *dst = bricks + c * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) dst[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
}

__global__ __launch_bounds__(kBlock) void k_build_rows(const double *planar, int n_th, int n_td, int n_pd, int phi_periodic,
                                                      double s0, double s1, double s2, int clamp, float4 *rows)
{
    const size_t H = n_th + 1, D = n_td + 1, P = n_pd + 1, total = H * D * P, plane = (size_t)n_th * n_td * n_pd;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
        const size_t ip = t % P, id = (t / P) % D, ih = t / (P * D);
        const size_t sh = ih < (size_t)n_th ? ih : n_th - 1, sd = id < (size_t)n_td ? id : n_td - 1, sp = ip == (size_t)n_pd ? (phi_periodic ? 0 : n_pd - 1) : ip;
        const size_t src = (sh * n_td + sd) * n_pd + sp;
        rows[t] = make_float4(scaled_texel(planar, plane, src, 0, s0, clamp), scaled_texel(planar, plane, src, 1, s1, clamp),
                              scaled_texel(planar, plane, src, 2, s2, clamp), 0.0f);
    }
}

// ---- layout to layout (the on-disk image cache keeps RGB tables in the compact rows form, whatever the context's layout) ----
// rows -> bricks: a cell's eight corners are eight texels of the padded rows image (its extra texels ARE the clamps / the phi wrap)
__global__ __launch_bounds__(kBlock) void k_rows_to_bricks(const float4 *rows, int n_th, int n_td, int n_pd, float4 *bricks)
{
    const size_t cells = (size_t)n_th * n_td * n_pd, D = n_td + 1, P = n_pd + 1;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x; c < cells; c += stride) {
        const size_t ip = c % (size_t)n_pd, id = (c / (size_t)n_pd) % (size_t)n_td, ih = c / ((size_t)n_pd * n_td);
        float v[32];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 t = rows[((ih + (size_t)(k >> 2)) * D + id + (size_t)((k >> 1) & 1)) * P + ip + (size_t)(k & 1)];
            v[3 * k] = t.x; v[3 * k + 1] = t.y; v[3 * k + 2] = t.z;
        }
#pragma unroll
        for (int pad = 24; pad < 32; ++pad) v[pad] = 0.0f;
        float4 *dst = bricks + c * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) dst[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
}
// bricks -> rows: corner 0 of a cell is the cell's own texel; the padding texels repeat the last one / wrap phi as k_build_rows does
__global__ __launch_bounds__(kBlock) void k_bricks_to_rows(const float4 *bricks, int n_th, int n_td, int n_pd, int phi_periodic, float4 *rows)
{
    const size_t H = n_th + 1, D = n_td + 1, P = n_pd + 1, total = H * D * P;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
        const size_t ip = t % P, id = (t / P) % D, ih = t / (P * D);
        const size_t sh = ih < (size_t)n_th ? ih : n_th - 1, sd = id < (size_t)n_td ? id : n_td - 1, sp = ip == (size_t)n_pd ? (phi_periodic ? 0 : n_pd - 1) : ip;
        const float4 q = bricks[((sh * n_td + sd) * n_pd + sp) * 8];
        rows[t] = make_float4(q.x, q.y, q.z, 0.0f);
    }
}

// ---- the conditional sampling table P(theta_h | theta_i) (definition: oracle/merl_oracle.h, "survey form") -------------
// Built on the device from the RESIDENT table, through the kernels' own lookup.  Pass 1, one wave per (incident bin i,
// theta_h bin j): its 64 lanes are the K_s x K_p = 4 x 16 quadrature midpoints of the bin; lane -> BRDF mass at its
// half vector -> wave sum (shuffle tree) -> W[i][j].  Pass 2, one block per incident bin: the row's total (block
// reduction), the 1 % floor, then an exclusive prefix scan over the n_th bins (per-thread serial chunks, a
// Hillis-Steele scan of the 256 partials in LDS) -> cdf, and c = mass / (Z pi ds).
constexpr int kS2dKs = 4, kS2dKp = 16;

template <int LAYOUT>
__global__ __launch_bounds__(64) void k_sampling2d_mass(MaterialDev m, Options o, int n_ti, double *W)
{
    const int j = (int)blockIdx.x, i = (int)blockIdx.y, lane = (int)threadIdx.x;
    const int n = m.n_th;
    const double s0 = m.sampling[j], s1 = m.sampling[j + 1], ds = s1 - s0;
    const double mu = ((double)i + 0.5) / (double)n_ti;
    const int a = lane / kS2dKp, b = lane % kS2dKp;
    const double s = s0 + ((double)a + 0.5) / kS2dKs * ds, phi = ((double)b + 0.5) / kS2dKp * kPi;
    double v = 0.0;
    if (m.param == PARAM_HALF_DIFF) {
        const fast::Vec3 in = { sqrt(fmax(1.0 - mu * mu, 0.0)), 0.0, mu };
        const double st = sqrt(s), ct = sqrt(fmax(1.0 - s, 0.0));
        const double hx = st * cos(phi), hy = st * sin(phi);
        const double c = in.x * hx + in.z * ct;
        const fast::Dir out = { 2.0 * c * hx - in.x, 2.0 * c * hy, 2.0 * c * ct - in.z, 1.0 };
        if (c > 0.0 && ct > 0.0 && out.z > 0.0) {
            const fast::TableMaps maps(m);
            // (the mass of a density: clamped values, whatever eval() does with negative texels)
            const Rgbf f = lookup_trilinear_t<LAYOUT>(m, maps(in, out), o.node, o.negative == NEGATIVE_CLAMP ? BLEND_STORED : BLEND_CLAMP);
            const double lum = 0.2126 * (double)f.r + 0.7152 * (double)f.g + 0.0722 * (double)f.b;
            v = lum * out.z * 4.0 * c / (2.0 * ct);
        }
    }
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane == 0) W[(size_t)i * n + j] = ds * v / (double)(kS2dKs * kS2dKp);
}

__global__ __launch_bounds__(kBlock) void k_sampling2d_scan(const double *s, int n, const double *W, double *rows)
{
    __shared__ double part[kBlock];
    const int i = (int)blockIdx.x, tid = (int)threadIdx.x;
    const double *w = W + (size_t)i * n;
    double *cdf = rows + (size_t)i * (2 * n + 1), *c = cdf + (n + 1);
    const int per = (n + kBlock - 1) / kBlock, lo = tid * per, hi = min(lo + per, n);
    // 1. the row's total -> the floor (1 % of the mass, uniform in s; a flat row when the table gives nothing)
    double sum = 0.0;
    for (int j = lo; j < hi; ++j) sum += w[j];
    part[tid] = sum;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) { if (tid < off) part[tid] += part[tid + off]; __syncthreads(); }
    const double total = part[0], span = s[n] - s[0];
    __syncthreads();
    auto mass = [&](int j) { const double ds = s[j + 1] - s[j]; return total > 0.0 ? w[j] + 0.01 * total * ds / span : ds / span; };
    // 2. exclusive prefix scan of the floored masses
    sum = 0.0;
    for (int j = lo; j < hi; ++j) sum += mass(j);
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {              // Hillis-Steele inclusive scan of the partials
        const double add = tid >= off ? part[tid - off] : 0.0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    const double Z = part[kBlock - 1];
    double run = tid ? part[tid - 1] : 0.0;
    for (int j = lo; j < hi; ++j) {
        const double wj = mass(j);
        cdf[j] = run / Z;
        c[j] = wj / (Z * kPi * (s[j + 1] - s[j]));
        run += wj;
    }
    if (tid == 0) cdf[n] = 1.0;
}

__global__ __launch_bounds__(kBlock) void k_generate_pairs(uint64_t seed, uint64_t first, size_t n,
                                                          float *wi, float *wo, float *u)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < n; t += stride) {
        uint64_t i = first + t;
        uint64_t r0 = mix64(seed ^ (3 * i)), r1 = mix64(seed ^ (3 * i + 1)), r2 = mix64(seed ^ (3 * i + 2));
        float v[3];
        hemisphere_dir(r0, v[0], v[1], v[2]);
        store3(wi, t, v);
        hemisphere_dir(r1, v[0], v[1], v[2]);
        store3(wo, t, v);
        u[2 * t + 0] = (float)(uint32_t)(r2 >> 40) * 0x1p-24f;
        u[2 * t + 1] = (float)(uint32_t)((r2 >> 16) & 0xFFFFFFu) * 0x1p-24f;
    }
}

__global__ __launch_bounds__(kBlock) void k_generate_materials(uint64_t seed, uint64_t first, size_t n,
                                                              int n_materials, int32_t *mat)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < n; t += stride) {
        uint64_t r = mix64((seed ^ 0x4D41544552494131ULL) + (first + t));
        mat[t] = (int32_t)(((r >> 32) * (uint64_t)n_materials) >> 32);
    }
}

inline unsigned grid_for(size_t n, int compute_units)
{
    size_t blocks = (n + kBlock - 1) / kBlock;
    size_t cap = (size_t)compute_units * 8;         // 8 x 256-thread blocks per CU, grid-stride the rest
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

// runtime (multi, nt, lookup, layout) -> compile-time kernel
template <int MODE, bool MULTI, bool NT, int LOOKUP>
void launch_table3(const BatchArgs &a, int layout, dim3 grid, dim3 block, hipStream_t stream)
{
    if (layout == LAYOUT_BRICK) hipLaunchKernelGGL((k_table<MODE, MULTI, NT, LOOKUP, LAYOUT_BRICK>), grid, block, 0, stream, a);
    else                        hipLaunchKernelGGL((k_table<MODE, MULTI, NT, LOOKUP, LAYOUT_ROWS>), grid, block, 0, stream, a);
}
template <int MODE, bool MULTI, bool NT>
void launch_table2(const BatchArgs &a, int lookup, int layout, dim3 grid, dim3 block, hipStream_t stream)
{
    if (lookup) launch_table3<MODE, MULTI, NT, 1>(a, layout, grid, block, stream);
    else        launch_table3<MODE, MULTI, NT, 0>(a, layout, grid, block, stream);
}
// the LDS-DMA kernel in its half/diff-only or its every-parameterisation build (BatchArgs::any_standard); MODE, g, b, stream, a in scope
#define MRL_DMA_LAUNCH(MULTI_, GGX_, INDEXED_)                                                                         \
    do {                                                                                                               \
        if (a.any_standard || a.opts.negative == NEGATIVE_RENORMALISE)                                                 \
            hipLaunchKernelGGL((k_table_dma<MODE, MULTI_, true, GGX_, INDEXED_, true>), g, b, 0, stream, a);           \
        else                hipLaunchKernelGGL((k_table_dma<MODE, MULTI_, true, GGX_, INDEXED_, false>), g, b, 0, stream, a); \
    } while (0)

template <int MODE>
void launch_table(const BatchArgs &a, bool multi, bool nt, int lookup, int layout, dim3 grid, dim3 block, hipStream_t stream)
{
    if (multi) { if (nt) launch_table2<MODE, true, true>(a, lookup, layout, grid, block, stream); else launch_table2<MODE, true, false>(a, lookup, layout, grid, block, stream); }
    else       { if (nt) launch_table2<MODE, false, true>(a, lookup, layout, grid, block, stream); else launch_table2<MODE, false, false>(a, lookup, layout, grid, block, stream); }
}

template <int MODE>
hipError_t launch_mode(const BatchArgs &a, bool multi, int variant, int layout, bool has_ggx, bool has_table, int compute_units, hipStream_t stream)
{
    dim3 grid(grid_for(a.n, compute_units)), block(kBlock);
    // variant 0: generic kernel (every kind, ocml math) — the A/B baseline;
    // variant 1: tuned table kernel; a single-material GGX launch has no table path and stays generic
    const bool tuned = variant >= 1 && (multi || a.single.kind != KIND_GGX);
    if (variant >= 1 && !multi && a.single.kind == KIND_GGX) {     // tuned analytic kernel
        hipLaunchKernelGGL((k_ggx<MODE, true>), grid, block, 0, stream, a);
        return hipGetLastError();
    }
    if (variant >= 1 && multi && has_ggx && !has_table) {          // a batch over analytic materials only
        hipLaunchKernelGGL((k_ggx<MODE, true, true>), grid, block, 0, stream, a);
        return hipGetLastError();
    }
    if constexpr (MODE != MODE_PDF) {
        // variant 3: cooperative LDS-DMA brick fetch (brick layout + trilinear only; otherwise variant 2)
        if (tuned && variant >= 3 && layout == LAYOUT_BRICK && a.opts.lookup == 1) {
            // 64 KB (two lookups) or 32 KB (one) of LDS per 256-thread block: 2 or 4 blocks per CU
            constexpr int per_cu = dma_blocks_per_cu(MODE);
            size_t blocks = (a.n + kDmaBlock - 1) / kDmaBlock;
            if (blocks > (size_t)compute_units * per_cu) blocks = (size_t)compute_units * per_cu;
            blocks = (blocks + 7) / 8 * 8;                    // whole rounds over the 8 XCDs (BatchArgs::block_map)
            const dim3 g((unsigned)blocks), b(kDmaBlock);
            if (multi && has_ggx)      MRL_DMA_LAUNCH(true, true, false);
            else if (multi)            MRL_DMA_LAUNCH(true, false, false);
            else                       MRL_DMA_LAUNCH(false, false, false);
            return hipGetLastError();
        }
    }
    // (k_table blends the texels as stored: a renormalising context's nearest lookups / rows-layout tables take the generic kernel)
    if (tuned && a.opts.negative != NEGATIVE_RENORMALISE) {
        launch_table<MODE>(a, multi, variant >= 2, a.opts.lookup, layout, grid, block, stream);
    } else {
        if (multi) hipLaunchKernelGGL((k_batch<MODE, true>), grid, block, 0, stream, a);
        else       hipLaunchKernelGGL((k_batch<MODE, false>), grid, block, 0, stream, a);
    }
    return hipGetLastError();
}

} // namespace

hipError_t launch_batch(int mode, const BatchArgs &a, bool multi, int variant, int layout, bool has_ggx, bool has_table, int compute_units, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    switch (mode) {
        case MODE_EVAL:        return launch_mode<MODE_EVAL>(a, multi, variant, layout, has_ggx, has_table, compute_units, stream);
        case MODE_PDF:         return launch_mode<MODE_PDF>(a, multi, variant, layout, has_ggx, has_table, compute_units, stream);
        case MODE_SAMPLE:      return launch_mode<MODE_SAMPLE>(a, multi, variant, layout, has_ggx, has_table, compute_units, stream);
        case MODE_EVAL_SAMPLE: return launch_mode<MODE_EVAL_SAMPLE>(a, multi, variant, layout, has_ggx, has_table, compute_units, stream);
        case MODE_EVAL_PDF:    return launch_mode<MODE_EVAL_PDF>(a, multi, variant, layout, has_ggx, has_table, compute_units, stream);
    }
    return hipErrorInvalidValue;
}

void partition_geometry(size_t n, int compute_units, uint32_t *segments, uint32_t *seg_len)
{
    size_t s = (n + kBlock - 1) / kBlock;
    const size_t cap = (size_t)compute_units * 8;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    size_t len = (n + s - 1) / s;
    len = (len + kBlock - 1) / kBlock * kBlock;
    *segments = (uint32_t)((n + len - 1) / len);
    *seg_len = (uint32_t)len;
}

// work: uint32 [2*segments counts][2*segments offsets][2 totals]; queues: dense, n entries each at most
hipError_t launch_partition_kinds(const int32_t *mat, size_t n, const MaterialDev *materials, int n_materials,
                                  uint32_t *queue_table, uint32_t *queue_ggx, uint32_t *work,
                                  uint32_t segments, uint32_t seg_len, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    uint32_t *counts = work, *offsets = work + 2 * (size_t)segments, *totals = work + 4 * (size_t)segments;
    hipLaunchKernelGGL(k_count_kinds, dim3(segments), dim3(kBlock), 0, stream, mat, n, materials, n_materials, counts, seg_len);
    hipLaunchKernelGGL(k_scan_segments, dim3(1), dim3(kBlock), 0, stream, counts, segments, offsets, totals);
    hipLaunchKernelGGL(k_partition_kinds, dim3(segments), dim3(kBlock), 0, stream, mat, n, materials, n_materials,
                       queue_table, queue_ggx, offsets, seg_len);
    return hipGetLastError();
}

namespace {
template <int MODE>
hipError_t launch_queue_mode(const BatchArgs &a, bool ggx_queue, int compute_units, hipStream_t stream)
{
    if constexpr (MODE == MODE_PDF) {
        return hipErrorInvalidValue;
    } else {
        if (ggx_queue) {
            hipLaunchKernelGGL((k_ggx<MODE, true, true, true>), dim3(grid_for(a.n, compute_units)), dim3(kBlock), 0, stream, a);
        } else {
            constexpr int per_cu = dma_blocks_per_cu(MODE);
            size_t blocks = (a.n + kDmaBlock - 1) / kDmaBlock;
            if (blocks > (size_t)compute_units * per_cu) blocks = (size_t)compute_units * per_cu;
            const dim3 g((unsigned)blocks), b(kDmaBlock);
            MRL_DMA_LAUNCH(true, false, true);
        }
        return hipGetLastError();
    }
}
} // namespace

// one kind's dense queue (a.idx, a.idx_count) of a partitioned mixed batch; a.n = units of the whole batch (grid sizing)
hipError_t launch_batch_queue(int mode, const BatchArgs &a, bool ggx_queue, int compute_units, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    switch (mode) {
        case MODE_EVAL:        return launch_queue_mode<MODE_EVAL>(a, ggx_queue, compute_units, stream);
        case MODE_SAMPLE:      return launch_queue_mode<MODE_SAMPLE>(a, ggx_queue, compute_units, stream);
        case MODE_EVAL_SAMPLE: return launch_queue_mode<MODE_EVAL_SAMPLE>(a, ggx_queue, compute_units, stream);
        case MODE_EVAL_PDF:    return launch_queue_mode<MODE_EVAL_PDF>(a, ggx_queue, compute_units, stream);
    }
    return hipErrorInvalidValue;
}

namespace {
template <int MODE>
hipError_t launch_indexed_mode(const BatchArgs &a, bool multi, int layout, bool has_ggx, bool has_table, int compute_units, hipStream_t stream)
{
    const dim3 grid(grid_for(a.n, compute_units)), block(kBlock);
    if (!multi && a.single.kind == KIND_GGX) {
        hipLaunchKernelGGL((k_ggx<MODE, true, false, true>), grid, block, 0, stream, a);
        return hipGetLastError();
    }
    if (multi && has_ggx && !has_table) {
        hipLaunchKernelGGL((k_ggx<MODE, true, true, true>), grid, block, 0, stream, a);
        return hipGetLastError();
    }
    if constexpr (MODE != MODE_PDF) {
        if (layout == LAYOUT_BRICK && a.opts.lookup == 1) {
            constexpr int per_cu = dma_blocks_per_cu(MODE);
            size_t blocks = (a.n + kDmaBlock - 1) / kDmaBlock;
            if (blocks > (size_t)compute_units * per_cu) blocks = (size_t)compute_units * per_cu;
            blocks = (blocks + 7) / 8 * 8;
            const dim3 g((unsigned)blocks), b(kDmaBlock);
            if (multi && has_ggx)      MRL_DMA_LAUNCH(true, true, true);
            else if (multi)            MRL_DMA_LAUNCH(true, false, true);
            else                       MRL_DMA_LAUNCH(false, false, true);
            return hipGetLastError();
        }
    }
    // rows layout, nearest lookup, pdf: the generic kernel walks the queue
    if (multi) hipLaunchKernelGGL((k_batch<MODE, true, true>), grid, block, 0, stream, a);
    else       hipLaunchKernelGGL((k_batch<MODE, false, true>), grid, block, 0, stream, a);
    return hipGetLastError();
}
} // namespace

// A caller's wavefront queue: units a.idx[0 .. min(*a.idx_count, a.n)) of the arrays in `a`; a.n is the queue's
// capacity (grid sizing and an upper clamp on the device-side count).
hipError_t launch_batch_indexed(int mode, const BatchArgs &a, bool multi, int layout, bool has_ggx, bool has_table,
                                int compute_units, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    switch (mode) {
        case MODE_EVAL:        return launch_indexed_mode<MODE_EVAL>(a, multi, layout, has_ggx, has_table, compute_units, stream);
        case MODE_PDF:         return launch_indexed_mode<MODE_PDF>(a, multi, layout, has_ggx, has_table, compute_units, stream);
        case MODE_SAMPLE:      return launch_indexed_mode<MODE_SAMPLE>(a, multi, layout, has_ggx, has_table, compute_units, stream);
        case MODE_EVAL_SAMPLE: return launch_indexed_mode<MODE_EVAL_SAMPLE>(a, multi, layout, has_ggx, has_table, compute_units, stream);
        case MODE_EVAL_PDF:    return launch_indexed_mode<MODE_EVAL_PDF>(a, multi, layout, has_ggx, has_table, compute_units, stream);
    }
    return hipErrorInvalidValue;
}

void material_partition_geometry(size_t n, int compute_units, uint32_t *chunks, uint32_t *chunk_len)
{
    size_t waves = (n + 63) / 64;
    const size_t cap = (size_t)compute_units * 8 * kPartWaves;
    if (waves > cap) waves = cap;
    if (waves < 1) waves = 1;
    waves = (waves + kPartWaves - 1) / kPartWaves * kPartWaves;       // whole blocks
    size_t len = (n + waves - 1) / waves;
    len = (len + 63) / 64 * 64;
    if (len < 64) len = 64;
    *chunks = (uint32_t)waves;
    *chunk_len = (uint32_t)len;
}

// work: chunks*K + K uint32 (per-chunk table, totals); K <= kMaxPartitionMaterials (LDS: 4 waves x K counters)
hipError_t launch_partition_materials(const int32_t *mat, size_t n, int K, uint32_t *queue, uint32_t *offsets, uint32_t *counts,
                                      uint32_t *work, uint32_t chunks, uint32_t chunk_len, int compute_units, hipStream_t stream)
{
    uint32_t *table = work, *totals = work + (size_t)chunks * K;
    const dim3 grid(chunks / kPartWaves), block(kBlock);
    const size_t lds = (size_t)kPartWaves * K * sizeof(unsigned);
    hipLaunchKernelGGL((k_partition_materials<false>), grid, block, lds, stream, mat, n, K, chunk_len, table, queue);
    hipLaunchKernelGGL(k_scan_material_columns, dim3(K), block, 0, stream, table, chunks, K, totals);
    hipLaunchKernelGGL(k_material_bases, dim3(1), block, 0, stream, totals, K, offsets, counts);
    hipLaunchKernelGGL(k_add_material_bases, dim3(grid_for((size_t)chunks * K, compute_units)), block, 0, stream, table, (size_t)chunks * K, K, offsets);
    hipLaunchKernelGGL((k_partition_materials<true>), grid, block, lds, stream, mat, n, K, chunk_len, table, queue);
    return hipGetLastError();
}

hipError_t launch_rows_to_bricks(const float4 *d_rows, const int dims[3], float4 *d_bricks, int compute_units, hipStream_t stream)
{
    hipLaunchKernelGGL(k_rows_to_bricks, dim3(grid_for((size_t)dims[0] * dims[1] * dims[2], compute_units)), dim3(kBlock), 0, stream,
                       d_rows, dims[0], dims[1], dims[2], d_bricks);
    return hipGetLastError();
}
hipError_t launch_bricks_to_rows(const float4 *d_bricks, const int dims[3], int param, float4 *d_rows, int compute_units, hipStream_t stream)
{
    hipLaunchKernelGGL(k_bricks_to_rows, dim3(grid_for((size_t)(dims[0] + 1) * (dims[1] + 1) * (dims[2] + 1), compute_units)), dim3(kBlock), 0, stream,
                       d_bricks, dims[0], dims[1], dims[2], (int)param_phi_periodic(param), d_rows);
    return hipGetLastError();
}

hipError_t launch_build_table(const double *d_planar, const int dims[3], const double scale[3], int layout, int param, int clamp, float4 *d_out,
                              int compute_units, hipStream_t stream)
{
    const size_t cells = (size_t)dims[0] * dims[1] * dims[2];
    if (layout == LAYOUT_BRICK)
        hipLaunchKernelGGL(k_build_bricks, dim3(grid_for(cells, compute_units)), dim3(kBlock), 0, stream, d_planar, dims[0], dims[1], dims[2],
                           (int)param_phi_periodic(param), scale[0], scale[1], scale[2], clamp, d_out);
    else
        hipLaunchKernelGGL(k_build_rows, dim3(grid_for((size_t)(dims[0] + 1) * (dims[1] + 1) * (dims[2] + 1), compute_units)), dim3(kBlock), 0, stream,
                           d_planar, dims[0], dims[1], dims[2], (int)param_phi_periodic(param), scale[0], scale[1], scale[2], clamp, d_out);
    return hipGetLastError();
}

// d_rows: n_ti x (2 n_th + 1) doubles; d_work: n_ti x n_th doubles.  m: the material with texels and the row marginal resident.
hipError_t launch_build_sampling2d(const MaterialDev &m, const Options &opts, int n_ti, double *d_rows, double *d_work, hipStream_t stream)
{
    const dim3 grid((unsigned)m.n_th, (unsigned)n_ti);
    if (m.layout == LAYOUT_BRICK) hipLaunchKernelGGL((k_sampling2d_mass<LAYOUT_BRICK>), grid, dim3(64), 0, stream, m, opts, n_ti, d_work);
    else                          hipLaunchKernelGGL((k_sampling2d_mass<LAYOUT_ROWS>), grid, dim3(64), 0, stream, m, opts, n_ti, d_work);
    hipLaunchKernelGGL(k_sampling2d_scan, dim3((unsigned)n_ti), dim3(kBlock), 0, stream, m.sampling, m.n_th, (const double *)d_work, d_rows);
    return hipGetLastError();
}

hipError_t launch_generate_pairs(uint64_t seed, uint64_t first, size_t n, float *wi, float *wo, float *u,
                                 int compute_units, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_generate_pairs, dim3(grid_for(n, compute_units)), dim3(kBlock), 0, stream, seed, first, n, wi, wo, u);
    return hipGetLastError();
}

hipError_t launch_generate_materials(uint64_t seed, uint64_t first, size_t n, int n_materials, int32_t *mat,
                                     int compute_units, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_generate_materials, dim3(grid_for(n, compute_units)), dim3(kBlock), 0, stream, seed, first, n, n_materials, mat);
    return hipGetLastError();
}

} // namespace mrl
