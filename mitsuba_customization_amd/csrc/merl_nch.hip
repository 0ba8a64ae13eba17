// merl_nch.hip — n-channel measured tables (customized_measurement beyond RGB: monochrome, RGB + alpha, spectral;
// SURVEY.md §8f item 3).  Same parameterisation, same transform, same trilinear blend as the RGB path; what changes
// is the width of a texel.
//
// HBM layout (bricks only): a cell's 8 corners x CPAD channels x f32, corner-major,
//     CPAD = 1 -> 32 B per cell (4 cells per 128-B line)        n_ch = 1
//     CPAD = 2 -> 64 B per cell                                 n_ch = 2
//     CPAD = 4 -> 128 B per (cell, group of 4 channels)         n_ch = 4 .. 32: ceil(n_ch / 4) lines per cell, contiguous
// (n_ch = 3 is the packed RGB brick of merl_kernels.hip and never comes here.)  The fabric moves whole 128-B lines
// whatever a lookup asks for (profiles/r02_partial_line_probe.json), so a cell is laid out to touch as few lines as
// its width allows: one line serves a lookup up to 4 channels, a 16-channel spectrum costs 4 contiguous lines.
//
// Kernel: one lane = one unit, the wave fetches its 64 bricks cooperatively with global_load_lds_dwordx4 exactly like
// k_table_dma — S = 2 CPAD sixteen-byte pieces per brick, S copy instructions per lookup, source-side XOR swizzle so
// that the unit's ds_read_b128 reads are bank-conflict-free — one channel group at a time (LDS: 64 x S x 16 B per wave
// and lookup; registers hold 4 channels, not n_ch).
#include "merl_kernels.hpp"
#include "merl_table_fast.hpp"
#include "merl_ggx_fast.hpp"

namespace mrl {

namespace {

constexpr int kNchBlock = 256;

enum Mode : int { MODE_EVAL = 0, MODE_PDF = 1, MODE_SAMPLE = 2, MODE_EVAL_SAMPLE = 3, MODE_EVAL_PDF = 4 };
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

// piece swizzle of unit u for bricks of S pieces: any 16 lanes that one ds_read_b128 group serves
// ({0-3, 12-15, 20-27} and its three siblings) then hit 16 different 16-B slots modulo 16
template <int S> __device__ __forceinline__ unsigned nch_swz(unsigned u)
{
    if constexpr (S == 8) return (u >> 1) & 7u;
    else if constexpr (S == 4) return (u >> 2) & 3u;
    else return (u >> 3) & 1u;
}

struct NchWeights { double w[8]; };

// a3 tail: cell index + the 8 corner weights.  Nearest lookups are the trilinear blend with all weight on corner 0.
__device__ __forceinline__ uint32_t nch_cell(int n_th, int n_td, int n_pd, bool phi_periodic, const Coords &c, const Options &o, NchWeights &out)
{
    int h0, d0, p0;
    double fh, fd, fp;
    if (o.lookup) {
        const double shift = o.node ? 0.5 : 0.0;
        split_clamped(c.xh - shift, n_th, h0, fh);
        split_clamped(c.xd - shift, n_td, d0, fd);
        split_phi(phi_periodic, c.xp - shift, n_pd, p0, fp);
    } else {
        h0 = clampi((int)c.xh, 0, n_th - 1); d0 = clampi((int)c.xd, 0, n_td - 1); p0 = clampi((int)c.xp, 0, n_pd - 1);
        fh = fd = fp = 0.0;
    }
    const double gh = 1.0 - fh, gd = 1.0 - fd, gp = 1.0 - fp;
    out.w[0] = gh * gd * gp; out.w[1] = gh * gd * fp; out.w[2] = gh * fd * gp; out.w[3] = gh * fd * fp;
    out.w[4] = fh * gd * gp; out.w[5] = fh * gd * fp; out.w[6] = fh * fd * gp; out.w[7] = fh * fd * fp;
    return (uint32_t)((h0 * n_td + d0) * n_pd + p0);
}

// cooperative copy of the wave's 64 bricks (S pieces each) into lds_slots[64 * S]; brick_addr = this lane's brick
template <int S>
__device__ __forceinline__ void nch_copy(uint64_t brick_addr, float4 *lds_slots, unsigned lane)
{
    constexpr unsigned PER = 64u / S;                         // bricks per copy instruction
    const uint32_t lo = (uint32_t)brick_addr, hi = (uint32_t)(brick_addr >> 32);
    uint32_t got_lo[S], got_hi[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const int sel = (int)((PER * k + lane / S) << 2);
        got_lo[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)lo);
        got_hi[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)hi);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const unsigned unit = PER * k + lane / S;
        const unsigned piece = (lane % S) ^ nch_swz<S>(unit);
        const float4 *src = (const float4 *)(((uint64_t)got_hi[k] << 32) | got_lo[k]) + piece;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds_slots + k * 64), 16, 0, 0);
    }
}

// this lane's brick out of LDS -> up to CPAD channel values
// renorm (MRL_OPT_NEGATIVE = 2, wave-uniform): a negative value marks a sample that was not measured — the valid corners only, divided
// by their weight (0 when none is valid); nearest lookups arrive here with all weight on corner 0 and come out as max(v, 0)
template <int CPAD>
__device__ __forceinline__ void nch_blend(const float4 *lds_slots, unsigned lane, const NchWeights &w, double out[CPAD], bool renorm)
{
    constexpr int S = 2 * CPAD;
    const unsigned f = nch_swz<S>(lane);
    const float4 *q = lds_slots + (unsigned)S * lane;
    float v[8 * CPAD];
#pragma unroll
    for (int p = 0; p < S; ++p) {
        const float4 t = q[(unsigned)p ^ f];
        v[4 * p] = t.x; v[4 * p + 1] = t.y; v[4 * p + 2] = t.z; v[4 * p + 3] = t.w;
    }
    if (renorm) {
#pragma unroll
        for (int ch = 0; ch < CPAD; ++ch) {
            double num = 0.0, den = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float t = v[k * CPAD + ch];
                const double wk = t >= 0.0f ? w.w[k] : 0.0;
                num += wk * (double)t; den += wk;
            }
            out[ch] = den > 0.0 ? num / den : 0.0;
        }
        return;
    }
#pragma unroll
    for (int ch = 0; ch < CPAD; ++ch) {
        double acc = w.w[0] * (double)v[ch];
#pragma unroll
        for (int k = 1; k < 8; ++k) acc += w.w[k] * (double)v[k * CPAD + ch];
        out[ch] = acc;
    }
}

// MODE as in merl_kernels.hip (pdf-only needs no table: the RGB pdf kernel serves every table kind).
// a.out_rgb / a.out_weight hold n x n_ch values.
// INDEXED: walk the caller's wavefront queue a.idx[0 .. min(*a.idx_count, a.n)) instead of the units [0, a.n)
template <bool INDEXED> __device__ __forceinline__ size_t nch_item_count(const BatchArgs &a)
{
    if constexpr (INDEXED) { const size_t c = (size_t)*a.idx_count; return c < a.n ? c : a.n; }
    else return a.n;
}

template <int MODE, bool MULTI, int CPAD, bool INDEXED = false>
__global__ __launch_bounds__(kNchBlock) void k_table_nch(BatchArgs a, int n_ch)
{
    static_assert(MODE != MODE_PDF, "pdf needs no table");
    constexpr bool HAS_EVAL = mode_eval(MODE);
    constexpr bool HAS_SAMPLE = mode_sample(MODE);
    constexpr int LOOKUPS = (HAS_EVAL ? 1 : 0) + (HAS_SAMPLE ? 1 : 0);
    constexpr int S = 2 * CPAD;
    __shared__ float4 lds[kNchBlock / 64][LOOKUPS][64 * S];

    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float4 *ldsA = lds[wave][0];
    float4 *ldsB = lds[wave][LOOKUPS - 1];
    const int groups = CPAD == 4 ? (n_ch + 3) / 4 : 1;
    const size_t stride = (size_t)gridDim.x * kNchBlock;
    const size_t n_items = nch_item_count<INDEXED>(a);
    const bool renorm = a.opts.negative == NEGATIVE_RENORMALISE;       // wave-uniform
    for (size_t base = (size_t)blockIdx.x * kNchBlock + wave * 64u; base < n_items; base += stride) {
        const size_t j = base + lane;
        const bool active = j < n_items;
        const size_t jj = active ? j : n_items - 1;          // tail lanes recompute the last unit, store nothing
        const size_t i = INDEXED ? (size_t)a.idx[jj] : jj;

        MaterialDev m;
        bool known = true;
        if constexpr (MULTI) {
            const int id = a.mat[i];
            known = id >= 0 && id < a.n_materials;
            m = a.materials[known ? id : 0];
            known = known && m.kind == KIND_TABLE_NCH && m.n_ch == n_ch;
        } else {
            m = a.single;
        }
        // a material this call cannot evaluate: a harmless, valid source (the material array itself) and zero outputs
        const float4 *texels = known ? m.texels : (const float4 *)a.materials;
        const int n_th = known ? m.n_th : 1, n_td = known ? m.n_td : 1, n_pd = known ? m.n_pd : 1;

        float wix, wiy, wiz, wox = 0.0f, woy = 0.0f, woz = 1.0f, u0 = 0.0f, u1 = 0.0f;
        load3(a.wi, i, wix, wiy, wiz);
        if (!known) wiz = 0.0f;
        if constexpr (HAS_EVAL) load3(a.wo, i, wox, woy, woz);
        if constexpr (HAS_SAMPLE) { u0 = a.u[2 * i]; u1 = a.u[2 * i + 1]; }
        const fast::Vec3 in = fast::normalize_f32(wix, wiy, wiz);
        const int param = known ? m.param : PARAM_HALF_DIFF;
        const bool phi_periodic = param_phi_periodic(param);
        const fast::TableMaps maps(n_th, n_td, n_pd, param);

        NchWeights wA, wB;
        uint32_t cellA = 0, cellB = 0;
        float sx = 0.0f, sy = 0.0f, sz = 1.0f, sp = 0.0f;
        if constexpr (HAS_EVAL)
            cellA = nch_cell(n_th, n_td, n_pd, phi_periodic, maps(in, fast::dir_f32(wox, woy, woz)), a.opts, wA);
        if constexpr (HAS_SAMPLE) {
            if (a.opts.sampling && known) {                   // option is wave-uniform
                fast::table_sample_dir(m, a.opts.disk_map, in, u0, u1, sx, sy, sz, a.opts.sampling);
                const bool up = sz > 0.0f;
                if (!up) { sx = 0.0f; sy = 0.0f; sz = 1.0f; }
                sp = up ? (float)fast::table_pdf(m, in, fast::normalize_f32(sx, sy, sz), sz, a.opts.sampling) : 0.0f;
            } else {
                square_to_cosine_hemisphere(a.opts.disk_map, u0, u1, sx, sy, sz);
                sp = sz > 0.0f ? sz * kInvPiF : 0.0f;
            }
            cellB = nch_cell(n_th, n_td, n_pd, phi_periodic, maps(in, fast::dir_f32(sx, sy, sz)), a.opts, wB);
        }
        const bool validA = (wiz > 0.0f) && (woz > 0.0f);
        const bool validB = (wiz > 0.0f) && (!a.opts.sampling || sp > 0.0f);
        const bool hasB = validB && (sp > 0.0f);
        const double cA = (double)fast::cos_or_nan32(wix + wiy + wiz, wox, woy, woz, a.opts.cosine != 0);
        const double cB = (double)fast::cos_or_nan32(wix + wiy + wiz, sx, sy, sz, a.opts.cosine != 0);
        const float ps = hasB ? sp : 1.0f;

        for (int g = 0; g < groups; ++g) {                    // wave-uniform trip count
            if constexpr (HAS_EVAL) nch_copy<S>((uint64_t)(texels + ((size_t)cellA * groups + g) * S), ldsA, lane);
            if constexpr (HAS_SAMPLE) nch_copy<S>((uint64_t)(texels + ((size_t)cellB * groups + g) * S), ldsB, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's copies have landed (own wave only: no barrier)
            const int first = CPAD == 4 ? 4 * g : 0;
            if constexpr (HAS_EVAL) {
                double v[CPAD];
                nch_blend<CPAD>(ldsA, lane, wA, v, renorm);
                if (active) {
#pragma unroll
                    for (int ch = 0; ch < CPAD; ++ch)
                        if (first + ch < n_ch) a.out_rgb[i * (size_t)n_ch + first + ch] = validA ? (float)(v[ch] * cA) : 0.0f;
                }
            }
            if constexpr (HAS_SAMPLE) {
                double v[CPAD];
                nch_blend<CPAD>(ldsB, lane, wB, v, renorm);
                if (active) {
#pragma unroll
                    for (int ch = 0; ch < CPAD; ++ch)
                        if (first + ch < n_ch) a.out_weight[i * (size_t)n_ch + first + ch] = (hasB ? (float)(v[ch] * cB) : 0.0f) / ps;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // LDS reads done before the next group's copies overwrite them
        }
        if (active) {
            if constexpr (mode_pdf(MODE)) {
                float p = validA ? woz * kInvPiF : 0.0f;
                if (a.opts.sampling && validA && known) p = (float)fast::table_pdf(m, in, fast::normalize_f32(wox, woy, woz), woz, a.opts.sampling);
                a.out_pdf[i] = p;
            }
            if constexpr (HAS_SAMPLE) {
                const float wo2[3] = { validB ? sx : 0.0f, validB ? sy : 0.0f, validB ? sz : 0.0f };
                store3(a.out_wo, i, wo2);
                a.out_pdf2[i] = validB ? sp : 0.0f;
            }
        }
    }
}

// ---- 4 .. 32 channels: one lookup at a time, channel groups staged in LDS, outputs written as dense spans ---------
// A unit's outputs are n_ch adjacent floats, so a wave's 64 units own ONE contiguous span of 64 x n_ch floats per
// output array.  Writing a group's 4 channels straight from the lanes would be 64 sixteen-byte pieces 4 n_ch bytes
// apart — partial-line writes that the memory side turns into read-modify-writes (measured: 32 channels ran at
// 0.23 G units/s that way, 1/10 of what its 8 lines per lookup cost).  Instead every group's result goes to an LDS
// row per unit (row stride n_ch + 1 floats: conflict-free), and once the lookup's groups are done the wave streams
// the span out with 256-B contiguous stores.  The two lookups of a fused unit run one after the other, so that one
// copy buffer (8 KB) and one staging area serve both: 2 blocks per CU up to 32 channels.
template <int MODE, bool MULTI, bool INDEXED = false>
__global__ __launch_bounds__(kNchBlock) void k_table_nch_wide(BatchArgs a, int n_ch)
{
    static_assert(MODE != MODE_PDF, "pdf needs no table");
    constexpr bool HAS_EVAL = mode_eval(MODE);
    constexpr bool HAS_SAMPLE = mode_sample(MODE);
    constexpr int S = 8;
    extern __shared__ float4 smem[];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const int row = n_ch + 1;                                  // staging row stride in floats
    const size_t wave_f4 = 64 * S + ((size_t)64 * row + 3) / 4;
    float4 *dma = smem + wave * wave_f4;
    float *stage = (float *)(dma + 64 * S);
    const int groups = (n_ch + 3) / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n_items = nch_item_count<INDEXED>(a);
    const bool renorm = a.opts.negative == NEGATIVE_RENORMALISE;       // wave-uniform
    for (size_t base = (size_t)blockIdx.x * blockDim.x + wave * 64u; base < n_items; base += stride) {
        const size_t j = base + lane;
        const bool active = j < n_items;
        const size_t jj = active ? j : n_items - 1;          // tail lanes recompute the last unit, store nothing
        const size_t i = INDEXED ? (size_t)a.idx[jj] : jj;
        const size_t span = (n_items - base < 64 ? n_items - base : 64) * (size_t)n_ch;   // floats this wave owns per output array

        MaterialDev m;
        bool known = true;
        if constexpr (MULTI) {
            const int id = a.mat[i];
            known = id >= 0 && id < a.n_materials;
            m = a.materials[known ? id : 0];
            known = known && m.kind == KIND_TABLE_NCH && m.n_ch == n_ch;
        } else {
            m = a.single;
        }
        const float4 *texels = known ? m.texels : (const float4 *)a.materials;
        const int n_th = known ? m.n_th : 1, n_td = known ? m.n_td : 1, n_pd = known ? m.n_pd : 1;

        float wix, wiy, wiz, wox = 0.0f, woy = 0.0f, woz = 1.0f, u0 = 0.0f, u1 = 0.0f;
        load3(a.wi, i, wix, wiy, wiz);
        if (!known) wiz = 0.0f;
        if constexpr (HAS_EVAL) load3(a.wo, i, wox, woy, woz);
        if constexpr (HAS_SAMPLE) { u0 = a.u[2 * i]; u1 = a.u[2 * i + 1]; }
        const fast::Vec3 in = fast::normalize_f32(wix, wiy, wiz);
        const int param = known ? m.param : PARAM_HALF_DIFF;
        const bool phi_periodic = param_phi_periodic(param);
        const fast::TableMaps maps(n_th, n_td, n_pd, param);

        // one lookup: copy + blend every group into the staging rows (scaled by `factor`), then stream the span out
        // (guards are selects on the finished value: a non-finite direction may have poisoned the blend)
        auto lookup = [&](const fast::Dir &out_dir, bool keep, double factor, float divide, float *dst) {
            NchWeights w;
            const uint32_t cell = nch_cell(n_th, n_td, n_pd, phi_periodic, maps(in, out_dir), a.opts, w);
            for (int g = 0; g < groups; ++g) {                // wave-uniform trip count
                nch_copy<S>((uint64_t)(texels + ((size_t)cell * groups + g) * S), dma, lane);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                double v[4];
                nch_blend<4>(dma, lane, w, v, renorm);
#pragma unroll
                for (int ch = 0; ch < 4; ++ch)
                    if (4 * g + ch < n_ch) stage[lane * row + 4 * g + ch] = keep ? (float)(v[ch] * factor) / divide : 0.0f;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads of the copy buffer done before the next copy lands
            }
            if constexpr (INDEXED) {
                // queued slots are scattered: every lane writes its own row (n_ch adjacent floats, back to back)
                if (active) {
                    float *mine = dst + i * (size_t)n_ch;
                    for (int ch = 0; ch < n_ch; ++ch) mine[ch] = stage[lane * row + ch];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                return;
            }
            // the wave's span of this array: float t*64 + lane belongs to unit (t*64 + lane) / n_ch
            const int q = 64 / n_ch, r = 64 % n_ch;
            int unit = (int)lane / n_ch, ch = (int)lane % n_ch;
            float *span_base = dst + base * (size_t)n_ch;
            for (int t = 0; t < n_ch; ++t) {
                const size_t at = (size_t)t * 64 + lane;
                if (at < span) span_base[at] = stage[unit * row + ch];
                unit += q; ch += r;
                if (ch >= n_ch) { ch -= n_ch; ++unit; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // staging rows are free for the next lookup
        };

        if constexpr (HAS_EVAL) {
            const bool valid = (wiz > 0.0f) && (woz > 0.0f);
            const double c = (double)fast::cos_or_nan32(wix + wiy + wiz, wox, woy, woz, a.opts.cosine != 0);
            lookup(fast::dir_f32(wox, woy, woz), valid, c, 1.0f, a.out_rgb);
            if constexpr (mode_pdf(MODE)) {
                float p = valid ? woz * kInvPiF : 0.0f;
                if (a.opts.sampling && valid && known) p = (float)fast::table_pdf(m, in, fast::normalize_f32(wox, woy, woz), woz, a.opts.sampling);
                if (active) a.out_pdf[i] = p;
            }
        }
        if constexpr (HAS_SAMPLE) {
            float sx, sy, sz, sp;
            if (a.opts.sampling && known) {                   // option is wave-uniform
                fast::table_sample_dir(m, a.opts.disk_map, in, u0, u1, sx, sy, sz, a.opts.sampling);
                const bool up = sz > 0.0f;
                if (!up) { sx = 0.0f; sy = 0.0f; sz = 1.0f; }
                sp = up ? (float)fast::table_pdf(m, in, fast::normalize_f32(sx, sy, sz), sz, a.opts.sampling) : 0.0f;
            } else {
                square_to_cosine_hemisphere(a.opts.disk_map, u0, u1, sx, sy, sz);
                sp = sz > 0.0f ? sz * kInvPiF : 0.0f;
            }
            const bool valid = (wiz > 0.0f) && (!a.opts.sampling || sp > 0.0f);
            const bool has = valid && (sp > 0.0f);
            const double c = (double)fast::cos_or_nan32(wix + wiy + wiz, sx, sy, sz, a.opts.cosine != 0);
            lookup(fast::dir_f32(sx, sy, sz), has, c, has ? sp : 1.0f, a.out_weight);
            if (active) {
                const float wo2[3] = { valid ? sx : 0.0f, valid ? sy : 0.0f, valid ? sz : 0.0f };
                store3(a.out_wo, i, wo2);
                a.out_pdf2[i] = valid ? sp : 0.0f;
            }
        }
    }
}

// ---- upload: planar f64 (n_ch planes, file order) -> n-channel bricks.  One thread per (cell, channel group). ----
template <int CPAD>
__global__ __launch_bounds__(kNchBlock) void k_build_bricks_nch(const double *planar, const double *scale, int n_th, int n_td, int n_pd,
                                                               int phi_periodic, int n_ch, int clamp, float4 *bricks)
{
    constexpr int S = 2 * CPAD;
    const int groups = CPAD == 4 ? (n_ch + 3) / 4 : 1;
    const size_t cells = (size_t)n_th * n_td * n_pd, plane = cells, total = cells * groups;
    const size_t stride = (size_t)gridDim.x * kNchBlock;
    for (size_t t = (size_t)blockIdx.x * kNchBlock + threadIdx.x; t < total; t += stride) {
        const size_t c = t / groups;
        const int g = (int)(t % groups);
        const int ip = (int)(c % (size_t)n_pd), id = (int)((c / (size_t)n_pd) % (size_t)n_td), ih = (int)(c / ((size_t)n_pd * n_td));
        float v[8 * CPAD];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int sh = min(ih + (k >> 2), n_th - 1), sd = min(id + ((k >> 1) & 1), n_td - 1),
                      sp = phi_periodic ? (ip + (k & 1)) % n_pd : min(ip + (k & 1), n_pd - 1);
            const size_t src = ((size_t)sh * n_td + sd) * n_pd + sp;
#pragma unroll
            for (int ch = 0; ch < CPAD; ++ch) {
                const int cidx = (CPAD == 4 ? 4 * g : 0) + ch;
                float out = 0.0f;
                if (cidx < n_ch) {
                    const double x = planar[src + (size_t)cidx * plane] * scale[cidx];
                    out = (x > 0.0 || !clamp) ? (float)x : 0.0f;          // negatives clamp to 0 unless MRL_OPT_NEGATIVE keeps them, as in the RGB path
                }
                v[k * CPAD + ch] = out;
            }
        }
        float4 *dst = bricks + t * S;
#pragma unroll
        for (int q = 0; q < S; ++q) dst[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
}

template <int MODE, bool MULTI, bool INDEXED>
hipError_t launch_nch_cpad(const BatchArgs &a, int n_ch, int compute_units, hipStream_t stream)
{
    size_t blocks = (a.n + kNchBlock - 1) / kNchBlock;
    const size_t cap = (size_t)compute_units * (MODE == MODE_EVAL_SAMPLE ? 2 : 4);
    if (blocks > cap) blocks = cap;
    const dim3 g((unsigned)blocks), b(kNchBlock);
    if (n_ch == 1)      hipLaunchKernelGGL((k_table_nch<MODE, MULTI, 1, INDEXED>), g, b, 0, stream, a, n_ch);
    else if (n_ch == 2) hipLaunchKernelGGL((k_table_nch<MODE, MULTI, 2, INDEXED>), g, b, 0, stream, a, n_ch);
    else if (n_ch == 4) hipLaunchKernelGGL((k_table_nch<MODE, MULTI, 4, INDEXED>), g, b, 0, stream, a, n_ch);   // one line, dense 16-B stores: both lookups in flight
    else {
        // per wave: one 8 KB copy buffer + 64 staging rows of n_ch + 1 floats (rounded up to float4s)
        const size_t wave_f4 = 64 * 8 + ((size_t)64 * (n_ch + 1) + 3) / 4;
        // 4 waves per block while that fits 64 KB of LDS (up to 28 channels: 41 .. 62.5 KB), 2 waves beyond (32: 33.3 KB)
        const unsigned threads = (kNchBlock / 64) * wave_f4 * sizeof(float4) <= 65536 ? kNchBlock : kNchBlock / 2;
        const size_t lds = (threads / 64) * wave_f4 * sizeof(float4);
        size_t wide_blocks = (a.n + threads - 1) / threads;
        const size_t wide_cap = (size_t)compute_units * (threads == kNchBlock ? 2 : 4);       // 8 waves per CU either way
        if (wide_blocks > wide_cap) wide_blocks = wide_cap;
        hipLaunchKernelGGL((k_table_nch_wide<MODE, MULTI, INDEXED>), dim3((unsigned)wide_blocks), dim3(threads), lds, stream, a, n_ch);
    }
    return hipGetLastError();
}

template <int MODE>
hipError_t launch_nch_mode(const BatchArgs &a, bool multi, bool indexed, int n_ch, int compute_units, hipStream_t stream)
{
    if (indexed) return multi ? launch_nch_cpad<MODE, true, true>(a, n_ch, compute_units, stream) : launch_nch_cpad<MODE, false, true>(a, n_ch, compute_units, stream);
    return multi ? launch_nch_cpad<MODE, true, false>(a, n_ch, compute_units, stream) : launch_nch_cpad<MODE, false, false>(a, n_ch, compute_units, stream);
}

} // namespace

hipError_t launch_batch_nch(int mode, const BatchArgs &a, bool multi, int n_ch, int compute_units, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    if (n_ch < 1 || n_ch > kMaxChannels || n_ch == 3) return hipErrorInvalidValue;
    const bool indexed = a.idx != nullptr;                     // a caller's wavefront queue (a.n = its capacity)
    switch (mode) {
        case MODE_EVAL:        return launch_nch_mode<MODE_EVAL>(a, multi, indexed, n_ch, compute_units, stream);
        case MODE_SAMPLE:      return launch_nch_mode<MODE_SAMPLE>(a, multi, indexed, n_ch, compute_units, stream);
        case MODE_EVAL_SAMPLE: return launch_nch_mode<MODE_EVAL_SAMPLE>(a, multi, indexed, n_ch, compute_units, stream);
        case MODE_EVAL_PDF:    return launch_nch_mode<MODE_EVAL_PDF>(a, multi, indexed, n_ch, compute_units, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_build_table_nch(const double *d_planar, const double *d_scale, const int dims[3], int n_ch, int param, int clamp, float4 *d_out,
                                  int compute_units, hipStream_t stream)
{
    const size_t cells = (size_t)dims[0] * dims[1] * dims[2];
    const size_t total = cells * (n_ch > 2 ? (size_t)((n_ch + 3) / 4) : 1);
    size_t blocks = (total + kNchBlock - 1) / kNchBlock;
    if (blocks > (size_t)compute_units * 8) blocks = (size_t)compute_units * 8;
    if (blocks < 1) blocks = 1;
    const dim3 g((unsigned)blocks), b(kNchBlock);
    if (n_ch == 1)      hipLaunchKernelGGL((k_build_bricks_nch<1>), g, b, 0, stream, d_planar, d_scale, dims[0], dims[1], dims[2], (int)param_phi_periodic(param), n_ch, clamp, d_out);
    else if (n_ch == 2) hipLaunchKernelGGL((k_build_bricks_nch<2>), g, b, 0, stream, d_planar, d_scale, dims[0], dims[1], dims[2], (int)param_phi_periodic(param), n_ch, clamp, d_out);
    else                hipLaunchKernelGGL((k_build_bricks_nch<4>), g, b, 0, stream, d_planar, d_scale, dims[0], dims[1], dims[2], (int)param_phi_periodic(param), n_ch, clamp, d_out);
    return hipGetLastError();
}

} // namespace mrl
