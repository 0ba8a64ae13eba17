// merl_rgl.hip — the adaptive-parameterisation measured BSDF (RGL *.bsdf; upstream Mitsuba 3 `measured`) on gfx950:
// the host-side image builder (normalisation + running integrals, f64, in the oracle's loop order) and the kernels.
// PARITY UNPINNED (see merl_rgl.hpp).  SURVEY.md §8f item 3.
//
//   k_rgl<MODE, INDEXED, MULTI, MASK>          one lane = one unit, grid-stride; the material's descriptor arrives by value in SGPRs
//                                              (MULTI: read per unit from behind the unit's material); the distributions' running
//                                              integrals come from the cell records in memory
//   k_rgl_lds<MODE, INDEXED, MARG_ONLY, MASK>  one workgroup per CU first copies those integrals (or the marginal rows alone) into LDS
//   k_rgl_spectral<MODE, LDS, MASK>            the same for spectral files, W values per unit
// MASK: the bracket shape the kernel is compiled for (5 isotropic, 15 anisotropic, 0 tested at run time).  What bounds them — round
// trips, then L1 line fills, then (isotropic) the vector ALU — and the measured rates: merl_rgl.hpp's header, DESIGN.md §5e.
#include "merl_kernels.hpp"
#include "merl_rgl.hpp"

#include <cmath>
#include <vector>

namespace mrl {

namespace {

constexpr int kRglBlock = 256;
// the LDS variant: ONE workgroup per CU holds the file's running integrals (up to 160 KB) — 4 waves per SIMD for eval and for pdf alone,
// 3 for the modes with a sample() (which carries two visited cells through its eval: 160 - 170 VGPRs), 2 for the anisotropic shape's
#ifndef MRL_RGL_LDS_BLOCK_SAMPLE
#define MRL_RGL_LDS_BLOCK_SAMPLE 768
#endif
#ifndef MRL_RGL_LDS_BLOCK_EVAL
#define MRL_RGL_LDS_BLOCK_EVAL 1024
#endif
#ifndef MRL_RGL_LDS_BLOCK_SAMPLE15
#define MRL_RGL_LDS_BLOCK_SAMPLE15 512
#endif
#ifndef MRL_RGL_LDS_BLOCK_EVALPDF5
#define MRL_RGL_LDS_BLOCK_EVALPDF5 768
#endif
constexpr int rgl_lds_block(int mode, int mask = 0)
{
    if (mode == 4 && mask == 5) return MRL_RGL_LDS_BLOCK_EVALPDF5;
    return mode >= 2 ? (mask == 15 ? MRL_RGL_LDS_BLOCK_SAMPLE15 : MRL_RGL_LDS_BLOCK_SAMPLE) : MRL_RGL_LDS_BLOCK_EVAL;
}

// ---- the running integrals in LDS ----
// Slice by slice (taken out of the records while they are copied: SearchLds below), for vndf and then luminance.  A ds_read gather of 64
// scattered addresses costs no line fill, and these are the reads sample() is made of: 2 x (log2 ny + log2 nx) dependent steps per unit.
extern __shared__ float4 rgl_lds[];

// ---- the parameter grids in LDS (every single-material kernel): phi_i then theta_i at the start of the block's LDS ----
struct GridLds {
    unsigned theta_at_, wl_at_;         // float index of theta_i's / the wavelengths' first node in rgl_lds (phi_i starts at 0)
    __device__ __forceinline__ float phi_at(int k) const { return ((const float *)rgl_lds)[k]; }
    __device__ __forceinline__ float theta_at(int k) const { return ((const float *)rgl_lds)[theta_at_ + (unsigned)k]; }
    __device__ __forceinline__ float wavelength_at(int k) const { return ((const float *)rgl_lds)[wl_at_ + (unsigned)k]; }
};
size_t grid_bytes_of(const RglDev &r) { return ((size_t)(r.n_phi + r.n_theta + r.n_wl) * sizeof(float) + 15) / 16 * 16; }
__device__ __forceinline__ GridLds stage_grids(const RglDev &r, unsigned &at_float4, int block)
{
    float *g = (float *)rgl_lds;
    for (int k = threadIdx.x; k < r.n_phi; k += block) g[k] = r.phi[k];
    for (int k = threadIdx.x; k < r.n_theta; k += block) g[r.n_phi + k] = r.theta[k];
    for (int k = threadIdx.x; k < r.n_wl; k += block) g[r.n_phi + r.n_theta + k] = r.wavelengths[k];
    at_float4 += (unsigned)((r.n_phi + r.n_theta + r.n_wl + 3) / 4);
    return GridLds{ (unsigned)r.n_phi, (unsigned)(r.n_phi + r.n_theta) };
}

// Slice by slice, taken out of the records while they are copied: left [slices][cell] float2 (node rows row / row + 1, up to node col),
// total [slices][ny - 1] float2, marg [slices][ny - 1] float — for vndf and then luminance.
struct SearchLds {
    unsigned left_at, total_at, marg_at;    // float2 / float2 / float index of the table's first element in rgl_lds
    unsigned per_row;
    // reads, then sums (rgl::fetch_raw), slice by slice
    struct PairRaw { float2 q0, q1, q2, q3; };
    struct MargRaw { float m0, m1, m2, m3; };
    __device__ __forceinline__ PairRaw pair_at(const rgl::Slices &s, const float2 *t, const unsigned (&off)[4]) const
    {
        PairRaw r;
        r.q1 = r.q2 = r.q3 = make_float2(0.0f, 0.0f);
        r.q0 = t[off[0]];
        if (s.mask & 2) r.q1 = t[off[1]];
        if (s.mask & 4) r.q2 = t[off[2]];
        if (s.mask & 8) r.q3 = t[off[3]];
        return r;
    }
    __device__ __forceinline__ PairRaw left_raw(const rgl::Slices &s, int cell) const { return pair_at(s, (const float2 *)rgl_lds + left_at + (unsigned)cell, s.soff); }
    __device__ __forceinline__ PairRaw total_raw(const rgl::Slices &s, int row) const { return pair_at(s, (const float2 *)rgl_lds + total_at + (unsigned)row, s.roff); }
    __device__ __forceinline__ rgl::D2 pair_blend(const rgl::Slices &s, const PairRaw &r) const
    {
#pragma clang fp contract(off)
        rgl::D2 v = { s.w[0] * (double)r.q0.x, s.w[0] * (double)r.q0.y };
        if (s.mask & 2) { v.x = __builtin_fma(s.w[1], (double)r.q1.x, v.x); v.y = __builtin_fma(s.w[1], (double)r.q1.y, v.y); }
        if (s.mask & 4) { v.x = __builtin_fma(s.w[2], (double)r.q2.x, v.x); v.y = __builtin_fma(s.w[2], (double)r.q2.y, v.y); }
        if (s.mask & 8) { v.x = __builtin_fma(s.w[3], (double)r.q3.x, v.x); v.y = __builtin_fma(s.w[3], (double)r.q3.y, v.y); }
        return v;
    }
    __device__ __forceinline__ MargRaw marg_raw(const rgl::Slices &s, int row) const
    {
        const float *t = (const float *)rgl_lds + marg_at + (unsigned)row;
        MargRaw r;
        r.m1 = r.m2 = r.m3 = 0.0f;
        r.m0 = t[s.roff[0]];
        if (s.mask & 2) r.m1 = t[s.roff[1]];
        if (s.mask & 4) r.m2 = t[s.roff[2]];
        if (s.mask & 8) r.m3 = t[s.roff[3]];
        return r;
    }
    __device__ __forceinline__ double marg_blend(const rgl::Slices &s, const MargRaw &r) const
    {
#pragma clang fp contract(off)
        double v = s.w[0] * (double)r.m0;
        if (s.mask & 2) v = __builtin_fma(s.w[1], (double)r.m1, v);
        if (s.mask & 4) v = __builtin_fma(s.w[2], (double)r.m2, v);
        if (s.mask & 8) v = __builtin_fma(s.w[3], (double)r.m3, v);
        return v;
    }
    struct HeadRaw { PairRaw total, p1, p2a, p2b; };
    // (the memory form's row header, from the slice-major copies: the four reads go out together)
    __device__ __forceinline__ HeadRaw head_raw(const rgl::Slices &s, int row) const
    {
        const rgl::Pivots pv = rgl::pivots_of(0, (int)per_row - 1);
        const int first = row * (int)per_row + 1;
        HeadRaw h;
        h.total = total_raw(s, row); h.p1 = left_raw(s, first + pv.m1); h.p2a = left_raw(s, first + pv.m2a); h.p2b = left_raw(s, first + pv.m2b);
        return h;
    }
    __device__ __forceinline__ HeadRaw quarter_raw(const rgl::Slices &s, int row, int, int lo, int hi) const
    {
        const rgl::Pivots pv = rgl::pivots_of(lo, hi);
        const int first = row * (int)per_row + 1;
        HeadRaw h;
        h.p1 = left_raw(s, first + pv.m1); h.p2a = left_raw(s, first + pv.m2a); h.p2b = left_raw(s, first + pv.m2b);
        h.total = h.p1;
        return h;
    }
    __device__ __forceinline__ rgl::D2 left(const rgl::Slices &s, int cell) const { return pair_blend(s, left_raw(s, cell)); }
    __device__ __forceinline__ rgl::D2 total(const rgl::Slices &s, int row) const { return pair_blend(s, total_raw(s, row)); }
    __device__ __forceinline__ double marg(const rgl::Slices &s, int row) const { return marg_blend(s, marg_raw(s, row)); }
};

// The partial form for files whose records do not fit (an anisotropic 16 x 8 x 32 x 32 file: 12.9 MB per distribution): the
// MARGINAL rows alone, in their bracket form (one ds_read_b128 per row-search step: 52 KB per distribution for that file), the
// conditional integrals from the records in memory.  Takes the row searches — 5 of the ~21 dependent reads of a forward warp — off
// the vector memory pipe.
struct SearchLdsMarg : rgl::SearchMem {
    unsigned marg_at;                   // float4 index of the table's first quad in rgl_lds
    __device__ __forceinline__ SearchLdsMarg(const WarpDev &w, unsigned at) : rgl::SearchMem(w), marg_at(at) {}
    __device__ __forceinline__ MargRaw marg_raw(const rgl::Slices &s, int row) const { return rgl_lds[marg_at + s.quad + (unsigned)row]; }
    __device__ __forceinline__ double marg(const rgl::Slices &s, int row) const { return marg_blend(s, marg_raw(s, row)); }
};

size_t lds_marg_bytes_of(const RglDev &r)
{
    const WarpDev w = r.vndf();
    const size_t tb = w.n_theta > 1 ? w.n_theta - 1 : 1, pb = w.n_phi > 1 ? w.n_phi - 1 : 1;
    return grid_bytes_of(r) + 2 * pb * tb * (size_t)(w.ny - 1) * sizeof(float4);
}

__device__ __forceinline__ SearchLdsMarg stage_marg(const WarpDev &w, unsigned &at_float4, int block)
{
    const int per_r = w.ny - 1;
    const int tb = w.n_theta > 1 ? w.n_theta - 1 : 1, pb = w.n_phi > 1 ? w.n_phi - 1 : 1;
    const SearchLdsMarg t(w, at_float4);
    const int n = pb * tb * per_r;
    for (int k = threadIdx.x; k < n; k += block) rgl_lds[at_float4 + (unsigned)k] = w.margq[k];
    at_float4 += (unsigned)n;
    return t;
}

// bytes of LDS the two distributions' running integrals take, slice by slice (16-B aligned pieces)
size_t lds_bytes_of(const RglDev &r)
{
    const WarpDev w = r.vndf();
    const size_t slices = (size_t)w.n_phi * (size_t)w.n_theta, per_c = (size_t)(w.nx - 1) * (size_t)(w.ny - 1), per_r = (size_t)(w.ny - 1);
    const size_t left = (slices * per_c * 8 + 15) / 16 * 16, total = (slices * per_r * 8 + 15) / 16 * 16, marg = (slices * per_r * 4 + 15) / 16 * 16;
    return grid_bytes_of(r) + 2 * (left + total + marg);
}

// one distribution's tables: memory (records, bracket-major) -> LDS (slice-major); every thread of the block takes part
__device__ __forceinline__ SearchLds stage_search(const WarpDev &w, unsigned &at_float4, int block)
{
    const int per_c = (w.nx - 1) * (w.ny - 1), per_r = w.ny - 1, slices = w.n_phi * w.n_theta;
    const int tb = w.n_theta > 1 ? w.n_theta - 1 : 1, pb = w.n_phi > 1 ? w.n_phi - 1 : 1;
    const unsigned stride = (unsigned)w.stride, totals_at = (unsigned)(w.phi_nodes() + w.slices());
    SearchLds t;
    t.per_row = (unsigned)(w.nx - 1);
    t.left_at = at_float4 * 2;                                   // in float2
    at_float4 += (unsigned)((slices * per_c + 1) / 2);
    t.total_at = at_float4 * 2;                                  // in float2
    at_float4 += (unsigned)((slices * per_r + 1) / 2);
    t.marg_at = at_float4 * 4;                                   // in float
    at_float4 += (unsigned)((slices * per_r + 3) / 4);
    float2 *left = (float2 *)rgl_lds + t.left_at, *total = (float2 *)rgl_lds + t.total_at;
    float *marg = (float *)rgl_lds + t.marg_at;
    // slice (ip, it) is read from a bracket it bounds: phi node ip - ipb of the bracket's pair, .xy / .zw by it - itb
    for (int k = threadIdx.x; k < slices * per_c; k += block) {
        const int sl = k / per_c, cell = k - sl * per_c, ip = sl / w.n_theta, it = sl - ip * w.n_theta;
        const int itb = it < tb ? it : tb - 1, ipb = ip < pb ? ip : pb - 1;
        const float4 v = w.cells[((unsigned)(ipb * tb + itb) * (unsigned)per_c + (unsigned)cell) * stride + (unsigned)(ip - ipb)];
        left[k] = it > itb ? make_float2(v.z, v.w) : make_float2(v.x, v.y);
    }
    for (int k = threadIdx.x; k < slices * per_r; k += block) {
        const int sl = k / per_r, row = k - sl * per_r, ip = sl / w.n_theta, it = sl - ip * w.n_theta;
        const int itb = it < tb ? it : tb - 1, ipb = ip < pb ? ip : pb - 1;
        const float4 tv = w.cells[((unsigned)(ipb * tb + itb) * (unsigned)per_c + (unsigned)row * t.per_row) * stride + totals_at + (unsigned)(ip - ipb)];
        total[k] = it > itb ? make_float2(tv.z, tv.w) : make_float2(tv.x, tv.y);
        const float4 v = w.margq[(unsigned)(ipb * tb + itb) * (unsigned)per_r + (unsigned)row];
        const int c = (ip - ipb) + 2 * (it - itb);
        marg[k] = c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
    }
    return t;
}

// One unit.  What depends on the incident direction alone — its angles, the parameter bracket, the projected area — is formed once
// and serves the eval, the pdf and the sample of the unit.
// g: where the parameter grids are read (GridLds: the launch's one material; rgl::GridMem: a batch with ids)
// MASK: which slices a parameter bracket of the launch's file has, known when the kernel is compiled (5: isotropic — theta_i alone;
// 15: anisotropic; 0: decided per launch) — every test on it then folds, and the reads of a lookup are straight-line code
template <int MODE, int MASK, class Grids, class Search>
__device__ __forceinline__ void rgl_unit(const BatchArgs &a, const RglDev &r, const Grids &g, const Search &tv, const Search &tl, size_t i)
{
    constexpr bool has_eval = MODE == 0 || MODE == 3 || MODE == 4, has_pdf = MODE == 1 || MODE == 3 || MODE == 4,
                   has_sample = MODE == 2 || MODE == 3;
    float wix = a.wi[3 * i], wiy = a.wi[3 * i + 1], wiz = a.wi[3 * i + 2];
    // (the unit's other stream reads are issued before the incident-only work waits for its own)
    float wox = 0.0f, woy = 0.0f, woz = 0.0f, u0 = 0.0f, u1 = 0.0f;
    if constexpr (has_eval || has_pdf) { wox = a.wo[3 * i]; woy = a.wo[3 * i + 1]; woz = a.wo[3 * i + 2]; }
    if constexpr (has_sample) { u0 = a.u[2 * i]; u1 = a.u[2 * i + 1]; }
    // (all three components of wi are wanted HERE: without this the compiler reads wi.z, tests it, and only then reads wi.xy — a round trip more)
    asm volatile("" : "+v"(wix), "+v"(wiy), "+v"(wiz));
    rgl::Incident in;
    const bool up = wiz > 0.0f && rgl::incident<has_eval || has_sample>(r, g, wix, wiy, wiz, in);
    if constexpr (MASK != 0) in.sv.mask = MASK;
    if constexpr (has_eval || has_pdf) {
        float rgb[3] = { 0.0f, 0.0f, 0.0f }, pdf = 0.0f;
        if (up) rgl::eval_pdf_at<has_eval, has_pdf>(r, tv, in, wox, woy, woz, rgb, pdf);
        if constexpr (has_eval) { a.out_rgb[3 * i] = rgb[0]; a.out_rgb[3 * i + 1] = rgb[1]; a.out_rgb[3 * i + 2] = rgb[2]; }
        if constexpr (has_pdf) a.out_pdf[i] = pdf;
    }
    if constexpr (has_sample) {
        float wo2[3] = { 0.0f, 0.0f, 0.0f }, pdf2 = 0.0f, w[3] = { 0.0f, 0.0f, 0.0f };
        if (up) rgl::sample_at(r, tv, tl, in, u0, u1, wo2, pdf2, w);
        a.out_wo[3 * i] = wo2[0]; a.out_wo[3 * i + 1] = wo2[1]; a.out_wo[3 * i + 2] = wo2[2];
        a.out_pdf2[i] = pdf2;
        a.out_weight[3 * i] = w[0]; a.out_weight[3 * i + 1] = w[1]; a.out_weight[3 * i + 2] = w[2];
    }
}

// MULTI: a batch with a material id per unit — the lanes whose id names an RGL material evaluate it through the descriptor
// stored behind that material's image (read on demand: a few more cache-resident loads per lookup) and overwrite the zeros
// the table / GGX kernel of the same call left there; every other lane skips.  Launched after that kernel, on the same stream.
// Blocks per CU the compiler is asked to make room for (= waves per SIMD).  With a lookup's reads in flight together a unit holds
// 12 - 16 vectors between the reads and the sums, and a spill costs more than a wave: per bracket shape (the kernels compiled for
// one) the largest bound without scratch — measured side by side with tools/rgl_ab_time.py; the run-time-shape kernels (batches with
// ids, odd shapes) keep the bounds they had, but for the fused unit inside a batch with ids (25 spilled registers at 3: 3.62 -> 3.32 ms
// at 2 on the mixed batch of tools/rgl_rates.py).
#ifndef MRL_RGL_EVALPDF_BLOCKS
#define MRL_RGL_EVALPDF_BLOCKS 4
#endif
#ifndef MRL_RGL_SAMPLE_BLOCKS
#define MRL_RGL_SAMPLE_BLOCKS 3
#endif
#ifndef MRL_RGL_EVAL_BLOCKS
#define MRL_RGL_EVAL_BLOCKS 4
#endif
#ifndef MRL_RGL_MULTI_FUSED_BLOCKS
#define MRL_RGL_MULTI_FUSED_BLOCKS 2
#endif
#ifndef MRL_RGL_BLOCKS15
#define MRL_RGL_BLOCKS15 3, 4, 2, 2, 3          // eval, pdf, sample, fused, eval + pdf of the anisotropic shape (four slices in flight per lookup)
#endif
#ifndef MRL_RGL_BLOCKS5
#define MRL_RGL_BLOCKS5 4, 4, 3, 3, 4
#endif
constexpr int rgl_min_blocks(int mode, bool multi, int mask)
{
    constexpr int b15[5] = { MRL_RGL_BLOCKS15 }, b5[5] = { MRL_RGL_BLOCKS5 };
    if (mask == 15) return b15[mode];
    if (mask == 5) return b5[mode];
    return mode == 3 ? (multi ? MRL_RGL_MULTI_FUSED_BLOCKS : 2) : (mode == 2 ? MRL_RGL_SAMPLE_BLOCKS : (mode == 4 ? MRL_RGL_EVALPDF_BLOCKS : MRL_RGL_EVAL_BLOCKS));
}
template <int MODE, bool INDEXED, bool MULTI, int MASK = 0>
__global__ __launch_bounds__(kRglBlock, rgl_min_blocks(MODE, MULTI, MASK)) void k_rgl(BatchArgs a, RglDev r)
{
    const size_t stride = (size_t)gridDim.x * kRglBlock;
    size_t n_items = a.n;
    if constexpr (INDEXED) { const size_t c = (size_t)*a.idx_count; n_items = c < a.n ? c : a.n; }
    GridLds grids{ 0u, 0u };
    if constexpr (!MULTI) {
        unsigned at = 0;
        grids = stage_grids(r, at, kRglBlock);
        __syncthreads();
    }
    for (size_t j = (size_t)blockIdx.x * kRglBlock + threadIdx.x; j < n_items; j += stride) {
        const size_t i = INDEXED ? (size_t)a.idx[j] : j;
        if constexpr (MULTI) {
            const int id = a.mat[i];
            if (id < 0 || id >= a.n_materials) continue;
            const MaterialDev &m = a.materials[id];
            if (m.kind != KIND_RGL) continue;
            // the unit's descriptor into registers, whole (31 dwords read together): through a reference every table read would be
            // two dependent loads — the pointer, then the data
            const RglDev rm = *(const RglDev *)m.rgl;
            rgl_unit<MODE, 0>(a, rm, rgl::GridMem{ rm.phi, rm.theta, rm.wavelengths }, rgl::SearchMem(rm.vndf()), rgl::SearchMem(rm.luminance()), i);
        } else {
            rgl_unit<MODE, MASK>(a, r, grids, rgl::SearchMem(r.vndf()), rgl::SearchMem(r.luminance()), i);
        }
    }
}

// The single-material launch when the file's running integrals fit a CU's LDS: one workgroup per CU copies them in (once: the grid is
// persistent) and every search step of every unit is a ds_read.  Same functions, same sums, same bits as k_rgl.
// MARG_ONLY: the partial form (SearchLdsMarg) for files whose conditional integrals do not fit
template <int MODE, bool INDEXED, bool MARG_ONLY = false, int MASK = 0>
__global__ __launch_bounds__(rgl_lds_block(MODE, MASK)) void k_rgl_lds(BatchArgs a, RglDev r)
{
    constexpr int kRglLdsBlock = rgl_lds_block(MODE, MASK);
    const size_t stride = (size_t)gridDim.x * kRglLdsBlock;
    size_t n_items = a.n;
    if constexpr (INDEXED) { const size_t c = (size_t)*a.idx_count; n_items = c < a.n ? c : a.n; }
    unsigned at = 0;
    const GridLds grids = stage_grids(r, at, kRglLdsBlock);
    if constexpr (MARG_ONLY) {
        const SearchLdsMarg tv = stage_marg(r.vndf(), at, kRglLdsBlock);
        const SearchLdsMarg tl = stage_marg(r.luminance(), at, kRglLdsBlock);
        __syncthreads();
        for (size_t j = (size_t)blockIdx.x * kRglLdsBlock + threadIdx.x; j < n_items; j += stride) rgl_unit<MODE, MASK>(a, r, grids, tv, tl, INDEXED ? (size_t)a.idx[j] : j);
    } else {
        const SearchLds tv = stage_search(r.vndf(), at, kRglLdsBlock);
        const SearchLds tl = stage_search(r.luminance(), at, kRglLdsBlock);
        __syncthreads();
        for (size_t j = (size_t)blockIdx.x * kRglLdsBlock + threadIdx.x; j < n_items; j += stride) rgl_unit<MODE, MASK>(a, r, grids, tv, tl, INDEXED ? (size_t)a.idx[j] : j);
    }
}

// ---- spectral files: W values per unit at the wavelengths wl[i * W .. ) (nullptr: the file's own nodes) ----
// a.out_rgb / a.out_weight hold n x W values.  Same structure as rgl_unit: what depends on wi alone is formed once.
template <int MODE, int MASK, class Search>
__device__ __forceinline__ void rgl_unit_spectral(const BatchArgs &a, const RglDev &r, const GridLds &g, const Search &tv, const Search &tl, size_t i, const float *wl_all, int W)
{
    constexpr bool has_eval = MODE == 0 || MODE == 3 || MODE == 4, has_pdf = MODE == 1 || MODE == 3 || MODE == 4,
                   has_sample = MODE == 2 || MODE == 3;
    const float wix = a.wi[3 * i], wiy = a.wi[3 * i + 1], wiz = a.wi[3 * i + 2];
    const float *wl = wl_all ? wl_all + i * (size_t)W : nullptr;
    rgl::Incident in;
    const bool up = wiz > 0.0f && rgl::incident<has_eval || has_sample>(r, g, wix, wiy, wiz, in);
    if constexpr (MASK != 0) in.sv.mask = MASK;
    if constexpr (has_eval || has_pdf) {
        const float wox = a.wo[3 * i], woy = a.wo[3 * i + 1], woz = a.wo[3 * i + 2];
        float *values = has_eval ? a.out_rgb + i * (size_t)W : nullptr;
        float pdf = 0.0f;
        if (up) rgl::eval_pdf_spectral_at<has_eval, has_pdf>(r, g, tv, in, wox, woy, woz, wl, W, values, pdf);
        else if constexpr (has_eval) for (int k = 0; k < W; ++k) values[k] = 0.0f;
        if constexpr (has_pdf) a.out_pdf[i] = pdf;
    }
    if constexpr (has_sample) {
        float wo2[3] = { 0.0f, 0.0f, 0.0f }, pdf2 = 0.0f;
        float *weight = a.out_weight + i * (size_t)W;
        if (up) rgl::sample_spectral_at(r, g, tv, tl, in, a.u[2 * i], a.u[2 * i + 1], wl, W, wo2, pdf2, weight);
        else for (int k = 0; k < W; ++k) weight[k] = 0.0f;
        a.out_wo[3 * i] = wo2[0]; a.out_wo[3 * i + 1] = wo2[1]; a.out_wo[3 * i + 2] = wo2[2];
        a.out_pdf2[i] = pdf2;
    }
}

template <int MODE, bool LDS, int MASK = 0>
__global__ __launch_bounds__(LDS ? rgl_lds_block(MODE) : kRglBlock) void k_rgl_spectral(BatchArgs a, RglDev r, const float *wl, int W)
{
    constexpr int kBlockThreads = LDS ? rgl_lds_block(MODE) : kRglBlock;
    const size_t stride = (size_t)gridDim.x * kBlockThreads;
    unsigned at = 0;
    const GridLds grids = stage_grids(r, at, kBlockThreads);
    if constexpr (LDS) {
        const SearchLds tv = stage_search(r.vndf(), at, kBlockThreads);
        const SearchLds tl = stage_search(r.luminance(), at, kBlockThreads);
        __syncthreads();
        for (size_t i = (size_t)blockIdx.x * kBlockThreads + threadIdx.x; i < a.n; i += stride) rgl_unit_spectral<MODE, MASK>(a, r, grids, tv, tl, i, wl, W);
    } else {
        __syncthreads();
        for (size_t i = (size_t)blockIdx.x * kBlockThreads + threadIdx.x; i < a.n; i += stride)
            rgl_unit_spectral<MODE, MASK>(a, r, grids, rgl::SearchMem(r.vndf()), rgl::SearchMem(r.luminance()), i, wl, W);
    }
}

int lds_limit()
{
    static const int limit = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
        return v;
    }();
    return limit;
}

template <int MODE, int MASK>
hipError_t launch_masked(const BatchArgs &a, const RglDev *r, bool indexed, int search, int compute_units, hipStream_t stream)
{
    // (kernels compiled for a bracket shape: the full LDS form for isotropic files — anisotropic ones rarely fit —, the marginal-rows
    // form for anisotropic ones; the other combinations take the kernel that tests the shape at run time)
    constexpr int kLdsMask = MASK == 5 ? 5 : 0, kMargMask = MASK == 15 ? 15 : 0;
    // LDS variant: a single-material launch large enough to pay for the copy (one image of the running integrals per CU)
    if (search == 0 && a.n >= (size_t)1 << 15) {
        // one workgroup per CU
        auto grid_of = [&](int threads) { size_t blocks = (a.n + (size_t)threads - 1) / (size_t)threads; return dim3((unsigned)(blocks > (size_t)compute_units ? (size_t)compute_units : blocks)); };
        const size_t need = lds_bytes_of(*r), need_marg = lds_marg_bytes_of(*r);
        if (need <= (size_t)lds_limit()) {
            constexpr int kThreads = rgl_lds_block(MODE, kLdsMask);
            if (indexed) {
                (void)hipFuncSetAttribute((const void *)k_rgl_lds<MODE, true, false, kLdsMask>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
                hipLaunchKernelGGL((k_rgl_lds<MODE, true, false, kLdsMask>), grid_of(kThreads), dim3(kThreads), need, stream, a, *r);
            } else {
                (void)hipFuncSetAttribute((const void *)k_rgl_lds<MODE, false, false, kLdsMask>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
                hipLaunchKernelGGL((k_rgl_lds<MODE, false, false, kLdsMask>), grid_of(kThreads), dim3(kThreads), need, stream, a, *r);
            }
            return hipGetLastError();
        }
        // the marginal rows alone — sample() alone: eval / pdf read one marginal value per unit (not worth a copy per CU), and the fused
        // unit of such a file is issued as two launches (launch_rgl)
        if constexpr (MODE == 2) {
            if (need_marg <= (size_t)lds_limit()) {
                constexpr int kThreads = rgl_lds_block(MODE, kMargMask);
                if (indexed) {
                    (void)hipFuncSetAttribute((const void *)k_rgl_lds<MODE, true, true, kMargMask>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need_marg);
                    hipLaunchKernelGGL((k_rgl_lds<MODE, true, true, kMargMask>), grid_of(kThreads), dim3(kThreads), need_marg, stream, a, *r);
                } else {
                    (void)hipFuncSetAttribute((const void *)k_rgl_lds<MODE, false, true, kMargMask>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need_marg);
                    hipLaunchKernelGGL((k_rgl_lds<MODE, false, true, kMargMask>), grid_of(kThreads), dim3(kThreads), need_marg, stream, a, *r);
                }
                return hipGetLastError();
            }
        }
    }
    size_t blocks = (a.n + kRglBlock - 1) / kRglBlock;
    const size_t cap = (size_t)compute_units * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const dim3 grid((unsigned)blocks), block(kRglBlock);
    const size_t grids = grid_bytes_of(*r);                     // (at most 32 KB: 4,096 nodes per grid)
    if (indexed) hipLaunchKernelGGL((k_rgl<MODE, true, false, MASK>), grid, block, grids, stream, a, *r);
    else hipLaunchKernelGGL((k_rgl<MODE, false, false, MASK>), grid, block, grids, stream, a, *r);
    return hipGetLastError();
}

template <int MODE>
hipError_t launch_mode(const BatchArgs &a, const RglDev *r, bool indexed, int search, int compute_units, hipStream_t stream)
{
    if (!r) {                                                   // a batch with material ids: descriptors come from the material array
        size_t blocks = (a.n + kRglBlock - 1) / kRglBlock;
        const size_t cap = (size_t)compute_units * 8;
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
        const dim3 grid((unsigned)blocks), block(kRglBlock);
        const RglDev none{};
        if (indexed) hipLaunchKernelGGL((k_rgl<MODE, true, true>), grid, block, 0, stream, a, none);
        else hipLaunchKernelGGL((k_rgl<MODE, false, true>), grid, block, 0, stream, a, none);
        return hipGetLastError();
    }
    if (r->n_phi == 1 && r->n_theta > 1) return launch_masked<MODE, 5>(a, r, indexed, search, compute_units, stream);
    if (r->n_phi > 1 && r->n_theta > 1) return launch_masked<MODE, 15>(a, r, indexed, search, compute_units, stream);
    return launch_masked<MODE, 0>(a, r, indexed, search, compute_units, stream);
}

bool all_finite(const float *p, size_t n)
{
    for (size_t i = 0; i < n; ++i) if (!std::isfinite(p[i])) return false;
    return true;
}

bool ascending(const float *p, int n)
{
    for (int i = 1; i < n; ++i) if (!(p[i] > p[i - 1])) return false;
    return true;
}

// appends one function's tables to the image (offsets in floats; every table on a 128-B boundary), bracket-major — WarpDev in
// merl_rgl.hpp has the forms: the measured values as corner bricks, a distribution as one RECORD per cell (integrals left of the cell,
// normalised corner bricks, row totals) plus `margq`; all rounded to Float once from f64 sums, in the oracle's loop order.
// src: [n_phi][n_theta][n_ch][ny][nx].
WarpOffsets append_warp(std::vector<float> &blob, const float *src_all, int nx, int ny, int n_phi, int n_theta, int n_ch, bool distribution)
{
    const size_t per = (size_t)nx * ny, cells = (size_t)(nx - 1) * (size_t)(ny - 1), slices = (size_t)n_phi * (size_t)n_theta;
    const size_t per_cond = (size_t)ny * (size_t)(nx - 1), per_marg = (size_t)(ny - 1);
    const int tb = n_theta > 1 ? n_theta - 1 : 1, pb = n_phi > 1 ? n_phi - 1 : 1;
    const size_t P = n_phi > 1 ? 2 : 1, S = rgl_bracket_slices(n_phi, n_theta);
    const size_t stride = distribution ? 2 * P + S : (size_t)n_ch * S;      // float4s per cell
    size_t at = blob.size();
    const WarpOffsets off = plan_warp(at, nx, ny, n_phi, n_theta, n_ch, distribution);
    blob.resize(at, 0.0f);
    std::vector<double> cond(per_cond), marg(per_marg);
    std::vector<float> node(per), condf(distribution ? per_cond * slices : 0), margf(distribution ? per_marg * slices : 0);
    for (size_t s = 0; s < slices; ++s)
        for (int ch = 0; ch < n_ch; ++ch) {
            const float *src = src_all + per * (s * (size_t)n_ch + (size_t)ch);
            double norm = 1.0;
            if (distribution) {
                for (int y = 0; y < ny; ++y) {
                    double sum = 0.0;
                    for (int x = 0; x < nx - 1; ++x) {
                        sum += 0.5 * ((double)src[y * nx + x] + (double)src[y * nx + x + 1]);
                        cond[(size_t)y * (size_t)(nx - 1) + (size_t)x] = sum;
                    }
                }
                double sum = 0.0;
                for (int y = 0; y < ny - 1; ++y) {
                    sum += 0.5 * (cond[(size_t)y * (size_t)(nx - 1) + (size_t)(nx - 2)] + cond[(size_t)(y + 1) * (size_t)(nx - 1) + (size_t)(nx - 2)]);
                    marg[(size_t)y] = sum;
                }
                norm = sum > 0.0 ? 1.0 / sum : 1.0;
                for (size_t k = 0; k < per_cond; ++k) condf[s * per_cond + k] = (float)(cond[k] * norm);
                for (size_t k = 0; k < per_marg; ++k) margf[s * per_marg + k] = (float)(marg[k] * norm);
            }
            for (size_t k = 0; k < per; ++k) node[k] = (float)((double)src[k] * norm);
            // a slice's corner bricks go to every bracket the slice bounds, as the bracket's slice (dp, dt), phi fastest
            const int ip = (int)(s / (size_t)n_theta), it = (int)(s % (size_t)n_theta);
            for (int dp = 0; dp < (n_phi > 1 ? 2 : 1); ++dp)
                for (int dt = 0; dt < (n_theta > 1 ? 2 : 1); ++dt) {
                    const int ipb = ip - dp, itb = it - dt;
                    if (ipb < 0 || ipb >= pb || itb < 0 || itb >= tb) continue;
                    const size_t slot = n_phi > 1 ? (size_t)dp + 2 * (size_t)dt : (size_t)dt, first = ((size_t)ipb * tb + itb) * cells;
                    for (int y = 0; y < ny - 1; ++y)
                        for (int x = 0; x < nx - 1; ++x) {
                            const size_t cell = (size_t)y * (size_t)(nx - 1) + (size_t)x;
                            float *q = &blob[off.cells + ((first + cell) * stride + (distribution ? P : (size_t)ch * S) + slot) * 4];
                            q[0] = node[(size_t)y * nx + x]; q[1] = node[(size_t)y * nx + x + 1];
                            q[2] = node[(size_t)(y + 1) * nx + x]; q[3] = node[(size_t)(y + 1) * nx + x + 1];
                        }
                }
        }
    if (distribution) {
        for (int ipb = 0; ipb < pb; ++ipb)
            for (int itb = 0; itb < tb; ++itb) {
                const size_t first = ((size_t)ipb * tb + itb) * cells;
                // the record's integrals: per phi node of the bracket one float4 = (slice (ip, it): rows row, row + 1 | slice (ip, it + 1): the same)
                for (size_t k = 0; k < P; ++k) {
                    const size_t s0 = (size_t)(ipb + (int)k) * n_theta + itb, s1 = n_theta > 1 ? s0 + 1 : s0;
                    for (int y = 0; y < ny - 1; ++y)
                        for (int x = 0; x < nx - 1; ++x) {
                            const size_t cell = (size_t)y * (size_t)(nx - 1) + (size_t)x;
                            float *l = &blob[off.cells + ((first + cell) * stride + k) * 4], *t = &blob[off.cells + ((first + cell) * stride + P + S + k) * 4];
                            const size_t lo = (size_t)y * (size_t)(nx - 1), hi = lo + (size_t)(nx - 1), last = (size_t)(nx - 2);
                            if (x > 0) {                 // up to node x: what is left of the cell (column 0: zeros)
                                l[0] = condf[s0 * per_cond + lo + x - 1]; l[1] = condf[s0 * per_cond + hi + x - 1];
                                l[2] = condf[s1 * per_cond + lo + x - 1]; l[3] = condf[s1 * per_cond + hi + x - 1];
                            }
                            t[0] = condf[s0 * per_cond + lo + last]; t[1] = condf[s0 * per_cond + hi + last];
                            t[2] = condf[s1 * per_cond + lo + last]; t[3] = condf[s1 * per_cond + hi + last];
                        }
                }
                // the row headers: block 0 = the row's totals + the integrals up to the three pivot columns of [0, nx - 2]; blocks 1..4 = (unused
                // entry) + the three pivot columns of the quarter's range (rgl::pivots_of / quarter_of); one float4 per phi node each
                {
                    const int last = nx - 2;
                    for (int y = 0; y < ny - 1; ++y)
                        for (int blk = 0; blk < 5; ++blk) {
                            const rgl::Range rg = blk == 0 ? rgl::Range{ 0, last } : rgl::quarter_of(0, last, blk - 1);
                            const rgl::Pivots pv = rgl::pivots_of(rg.lo, rg.hi);
                            const int piv[4] = { blk == 0 ? last : pv.m1, pv.m1, pv.m2a, pv.m2b };
                            for (int g = 0; g < 4; ++g)
                                for (size_t k = 0; k < P; ++k) {
                                    const size_t s0 = (size_t)(ipb + (int)k) * n_theta + itb, s1 = n_theta > 1 ? s0 + 1 : s0;
                                    const size_t lo = (size_t)y * (size_t)(nx - 1) + (size_t)piv[g], hi = lo + (size_t)(nx - 1);
                                    float *h = &blob[off.rowh + ((((((size_t)ipb * tb + itb) * per_marg + (size_t)y) * 5 + (size_t)blk) * 4 + (size_t)g) * P + k) * 4];
                                    h[0] = condf[s0 * per_cond + lo]; h[1] = condf[s0 * per_cond + hi];
                                    h[2] = condf[s1 * per_cond + lo]; h[3] = condf[s1 * per_cond + hi];
                                }
                        }
                }
                const size_t dp = n_phi > 1 ? (size_t)n_theta : 0, dt = n_theta > 1 ? 1 : 0, s0 = (size_t)ipb * n_theta + itb;
                float *m = &blob[off.margq + ((size_t)ipb * tb + itb) * per_marg * 4];
                for (size_t y = 0; y < per_marg; ++y, m += 4) {
                    m[0] = margf[s0 * per_marg + y]; m[1] = margf[(s0 + dp) * per_marg + y];
                    m[2] = margf[(s0 + dt) * per_marg + y]; m[3] = margf[(s0 + dp + dt) * per_marg + y];
                }
            }
    }
    return off;
}

} // namespace

// 2 pi / (span of phi_i), rounded: which part of the azimuth an anisotropic file stores
int rgl_reduction(const RglFields &f)
{
    if (f.n_phi <= 2) return 1;
    const double span = (double)f.phi_i[f.n_phi - 1] - (double)f.phi_i[0];
    const double q = span > 0.0 ? 2.0 * kPi / span : 0.0;                 // (a denormal span gives inf: no cast of that)
    return (q >= 0.5 && q < 4.5) ? (int)std::floor(q + 0.5) : 0;
}

const char *rgl_check_fields(const RglFields &f)
{
    if (const char *why = rgl_check_shapes(f)) return why;
    if (!f.phi_i || !f.theta_i || !f.ndf || !f.sigma || !f.vndf || !f.luminance || !f.rgb || (f.n_wl > 0 && !f.wavelengths)) return "null field";
    const size_t slices = (size_t)f.n_phi * (size_t)f.n_theta, per = (size_t)f.res[0] * (size_t)f.res[1];
    if (!all_finite(f.phi_i, (size_t)f.n_phi) || !all_finite(f.theta_i, (size_t)f.n_theta) || !ascending(f.phi_i, f.n_phi) || !ascending(f.theta_i, f.n_theta))
        return "phi_i / theta_i must be finite and strictly ascending";
    if (!all_finite(f.ndf, (size_t)f.res_ndf[0] * f.res_ndf[1]) || !all_finite(f.sigma, (size_t)f.res_sigma[0] * f.res_sigma[1]) ||
        !all_finite(f.vndf, slices * per) || !all_finite(f.luminance, slices * per) || !all_finite(f.rgb, slices * per * (size_t)rgl_value_channels(f)))
        return "non-finite table value";
    if (f.n_wl > 0 && (!all_finite(f.wavelengths, (size_t)f.n_wl) || !ascending(f.wavelengths, f.n_wl))) return "wavelengths must be finite and strictly ascending";
    for (size_t k = 0; k < slices * per; ++k)
        if (f.vndf[k] < 0.0f || f.luminance[k] < 0.0f) return "vndf / luminance must be non-negative (they are densities)";
    // an anisotropic file covers the whole azimuth, or the half / quarter a sample with a point symmetry / two mirror planes needs
    if (f.n_phi > 2) {
        const int reduction = rgl_reduction(f);
        if (reduction != 1 && reduction != 2 && reduction != 4) return "anisotropic file: phi_i must span the whole azimuth, a half or a quarter of it";
    }
    return nullptr;
}

RglLayout rgl_build_image(const RglFields &f, std::vector<float> &blob)
{
    blob.clear();
    RglLayout l;
    l.phi = 0; l.theta = (size_t)f.n_phi;
    blob.insert(blob.end(), f.phi_i, f.phi_i + f.n_phi);
    blob.insert(blob.end(), f.theta_i, f.theta_i + f.n_theta);
    l.wavelengths = blob.size();
    if (f.n_wl > 0) blob.insert(blob.end(), f.wavelengths, f.wavelengths + f.n_wl);
    auto put = [&](int which, const float *src, const int res[2], int n_phi, int n_theta, int n_ch, bool distribution) {
        const WarpOffsets o = append_warp(blob, src, res[0], res[1], n_phi, n_theta, n_ch, distribution);
        l.cells[which] = o.cells; l.margq[which] = o.margq; l.rowh[which] = o.rowh;
    };
    put(0, f.ndf, f.res_ndf, 1, 1, 1, false);
    put(1, f.sigma, f.res_sigma, 1, 1, 1, false);
    put(2, f.vndf, f.res, f.n_phi, f.n_theta, 1, true);
    put(3, f.luminance, f.res, f.n_phi, f.n_theta, 1, true);
    put(4, f.rgb, f.res, f.n_phi, f.n_theta, rgl_value_channels(f), false);
    return l;
}

// base: 128-B aligned on the device (the tables then sit on cache-line boundaries); the host copy need not be
RglDev rgl_descriptor(const RglFields &f, const RglLayout &l, const float *base)
{
    auto at = [&](size_t off) { return (const float4 *)(base + off); };
    RglDev r;
    r.ndf_cells = at(l.cells[0]); r.sigma_cells = at(l.cells[1]); r.vndf_cells = at(l.cells[2]); r.lum_cells = at(l.cells[3]); r.rgb_cells = at(l.cells[4]);
    r.vndf_margq = at(l.margq[2]); r.lum_margq = at(l.margq[3]); r.vndf_rowh = at(l.rowh[2]); r.lum_rowh = at(l.rowh[3]);
    r.phi = base + l.phi; r.theta = base + l.theta;
    r.ndf_nx = f.res_ndf[0]; r.ndf_ny = f.res_ndf[1]; r.sigma_nx = f.res_sigma[0]; r.sigma_ny = f.res_sigma[1];
    r.nx = f.res[0]; r.ny = f.res[1]; r.n_phi = f.n_phi; r.n_theta = f.n_theta;
    r.n_values = rgl_value_channels(f);
    r.wavelengths = f.n_wl > 0 ? base + l.wavelengths : nullptr;
    r.n_wl = f.n_wl;
    r.isotropic = f.n_phi <= 2;
    r.jacobian = f.jacobian ? 1 : 0;
    r.reduction = rgl_reduction(f);
    return r;
}

namespace {
template <int MODE, int MASK>
hipError_t launch_spectral_masked(const BatchArgs &a, const RglDev &r, const float *wl, int W, int search, int compute_units, hipStream_t stream)
{
    if (search == 0 && a.n >= (size_t)1 << 15) {
        const size_t need = lds_bytes_of(r);
        if (need <= (size_t)lds_limit()) {
            constexpr int kThreads = rgl_lds_block(MODE);
            size_t blocks = (a.n + kThreads - 1) / kThreads;
            if (blocks > (size_t)compute_units) blocks = (size_t)compute_units;
            (void)hipFuncSetAttribute((const void *)k_rgl_spectral<MODE, true, MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
            hipLaunchKernelGGL((k_rgl_spectral<MODE, true, MASK>), dim3((unsigned)blocks), dim3(kThreads), need, stream, a, r, wl, W);
            return hipGetLastError();
        }
    }
    size_t blocks = (a.n + kRglBlock - 1) / kRglBlock;
    const size_t cap = (size_t)compute_units * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((k_rgl_spectral<MODE, false, MASK>), dim3((unsigned)blocks), dim3(kRglBlock), grid_bytes_of(r), stream, a, r, wl, W);
    return hipGetLastError();
}
// (the database's spectral files are isotropic: that shape has its own kernels, every other one tests the shape at run time)
template <int MODE>
hipError_t launch_spectral_mode(const BatchArgs &a, const RglDev &r, const float *wl, int W, int search, int compute_units, hipStream_t stream)
{
    if (r.n_phi == 1 && r.n_theta > 1) return launch_spectral_masked<MODE, 5>(a, r, wl, W, search, compute_units, stream);
    return launch_spectral_masked<MODE, 0>(a, r, wl, W, search, compute_units, stream);
}

} // namespace

hipError_t launch_rgl_spectral(int mode, const BatchArgs &a, const RglDev &r, const float *wl, int W, int search, int compute_units, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    switch (mode) {
        case 0: return launch_spectral_mode<0>(a, r, wl, W, search, compute_units, stream);
        case 1: return launch_spectral_mode<1>(a, r, wl, W, search, compute_units, stream);
        case 2: return launch_spectral_mode<2>(a, r, wl, W, search, compute_units, stream);
        case 3: return launch_spectral_mode<3>(a, r, wl, W, search, compute_units, stream);
        case 4: return launch_spectral_mode<4>(a, r, wl, W, search, compute_units, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_rgl(int mode, const BatchArgs &a, const RglDev *r, bool indexed, int search, int compute_units, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    // The fused unit of a file whose integrals do not fit a CU's LDS (an anisotropic file: its lookups blend four slices of a 45 MB
    // image, the launch waits on L1 fills): eval + pdf and sample() as TWO launches on the stream.  The fused kernel carries sample()'s
    // 220 VGPRs through its eval as well (2 waves per SIMD); apart, eval + pdf runs at 3 waves per SIMD and sample() with its marginal
    // rows in LDS — 16M units: 3.84 ms fused, 3.34 ms apart (profiles/r04_rgl_rates.json); the 12 B per unit of wi read twice do not
    // show.  Same functions, same bits (the separate entry points are bit-compared with the fused one).
    if (mode == 3 && r && search == 0 && a.n >= (size_t)1 << 15 && lds_bytes_of(*r) > (size_t)lds_limit()) {
        const hipError_t e = launch_mode<4>(a, r, indexed, search, compute_units, stream);
        return e != hipSuccess ? e : launch_mode<2>(a, r, indexed, search, compute_units, stream);
    }
    switch (mode) {
        case 0: return launch_mode<0>(a, r, indexed, search, compute_units, stream);
        case 1: return launch_mode<1>(a, r, indexed, search, compute_units, stream);
        case 2: return launch_mode<2>(a, r, indexed, search, compute_units, stream);
        case 3: return launch_mode<3>(a, r, indexed, search, compute_units, stream);
        case 4: return launch_mode<4>(a, r, indexed, search, compute_units, stream);
    }
    return hipErrorInvalidValue;
}

} // namespace mrl
