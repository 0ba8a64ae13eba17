// merl_rgl.hpp — per-unit math of the adaptive-parameterisation measured BSDF (the RGL material database's *.bsdf files;
// Dupuy & Jakob 2018; upstream Mitsuba 3's stock `measured` plugin).  SURVEY.md §8f item 3 ("next" row).
//
// PARITY UNPINNED: no RGL file, no reader and no plugin source exists in the reference snapshot; the model is restated from
// its published description (oracle/rgl_oracle.c is the independent CPU restatement the GPU tests compare against).
//
// The model.  Five piecewise-bilinear functions over the unit square, three of them with interpolated parameters
// (phi_i, theta_i) of the incident direction:
//     ndf, sigma                  plain 2-D tables (microfacet normal density, projected area)
//     vndf, luminance             normalised 2-D distributions per (phi_i, theta_i) node, with running integrals for
//                                 sampling (conditional: along x per node row, marginal: over rows)
//     rgb                         the measured values in the warped domain, per (phi_i, theta_i, channel)
// eval(wi, wo):  m = (wi + wo) normalised;  u_m = (sqrt(2 theta_m / pi), (phi_m [- phi_i]) / 2pi + 1/2);
//                s = vndf.invert(u_m);  f cos = rgb(s) * ndf(u_m) / (4 sigma(u_wi))
// pdf(wi, wo):   vndf.pdf(u_m) * luminance(s) / (max(2 pi^2 u_m.x sin theta_m, 1e-6) * 4 (wi . m))
// sample(wi, u): s = luminance.sample(u), u_m = vndf.sample(s), wo = reflect(wi, m(u_m)); what is reported is eval / pdf
//                AT the Float direction that is returned (so pdf(wi, sample.wo) == sample.pdf and weight == eval / pdf).
// Math in f64 on Float tables with the table path's own building blocks (v_rcp / v_rsq seeds + one Newton step, the
// pi/8-rotated atan polynomial, Taylor sin / cos: relative error ~1e-15 each) in place of ocml's correctly rounded
// division, sqrt, asin, atan2, sin and cos.  One lane per unit.
//
// What bounds the kernel and what this file does about it (profiles/r04_rgl_pmc.json, r04_rgl_l2.json,
// r04_gather_quad.json).  Two things, found in this order:
//   (1) ROUND TRIPS.  A table read that sits next to its use inside a conditional block (`if (slice exists) { q = table[..];
//       v = fma(w, q, v); }`) cannot be hoisted, so the wave waits for every read on its own — 25 serial round trips per
//       anisotropic eval, the CU's L1 idling on one outstanding line per wave.  Hence every lookup here comes in two steps,
//       READS (fetch_raw, Search::*_raw: loads only, issued back to back) and SUMS (blend4, *_blend), and the callers issue
//       everything a stage needs before they sum anything: an inverse warp its whole cell, eval its three channels and the
//       luminance cell.  The kernels compiled for one bracket shape (merl_rgl.hip, MASK) make the reads straight-line code.
//   (2) LINE FILLS.  With the reads in flight together a lane's reads that fall in one 128-B line merge into one L1 fill, and
//       the cost of a unit is its number of distinct lines (~4-5 CU-cycles each from L2).  Hence the image is stored per
//       parameter BRACKET — the 2 / 4 slices a lookup blends side by side — and a distribution's cell is ONE RECORD (WarpDev):
//       the integrals left of the cell, the corner values per slice, the row totals: an inverse warp = one line.
//   * The running integrals are read through a policy (`Search`): from the records in memory (SearchMem: host images, batches
//     with material ids, files too large for a CU's LDS) or from slice-major copies in LDS (merl_rgl.hip: every step of
//     sample()'s two binary searches a ds_read); the parameter grids through `Grids` (memory, or LDS in every
//     single-material kernel).
//   * sample() hands the cells its two searches ended in to the eval / pdf it reports at the returned direction
//     (`Found`): the inverse warp of a point the forward warp has just produced lands in the same cell (but for one unit in
//     ~10^5, which re-reads), so nothing is read twice;
//   * what depends on the incident direction alone (angles, parameter bracket, projected area) is formed once per unit.
// Every blend is spelt with explicit FMAs and contraction is off: which multiply the compiler would fuse depends on the
// inlining context, and the entry points (separate, fused, queue, batch with ids, LDS or memory) must agree bit for bit.
#pragma once
#include "merl_device.hpp"
#include "merl_table_fast.hpp"     // rcp_nr / div_fast / sqrt_fast / rsqrt_pos / atan2_q1: the table path's f64 building blocks
#include "merl_ggx_fast.hpp"       // sincos_2pi

namespace mrl {

// One piecewise-bilinear function.  A slice is (ny - 1) x (nx - 1) cells, x fastest; everything is stored per parameter BRACKET
// (tb = max(n_theta - 1, 1), pb = max(n_phi - 1, 1); a bracket has S = 1 / 2 / 4 slices, phi fastest, and P = 1 / 2 phi nodes):
//   measured values:  cells [pb][tb][cell][n_ch][S]   the cell's four corner values (v00, v10, v01, v11) per slice
//   distributions:    cells [pb][tb][cell] RECORDS of 2 P + S float4 — everything an inverse warp or a search step needs about the cell
//                     in one 64-B (isotropic) or 128-B (anisotropic) piece of ONE cache line:
//                       [0, P)          running integrals along x LEFT of the cell (up to node col) of node rows (row, row + 1): .xy of
//                                       slice (ip, it), .zw of slice (ip, it + 1); one float4 per phi node; zeros in column 0
//                       [P, P + S)      the corner values per slice, normalised
//                       [P + S, 2P + S) the totals of node rows (row, row + 1), same form as the first part
//                     margq [pb][tb][ny - 1]  marginal cdf after the cell row, of slices (ip, it) (ip+1, it) (ip, it+1) (ip+1, it+1)
//                     rowh  [pb][tb][ny - 1]  ROW HEADERS of 5 blocks x 4 P float4: the row's totals with the integrals up to the three columns
//                                             the column search's first two halvings test (pivots_of), then per quarter of the row the three
//                                             columns of the next two halvings — a forward warp reads the totals and decides four of its
//                                             log2(nx) halvings from TWO lines
struct WarpDev {
    const float4 *cells;
    const float4 *margq, *rowh; // distributions only
    int nx, ny, n_phi, n_theta;
    int normalized;             // a distribution: values are densities over the unit square
    int stride, first;          // float4s per cell; where the corner values (of channel 0) start in a cell's piece
    MRL_HD int phi_nodes() const { return n_phi > 1 ? 2 : 1; }
    MRL_HD int slices() const { return (n_phi > 1 ? 2 : 1) * (n_theta > 1 ? 2 : 1); }
};

// The material's descriptor, compact: it travels to the kernels by value in SGPRs.  vndf, luminance and the measured values share the
// parameter grids and the resolution; the five functions are handed out as WarpDev views.
struct RglDev {
    const float4 *ndf_cells, *sigma_cells, *vndf_cells, *lum_cells, *rgb_cells;
    const float4 *vndf_margq, *lum_margq, *vndf_rowh, *lum_rowh;
    const float *phi, *theta;   // ascending parameter grids of vndf / luminance / rgb
    // spectral files ("spectra" + "wavelengths" instead of "rgb"): `rgb_cells` then holds the spectra, one channel per wavelength
    // node (n_values = n_wl), and a value is interpolated linearly between the nodes around the wavelength asked for
    const float *wavelengths;   // [n_wl], ascending; nullptr for an RGB file
    int ndf_nx, ndf_ny, sigma_nx, sigma_ny, nx, ny, n_phi, n_theta;
    int n_values;           // channels of the measured values: 3 (RGB) or n_wl
    int isotropic;          // n_phi <= 2: phi_m is measured relative to phi_i
    int jacobian;           // the file's flag: multiply the spectrum by ndf / (4 sigma)
    int reduction;          // anisotropic: 2 pi / (span of phi_i) — 1 the whole azimuth, 2 [-pi, 0] (point symmetry), 4 [-pi, -pi/2] (+ two mirror planes)
    int n_wl;               // 0: an RGB file
    // the same descriptor over a copy of the image at another address (the host table: merl_materials.hip)
    void rebase(const char *from, const char *to)
    {
        const float4 **q[9] = { &ndf_cells, &sigma_cells, &vndf_cells, &lum_cells, &rgb_cells, &vndf_margq, &lum_margq, &vndf_rowh, &lum_rowh };
        for (const float4 **x : q) *x = (const float4 *)(to + ((const char *)*x - from));
        const float **g[3] = { &phi, &theta, &wavelengths };
        for (const float **x : g) if (*x) *x = (const float *)(to + ((const char *)*x - from));
    }
    MRL_HD int phi_nodes() const { return n_phi > 1 ? 2 : 1; }
    MRL_HD int slices() const { return (n_phi > 1 ? 2 : 1) * (n_theta > 1 ? 2 : 1); }
    MRL_HD WarpDev ndf() const { return { ndf_cells, nullptr, nullptr, ndf_nx, ndf_ny, 1, 1, 0, 1, 0 }; }
    MRL_HD WarpDev sigma() const { return { sigma_cells, nullptr, nullptr, sigma_nx, sigma_ny, 1, 1, 0, 1, 0 }; }
    MRL_HD WarpDev vndf() const { return { vndf_cells, vndf_margq, vndf_rowh, nx, ny, n_phi, n_theta, 1, 2 * phi_nodes() + slices(), phi_nodes() }; }
    MRL_HD WarpDev luminance() const { return { lum_cells, lum_margq, lum_rowh, nx, ny, n_phi, n_theta, 1, 2 * phi_nodes() + slices(), phi_nodes() }; }
    MRL_HD WarpDev rgb() const { return { rgb_cells, nullptr, nullptr, nx, ny, n_phi, n_theta, 0, n_values * slices(), 0 }; }
};

namespace rgl {

#ifndef MRL_RGL_SAMPLE_BATCH
#define MRL_RGL_SAMPLE_BATCH true
#endif

// the four parameter slices around (phi_i, theta_i) and their weights, phi fastest (the order the oracle sums in); `mask`
// says which entries exist (a grid of one node has no upper neighbour) — uniform over a launch (a constant in the kernels compiled
// for a bracket shape), so the tests on it are scalar branches or none and the arrays stay in registers.
// Offsets instead of indices (formed once per unit: a 32-bit multiply costs what an f64 FMA costs): cell0 = the bracket's first cell
// in the bracket-major tables, quad = its first row in margq; soff[k] / roff[k] = slice k x cells / cell rows of a slice (the slice-major
// copies of the running integrals in LDS).
struct Slices { unsigned soff[4], roff[4]; double w[4]; int mask; unsigned cell0, quad; };

// largest i in [0, n - 2] with node(i) <= p, and p's position in that bracket; node(k): the ascending grid's k-th value
template <class Node>
MRL_HD void bracket_by(const Node &node, int n, double p, int &i, double &t)
{
#pragma clang fp contract(off)
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((double)node(mid) <= p) lo = mid; else hi = mid; }
    i = lo;
    const double p0 = node(lo), p1 = node(lo + 1);
    t = fast::div_fast(p - p0, p1 - p0);
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
}
MRL_HD void bracket(const float *grid, int n, double p, int &i, double &t)
{
    bracket_by([grid](int k) { return grid[k]; }, n, p, i, t);
}

// Where the two parameter grids are read: from the image (host code, batches with material ids) or from the copy a single-material
// kernel keeps in LDS (merl_rgl.hip: GridLds) — the bracket search is 3 - 4 dependent reads per grid and unit, a third of an
// anisotropic eval while each was a round trip through the vector memory pipe.
struct GridMem {
    const float *phi, *theta, *wavelengths;     // (wavelengths: spectral files)
    MRL_HD float phi_at(int k) const { return phi[k]; }
    MRL_HD float theta_at(int k) const { return theta[k]; }
    MRL_HD float wavelength_at(int k) const { return wavelengths[k]; }
};

template <class Grids>
MRL_HD Slices find_slices(const WarpDev &w, const Grids &g, double phi_i, double theta_i)
{
#pragma clang fp contract(off)
    int ip = 0, it = 0;
    double tp = 0.0, tt = 0.0;
    if (w.n_phi > 1) bracket_by([&g](int k) { return g.phi_at(k); }, w.n_phi, phi_i, ip, tp);
    if (w.n_theta > 1) bracket_by([&g](int k) { return g.theta_at(k); }, w.n_theta, theta_i, it, tt);
    const int ip1 = w.n_phi > 1 ? ip + 1 : ip, it1 = w.n_theta > 1 ? it + 1 : it;
    const int tb = w.n_theta > 1 ? w.n_theta - 1 : 1;
    const unsigned per_c = (unsigned)((w.nx - 1) * (w.ny - 1)), per_r = (unsigned)(w.ny - 1);
    const unsigned sl[4] = { (unsigned)(ip * w.n_theta + it), (unsigned)(ip1 * w.n_theta + it), (unsigned)(ip * w.n_theta + it1), (unsigned)(ip1 * w.n_theta + it1) };
    Slices out;
    for (int k = 0; k < 4; ++k) { out.soff[k] = sl[k] * per_c; out.roff[k] = sl[k] * per_r; }
    out.w[0] = (1.0 - tp) * (1.0 - tt); out.w[1] = tp * (1.0 - tt); out.w[2] = (1.0 - tp) * tt; out.w[3] = tp * tt;
    out.mask = 1 | (w.n_phi > 1 ? 2 : 0) | (w.n_theta > 1 ? 4 : 0) | (w.n_phi > 1 && w.n_theta > 1 ? 8 : 0);
    out.cell0 = (unsigned)(ip * tb + it) * per_c;
    out.quad = (unsigned)(ip * tb + it) * per_r;
    return out;
}

MRL_HD Slices single_slice()
{
    Slices out;
    for (int k = 0; k < 4; ++k) { out.soff[k] = out.roff[k] = 0u; out.w[k] = 0.0; }
    out.w[0] = 1.0; out.mask = 1;
    out.cell0 = out.quad = 0u;
    return out;
}

// weighted sums over the parameter slices, component by component in slice order (what the oracle's scalar loop does; the
// first term is a product, every later one an explicit FMA); 32-bit offsets: a function's tables hold at most 2^28 values;
// index: the cell inside a slice (ndf / sigma: single_slice(), whose offsets are zero); the cells are stored per (phi, theta)
// BRACKET — pair[0] is the bracket's first cell —, [cell][channel][slice of the bracket]
struct D4 { double x, y, z, w; };
struct D2 { double x, y; };
// A lookup in two steps, so that the loads of everything a stage needs are in flight together: the compiler waits for a load where
// its value is first used, and a load inside a conditional block next to its use is a round trip of its own — 25 serial round
// trips per anisotropic eval was what bounded the kernel, not the addresser and not the lines fetched.
//   fetch_raw: the corner vectors of one cell (and channel), one per slice of the bracket — loads only;
//   blend4:    their weighted sum.
struct Raw4 { float4 q0, q1, q2, q3; };      // q1: the phi neighbour, q2: the theta neighbour, q3: both
MRL_HD unsigned bracket_slices(const Slices &s) { return (unsigned)((s.mask & 1) + ((s.mask >> 1) & 1) + ((s.mask >> 2) & 1) + ((s.mask >> 3) & 1)); }
MRL_HD Raw4 fetch_raw(const Slices &s, const WarpDev &w, int index, int channel = 0)
{
    // the bracket's slices lie side by side (16 / 32 / 64 B per cell and channel), phi fastest; a distribution's behind the cell's integrals
    const float4 *p = w.cells + (s.cell0 + (unsigned)index) * (unsigned)w.stride + (unsigned)w.first + (unsigned)channel * bracket_slices(s);
    Raw4 r;
    // absent slices: zeros (never summed).  Defined values on purpose: left undefined, the compiler merges each conditional read with the
    // conditional sum that uses it — the load next to its use again, one round trip per slice (measured: 1.49 -> 1.97 ms)
    r.q1 = r.q2 = r.q3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    r.q0 = p[0];
    if (s.mask & 2) r.q1 = p[1];
    if (s.mask & 4) r.q2 = p[(s.mask & 2) ? 2 : 1];
    if (s.mask & 8) r.q3 = p[3];
    return r;
}
MRL_HD D4 blend4(const Slices &s, const Raw4 &r)
{
#pragma clang fp contract(off)
    D4 v = { s.w[0] * (double)r.q0.x, s.w[0] * (double)r.q0.y, s.w[0] * (double)r.q0.z, s.w[0] * (double)r.q0.w };
    auto add = [&](double w, const float4 &q) {
        v.x = __builtin_fma(w, (double)q.x, v.x); v.y = __builtin_fma(w, (double)q.y, v.y);
        v.z = __builtin_fma(w, (double)q.z, v.z); v.w = __builtin_fma(w, (double)q.w, v.w);
    };
    if (s.mask & 2) add(s.w[1], r.q1);
    if (s.mask & 4) add(s.w[2], r.q2);
    if (s.mask & 8) add(s.w[3], r.q3);
    return v;
}
MRL_HD D4 fetch4(const Slices &s, const WarpDev &w, int index, int channel = 0)
{
    return blend4(s, fetch_raw(s, w, index, channel));
}

// the blends of the bracket vectors, in slice order (the same sums, in the same order, as a slice-by-slice read):
// a / b: a record's integrals at the bracket's phi node ip / ip + 1 (b unused without an upper phi neighbour)
MRL_HD D2 blend_pairs(const Slices &s, const float4 &a, const float4 &b)
{
#pragma clang fp contract(off)
    D2 v = { s.w[0] * (double)a.x, s.w[0] * (double)a.y };
    if (s.mask & 2) { v.x = __builtin_fma(s.w[1], (double)b.x, v.x); v.y = __builtin_fma(s.w[1], (double)b.y, v.y); }
    if (s.mask & 4) { v.x = __builtin_fma(s.w[2], (double)a.z, v.x); v.y = __builtin_fma(s.w[2], (double)a.w, v.y); }
    if (s.mask & 8) { v.x = __builtin_fma(s.w[3], (double)b.z, v.x); v.y = __builtin_fma(s.w[3], (double)b.w, v.y); }
    return v;
}
MRL_HD double blend_quad(const Slices &s, const float4 &q)
{
#pragma clang fp contract(off)
    double v = s.w[0] * (double)q.x;
    if (s.mask & 2) v = __builtin_fma(s.w[1], (double)q.y, v);
    if (s.mask & 4) v = __builtin_fma(s.w[2], (double)q.z, v);
    if (s.mask & 8) v = __builtin_fma(s.w[3], (double)q.w, v);
    return v;
}

// A distribution's running integrals read from memory — from its cell records (the integrals LEFT of a cell, the totals of its two
// node rows) and from margq (the marginal after a cell row) — in two steps each: the loads, then the sums (see fetch_raw).
struct SearchMem {
    const float4 *cells, *margq, *rowh;
    unsigned stride, totals_at, per_row, nodes; // float4s per record; where a record's totals start; cells per row; phi nodes of a bracket
    MRL_HD explicit SearchMem(const WarpDev &w)
        : cells(w.cells), margq(w.margq), rowh(w.rowh), stride((unsigned)w.stride), totals_at((unsigned)(w.phi_nodes() + w.slices())),
          per_row((unsigned)(w.nx - 1)), nodes((unsigned)w.phi_nodes()) {}
    struct PairRaw { float4 a, b; };
    typedef float4 MargRaw;
    struct HeadRaw { PairRaw total, p1, p2a, p2b; };
    // a cell row's header: its totals and the integrals at the three pivot columns, one line
    MRL_HD HeadRaw head_raw(const Slices &s, int row) const
    {
        const float4 *p = rowh + (s.quad + (unsigned)row) * (20u * nodes);
        HeadRaw h;
        h.total = pair_at(s, p); h.p1 = pair_at(s, p + nodes); h.p2a = pair_at(s, p + 2u * nodes); h.p2b = pair_at(s, p + 3u * nodes);
        return h;
    }
    // ... and the block of the quarter q = 2 (first halving went right) + (second went right): the pivots of its range [lo, hi], one line
    MRL_HD HeadRaw quarter_raw(const Slices &s, int row, int q, int, int) const
    {
        const float4 *p = rowh + (s.quad + (unsigned)row) * (20u * nodes) + (unsigned)(1 + q) * (4u * nodes);
        HeadRaw h;
        h.total = pair_at(s, p); h.p1 = pair_at(s, p + nodes); h.p2a = pair_at(s, p + 2u * nodes); h.p2b = pair_at(s, p + 3u * nodes);
        return h;
    }
    MRL_HD PairRaw pair_at(const Slices &s, const float4 *p) const
    {
        PairRaw r;
        r.b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        r.a = p[0];
        if (s.mask & 2) r.b = p[1];
        return r;
    }
    // node rows (row, row + 1) of cell `cell` = row (nx - 1) + col: the conditional running integrals up to node col
    MRL_HD PairRaw left_raw(const Slices &s, int cell) const { return pair_at(s, cells + (s.cell0 + (unsigned)cell) * stride); }
    // ... and over the whole rows
    MRL_HD PairRaw total_raw(const Slices &s, int row) const { return pair_at(s, cells + (s.cell0 + (unsigned)row * per_row) * stride + totals_at); }
    MRL_HD D2 pair_blend(const Slices &s, const PairRaw &r) const { return blend_pairs(s, r.a, r.b); }
    // the marginal cdf after cell row `row`
    MRL_HD MargRaw marg_raw(const Slices &s, int row) const { return margq[s.quad + (unsigned)row]; }
    MRL_HD double marg_blend(const Slices &s, const MargRaw &q) const { return blend_quad(s, q); }
    MRL_HD D2 left(const Slices &s, int cell) const { return pair_blend(s, left_raw(s, cell)); }
    MRL_HD D2 total(const Slices &s, int row) const { return pair_blend(s, total_raw(s, row)); }
    MRL_HD double marg(const Slices &s, int row) const { return marg_blend(s, marg_raw(s, row)); }
};

MRL_HD int clamp_cell(double p, int last)
{
    int i = trunc_i(p);
    return i < 0 ? 0 : (i > last ? last : i);
}

// (1 - t) a + t b
MRL_HD double lerp(double t, double a, double b)
{
#pragma clang fp contract(off)
    return __builtin_fma(t, b, (1.0 - t) * a);
}

// What a search (or an inverse warp) knows about the cell it ended in: the blended corner values, the conditional
// integrals left of the cell (node rows row / row + 1; zero in column 0), the marginal below the row (zero in row 0) and
// the totals of the two node rows.  sample() passes it on to the eval / pdf at the direction it returns.
struct Found { int row, col; D4 q; D2 left; double before, r0, r1; };

MRL_HD double bilinear(const D4 &q, double fx, double fy)
{
    return lerp(fy, lerp(fx, q.x, q.y), lerp(fx, q.z, q.w));
}

// the cell of a table a position falls in, and the position inside the cell
struct Cell { int ox, oy, index; double fx, fy; };
MRL_HD Cell locate(const WarpDev &w, double x_in, double y_in)
{
#pragma clang fp contract(off)
    const double px = x_in * (double)(w.nx - 1), py = y_in * (double)(w.ny - 1);
    Cell c;
    c.ox = clamp_cell(px, w.nx - 2); c.oy = clamp_cell(py, w.ny - 2);
    c.fx = px - (double)c.ox; c.fy = py - (double)c.oy;
    c.index = c.oy * (w.nx - 1) + c.ox;
    return c;
}
// the value at the position from the cell's blended corners
MRL_HD double cell_value(const WarpDev &w, const Cell &c, const D4 &q)
{
#pragma clang fp contract(off)
    const double v = bilinear(q, c.fx, c.fy);
    return w.normalized ? v * (double)(w.nx - 1) * (double)(w.ny - 1) : v;
}

// a plain lookup; `known` (may be null): a cell of this table whose corner values are already held
MRL_HD double warp_eval(const WarpDev &w, const Slices &s, double x_in, double y_in, int channel = 0, const Found *known = nullptr)
{
    const Cell c = locate(w, x_in, y_in);
    D4 q;
    if (known && known->row == c.oy && known->col == c.ox) q = known->q;
    else q = fetch4(s, w, c.index, channel);
    return cell_value(w, c, q);
}

MRL_HD double safe_sqrt(double x) { return x > 0.0 ? fast::sqrt_fast(x) : 0.0; }
MRL_HD double clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }

// position in [0, 1] at which a density running linearly from c0 to c1 has accumulated the mass u
MRL_HD double invert_linear(double c0, double c1, double u)
{
#pragma clang fp contract(off)
    const bool is_const = fabs(c0 - c1) < 1e-4 * (c0 + c1);
    const double num = is_const ? 2.0 * u : c0 - safe_sqrt(__builtin_fma(-2.0 * u, c0 - c1, c0 * c0));
    const double den = is_const ? c0 + c1 : c0 - c1;
    return den != 0.0 ? fast::div_fast(num, den) : 0.0;
}

// The columns a binary search of [lo, hi] tests first — (lo + hi) / 2 — and next, in the half it moved into (where that half still has
// more than one column; else the first pivot again, unused): what a header block holds the integrals of.
struct Pivots { int m1, m2a, m2b; };
MRL_HD Pivots pivots_of(int lo, int hi)
{
    const int m1 = (lo + hi) >> 1;
    return { m1, m1 > lo ? (lo + m1) >> 1 : m1, m1 + 1 < hi ? (m1 + 1 + hi) >> 1 : m1 };
}
// the range two halvings of [lo, hi] end in when they went (q & 2: the first, q & 1: the second) right; a halving of a single column does not happen
struct Range { int lo, hi; };
MRL_HD Range quarter_of(int lo, int hi, int q)
{
    for (int level = 0; level < 2; ++level)
        if (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (q & (2 >> level)) lo = mid + 1; else hi = mid;
        }
    return { lo, hi };
}

// uniform sample -> position; returns the density there and what it found on the way
template <class Search>
MRL_HD double warp_sample(const WarpDev &w, const Search &t, const Slices &s, double ux, double uy, double &x_out, double &y_out, Found &f)
{
#pragma clang fp contract(off)
    const int nx = w.nx, ny = w.ny;
    ux = clamp01(ux); uy = clamp01(uy);
    int lo = 0, hi = ny - 2;
    double before = 0.0;                                     // the marginal cdf below row lo: every step that raises lo has just read it
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const double c = t.marg(s, mid);
        if (c < uy) { lo = mid + 1; before = c; } else hi = mid;
    }
    const int row = lo;
    uy -= before;
    // the row's header: the totals of node rows (row, row + 1) and the integrals the first two halvings of the column search test; then
    // the header block of the quarter those end in: the next two — the very values the search would read cell by cell (so the same
    // decisions), two lines and two round trips for four halvings
    const auto hr = t.head_raw(s, row);
    const D2 tot = t.pair_blend(s, hr.total);
    const double r0 = tot.x, r1 = tot.y;
    const double y = clamp01(invert_linear(r0, r1, uy));
    ux *= lerp(y, r0, r1);
    lo = 0; hi = nx - 2;
    D2 left = { 0.0, 0.0 };                                  // the conditional integrals left of column lo, likewise
    // two halvings from a block's three pivots; returns 2 (the first went right) + 1 (the second did)
    auto two_halvings = [&](const decltype(hr) &h) {
        int q = 0;
        if (lo < hi) {
            const D2 p = t.pair_blend(s, h.p1);              // at column (lo + hi) >> 1
            if (lerp(y, p.x, p.y) < ux) { lo = ((lo + hi) >> 1) + 1; left = p; q = 2; } else hi = (lo + hi) >> 1;
        }
        if (lo < hi) {
            auto pr = h.p2a;
            if (q) pr = h.p2b;
            const D2 p = t.pair_blend(s, pr);                // at the middle of the half it moved into
            if (lerp(y, p.x, p.y) < ux) { lo = ((lo + hi) >> 1) + 1; left = p; q |= 1; } else hi = (lo + hi) >> 1;
        }
        return q;
    };
    const int quarter = two_halvings(hr);
    if (lo < hi) {
        const auto qr = t.quarter_raw(s, row, quarter, lo, hi);
        (void)two_halvings(qr);
    }
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const D2 p = t.left(s, row * (nx - 1) + mid + 1);   // the integrals up to node mid + 1: left of cell mid + 1
        if (lerp(y, p.x, p.y) < ux) { lo = mid + 1; left = p; } else hi = mid;
    }
    const int col = lo;
    ux -= lerp(y, left.x, left.y);
    const D4 q = fetch4(s, w, row * (nx - 1) + col);
    const double c0 = lerp(y, q.x, q.z), c1 = lerp(y, q.y, q.w);
    const double x = clamp01(invert_linear(c0, c1, ux));
    x_out = fast::div_fast((double)col + x, (double)(nx - 1));
    y_out = fast::div_fast((double)row + y, (double)(ny - 1));
    f.row = row; f.col = col; f.q = q; f.left = left; f.before = before; f.r0 = r0; f.r1 = r1;
    return lerp(x, c0, c1) * (double)(nx - 1) * (double)(ny - 1);
}

// position -> the uniform sample that maps to it; returns the density at the position.  `known` (may be null): the cell a
// forward warp of this table ended in — when the position lies in it nothing is read.
template <class Search>
MRL_HD double warp_invert(const WarpDev &w, const Search &t, const Slices &s, double x_in, double y_in, double &ux_out, double &uy_out,
                          const Found *known = nullptr)
{
#pragma clang fp contract(off)
    const int nx = w.nx, ny = w.ny;
    const double px = x_in * (double)(nx - 1), py = y_in * (double)(ny - 1);
    const int col = clamp_cell(px, nx - 2), row = clamp_cell(py, ny - 2);
    const double x = px - (double)col, y = py - (double)row;
    D4 q;
    D2 left = { 0.0, 0.0 };
    double before = 0.0, r0, r1;
    if (known && known->row == row && known->col == col) {
        q = known->q; left = known->left; before = known->before; r0 = known->r0; r1 = known->r1;
    } else {
        // everything the cell needs is read before anything is summed, and from memory it is ONE record (column 0's holds zeros on its
        // left) plus the marginal below the row (read at a clamped index and dropped in row 0)
        const Raw4 qr = fetch_raw(s, w, row * (nx - 1) + col);
        const auto lr = t.left_raw(s, row * (nx - 1) + col);
        const auto tr = t.total_raw(s, row);
        const auto br = t.marg_raw(s, row > 0 ? row - 1 : 0);
        q = blend4(s, qr);
        left = t.pair_blend(s, lr);
        const D2 tot = t.pair_blend(s, tr);
        r0 = tot.x; r1 = tot.y;
        const double bf = t.marg_blend(s, br);
        if (row > 0) before = bf;
    }
    const double c0 = lerp(y, q.x, q.z), c1 = lerp(y, q.y, q.w);
    const double pdf = lerp(x, c0, c1) * (double)(nx - 1) * (double)(ny - 1);
    const double sx = __builtin_fma(x, __builtin_fma(0.5 * x, c1 - c0, c0), lerp(y, left.x, left.y));
    const double tot = lerp(y, r0, r1);
    ux_out = tot > 0.0 ? fast::div_fast(sx, tot) : 0.0;
    uy_out = __builtin_fma(y, __builtin_fma(0.5 * y, r1 - r0, r0), before);
    return pdf;
}

// polar angle of a unit direction of the upper hemisphere (d.z > 0): atan2(|d_xy|, d_z), well conditioned at the pole
MRL_HD double elevation(const Vec3d &d)
{
#pragma clang fp contract(off)
    return fast::atan2_q1<false>(fast::sqrt_fast(__builtin_fma(d.x, d.x, d.y * d.y)), d.z);
}
// atan2(y, x) over the full circle, IEEE signs (atan2(+-0, -x) = +-pi, atan2(+-0, +-0) = +-0 or +-pi)
MRL_HD double azimuth(double y, double x)
{
#pragma clang fp contract(off)
    const bool neg_x = __builtin_signbit(x), neg_y = __builtin_signbit(y);
    const double t = (x == 0.0 && y == 0.0) ? 0.0 : fast::atan2_q1<false>(__builtin_fabs(y), __builtin_fabs(x));
    const double r = neg_x ? kPi - t : t;
    return neg_y ? -r : r;
}
MRL_HD double theta2u(double t) { return fast::sqrt_fast(t * (2.0 / kPi)); }
MRL_HD double phi2u(double p) { return (p + kPi) * (0.5 / kPi); }

// (contraction off: wi * rs + wo * rs must cancel exactly for a mirror pair — fused into fma(wi, rs, round(wo * rs)) it leaves
// the product's rounding error, and the half vector of the specular direction gets an arbitrary azimuth)
MRL_HD bool unit3(Vec3d &v)
{
#pragma clang fp contract(off)
    const double n2 = v.x * v.x + v.y * v.y + v.z * v.z;
    if (!(n2 > 0.0)) return false;
    // two Newton steps: for a near-mirror pair the half vector's transverse part is the DIFFERENCE of the two
    // normalisations (|m_xy| ~ 1e-10 for inputs that differ by a few Float ulps), so the reciprocal norm must be good to
    // an f64 ulp or the half vector's azimuth is noise
    double y = fast::rsqrt_pos(n2);
    const double e = __builtin_fma(-(n2 * y), y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
    v.x *= y; v.y *= y; v.z *= y;
    return true;
}

// Symmetry-reduced files: both directions of a pair go into the stored part of the azimuth with the signs of wi — x and y negated
// together when wi.y is not negative (2), x when wi.x and y when wi.y is not negative (4); by the sign BIT (+0 is positive).
// Returns the two factors so that sample() can take the direction it draws back out.
MRL_HD void reduce_signs(int reduction, float wix, float wiy, float &sx, float &sy)
{
    sx = sy = 1.0f;
    if (reduction < 2) return;
    sy = __builtin_signbit(wiy) ? 1.0f : -1.0f;
    sx = reduction == 4 ? (__builtin_signbit(wix) ? 1.0f : -1.0f) : sy;
}

// What eval / pdf / sample share about the incident direction: wi in the stored part of the azimuth, normalised; its angles; the
// parameter slices around them (vndf, luminance and rgb share the parameter grids); four times the projected area (the
// jacobian's denominator: a function of wi alone)
struct Incident { Vec3d wi; float fx, fy; double theta_i, phi_i, sigma4; Slices sv; };

template <bool WANT_SIGMA, class Grids>
MRL_HD bool incident(const RglDev &b, const Grids &g, float wix, float wiy, float wiz, Incident &in)
{
#pragma clang fp contract(off)
    reduce_signs(b.reduction, wix, wiy, in.fx, in.fy);
    in.wi = { (double)(wix * in.fx), (double)(wiy * in.fy), (double)wiz };
    if (!unit3(in.wi)) return false;
    in.theta_i = elevation(in.wi); in.phi_i = azimuth(in.wi.y, in.wi.x);
    in.sv = find_slices(b.vndf(), g, in.phi_i, in.theta_i);
    in.sigma4 = 1.0;
    if constexpr (WANT_SIGMA)
        if (b.jacobian) in.sigma4 = 4.0 * warp_eval(b.sigma(), single_slice(), theta2u(in.theta_i), phi2u(in.phi_i));
    return true;
}

// What eval and pdf share about a pair: the half vector's warp coordinates, their pre-image (sx, sy) under the vndf warp with the
// warp's density there, and the two geometric factors of the pdf.  ok = false: the pair evaluates to zero.
struct Half { bool ok; double u_m_x, u_m_y, sx, sy, vndf_pdf, sin_theta_m, wi_dot_m; };

// tv: where vndf's running integrals are read; fv (may be null): the cell of vndf a sample() has just visited
template <class Search>
MRL_HD Half half_lookup(const RglDev &b, const Search &tv, const Incident &in, float wox, float woy, float woz, const Found *fv)
{
#pragma clang fp contract(off)
    Half h;
    h.ok = false;
    if (!(woz > 0.0f)) return h;
    const Vec3d &wi = in.wi;
    Vec3d wo = { (double)(wox * in.fx), (double)(woy * in.fy), (double)woz };
    if (!unit3(wo)) return h;
    Vec3d m = { wi.x + wo.x, wi.y + wo.y, wi.z + wo.z };
    if (!unit3(m)) return h;
    const double theta_m = elevation(m), phi_m = azimuth(m.y, m.x);
    h.u_m_x = theta2u(theta_m);
    h.u_m_y = phi2u(b.isotropic ? phi_m - in.phi_i : phi_m);
    h.u_m_y -= floor(h.u_m_y);
    h.vndf_pdf = warp_invert(b.vndf(), tv, in.sv, h.u_m_x, h.u_m_y, h.sx, h.sy, fv);
    h.sin_theta_m = fast::sqrt_fast(__builtin_fma(m.x, m.x, m.y * m.y));
    h.wi_dot_m = __builtin_fma(wi.x, m.x, __builtin_fma(wi.y, m.y, wi.z * m.z));
    h.ok = true;
    return h;
}

// the jacobian's factor ndf(u_m) / (4 sigma(u_wi)) of the measured values (1 when the file's flag is off)
MRL_HD double value_scale(const RglDev &b, const Incident &in, const Half &h)
{
    return b.jacobian ? fast::div_fast(warp_eval(b.ndf(), single_slice(), h.u_m_x, h.u_m_y), in.sigma4) : 1.0;
}

MRL_HD float pdf_from(const Incident &in, const Half &h, double lum_pdf)
{
#pragma clang fp contract(off)
    const double jac = fmax(2.0 * kPi * kPi * h.u_m_x * h.sin_theta_m, 1e-6) * 4.0 * h.wi_dot_m;
    return (float)fast::div_fast(h.vndf_pdf * lum_pdf, jac);
}
// fl (may be null): the cell of luminance a sample() has just visited
MRL_HD float pdf_of(const RglDev &b, const Incident &in, const Half &h, const Found *fl)
{
    return pdf_from(in, h, warp_eval(b.luminance(), in.sv, h.sx, h.sy, 0, fl));
}

// eval (f cos theta_o, RGB) and / or pdf for an incident direction that is above the horizon; wo as the caller holds it.
// BATCH: the three channels' reads in flight together (12 vectors for an anisotropic file: 48 registers) — or channel by channel,
// for the callers that are short of registers (sample(): it carries two visited cells)
template <bool WANT_RGB, bool WANT_PDF, bool BATCH = true, class Search>
MRL_HD void eval_pdf_at(const RglDev &b, const Search &tv, const Incident &in, float wox, float woy, float woz, float rgb[3], float &pdf,
                        const Found *fv = nullptr, const Found *fl = nullptr)
{
#pragma clang fp contract(off)
    rgb[0] = rgb[1] = rgb[2] = 0.0f; pdf = 0.0f;
    const Half h = half_lookup(b, tv, in, wox, woy, woz, fv);
    if (!h.ok) return;
    // the luminance cell (pdf) and the three channels' cells lie at the same position of tables of one shape: located once, read together
    const WarpDev wr = b.rgb(), wl = b.luminance();
    const Cell c = locate(wr, h.sx, h.sy);
    Raw4 raw[BATCH ? 3 : 1], rawl{};
    const bool have_l = fl && fl->row == c.oy && fl->col == c.ox;
    if constexpr (WANT_RGB && BATCH)
        for (int k = 0; k < 3; ++k) raw[k] = fetch_raw(in.sv, wr, c.index, k);
    if constexpr (WANT_PDF)
        if (!have_l) rawl = fetch_raw(in.sv, wl, c.index);
    if constexpr (WANT_RGB) {
        const double scale = value_scale(b, in, h);
        for (int k = 0; k < 3; ++k) {
            if constexpr (!BATCH) raw[0] = fetch_raw(in.sv, wr, c.index, k);
            double v = cell_value(wr, c, blend4(in.sv, raw[BATCH ? k : 0]));
            v = v < 0.0 ? 0.0 : v;
            rgb[k] = (float)(v * scale);
        }
    }
    if constexpr (WANT_PDF) {
        D4 ql;
        if (have_l) ql = fl->q; else ql = blend4(in.sv, rawl);
        pdf = pdf_from(in, h, cell_value(wl, c, ql));
    }
}

// sample()'s direction for an incident direction that is above the horizon: false when the draw is rejected (reflected below the
// horizon); fl / fv receive the cells the two forward warps ended in
template <class Search>
MRL_HD bool sample_direction(const RglDev &b, const Search &tv, const Search &tl, const Incident &in, float u0, float u1, float wof[3], Found &fl, Found &fv)
{
#pragma clang fp contract(off)
    const Vec3d &wi = in.wi;
    const float fx = in.fx, fy = in.fy;
    double sx, sy, umx, umy;
    (void)warp_sample(b.luminance(), tl, in.sv, (double)u1, (double)u0, sx, sy, fl);
    (void)warp_sample(b.vndf(), tv, in.sv, sx, sy, umx, umy, fv);
    // m = (theta_m, phi_m) with theta_m = umx^2 pi/2 and phi_m = (2 umy - 1) pi [+ phi_i]: sin / cos of 2 pi (umx^2 / 4) and
    // of 2 pi umy - pi; the isotropic offset is a rotation by wi's own azimuth (cos, sin = wi_xy / |wi_xy|), no second sincos
    double st, ct, sp, cp;
    fast::sincos_2pi(0.25 * umx * umx, st, ct);
    fast::sincos_2pi(umy, sp, cp);
    sp = -sp; cp = -cp;
    if (b.isotropic) {
        const double rho2 = __builtin_fma(wi.x, wi.x, wi.y * wi.y);
        const double rr = rho2 > 0.0 ? fast::rsqrt_pos(rho2) : 0.0;
        const double ci = rho2 > 0.0 ? wi.x * rr : (__builtin_signbit(wi.x) ? -1.0 : 1.0), si = wi.y * rr;
        const double c2 = __builtin_fma(cp, ci, -(sp * si)), s2 = __builtin_fma(sp, ci, cp * si);
        cp = c2; sp = s2;
    }
    const Vec3d m = { cp * st, sp * st, ct };
    const double c = __builtin_fma(wi.x, m.x, __builtin_fma(wi.y, m.y, wi.z * m.z));
    // (the direction drawn in the stored part of the azimuth goes back through the same sign flips)
    wof[0] = (float)(__builtin_fma(2.0 * c, m.x, -wi.x) * (double)fx);
    wof[1] = (float)(__builtin_fma(2.0 * c, m.y, -wi.y) * (double)fy);
    wof[2] = (float)__builtin_fma(2.0 * c, m.z, -wi.z);
    return wof[2] > 0.0f && c > 0.0;
}

// sample() for an incident direction that is above the horizon
template <class Search>
MRL_HD void sample_at(const RglDev &b, const Search &tv, const Search &tl, const Incident &in, float u0, float u1,
                      float wo_out[3], float &pdf_out, float weight[3])
{
    wo_out[0] = wo_out[1] = wo_out[2] = 0.0f; pdf_out = 0.0f; weight[0] = weight[1] = weight[2] = 0.0f;
    float wof[3];
    Found fl, fv;
    if (!sample_direction(b, tv, tl, in, u0, u1, wof, fl, fv)) return;
    float f[3], p;
    eval_pdf_at<true, true, MRL_RGL_SAMPLE_BATCH>(b, tv, in, wof[0], wof[1], wof[2], f, p, &fv, &fl);    // at the Float direction that is returned
    if (!(p > 0.0f)) return;
    wo_out[0] = wof[0]; wo_out[1] = wof[1]; wo_out[2] = wof[2];
    pdf_out = p;
    weight[0] = f[0] / p; weight[1] = f[1] / p; weight[2] = f[2] / p;
}

// ---- spectral files: W values per unit, at the wavelengths wl[0 .. W) (wl == nullptr: at the file's own nodes, W = n_wl) ----
// the measured spectrum at warp position (sx, sy): linear between the file's wavelength nodes, clamped outside them — the
// wavelength as the third interpolated parameter, as upstream's spectral variants evaluate `spectra`
// c: the cell of the position in the values table (one per unit: every wavelength reads the same cell); g: where the file's
// wavelength grid is read.  The two nodes' corner values are read together.
template <class Grids>
MRL_HD double spectrum_at(const RglDev &b, const Grids &g, const WarpDev &wr, const Slices &sv, const Cell &c, const float *wl, int k)
{
#pragma clang fp contract(off)
    double v;
    if (!wl) {
        v = cell_value(wr, c, fetch4(sv, wr, c.index, k));
    } else {
        int c0 = 0;
        double t = 0.0;
        if (b.n_wl > 1) bracket_by([&g](int j) { return g.wavelength_at(j); }, b.n_wl, (double)wl[k], c0, t);
        const Raw4 r0 = fetch_raw(sv, wr, c.index, c0), r1 = fetch_raw(sv, wr, c.index, b.n_wl > 1 ? c0 + 1 : c0);
        v = cell_value(wr, c, blend4(sv, r0));
        if (b.n_wl > 1) v = lerp(t, v, cell_value(wr, c, blend4(sv, r1)));
    }
    return v < 0.0 ? 0.0 : v;
}

// values[0 .. W) and / or pdf; values is the caller's row (written once per wavelength, no register array)
template <bool WANT_VALUES, bool WANT_PDF, class Grids, class Search>
MRL_HD void eval_pdf_spectral_at(const RglDev &b, const Grids &g, const Search &tv, const Incident &in, float wox, float woy, float woz, const float *wl, int W,
                                 float *values, float &pdf, const Found *fv = nullptr, const Found *fl = nullptr)
{
#pragma clang fp contract(off)
    pdf = 0.0f;
    const Half h = half_lookup(b, tv, in, wox, woy, woz, fv);
    if (!h.ok) {
        if constexpr (WANT_VALUES) for (int k = 0; k < W; ++k) values[k] = 0.0f;
        return;
    }
    if constexpr (WANT_VALUES) {
        const double scale = value_scale(b, in, h);
        const WarpDev wr = b.rgb();
        const Cell c = locate(wr, h.sx, h.sy);
        for (int k = 0; k < W; ++k) values[k] = (float)(spectrum_at(b, g, wr, in.sv, c, wl, k) * scale);
    }
    if constexpr (WANT_PDF) pdf = pdf_of(b, in, h, fl);
}

template <class Grids, class Search>
MRL_HD void sample_spectral_at(const RglDev &b, const Grids &g, const Search &tv, const Search &tl, const Incident &in, float u0, float u1, const float *wl, int W,
                               float wo_out[3], float &pdf_out, float *weight)
{
#pragma clang fp contract(off)
    wo_out[0] = wo_out[1] = wo_out[2] = 0.0f; pdf_out = 0.0f;
    float wof[3];
    Found fl, fv;
    bool live = sample_direction(b, tv, tl, in, u0, u1, wof, fl, fv);
    Half h;
    float p = 0.0f;
    if (live) {
        h = half_lookup(b, tv, in, wof[0], wof[1], wof[2], &fv);                  // at the Float direction that is returned
        live = h.ok;
        if (live) { p = pdf_of(b, in, h, &fl); live = p > 0.0f; }
    }
    if (!live) { for (int k = 0; k < W; ++k) weight[k] = 0.0f; return; }
    wo_out[0] = wof[0]; wo_out[1] = wof[1]; wo_out[2] = wof[2];
    pdf_out = p;
    const double scale = value_scale(b, in, h);
    const WarpDev wr = b.rgb();
    const Cell c = locate(wr, h.sx, h.sy);
    for (int k = 0; k < W; ++k) {
        const float f = (float)(spectrum_at(b, g, wr, in.sv, c, wl, k) * scale);
        weight[k] = f / p;
    }
}

// ---- one unit through the image in memory (host images: mrl_host_*; tests/rgl_host_harness.hip) ----
// eval (f cos theta_o, RGB) and / or pdf of one unit; every output zero outside the upper hemisphere
template <bool WANT_RGB, bool WANT_PDF>
MRL_HD void eval_pdf(const RglDev &b, float wix, float wiy, float wiz, float wox, float woy, float woz, float rgb[3], float &pdf)
{
    rgb[0] = rgb[1] = rgb[2] = 0.0f; pdf = 0.0f;
    Incident in;
    if (!(wiz > 0.0f) || !(woz > 0.0f) || !incident<WANT_RGB>(b, GridMem{ b.phi, b.theta }, wix, wiy, wiz, in)) return;
    eval_pdf_at<WANT_RGB, WANT_PDF>(b, SearchMem(b.vndf()), in, wox, woy, woz, rgb, pdf);
}

MRL_HD void sample(const RglDev &b, float wix, float wiy, float wiz, float u0, float u1, float wo_out[3], float &pdf_out, float weight[3])
{
    wo_out[0] = wo_out[1] = wo_out[2] = 0.0f; pdf_out = 0.0f; weight[0] = weight[1] = weight[2] = 0.0f;
    Incident in;
    if (!(wiz > 0.0f) || !incident<true>(b, GridMem{ b.phi, b.theta }, wix, wiy, wiz, in)) return;
    sample_at(b, SearchMem(b.vndf()), SearchMem(b.luminance()), in, u0, u1, wo_out, pdf_out, weight);
}

// a spectral file, one unit through the image in memory
template <bool WANT_VALUES, bool WANT_PDF>
MRL_HD void eval_pdf_spectral(const RglDev &b, float wix, float wiy, float wiz, float wox, float woy, float woz, const float *wl, int W, float *values, float &pdf)
{
    pdf = 0.0f;
    Incident in;
    if (!(wiz > 0.0f) || !(woz > 0.0f) || !incident<WANT_VALUES>(b, GridMem{ b.phi, b.theta }, wix, wiy, wiz, in)) {
        if constexpr (WANT_VALUES) for (int k = 0; k < W; ++k) values[k] = 0.0f;
        return;
    }
    eval_pdf_spectral_at<WANT_VALUES, WANT_PDF>(b, GridMem{ b.phi, b.theta, b.wavelengths }, SearchMem(b.vndf()), in, wox, woy, woz, wl, W, values, pdf);
}

MRL_HD void sample_spectral(const RglDev &b, float wix, float wiy, float wiz, float u0, float u1, const float *wl, int W,
                            float wo_out[3], float &pdf_out, float *weight)
{
    wo_out[0] = wo_out[1] = wo_out[2] = 0.0f; pdf_out = 0.0f;
    Incident in;
    if (!(wiz > 0.0f) || !incident<true>(b, GridMem{ b.phi, b.theta }, wix, wiy, wiz, in)) { for (int k = 0; k < W; ++k) weight[k] = 0.0f; return; }
    sample_spectral_at(b, GridMem{ b.phi, b.theta, b.wavelengths }, SearchMem(b.vndf()), SearchMem(b.luminance()), in, u0, u1, wl, W, wo_out, pdf_out, weight);
}

} // namespace rgl
} // namespace mrl
