// merl_rgl.hip — the adaptive-parameterisation measured BSDF (RGL *.bsdf; upstream Mitsuba 3 `measured`) on gfx950:
// the host-side image builder (normalisation + running integrals, f64, in the oracle's loop order) and the kernels.
// PARITY UNPINNED (see merl_rgl.hpp).  SURVEY.md §8f item 3.
//
//   k_rgl<MODE, INDEXED, MULTI>   one lane = one unit, grid-stride; the material's descriptor (five WarpDev) arrives by value in
//                          SGPRs; every table read is a per-lane gather served by L1/L2 (a material is 0.5 - 50 MB).
// What bounds it: the number of scattered lane-addresses the CU's texture addresser resolves (93 % of wave cycles wait on
// memory, VALU is under 10 % busy: profiles/r03_rgl_pmc.json) — hence the image's cell bricks (one 16-B load per cell and
// slice where the file's node-major layout needs four); measured rates: DESIGN.md §5c.
#include "merl_kernels.hpp"
#include "merl_rgl.hpp"

#include <cmath>
#include <vector>

namespace mrl {

namespace {

constexpr int kRglBlock = 256;

// MULTI: a batch with a material id per unit — the lanes whose id names an RGL material evaluate it through the descriptor
// stored behind that material's image (read on demand: a few more cache-resident loads per lookup) and overwrite the zeros
// the table / GGX kernel of the same call left there; every other lane skips.  Launched after that kernel, on the same stream.
template <int MODE>
__device__ __forceinline__ void rgl_unit(const BatchArgs &a, const RglDev &r, size_t i)
{
    constexpr bool has_eval = MODE == 0 || MODE == 3 || MODE == 4, has_pdf = MODE == 1 || MODE == 3 || MODE == 4,
                   has_sample = MODE == 2 || MODE == 3;
    const float wix = a.wi[3 * i], wiy = a.wi[3 * i + 1], wiz = a.wi[3 * i + 2];
    if constexpr (has_eval || has_pdf) {
        const float wox = a.wo[3 * i], woy = a.wo[3 * i + 1], woz = a.wo[3 * i + 2];
        float rgb[3], pdf;
        rgl::eval_pdf<has_eval, has_pdf>(r, wix, wiy, wiz, wox, woy, woz, rgb, pdf);
        if constexpr (has_eval) { a.out_rgb[3 * i] = rgb[0]; a.out_rgb[3 * i + 1] = rgb[1]; a.out_rgb[3 * i + 2] = rgb[2]; }
        if constexpr (has_pdf) a.out_pdf[i] = pdf;
    }
    if constexpr (has_sample) {
        float wo2[3], pdf2, w[3];
        rgl::sample(r, wix, wiy, wiz, a.u[2 * i], a.u[2 * i + 1], wo2, pdf2, w);
        a.out_wo[3 * i] = wo2[0]; a.out_wo[3 * i + 1] = wo2[1]; a.out_wo[3 * i + 2] = wo2[2];
        a.out_pdf2[i] = pdf2;
        a.out_weight[3 * i] = w[0]; a.out_weight[3 * i + 1] = w[1]; a.out_weight[3 * i + 2] = w[2];
    }
}

template <int MODE, bool INDEXED, bool MULTI>
__global__ __launch_bounds__(kRglBlock) void k_rgl(BatchArgs a, RglDev r)
{
    const size_t stride = (size_t)gridDim.x * kRglBlock;
    size_t n_items = a.n;
    if constexpr (INDEXED) { const size_t c = (size_t)*a.idx_count; n_items = c < a.n ? c : a.n; }
    for (size_t j = (size_t)blockIdx.x * kRglBlock + threadIdx.x; j < n_items; j += stride) {
        const size_t i = INDEXED ? (size_t)a.idx[j] : j;
        if constexpr (MULTI) {
            const int id = a.mat[i];
            if (id < 0 || id >= a.n_materials) continue;
            const MaterialDev &m = a.materials[id];
            if (m.kind != KIND_RGL) continue;
            rgl_unit<MODE>(a, *(const RglDev *)m.rgl, i);
        } else {
            rgl_unit<MODE>(a, r, i);
        }
    }
}

template <int MODE>
hipError_t launch_mode(const BatchArgs &a, const RglDev *r, bool indexed, int compute_units, hipStream_t stream)
{
    size_t blocks = (a.n + kRglBlock - 1) / kRglBlock;
    const size_t cap = (size_t)compute_units * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const dim3 grid((unsigned)blocks), block(kRglBlock);
    if (!r) {                                                   // a batch with material ids: descriptors come from the material array
        const RglDev none{};
        if (indexed) hipLaunchKernelGGL((k_rgl<MODE, true, true>), grid, block, 0, stream, a, none);
        else hipLaunchKernelGGL((k_rgl<MODE, false, true>), grid, block, 0, stream, a, none);
    } else {
        if (indexed) hipLaunchKernelGGL((k_rgl<MODE, true, false>), grid, block, 0, stream, a, *r);
        else hipLaunchKernelGGL((k_rgl<MODE, false, false>), grid, block, 0, stream, a, *r);
    }
    return hipGetLastError();
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

// appends one function's tables to the image (offsets in floats, each a multiple of 4 so that the vectors are 16-B aligned):
// the corner bricks (normalised if a distribution) and, for distributions, the running integrals `cond` (along x, node rows
// row / row + 1 side by side) and `rows` (marginal cdf before / after the cell row, totals of its two node rows) — all rounded
// to Float once from f64 sums, in the oracle's loop order.  src: [slices][n_ch][ny][nx].
WarpOffsets append_warp(std::vector<float> &blob, const float *src_all, int nx, int ny, size_t slices, int n_ch, bool distribution)
{
    const size_t per = (size_t)nx * ny, cells = (size_t)(nx - 1) * (size_t)(ny - 1);
    size_t at = blob.size();
    const WarpOffsets off = plan_warp(at, nx, ny, slices, n_ch, distribution);
    blob.resize(at, 0.0f);
    std::vector<double> cond((size_t)ny * (size_t)(nx - 1)), marg((size_t)(ny - 1));
    std::vector<float> node(per), condf(cond.size()), margf(marg.size());
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
                for (size_t k = 0; k < cond.size(); ++k) condf[k] = (float)(cond[k] * norm);
                for (size_t k = 0; k < marg.size(); ++k) margf[k] = (float)(marg[k] * norm);
            }
            for (size_t k = 0; k < per; ++k) node[k] = (float)((double)src[k] * norm);
            for (int y = 0; y < ny - 1; ++y)
                for (int x = 0; x < nx - 1; ++x) {
                    const size_t cell = (size_t)y * (size_t)(nx - 1) + (size_t)x;
                    float *q = &blob[off.cells + ((s * cells + cell) * (size_t)n_ch + (size_t)ch) * 4];
                    q[0] = node[(size_t)y * nx + x]; q[1] = node[(size_t)y * nx + x + 1];
                    q[2] = node[(size_t)(y + 1) * nx + x]; q[3] = node[(size_t)(y + 1) * nx + x + 1];
                    if (distribution) {
                        float *c = &blob[off.cond + (s * cells + cell) * 2];
                        c[0] = condf[(size_t)y * (size_t)(nx - 1) + (size_t)x]; c[1] = condf[(size_t)(y + 1) * (size_t)(nx - 1) + (size_t)x];
                    }
                }
            if (distribution)
                for (int y = 0; y < ny - 1; ++y) {
                    float *r = &blob[off.rows + (s * (size_t)(ny - 1) + (size_t)y) * 4];
                    r[0] = y > 0 ? margf[(size_t)y - 1] : 0.0f; r[1] = margf[(size_t)y];
                    r[2] = condf[(size_t)y * (size_t)(nx - 1) + (size_t)(nx - 2)]; r[3] = condf[(size_t)(y + 1) * (size_t)(nx - 1) + (size_t)(nx - 2)];
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
    return span > 0.0 ? (int)std::floor(2.0 * kPi / span + 0.5) : 0;
}

const char *rgl_check_fields(const RglFields &f)
{
    if (const char *why = rgl_check_shapes(f)) return why;
    if (!f.phi_i || !f.theta_i || !f.ndf || !f.sigma || !f.vndf || !f.luminance || !f.rgb) return "null field";
    const size_t slices = (size_t)f.n_phi * (size_t)f.n_theta, per = (size_t)f.res[0] * (size_t)f.res[1];
    if (!all_finite(f.phi_i, (size_t)f.n_phi) || !all_finite(f.theta_i, (size_t)f.n_theta) || !ascending(f.phi_i, f.n_phi) || !ascending(f.theta_i, f.n_theta))
        return "phi_i / theta_i must be finite and strictly ascending";
    if (!all_finite(f.ndf, (size_t)f.res_ndf[0] * f.res_ndf[1]) || !all_finite(f.sigma, (size_t)f.res_sigma[0] * f.res_sigma[1]) ||
        !all_finite(f.vndf, slices * per) || !all_finite(f.luminance, slices * per) || !all_finite(f.rgb, slices * per * 3))
        return "non-finite table value";
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
    const size_t slices = (size_t)f.n_phi * (size_t)f.n_theta;
    RglLayout l;
    l.phi = 0; l.theta = (size_t)f.n_phi;
    blob.insert(blob.end(), f.phi_i, f.phi_i + f.n_phi);
    blob.insert(blob.end(), f.theta_i, f.theta_i + f.n_theta);
    auto put = [&](int which, const float *src, const int res[2], size_t n, int n_ch, bool distribution) {
        const WarpOffsets o = append_warp(blob, src, res[0], res[1], n, n_ch, distribution);
        l.cells[which] = o.cells; l.cond[which] = o.cond; l.rows[which] = o.rows;
    };
    put(0, f.ndf, f.res_ndf, 1, 1, false);
    put(1, f.sigma, f.res_sigma, 1, 1, false);
    put(2, f.vndf, f.res, slices, 1, true);
    put(3, f.luminance, f.res, slices, 1, true);
    put(4, f.rgb, f.res, slices, 3, false);
    return l;
}

// base: 16-B aligned
RglDev rgl_descriptor(const RglFields &f, const RglLayout &l, const float *base)
{
    auto warp = [&](int which, const int res[2], int n_phi, int n_theta, int n_ch, bool distribution) {
        WarpDev w;
        w.cells = (const float4 *)(base + l.cells[which]);
        w.cond = distribution ? (const float2 *)(base + l.cond[which]) : nullptr;
        w.rows = distribution ? (const float4 *)(base + l.rows[which]) : nullptr;
        w.phi = base + l.phi; w.theta = base + l.theta;
        w.nx = res[0]; w.ny = res[1]; w.n_phi = n_phi; w.n_theta = n_theta; w.n_ch = n_ch;
        w.normalized = distribution ? 1 : 0;
        return w;
    };
    RglDev r;
    r.ndf = warp(0, f.res_ndf, 1, 1, 1, false);
    r.sigma = warp(1, f.res_sigma, 1, 1, 1, false);
    r.vndf = warp(2, f.res, f.n_phi, f.n_theta, 1, true);
    r.luminance = warp(3, f.res, f.n_phi, f.n_theta, 1, true);
    r.rgb = warp(4, f.res, f.n_phi, f.n_theta, 3, false);
    r.isotropic = f.n_phi <= 2;
    r.jacobian = f.jacobian ? 1 : 0;
    r.reduction = rgl_reduction(f);
    return r;
}

hipError_t launch_rgl(int mode, const BatchArgs &a, const RglDev *r, bool indexed, int compute_units, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    switch (mode) {
        case 0: return launch_mode<0>(a, r, indexed, compute_units, stream);
        case 1: return launch_mode<1>(a, r, indexed, compute_units, stream);
        case 2: return launch_mode<2>(a, r, indexed, compute_units, stream);
        case 3: return launch_mode<3>(a, r, indexed, compute_units, stream);
        case 4: return launch_mode<4>(a, r, indexed, compute_units, stream);
    }
    return hipErrorInvalidValue;
}

} // namespace mrl
