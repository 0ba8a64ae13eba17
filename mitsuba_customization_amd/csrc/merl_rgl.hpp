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
//                                 sampling (`cond`: along x per node row, `marg`: over rows)
//     rgb                         the measured values in the warped domain, per (phi_i, theta_i, channel)
// eval(wi, wo):  m = (wi + wo) normalised;  u_m = (sqrt(2 theta_m / pi), (phi_m [- phi_i]) / 2pi + 1/2);
//                s = vndf.invert(u_m);  f cos = rgb(s) * ndf(u_m) / (4 sigma(u_wi))
// pdf(wi, wo):   vndf.pdf(u_m) * luminance(s) / (max(2 pi^2 u_m.x sin theta_m, 1e-6) * 4 (wi . m))
// sample(wi, u): s = luminance.sample(u), u_m = vndf.sample(s), wo = reflect(wi, m(u_m)); what is reported is eval / pdf
//                AT the Float direction that is returned (so pdf(wi, sample.wo) == sample.pdf and weight == eval / pdf).
// Math in f64 on Float tables, one lane per unit; every table read is a plain gather (tables are a few hundred KB to a
// few MB per material: cache resident), so this path is latency / VALU bound, not HBM bound.
#pragma once
#include "merl_device.hpp"

namespace mrl {

// one piecewise-bilinear function; slices are row-major in (phi, theta, channel), a slice is [ny][nx] nodes, x fastest
struct WarpDev {
    const float *data;      // [slices][ny][nx]        (divided by the slice's integral when normalised)
    const float *marg;      // [slices][ny - 1]        running integral over rows         (distributions only)
    const float *cond;      // [slices][ny][nx - 1]    running integral along a node row  (distributions only)
    const float *phi, *theta;   // ascending parameter grids (unused when the count is 1)
    int nx, ny, n_phi, n_theta, n_ch;
    int normalized;
};

struct RglDev {
    WarpDev ndf, sigma, vndf, luminance, rgb;
    int isotropic;          // n_phi <= 2: phi_m is measured relative to phi_i
    int jacobian;           // the file's flag: multiply the spectrum by ndf / (4 sigma)
};

namespace rgl {

// the four parameter slices around (phi_i, theta_i) and their weights, phi fastest (the order the oracle sums in); `mask`
// says which entries exist (a grid of one node has no upper neighbour) — uniform over a launch, so fetch()'s tests are
// scalar branches and the arrays stay in registers
struct Slices { int s[4]; double w[4]; int mask; };

MRL_HD void bracket(const float *grid, int n, double p, int &i, double &t)
{
    // largest i in [0, n - 2] with grid[i] <= p
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((double)grid[mid] <= p) lo = mid; else hi = mid; }
    i = lo;
    const double p0 = grid[lo], p1 = grid[lo + 1];
    t = (p - p0) / (p1 - p0);
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
}

MRL_HD Slices find_slices(const WarpDev &w, double phi_i, double theta_i)
{
    int ip = 0, it = 0;
    double tp = 0.0, tt = 0.0;
    if (w.n_phi > 1) bracket(w.phi, w.n_phi, phi_i, ip, tp);
    if (w.n_theta > 1) bracket(w.theta, w.n_theta, theta_i, it, tt);
    const int ip1 = w.n_phi > 1 ? ip + 1 : ip, it1 = w.n_theta > 1 ? it + 1 : it;
    Slices out;
    out.s[0] = ip * w.n_theta + it;  out.w[0] = (1.0 - tp) * (1.0 - tt);
    out.s[1] = ip1 * w.n_theta + it; out.w[1] = tp * (1.0 - tt);
    out.s[2] = ip * w.n_theta + it1; out.w[2] = (1.0 - tp) * tt;
    out.s[3] = ip1 * w.n_theta + it1; out.w[3] = tp * tt;
    out.mask = 1 | (w.n_phi > 1 ? 2 : 0) | (w.n_theta > 1 ? 4 : 0) | (w.n_phi > 1 && w.n_theta > 1 ? 8 : 0);
    return out;
}

MRL_HD Slices single_slice()
{
    Slices out;
    for (int k = 0; k < 4; ++k) { out.s[k] = 0; out.w[k] = 0.0; }
    out.w[0] = 1.0; out.mask = 1;
    return out;
}

MRL_HD double fetch(const Slices &s, const float *base, int per_slice, int index)
{
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if ((s.mask >> k) & 1) v += s.w[k] * (double)base[(size_t)s.s[k] * (size_t)per_slice + (size_t)index];
    return v;
}

MRL_HD int clamp_cell(double p, int last)
{
    int i = trunc_i(p);
    return i < 0 ? 0 : (i > last ? last : i);
}

MRL_HD double warp_eval(const WarpDev &w, const Slices &s, double x_in, double y_in)
{
    const double px = x_in * (double)(w.nx - 1), py = y_in * (double)(w.ny - 1);
    const int ox = clamp_cell(px, w.nx - 2), oy = clamp_cell(py, w.ny - 2);
    const double fx = px - (double)ox, fy = py - (double)oy;
    const int per = w.nx * w.ny, idx = oy * w.nx + ox;
    const double v00 = fetch(s, w.data, per, idx), v10 = fetch(s, w.data, per, idx + 1);
    const double v01 = fetch(s, w.data, per, idx + w.nx), v11 = fetch(s, w.data, per, idx + w.nx + 1);
    const double v = (1.0 - fy) * ((1.0 - fx) * v00 + fx * v10) + fy * ((1.0 - fx) * v01 + fx * v11);
    return w.normalized ? v * (double)(w.nx - 1) * (double)(w.ny - 1) : v;
}

MRL_HD double safe_sqrt(double x) { return x > 0.0 ? sqrt(x) : 0.0; }
MRL_HD double clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }

// position in [0, 1] at which a density running linearly from c0 to c1 has accumulated the mass u
MRL_HD double invert_linear(double c0, double c1, double u)
{
    const bool is_const = fabs(c0 - c1) < 1e-4 * (c0 + c1);
    const double num = is_const ? 2.0 * u : c0 - safe_sqrt(c0 * c0 - 2.0 * u * (c0 - c1));
    const double den = is_const ? c0 + c1 : c0 - c1;
    return den != 0.0 ? num / den : 0.0;
}

// uniform sample -> position; returns the density there
MRL_HD double warp_sample(const WarpDev &w, const Slices &s, double ux, double uy, double &x_out, double &y_out)
{
    const int nx = w.nx, ny = w.ny;
    const int per_m = ny - 1, per_c = ny * (nx - 1), per_d = nx * ny;
    ux = clamp01(ux); uy = clamp01(uy);
    int lo = 0, hi = ny - 2;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (fetch(s, w.marg, per_m, mid) < uy) lo = mid + 1; else hi = mid; }
    const int row = lo;
    if (row > 0) uy -= fetch(s, w.marg, per_m, row - 1);
    const double r0 = fetch(s, w.cond, per_c, row * (nx - 1) + (nx - 2));
    const double r1 = fetch(s, w.cond, per_c, (row + 1) * (nx - 1) + (nx - 2));
    const double y = clamp01(invert_linear(r0, r1, uy));
    ux *= (1.0 - y) * r0 + y * r1;
    lo = 0; hi = nx - 2;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const double c = (1.0 - y) * fetch(s, w.cond, per_c, row * (nx - 1) + mid) + y * fetch(s, w.cond, per_c, (row + 1) * (nx - 1) + mid);
        if (c < ux) lo = mid + 1; else hi = mid;
    }
    const int col = lo;
    if (col > 0)
        ux -= (1.0 - y) * fetch(s, w.cond, per_c, row * (nx - 1) + col - 1) + y * fetch(s, w.cond, per_c, (row + 1) * (nx - 1) + col - 1);
    const int idx = row * nx + col;
    const double v00 = fetch(s, w.data, per_d, idx), v10 = fetch(s, w.data, per_d, idx + 1);
    const double v01 = fetch(s, w.data, per_d, idx + nx), v11 = fetch(s, w.data, per_d, idx + nx + 1);
    const double c0 = (1.0 - y) * v00 + y * v01, c1 = (1.0 - y) * v10 + y * v11;
    const double x = clamp01(invert_linear(c0, c1, ux));
    x_out = ((double)col + x) / (double)(nx - 1);
    y_out = ((double)row + y) / (double)(ny - 1);
    return ((1.0 - x) * c0 + x * c1) * (double)(nx - 1) * (double)(ny - 1);
}

// position -> the uniform sample that maps to it; returns the density at the position
MRL_HD double warp_invert(const WarpDev &w, const Slices &s, double x_in, double y_in, double &ux_out, double &uy_out)
{
    const int nx = w.nx, ny = w.ny;
    const int per_m = ny - 1, per_c = ny * (nx - 1), per_d = nx * ny;
    const double px = x_in * (double)(nx - 1), py = y_in * (double)(ny - 1);
    const int col = clamp_cell(px, nx - 2), row = clamp_cell(py, ny - 2);
    const double x = px - (double)col, y = py - (double)row;
    const int idx = row * nx + col;
    const double v00 = fetch(s, w.data, per_d, idx), v10 = fetch(s, w.data, per_d, idx + 1);
    const double v01 = fetch(s, w.data, per_d, idx + nx), v11 = fetch(s, w.data, per_d, idx + nx + 1);
    const double c0 = (1.0 - y) * v00 + y * v01, c1 = (1.0 - y) * v10 + y * v11;
    const double pdf = ((1.0 - x) * c0 + x * c1) * (double)(nx - 1) * (double)(ny - 1);
    double sx = x * (c0 + 0.5 * x * (c1 - c0));
    if (col > 0)
        sx += (1.0 - y) * fetch(s, w.cond, per_c, row * (nx - 1) + col - 1) + y * fetch(s, w.cond, per_c, (row + 1) * (nx - 1) + col - 1);
    const double r0 = fetch(s, w.cond, per_c, row * (nx - 1) + (nx - 2));
    const double r1 = fetch(s, w.cond, per_c, (row + 1) * (nx - 1) + (nx - 2));
    const double tot = (1.0 - y) * r0 + y * r1;
    ux_out = tot > 0.0 ? sx / tot : 0.0;
    double sy = y * (r0 + 0.5 * y * (r1 - r0));
    if (row > 0) sy += fetch(s, w.marg, per_m, row - 1);
    uy_out = sy;
    return pdf;
}

// 2 asin(|d - z| / 2): acos(d.z) without its cancellation near the pole
MRL_HD double elevation(const Vec3d &d)
{
    const double dz = d.z - 1.0;
    const double h = 0.5 * sqrt(d.x * d.x + d.y * d.y + dz * dz);
    return 2.0 * asin(h > 1.0 ? 1.0 : h);
}
MRL_HD double theta2u(double t) { return sqrt(t * (2.0 / kPi)); }
MRL_HD double phi2u(double p) { return (p + kPi) * (0.5 / kPi); }
MRL_HD double u2theta(double u) { return u * u * (kPi / 2.0); }
MRL_HD double u2phi(double u) { return (2.0 * u - 1.0) * kPi; }

MRL_HD bool unit3(Vec3d &v)
{
    const double n = sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    if (!(n > 0.0)) return false;
    v.x /= n; v.y /= n; v.z /= n;
    return true;
}

// eval (f cos theta_o, RGB) and / or pdf of one unit; every output zero outside the upper hemisphere
template <bool WANT_RGB, bool WANT_PDF>
MRL_HD void eval_pdf(const RglDev &b, float wix, float wiy, float wiz, float wox, float woy, float woz, float rgb[3], float &pdf)
{
    rgb[0] = rgb[1] = rgb[2] = 0.0f; pdf = 0.0f;
    if (!(wiz > 0.0f) || !(woz > 0.0f)) return;
    Vec3d wi = { (double)wix, (double)wiy, (double)wiz }, wo = { (double)wox, (double)woy, (double)woz };
    if (!unit3(wi) || !unit3(wo)) return;
    Vec3d m = { wi.x + wo.x, wi.y + wo.y, wi.z + wo.z };
    if (!unit3(m)) return;
    const double theta_i = elevation(wi), phi_i = atan2(wi.y, wi.x);
    const double theta_m = elevation(m), phi_m = atan2(m.y, m.x);
    const double u_wi_x = theta2u(theta_i), u_wi_y = phi2u(phi_i);
    const double u_m_x = theta2u(theta_m);
    double u_m_y = phi2u(b.isotropic ? phi_m - phi_i : phi_m);
    u_m_y -= floor(u_m_y);
    const Slices sv = find_slices(b.vndf, phi_i, theta_i);          // vndf, luminance and rgb share the parameter grids
    double sx, sy;
    const double vndf_pdf = warp_invert(b.vndf, sv, u_m_x, u_m_y, sx, sy);
    if constexpr (WANT_RGB) {
        double scale = 1.0;
        if (b.jacobian) {
            const Slices one = single_slice();
            scale = warp_eval(b.ndf, one, u_m_x, u_m_y) / (4.0 * warp_eval(b.sigma, one, u_wi_x, u_wi_y));
        }
        for (int c = 0; c < 3; ++c) {
            Slices sc = sv;
            for (int k = 0; k < 4; ++k) sc.s[k] = sv.s[k] * 3 + c;
            double v = warp_eval(b.rgb, sc, sx, sy);
            v = v < 0.0 ? 0.0 : v;
            rgb[c] = (float)(v * scale);
        }
    }
    if constexpr (WANT_PDF) {
        const double lum_pdf = warp_eval(b.luminance, sv, sx, sy);
        const double sin_theta_m = sqrt(m.x * m.x + m.y * m.y);
        const double jac = fmax(2.0 * kPi * kPi * u_m_x * sin_theta_m, 1e-6) * 4.0 * (wi.x * m.x + wi.y * m.y + wi.z * m.z);
        pdf = (float)(vndf_pdf * lum_pdf / jac);
    }
}

MRL_HD void sample(const RglDev &b, float wix, float wiy, float wiz, float u0, float u1, float wo_out[3], float &pdf_out, float weight[3])
{
    wo_out[0] = wo_out[1] = wo_out[2] = 0.0f; pdf_out = 0.0f; weight[0] = weight[1] = weight[2] = 0.0f;
    if (!(wiz > 0.0f)) return;
    Vec3d wi = { (double)wix, (double)wiy, (double)wiz };
    if (!unit3(wi)) return;
    const double theta_i = elevation(wi), phi_i = atan2(wi.y, wi.x);
    const Slices sv = find_slices(b.vndf, phi_i, theta_i);
    double sx, sy, umx, umy;
    (void)warp_sample(b.luminance, sv, (double)u1, (double)u0, sx, sy);
    (void)warp_sample(b.vndf, sv, sx, sy, umx, umy);
    double phi_m = u2phi(umy);
    const double theta_m = u2theta(umx);
    if (b.isotropic) phi_m += phi_i;
    const double st = sin(theta_m), ct = cos(theta_m);
    const Vec3d m = { cos(phi_m) * st, sin(phi_m) * st, ct };
    const double c = wi.x * m.x + wi.y * m.y + wi.z * m.z;
    const float wof[3] = { (float)(2.0 * c * m.x - wi.x), (float)(2.0 * c * m.y - wi.y), (float)(2.0 * c * m.z - wi.z) };
    if (!(wof[2] > 0.0f) || !(c > 0.0)) return;
    float f[3], p;
    eval_pdf<true, true>(b, wix, wiy, wiz, wof[0], wof[1], wof[2], f, p);
    if (!(p > 0.0f)) return;
    wo_out[0] = wof[0]; wo_out[1] = wof[1]; wo_out[2] = wof[2];
    pdf_out = p;
    weight[0] = f[0] / p; weight[1] = f[1] / p; weight[2] = f[2] / p;
}

} // namespace rgl
} // namespace mrl
