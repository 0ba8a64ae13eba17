// merl_table_fast.hpp — the tuned per-lane path for table materials (MERL / customized_measurement).
//
// Same real-valued functions as merl_device.hpp (rows a2–a6 of SURVEY.md §8a), re-expressed for
// the CDNA4 VALU:
//   * f64 division / sqrt / atan2 from ocml (IEEE-exact, ~100+ instructions each with their
//     scaling and special-case paths) are replaced by v_rcp_f64 / v_rsq_f64 seeds + one Newton
//     step (relative error ~1e-15, no denormal scaling: operands here are O(1)), and a
//     reduced-range odd polynomial for atan (|r| <= tan(pi/8), degree 8 in r^2, abs error
//     9.4e-15).  The table coordinate stays good to ~1e-12 texel — far inside the 1e-8 budget.
//   * branch-free: guards become selects at the very end, so the eval lookup and the sample
//     lookup of one unit sit in one basic block and their 16 gathers overlap with the ALU work.
#pragma once
#include "merl_device.hpp"

namespace mrl {
namespace fast {

// n / d for finite d != 0: v_rcp_f64 seed + one Newton step (relative error ~1e-15; the
// coordinate budget is 1e-10, so no residual correction)
MRL_HD double div_fast(double n, double d)
{
    return n * rcp_nr(d);
}

constexpr double kTiny = 1e-280;

// sqrt(x) and 1/sqrt(x) for x >= 0: v_rsq_f64 seed + one coupled Newton step.  x is floored at
// kTiny so that x == 0 needs no select (sqrt -> 1e-140 ~ 0).
MRL_HD void sqrt_rsqrt(double x, double &s, double &rs)
{
    x = __builtin_fmax(x, kTiny);
    double y = rsq_seed(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    s = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    rs = h + h;
}
MRL_HD double sqrt_fast(double x)
{
    x = __builtin_fmax(x, kTiny);
    double y = rsq_seed(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    return __builtin_fma(g, r, g);
}
// sqrt(x) for x > 0 known (a sum of squares of a direction that passed the cosine guards): no floor
MRL_HD double sqrt_pos(double x)
{
    double y = rsq_seed(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    return __builtin_fma(g, r, g);
}
// 1/sqrt(x) for x > 0: seed + one Newton step, y (1 + e/2) with e = 1 - x y^2.  x == 0 gives NaN (0 * inf): callers pass
// squared lengths of directions whose lanes are masked by the cosine guards when the length is zero.
MRL_HD double rsqrt_pos(double x)
{
    const double y = rsq_seed(x);
    const double e = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(0.5 * y, e, y);
}

// atan2(a, b) for a >= 0, b >= 0 -> [0, pi/2].
// With mn = min, mx = max:  atan(mn/mx) = pi/8 + atan(r),  r = (mn - c mx) / (mx + c mn),  c = tan(pi/8),
// and |r| <= c for every mn/mx in [0,1] — a fixed rotation by pi/8, so no range select is needed.
// atan(r) = r P(r^2), P of degree 8 (abs error 9.4e-15 on |r| <= c).  a > b mirrors about pi/4.
// (a, b) = (0, 0) returns pi/8: phi_d is undefined there (degenerate h or retro-reflection).
// FLOOR = false: (a, b) != (0, 0) is known (den > 0), the kTiny floor on the denominator is dropped.
template <bool FLOOR = true>
MRL_HD double atan2_q1(double a, double b)
{
    constexpr double C = 0.41421356237309503;            // tan(pi/8)
    constexpr double PI_8 = 0.39269908169872415481;
    constexpr double PI_3_8 = 1.17809724509617246442;
    const double mn = __builtin_fmin(a, b), mx = __builtin_fmax(a, b);
    const double num = __builtin_fma(-C, mx, mn);
    double den = __builtin_fma(C, mn, mx);
    if constexpr (FLOOR) den = __builtin_fmax(den, kTiny);
    const double r = div_fast(num, den);
    const double z = r * r;
    double p = 0x1.f5ef263ad0056p-6;
    p = __builtin_fma(p, z, -0x1.e116a805760c8p-5);
    p = __builtin_fma(p, z, 0x1.35c9c1f5f80ecp-4);
    p = __builtin_fma(p, z, -0x1.73d90b0295f34p-4);
    p = __builtin_fma(p, z, 0x1.c714c3f46eb8ep-4);
    p = __builtin_fma(p, z, -0x1.249228bbcc3c1p-3);
    p = __builtin_fma(p, z, 0x1.9999990f93ea4p-3);
    p = __builtin_fma(p, z, -0x1.55555554e3467p-2);
    p = __builtin_fma(p, z, 0x1.fffffffffff02p-1);
    const bool swap = a > b;
    // swap ? (3pi/8 - r p) : (pi/8 + r p)
    return __builtin_fma(swap ? -r : r, p, swap ? PI_3_8 : PI_8);
}

struct Vec3 { double x, y, z; };

MRL_HD Vec3 normalize_f32(float x, float y, float z)
{
    const double dx = x, dy = y, dz = z;
    const double rs = rsqrt_pos(__builtin_fma(dx, dx, __builtin_fma(dy, dy, dz * dz)));
    return { dx * rs, dy * rs, dz * rs };
}

// An outgoing direction as the coordinate maps take it: the raw components and the reciprocal of their length.  The maps
// fold the normalisation into the sum / difference with the (unit) incident direction — s = in + rs out, e = in - rs out
// as six explicit FMAs.  Left to the compiler's contraction the same six FMAs come out, but WHICH multiply is fused
// depends on the inlining context (basic-block boundaries), the coordinates then differ by an ulp (1e-14 texel) from one
// entry point to the next, and the Float corner weights turn that into a one-ulp difference in about two results per
// million: explicit here, and contraction is off in everything downstream of the coordinates.
struct Dir { double x, y, z, rs; };
MRL_HD Dir dir_f32(float x, float y, float z)
{
    const double dx = x, dy = y, dz = z;
    return { dx, dy, dz, rsqrt_pos(__builtin_fma(dx, dx, __builtin_fma(dy, dy, dz * dz))) };
}
MRL_HD Vec3 unit(const Dir &d)
{
#pragma clang fp contract(off)
    return { d.x * d.rs, d.y * d.rs, d.z * d.rs };
}

// a2 + a3 for a unit incident and any outgoing direction (see merl_device.hpp::half_diff_coords for the derivation)
MRL_HD Coords coords(const Vec3 &in, const Dir &out, double k_th, double k_td, double k_pd)
{
#pragma clang fp contract(off)
    const double sx = __builtin_fma(out.x, out.rs, in.x), sy = __builtin_fma(out.y, out.rs, in.y), sz = __builtin_fma(out.z, out.rs, in.z);
    const double ex = __builtin_fma(-out.x, out.rs, in.x), ey = __builtin_fma(-out.y, out.rs, in.y), ez = __builtin_fma(-out.z, out.rs, in.z);
    const double rho2 = __builtin_fma(sx, sx, sy * sy);
    const double s2 = __builtin_fma(sz, sz, rho2);
    const double e2 = __builtin_fma(ex, ex, __builtin_fma(ey, ey, ez * ez));
    // |s| > 0 for every pair that passes the cosine guards (s_z > 0); rho and |e| are zero for h == n / retro-reflection
    const double rho = sqrt_fast(rho2), ns = sqrt_pos(s2), ne = sqrt_fast(e2);
    const double th = atan2_q1<false>(rho, sz);
    const double td = atan2_q1<false>(ne, ns);
    double py = __builtin_fma(ey, sx, -(ex * sy));
    double px = -ez * ns;
    const bool degenerate = rho2 == 0.0;                  // h == n: phi_h = atan2(0,0) = 0
    py = degenerate ? in.y : py;
    px = degenerate ? in.x : px;
    const double t = atan2_q1(__builtin_fabs(py), __builtin_fabs(px));
    const double pd = ((px < 0.0) != (py < 0.0)) ? kPi - t : t;   // atan2(py,px) folded into [0,pi]
    Coords c;
    c.xh = sqrt_fast(th * k_th);                          // k_th = n_th^2 / (pi/2)
    c.xd = td * k_td;                                     // k_td = n_td / (pi/2)
    c.xp = pd * k_pd;                                     // k_pd = n_pd / pi
    return c;
}

// the standard parameterisations (merl_device.hpp::standard_coords): theta = atan2(|v_xy|, v_z) for both directions,
// dphi = atan2(cross_z, dot_xy); all three axes linear.  k_0 = n_0 / (pi/2), k_1 = n_1 / (pi/2), k_2 = n_2 / pi (mirrored)
// or n_2 / 2pi (full).  Directions below the horizon are discarded by the caller; |z| keeps atan2_q1 in its domain.
MRL_HD Coords coords_standard(const Vec3 &in, const Dir &out_dir, bool full, double k_0, double k_1, double k_2)
{
#pragma clang fp contract(off)
    const Vec3 out = unit(out_dir);
    const double ti = atan2_q1(sqrt_fast(__builtin_fma(in.x, in.x, in.y * in.y)), __builtin_fabs(in.z));
    const double to = atan2_q1(sqrt_fast(__builtin_fma(out.x, out.x, out.y * out.y)), __builtin_fabs(out.z));
    const double cr = __builtin_fma(in.x, out.y, -(in.y * out.x));
    const double dt = __builtin_fma(in.x, out.x, in.y * out.y);
    const double t = (cr == 0.0 && dt == 0.0) ? 0.0 : atan2_q1(__builtin_fabs(cr), __builtin_fabs(dt));   // atan2(0,0) = 0
    const double ap = dt < 0.0 ? kPi - t : t;                      // |dphi| in [0, pi]
    Coords c;
    c.xh = ti * k_0;
    c.xd = to * k_1;
    c.xp = ((full && cr < 0.0) ? 2.0 * kPi - ap : ap) * k_2;
    return c;
}

// per-material constants of the coordinate maps
struct TableMaps {
    double k_th, k_td, k_pd;
    int param;
    MRL_HD explicit TableMaps(const MaterialDev &m)
        : k_th((double)m.n_th * (m.param == PARAM_HALF_DIFF ? (double)m.n_th : 1.0) / kHalfPi), k_td((double)m.n_td / kHalfPi),
          k_pd((double)m.n_pd / (m.param == PARAM_STANDARD_FULL ? 2.0 * kPi : kPi)), param(m.param) {}
    MRL_HD TableMaps(int n_th, int n_td, int n_pd, int prm)
        : k_th((double)n_th * (prm == PARAM_HALF_DIFF ? (double)n_th : 1.0) / kHalfPi), k_td((double)n_td / kHalfPi),
          k_pd((double)n_pd / (prm == PARAM_STANDARD_FULL ? 2.0 * kPi : kPi)), param(prm) {}
    // a2 + a3 under the material's parameterisation (wave-uniform branch for a single-material launch)
    MRL_HD Coords operator()(const Vec3 &in, const Dir &out) const
    {
        return param == PARAM_HALF_DIFF ? coords(in, out, k_th, k_td, k_pd)
                                        : coords_standard(in, out, param == PARAM_STANDARD_FULL, k_th, k_td, k_pd);
    }
};

// BRDF value (no cosine).  LOOKUP and LAYOUT are compile-time so that the eval lookup and the
// sample lookup of a unit stay in ONE basic block: their gathers are then all in flight together
// (a wave-uniform runtime branch here halves the memory-level parallelism and doubles the time).
// POLICY: the lookup follows MRL_OPT_NEGATIVE's renormalising blend when the option asks for it (one-unit callers); false: the
// texels are blended as stored and no branch enters the block (k_table — the batch calls send a renormalising context to the
// generic kernel or the LDS-DMA kernel instead)
template <int LOOKUP, int LAYOUT, bool POLICY = false>
MRL_HD Rgbf table_brdf(const MaterialDev &m, const Options &o, const Vec3 &in, const Dir &out)
{
    const TableMaps k(m);
    const Coords c = k(in, out);
    const int blend = POLICY ? blend_of(o) : (int)BLEND_STORED;
    if constexpr (LOOKUP) return lookup_trilinear_t<LAYOUT>(m, c, o.node, blend);
    else return lookup_nearest_t<LAYOUT>(m, c, blend);
}

// a5: eval (cosine included); valid == false gives zeros
template <int LOOKUP, int LAYOUT, bool POLICY = false>
MRL_HD void unit_eval(const MaterialDev &m, const Options &o, const Vec3 &in,
                                          float wix, float wiy, float wiz, float wox, float woy, float woz, float rgb[3])
{
    const Rgbf v = table_brdf<LOOKUP, LAYOUT, POLICY>(m, o, in, dir_f32(wox, woy, woz));
    eval_tail(v, (wix + wiy + wiz), wiz, wox, woy, woz, rgb, o.cosine != 0);
}

// ---- table importance sampling, tuned forms of merl_device.hpp::table_pdf / table_sample_dir ----
MRL_HD void sincos_2pi(double u, double &s, double &c);      // merl_ggx_fast.hpp

MRL_HD double table_pdf(const MaterialDev &m, const Vec3 &in, const Vec3 &out, float woz, int mode)
{
#pragma clang fp contract(off)
    const SamplingRow r = sampling_row(m, mode, in.z);
    double hx = in.x + out.x, hy = in.y + out.y, hz = in.z + out.z;
    double hs, hrs;
    sqrt_rsqrt(__builtin_fma(hx, hx, __builtin_fma(hy, hy, hz * hz)), hs, hrs);
    hx *= hrs; hy *= hrs; hz *= hrs;
    const int i = bin_of(r.s, r.n, __builtin_fma(hx, hx, hy * hy));
    const double ih = __builtin_fma(in.x, hx, __builtin_fma(in.y, hy, in.z * hz));
    const double ph = r.c[i] * hz * 0.25 * rcp_nr(__builtin_fmax(ih, kTiny));
    return (double)r.alpha * ((double)woz * 0.31830988618379067154) + (1.0 - (double)r.alpha) * ph;
}

MRL_HD void table_sample_dir(const MaterialDev &m, int disk_map, const Vec3 &in, float u0, float u1,
                                                 float &x, float &y, float &z, int mode)
{
    const SamplingRow r = sampling_row(m, mode, in.z);
    if (u0 < r.alpha) {
        square_to_cosine_hemisphere(disk_map, u0 * (1.0f / r.alpha), u1, x, y, z);
        return;
    }
    const double t = (double)(u0 - r.alpha) * (1.0 / (1.0 - (double)r.alpha));
    const int i = bin_of(r.cdf, r.n, t);
    const double c0 = r.cdf[i], s0 = r.s[i];
    const double xi = (t - c0) * rcp_nr(r.cdf[i + 1] - c0);
    const double sin2 = __builtin_fma(xi, r.s[i + 1] - s0, s0);
    const double ct = sqrt_fast(__builtin_fmax(1.0 - sin2, 0.0)), st = sqrt_fast(sin2);
    double sp, cp;
    sincos_2pi((double)u1, sp, cp);
    const double hx = st * cp, hy = st * sp;
    const double c2 = 2.0 * __builtin_fma(in.x, hx, __builtin_fma(in.y, hy, in.z * ct));
    x = (float)__builtin_fma(c2, hx, -in.x); y = (float)__builtin_fma(c2, hy, -in.y); z = (float)__builtin_fma(c2, ct, -in.z);
}

// a6: sample
template <int LOOKUP, int LAYOUT, bool POLICY = false>
MRL_HD void unit_sample(const MaterialDev &m, const Options &o, const Vec3 &in,
                                            float wix, float wiy, float wiz,
                                            float u0, float u1, float wo[3], float &pdf, float weight[3])
{
    float x, y, z, p;
    if (o.sampling) {                                       // wave-uniform
        table_sample_dir(m, o.disk_map, in, u0, u1, x, y, z, o.sampling);
        const bool up = z > 0.0f;
        if (!up) { x = 0.0f; y = 0.0f; z = 1.0f; }          // rejected: evaluate a harmless direction, report zeros
        p = up ? (float)table_pdf(m, in, normalize_f32(x, y, z), z, o.sampling) : 0.0f;
    } else {
        square_to_cosine_hemisphere(o.disk_map, u0, u1, x, y, z);
        p = z > 0.0f ? z * kInvPiF : 0.0f;
    }
    const Rgbf v = table_brdf<LOOKUP, LAYOUT, POLICY>(m, o, in, dir_f32(x, y, z));
    sample_tail(v, (wix + wiy + wiz), wiz, x, y, z, p, o.sampling != 0, wo, pdf, weight, o.cosine != 0);
}

} // namespace fast
} // namespace mrl
