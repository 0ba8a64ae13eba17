// merl_ggx_fast.hpp — tuned per-lane GGX rough conductor (row a9; BASELINE config 3).
//
// Same real-valued functions as merl_device.hpp's ggx_* (which mirror the oracle formula for
// formula), re-expressed without libm calls: every division / sqrt goes through the v_rcp_f64 /
// v_rsq_f64 + Newton helpers of merl_table_fast.hpp, and the visible-normal sampler never forms an
// angle — with s the stretched unit incident direction, tan(theta) = |s_xy| / s_z and
// (cos phi, sin phi) = s_xy / |s_xy|, where the generic path calls acos, atan2, tan, cos and sin.
#pragma once
#include "merl_table_fast.hpp"

namespace mrl {
namespace fast {

struct GgxConsts {                 // per material, computed once per launch (SGPRs for a single-material launch)
    double alpha, inv_alpha2, inv_pi_alpha2;
    double eta2_k2[3];             // eta^2 - k^2
    double four_k2_eta2[3];        // 4 k^2 eta^2
    __device__ __forceinline__ explicit GgxConsts(const MaterialDev &m)
    {
        alpha = m.alpha;
        inv_alpha2 = rcp_nr(m.alpha * m.alpha);
        inv_pi_alpha2 = inv_alpha2 * 0.31830988618379067154;
        for (int c = 0; c < 3; ++c) {
            eta2_k2[c] = m.eta[c] * m.eta[c] - m.k[c] * m.k[c];
            four_k2_eta2[c] = 4.0 * m.k[c] * m.k[c] * m.eta[c] * m.eta[c];
        }
    }
};

// D = 1 / (pi a^2 (cos^2 + sin^2/a^2)^2): the oracle's (1 + tan^2/a^2) cos^2 with the division folded away
__device__ __forceinline__ double ggx_D(const GgxConsts &g, const Vec3 &m)
{
    const double root = __builtin_fma(__builtin_fma(m.x, m.x, m.y * m.y), g.inv_alpha2, m.z * m.z);
    const double r = g.inv_pi_alpha2 * rcp_nr(__builtin_fmax(root * root, kTiny));
    return (m.z <= 0.0 || r * m.z < 1e-20) ? 0.0 : r;
}

// G1 = 2 / (1 + sqrt(1 + a^2 tan^2)) = 2 |vz| / (|vz| + sqrt(vz^2 + a^2 (1 - vz^2))): one sqrt, one reciprocal
__device__ __forceinline__ double ggx_G1(const GgxConsts &g, const Vec3 &v, const Vec3 &m)
{
    const double vm = __builtin_fma(v.x, m.x, __builtin_fma(v.y, m.y, v.z * m.z));
    const double vz2 = v.z * v.z;
    const double s2 = 1.0 - vz2;
    const double az = __builtin_fabs(v.z);
    const double r = 2.0 * az * rcp_nr(__builtin_fmax(az + sqrt_fast(__builtin_fma(g.alpha * g.alpha, s2, vz2)), kTiny));
    const double res = s2 <= 0.0 ? 1.0 : r;
    return vm * v.z <= 0.0 ? 0.0 : res;
}

__device__ __forceinline__ double fresnel_conductor(const GgxConsts &g, int ch, double c)
{
    const double c2 = c * c, s2 = 1.0 - c2, s4 = s2 * s2;
    const double t1 = g.eta2_k2[ch] - s2;
    const double a2pb2 = sqrt_fast(__builtin_fma(t1, t1, g.four_k2_eta2[ch]));
    const double a = sqrt_fast(0.5 * (a2pb2 + t1));
    const double term1 = a2pb2 + c2, term2 = 2.0 * a * c;
    const double term3 = __builtin_fma(a2pb2, c2, s4), term4 = term2 * s2;
    // Rs = (t1-t2)/(t1+t2), Rp = Rs (t3-t4)/(t3+t4): one reciprocal of the product of both denominators
    const double d12 = term1 + term2, d34 = term3 + term4;
    const double inv = rcp_nr(d12 * d34);
    const double rs2 = (term1 - term2) * d34 * inv;
    const double rp2 = rs2 * (term3 - term4) * d12 * inv;
    return 0.5 * (rp2 + rs2);
}

__device__ __forceinline__ Vec3 unit_sum(const Vec3 &a, const Vec3 &b)
{
    const double x = a.x + b.x, y = a.y + b.y, z = a.z + b.z;
    double s, rs;
    sqrt_rsqrt(__builtin_fma(x, x, __builtin_fma(y, y, z * z)), s, rs);
    return { x * rs, y * rs, z * rs };
}

// eval = F D G / (4 cos theta_i) (cosine of wo folded in), and pdf = D G1(wi) / (4 cos theta_i)
__device__ __forceinline__ void ggx_eval_pdf(const GgxConsts &g, const Vec3 &in, const Vec3 &out, double rgb[3], double &pdf)
{
    const Vec3 m = unit_sum(in, out);
    const double D = ggx_D(g, m);
    const double G1i = ggx_G1(g, in, m);
    const double quarter_inv_cos = 0.25 * rcp_nr(in.z);
    pdf = D * G1i * quarter_inv_cos;
    const double model = pdf * ggx_G1(g, out, m);
    const double c = __builtin_fma(in.x, m.x, __builtin_fma(in.y, m.y, in.z * m.z));
    rgb[0] = D == 0.0 ? 0.0 : fresnel_conductor(g, 0, c) * model;
    rgb[1] = D == 0.0 ? 0.0 : fresnel_conductor(g, 1, c) * model;
    rgb[2] = D == 0.0 ? 0.0 : fresnel_conductor(g, 2, c) * model;
}

// sin / cos of 2 pi u for u in [0,1): only the (rare) normal-incidence branch of the sampler needs it
MRL_HD void sincos_2pi(double u, double &s, double &c)
{   // declared in merl_table_fast.hpp
    // octant reduction: 2 pi u = q pi/2 + t, |t| <= pi/4
    const double x = 4.0 * u;
    const double q = __builtin_rint(x);
    const double t = (x - q) * kHalfPi;
    const double z = t * t;
    double ps = -2.5052108385441720e-08;           // Taylor coefficients: |t| <= pi/4 gives < 1e-13 abs error
    ps = __builtin_fma(ps, z, 2.7557319223985893e-06);
    ps = __builtin_fma(ps, z, -1.9841269841269841e-04);
    ps = __builtin_fma(ps, z, 8.3333333333333332e-03);
    ps = __builtin_fma(ps, z, -1.6666666666666666e-01);
    const double st = __builtin_fma(ps * z, t, t);
    double pc = 2.0876756987868099e-09;
    pc = __builtin_fma(pc, z, -2.7557319223985888e-07);
    pc = __builtin_fma(pc, z, 2.4801587301587302e-05);
    pc = __builtin_fma(pc, z, -1.3888888888888889e-03);
    pc = __builtin_fma(pc, z, 4.1666666666666664e-02);
    pc = __builtin_fma(pc, z, -0.5);
    const double ct = __builtin_fma(pc, z, 1.0);
    const int qi = (int)q & 3;
    s = qi == 0 ? st : (qi == 1 ? ct : (qi == 2 ? -st : -ct));
    c = qi == 0 ? ct : (qi == 1 ? -st : (qi == 2 ? -ct : st));
}

// visible-normal sampling (Heitz & d'Eon 2014); returns false when the sample is rejected
__device__ __forceinline__ bool ggx_sample(const GgxConsts &g, const Vec3 &in, float u0, float u1,
                                           float wo[3], float &pdf, float weight[3])
{
    const double al = g.alpha;
    // 1. stretch
    double sx = al * in.x, sy = al * in.y, sz = in.z;
    double sl, srs;
    sqrt_rsqrt(__builtin_fma(sx, sx, __builtin_fma(sy, sy, sz * sz)), sl, srs);
    sx *= srs; sy *= srs; sz *= srs;
    double slx, sly, cp = 1.0, sp = 0.0;
    double u2 = (double)u1;
    if (sz < 0.99999) {
        // 2. P22 slopes for alpha = 1; tan(theta) and (cos phi, sin phi) straight from the stretched vector
        double rho, rrho;
        sqrt_rsqrt(__builtin_fma(sx, sx, sy * sy), rho, rrho);
        cp = sx * rrho; sp = sy * rrho;
        const double inv_tan = sz * rrho;
        const double tan_i = rho * rcp_nr(sz);
        const double G1 = 2.0 * rcp_nr(1.0 + sqrt_fast(__builtin_fma(tan_i, tan_i, 1.0)));
        double A = __builtin_fma(2.0 * (double)u0, rcp_nr(G1), -1.0);
        if (__builtin_fabs(A) == 1.0) A -= (A > 0 ? 1.0 : -1.0) * 1e-12;
        const double tmp = rcp_nr(__builtin_fma(A, A, -1.0));
        const double B = tan_i;
        const double disc = B * B * tmp * tmp - (A * A - B * B) * tmp;
        const double D = disc > 0.0 ? sqrt_fast(disc) : 0.0;
        const double s1 = B * tmp - D, s2 = B * tmp + D;
        slx = (A < 0.0 || s2 > inv_tan) ? s1 : s2;
        double S;
        if (u2 > 0.5) { S = 1.0; u2 = 2.0 * (u2 - 0.5); }
        else { S = -1.0; u2 = 2.0 * (0.5 - u2); }
        const double num = u2 * (u2 * (u2 * (-0.365728915865723) + 0.790235037209296) - 0.424965825137544) + 0.000152998850436920;
        const double den = u2 * (u2 * (u2 * (u2 * 0.169507819808272 - 0.397203533833404) - 0.232500544458471) + 1.0) - 0.539825872510702;
        sly = S * num * rcp_nr(den) * sqrt_fast(__builtin_fma(slx, slx, 1.0));
    } else {
        // normal incidence: theta = phi = 0
        const double q = (double)u0 * rcp_nr(1.0 - (double)u0);
        const double r = q > 0.0 ? sqrt_fast(q) : 0.0;
        double s2pi, c2pi;
        sincos_2pi(u2, s2pi, c2pi);
        slx = r * c2pi; sly = r * s2pi;
    }
    // 3. rotate, 4. unstretch, 5. normal
    const double mx = (cp * slx - sp * sly) * al, my = (sp * slx + cp * sly) * al;
    double nl, nrm;
    sqrt_rsqrt(__builtin_fma(mx, mx, __builtin_fma(my, my, 1.0)), nl, nrm);
    const Vec3 m = { -mx * nrm, -my * nrm, nrm };
    const double c = __builtin_fma(in.x, m.x, __builtin_fma(in.y, m.y, in.z * m.z));
    const Vec3 out = { __builtin_fma(2.0 * c, m.x, -in.x), __builtin_fma(2.0 * c, m.y, -in.y), __builtin_fma(2.0 * c, m.z, -in.z) };
    const double D = ggx_D(g, m);
    const double p = D * ggx_G1(g, in, m) * 0.25 * rcp_nr(in.z);
    const float wx = (float)out.x, wy = (float)out.y, wz = (float)out.z;
    const bool ok = (out.z > 0.0) && (c > 0.0) && (p > 0.0) && (wz > 0.0f);
    const double G1o = ggx_G1(g, out, m);
    wo[0] = ok ? wx : 0.0f; wo[1] = ok ? wy : 0.0f; wo[2] = ok ? wz : 0.0f;
    pdf = ok ? (float)p : 0.0f;
    weight[0] = ok ? (float)(fresnel_conductor(g, 0, c) * G1o) : 0.0f;
    weight[1] = ok ? (float)(fresnel_conductor(g, 1, c) * G1o) : 0.0f;
    weight[2] = ok ? (float)(fresnel_conductor(g, 2, c) * G1o) : 0.0f;
    return ok;
}

} // namespace fast
} // namespace mrl
