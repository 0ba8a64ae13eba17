// merl_device.hpp — per-lane device math of the MERL / customized_measurement / GGX hot path.
//
// Reference rows (reference files absent, /root/reference/README.md:1 names them):
//   a2 half/diff transform, a3 index maps, a4 table fetch, a5 eval, a6 sample, a7 pdf, a9 GGX.
//
// Numerics.  Inputs/outputs are f32 (Mitsuba Float).  The transform runs in f64 VALU: the table
// coordinate x in [0,180) must be good to ~1e-8 texel for the interpolated value to match an
// f64 CPU evaluation to 1e-6 relative on high-contrast tables, which f32 (ulp(90) = 7.6e-6)
// cannot give.  Angles come from atan2 forms that are well conditioned for unit
// vectors s = in+out, e = in-out:
//     theta_h = atan2(|s_xy|, s_z)      theta_d = atan2(|e|, |s|)
//     phi_d   = atan2(e_y s_x - e_x s_y, -e_z |s|)
// (algebraically the Rodrigues form of BRDFRead; no sin/cos, no acos near 1).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Every per-unit function below is compiled for the device AND for the host: the host build is the product's own
// one-unit path (merl_host_scalar.hip: the scalar virtual BSDF::eval / sample / pdf of a per-ray integrator evaluates
// on the calling CPU thread, SURVEY.md §8b "what calls it (2)") — one formulation, two targets.  Only the hardware
// reciprocal / reciprocal-square-root seeds differ (v_rcp_f64 / v_rsq_f64 on the device, an exact quotient on the
// host); both are followed by the same Newton steps, so the two builds agree to ~1e-15 before the outputs are rounded.
#define MRL_HD __host__ __device__ __forceinline__

namespace mrl {

MRL_HD double rcp_seed(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(x);
#else
    return 1.0 / x;
#endif
}
MRL_HD double rsq_seed(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsq(x);
#else
    return 1.0 / __builtin_sqrt(x);
#endif
}
// double -> int, truncating.  The device's v_cvt_i32_f64 saturates and maps NaN to 0, and the index maps below lean on that
// (garbage coordinates of masked lanes must still address the table); x86's cvttsd2si returns INT_MIN for both, so the host
// build spells the same semantics out.
MRL_HD int trunc_i(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)x;
#else
    if (!(x == x)) return 0;
    if (x >= 2147483647.0) return 2147483647;
    if (x <= -2147483648.0) return -2147483647 - 1;
    return (int)x;
#endif
}
MRL_HD int min_i(int a, int b) { return a < b ? a : b; }
MRL_HD unsigned min_u(unsigned a, unsigned b) { return a < b ? a : b; }

constexpr double kPi = 3.14159265358979323846;
constexpr double kHalfPi = 1.57079632679489661923;
constexpr float kInvPiF = 0.31830988618379067154f;

enum Kind : int { KIND_MERL = 0, KIND_TABLE = 1, KIND_GGX = 2,
                  KIND_RELEASED = 3,     // tombstone of mrl_material_release: a valid 1x1x1 zero table, treated like an unknown id
                  KIND_TABLE_NCH = 4,    // n-channel table (merl_nch.hip): only the *_nch entry points evaluate it
                  KIND_RGL = 5,          // adaptive-parameterisation measured BSDF (merl_rgl.hip): its own kernel, also behind mixed batches
                  KIND_RGL_SPECTRAL = 6 };   // the same from a spectral file: W values per unit at caller-supplied wavelengths (mrl_*_spectral_batch)
// the RGB kernels evaluate kinds 0..2; anything above renders as an unknown id (every output zero)
__host__ __device__ constexpr bool kind_is_rgb_path(int kind) { return kind >= KIND_MERL && kind <= KIND_GGX; }
enum Layout : int { LAYOUT_ROWS = 0, LAYOUT_BRICK = 1 };
// table parameterisation (include/merl_hip.h MRL_PARAM_*): which three angles index the table
enum Param : int { PARAM_HALF_DIFF = 0,        // (theta_h sqrt-warped, theta_d, phi_d mod pi): MERL
                   PARAM_STANDARD = 1,         // (theta_i, theta_o, |phi_o - phi_i| in [0,pi]): linear axes, azimuth clamped
                   PARAM_STANDARD_FULL = 2 };  // (theta_i, theta_o, phi_o - phi_i in [0,2pi)): linear axes, azimuth periodic
__host__ __device__ constexpr bool param_phi_periodic(int param) { return param != PARAM_STANDARD; }

// One material as the kernels see it (array in device memory; single-material launches get it
// by value, i.e. in SGPRs).
struct MaterialDev {
    int kind;
    int n_th, n_td, n_pd;        // logical dims
    int row_td;                  // texels per theta_d row      = n_pd + 1 (phi wrap texel appended)
    int row_th;                  // texels per theta_h slab     = (n_td + 1) * (n_pd + 1)
    const float4 *texels;        // layout 0: [(n_th+1)][(n_td+1)][(n_pd+1)] RGBA f32, scaled, negatives clamped
                                 // layout 1: [n_th][n_td][n_pd] bricks of 128 B = the cell's 8 corners, RGB f32 packed
    int layout;                  // LAYOUT_ROWS / LAYOUT_BRICK
    int n_ch;                    // channels: 3 for the RGB kinds; KIND_TABLE_NCH: 1..32 (bricks of ceil(n_ch/4) x 128 B, or 32 / 64 B for 1 / 2 channels)
    const double *sampling;      // table importance sampling: s[n_th+1] | cdf[n_th+1] | c[n_th]  (see table_pdf below)
    const double *sampling2d;    // conditional rows P(theta_h | theta_i): [n_ti][ cdf[n_th+1] | c[n_th] ], nullptr = none (see SamplingRow)
    int n_ti;                    // incident bins of sampling2d (uniform in cos theta_i)
    int param;                   // PARAM_*: the axes are (n_th, n_td, n_pd) whatever they mean
    const void *rgl;             // KIND_RGL: the material's RglDev (merl_rgl.hpp), stored behind its image in device memory
    double alpha;                // GGX
    double eta[3], k[3];
};

struct Options {
    int lookup;    // 0 nearest, 1 trilinear
    int node;      // 0 integer node, 1 texel centre
    int disk_map;  // 0 Mitsuba 0.6, 1 Mitsuba 3
    int sampling;  // 0 cosine hemisphere (upstream convention), 1 table importance sampling
    int cosine;    // MRL_OPT_COSINE_FACTOR: 0 eval() = f cos(theta_o) (upstream convention), 1 eval() = f
    int negative;  // MRL_OPT_NEGATIVE: what a negative stored value does to a lookup — 0 clamped to 0 when the table is built,
                   // 1 kept as stored, 2 left out and the valid corners renormalised
};
enum Negative : int { NEGATIVE_CLAMP = 0, NEGATIVE_KEEP = 1, NEGATIVE_RENORMALISE = 2 };
// how a lookup treats the texels it blends: as stored (clamped at build time under NEGATIVE_CLAMP, raw otherwise); clamped at
// lookup time (the sampling-table builders on a raw table: a density needs a non-negative mass); valid corners only
enum Blend : int { BLEND_STORED = 0, BLEND_CLAMP = 1, BLEND_RENORMALISE = 2 };
MRL_HD int blend_of(const Options &o) { return o.negative == NEGATIVE_RENORMALISE ? BLEND_RENORMALISE : BLEND_STORED; }

struct Vec3d { double x, y, z; };

MRL_HD Vec3d normalized(float x, float y, float z)
{
    double dx = x, dy = y, dz = z;
    double inv = 1.0 / sqrt(dx * dx + dy * dy + dz * dz);
    return { dx * inv, dy * inv, dz * inv };
}

// ---- a2 + a3: unit in/out -> continuous table coordinates ---------------------------------
struct Coords { double xh, xd, xp; };

MRL_HD Coords half_diff_coords(const Vec3d &in, const Vec3d &out, int n_th, int n_td, int n_pd)
{
    const double sx = in.x + out.x, sy = in.y + out.y, sz = in.z + out.z;
    const double ex = in.x - out.x, ey = in.y - out.y, ez = in.z - out.z;
    const double rho2 = sx * sx + sy * sy;
    const double ns = sqrt(rho2 + sz * sz);
    const double ne = sqrt(ex * ex + ey * ey + ez * ez);
    const double rho = sqrt(rho2);
    const double th = atan2(rho, sz);
    const double td = atan2(ne, ns);
    double py = ey * sx - ex * sy;
    double px = -ez * ns;
    if (rho == 0.0) { py = in.y; px = in.x; }       // h == n: phi_h = atan2(0,0) = 0
    double pd = atan2(py, px);
    if (pd < 0.0) pd += kPi;                          // reciprocity fold
    Coords c;
    c.xh = th <= 0.0 ? 0.0 : sqrt((th / kHalfPi) * n_th * n_th);
    c.xd = td / kHalfPi * n_td;
    c.xp = pd / kPi * n_pd;
    return c;
}

// the standard parameterisations: polar angles of both directions and their azimuth difference, linear axes
MRL_HD Coords standard_coords(const Vec3d &in, const Vec3d &out, int param, int n_0, int n_1, int n_2)
{
    const double ti = atan2(sqrt(in.x * in.x + in.y * in.y), in.z);
    const double to = atan2(sqrt(out.x * out.x + out.y * out.y), out.z);
    const double cr = in.x * out.y - in.y * out.x, dt = in.x * out.x + in.y * out.y;
    const double dp = (cr == 0.0 && dt == 0.0) ? 0.0 : atan2(cr, dt);   // a direction AT the normal has no azimuth: 0
    Coords c;
    c.xh = ti / kHalfPi * n_0;
    c.xd = to / kHalfPi * n_1;
    if (param == PARAM_STANDARD) c.xp = fabs(dp) / kPi * n_2;
    else c.xp = (dp < 0.0 ? dp + 2.0 * kPi : dp) / (2.0 * kPi) * n_2;
    return c;
}

// ---- a4: table fetch -------------------------------------------------------------------------
MRL_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct Rgbd { double r, g, b; };
struct Rgbf { float r, g, b; };       // a looked-up BRDF value: Float, like the texels it is blended from

// The 8 trilinear corner weights, formed in f64 from the f64 fractions (a fraction near 1 would lose its complement's
// relative accuracy in Float) and rounded ONCE to Float; corner k = 4 a + 2 b + c along (axis 0, axis 1, azimuth).
struct CornerWeights { float w[8]; };
MRL_HD CornerWeights corner_weights(double fh, double fd, double fp)
{
#pragma clang fp contract(off)
    const double gh = 1.0 - fh, gd = 1.0 - fd, gp = 1.0 - fp;
    const double a00 = gh * gd, a01 = gh * fd, a10 = fh * gd, a11 = fh * fd;
    CornerWeights c;
    c.w[0] = (float)(a00 * gp); c.w[1] = (float)(a00 * fp); c.w[2] = (float)(a01 * gp); c.w[3] = (float)(a01 * fp);
    c.w[4] = (float)(a10 * gp); c.w[5] = (float)(a10 * fp); c.w[6] = (float)(a11 * gp); c.w[7] = (float)(a11 * fp);
    return c;
}

// Blend of one 96-B brick (8 corners x RGB f32, corner-major: float 3 k + ch) in packed Float math.  The brick's 24
// floats are 12 aligned register pairs that cycle through (r,g) (b,r) (g,b): three pair accumulators A, B, C take four
// v_pk_fma_f32 each — 12 packed FMAs instead of 24 f64 FMAs and 24 f32 -> f64 converts — and r = A.x + B.y, g = A.y + C.x,
// b = B.x + C.y.  Texels and weights are non-negative, so nothing cancels: the result is within 6 roundings
// (3.6e-7 relative; measured worst 2.8e-7 incl. the oracle's own rounding) of the exact blend of the same texels.
typedef float v2f_t __attribute__((ext_vector_type(2)));
MRL_HD Rgbf blend_brick(const float4 &q0, const float4 &q1, const float4 &q2, const float4 &q3, const float4 &q4,
                                            const float4 &q5, const CornerWeights &c)
{
    // contraction off: every operation below is exactly what is written (entry points must agree bit for bit, and a
    // multiply left to the compiler's fusing would round differently from one inlining context to the next)
#pragma clang fp contract(off)
    const float *w = c.w;
    v2f_t A = v2f_t{w[0], w[0]} * v2f_t{q0.x, q0.y};
    v2f_t B = v2f_t{w[0], w[1]} * v2f_t{q0.z, q0.w};
    v2f_t C = v2f_t{w[1], w[1]} * v2f_t{q1.x, q1.y};
    A = __builtin_elementwise_fma(v2f_t{w[2], w[2]}, v2f_t{q1.z, q1.w}, A);
    B = __builtin_elementwise_fma(v2f_t{w[2], w[3]}, v2f_t{q2.x, q2.y}, B);
    C = __builtin_elementwise_fma(v2f_t{w[3], w[3]}, v2f_t{q2.z, q2.w}, C);
    A = __builtin_elementwise_fma(v2f_t{w[4], w[4]}, v2f_t{q3.x, q3.y}, A);
    B = __builtin_elementwise_fma(v2f_t{w[4], w[5]}, v2f_t{q3.z, q3.w}, B);
    C = __builtin_elementwise_fma(v2f_t{w[5], w[5]}, v2f_t{q4.x, q4.y}, C);
    A = __builtin_elementwise_fma(v2f_t{w[6], w[6]}, v2f_t{q4.z, q4.w}, A);
    B = __builtin_elementwise_fma(v2f_t{w[6], w[7]}, v2f_t{q5.x, q5.y}, B);
    C = __builtin_elementwise_fma(v2f_t{w[7], w[7]}, v2f_t{q5.z, q5.w}, C);
    return { A.x + B.y, A.y + C.x, B.x + C.y };
}

// MRL_OPT_NEGATIVE = 2 ("skip and renormalise"): a negative texel is a sample that was not measured; the
// lookup blends the valid corners only and divides by their weight, per channel — sum_k w_k v_k [v_k >= 0] / sum_k w_k [v_k >= 0],
// 0 when no corner is valid.  In f64 on the Float corner weights (the quotient of two eight-term Float sums would spend the whole
// 1e-6 margin); every entry point runs this one function.
MRL_HD Rgbf blend_brick_valid(const float4 &q0, const float4 &q1, const float4 &q2, const float4 &q3, const float4 &q4,
                              const float4 &q5, const CornerWeights &c)
{
#pragma clang fp contract(off)
    const float f[24] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w,
                          q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w };
    double num[3] = { 0.0, 0.0, 0.0 }, den[3] = { 0.0, 0.0, 0.0 };
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double w = (double)c.w[k];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float v = f[3 * k + ch];
            const bool ok = v >= 0.0f;
            num[ch] = __builtin_fma(ok ? w : 0.0, (double)v, num[ch]);
            den[ch] += ok ? w : 0.0;
        }
    }
    return { den[0] > 0.0 ? (float)(num[0] / den[0]) : 0.0f, den[1] > 0.0 ? (float)(num[1] / den[1]) : 0.0f,
             den[2] > 0.0 ? (float)(num[2] / den[2]) : 0.0f };
}

MRL_HD float4 max0(const float4 &q) { return make_float4(__builtin_fmaxf(q.x, 0.0f), __builtin_fmaxf(q.y, 0.0f), __builtin_fmaxf(q.z, 0.0f), __builtin_fmaxf(q.w, 0.0f)); }

// the blend of one brick under a policy (enum Blend; wave-uniform)
MRL_HD Rgbf blend_brick_as(int blend, const float4 &q0, const float4 &q1, const float4 &q2, const float4 &q3, const float4 &q4,
                           const float4 &q5, const CornerWeights &c)
{
    if (blend == BLEND_RENORMALISE) return blend_brick_valid(q0, q1, q2, q3, q4, q5, c);
    if (blend == BLEND_CLAMP) return blend_brick(max0(q0), max0(q1), max0(q2), max0(q3), max0(q4), max0(q5), c);
    return blend_brick(q0, q1, q2, q3, q4, q5, c);
}

// blend: BLEND_STORED returns the texel as stored; the other two policies have nothing to blend and return it clamped at 0
template <int LAYOUT>
MRL_HD Rgbf lookup_nearest_t(const MaterialDev &m, const Coords &c, int blend = BLEND_STORED)
{
    int ih = clampi(trunc_i(c.xh), 0, m.n_th - 1);
    int id = clampi(trunc_i(c.xd), 0, m.n_td - 1);
    int ip = clampi(trunc_i(c.xp), 0, m.n_pd - 1);
    float4 t;
    if constexpr (LAYOUT == LAYOUT_BRICK) t = m.texels[(((size_t)ih * m.n_td + id) * m.n_pd + ip) * 8];     // corner 0 = the texel itself
    else t = m.texels[(size_t)ih * m.row_th + (size_t)id * m.row_td + ip];
    if (blend != BLEND_STORED) t = max0(t);
    return { t.x, t.y, t.z };
}

// Both splits take x in [-1, n] (every coordinate map of this file lands there: angles are atan2 results scaled by n / range,
// minus the half-texel shift of the centre-node convention) or NaN, and stay inside the table for anything else.
// clamped axis: i0 in [0,n-1], f in [0,1]; i0+1 is always a valid (padded) index.  (int)x truncates towards zero, which for
// x > -1 is floor(x) clamped at 0 (v_cvt_i32_f64 saturates and maps NaN to 0): no floor, no lower clamp.
// (contraction off in everything between a continuous coordinate and its Float corner weights: fused with the multiply
// that produced x in one inlining context and not in another, x - i differs by ~1e-14, and the weights' rounding to
// Float turns that into a one-ulp difference between entry points for about one unit in a million)
MRL_HD void split_clamped(double x, int n, int &i0, double &f)
{
#pragma clang fp contract(off)
    int i = min_i(trunc_i(x), n - 1);
#if !defined(__HIP_DEVICE_COMPILE__)
    i = i < 0 ? 0 : i;                         // x <= -1 only happens for masked garbage; the device's (x > -1) never needs it
#endif
    f = __builtin_fmin(__builtin_fmax(x - (double)i, 0.0), 1.0);
    i0 = i;
}
// periodic axis: i0 in [0,n-1]; i0+1 <= n hits the appended wrap texel.  floor(x) is in [-1, n]: shifted by n it lies in
// [n-1, 2n] and two unsigned min(j, j - n) steps bring it to [0, n-1] (j < n: j - n wraps around to a huge value and j wins).
MRL_HD void split_periodic(double x, int n, int &i0, double &f)
{
#pragma clang fp contract(off)
    const double fl = floor(x);
    f = x - fl;
    unsigned j = (unsigned)trunc_i(fl) + (unsigned)n;
    j = min_u(j, j - (unsigned)n);
    j = min_u(j, j - (unsigned)n);
    i0 = (int)min_u(j, (unsigned)(n - 1));          // only reached by x outside [-1, n]: stay inside the table
}

// the azimuth axis: periodic, except for the mirrored standard form (0 and pi are its two ends)
MRL_HD void split_phi(bool periodic, double x, int n, int &i0, double &f)
{
    int ip, ic; double fp, fc;
    split_periodic(x, n, ip, fp);
    split_clamped(x, n, ic, fc);
    i0 = periodic ? ip : ic;
    f = periodic ? fp : fc;
}

template <int LAYOUT>
MRL_HD Rgbf lookup_trilinear_t(const MaterialDev &m, const Coords &c, int node, int blend = BLEND_STORED)
{
#pragma clang fp contract(off)
    const double shift = node ? 0.5 : 0.0;
    int h0, d0, p0; double fh, fd, fp;
    split_clamped(c.xh - shift, m.n_th, h0, fh);
    split_clamped(c.xd - shift, m.n_td, d0, fd);
    split_phi(param_phi_periodic(m.param), c.xp - shift, m.n_pd, p0, fp);
    if constexpr (LAYOUT == LAYOUT_BRICK) {
        // one 128-B line holds the whole neighbourhood: 8 corners x RGB f32 = 96 B, six 16-B loads; the same blend as
        // k_table_dma's (entry points agree bit for bit)
        const float4 *q = m.texels + (((size_t)h0 * m.n_td + d0) * m.n_pd + p0) * 8;
        const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5];
        return blend_brick_as(blend, q0, q1, q2, q3, q4, q5, corner_weights(fh, fd, fp));
    }
    const float4 *b = m.texels + ((size_t)h0 * m.row_th + (size_t)d0 * m.row_td + p0);
    // issue all eight 16-B gathers before any use
    const float4 t000 = b[0],                 t001 = b[1];
    const float4 t010 = b[m.row_td],          t011 = b[m.row_td + 1];
    const float4 t100 = b[m.row_th],          t101 = b[m.row_th + 1];
    const float4 t110 = b[m.row_th + m.row_td], t111 = b[m.row_th + m.row_td + 1];
    // the eight texels in brick order (corner-major RGB) through the SAME blend: a rows-layout table, a brick table and the
    // host image of either give the same bits for the same coordinates
    return blend_brick_as(blend, make_float4(t000.x, t000.y, t000.z, t001.x), make_float4(t001.y, t001.z, t010.x, t010.y),
                          make_float4(t010.z, t011.x, t011.y, t011.z), make_float4(t100.x, t100.y, t100.z, t101.x),
                          make_float4(t101.y, t101.z, t110.x, t110.y), make_float4(t110.z, t111.x, t111.y, t111.z),
                          corner_weights(fh, fd, fp));
}

// runtime-layout wrappers (generic kernel)
MRL_HD Rgbf lookup_nearest(const MaterialDev &m, const Coords &c, int blend = BLEND_STORED)
{
    return m.layout == LAYOUT_BRICK ? lookup_nearest_t<LAYOUT_BRICK>(m, c, blend) : lookup_nearest_t<LAYOUT_ROWS>(m, c, blend);
}
MRL_HD Rgbf lookup_trilinear(const MaterialDev &m, const Coords &c, int node, int blend = BLEND_STORED)
{
    return m.layout == LAYOUT_BRICK ? lookup_trilinear_t<LAYOUT_BRICK>(m, c, node, blend) : lookup_trilinear_t<LAYOUT_ROWS>(m, c, node, blend);
}

// BRDF value (no cosine) of a table material for unit in/out
MRL_HD Rgbf table_brdf(const MaterialDev &m, const Options &o, const Vec3d &in, const Vec3d &out)
{
    Coords c = m.param == PARAM_HALF_DIFF ? half_diff_coords(in, out, m.n_th, m.n_td, m.n_pd)
                                          : standard_coords(in, out, m.param, m.n_th, m.n_td, m.n_pd);
    return o.lookup ? lookup_trilinear(m, c, o.node, blend_of(o)) : lookup_nearest(m, c, blend_of(o));
}

// ---- a6: cosine-hemisphere sampling, pinned f32 sequence (bit-identical to oracle/merl_oracle.c) --
MRL_HD void sincos_quarter_f32(float t, float &s, float &c)
{
#pragma clang fp contract(off)
    const float S0 = -0x1.555552p-3f, S1 = 0x1.110c28p-7f, S2 = -0x1.9ac98ep-13f;
    const float C0 = 0x1.555552p-5f, C1 = -0x1.6c10dp-10f, C2 = 0x1.9b31dep-16f;
    float z = t * t;
    float p = __builtin_fmaf(S2, z, S1); p = __builtin_fmaf(p, z, S0);
    float pz = p * z;
    s = __builtin_fmaf(pz, t, t);
    float q = __builtin_fmaf(C2, z, C1); q = __builtin_fmaf(q, z, C0);
    float qz = q * z;
    c = __builtin_fmaf(qz, z, __builtin_fmaf(-0.5f, z, 1.0f));
}

MRL_HD void square_to_cosine_hemisphere(int disk_map, float u0, float u1, float &x, float &y, float &z)
{
#pragma clang fp contract(off)
    const float QUARTER_PI = 0.78539816339744830962f;
    float a = 2.0f * u0 - 1.0f, b = 2.0f * u1 - 1.0f;
    {
        // one IEEE division for both branches of the map (num / den selected first); a == b == 0 divides 0 by 0 and is
        // replaced below — the same values, instruction for instruction, as the two-branch form of the oracle
        float aa = a * a, bb = b * b;
        bool first = disk_map ? !(fabsf(a) < fabsf(b)) : (aa > bb);
        float r = first ? a : b;
        float ratio = (first ? b : a) / r;
        float s, c;
        sincos_quarter_f32(QUARTER_PI * ratio, s, c);
        x = r * (first ? c : s);
        y = r * (first ? s : c);
        const bool origin = (a == 0.0f) && (b == 0.0f);
        x = origin ? 0.0f : x;
        y = origin ? 0.0f : y;
    }
    float xx = x * x;
    float zz = 1.0f - __builtin_fmaf(y, y, xx);
    z = zz > 0.0f ? __builtin_sqrtf(zz) : 0.0f;
    if (disk_map == 0 && z == 0.0f) z = 1e-10f;
}

// ---- a9: GGX rough conductor, f64, formula-for-formula the oracle's ---------
MRL_HD double ggx_D(double alpha, const Vec3d &m)
{
    if (m.z <= 0.0) return 0.0;
    double c2 = m.z * m.z;
    double e = (m.x * m.x + m.y * m.y) / (alpha * alpha) / c2;
    double root = (1.0 + e) * c2;
    double r = 1.0 / (kPi * alpha * alpha * root * root);
    return r * m.z < 1e-20 ? 0.0 : r;
}
MRL_HD double ggx_G1(double alpha, const Vec3d &v, const Vec3d &m)
{
    double vm = v.x * m.x + v.y * m.y + v.z * m.z;
    if (vm * v.z <= 0.0) return 0.0;
    double s2 = 1.0 - v.z * v.z;
    if (s2 <= 0.0) return 1.0;
    double tan2 = s2 / (v.z * v.z);
    return 2.0 / (1.0 + sqrt(1.0 + alpha * alpha * tan2));
}
MRL_HD double safe_sqrt(double x) { return x > 0.0 ? sqrt(x) : 0.0; }
MRL_HD double fresnel_conductor(double c, double eta, double k)
{
    double c2 = c * c, s2 = 1.0 - c2, s4 = s2 * s2;
    double t1 = eta * eta - k * k - s2;
    double a2pb2 = safe_sqrt(t1 * t1 + 4.0 * k * k * eta * eta);
    double a = safe_sqrt(0.5 * (a2pb2 + t1));
    double term1 = a2pb2 + c2, term2 = 2.0 * a * c;
    double rs2 = (term1 - term2) / (term1 + term2);
    double term3 = a2pb2 * c2 + s4, term4 = term2 * s2;
    double rp2 = rs2 * (term3 - term4) / (term3 + term4);
    return 0.5 * (rp2 + rs2);
}
MRL_HD Vec3d unit_sum(const Vec3d &a, const Vec3d &b)
{
    double x = a.x + b.x, y = a.y + b.y, z = a.z + b.z;
    double len = sqrt(x * x + y * y + z * z);
    if (len > 0.0) { x /= len; y /= len; z /= len; }
    return { x, y, z };
}
// eval with the cosine folded in: F D G / (4 cos ti)
MRL_HD Rgbd ggx_eval(const MaterialDev &g, const Vec3d &in, const Vec3d &out)
{
    Vec3d m = unit_sum(in, out);
    Rgbd o = { 0.0, 0.0, 0.0 };
    double D = ggx_D(g.alpha, m);
    if (D == 0.0) return o;
    double G = ggx_G1(g.alpha, in, m) * ggx_G1(g.alpha, out, m);
    double model = D * G / (4.0 * in.z);
    double c = in.x * m.x + in.y * m.y + in.z * m.z;
    o.r = fresnel_conductor(c, g.eta[0], g.k[0]) * model;
    o.g = fresnel_conductor(c, g.eta[1], g.k[1]) * model;
    o.b = fresnel_conductor(c, g.eta[2], g.k[2]) * model;
    return o;
}
MRL_HD double ggx_pdf(const MaterialDev &g, const Vec3d &in, const Vec3d &out)
{
    Vec3d m = unit_sum(in, out);
    return ggx_D(g.alpha, m) * ggx_G1(g.alpha, in, m) / (4.0 * in.z);
}
MRL_HD void ggx_sample_visible_11(double theta_i, double u1, double u2, double &slx, double &sly)
{
    if (theta_i < 1e-4) {
        double r = safe_sqrt(u1 / (1.0 - u1));
        double phi = 2.0 * kPi * u2;
        slx = r * cos(phi); sly = r * sin(phi);
        return;
    }
    double tan_i = tan(theta_i);
    double a = 1.0 / tan_i;
    double G1 = 2.0 / (1.0 + safe_sqrt(1.0 + 1.0 / (a * a)));
    double A = 2.0 * u1 / G1 - 1.0;
    if (fabs(A) == 1.0) A -= (A > 0 ? 1.0 : -1.0) * 1e-12;
    double tmp = 1.0 / (A * A - 1.0);
    double B = tan_i;
    double D = safe_sqrt(B * B * tmp * tmp - (A * A - B * B) * tmp);
    double s1 = B * tmp - D, s2 = B * tmp + D;
    slx = (A < 0.0 || s2 > 1.0 / tan_i) ? s1 : s2;
    double S;
    if (u2 > 0.5) { S = 1.0; u2 = 2.0 * (u2 - 0.5); }
    else { S = -1.0; u2 = 2.0 * (0.5 - u2); }
    double z = (u2 * (u2 * (u2 * (-0.365728915865723) + 0.790235037209296) - 0.424965825137544) + 0.000152998850436920)
             / (u2 * (u2 * (u2 * (u2 * 0.169507819808272 - 0.397203533833404) - 0.232500544458471) + 1.0) - 0.539825872510702);
    sly = S * z * sqrt(1.0 + slx * slx);
}
// returns false when the sample is rejected (all outputs zero)
MRL_HD bool ggx_sample(const MaterialDev &g, const Vec3d &in, float u0, float u1,
                                           float wo[3], float &pdf, float weight[3])
{
    const double al = g.alpha;
    double sx = al * in.x, sy = al * in.y, sz = in.z;
    double sl = sqrt(sx * sx + sy * sy + sz * sz);
    if (sl > 0.0) { sx /= sl; sy /= sl; sz /= sl; }
    double theta = 0.0, phi = 0.0;
    if (sz < 0.99999) { theta = acos(sz); phi = atan2(sy, sx); }
    double slx, sly;
    ggx_sample_visible_11(theta, (double)u0, (double)u1, slx, sly);
    double cp = cos(phi), sp = sin(phi);
    double mx = (cp * slx - sp * sly) * al, my = (sp * slx + cp * sly) * al;
    double nrm = 1.0 / sqrt(mx * mx + my * my + 1.0);
    Vec3d m = { -mx * nrm, -my * nrm, nrm };
    double c = in.x * m.x + in.y * m.y + in.z * m.z;
    Vec3d out = { 2.0 * c * m.x - in.x, 2.0 * c * m.y - in.y, 2.0 * c * m.z - in.z };
    if (!(out.z > 0.0) || !(c > 0.0)) return false;
    double D = ggx_D(al, m);
    double p = D * ggx_G1(al, in, m) / (4.0 * in.z);
    if (!(p > 0.0)) return false;
    float wx = (float)out.x, wy = (float)out.y, wz = (float)out.z;
    if (!(wz > 0.0f)) return false;
    wo[0] = wx; wo[1] = wy; wo[2] = wz;
    pdf = (float)p;
    double G1o = ggx_G1(al, out, m);
    weight[0] = (float)(fresnel_conductor(c, g.eta[0], g.k[0]) * G1o);
    weight[1] = (float)(fresnel_conductor(c, g.eta[1], g.k[1]) * G1o);
    weight[2] = (float)(fresnel_conductor(c, g.eta[2], g.k[2]) * G1o);
    return true;
}

// ---- table importance sampling (definition in oracle/merl_oracle.h) --------
// one-sample mixture: u0 < 1/2 -> cosine hemisphere with (2 u0, u1); else theta_h from the table's row
// marginal (sin^2 theta_h is uniform inside a row bin), phi_h = 2 pi u1, wo = reflect(wi, h).
// pdf(wi, wo) = 1/2 cos(theta_o)/pi + 1/2 c_i h.z / (4 wi.h)
MRL_HD int bin_of(const double *a, int n, double x)      // largest i in [0,n-1] with a[i] <= x
{
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

// The distribution one sample()/pdf() call uses (MRL_OPT_SAMPLING): 1 = the table's row marginal with the cosine lobe at
// weight 1/2; 2 = the row of the incident direction's bin in the conditional table P(theta_h | theta_i) — it follows the
// BRDF itself at that incidence, so the cosine lobe only keeps the estimator bounded: weight 1/8 (definition:
// oracle/merl_oracle.h; measured on the GGX-shaped table: the weights' variance falls by a third against mode 1).
// A material without a conditional table (n-channel tables) answers mode 2 with its marginal.
struct SamplingRow {
    const double *s, *cdf, *c;
    int n;
    float alpha;             // 1/2 or 1/8: u / alpha and u - alpha are exact in Float
};
MRL_HD SamplingRow sampling_row(const MaterialDev &m, int mode, double in_z)
{
    if (mode == 2 && m.sampling2d) {
        int i = (int)(in_z * (double)m.n_ti);
        i = i < 0 ? 0 : (i >= m.n_ti ? m.n_ti - 1 : i);
        const double *row = m.sampling2d + (size_t)i * (size_t)(2 * m.n_th + 1);
        return { m.sampling, row, row + (m.n_th + 1), m.n_th, 0.125f };
    }
    return { m.sampling, m.sampling + (m.n_th + 1), m.sampling + 2 * (m.n_th + 1), m.n_th, 0.5f };
}

MRL_HD double table_pdf(const MaterialDev &m, const Vec3d &in, const Vec3d &out, float woz, int mode)
{
    const SamplingRow r = sampling_row(m, mode, in.z);
    Vec3d h = { in.x + out.x, in.y + out.y, in.z + out.z };
    const double inv = 1.0 / sqrt(h.x * h.x + h.y * h.y + h.z * h.z);
    h.x *= inv; h.y *= inv; h.z *= inv;
    const int i = bin_of(r.s, r.n, h.x * h.x + h.y * h.y);
    const double ih = in.x * h.x + in.y * h.y + in.z * h.z;
    const double ph = r.c[i] * h.z / (4.0 * ih);
    return (double)r.alpha * ((double)woz * 0.31830988618379067154) + (1.0 - (double)r.alpha) * ph;
}

// direction of the mixture sample (Float); z <= 0 means "rejected"
MRL_HD void table_sample_dir(const MaterialDev &m, int disk_map, const Vec3d &in, float u0, float u1,
                                                 float &x, float &y, float &z, int mode)
{
    const SamplingRow r = sampling_row(m, mode, in.z);
    if (u0 < r.alpha) {
        square_to_cosine_hemisphere(disk_map, u0 * (1.0f / r.alpha), u1, x, y, z);
        return;
    }
    const double t = (double)(u0 - r.alpha) * (1.0 / (1.0 - (double)r.alpha));
    const int i = bin_of(r.cdf, r.n, t);
    const double xi = (t - r.cdf[i]) / (r.cdf[i + 1] - r.cdf[i]);
    const double sin2 = r.s[i] + xi * (r.s[i + 1] - r.s[i]);
    const double ct = sqrt(1.0 - sin2 > 0.0 ? 1.0 - sin2 : 0.0), st = sqrt(sin2);
    const double phi = 2.0 * kPi * (double)u1;
    const Vec3d h = { st * cos(phi), st * sin(phi), ct };
    const double c = in.x * h.x + in.y * h.y + in.z * h.z;
    x = (float)(2.0 * c * h.x - in.x); y = (float)(2.0 * c * h.y - in.y); z = (float)(2.0 * c * h.z - in.z);
}

// ---- tails shared by every kernel (generic and tuned), so that entry points agree bit for bit on the same lookup value ----
namespace fast {
// 1 / x: v_rcp_f64 seed + one Newton step (relative error ~1e-15)
MRL_HD double rcp_nr(double x)
{
    double y = rcp_seed(x);
    double e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}

// NaN / inf directions: the floors and guards of this file would turn them into finite garbage; an f64 CPU evaluation
// propagates NaN instead, so the cosine factor is poisoned when a component of either input is not finite.
// wi_sum = wix + wiy + wiz is shared by the two lookups of a unit.
MRL_HD float cos_or_nan32(float wi_sum, float wox, float woy, float woz, bool no_cosine = false)
{
    const float t = wi_sum + (wox + woy + woz);                     // NaN or inf iff some component is
    return (__builtin_fabsf(t) <= 3.0e38f) ? (no_cosine ? 1.0f : woz) : __builtin_nanf("");
}
MRL_HD double cos_or_nan(float wix, float wiy, float wiz, float wox, float woy, float woz)
{
    return (double)cos_or_nan32((wix + wiy + wiz), wox, woy, woz);
}

// a5 tail: rgb = f cos(theta_o) in Float (the plugin's own arithmetic: Spectrum * Float); zero for a pair that fails the
// cosine guards (texels are finite by construction, so the factor 0 is enough), NaN where an input component is not
// finite and the guards pass
// no_cosine (MRL_OPT_COSINE_FACTOR = 1, wave-uniform): the factor is 1 — eval() returns f alone
MRL_HD void eval_tail(const Rgbf &v, float wi_sum, float wiz, float wox, float woy, float woz, float rgb[3], bool no_cosine = false)
{
#pragma clang fp contract(off)
    const bool valid = (wiz > 0.0f) && (woz > 0.0f);
    const float c32 = cos_or_nan32(wi_sum, wox, woy, woz, no_cosine);
    const float c = valid ? c32 : 0.0f;
    rgb[0] = v.r * c; rgb[1] = v.g * c; rgb[2] = v.b * c;
}
// a6 tail: weight = eval(wi, wo') / pdf IN Float, as the plugin computes it: f = f_d cos theta_o' first, then the
// Float quotient f / pdf.  The three IEEE divisions share one reciprocal: q = double(f) * (1/pdf) with 1/pdf good to
// ~3e-16 (two Newton steps) is within 5e-16 of the true quotient, and a quotient of two 24-bit floats is never closer
// than 2^-49 = 1.8e-15 (relative) to a rounding boundary of Float, so Float(q) IS the correctly rounded f / pdf
// (tests/test_gpu_fullsize.py checks weight == eval / pdf bit for bit on 64M units).  Zero when the sample is invalid
// or its pdf is zero.
MRL_HD void sample_tail(const Rgbf &v, float wi_sum, float wiz, float sx, float sy, float sz, float p, bool table_sampling,
                                            float wo[3], float &pdf, float weight[3], bool no_cosine = false)
{
#pragma clang fp contract(off)
    const bool valid = (wiz > 0.0f) && (!table_sampling || p > 0.0f);
    const bool has = valid && (p > 0.0f);
    const float c32 = cos_or_nan32(wi_sum, sx, sy, sz, no_cosine);
    const float c = has ? c32 : 0.0f;
    const double pd = (double)(has ? p : 1.0f);
    double y = rcp_seed(pd);
    y = __builtin_fma(y, __builtin_fma(-pd, y, 1.0), y);
    y = __builtin_fma(y, __builtin_fma(-pd, y, 1.0), y);
    wo[0] = valid ? sx : 0.0f; wo[1] = valid ? sy : 0.0f; wo[2] = valid ? sz : 0.0f;
    pdf = valid ? p : 0.0f;
    weight[0] = (float)((double)(v.r * c) * y);
    weight[1] = (float)((double)(v.g * c) * y);
    weight[2] = (float)((double)(v.b * c) * y);
}
} // namespace fast

// ---- a5 / a6 / a7 for one unit, any material kind ------------------------------------------
// eval(): rgb = f * cos(theta_o), zero unless cos(theta_i) > 0 and cos(theta_o) > 0
MRL_HD void unit_eval(const MaterialDev &m, const Options &o,
                                          float wix, float wiy, float wiz, float wox, float woy, float woz,
                                          float rgb[3])
{
    rgb[0] = rgb[1] = rgb[2] = 0.0f;
    if (!(wiz > 0.0f) || !(woz > 0.0f)) return;
    Vec3d in = normalized(wix, wiy, wiz), out = normalized(wox, woy, woz);
    if (m.kind != KIND_GGX) {
        fast::eval_tail(table_brdf(m, o, in, out), (wix + wiy + wiz), wiz, wox, woy, woz, rgb, o.cosine != 0);
        return;
    }
    const Rgbd v = ggx_eval(m, in, out);
    rgb[0] = (float)v.r; rgb[1] = (float)v.g; rgb[2] = (float)v.b;
}

MRL_HD float unit_pdf(const MaterialDev &m, const Options &o, float wix, float wiy, float wiz, float wox, float woy, float woz)
{
    if (!(wiz > 0.0f) || !(woz > 0.0f)) return 0.0f;
    if (m.kind == KIND_GGX) {
        Vec3d in = normalized(wix, wiy, wiz), out = normalized(wox, woy, woz);
        return (float)ggx_pdf(m, in, out);
    }
    if (o.sampling) return (float)table_pdf(m, normalized(wix, wiy, wiz), normalized(wox, woy, woz), woz, o.sampling);
    return woz * kInvPiF;
}

MRL_HD void unit_sample(const MaterialDev &m, const Options &o,
                                            float wix, float wiy, float wiz, float u0, float u1,
                                            float wo[3], float &pdf, float weight[3])
{
    wo[0] = wo[1] = wo[2] = 0.0f; pdf = 0.0f; weight[0] = weight[1] = weight[2] = 0.0f;
    if (!(wiz > 0.0f)) return;
    if (m.kind == KIND_GGX) {
        Vec3d in = normalized(wix, wiy, wiz);
        ggx_sample(m, in, u0, u1, wo, pdf, weight);
        return;
    }
    float x, y, z, p;
    if (o.sampling) {
        const Vec3d in = normalized(wix, wiy, wiz);
        table_sample_dir(m, o.disk_map, in, u0, u1, x, y, z, o.sampling);
        if (!(z > 0.0f)) return;                          // reflected below the horizon: rejected
        p = (float)table_pdf(m, in, normalized(x, y, z), z, o.sampling);
        if (!(p > 0.0f)) return;
    } else {
        square_to_cosine_hemisphere(o.disk_map, u0, u1, x, y, z);
        p = z > 0.0f ? z * kInvPiF : 0.0f;
    }
    Vec3d in = normalized(wix, wiy, wiz), out = normalized(x, y, z);
    fast::sample_tail(table_brdf(m, o, in, out), (wix + wiy + wiz), wiz, x, y, z, p, o.sampling != 0, wo, pdf, weight, o.cosine != 0);
}

// ---- synthetic inputs, bit-identical to the oracle's generator ---------------
MRL_HD uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
MRL_HD void hemisphere_dir(uint64_t r, float &dx, float &dy, float &dz)
{
#pragma clang fp contract(off)
    const float TWO_NEG24 = 0x1p-24f, STEP = 0x1.921fb6p-22f;
    float z = (float)(uint32_t)(((r >> 41) << 1) | 1u) * TWO_NEG24;
    int32_t k = (int32_t)((r >> 8) & 0xFFFFFFu);
    int32_t q = (k + (1 << 21)) >> 22;
    int32_t j = k - (q << 22);
    float s, c;
    sincos_quarter_f32((float)j * STEP, s, c);
    float cs, sn;
    switch (q & 3) {
        case 0:  cs = c;  sn = s;  break;
        case 1:  cs = -s; sn = c;  break;
        case 2:  cs = -c; sn = -s; break;
        default: cs = s;  sn = -c; break;
    }
    float rr = __builtin_sqrtf(__builtin_fmaf(-z, z, 1.0f));
    dx = rr * cs; dy = rr * sn; dz = z;
}

} // namespace mrl
