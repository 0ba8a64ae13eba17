/*
 * merl_oracle.c — CPU oracle (test infrastructure; see merl_oracle.h for the rules).
 *
 * PARITY UNPINNED: /root/reference contains no source, tests or vectors for this path
 * (reference README.md:1 is its only statement).  Every function below
 * restates the PUBLIC algorithm named there and cites the SURVEY.md appendix item it follows.
 * Plain C, double precision, one pair per call, no SIMD intrinsics.  Build: oracle/Makefile
 * (-O2 -ffp-contract=off so that the f32 parts round exactly like the HIP kernels' f32 parts).
 */
#include "merl_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------
 * a1 — file format (reference location: absent, named by README.md:1).
 * little-endian int32 dims[3], then 3*n doubles, planar R,G,B.
 * ---------------------------------------------------------------------------------------- */
int orc_read_table(const char *path, int require_merl_dims, double **out_data, int dims[3])
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    int32_t d[3];
    if (fread(d, sizeof(int32_t), 3, f) != 3) { fclose(f); return -2; }
    if (d[0] <= 0 || d[1] <= 0 || d[2] <= 0) { fclose(f); return -3; }
    long long n = (long long)d[0] * d[1] * d[2];
    if (require_merl_dims && n != (long long)ORC_MERL_N) { fclose(f); return -3; }
    if (n > (1LL << 28)) { fclose(f); return -3; }
    double *buf = (double *)malloc(sizeof(double) * 3 * (size_t)n);
    if (!buf) { fclose(f); return -4; }
    if (fread(buf, sizeof(double), 3 * (size_t)n, f) != 3 * (size_t)n) { free(buf); fclose(f); return -2; }
    fclose(f);
    dims[0] = d[0]; dims[1] = d[1]; dims[2] = d[2];
    *out_data = buf;
    return 0;
}

int orc_write_table(const char *path, const double *planar, const int dims[3])
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    int32_t d[3] = { dims[0], dims[1], dims[2] };
    size_t n = (size_t)dims[0] * dims[1] * dims[2];
    int ok = fwrite(d, sizeof(int32_t), 3, f) == 3 && fwrite(planar, sizeof(double), 3 * n, f) == 3 * n;
    fclose(f);
    return ok ? 0 : -2;
}

void orc_free(void *p) { free(p); }

void orc_merl_table(orc_table *t, const double *planar)
{
    t->n_th = ORC_MERL_N_TH; t->n_td = ORC_MERL_N_TD; t->n_pd = ORC_MERL_N_PD;
    t->data = planar;
    t->scale[0] = 1.0 / 1500.0;      /* A.1 channel scales */
    t->scale[1] = 1.15 / 1500.0;
    t->scale[2] = 1.66 / 1500.0;
    t->param = ORC_PARAM_HALF_DIFF;
}

/* ------------------------------------------------------------------------------------------
 * a2 — half/diff transform: h = normalize((in+out)/2); theta_h = acos h.z;
 * phi_h = atan2(h.y,h.x); diff = R_y(-theta_h) R_z(-phi_h) in (Rodrigues rotations);
 * theta_d = acos diff.z; phi_d = atan2(diff.y, diff.x).
 * ---------------------------------------------------------------------------------------- */
static void rotate_about(const double v[3], const double axis[3], double angle, double r[3])
{
    double c = cos(angle), s = sin(angle);
    double along = (axis[0] * v[0] + axis[1] * v[1] + axis[2] * v[2]) * (1.0 - c);
    double cx = axis[1] * v[2] - axis[2] * v[1];
    double cy = axis[2] * v[0] - axis[0] * v[2];
    double cz = axis[0] * v[1] - axis[1] * v[0];
    r[0] = v[0] * c + axis[0] * along + cx * s;
    r[1] = v[1] * c + axis[1] * along + cy * s;
    r[2] = v[2] * c + axis[2] * along + cz * s;
}

static double clamp_unit(double x) { return x > 1.0 ? 1.0 : (x < -1.0 ? -1.0 : x); }

static void unit3(double v[3])
{
    double len = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (len > 0.0) { v[0] /= len; v[1] /= len; v[2] /= len; }
}

void orc_half_diff(const double in[3], const double out[3],
                   double *theta_half, double *phi_half, double *theta_diff, double *phi_diff)
{
    static const double z_axis[3] = { 0.0, 0.0, 1.0 };
    static const double y_axis[3] = { 0.0, 1.0, 0.0 };
    double h[3] = { (in[0] + out[0]) * 0.5, (in[1] + out[1]) * 0.5, (in[2] + out[2]) * 0.5 };
    unit3(h);
    /* acos argument clamped: rounding can push |h.z| past 1 (robustness; not in BRDFRead) */
    double th = acos(clamp_unit(h[2]));
    double ph = atan2(h[1], h[0]);
    double tmp[3], d[3];
    rotate_about(in, z_axis, -ph, tmp);
    rotate_about(tmp, y_axis, -th, d);
    *theta_half = th;
    *phi_half = ph;
    *theta_diff = acos(clamp_unit(d[2]));
    *phi_diff = atan2(d[1], d[0]);
}

/* the standard parameterisations (merl_oracle.h): polar angles and the azimuth difference, atan2 forms */
void orc_standard_angles(const double in[3], const double out[3], double *theta_i, double *theta_o, double *dphi)
{
    *theta_i = atan2(sqrt(in[0] * in[0] + in[1] * in[1]), in[2]);
    *theta_o = atan2(sqrt(out[0] * out[0] + out[1] * out[1]), out[2]);
    const double cr = in[0] * out[1] - in[1] * out[0], dt = in[0] * out[0] + in[1] * out[1];
    *dphi = (cr == 0.0 && dt == 0.0) ? 0.0 : atan2(cr, dt);       /* a direction AT the normal has no azimuth: 0 (atan2 of signed zeros may say pi) */
}

void orc_table_angles(const orc_table *t, const double in[3], const double out[3], double a[3])
{
    if (t->param == ORC_PARAM_HALF_DIFF) {
        double ph;
        orc_half_diff(in, out, &a[0], &ph, &a[1], &a[2]);
    } else
        orc_standard_angles(in, out, &a[0], &a[1], &a[2]);
}

/* ------------------------------------------------------------------------------------------
 * a3 — index maps, generalised from 90/90/180 to the table's dims and, for the
 * standard parameterisations, to linear axes (no sqrt warp; azimuth over [0,pi] or [0,2pi)).
 * ---------------------------------------------------------------------------------------- */
static double x_theta_half(const orc_table *t, double theta_half)
{
    if (t->param != ORC_PARAM_HALF_DIFF) return theta_half / (M_PI * 0.5) * t->n_th;
    if (theta_half <= 0.0) return 0.0;
    double deg = (theta_half / (M_PI / 2.0)) * t->n_th;
    return sqrt(deg * t->n_th);
}
static double x_theta_diff(const orc_table *t, double theta_diff)
{
    return theta_diff / (M_PI * 0.5) * t->n_td;
}
static double x_phi_diff(const orc_table *t, double phi_diff)
{
    if (t->param == ORC_PARAM_STANDARD) return fabs(phi_diff) / M_PI * t->n_pd;               /* mirror symmetry */
    if (t->param == ORC_PARAM_STANDARD_FULL) return (phi_diff < 0.0 ? phi_diff + 2.0 * M_PI : phi_diff) / (2.0 * M_PI) * t->n_pd;
    if (phi_diff < 0.0) phi_diff += M_PI;       /* reciprocity fold: phi_d == phi_d + pi */
    return phi_diff / M_PI * t->n_pd;
}
static int clamp_index(double x, int n)
{
    int i = (int)x;                              /* truncation, as BRDFRead */
    if (i < 0) return 0;
    if (i > n - 1) return n - 1;
    return i;
}
int orc_theta_half_index(const orc_table *t, double th) { return clamp_index(x_theta_half(t, th), t->n_th); }
int orc_theta_diff_index(const orc_table *t, double td) { return clamp_index(x_theta_diff(t, td), t->n_td); }
int orc_phi_diff_index(const orc_table *t, double pd)   { return clamp_index(x_phi_diff(t, pd), t->n_pd); }

void orc_coords(const orc_table *t, double th, double td, double pd, double *x_th, double *x_td, double *x_pd)
{
    *x_th = x_theta_half(t, th);
    *x_td = x_theta_diff(t, td);
    *x_pd = x_phi_diff(t, pd);
}

/* ------------------------------------------------------------------------------------------
 * a4 — table fetch.  Texel = scaled value, negatives (MERL's below-horizon
 * markers) clamped to 0 BEFORE interpolation.
 * ---------------------------------------------------------------------------------------- */
static void texel_raw(const orc_table *t, int ith, int itd, int ipd, double rgb[3])
{
    size_t n = (size_t)t->n_th * t->n_td * t->n_pd;
    size_t ind = (size_t)ipd + (size_t)t->n_pd * ((size_t)itd + (size_t)t->n_td * (size_t)ith);
    for (int c = 0; c < 3; ++c) rgb[c] = t->data[ind + c * n] * t->scale[c];
}
static void texel(const orc_table *t, int ith, int itd, int ipd, double rgb[3])
{
    texel_raw(t, ith, itd, ipd, rgb);
    for (int c = 0; c < 3; ++c) rgb[c] = rgb[c] > 0.0 ? rgb[c] : 0.0;
}
/* one channel of one corner into the running sums of a lookup, under the negative-value policy (merl_oracle.h):
 * num += w v (clamped / as stored / only if valid), den += w (RENORMALISE: only if valid) */
static void corner_accumulate(int negative, double w, double v, double *num, double *den)
{
    if (negative == ORC_NEGATIVE_KEEP) { *num += w * v; *den += w; }
    else if (negative == ORC_NEGATIVE_RENORMALISE) { if (v >= 0.0) { *num += w * v; *den += w; } }
    else { *num += w * (v > 0.0 ? v : 0.0); *den += w; }
}
static double corner_finish(int negative, double num, double den)
{
    if (negative != ORC_NEGATIVE_RENORMALISE) return num;
    return den > 0.0 ? num / den : 0.0;
}
static double nearest_value(int negative, double v)
{
    return negative == ORC_NEGATIVE_KEEP ? v : (v > 0.0 ? v : 0.0);
}

/* split a continuous coordinate into (i0, i1, f) for a clamped axis */
static void split_clamped(double x, int n, int *i0, int *i1, double *f)
{
    double fl = floor(x);
    int i = (int)fl;
    if (i < 0) i = 0;
    if (i > n - 1) i = n - 1;
    double fr = x - (double)i;
    if (fr < 0.0) fr = 0.0;
    if (fr > 1.0) fr = 1.0;
    *i0 = i;
    *i1 = i + 1 > n - 1 ? n - 1 : i + 1;
    *f = fr;
}
/* … and for the periodic phi_d axis (period n: phi_d and phi_d + pi are the same sample) */
static void split_periodic(double x, int n, int *i0, int *i1, double *f)
{
    double fl = floor(x);
    int i = (int)fl;
    *f = x - fl;
    i %= n; if (i < 0) i += n;
    *i0 = i;
    *i1 = (i + 1) % n;
}

/* azimuth axis: periodic, except for the mirrored standard form where 0 and pi are the two ends */
static void split_phi(int param, double x, int n, int *i0, int *i1, double *f)
{
    if (param == ORC_PARAM_STANDARD) split_clamped(x, n, i0, i1, f);
    else split_periodic(x, n, i0, i1, f);
}

void orc_lookup(const orc_table *t, const orc_opts *o, double th, double td, double pd, double rgb[3])
{
    if (o->lookup == ORC_LOOKUP_NEAREST) {
        texel_raw(t, orc_theta_half_index(t, th), orc_theta_diff_index(t, td), orc_phi_diff_index(t, pd), rgb);
        for (int c = 0; c < 3; ++c) rgb[c] = nearest_value(o->negative, rgb[c]);
        return;
    }
    double shift = o->node == ORC_NODE_CENTER ? 0.5 : 0.0;
    double xh, xd, xp;
    orc_coords(t, th, td, pd, &xh, &xd, &xp);
    int h0, h1, d0, d1, p0, p1; double fh, fd, fp;
    split_clamped(xh - shift, t->n_th, &h0, &h1, &fh);
    split_clamped(xd - shift, t->n_td, &d0, &d1, &fd);
    split_phi(t->param, xp - shift, t->n_pd, &p0, &p1, &fp);
    const int hs[2] = { h0, h1 }, ds[2] = { d0, d1 }, ps[2] = { p0, p1 };
    const double wh[2] = { 1.0 - fh, fh }, wd[2] = { 1.0 - fd, fd }, wp[2] = { 1.0 - fp, fp };
    double num[3] = { 0.0, 0.0, 0.0 }, den[3] = { 0.0, 0.0, 0.0 };
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int c = 0; c < 2; ++c) {
                double v[3];
                texel_raw(t, hs[a], ds[b], ps[c], v);
                double w = wh[a] * wd[b] * wp[c];
                for (int k = 0; k < 3; ++k) corner_accumulate(o->negative, w, v[k], &num[k], &den[k]);
            }
    for (int k = 0; k < 3; ++k) rgb[k] = corner_finish(o->negative, num[k], den[k]);
}

/* ------------------------------------------------------------------------------------------
 * a5 — BSDF::eval: zero unless cos(theta_i) > 0 and cos(theta_o) > 0; the
 * directions (Mitsuba Float = float, local frame, z = normal) are normalised in double, run
 * through a2→a4, and the value is multiplied by cos(theta_o) = wo.z.
 * ---------------------------------------------------------------------------------------- */
static void eval_f64(const orc_table *t, const orc_opts *o, const float wi[3], const float wo[3], double rgb[3])
{
    rgb[0] = rgb[1] = rgb[2] = 0.0;
    if (!(wi[2] > 0.0f) || !(wo[2] > 0.0f)) return;
    double in[3] = { wi[0], wi[1], wi[2] }, out[3] = { wo[0], wo[1], wo[2] };
    unit3(in); unit3(out);
    double a[3];
    orc_table_angles(t, in, out, a);
    orc_lookup(t, o, a[0], a[1], a[2], rgb);
    double c = o->cosine == ORC_COSINE_OMITTED ? 1.0 : (double)wo[2];
    rgb[0] *= c; rgb[1] *= c; rgb[2] *= c;
}

void orc_eval(const orc_table *t, const orc_opts *o, const float wi[3], const float wo[3], float rgb[3])
{
    double v[3];
    eval_f64(t, o, wi, wo, v);
    rgb[0] = (float)v[0]; rgb[1] = (float)v[1]; rgb[2] = (float)v[2];
}

/* a7 — BSDF::pdf: cosine-hemisphere density cos(theta_o)/pi in Float (A.5) */
static const float ORC_INV_PI_F = 0.31830988618379067154f;
float orc_pdf(const float wi[3], const float wo[3])
{
    if (!(wi[2] > 0.0f) || !(wo[2] > 0.0f)) return 0.0f;
    return wo[2] * ORC_INV_PI_F;
}

/* ------------------------------------------------------------------------------------------
 * a6 — BSDF::sample (A.5): wo = squareToCosineHemisphere(u) via the concentric disk map, in
 * Float (f32).  libm sinf/cosf differ between hosts in the last ulp, so the map's sincos is
 * pinned here to an explicit f32 polynomial (every operation an IEEE f32 op or fmaf); the
 * HIP kernel runs the same sequence, which makes the sampled direction bit-identical.
 * ---------------------------------------------------------------------------------------- */
static void sincos_quarter_f32(float t, float *s, float *c)   /* |t| <= pi/4 */
{
    const float S0 = -0x1.555552p-3f, S1 = 0x1.110c28p-7f, S2 = -0x1.9ac98ep-13f;
    const float C0 = 0x1.555552p-5f, C1 = -0x1.6c10dp-10f, C2 = 0x1.9b31dep-16f;
    float z = t * t;
    float p = fmaf(S2, z, S1); p = fmaf(p, z, S0);
    *s = fmaf(p * z, t, t);
    float q = fmaf(C2, z, C1); q = fmaf(q, z, C0);
    *c = fmaf(q * z, z, fmaf(-0.5f, z, 1.0f));
}

void orc_square_to_cosine_hemisphere(int disk_map, const float u[2], float wo[3])
{
    const float QUARTER_PI = 0.78539816339744830962f;
    float a = 2.0f * u[0] - 1.0f, b = 2.0f * u[1] - 1.0f;
    float x, y;
    if (a == 0.0f && b == 0.0f) {
        x = 0.0f; y = 0.0f;
    } else {
        /* Mitsuba 0.6: (a*a > b*b) ? first : second;  Mitsuba 3: (|a| < |b|) ? second : first */
        int first = disk_map == ORC_DISK_MITSUBA3 ? !(fabsf(a) < fabsf(b)) : (a * a > b * b);
        float r = first ? a : b;
        float ratio = first ? b / a : a / b;
        float s, c;
        sincos_quarter_f32(QUARTER_PI * ratio, &s, &c);
        /* first: phi = t; second: phi = pi/2 - t  =>  (cos phi, sin phi) = (sin t, cos t) */
        x = r * (first ? c : s);
        y = r * (first ? s : c);
    }
    float zz = 1.0f - fmaf(y, y, x * x);
    float z = zz > 0.0f ? sqrtf(zz) : 0.0f;
    if (disk_map == ORC_DISK_MITSUBA06 && z == 0.0f) z = 1e-10f;   /* 0.6 guard against z == 0 */
    wo[0] = x; wo[1] = y; wo[2] = z;
}

void orc_sample(const orc_table *t, const orc_opts *o, const float wi[3], const float u[2],
                float wo[3], float *pdf, float weight[3])
{
    wo[0] = wo[1] = wo[2] = 0.0f; *pdf = 0.0f; weight[0] = weight[1] = weight[2] = 0.0f;
    if (!(wi[2] > 0.0f)) return;
    orc_square_to_cosine_hemisphere(o->disk_map, u, wo);
    float p = orc_pdf(wi, wo);
    *pdf = p;
    if (!(p > 0.0f)) return;
    float f[3];
    orc_eval(t, o, wi, wo, f);                  /* eval(bRec) / pdf in Float, as 0.6's sample() */
    weight[0] = f[0] / p; weight[1] = f[1] / p; weight[2] = f[2] / p;
}

/* ------------------------------------------------------------------------------------------
 * §8f item 2 — table importance sampling (see merl_oracle.h for the definition).
 * ---------------------------------------------------------------------------------------- */
int orc_build_sampling(const orc_table *t, orc_sampling *out)
{
    const int n = t->n_th;
    out->n = n;
    out->s = (double *)malloc(sizeof(double) * (size_t)(n + 1));
    out->cdf = (double *)malloc(sizeof(double) * (size_t)(n + 1));
    out->c = (double *)malloc(sizeof(double) * (size_t)n);
    double *D = (double *)malloc(sizeof(double) * (size_t)n);
    if (!out->s || !out->cdf || !out->c || !D) { free(D); orc_free_sampling(out); return -4; }
    double mean = 0.0;
    for (int i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int j = 0; j < t->n_td; ++j)
            for (int k = 0; k < t->n_pd; ++k) {
                double v[3];
                texel(t, i, j, k, v);
                acc += 0.2126 * v[0] + 0.7152 * v[1] + 0.0722 * v[2];
            }
        D[i] = acc / ((double)t->n_td * (double)t->n_pd);
        mean += D[i];
    }
    mean /= (double)n;
    if (t->param != ORC_PARAM_HALF_DIFF) mean = 0.0;                  /* rows are not theta_h: flat lobe */
    for (int i = 0; i < n; ++i) D[i] = mean > 0.0 ? D[i] + 0.01 * mean : 1.0;
    for (int i = 0; i <= n; ++i) {
        double r = (double)i / (double)n;
        double sn = sin(r * r * (M_PI / 2.0));
        out->s[i] = i == n ? 1.0 : sn * sn;
    }
    double Z = 0.0;
    for (int i = 0; i < n; ++i) Z += D[i] * (out->s[i + 1] - out->s[i]);
    double run = 0.0;
    for (int i = 0; i < n; ++i) {
        out->cdf[i] = run / Z;
        run += D[i] * (out->s[i + 1] - out->s[i]);
        out->c[i] = D[i] / (M_PI * Z);
    }
    out->cdf[n] = 1.0;
    out->alpha = 0.5;
    free(D);
    return 0;
}

void orc_free_sampling(orc_sampling *sp)
{
    free(sp->s); free(sp->cdf); free(sp->c);
    sp->s = sp->cdf = sp->c = NULL;
}

/* largest i in [0, n-1] with a[i] <= x */
static int bin_of(const double *a, int n, double x)
{
    int lo = 0, hi = n;           /* invariant: a[lo] <= x < a[hi] (a[n] treated as +inf) */
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (a[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

static double pdf_table_f64(const orc_sampling *sp, const float wi[3], const float wo[3])
{
    if (!(wi[2] > 0.0f) || !(wo[2] > 0.0f)) return 0.0;
    double in[3] = { wi[0], wi[1], wi[2] }, out[3] = { wo[0], wo[1], wo[2] };
    unit3(in); unit3(out);
    double h[3] = { in[0] + out[0], in[1] + out[1], in[2] + out[2] };
    unit3(h);
    double sin2 = h[0] * h[0] + h[1] * h[1];
    int i = bin_of(sp->s, sp->n, sin2);
    double ih = in[0] * h[0] + in[1] * h[1] + in[2] * h[2];
    double ph = sp->c[i] * h[2] / (4.0 * ih);
    return sp->alpha * ((double)wo[2] * (1.0 / M_PI)) + (1.0 - sp->alpha) * ph;
}

float orc_pdf_table(const orc_sampling *sp, const float wi[3], const float wo[3])
{
    return (float)pdf_table_f64(sp, wi, wo);
}

void orc_sample_table(const orc_table *t, const orc_opts *o, const orc_sampling *sp, const float wi[3], const float u[2],
                      float wo[3], float *pdf, float weight[3])
{
    wo[0] = wo[1] = wo[2] = 0.0f; *pdf = 0.0f; weight[0] = weight[1] = weight[2] = 0.0f;
    if (!(wi[2] > 0.0f)) return;
    float d[3];
    const float alpha = (float)sp->alpha;                     /* 1/2 or 1/8: u / alpha and u - alpha are exact in Float */
    if (u[0] < alpha) {
        const float uu[2] = { u[0] * (1.0f / alpha), u[1] };
        orc_square_to_cosine_hemisphere(o->disk_map, uu, d);
    } else {
        double x = (double)(u[0] - alpha) * (1.0 / (1.0 - sp->alpha));
        int i = bin_of(sp->cdf, sp->n, x);
        double xi = (x - sp->cdf[i]) / (sp->cdf[i + 1] - sp->cdf[i]);
        double sin2 = sp->s[i] + xi * (sp->s[i + 1] - sp->s[i]);
        double ct = sqrt(1.0 - sin2 > 0.0 ? 1.0 - sin2 : 0.0), st = sqrt(sin2);
        double phi = 2.0 * M_PI * (double)u[1];
        double h[3] = { st * cos(phi), st * sin(phi), ct };
        double in[3] = { wi[0], wi[1], wi[2] };
        unit3(in);
        double c = in[0] * h[0] + in[1] * h[1] + in[2] * h[2];
        d[0] = (float)(2.0 * c * h[0] - in[0]); d[1] = (float)(2.0 * c * h[1] - in[1]); d[2] = (float)(2.0 * c * h[2] - in[2]);
    }
    if (!(d[2] > 0.0f)) return;                 /* below the horizon: rejected sample */
    float p = orc_pdf_table(sp, wi, d);
    if (!(p > 0.0f)) return;
    wo[0] = d[0]; wo[1] = d[1]; wo[2] = d[2];
    *pdf = p;
    float f[3];
    orc_eval(t, o, wi, wo, f);
    weight[0] = f[0] / p; weight[1] = f[1] / p; weight[2] = f[2] / p;
}

void orc_pdf_table_batch(const orc_sampling *sp, const float *wi, const float *wo, size_t n, float *pdf)
{
    for (size_t i = 0; i < n; ++i) pdf[i] = orc_pdf_table(sp, wi + 3 * i, wo + 3 * i);
}
void orc_sample_table_batch(const orc_table *t, const orc_opts *o, const orc_sampling *sp, const float *wi, const float *u,
                            size_t n, float *wo, float *pdf, float *weight)
{
    for (size_t i = 0; i < n; ++i) orc_sample_table(t, o, sp, wi + 3 * i, u + 2 * i, wo + 3 * i, pdf + i, weight + 3 * i);
}

/* ---- §8f item 2, survey form: P(theta_h | theta_i) (definition: merl_oracle.h) ---- */
static double brdf_mass_sample(const orc_table *t, const orc_opts *o, double mu, double s, double phi)
{
    /* lum f(wi, wo) cos(theta_o) 4 (wi.h) / (2 cos(theta_h)) for wi = (sqrt(1 - mu^2), 0, mu), h from (s, phi) */
    const double in[3] = { sqrt(1.0 - mu * mu > 0.0 ? 1.0 - mu * mu : 0.0), 0.0, mu };
    const double st = sqrt(s), ct = sqrt(1.0 - s > 0.0 ? 1.0 - s : 0.0);
    const double h[3] = { st * cos(phi), st * sin(phi), ct };
    const double c = in[0] * h[0] + in[1] * h[1] + in[2] * h[2];
    if (!(c > 0.0) || !(ct > 0.0)) return 0.0;
    double out[3] = { 2.0 * c * h[0] - in[0], 2.0 * c * h[1] - in[1], 2.0 * c * h[2] - in[2] };
    if (!(out[2] > 0.0)) return 0.0;
    double uin[3] = { in[0], in[1], in[2] }, a[3], rgb[3];
    unit3(uin); unit3(out);
    orc_table_angles(t, uin, out, a);
    orc_lookup(t, o, a[0], a[1], a[2], rgb);
    const double lum = 0.2126 * rgb[0] + 0.7152 * rgb[1] + 0.0722 * rgb[2];
    return lum * out[2] * 4.0 * c / (2.0 * ct);
}

int orc_build_sampling2d(const orc_table *t, const orc_opts *o, int n_i, orc_sampling2d *out)
{
    const int n = t->n_th;
    out->n_i = 0; out->rows = NULL;
    if (n_i < 1 || n < 1) return -1;
    out->rows = (orc_sampling *)calloc((size_t)n_i, sizeof(orc_sampling));
    double *W = (double *)malloc(sizeof(double) * (size_t)n);
    if (!out->rows || !W) { free(W); free(out->rows); out->rows = NULL; return -4; }
    out->n_i = n_i;
    orc_opts lo = *o;
    lo.lookup = 1;                                            /* the mass is measured with the interpolated table */
    lo.negative = ORC_NEGATIVE_CLAMP;                         /* ... of clamped values, whatever eval() does with negative ones */
    for (int i = 0; i < n_i; ++i) {
        orc_sampling *r = &out->rows[i];
        r->n = n;
        r->s = (double *)malloc(sizeof(double) * (size_t)(n + 1));
        r->cdf = (double *)malloc(sizeof(double) * (size_t)(n + 1));
        r->c = (double *)malloc(sizeof(double) * (size_t)n);
        if (!r->s || !r->cdf || !r->c) { free(W); orc_free_sampling2d(out); return -4; }
        for (int j = 0; j <= n; ++j) {
            const double q = (double)j / (double)n, sn = sin(q * q * (M_PI / 2.0));
            r->s[j] = j == n ? 1.0 : sn * sn;
        }
        const double mu = ((double)i + 0.5) / (double)n_i;
        double total = 0.0;
        for (int j = 0; j < n; ++j) {
            const double ds = r->s[j + 1] - r->s[j];
            double acc = 0.0;
            if (t->param == ORC_PARAM_HALF_DIFF)
                for (int a = 0; a < ORC_S2D_KS; ++a)
                    for (int b = 0; b < ORC_S2D_KP; ++b)
                        acc += brdf_mass_sample(t, &lo, mu, r->s[j] + ((double)a + 0.5) / ORC_S2D_KS * ds, ((double)b + 0.5) / ORC_S2D_KP * M_PI);
            W[j] = ds * acc / (double)(ORC_S2D_KS * ORC_S2D_KP);
            total += W[j];
        }
        /* the floor: 1 % of the row's mass, uniform in s (a flat row when the table gives nothing, e.g. a standard parameterisation) */
        const double span = r->s[n] - r->s[0];
        double Z = 0.0;
        for (int j = 0; j < n; ++j) {
            const double ds = r->s[j + 1] - r->s[j];
            W[j] = total > 0.0 ? W[j] + 0.01 * total * ds / span : ds / span;
            Z += W[j];
        }
        double run = 0.0;
        for (int j = 0; j < n; ++j) {
            r->cdf[j] = run / Z;
            run += W[j];
            r->c[j] = W[j] / (Z * M_PI * (r->s[j + 1] - r->s[j]));
        }
        r->cdf[n] = 1.0;
        r->alpha = 0.125;
    }
    free(W);
    return 0;
}

int orc_sampling2d_from_arrays(int n_i, int n, const double *s, const double *flat, orc_sampling2d *out)
{
    out->n_i = 0; out->rows = NULL;
    if (n_i < 1 || n < 1 || !s || !flat) return -1;
    out->rows = (orc_sampling *)calloc((size_t)n_i, sizeof(orc_sampling));
    if (!out->rows) return -4;
    out->n_i = n_i;
    for (int i = 0; i < n_i; ++i) {
        orc_sampling *r = &out->rows[i];
        r->n = n;
        r->s = (double *)malloc(sizeof(double) * (size_t)(n + 1));
        r->cdf = (double *)malloc(sizeof(double) * (size_t)(n + 1));
        r->c = (double *)malloc(sizeof(double) * (size_t)n);
        if (!r->s || !r->cdf || !r->c) { orc_free_sampling2d(out); return -4; }
        memcpy(r->s, s, sizeof(double) * (size_t)(n + 1));
        memcpy(r->cdf, flat + (size_t)i * (size_t)(2 * n + 1), sizeof(double) * (size_t)(n + 1));
        memcpy(r->c, flat + (size_t)i * (size_t)(2 * n + 1) + (size_t)(n + 1), sizeof(double) * (size_t)n);
        r->alpha = 0.125;
    }
    return 0;
}

void orc_free_sampling2d(orc_sampling2d *sp)
{
    if (sp->rows) for (int i = 0; i < sp->n_i; ++i) orc_free_sampling(&sp->rows[i]);
    free(sp->rows);
    sp->rows = NULL; sp->n_i = 0;
}

int orc_sampling2d_bin(const orc_sampling2d *sp, const float wi[3])
{
    double in[3] = { wi[0], wi[1], wi[2] };
    unit3(in);
    int i = (int)(in[2] * (double)sp->n_i);
    return i < 0 ? 0 : (i >= sp->n_i ? sp->n_i - 1 : i);
}

void orc_pdf_table2d_batch(const orc_sampling2d *sp, const float *wi, const float *wo, size_t n, float *pdf)
{
    for (size_t k = 0; k < n; ++k) pdf[k] = orc_pdf_table(&sp->rows[orc_sampling2d_bin(sp, wi + 3 * k)], wi + 3 * k, wo + 3 * k);
}

void orc_sample_table2d_batch(const orc_table *t, const orc_opts *o, const orc_sampling2d *sp, const float *wi, const float *u,
                              size_t n, float *wo, float *pdf, float *weight)
{
    for (size_t k = 0; k < n; ++k)
        orc_sample_table(t, o, &sp->rows[orc_sampling2d_bin(sp, wi + 3 * k)], wi + 3 * k, u + 2 * k, wo + 3 * k, pdf + k, weight + 3 * k);
}

/* ---- batches ---- */
void orc_eval_batch(const orc_table *t, const orc_opts *o, const float *wi, const float *wo, size_t n, float *rgb)
{
    for (size_t i = 0; i < n; ++i) orc_eval(t, o, wi + 3 * i, wo + 3 * i, rgb + 3 * i);
}
void orc_pdf_batch(const float *wi, const float *wo, size_t n, float *pdf)
{
    for (size_t i = 0; i < n; ++i) pdf[i] = orc_pdf(wi + 3 * i, wo + 3 * i);
}
void orc_sample_batch(const orc_table *t, const orc_opts *o, const float *wi, const float *u, size_t n,
                      float *wo, float *pdf, float *weight)
{
    for (size_t i = 0; i < n; ++i) orc_sample(t, o, wi + 3 * i, u + 2 * i, wo + 3 * i, pdf + i, weight + 3 * i);
}
void orc_eval_sample_batch_multi(const orc_table *tables, int n_tables, const orc_opts *o,
                                 const float *wi, const float *wo, const float *u, const int32_t *mat,
                                 size_t n, float *rgb, float *pdf, float *wo2, float *pdf2, float *weight)
{
    for (size_t i = 0; i < n; ++i) {
        int m = mat ? mat[i] : 0;
        if (m < 0 || m >= n_tables) {
            memset(rgb + 3 * i, 0, 12); pdf[i] = 0; memset(wo2 + 3 * i, 0, 12); pdf2[i] = 0; memset(weight + 3 * i, 0, 12);
            continue;
        }
        const orc_table *t = tables + m;
        orc_eval(t, o, wi + 3 * i, wo + 3 * i, rgb + 3 * i);
        pdf[i] = orc_pdf(wi + 3 * i, wo + 3 * i);
        orc_sample(t, o, wi + 3 * i, u + 2 * i, wo2 + 3 * i, pdf2 + i, weight + 3 * i);
    }
}

/* ------------------------------------------------------------------------------------------
 * §8f item 3 — n-channel tables (see merl_oracle.h).  The index maps only look at the dims, so a
 * dims-only orc_table view of the n-channel table feeds the functions above.
 * ---------------------------------------------------------------------------------------- */
static orc_table dims_view(const orc_table_nch *t)
{
    orc_table v;
    v.n_th = t->n_th; v.n_td = t->n_td; v.n_pd = t->n_pd; v.data = NULL;
    v.scale[0] = v.scale[1] = v.scale[2] = 1.0;
    v.param = t->param;
    return v;
}

static void texel_nch_raw(const orc_table_nch *t, int ith, int itd, int ipd, double *out)
{
    size_t n = (size_t)t->n_th * t->n_td * t->n_pd;
    size_t ind = (size_t)ipd + (size_t)t->n_pd * ((size_t)itd + (size_t)t->n_td * (size_t)ith);
    for (int c = 0; c < t->n_ch; ++c) out[c] = t->data[ind + (size_t)c * n] * t->scale[c];
}
static void texel_nch(const orc_table_nch *t, int ith, int itd, int ipd, double *out)
{
    texel_nch_raw(t, ith, itd, ipd, out);
    for (int c = 0; c < t->n_ch; ++c) out[c] = out[c] > 0.0 ? out[c] : 0.0;
}

#define ORC_MAX_CH 64

void orc_lookup_nch(const orc_table_nch *t, const orc_opts *o, double th, double td, double pd, double *out)
{
    const orc_table dv = dims_view(t);
    if (o->lookup == ORC_LOOKUP_NEAREST) {
        texel_nch_raw(t, orc_theta_half_index(&dv, th), orc_theta_diff_index(&dv, td), orc_phi_diff_index(&dv, pd), out);
        for (int c = 0; c < t->n_ch; ++c) out[c] = nearest_value(o->negative, out[c]);
        return;
    }
    double shift = o->node == ORC_NODE_CENTER ? 0.5 : 0.0;
    double xh, xd, xp;
    orc_coords(&dv, th, td, pd, &xh, &xd, &xp);
    int h0, h1, d0, d1, p0, p1; double fh, fd, fp;
    split_clamped(xh - shift, t->n_th, &h0, &h1, &fh);
    split_clamped(xd - shift, t->n_td, &d0, &d1, &fd);
    split_phi(t->param, xp - shift, t->n_pd, &p0, &p1, &fp);
    const int hs[2] = { h0, h1 }, ds[2] = { d0, d1 }, ps[2] = { p0, p1 };
    const double wh[2] = { 1.0 - fh, fh }, wd[2] = { 1.0 - fd, fd }, wp[2] = { 1.0 - fp, fp };
    double den[ORC_MAX_CH];
    for (int c = 0; c < t->n_ch; ++c) out[c] = den[c] = 0.0;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int c = 0; c < 2; ++c) {
                double v[ORC_MAX_CH];
                texel_nch_raw(t, hs[a], ds[b], ps[c], v);
                double w = wh[a] * wd[b] * wp[c];
                for (int k = 0; k < t->n_ch; ++k) corner_accumulate(o->negative, w, v[k], &out[k], &den[k]);
            }
    for (int k = 0; k < t->n_ch; ++k) out[k] = corner_finish(o->negative, out[k], den[k]);
}

void orc_eval_nch(const orc_table_nch *t, const orc_opts *o, const float wi[3], const float wo[3], float *out)
{
    for (int c = 0; c < t->n_ch; ++c) out[c] = 0.0f;
    if (!(wi[2] > 0.0f) || !(wo[2] > 0.0f)) return;
    double in[3] = { wi[0], wi[1], wi[2] }, od[3] = { wo[0], wo[1], wo[2] };
    unit3(in); unit3(od);
    double a[3], v[ORC_MAX_CH];
    const orc_table dv = dims_view(t);
    orc_table_angles(&dv, in, od, a);
    orc_lookup_nch(t, o, a[0], a[1], a[2], v);
    const double cosine = o->cosine == ORC_COSINE_OMITTED ? 1.0 : (double)wo[2];
    for (int c = 0; c < t->n_ch; ++c) out[c] = (float)(v[c] * cosine);
}

void orc_sample_nch(const orc_table_nch *t, const orc_opts *o, const float wi[3], const float u[2],
                    float wo[3], float *pdf, float *weight)
{
    wo[0] = wo[1] = wo[2] = 0.0f; *pdf = 0.0f;
    for (int c = 0; c < t->n_ch; ++c) weight[c] = 0.0f;
    if (!(wi[2] > 0.0f)) return;
    orc_square_to_cosine_hemisphere(o->disk_map, u, wo);
    float p = orc_pdf(wi, wo);
    *pdf = p;
    if (!(p > 0.0f)) return;
    float f[ORC_MAX_CH];
    orc_eval_nch(t, o, wi, wo, f);
    for (int c = 0; c < t->n_ch; ++c) weight[c] = f[c] / p;
}

int orc_build_sampling_nch(const orc_table_nch *t, orc_sampling *out)
{
    const int n = t->n_th;
    out->n = n;
    out->s = (double *)malloc(sizeof(double) * (size_t)(n + 1));
    out->cdf = (double *)malloc(sizeof(double) * (size_t)(n + 1));
    out->c = (double *)malloc(sizeof(double) * (size_t)n);
    double *D = (double *)malloc(sizeof(double) * (size_t)n);
    if (!out->s || !out->cdf || !out->c || !D) { free(D); orc_free_sampling(out); return -4; }
    double mean = 0.0;
    for (int i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int j = 0; j < t->n_td; ++j)
            for (int k = 0; k < t->n_pd; ++k) {
                double v[ORC_MAX_CH], sum = 0.0;
                texel_nch(t, i, j, k, v);
                for (int c = 0; c < t->n_ch; ++c) sum += v[c];
                acc += sum / (double)t->n_ch;
            }
        D[i] = acc / ((double)t->n_td * (double)t->n_pd);
        mean += D[i];
    }
    mean /= (double)n;
    if (t->param != ORC_PARAM_HALF_DIFF) mean = 0.0;                  /* rows are not theta_h: flat lobe */
    for (int i = 0; i < n; ++i) D[i] = mean > 0.0 ? D[i] + 0.01 * mean : 1.0;
    for (int i = 0; i <= n; ++i) {
        double r = (double)i / (double)n;
        double sn = sin(r * r * (M_PI / 2.0));
        out->s[i] = i == n ? 1.0 : sn * sn;
    }
    double Z = 0.0;
    for (int i = 0; i < n; ++i) Z += D[i] * (out->s[i + 1] - out->s[i]);
    double run = 0.0;
    for (int i = 0; i < n; ++i) {
        out->cdf[i] = run / Z;
        run += D[i] * (out->s[i + 1] - out->s[i]);
        out->c[i] = D[i] / (M_PI * Z);
    }
    out->cdf[n] = 1.0;
    out->alpha = 0.5;
    free(D);
    return 0;
}

void orc_sample_table_nch(const orc_table_nch *t, const orc_opts *o, const orc_sampling *sp, const float wi[3], const float u[2],
                          float wo[3], float *pdf, float *weight)
{
    wo[0] = wo[1] = wo[2] = 0.0f; *pdf = 0.0f;
    for (int c = 0; c < t->n_ch; ++c) weight[c] = 0.0f;
    if (!(wi[2] > 0.0f)) return;
    float d[3];
    if (u[0] < 0.5f) {
        const float uu[2] = { 2.0f * u[0], u[1] };
        orc_square_to_cosine_hemisphere(o->disk_map, uu, d);
    } else {
        double x = (double)(2.0f * u[0] - 1.0f);
        int i = bin_of(sp->cdf, sp->n, x);
        double xi = (x - sp->cdf[i]) / (sp->cdf[i + 1] - sp->cdf[i]);
        double sin2 = sp->s[i] + xi * (sp->s[i + 1] - sp->s[i]);
        double ct = sqrt(1.0 - sin2 > 0.0 ? 1.0 - sin2 : 0.0), st = sqrt(sin2);
        double phi = 2.0 * M_PI * (double)u[1];
        double h[3] = { st * cos(phi), st * sin(phi), ct };
        double in[3] = { wi[0], wi[1], wi[2] };
        unit3(in);
        double c = in[0] * h[0] + in[1] * h[1] + in[2] * h[2];
        d[0] = (float)(2.0 * c * h[0] - in[0]); d[1] = (float)(2.0 * c * h[1] - in[1]); d[2] = (float)(2.0 * c * h[2] - in[2]);
    }
    if (!(d[2] > 0.0f)) return;
    float p = orc_pdf_table(sp, wi, d);
    if (!(p > 0.0f)) return;
    wo[0] = d[0]; wo[1] = d[1]; wo[2] = d[2];
    *pdf = p;
    float f[ORC_MAX_CH];
    orc_eval_nch(t, o, wi, wo, f);
    for (int c = 0; c < t->n_ch; ++c) weight[c] = f[c] / p;
}

void orc_eval_sample_batch_nch(const orc_table_nch *tables, int n_tables, int n_ch, const orc_opts *o, const orc_sampling *sp,
                               const float *wi, const float *wo, const float *u, const int32_t *mat, size_t n,
                               float *values, float *pdf, float *wo2, float *pdf2, float *weight)
{
    for (size_t i = 0; i < n; ++i) {
        int m = mat ? mat[i] : 0;
        float *val = values + (size_t)n_ch * i, *w = weight + (size_t)n_ch * i;
        if (m < 0 || m >= n_tables || tables[m].n_ch != n_ch) {
            for (int c = 0; c < n_ch; ++c) { val[c] = 0.0f; w[c] = 0.0f; }
            pdf[i] = 0; memset(wo2 + 3 * i, 0, 12); pdf2[i] = 0;
            continue;
        }
        const orc_table_nch *t = tables + m;
        orc_eval_nch(t, o, wi + 3 * i, wo + 3 * i, val);
        if (sp) {
            pdf[i] = orc_pdf_table(sp + m, wi + 3 * i, wo + 3 * i);
            orc_sample_table_nch(t, o, sp + m, wi + 3 * i, u + 2 * i, wo2 + 3 * i, pdf2 + i, w);
        } else {
            pdf[i] = orc_pdf(wi + 3 * i, wo + 3 * i);
            orc_sample_nch(t, o, wi + 3 * i, u + 2 * i, wo2 + 3 * i, pdf2 + i, w);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * a9 — GGX rough conductor (stock upstream plugin, BASELINE config 3).
 * D = 1/(pi a^2 cos^4(tm) (1 + tan^2(tm)/a^2)^2); Smith G1(v) = 2/(1 + sqrt(1 + a^2 tan^2 tv));
 * eval = F(wi.m) D G /(4 cos ti); visible-normal sampling (Heitz & d'Eon 2014);
 * pdf = D G1(wi) / (4 cos ti); weight = F G1(wo).  All in double; outputs rounded to Float.
 * ---------------------------------------------------------------------------------------- */
static double ggx_D(double alpha, const double m[3])
{
    if (m[2] <= 0.0) return 0.0;
    double c2 = m[2] * m[2];
    double e = (m[0] * m[0] + m[1] * m[1]) / (alpha * alpha) / c2;   /* tan^2 / alpha^2 */
    double root = (1.0 + e) * c2;
    double r = 1.0 / (M_PI * alpha * alpha * root * root);
    return r * m[2] < 1e-20 ? 0.0 : r;
}
static double ggx_G1(double alpha, const double v[3], const double m[3])
{
    double vm = v[0] * m[0] + v[1] * m[1] + v[2] * m[2];
    if (vm * v[2] <= 0.0) return 0.0;
    double s2 = 1.0 - v[2] * v[2];
    if (s2 <= 0.0) return 1.0;
    double tan2 = s2 / (v[2] * v[2]);
    return 2.0 / (1.0 + sqrt(1.0 + alpha * alpha * tan2));
}
static double safe_sqrt(double x) { return x > 0.0 ? sqrt(x) : 0.0; }
static double fresnel_conductor(double c, double eta, double k)
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
static int ggx_dirs(const float wi[3], const float wo[3], double in[3], double out[3])
{
    if (!(wi[2] > 0.0f) || !(wo[2] > 0.0f)) return 0;
    in[0] = wi[0]; in[1] = wi[1]; in[2] = wi[2]; out[0] = wo[0]; out[1] = wo[1]; out[2] = wo[2];
    unit3(in); unit3(out);
    return 1;
}
static void ggx_eval_core(const orc_ggx *g, const double in[3], const double out[3], double rgb[3])
{
    double m[3] = { in[0] + out[0], in[1] + out[1], in[2] + out[2] };
    unit3(m);
    rgb[0] = rgb[1] = rgb[2] = 0.0;
    double D = ggx_D(g->alpha, m);
    if (D == 0.0) return;
    double G = ggx_G1(g->alpha, in, m) * ggx_G1(g->alpha, out, m);
    double model = D * G / (4.0 * in[2]);
    double c = in[0] * m[0] + in[1] * m[1] + in[2] * m[2];
    for (int ch = 0; ch < 3; ++ch) rgb[ch] = fresnel_conductor(c, g->eta[ch], g->k[ch]) * model;
}
void orc_ggx_eval(const orc_ggx *g, const float wi[3], const float wo[3], float rgb[3])
{
    double in[3], out[3], v[3] = { 0, 0, 0 };
    if (ggx_dirs(wi, wo, in, out)) ggx_eval_core(g, in, out, v);
    rgb[0] = (float)v[0]; rgb[1] = (float)v[1]; rgb[2] = (float)v[2];
}
static double ggx_pdf_core(const orc_ggx *g, const double in[3], const double out[3])
{
    double m[3] = { in[0] + out[0], in[1] + out[1], in[2] + out[2] };
    unit3(m);
    /* pdfVisible(wi,m) / (4 |wo.m|) with pdfVisible = D G1(wi) |wi.m| / cos ti  and wi.m = wo.m */
    return ggx_D(g->alpha, m) * ggx_G1(g->alpha, in, m) / (4.0 * in[2]);
}
float orc_ggx_pdf(const orc_ggx *g, const float wi[3], const float wo[3])
{
    double in[3], out[3];
    if (!ggx_dirs(wi, wo, in, out)) return 0.0f;
    return (float)ggx_pdf_core(g, in, out);
}
/* P22 slope sampling for alpha = 1 (Heitz & d'Eon 2014, "sample_visible_11") */
static void ggx_sample_visible_11(double theta_i, double u1, double u2, double slope[2])
{
    if (theta_i < 1e-4) {
        double r = safe_sqrt(u1 / (1.0 - u1));
        double phi = 2.0 * M_PI * u2;
        slope[0] = r * cos(phi); slope[1] = r * sin(phi);
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
    slope[0] = (A < 0.0 || s2 > 1.0 / tan_i) ? s1 : s2;
    double S;
    if (u2 > 0.5) { S = 1.0; u2 = 2.0 * (u2 - 0.5); }
    else { S = -1.0; u2 = 2.0 * (0.5 - u2); }
    double z = (u2 * (u2 * (u2 * (-0.365728915865723) + 0.790235037209296) - 0.424965825137544) + 0.000152998850436920)
             / (u2 * (u2 * (u2 * (u2 * 0.169507819808272 - 0.397203533833404) - 0.232500544458471) + 1.0) - 0.539825872510702);
    slope[1] = S * z * sqrt(1.0 + slope[0] * slope[0]);
}
void orc_ggx_sample(const orc_ggx *g, const float wi[3], const float u[2], float wo[3], float *pdf, float weight[3])
{
    wo[0] = wo[1] = wo[2] = 0.0f; *pdf = 0.0f; weight[0] = weight[1] = weight[2] = 0.0f;
    if (!(wi[2] > 0.0f)) return;
    double in[3] = { wi[0], wi[1], wi[2] };
    unit3(in);
    double al = g->alpha;
    /* 1 stretch */
    double s[3] = { al * in[0], al * in[1], in[2] };
    unit3(s);
    double theta = 0.0, phi = 0.0;
    if (s[2] < 0.99999) { theta = acos(s[2]); phi = atan2(s[1], s[0]); }
    /* 2 sample P22 */
    double sl[2];
    ggx_sample_visible_11(theta, (double)u[0], (double)u[1], sl);
    /* 3 rotate, 4 unstretch */
    double cp = cos(phi), sp = sin(phi);
    double sx = (cp * sl[0] - sp * sl[1]) * al, sy = (sp * sl[0] + cp * sl[1]) * al;
    /* 5 normal */
    double nrm = 1.0 / sqrt(sx * sx + sy * sy + 1.0);
    double m[3] = { -sx * nrm, -sy * nrm, nrm };
    double c = in[0] * m[0] + in[1] * m[1] + in[2] * m[2];
    double out[3] = { 2.0 * c * m[0] - in[0], 2.0 * c * m[1] - in[1], 2.0 * c * m[2] - in[2] };
    if (!(out[2] > 0.0) || !(c > 0.0)) return;
    double D = ggx_D(al, m);
    double p = D * ggx_G1(al, in, m) / (4.0 * in[2]);
    if (!(p > 0.0)) return;
    float wof[3] = { (float)out[0], (float)out[1], (float)out[2] };
    if (!(wof[2] > 0.0f)) return;
    wo[0] = wof[0]; wo[1] = wof[1]; wo[2] = wof[2];
    *pdf = (float)p;
    double G1o = ggx_G1(al, out, m);
    for (int ch = 0; ch < 3; ++ch) weight[ch] = (float)(fresnel_conductor(c, g->eta[ch], g->k[ch]) * G1o);
}
void orc_ggx_eval_batch(const orc_ggx *g, const float *wi, const float *wo, size_t n, float *rgb)
{
    for (size_t i = 0; i < n; ++i) orc_ggx_eval(g, wi + 3 * i, wo + 3 * i, rgb + 3 * i);
}
void orc_ggx_pdf_batch(const orc_ggx *g, const float *wi, const float *wo, size_t n, float *pdf)
{
    for (size_t i = 0; i < n; ++i) pdf[i] = orc_ggx_pdf(g, wi + 3 * i, wo + 3 * i);
}
void orc_ggx_sample_batch(const orc_ggx *g, const float *wi, const float *u, size_t n, float *wo, float *pdf, float *weight)
{
    for (size_t i = 0; i < n; ++i) orc_ggx_sample(g, wi + 3 * i, u + 2 * i, wo + 3 * i, pdf + i, weight + 3 * i);
}

/* ------------------------------------------------------------------------------------------
 * Synthetic inputs: pair i -> splitmix64(seed ^ counter) -> 6 uniforms ->
 * wi, wo uniform on the upper hemisphere (z = u1, phi = 2 pi u2), u = (u5, u6).  Integer and
 * IEEE-f32 operations only, so the HIP generator produces the same bits.
 * ---------------------------------------------------------------------------------------- */
static uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static void hemisphere_dir(uint64_t r, float d[3])
{
    const float TWO_NEG24 = 0x1p-24f, STEP = 0x1.921fb6p-22f; /* 2*pi / 2^24 */
    float z = (float)(((r >> 41) << 1) | 1u) * TWO_NEG24;     /* odd 24-bit integer: z in (0,1) */
    int32_t k = (int32_t)((r >> 8) & 0xFFFFFFu);               /* phi = 2 pi k / 2^24 */
    int32_t q = (k + (1 << 21)) >> 22;                         /* nearest quadrant */
    int32_t j = k - (q << 22);
    float s, c;
    sincos_quarter_f32((float)j * STEP, &s, &c);
    float cs, sn;
    switch (q & 3) {
        case 0:  cs = c;  sn = s;  break;
        case 1:  cs = -s; sn = c;  break;
        case 2:  cs = -c; sn = -s; break;
        default: cs = s;  sn = -c; break;
    }
    float rr = sqrtf(fmaf(-z, z, 1.0f));
    d[0] = rr * cs; d[1] = rr * sn; d[2] = z;
}
void orc_generate_pairs(uint64_t seed, uint64_t first, size_t n, float *wi, float *wo, float *u)
{
    for (size_t t = 0; t < n; ++t) {
        uint64_t i = first + t;
        uint64_t r0 = mix64(seed ^ (3 * i)), r1 = mix64(seed ^ (3 * i + 1)), r2 = mix64(seed ^ (3 * i + 2));
        hemisphere_dir(r0, wi + 3 * t);
        hemisphere_dir(r1, wo + 3 * t);
        u[2 * t + 0] = (float)(r2 >> 40) * 0x1p-24f;
        u[2 * t + 1] = (float)((r2 >> 16) & 0xFFFFFFu) * 0x1p-24f;
    }
}
void orc_generate_materials(uint64_t seed, uint64_t first, size_t n, int n_materials, int32_t *mat)
{
    for (size_t t = 0; t < n; ++t) {
        uint64_t r = mix64((seed ^ 0x4D41544552494131ULL) + (first + t));
        mat[t] = (int32_t)(((r >> 32) * (uint64_t)n_materials) >> 32);
    }
}

/* ------------------------------------------------------------------------------------------
 * CPU baseline (SURVEY.md §8d "CPU baseline beside it"): the scalar oracle behind a
 * Mitsuba-0.6-style virtual BSDF — one indirect call per pair for eval, pdf and sample.
 * ---------------------------------------------------------------------------------------- */
static void v_merl_eval(const orc_bsdf *b, const float wi[3], const float wo[3], float rgb[3]) { orc_eval(&b->table, &b->opts, wi, wo, rgb); }
static float v_merl_pdf(const orc_bsdf *b, const float wi[3], const float wo[3]) { (void)b; return orc_pdf(wi, wo); }
static void v_merl_sample(const orc_bsdf *b, const float wi[3], const float u[2], float wo[3], float *pdf, float w[3]) { orc_sample(&b->table, &b->opts, wi, u, wo, pdf, w); }
static void v_ggx_eval(const orc_bsdf *b, const float wi[3], const float wo[3], float rgb[3]) { orc_ggx_eval(&b->ggx, wi, wo, rgb); }
static float v_ggx_pdf(const orc_bsdf *b, const float wi[3], const float wo[3]) { return orc_ggx_pdf(&b->ggx, wi, wo); }
static void v_ggx_sample(const orc_bsdf *b, const float wi[3], const float u[2], float wo[3], float *pdf, float w[3]) { orc_ggx_sample(&b->ggx, wi, u, wo, pdf, w); }
static const orc_bsdf_vtbl MERL_VTBL = { v_merl_eval, v_merl_pdf, v_merl_sample };
static const orc_bsdf_vtbl GGX_VTBL = { v_ggx_eval, v_ggx_pdf, v_ggx_sample };

void orc_bsdf_init_merl(orc_bsdf *b, const double *planar, const orc_opts *o)
{
    memset(b, 0, sizeof *b);
    b->vtbl = &MERL_VTBL;
    orc_merl_table(&b->table, planar);
    b->opts = *o;
}
void orc_bsdf_init_ggx(orc_bsdf *b, const orc_ggx *g)
{
    memset(b, 0, sizeof *b);
    b->vtbl = &GGX_VTBL;
    b->ggx = *g;
}

typedef struct bench_job {
    const orc_bsdf *b; uint64_t seed, first; size_t n; int with_sample; double checksum;
} bench_job;

static void *bench_worker(void *arg)
{
    bench_job *j = (bench_job *)arg;
    const orc_bsdf *b = j->b;
    double acc = 0.0;
    enum { CH = 1024 };
    float wi[3 * CH], wo[3 * CH], u[2 * CH];
    for (size_t off = 0; off < j->n; off += CH) {
        size_t m = j->n - off < CH ? j->n - off : CH;
        orc_generate_pairs(j->seed, j->first + off, m, wi, wo, u);
        for (size_t i = 0; i < m; ++i) {
            float rgb[3], wo2[3], pdf2, w[3];
            b->vtbl->eval(b, wi + 3 * i, wo + 3 * i, rgb);
            acc += rgb[0] + rgb[1] + rgb[2];
            if (j->with_sample) {
                float pdf = b->vtbl->pdf(b, wi + 3 * i, wo + 3 * i);
                b->vtbl->sample(b, wi + 3 * i, u + 2 * i, wo2, &pdf2, w);
                acc += pdf + wo2[0] + wo2[1] + wo2[2] + pdf2 + w[0] + w[1] + w[2];
            }
        }
    }
    j->checksum = acc;
    return NULL;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static double bench_run(const orc_bsdf *b, uint64_t seed, uint64_t first, size_t n, int n_threads, int with_sample, double *checksum)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    bench_job *jobs = (bench_job *)malloc(sizeof(bench_job) * (size_t)n_threads);
    size_t per = (n + (size_t)n_threads - 1) / (size_t)n_threads;
    /* generation of the inputs is inside the timed region for every thread alike; it is
     * ~3 % of a unit's cost (two polynomial sincos vs. ~10 libm calls per lookup) */
    double t0 = now_s();
    for (int t = 0; t < n_threads; ++t) {
        size_t lo = per * (size_t)t, hi = lo + per > n ? n : lo + per;
        if (lo > n) lo = n;
        jobs[t].b = b; jobs[t].seed = seed; jobs[t].first = first + lo; jobs[t].n = hi - lo;
        jobs[t].with_sample = with_sample; jobs[t].checksum = 0.0;
        pthread_create(&th[t], NULL, bench_worker, &jobs[t]);
    }
    double acc = 0.0;
    for (int t = 0; t < n_threads; ++t) { pthread_join(th[t], NULL); acc += jobs[t].checksum; }
    double t1 = now_s();
    free(th); free(jobs);
    if (checksum) *checksum = acc;
    return t1 - t0;
}
double orc_bench_eval_sample(const orc_bsdf *b, uint64_t seed, uint64_t first, size_t n, int n_threads, double *checksum)
{
    return bench_run(b, seed, first, n, n_threads, 1, checksum);
}
double orc_bench_eval(const orc_bsdf *b, uint64_t seed, uint64_t first, size_t n, int n_threads, double *checksum)
{
    return bench_run(b, seed, first, n, n_threads, 0, checksum);
}
