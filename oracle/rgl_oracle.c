/* rgl_oracle.c — TEST INFRASTRUCTURE: CPU restatement of the adaptive-parameterisation measured BSDF that the RGL material
 * database's *.bsdf files describe (Dupuy & Jakob 2018, "An adaptive parameterization for efficient material acquisition and
 * rendering"; upstream Mitsuba 3's stock `measured` plugin evaluates it).  SURVEY.md §8f item 3 names the format as a "next" row.
 *
 * PARITY UNPINNED: neither the paper's code nor upstream's plugin nor a *.bsdf file exists in this container; this file
 * restates the published model from its description — the piecewise-bilinear 2-D distribution with parameter interpolation
 * ("Marginal2D" upstream), its sample / invert / eval, and the BSDF's eval / sample / pdf on top — and is pinned only by
 * self-consistency KATs (tests/test_rgl_cpu.py: invert(sample(u)) == u, densities integrate to 1, constant tables are the
 * identity warp, weight == eval / pdf, chi-square).  Only tests/, __graft_entry__.smoke() and bench.py's checker legs may use it.
 *
 * Math in f64 on Float tables.  Layout of a warp with parameter dimensions (phi_i, theta_i[, channel]): slices are row-major
 * in the parameter indices; a slice is [ny][nx] nodes, x fastest.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "rgl_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- the 2-D piecewise-bilinear distribution -------------------------------------------------------------------------- */
int rgl_warp_init(rgl_warp *w, int nx, int ny, int n_dim, const int *n_par, const float *const *par, const float *data,
                  int normalize, int build_cdf)
{
    memset(w, 0, sizeof *w);
    if (nx < 2 || ny < 2 || n_dim < 0 || n_dim > RGL_MAX_DIM) return -1;
    w->nx = nx; w->ny = ny; w->n_dim = n_dim;
    w->n_slices = 1;
    for (int d = 0; d < n_dim; ++d) {
        if (n_par[d] < 1) return -1;
        w->n_par[d] = n_par[d];
        w->par[d] = (float *)malloc(sizeof(float) * (size_t)n_par[d]);
        if (!w->par[d]) return -4;
        memcpy(w->par[d], par[d], sizeof(float) * (size_t)n_par[d]);
        w->n_slices *= n_par[d];
    }
    int stride = 1;
    for (int d = n_dim - 1; d >= 0; --d) { w->stride[d] = stride; stride *= n_par[d]; }
    const size_t slice = (size_t)nx * ny;
    w->data = (float *)malloc(sizeof(float) * slice * (size_t)w->n_slices);
    if (!w->data) return -4;
    w->normalized = normalize;
    if (build_cdf) {
        w->marg = (float *)malloc(sizeof(float) * (size_t)(ny - 1) * (size_t)w->n_slices);
        w->cond = (float *)malloc(sizeof(float) * (size_t)ny * (size_t)(nx - 1) * (size_t)w->n_slices);
        if (!w->marg || !w->cond) return -4;
    }
    double *cond = (double *)malloc(sizeof(double) * (size_t)ny * (size_t)(nx - 1));
    double *marg = (double *)malloc(sizeof(double) * (size_t)(ny - 1));
    if (!cond || !marg) { free(cond); free(marg); return -4; }
    for (int s = 0; s < w->n_slices; ++s) {
        const float *src = data + slice * (size_t)s;
        /* conditional: running integral along x of each node row; marginal: running integral along y of the row totals */
        for (int y = 0; y < ny; ++y) {
            double sum = 0.0;
            for (int x = 0; x < nx - 1; ++x) {
                sum += 0.5 * ((double)src[y * nx + x] + (double)src[y * nx + x + 1]);
                cond[y * (nx - 1) + x] = sum;
            }
        }
        double sum = 0.0;
        for (int y = 0; y < ny - 1; ++y) {
            sum += 0.5 * (cond[y * (nx - 1) + nx - 2] + cond[(y + 1) * (nx - 1) + nx - 2]);
            marg[y] = sum;
        }
        const double norm = (normalize && sum > 0.0) ? 1.0 / sum : 1.0;
        for (size_t k = 0; k < slice; ++k) w->data[slice * (size_t)s + k] = (float)((double)src[k] * norm);
        if (build_cdf) {
            for (int k = 0; k < ny * (nx - 1); ++k) w->cond[(size_t)s * (size_t)ny * (size_t)(nx - 1) + (size_t)k] = (float)(cond[k] * norm);
            for (int k = 0; k < ny - 1; ++k) w->marg[(size_t)s * (size_t)(ny - 1) + (size_t)k] = (float)(marg[k] * norm);
        }
    }
    free(cond); free(marg);
    return 0;
}

void rgl_warp_free(rgl_warp *w)
{
    for (int d = 0; d < RGL_MAX_DIM; ++d) free(w->par[d]);
    free(w->data); free(w->marg); free(w->cond);
    memset(w, 0, sizeof *w);
}

/* the 2^n_dim parameter slices around `params` and their weights */
typedef struct { int slice[1 << RGL_MAX_DIM]; double weight[1 << RGL_MAX_DIM]; int n; } slices_t;

static void find_slices(const rgl_warp *w, const double *params, slices_t *out)
{
    out->n = 1; out->slice[0] = 0; out->weight[0] = 1.0;
    for (int d = 0; d < w->n_dim; ++d) {
        const int n = w->n_par[d];
        int i = 0; double t = 0.0;
        if (n > 1) {
            /* largest i in [0, n-2] with par[i] <= p */
            int lo = 0, hi = n - 1;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((double)w->par[d][mid] <= params[d]) lo = mid; else hi = mid; }
            i = lo;
            const double p0 = w->par[d][i], p1 = w->par[d][i + 1];
            t = (params[d] - p0) / (p1 - p0);
            t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        }
        const int m = out->n;
        for (int k = 0; k < m; ++k) {
            out->slice[k + m] = out->slice[k] + (i + (n > 1 ? 1 : 0)) * w->stride[d];
            out->weight[k + m] = out->weight[k] * t;
            out->slice[k] += i * w->stride[d];
            out->weight[k] *= 1.0 - t;
        }
        out->n = 2 * m;
    }
}

static double fetch(const slices_t *s, const float *base, size_t per_slice, size_t index)
{
    double v = 0.0;
    for (int k = 0; k < s->n; ++k)
        if (s->weight[k] != 0.0) v += s->weight[k] * (double)base[(size_t)s->slice[k] * per_slice + index];
    return v;
}

double rgl_warp_eval(const rgl_warp *w, const double pos_in[2], const double *params)
{
    slices_t s; find_slices(w, params, &s);
    double px = pos_in[0] * (w->nx - 1), py = pos_in[1] * (w->ny - 1);
    int ox = (int)px, oy = (int)py;
    ox = ox < 0 ? 0 : (ox > w->nx - 2 ? w->nx - 2 : ox);
    oy = oy < 0 ? 0 : (oy > w->ny - 2 ? w->ny - 2 : oy);
    const double fx = px - ox, fy = py - oy;
    const size_t per = (size_t)w->nx * w->ny, idx = (size_t)oy * w->nx + ox;
    const double v00 = fetch(&s, w->data, per, idx), v10 = fetch(&s, w->data, per, idx + 1);
    const double v01 = fetch(&s, w->data, per, idx + w->nx), v11 = fetch(&s, w->data, per, idx + w->nx + 1);
    const double v = (1.0 - fy) * ((1.0 - fx) * v00 + fx * v10) + fy * ((1.0 - fx) * v01 + fx * v11);
    return w->normalized ? v * (double)(w->nx - 1) * (double)(w->ny - 1) : v;
}

static double safe_sqrt(double x) { return x > 0.0 ? sqrt(x) : 0.0; }

/* invert a linear density c0 -> c1 over [0, 1] given the mass u in units where the patch integral is (c0 + c1) / 2 */
static double invert_linear(double c0, double c1, double u)
{
    const int is_const = fabs(c0 - c1) < 1e-4 * (c0 + c1);
    const double num = is_const ? 2.0 * u : c0 - safe_sqrt(c0 * c0 - 2.0 * u * (c0 - c1));
    const double den = is_const ? c0 + c1 : c0 - c1;
    return den != 0.0 ? num / den : 0.0;
}

void rgl_warp_sample(const rgl_warp *w, const double u_in[2], const double *params, double pos[2], double *pdf)
{
    slices_t s; find_slices(w, params, &s);
    const int nx = w->nx, ny = w->ny;
    const size_t per_m = (size_t)(ny - 1), per_c = (size_t)ny * (size_t)(nx - 1), per_d = (size_t)nx * ny;
    double ux = u_in[0] < 0.0 ? 0.0 : (u_in[0] > 1.0 ? 1.0 : u_in[0]);
    double uy = u_in[1] < 0.0 ? 0.0 : (u_in[1] > 1.0 ? 1.0 : u_in[1]);
    /* row: first index whose marginal cdf is not below uy */
    int lo = 0, hi = ny - 2;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (fetch(&s, w->marg, per_m, (size_t)mid) < uy) lo = mid + 1; else hi = mid; }
    const int row = lo;
    if (row > 0) uy -= fetch(&s, w->marg, per_m, (size_t)(row - 1));
    const double r0 = fetch(&s, w->cond, per_c, (size_t)row * (nx - 1) + (nx - 2));
    const double r1 = fetch(&s, w->cond, per_c, (size_t)(row + 1) * (nx - 1) + (nx - 2));
    double y = invert_linear(r0, r1, uy);
    y = y < 0.0 ? 0.0 : (y > 1.0 ? 1.0 : y);
    /* column, in the conditional cdf interpolated between the two node rows */
    ux *= (1.0 - y) * r0 + y * r1;
    lo = 0; hi = nx - 2;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const double c = (1.0 - y) * fetch(&s, w->cond, per_c, (size_t)row * (nx - 1) + mid) + y * fetch(&s, w->cond, per_c, (size_t)(row + 1) * (nx - 1) + mid);
        if (c < ux) lo = mid + 1; else hi = mid;
    }
    const int col = lo;
    if (col > 0)
        ux -= (1.0 - y) * fetch(&s, w->cond, per_c, (size_t)row * (nx - 1) + col - 1) + y * fetch(&s, w->cond, per_c, (size_t)(row + 1) * (nx - 1) + col - 1);
    const size_t idx = (size_t)row * nx + col;
    const double v00 = fetch(&s, w->data, per_d, idx), v10 = fetch(&s, w->data, per_d, idx + 1);
    const double v01 = fetch(&s, w->data, per_d, idx + nx), v11 = fetch(&s, w->data, per_d, idx + nx + 1);
    const double c0 = (1.0 - y) * v00 + y * v01, c1 = (1.0 - y) * v10 + y * v11;
    double x = invert_linear(c0, c1, ux);
    x = x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x);
    pos[0] = ((double)col + x) / (double)(nx - 1);
    pos[1] = ((double)row + y) / (double)(ny - 1);
    *pdf = ((1.0 - x) * c0 + x * c1) * (double)(nx - 1) * (double)(ny - 1);
}

void rgl_warp_invert(const rgl_warp *w, const double pos_in[2], const double *params, double u[2], double *pdf)
{
    slices_t s; find_slices(w, params, &s);
    const int nx = w->nx, ny = w->ny;
    const size_t per_m = (size_t)(ny - 1), per_c = (size_t)ny * (size_t)(nx - 1), per_d = (size_t)nx * ny;
    const double px = pos_in[0] * (nx - 1), py = pos_in[1] * (ny - 1);
    int col = (int)px, row = (int)py;
    col = col < 0 ? 0 : (col > nx - 2 ? nx - 2 : col);
    row = row < 0 ? 0 : (row > ny - 2 ? ny - 2 : row);
    const double x = px - col, y = py - row;
    const size_t idx = (size_t)row * nx + col;
    const double v00 = fetch(&s, w->data, per_d, idx), v10 = fetch(&s, w->data, per_d, idx + 1);
    const double v01 = fetch(&s, w->data, per_d, idx + nx), v11 = fetch(&s, w->data, per_d, idx + nx + 1);
    const double c0 = (1.0 - y) * v00 + y * v01, c1 = (1.0 - y) * v10 + y * v11;
    *pdf = ((1.0 - x) * c0 + x * c1) * (double)(nx - 1) * (double)(ny - 1);
    double sx = x * (c0 + 0.5 * x * (c1 - c0));
    if (col > 0)
        sx += (1.0 - y) * fetch(&s, w->cond, per_c, (size_t)row * (nx - 1) + col - 1) + y * fetch(&s, w->cond, per_c, (size_t)(row + 1) * (nx - 1) + col - 1);
    const double r0 = fetch(&s, w->cond, per_c, (size_t)row * (nx - 1) + (nx - 2));
    const double r1 = fetch(&s, w->cond, per_c, (size_t)(row + 1) * (nx - 1) + (nx - 2));
    const double tot = (1.0 - y) * r0 + y * r1;
    u[0] = tot > 0.0 ? sx / tot : 0.0;
    double sy = y * (r0 + 0.5 * y * (r1 - r0));
    if (row > 0) sy += fetch(&s, w->marg, per_m, (size_t)(row - 1));
    u[1] = sy;
}

/* ---- the BSDF ---------------------------------------------------------------------------------------------------------- */
static double elevation(const double d[3])
{
    /* 2 asin(|d - z| / 2): acos(d.z) without its cancellation near the pole */
    const double dx = d[0], dy = d[1], dz = d[2] - 1.0;
    const double h = 0.5 * sqrt(dx * dx + dy * dy + dz * dz);
    return 2.0 * asin(h > 1.0 ? 1.0 : h);
}
static double theta2u(double t) { return sqrt(t * (2.0 / M_PI)); }
static double phi2u(double p) { return (p + M_PI) * (0.5 / M_PI); }
static double u2theta(double u) { return u * u * (M_PI / 2.0); }
static double u2phi(double u) { return (2.0 * u - 1.0) * M_PI; }

static int unit3(double v[3])
{
    const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (!(n > 0.0)) return 0;
    v[0] /= n; v[1] /= n; v[2] /= n;
    return 1;
}

/* the measured values at `n` settings of the third parameter: the channel number 0, 1, 2 of an RGB file (third == NULL), or n
 * wavelengths of a spectral file — interpolated linearly between the file's wavelength nodes, clamped outside them, as upstream's
 * spectral variants evaluate `spectra` with the wavelength as the third interpolated parameter */
static void spectrum(const rgl_bsdf *b, const double sample[2], double phi_i, double theta_i, int n, const float *third, double *out)
{
    for (int c = 0; c < n; ++c) {
        const double p[3] = { phi_i, theta_i, third ? (double)third[c] : (double)c };
        out[c] = rgl_warp_eval(&b->rgb, sample, p);
        if (out[c] < 0.0) out[c] = 0.0;
    }
}

/* Symmetry-reduced anisotropic files hold phi_i in [-pi, 0] (reduction 2: the sample looks the same turned by 180 degrees) or in
 * [-pi, -pi/2] (reduction 4: it also has two mirror planes).  Both directions of a pair are mapped into the stored part with the
 * signs of wi: x and y are negated together when wi.y is not negative (2), resp. x when wi.x and y when wi.y is not negative (4)
 * — "not negative" by the sign BIT, +0 counts as positive.  flip[] receives the two factors (+-1) so that sample() can map the
 * direction it draws back. */
static void reduce_pair(const rgl_bsdf *b, double wi[3], double wo[3], double flip[2])
{
    flip[0] = flip[1] = 1.0;
    if (b->reduction < 2) return;
    const double sy = signbit(wi[1]) ? 1.0 : -1.0;
    const double sx = b->reduction == 4 ? (signbit(wi[0]) ? 1.0 : -1.0) : sy;
    flip[0] = sx; flip[1] = sy;
    wi[0] *= sx; wi[1] *= sy; wo[0] *= sx; wo[1] *= sy;
}

/* eval / pdf from the normalised incident direction (already in the stored part of the azimuth) and the UNNORMALISED half vector
 * m = wi + wo.  rgl_eval_pdf below is this after its prelude; the tests call it directly to measure conditioning: for a
 * near-mirror pair m's transverse part is the difference of two normalisations and carries their rounding errors (a few 1e-16
 * absolute on a length that can be 1e-9), so they evaluate it over the box of half vectors the f64 arithmetic can land on. */
#define RGL_MAX_VALUES 4096
/* n values (third == NULL: n = 3 RGB channels; else n wavelengths) and / or the pdf */
static void eval_pdf_half_n(const rgl_bsdf *b, const double wi[3], const double m_in[3], int n, const float *third, float *values, float *pdf_out)
{
    for (int c = 0; c < n; ++c) values[c] = 0.0f;
    if (pdf_out) *pdf_out = 0.0f;
    double m[3] = { m_in[0], m_in[1], m_in[2] };
    if (!unit3(m) || n > RGL_MAX_VALUES) return;
    const double theta_i = elevation(wi), phi_i = atan2(wi[1], wi[0]);
    const double theta_m = elevation(m), phi_m = atan2(m[1], m[0]);
    const double params[2] = { phi_i, theta_i };
    const double u_wi[2] = { theta2u(theta_i), phi2u(phi_i) };
    double u_m[2] = { theta2u(theta_m), phi2u(b->isotropic ? phi_m - phi_i : phi_m) };
    u_m[1] -= floor(u_m[1]);
    double sample[2], vndf_pdf;
    rgl_warp_invert(&b->vndf, u_m, params, sample, &vndf_pdf);
    double spec[RGL_MAX_VALUES];
    spectrum(b, sample, phi_i, theta_i, n, third, spec);
    double scale = 1.0;
    if (b->jacobian) scale = rgl_warp_eval(&b->ndf, u_m, params) / (4.0 * rgl_warp_eval(&b->sigma, u_wi, params));
    for (int c = 0; c < n; ++c) values[c] = (float)(spec[c] * scale);
    if (pdf_out) {
        const double lum_pdf = rgl_warp_eval(&b->luminance, sample, params);
        const double sin_theta_m = sqrt(m[0] * m[0] + m[1] * m[1]);
        const double jac = fmax(2.0 * M_PI * M_PI * u_m[0] * sin_theta_m, 1e-6) * 4.0 * (wi[0] * m[0] + wi[1] * m[1] + wi[2] * m[2]);
        *pdf_out = (float)(vndf_pdf * lum_pdf / jac);
    }
}

/* eval / pdf from the normalised incident direction (already in the stored part of the azimuth) and the UNNORMALISED half vector
 * m = wi + wo.  rgl_eval_pdf below is this after its prelude; the tests call it directly to measure conditioning: for a
 * near-mirror pair m's transverse part is the difference of two normalisations and carries their rounding errors (a few 1e-16
 * absolute on a length that can be 1e-9), so they evaluate it over the box of half vectors the f64 arithmetic can land on. */
void rgl_eval_pdf_half(const rgl_bsdf *b, const double wi[3], const double m_in[3], float rgb[3], float *pdf_out)
{
    eval_pdf_half_n(b, wi, m_in, 3, NULL, rgb, pdf_out);
}

/* the prelude alone: wi, wo in the stored part of the azimuth and normalised, m = wi + wo; 0 when the pair evaluates to zero */
int rgl_half_vector(const rgl_bsdf *b, const float wi_f[3], const float wo_f[3], double wi[3], double m[3])
{
    if (!(wi_f[2] > 0.0f) || !(wo_f[2] > 0.0f)) return 0;
    double wo[3] = { wo_f[0], wo_f[1], wo_f[2] }, flip[2];
    wi[0] = wi_f[0]; wi[1] = wi_f[1]; wi[2] = wi_f[2];
    reduce_pair(b, wi, wo, flip);
    if (!unit3(wi) || !unit3(wo)) return 0;
    m[0] = wi[0] + wo[0]; m[1] = wi[1] + wo[1]; m[2] = wi[2] + wo[2];
    return 1;
}

void rgl_eval_pdf(const rgl_bsdf *b, const float wi_f[3], const float wo_f[3], float rgb[3], float *pdf_out)
{
    rgb[0] = rgb[1] = rgb[2] = 0.0f;
    if (pdf_out) *pdf_out = 0.0f;
    double wi[3], m[3];
    if (!rgl_half_vector(b, wi_f, wo_f, wi, m)) return;
    rgl_eval_pdf_half(b, wi, m, rgb, pdf_out);
}

/* a spectral file: the values at W wavelengths (wl == NULL: at the file's own wavelength nodes, W = their number) */
void rgl_eval_pdf_spectral(const rgl_bsdf *b, const float wi_f[3], const float wo_f[3], const float *wl, int W, float *values, float *pdf_out)
{
    for (int c = 0; c < W; ++c) values[c] = 0.0f;
    if (pdf_out) *pdf_out = 0.0f;
    double wi[3], m[3];
    if (!rgl_half_vector(b, wi_f, wo_f, wi, m)) return;
    eval_pdf_half_n(b, wi, m, W, wl ? wl : b->rgb.par[2], values, pdf_out);
}

static void sample_n(const rgl_bsdf *b, const float wi_f[3], const float u[2], int n, const float *third, float wo_out[3], float *pdf_out, float *weight);

void rgl_sample_spectral(const rgl_bsdf *b, const float wi_f[3], const float u[2], const float *wl, int W, float wo_out[3], float *pdf_out, float *weight)
{
    sample_n(b, wi_f, u, W, wl ? wl : b->rgb.par[2], wo_out, pdf_out, weight);
}

void rgl_sample(const rgl_bsdf *b, const float wi_f[3], const float u[2], float wo_out[3], float *pdf_out, float weight[3])
{
    sample_n(b, wi_f, u, 3, NULL, wo_out, pdf_out, weight);
}

static void sample_n(const rgl_bsdf *b, const float wi_f[3], const float u[2], int n, const float *third, float wo_out[3], float *pdf_out, float *weight)
{
    wo_out[0] = wo_out[1] = wo_out[2] = 0.0f; *pdf_out = 0.0f;
    for (int c = 0; c < n; ++c) weight[c] = 0.0f;
    if (n > RGL_MAX_VALUES) return;
    if (!(wi_f[2] > 0.0f)) return;
    double wi[3] = { wi_f[0], wi_f[1], wi_f[2] };
    double unused[3] = { 0.0, 0.0, 1.0 }, flip[2];
    reduce_pair(b, wi, unused, flip);
    if (!unit3(wi)) return;
    const double theta_i = elevation(wi), phi_i = atan2(wi[1], wi[0]);
    const double params[2] = { phi_i, theta_i };
    const double u_wi[2] = { theta2u(theta_i), phi2u(phi_i) };
    const double s_in[2] = { (double)u[1], (double)u[0] };
    double sample[2], lum_pdf, u_m[2], ndf_pdf;
    rgl_warp_sample(&b->luminance, s_in, params, sample, &lum_pdf);
    rgl_warp_sample(&b->vndf, sample, params, u_m, &ndf_pdf);
    double phi_m = u2phi(u_m[1]);
    const double theta_m = u2theta(u_m[0]);
    if (b->isotropic) phi_m += phi_i;
    const double st = sin(theta_m), ct = cos(theta_m);
    const double m[3] = { cos(phi_m) * st, sin(phi_m) * st, ct };
    const double c = wi[0] * m[0] + wi[1] * m[1] + wi[2] * m[2];
    /* the direction drawn in the stored part of the azimuth goes back through the same sign flips */
    const double wo[3] = { (2.0 * c * m[0] - wi[0]) * flip[0], (2.0 * c * m[1] - wi[1]) * flip[1], 2.0 * c * m[2] - wi[2] };
    const float wof[3] = { (float)wo[0], (float)wo[1], (float)wo[2] };
    if (!(wof[2] > 0.0f) || !(c > 0.0)) return;
    /* report what eval / pdf say AT the Float direction returned, so that pdf(wi, sample.wo) == sample.pdf and weight == eval / pdf */
    float f[RGL_MAX_VALUES], p;
    {
        double wi_d[3], m_d[3];
        for (int c = 0; c < n; ++c) f[c] = 0.0f;
        p = 0.0f;
        if (rgl_half_vector(b, wi_f, wof, wi_d, m_d)) eval_pdf_half_n(b, wi_d, m_d, n, third, f, &p);       /* = rgl_eval_pdf(b, wi_f, wof, ...) */
    }
    (void)u_wi; (void)lum_pdf; (void)ndf_pdf;
    if (!(p > 0.0f)) return;
    wo_out[0] = wof[0]; wo_out[1] = wof[1]; wo_out[2] = wof[2];
    *pdf_out = p;
    for (int c = 0; c < n; ++c) weight[c] = f[c] / p;
}

void rgl_eval_pdf_spectral_batch(const rgl_bsdf *b, const float *wi, const float *wo, const float *wl, int W, size_t n, float *values, float *pdf)
{
    for (size_t i = 0; i < n; ++i) rgl_eval_pdf_spectral(b, wi + 3 * i, wo + 3 * i, wl ? wl + (size_t)W * i : NULL, W, values + (size_t)W * i, pdf ? pdf + i : NULL);
}
void rgl_sample_spectral_batch(const rgl_bsdf *b, const float *wi, const float *u, const float *wl, int W, size_t n, float *wo, float *pdf, float *weight)
{
    for (size_t i = 0; i < n; ++i) rgl_sample_spectral(b, wi + 3 * i, u + 2 * i, wl ? wl + (size_t)W * i : NULL, W, wo + 3 * i, pdf + i, weight + (size_t)W * i);
}

void rgl_eval_pdf_batch(const rgl_bsdf *b, const float *wi, const float *wo, size_t n, float *rgb, float *pdf)
{
    for (size_t i = 0; i < n; ++i) rgl_eval_pdf(b, wi + 3 * i, wo + 3 * i, rgb + 3 * i, pdf ? pdf + i : NULL);
}
void rgl_sample_batch(const rgl_bsdf *b, const float *wi, const float *u, size_t n, float *wo, float *pdf, float *weight)
{
    for (size_t i = 0; i < n; ++i) rgl_sample(b, wi + 3 * i, u + 2 * i, wo + 3 * i, pdf + i, weight + 3 * i);
}

int rgl_bsdf_init(rgl_bsdf *b, int n_phi, int n_theta, const float *phi_i, const float *theta_i, int res_ndf_x, int res_ndf_y, const float *ndf,
                  int res_sigma_x, int res_sigma_y, const float *sigma, int res_x, int res_y, const float *vndf, const float *luminance,
                  const float *rgb, int jacobian)
{
    return rgl_bsdf_init_spectral(b, n_phi, n_theta, phi_i, theta_i, res_ndf_x, res_ndf_y, ndf, res_sigma_x, res_sigma_y, sigma, res_x, res_y, vndf, luminance,
                                  0, NULL, rgb, jacobian);
}

/* n_wavelengths == 0: an RGB file (values [n_phi][n_theta][3][res_y][res_x]); else a spectral one: `spectra`
 * [n_phi][n_theta][n_wavelengths][res_y][res_x] over the ascending grid `wavelengths` */
int rgl_bsdf_init_spectral(rgl_bsdf *b, int n_phi, int n_theta, const float *phi_i, const float *theta_i, int res_ndf_x, int res_ndf_y, const float *ndf,
                           int res_sigma_x, int res_sigma_y, const float *sigma, int res_x, int res_y, const float *vndf, const float *luminance,
                           int n_wavelengths, const float *wavelengths, const float *rgb, int jacobian)
{
    memset(b, 0, sizeof *b);
    b->isotropic = n_phi <= 2;
    b->jacobian = jacobian;
    b->reduction = 1;
    if (!b->isotropic) {
        const double span = (double)phi_i[n_phi - 1] - (double)phi_i[0];
        b->reduction = span > 0.0 ? (int)floor(2.0 * M_PI / span + 0.5) : 0;
        if (b->reduction != 1 && b->reduction != 2 && b->reduction != 4) return -1;
    }
    const int np2[2] = { n_phi, n_theta };
    const float *par2[2] = { phi_i, theta_i };
    const float chan[3] = { 0.f, 1.f, 2.f };
    const int np3[3] = { n_phi, n_theta, n_wavelengths > 0 ? n_wavelengths : 3 };
    const float *par3[3] = { phi_i, theta_i, n_wavelengths > 0 ? wavelengths : chan };
    b->n_wavelengths = n_wavelengths > 0 ? n_wavelengths : 0;
    int rc = rgl_warp_init(&b->ndf, res_ndf_x, res_ndf_y, 0, NULL, NULL, ndf, 0, 0);
    if (!rc) rc = rgl_warp_init(&b->sigma, res_sigma_x, res_sigma_y, 0, NULL, NULL, sigma, 0, 0);
    if (!rc) rc = rgl_warp_init(&b->vndf, res_x, res_y, 2, np2, par2, vndf, 1, 1);
    if (!rc) rc = rgl_warp_init(&b->luminance, res_x, res_y, 2, np2, par2, luminance, 1, 1);
    if (!rc) rc = rgl_warp_init(&b->rgb, res_x, res_y, 3, np3, par3, rgb, 0, 0);
    if (rc) rgl_bsdf_free(b);
    return rc;
}

void rgl_bsdf_free(rgl_bsdf *b)
{
    rgl_warp_free(&b->ndf); rgl_warp_free(&b->sigma); rgl_warp_free(&b->vndf); rgl_warp_free(&b->luminance); rgl_warp_free(&b->rgb);
}
