/*
 * merl_oracle.h — CPU oracle for the MERL / customized_measurement BSDF hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the timed CPU baseline.  The product (libmerl_hip.so) never links it.
 *
 * PARITY UNPINNED.  /root/reference holds no source for this path (README.md:1 names the
 * plugins; mitsuba/ and mitsuba3/ are empty gitlinks), and ships no golden
 * vectors.  This file is therefore a plain-C, double-precision restatement of the PUBLIC
 * algorithm the README names — the MERL distribution's BRDFRead lookup (Matusik et al. 2003)
 * under upstream Mitsuba 0.6 / Mitsuba 3 BSDF conventions — as specified in SURVEY.md
 * Appendix A (A.1 file format, A.2 half/diff transform, A.3 index maps, A.4 lookup,
 * A.5 Mitsuba conventions, A.6 GGX rough conductor).  It is pinned by analytic known-answer
 * tests and an independent numpy restatement (tests/), not by the reference.
 */
#ifndef MERL_ORACLE_H
#define MERL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* MERL grid */
#define ORC_MERL_N_TH 90
#define ORC_MERL_N_TD 90
#define ORC_MERL_N_PD 180
#define ORC_MERL_N (ORC_MERL_N_TH * ORC_MERL_N_TD * ORC_MERL_N_PD) /* 1,458,000 */

enum { ORC_LOOKUP_NEAREST = 0, ORC_LOOKUP_TRILINEAR = 1 };
enum { ORC_NODE_INTEGER = 0, ORC_NODE_CENTER = 1 };   /* trilinear node position: i or i+1/2 */
enum { ORC_DISK_MITSUBA06 = 0, ORC_DISK_MITSUBA3 = 1 }; /* concentric-disk flavour (A.5) */

/* Table parameterisation ("generalise table dims/parameterisation"):
 *   HALF_DIFF      (theta_h sqrt-warped, theta_d, phi_d mod pi)  — MERL, the default;
 *   STANDARD       (theta_i, theta_o, |phi_o - phi_i| in [0, pi]) — an isotropic, bilaterally symmetric gonioreflectometer
 *                  grid: all three axes linear in the angle, the azimuth axis CLAMPED (0 and pi are different samples);
 *   STANDARD_FULL  (theta_i, theta_o, phi_o - phi_i in [0, 2 pi)) — isotropic without the mirror symmetry, azimuth PERIODIC.
 * Angles of the standard forms come from cancellation-free atan2 expressions (this is the build's own definition, not
 * BRDFRead's):  theta = atan2(|v_xy|, v_z),  dphi = atan2(wi_x wo_y - wi_y wo_x, wi_x wo_x + wi_y wo_y)  (0 when either
 * direction is the normal).  Nearest lookup truncates, trilinear uses the node convention of orc_opts, as for MERL. */
enum { ORC_PARAM_HALF_DIFF = 0, ORC_PARAM_STANDARD = 1, ORC_PARAM_STANDARD_FULL = 2 };

/* does the plugin's eval() multiply the BRDF by cos(theta_o)?  Upstream's convention says yes (the
 * default); OMITTED returns f alone — from eval() and from the eval() inside sample()'s weight (weight == eval / pdf stays true). */
enum { ORC_COSINE_INCLUDED = 0, ORC_COSINE_OMITTED = 1 };
/* what a negative stored value (MERL's marker for a sample that was not measured) does to a lookup:
 *   CLAMP         it counts as 0 (the default; interpolation runs on the clamped values);
 *   KEEP          it is used as it is stored (BRDFRead itself: it only prints a warning) — eval() can come out negative;
 *   RENORMALISE   it is left out: a trilinear lookup blends the valid corners only and divides by their weight,
 *                 sum_k w_k v_k [v_k >= 0] / sum_k w_k [v_k >= 0] per channel (0 when no corner is valid); nearest: 0.
 * The sampling marginals of §8f item 2 are built from clamped values under every setting (a density needs a non-negative mass). */
enum { ORC_NEGATIVE_CLAMP = 0, ORC_NEGATIVE_KEEP = 1, ORC_NEGATIVE_RENORMALISE = 2 };

typedef struct orc_opts {
    int lookup;       /* ORC_LOOKUP_* */
    int node;         /* ORC_NODE_*   */
    int disk_map;     /* ORC_DISK_*   */
    int cosine;       /* ORC_COSINE_*   (0: the default) */
    int negative;     /* ORC_NEGATIVE_* (0: the default) */
} orc_opts;

/* A measured table in MERL parameterisation with free dims (customized_measurement = same
 * layout, other dims/scales).  data is planar: channel c at data + c*n_th*n_td*n_pd, and
 * inside a plane ind = i_pd + n_pd*(i_td + n_td*i_th)  (A.1). */
typedef struct orc_table {
    int n_th, n_td, n_pd;         /* axis 0, 1, 2 (theta_h, theta_d, phi_d for MERL; theta_i, theta_o, dphi for the standard forms) */
    const double *data;
    double scale[3];
    int param;                    /* ORC_PARAM_* */
} orc_table;

/* ---- a1: file format ---- */
int  orc_read_table(const char *path, int require_merl_dims, double **out_data, int dims[3]);
int  orc_write_table(const char *path, const double *planar, const int dims[3]);
void orc_free(void *p);
void orc_merl_table(orc_table *t, const double *planar);  /* dims 90/90/180, MERL scales */

/* ---- a2: half/diff transform on unit f64 vectors (Rodrigues form, as BRDFRead) ---- */
void orc_half_diff(const double in[3], const double out[3],
                   double *theta_half, double *phi_half, double *theta_diff, double *phi_diff);

/* the three lookup angles of a direction pair under the table's parameterisation (unit f64 vectors) */
void orc_standard_angles(const double in[3], const double out[3], double *theta_i, double *theta_o, double *dphi);
void orc_table_angles(const orc_table *t, const double in[3], const double out[3], double angles[3]);

/* ---- a3: index maps (param-aware: the names are MERL's, the axes are the table's) ---- */
int orc_theta_half_index(const orc_table *t, double theta_half);
int orc_theta_diff_index(const orc_table *t, double theta_diff);
int orc_phi_diff_index(const orc_table *t, double phi_diff);
/* continuous (pre-truncation) coordinates; phi_diff folded into [0,pi] first */
void orc_coords(const orc_table *t, double theta_half, double theta_diff, double phi_diff,
                double *x_th, double *x_td, double *x_pd);

/* ---- a4: lookup (scaled; negatives per orc_opts.negative, clamped to 0 by default) ---- */
void orc_lookup(const orc_table *t, const orc_opts *o,
                double theta_half, double theta_diff, double phi_diff, double rgb[3]);

/* ---- a5..a7: Mitsuba-convention eval / pdf / sample on f32 directions ---- */
void  orc_eval(const orc_table *t, const orc_opts *o, const float wi[3], const float wo[3], float rgb[3]);
float orc_pdf(const float wi[3], const float wo[3]);
void  orc_square_to_cosine_hemisphere(int disk_map, const float u[2], float wo[3]);
void  orc_sample(const orc_table *t, const orc_opts *o, const float wi[3], const float u[2],
                 float wo[3], float *pdf, float weight[3]);

/* batches (AoS f32: xyzxyz…, uvuv…) */
void orc_eval_batch(const orc_table *t, const orc_opts *o, const float *wi, const float *wo,
                    size_t n, float *rgb);
void orc_pdf_batch(const float *wi, const float *wo, size_t n, float *pdf);
void orc_sample_batch(const orc_table *t, const orc_opts *o, const float *wi, const float *u,
                      size_t n, float *wo, float *pdf, float *weight);
/* mixed materials: tables[mat[i]] */
void orc_eval_sample_batch_multi(const orc_table *tables, int n_tables, const orc_opts *o,
                                 const float *wi, const float *wo, const float *u, const int32_t *mat,
                                 size_t n, float *rgb, float *pdf, float *wo2, float *pdf2, float *weight);

/* ---- §8f item 2 ("next"): table importance sampling for sample()/pdf() ----------------------------
 * A one-sample mixture of the cosine lobe and a half-vector lobe read off the table:
 *   theta_h bins are the table's own rows (bin i = [theta_i, theta_i+1], theta_i = (i/n_th)^2 pi/2);
 *   D_i = mean luminance of row i (+1 % of the mean over rows, so no bin has zero probability);
 *   p_h(h) = c_i cos(theta_h) inside bin i,  c_i = D_i / (pi sum_j D_j (s_j+1 - s_j)),  s_i = sin^2 theta_i;
 *   sample(): u1 < 1/2 -> cosine hemisphere with (2 u1, u2); else theta_h by inverting the bin CDF with
 *             2 u1 - 1 (exact: sin^2 theta_h is uniform inside a bin), phi_h = 2 pi u2, wo = reflect(wi, h);
 *   pdf(wi, wo) = 1/2 cos(theta_o)/pi + 1/2 p_h(h) / (4 wi.h), h = normalize(wi + wo);
 *   sample() reports pdf(wi, wo_rounded_to_Float), so pdf(wi, sample.wo) == sample.pdf exactly.
 * A table in one of the standard parameterisations has no theta_h rows to read D_i from: its lobe is flat (D_i = 1,
 * p_h = cos(theta_h)/pi) — a valid mixture with nothing learnt from the table. */
typedef struct orc_sampling {
    int n;            /* = n_th */
    double *s;        /* [n+1] sin^2(theta_i) */
    double *cdf;      /* [n+1] */
    double *c;        /* [n]   */
    double alpha;     /* weight of the cosine lobe in the one-sample mixture: 1/2 (the row marginal), 1/8 (a row of the 2-D table) */
} orc_sampling;
int   orc_build_sampling(const orc_table *t, orc_sampling *out);
void  orc_free_sampling(orc_sampling *sp);
float orc_pdf_table(const orc_sampling *sp, const float wi[3], const float wo[3]);
void  orc_sample_table(const orc_table *t, const orc_opts *o, const orc_sampling *sp, const float wi[3], const float u[2],
                       float wo[3], float *pdf, float weight[3]);
void  orc_pdf_table_batch(const orc_sampling *sp, const float *wi, const float *wo, size_t n, float *pdf);
void  orc_sample_table_batch(const orc_table *t, const orc_opts *o, const orc_sampling *sp, const float *wi, const float *u,
                             size_t n, float *wo, float *pdf, float *weight);

/* ---- §8f item 2, survey form: conditional CDFs in two dimensions, P(theta_h | theta_i) --------------------------
 * The 1-D lobe above knows how bright a theta_h row is on average; it does not know that the lobe a surface shows depends
 * on the incident angle (Fresnel towards grazing, the 1 / (cos_i cos_o) of a microfacet shape, shadowed halves of the
 * hemisphere).  Here the half-vector lobe is conditional on the incident direction: n_i bins of mu = cos(theta_i)
 * (uniform in mu, bin i = [i / n_i, (i + 1) / n_i), centre mu_i), and per bin its own row distribution over the table's
 * theta_h bins.  The mass of (i, j) is what the BRDF itself puts there, measured in half-vector space:
 *     W_ij = ds_j * mean over K_s x K_p midpoints (s, phi) of  [ lum f(wi_i, wo) * cos(theta_o) * 4 (wi_i . h) / (2 cos(theta_h)) ]
 * with wi_i = (sqrt(1 - mu_i^2), 0, mu_i), s = sin^2(theta_h) uniform in the bin, phi in (0, pi) (the table is symmetric
 * in phi), h = (sqrt(s) cos phi, sqrt(s) sin phi, sqrt(1 - s)), wo = reflect(wi_i, h); samples with wi.h <= 0 or wo below
 * the horizon contribute 0; f is the table's own trilinear lookup; + a floor of 1 % of the row's mass spread uniformly in s.
 * A row is then exactly an orc_sampling (same s, its own cdf and c), and sample() / pdf() are the 1-D ones on the row of
 * wi's bin — a valid density for every wi, piecewise constant in mu.  Because the row follows the BRDF itself (diffuse floor
 * included) the cosine lobe only has to keep the estimator bounded: its mixture weight is alpha = 1/8 instead of 1/2
 * (u1 < alpha -> cosine with (u1 / alpha, u2); else the row CDF with (u1 - alpha) / (1 - alpha)). */
#define ORC_S2D_KS 4
#define ORC_S2D_KP 16
typedef struct orc_sampling2d {
    int n_i;               /* incident bins */
    orc_sampling *rows;    /* [n_i], rows[i].s all equal */
} orc_sampling2d;
int   orc_build_sampling2d(const orc_table *t, const orc_opts *o, int n_i, orc_sampling2d *out);
/* a table built elsewhere (the device's prefix-scan build, downloaded): flat[n_i][(n+1) cdf | n c], s: [n+1] */
int   orc_sampling2d_from_arrays(int n_i, int n, const double *s, const double *flat, orc_sampling2d *out);
void  orc_free_sampling2d(orc_sampling2d *sp);
int   orc_sampling2d_bin(const orc_sampling2d *sp, const float wi[3]);
void  orc_pdf_table2d_batch(const orc_sampling2d *sp, const float *wi, const float *wo, size_t n, float *pdf);
void  orc_sample_table2d_batch(const orc_table *t, const orc_opts *o, const orc_sampling2d *sp, const float *wi, const float *u,
                               size_t n, float *wo, float *pdf, float *weight);

/* ---- §8f item 3 ("next"): n-channel tables (customized_measurement beyond RGB) ---------------------------
 * Same parameterisation, transform, index maps and trilinear blend; a texel has n_ch values (planar: channel c at
 * data + c*n_th*n_td*n_pd), each with its own scale, negatives clamped to 0.  The sampling marginal weighs the
 * channels equally (the RGB one uses luminance).  PARITY UNPINNED like everything else here: the reference's
 * customized_measurement format is unknown. */
typedef struct orc_table_nch {
    int n_th, n_td, n_pd, n_ch;
    const double *data;
    const double *scale;          /* n_ch factors */
    int param;                    /* ORC_PARAM_* */
} orc_table_nch;
void  orc_lookup_nch(const orc_table_nch *t, const orc_opts *o, double theta_half, double theta_diff, double phi_diff, double *out);
void  orc_eval_nch(const orc_table_nch *t, const orc_opts *o, const float wi[3], const float wo[3], float *out);
void  orc_sample_nch(const orc_table_nch *t, const orc_opts *o, const float wi[3], const float u[2],
                     float wo[3], float *pdf, float *weight);
int   orc_build_sampling_nch(const orc_table_nch *t, orc_sampling *out);
void  orc_sample_table_nch(const orc_table_nch *t, const orc_opts *o, const orc_sampling *sp, const float wi[3], const float u[2],
                           float wo[3], float *pdf, float *weight);
/* mixed batch of n_ch-channel tables: tables[mat[i]] (mat NULL: tables[0]); ids outside [0, n_tables) or tables of
 * another width give zeros.  sp: NULL = cosine sampling; else sp[m] is table m's marginal (table sampling for
 * sample() and pdf()).  values / weight: n x n_ch. */
void  orc_eval_sample_batch_nch(const orc_table_nch *tables, int n_tables, int n_ch, const orc_opts *o, const orc_sampling *sp,
                                const float *wi, const float *wo, const float *u, const int32_t *mat, size_t n,
                                float *values, float *pdf, float *wo2, float *pdf2, float *weight);

/* ---- a9: GGX rough conductor (A.6), isotropic alpha, visible-normal sampling ---- */
typedef struct orc_ggx {
    double alpha;
    double eta[3], k[3];
} orc_ggx;
void  orc_ggx_eval(const orc_ggx *g, const float wi[3], const float wo[3], float rgb[3]);
float orc_ggx_pdf(const orc_ggx *g, const float wi[3], const float wo[3]);
void  orc_ggx_sample(const orc_ggx *g, const float wi[3], const float u[2],
                     float wo[3], float *pdf, float weight[3]);
void  orc_ggx_eval_batch(const orc_ggx *g, const float *wi, const float *wo, size_t n, float *rgb);
void  orc_ggx_pdf_batch(const orc_ggx *g, const float *wi, const float *wo, size_t n, float *pdf);
void  orc_ggx_sample_batch(const orc_ggx *g, const float *wi, const float *u, size_t n,
                           float *wo, float *pdf, float *weight);

/* ---- synthetic inputs: pair i -> splitmix64 -> (wi, wo, u) ---- */
void orc_generate_pairs(uint64_t seed, uint64_t first, size_t n, float *wi, float *wo, float *u);
void orc_generate_materials(uint64_t seed, uint64_t first, size_t n, int n_materials, int32_t *mat);

/* ---- CPU baseline: one *virtual* call per pair, Mitsuba-0.6 style ---- */
typedef struct orc_bsdf orc_bsdf;
typedef struct orc_bsdf_vtbl {
    void  (*eval)(const orc_bsdf *self, const float wi[3], const float wo[3], float rgb[3]);
    float (*pdf)(const orc_bsdf *self, const float wi[3], const float wo[3]);
    void  (*sample)(const orc_bsdf *self, const float wi[3], const float u[2],
                    float wo[3], float *pdf, float weight[3]);
} orc_bsdf_vtbl;
struct orc_bsdf {
    const orc_bsdf_vtbl *vtbl;
    orc_table table;
    orc_ggx   ggx;
    orc_opts  opts;
};
void orc_bsdf_init_merl(orc_bsdf *b, const double *planar, const orc_opts *o);
void orc_bsdf_init_ggx(orc_bsdf *b, const orc_ggx *g);
/* runs n eval+sample units [first, first+n) through the vtable on n_threads pthreads;
 * returns wall seconds; checksum (sum of all outputs) defeats dead-code elimination */
double orc_bench_eval_sample(const orc_bsdf *b, uint64_t seed, uint64_t first, size_t n,
                             int n_threads, double *checksum);
double orc_bench_eval(const orc_bsdf *b, uint64_t seed, uint64_t first, size_t n,
                      int n_threads, double *checksum);

#ifdef __cplusplus
}
#endif
#endif
