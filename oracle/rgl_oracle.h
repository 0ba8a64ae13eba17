/* rgl_oracle.h — TEST INFRASTRUCTURE (see rgl_oracle.c): CPU restatement of the RGL adaptive-parameterisation BSDF.  PARITY UNPINNED. */
#pragma once
#include <stddef.h>

#define RGL_MAX_DIM 3

/* piecewise-bilinear 2-D function / distribution over the unit square with up to 3 interpolated parameters */
typedef struct rgl_warp {
    int nx, ny;                 /* nodes per axis (x fastest) */
    int n_dim, n_par[RGL_MAX_DIM], stride[RGL_MAX_DIM], n_slices;
    float *par[RGL_MAX_DIM];    /* ascending parameter grids */
    float *data;                /* [n_slices][ny][nx]; divided by the slice's integral when normalized */
    float *marg;                /* [n_slices][ny - 1] running integral over rows (cdf) */
    float *cond;                /* [n_slices][ny][nx - 1] running integral along each node row */
    int normalized;
} rgl_warp;

int    rgl_warp_init(rgl_warp *w, int nx, int ny, int n_dim, const int *n_par, const float *const *par, const float *data,
                     int normalize, int build_cdf);
void   rgl_warp_free(rgl_warp *w);
double rgl_warp_eval(const rgl_warp *w, const double pos[2], const double *params);      /* density (normalized) or value */
void   rgl_warp_sample(const rgl_warp *w, const double u[2], const double *params, double pos[2], double *pdf);
void   rgl_warp_invert(const rgl_warp *w, const double pos[2], const double *params, double u[2], double *pdf);

typedef struct rgl_bsdf {
    int isotropic, jacobian;
    int reduction;              /* anisotropic files: 2 pi / (span of phi_i), rounded: 1 full azimuth, 2 half (point symmetry), 4 quarter */
    int n_wavelengths;          /* 0: an RGB file; else `rgb` holds the spectra, its third parameter grid the wavelengths */
    rgl_warp ndf, sigma, vndf, luminance, rgb;
} rgl_bsdf;

/* phi_i[n_phi], theta_i[n_theta]; ndf [res_ndf_y][res_ndf_x]; sigma [res_sigma_y][res_sigma_x]; vndf, luminance [n_phi][n_theta][res_y][res_x];
 * rgb [n_phi][n_theta][3][res_y][res_x] */
int  rgl_bsdf_init(rgl_bsdf *b, int n_phi, int n_theta, const float *phi_i, const float *theta_i, int res_ndf_x, int res_ndf_y, const float *ndf,
                   int res_sigma_x, int res_sigma_y, const float *sigma, int res_x, int res_y, const float *vndf, const float *luminance,
                   const float *rgb, int jacobian);
/* spectral files ("spectra" [n_phi][n_theta][n_wavelengths][res_y][res_x] + "wavelengths" instead of "rgb"); n_wavelengths == 0: as above */
int  rgl_bsdf_init_spectral(rgl_bsdf *b, int n_phi, int n_theta, const float *phi_i, const float *theta_i, int res_ndf_x, int res_ndf_y, const float *ndf,
                            int res_sigma_x, int res_sigma_y, const float *sigma, int res_x, int res_y, const float *vndf, const float *luminance,
                            int n_wavelengths, const float *wavelengths, const float *spectra, int jacobian);
void rgl_bsdf_free(rgl_bsdf *b);
/* a spectral file evaluated at W wavelengths per unit — wl [W] (NULL: the file's own nodes, W = n_wavelengths): the wavelength is
 * the third interpolated parameter of `spectra` (linear between nodes, clamped outside); pdf and sampled direction are wavelength-free */
void rgl_eval_pdf_spectral(const rgl_bsdf *b, const float wi[3], const float wo[3], const float *wl, int W, float *values, float *pdf_out);
void rgl_sample_spectral(const rgl_bsdf *b, const float wi[3], const float u[2], const float *wl, int W, float wo[3], float *pdf, float *weight);
void rgl_eval_pdf_spectral_batch(const rgl_bsdf *b, const float *wi, const float *wo, const float *wl /* [n][W] or NULL */, int W, size_t n, float *values, float *pdf);
void rgl_sample_spectral_batch(const rgl_bsdf *b, const float *wi, const float *u, const float *wl, int W, size_t n, float *wo, float *pdf, float *weight);
/* eval: f cos(theta_o) (RGB); pdf_out may be NULL */
void rgl_eval_pdf(const rgl_bsdf *b, const float wi[3], const float wo[3], float rgb[3], float *pdf_out);
/* rgl_eval_pdf in its two steps (for the tests' conditioning range): the prelude — both directions into the stored part of the
 * azimuth, normalised, m = wi + wo unnormalised; returns 0 when the pair evaluates to zero — and the evaluation from (wi, m) */
int  rgl_half_vector(const rgl_bsdf *b, const float wi_f[3], const float wo_f[3], double wi[3], double m[3]);
void rgl_eval_pdf_half(const rgl_bsdf *b, const double wi[3], const double m[3], float rgb[3], float *pdf_out);
void rgl_sample(const rgl_bsdf *b, const float wi[3], const float u[2], float wo[3], float *pdf, float weight[3]);
void rgl_eval_pdf_batch(const rgl_bsdf *b, const float *wi, const float *wo, size_t n, float *rgb, float *pdf);
void rgl_sample_batch(const rgl_bsdf *b, const float *wi, const float *u, size_t n, float *wo, float *pdf, float *weight);
