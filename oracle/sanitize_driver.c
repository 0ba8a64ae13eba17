/* sanitize_driver.c — runs the oracle over its whole surface under -fsanitize=address,undefined
 * (CPU build only: GPU sanitizers are not available on the pool).  Built and run by
 * tests/test_sanitize_cpu.py;  exit code 0 = clean.  Test infrastructure, like everything in oracle/. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "merl_oracle.h"

int main(int argc, char **argv)
{
    const char *tmp = argc > 1 ? argv[1] : "/tmp/orc_sanitize.binary";
    const int dims[3] = { 9, 7, 12 };
    const size_t n = (size_t)dims[0] * dims[1] * dims[2];
    double *planar = (double *)malloc(sizeof(double) * 3 * n);
    for (size_t i = 0; i < 3 * n; ++i) planar[i] = (i % 17 == 0) ? -1.0 : 10.0 + (double)(i % 101);
    if (orc_write_table(tmp, planar, dims) != 0) return 2;
    double *back = NULL; int d[3];
    if (orc_read_table(tmp, 0, &back, d) != 0 || memcmp(back, planar, sizeof(double) * 3 * n) != 0) return 3;
    if (orc_read_table(tmp, 1, &back + 0, d) != -3) return 4;             /* not MERL dims */
    orc_table t = { dims[0], dims[1], dims[2], back, { 0.5, 1.0, 2.0 } };
    orc_sampling sp;
    if (orc_build_sampling(&t, &sp) != 0) return 5;

    enum { N = 20000 };
    float *wi = malloc(12 * N), *wo = malloc(12 * N), *u = malloc(8 * N);
    float *rgb = malloc(12 * N), *pdf = malloc(4 * N), *wo2 = malloc(12 * N), *pdf2 = malloc(4 * N), *w = malloc(12 * N);
    int32_t *mat = malloc(4 * N);
    orc_generate_pairs(0x5EED, 12345, N, wi, wo, u);
    orc_generate_materials(0x5EED, 0, N, 1, mat);
    /* edge inputs: below horizon, zero vector, NaN, inf, huge, exact mirror / retro pairs */
    wi[2] = -wi[2]; wo[3 * 1 + 2] = 0.0f;
    wi[3 * 2] = wi[3 * 2 + 1] = wi[3 * 2 + 2] = 0.0f;
    wi[3 * 3] = NAN; wo[3 * 4 + 2] = INFINITY; wi[3 * 5] = 3e38f;
    memcpy(wo + 3 * 6, wi + 3 * 6, 12);
    wo[3 * 7] = -wi[3 * 7]; wo[3 * 7 + 1] = -wi[3 * 7 + 1]; wo[3 * 7 + 2] = wi[3 * 7 + 2];
    u[0] = 0.0f; u[1] = 0.0f; u[2] = 1.0f; u[3] = 0.5f; u[4] = 0.5f; u[5] = 0.5f;
    mat[9] = -3; mat[10] = 7;
    for (int param = 0; param < 3; ++param)                       /* half/diff, standard, standard-full axes */
    for (int lookup = 0; lookup < 2; ++lookup)
        for (int node = 0; node < 2; ++node)
            for (int disk = 0; disk < 2; ++disk) {
                orc_opts o = { lookup, node, disk };
                t.param = param;
                orc_eval_sample_batch_multi(&t, 1, &o, wi, wo, u, mat, N, rgb, pdf, wo2, pdf2, w);
                orc_sample_table_batch(&t, &o, &sp, wi, u, N, wo2, pdf2, w);
                orc_pdf_table_batch(&sp, wi, wo, N, pdf);
            }
    orc_ggx g = { 0.1, { 0.143, 0.375, 1.442 }, { 3.983, 2.386, 1.603 } };
    orc_ggx_eval_batch(&g, wi, wo, N, rgb);
    orc_ggx_pdf_batch(&g, wi, wo, N, pdf);
    orc_ggx_sample_batch(&g, wi, u, N, wo2, pdf2, w);

    /* n-channel tables: 1, 5 and 32 channels, mixed batch with a table of another width and unknown ids, both samplers */
    {
        enum { CMAX = 32 };
        double *wide = (double *)malloc(sizeof(double) * CMAX * n), scale[CMAX];
        for (size_t i = 0; i < CMAX * n; ++i) wide[i] = (i % 23 == 0) ? -2.0 : 1.0 + (double)(i % 57);
        for (int c = 0; c < CMAX; ++c) scale[c] = 0.25 + 0.125 * c;
        const int widths[3] = { 1, 5, 32 };
        for (int k = 0; k < 3; ++k) {
            const int C = widths[k];
            orc_table_nch tabs[2] = { { dims[0], dims[1], dims[2], C, wide, scale }, { dims[0], dims[1], dims[2], C == 1 ? 2 : 1, wide, scale } };
            orc_sampling sps[2];
            if (orc_build_sampling_nch(&tabs[0], &sps[0]) != 0 || orc_build_sampling_nch(&tabs[1], &sps[1]) != 0) return 7;
            float *val = malloc(4 * (size_t)C * N), *wgt = malloc(4 * (size_t)C * N);
            for (int i = 0; i < N; ++i) mat[i] = (i % 5) - 1;                        /* -1, 0, 1 (other width), 2, 3 (unknown) */
            for (int lookup = 0; lookup < 2; ++lookup) {
                orc_opts on = { lookup, lookup, 1 - lookup };
                tabs[0].param = (k + lookup) % 3; tabs[1].param = (k + lookup + 1) % 3;
                orc_eval_sample_batch_nch(tabs, 2, C, &on, NULL, wi, wo, u, mat, N, val, pdf, wo2, pdf2, wgt);
                orc_eval_sample_batch_nch(tabs, 2, C, &on, sps, wi, wo, u, mat, N, val, pdf, wo2, pdf2, wgt);
                orc_eval_sample_batch_nch(tabs, 1, C, &on, NULL, wi, wo, u, NULL, N, val, pdf, wo2, pdf2, wgt);
            }
            double out[CMAX];
            orc_opts on = { 1, 0, 0 };
            orc_lookup_nch(&tabs[0], &on, 0.0, 0.0, 0.0, out);
            orc_lookup_nch(&tabs[0], &on, 1.5707963, 1.5707963, 3.1415926, out);
            orc_free_sampling(&sps[0]); orc_free_sampling(&sps[1]);
            free(val); free(wgt);
        }
        free(wide);
        orc_generate_materials(0x5EED, 0, N, 1, mat);
    }

    orc_bsdf b; orc_opts o = { 1, 0, 0 };
    /* the baseline driver needs MERL dims */
    double *merl = (double *)calloc(3 * (size_t)ORC_MERL_N, sizeof(double));
    for (size_t i = 0; i < 3 * (size_t)ORC_MERL_N; i += 7) merl[i] = 100.0;
    orc_bsdf_init_merl(&b, merl, &o);
    double chk = 0.0;
    if (!(orc_bench_eval_sample(&b, 0x5EED, 0, 5000, 3, &chk) > 0.0)) return 6;
    orc_bsdf_init_ggx(&b, &g);
    if (!(orc_bench_eval(&b, 0x5EED, 0, 5000, 2, &chk) > 0.0)) return 6;

    orc_free_sampling(&sp);
    orc_free(back); free(planar); free(merl);
    free(wi); free(wo); free(u); free(rgb); free(pdf); free(wo2); free(pdf2); free(w); free(mat);
    remove(tmp);
    puts("sanitize ok");
    return 0;
}
