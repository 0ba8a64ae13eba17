/* abi_example.c — the C ABI from plain C (C99): load a MERL-layout table, evaluate a batch held in
 * pinned host memory (zero copy over PCIe) and the same batch from plain host memory (staged), compare.
 *   gcc -std=c99 -I include examples/abi_example.c -L mitsuba_customization_amd/lib -lmerl_hip -Wl,-rpath,... -lm
 *   ./abi_example table.binary          (exit 0 and "example ok" on success; needs an MI355X) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "merl_hip.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_ != MRL_OK) {                                                                     \
            fprintf(stderr, "%s -> %s (%s)\n", #call, mrl_strerror(rc_), ctx ? mrl_last_error(ctx) : ""); \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

int main(int argc, char **argv)
{
    mrl_ctx *ctx = NULL;
    if (argc < 2) { fprintf(stderr, "usage: %s table.binary\n", argv[0]); return 2; }
    CHECK(mrl_init(0, &ctx));
    char name[128]; int cus = 0; size_t mem = 0;
    CHECK(mrl_device_info(ctx, name, sizeof name, &cus, &mem));
    printf("device: %s, %d CUs, %.0f GB\n", name, cus, (double)mem / 1e9);
    int id = -1;
    CHECK(mrl_material_load_merl(ctx, argv[1], &id));

    enum { N = 100000 };
    float *pin = NULL;                                  /* wi wo u | rgb pdf wo2 pdf2 weight : 19 floats per unit */
    CHECK(mrl_host_alloc(ctx, sizeof(float) * 19 * N, (void **)&pin));
    float *wi = pin, *wo = pin + 3 * N, *u = pin + 6 * N;
    float *rgb = pin + 8 * N, *pdf = pin + 11 * N, *wo2 = pin + 12 * N, *pdf2 = pin + 15 * N, *w = pin + 16 * N;
    /* device-side generator writes straight into the pinned block (it is device-accessible) */
    CHECK(mrl_generate_pairs(ctx, 0x5EEDu, 0, N, wi, wo, u));
    CHECK(mrl_eval_sample_batch(ctx, wi, wo, u, NULL, id, N, rgb, pdf, wo2, pdf2, w));
    CHECK(mrl_synchronize(ctx));

    /* the same batch from ordinary malloc memory: staged through HBM, returns when the outputs are written */
    float *h = (float *)malloc(sizeof(float) * 19 * N);
    memcpy(h, pin, sizeof(float) * 8 * N);
    CHECK(mrl_eval_sample_batch(ctx, h, h + 3 * N, h + 6 * N, NULL, id, N, h + 8 * N, h + 11 * N, h + 12 * N, h + 15 * N, h + 16 * N));
    int same = memcmp(h + 8 * N, pin + 8 * N, sizeof(float) * 11 * N) == 0;
    double sum = 0.0;
    for (int i = 0; i < 3 * N; ++i) sum += rgb[i];
    printf("mean rgb %.6f, pinned and staged paths %s\n", sum / (3.0 * N), same ? "agree bit for bit" : "DIFFER");

    /* error paths return codes, never abort */
    int bad = mrl_eval_batch(ctx, wi, wo, NULL, 12345, N, rgb);
    printf("unknown material -> %d (%s)\n", bad, mrl_strerror(bad));
    free(h);
    CHECK(mrl_host_free(ctx, pin));
    CHECK(mrl_destroy(ctx));
    if (!same || bad != MRL_ERR_MATERIAL || !(sum > 0.0)) return 1;
    puts("example ok");
    return 0;
}
