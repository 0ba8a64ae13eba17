// valu_issue.hip — issue cost (cycles per wave-instruction on one SIMD) of the vector instructions the table
// kernels are made of, measured the way the MI355X guide's constants table measures v_add_f32 / v_exp_f32:
// one wave's stream of INDEPENDENT instructions (8 rotating destination registers), s_memtime around it.
// Run with 1 wave per SIMD (issue cost of one stream) and with 2 waves per SIMD (what two co-resident waves
// of the fused kernel sustain together): the VALU budget of k_table_dma is priced with these numbers
// (profiles/r03_valu_issue.json).
//   hipcc -O3 --offload-arch=gfx950 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kIters = 512;     // loop trips
constexpr int kPerTrip = 32;    // instructions per trip (4 rounds over 8 registers)

// 8 independent destinations: d0..d7; sources a, b are never written
#define ROUND64(OP)                                                                                              \
    asm volatile(OP " %0, %8, %9\n\t" OP " %1, %8, %9\n\t" OP " %2, %8, %9\n\t" OP " %3, %8, %9\n\t"              \
                 OP " %4, %8, %9\n\t" OP " %5, %8, %9\n\t" OP " %6, %8, %9\n\t" OP " %7, %8, %9"                  \
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b))
#define ROUND64_FMA(OP)                                                                                          \
    asm volatile(OP " %0, %8, %9, %0\n\t" OP " %1, %8, %9, %1\n\t" OP " %2, %8, %9, %2\n\t" OP " %3, %8, %9, %3\n\t" \
                 OP " %4, %8, %9, %4\n\t" OP " %5, %8, %9, %5\n\t" OP " %6, %8, %9, %6\n\t" OP " %7, %8, %9, %7"  \
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a), "v"(b))
#define ROUND64_UN(OP)                                                                                           \
    asm volatile(OP " %0, %8\n\t" OP " %1, %8\n\t" OP " %2, %8\n\t" OP " %3, %8\n\t"                              \
                 OP " %4, %8\n\t" OP " %5, %8\n\t" OP " %6, %8\n\t" OP " %7, %8"                                  \
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a))

enum Op { FMA_F64, MUL_F64, ADD_F64, MAX_F64, RCP_F64, RSQ_F64, SQRT_F64, FLOOR_F64, FRACT_F64, CVT_F64_F32, CVT_F32_F64, CVT_I32_F64,
          CVT_F64_I32, CMP_F64, CNDMASK, MOV_B32, FMA_F32, MUL_F32, PK_FMA_F32, PK_MUL_F32, RCP_F32, SQRT_F32, RSQ_F32, MUL_LO_U32,
          MAD_U32_U24, ADD_U32, LSHL_ADD, BPERMUTE, DS_READ_B128, DIV_SCALE_F32, DIV_FMAS_F32, DIV_FIXUP_F32, N_OPS };
static const char *kNames[N_OPS] = { "v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_floor_f64",
    "v_fract_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_cvt_i32_f64", "v_cvt_f64_i32", "v_cmp_lt_f64", "v_cndmask_b32", "v_mov_b32", "v_fma_f32",
    "v_mul_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_mul_lo_u32", "v_mad_u32_u24", "v_add_u32", "v_lshl_add_u32",
    "ds_bpermute_b32", "ds_read_b128", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32" };

template <int OP>
__global__ __launch_bounds__(512) void k_issue(uint64_t *cycles, double *sink, double seed)
{
    __shared__ float4 lds[512];
    lds[threadIdx.x & 511] = make_float4((float)seed, 1.f, 2.f, 3.f);
    __syncthreads();
    double a = seed + 1.0 + threadIdx.x * 1e-9, b = seed + 0.999;
    double d0 = a, d1 = a + 1, d2 = a + 2, d3 = a + 3, d4 = a + 4, d5 = a + 5, d6 = a + 6, d7 = a + 7;
    float fa = (float)a, fb = (float)b;
    float f0 = fa, f1 = fa + 1, f2 = fa + 2, f3 = fa + 3, f4 = fa + 4, f5 = fa + 5, f6 = fa + 6, f7 = fa + 7;
    int ia = (int)threadIdx.x * 4 & 255, ib = 3;
    int i0 = ia, i1 = ia + 1, i2 = ia + 2, i3 = ia + 3, i4 = ia + 4, i5 = ia + 5, i6 = ia + 6, i7 = ia + 7;
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f q0 = {}, q1 = {}, q2 = {}, q3 = {};
    const unsigned lds_addr = (threadIdx.x & 63u) * 16u;
    __builtin_amdgcn_s_barrier();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int r = 0; r < kPerTrip / 8; ++r) {
            if constexpr (OP == FMA_F64) ROUND64_FMA("v_fma_f64");
            else if constexpr (OP == MUL_F64) ROUND64("v_mul_f64");
            else if constexpr (OP == ADD_F64) ROUND64("v_add_f64");
            else if constexpr (OP == MAX_F64) ROUND64("v_max_f64");
            else if constexpr (OP == RCP_F64) ROUND64_UN("v_rcp_f64");
            else if constexpr (OP == RSQ_F64) ROUND64_UN("v_rsq_f64");
            else if constexpr (OP == SQRT_F64) ROUND64_UN("v_sqrt_f64");
            else if constexpr (OP == FLOOR_F64) ROUND64_UN("v_floor_f64");
            else if constexpr (OP == FRACT_F64) ROUND64_UN("v_fract_f64");
            else if constexpr (OP == CVT_F64_F32)
                asm volatile("v_cvt_f64_f32 %0, %8\n\tv_cvt_f64_f32 %1, %8\n\tv_cvt_f64_f32 %2, %8\n\tv_cvt_f64_f32 %3, %8\n\t"
                             "v_cvt_f64_f32 %4, %8\n\tv_cvt_f64_f32 %5, %8\n\tv_cvt_f64_f32 %6, %8\n\tv_cvt_f64_f32 %7, %8"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(fa));
            else if constexpr (OP == CVT_F32_F64)
                asm volatile("v_cvt_f32_f64 %0, %8\n\tv_cvt_f32_f64 %1, %8\n\tv_cvt_f32_f64 %2, %8\n\tv_cvt_f32_f64 %3, %8\n\t"
                             "v_cvt_f32_f64 %4, %8\n\tv_cvt_f32_f64 %5, %8\n\tv_cvt_f32_f64 %6, %8\n\tv_cvt_f32_f64 %7, %8"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(a));
            else if constexpr (OP == CVT_I32_F64)
                asm volatile("v_cvt_i32_f64 %0, %8\n\tv_cvt_i32_f64 %1, %8\n\tv_cvt_i32_f64 %2, %8\n\tv_cvt_i32_f64 %3, %8\n\t"
                             "v_cvt_i32_f64 %4, %8\n\tv_cvt_i32_f64 %5, %8\n\tv_cvt_i32_f64 %6, %8\n\tv_cvt_i32_f64 %7, %8"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(a));
            else if constexpr (OP == CVT_F64_I32)
                asm volatile("v_cvt_f64_i32 %0, %8\n\tv_cvt_f64_i32 %1, %8\n\tv_cvt_f64_i32 %2, %8\n\tv_cvt_f64_i32 %3, %8\n\t"
                             "v_cvt_f64_i32 %4, %8\n\tv_cvt_f64_i32 %5, %8\n\tv_cvt_f64_i32 %6, %8\n\tv_cvt_f64_i32 %7, %8"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(ia));
            else if constexpr (OP == CMP_F64)
                asm volatile("v_cmp_lt_f64 vcc, %0, %8\n\tv_cmp_lt_f64 vcc, %1, %8\n\tv_cmp_lt_f64 vcc, %2, %8\n\tv_cmp_lt_f64 vcc, %3, %8\n\t"
                             "v_cmp_lt_f64 vcc, %4, %8\n\tv_cmp_lt_f64 vcc, %5, %8\n\tv_cmp_lt_f64 vcc, %6, %8\n\tv_cmp_lt_f64 vcc, %7, %8"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a) : "vcc");
            else if constexpr (OP == CNDMASK)
                asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n\tv_cndmask_b32 %1, %8, %9, vcc\n\tv_cndmask_b32 %2, %8, %9, vcc\n\tv_cndmask_b32 %3, %8, %9, vcc\n\t"
                             "v_cndmask_b32 %4, %8, %9, vcc\n\tv_cndmask_b32 %5, %8, %9, vcc\n\tv_cndmask_b32 %6, %8, %9, vcc\n\tv_cndmask_b32 %7, %8, %9, vcc"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fa), "v"(fb) : "vcc");
            else if constexpr (OP == MOV_B32)
                asm volatile("v_mov_b32 %0, %8\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %8\n\t"
                             "v_mov_b32 %4, %8\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %8"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fa));
#define ROUND32(OPS) asm volatile(OPS " %0, %8, %9\n\t" OPS " %1, %8, %9\n\t" OPS " %2, %8, %9\n\t" OPS " %3, %8, %9\n\t" \
                                  OPS " %4, %8, %9\n\t" OPS " %5, %8, %9\n\t" OPS " %6, %8, %9\n\t" OPS " %7, %8, %9"     \
                                  : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fa), "v"(fb))
#define ROUND32_3(OPS) asm volatile(OPS " %0, %8, %9, %0\n\t" OPS " %1, %8, %9, %1\n\t" OPS " %2, %8, %9, %2\n\t" OPS " %3, %8, %9, %3\n\t" \
                                    OPS " %4, %8, %9, %4\n\t" OPS " %5, %8, %9, %5\n\t" OPS " %6, %8, %9, %6\n\t" OPS " %7, %8, %9, %7"     \
                                    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fa), "v"(fb) : "vcc")
#define ROUND32_UN(OPS) asm volatile(OPS " %0, %8\n\t" OPS " %1, %8\n\t" OPS " %2, %8\n\t" OPS " %3, %8\n\t" \
                                     OPS " %4, %8\n\t" OPS " %5, %8\n\t" OPS " %6, %8\n\t" OPS " %7, %8"     \
                                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fa))
#define ROUNDI(OPS) asm volatile(OPS " %0, %8, %9\n\t" OPS " %1, %8, %9\n\t" OPS " %2, %8, %9\n\t" OPS " %3, %8, %9\n\t" \
                                 OPS " %4, %8, %9\n\t" OPS " %5, %8, %9\n\t" OPS " %6, %8, %9\n\t" OPS " %7, %8, %9"     \
                                 : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ia), "v"(ib))
#define ROUNDI_3(OPS) asm volatile(OPS " %0, %8, %9, %0\n\t" OPS " %1, %8, %9, %1\n\t" OPS " %2, %8, %9, %2\n\t" OPS " %3, %8, %9, %3\n\t" \
                                   OPS " %4, %8, %9, %4\n\t" OPS " %5, %8, %9, %5\n\t" OPS " %6, %8, %9, %6\n\t" OPS " %7, %8, %9, %7"     \
                                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ia), "v"(ib))
            else if constexpr (OP == FMA_F32) ROUND32_3("v_fma_f32");
            else if constexpr (OP == MUL_F32) ROUND32("v_mul_f32");
            else if constexpr (OP == PK_FMA_F32) ROUND64_FMA("v_pk_fma_f32");
            else if constexpr (OP == PK_MUL_F32) ROUND64("v_pk_mul_f32");
            else if constexpr (OP == RCP_F32) ROUND32_UN("v_rcp_f32");
            else if constexpr (OP == SQRT_F32) ROUND32_UN("v_sqrt_f32");
            else if constexpr (OP == RSQ_F32) ROUND32_UN("v_rsq_f32");
            else if constexpr (OP == MUL_LO_U32) ROUNDI("v_mul_lo_u32");
            else if constexpr (OP == MAD_U32_U24) ROUNDI_3("v_mad_u32_u24");
            else if constexpr (OP == ADD_U32) ROUNDI("v_add_u32");
            else if constexpr (OP == LSHL_ADD) ROUNDI_3("v_lshl_add_u32");
            else if constexpr (OP == BPERMUTE) ROUNDI("ds_bpermute_b32");
            else if constexpr (OP == DS_READ_B128)
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t"
                             "ds_read_b128 %0, %4 offset:4096\n\tds_read_b128 %1, %4 offset:5120\n\tds_read_b128 %2, %4 offset:6144\n\tds_read_b128 %3, %4 offset:7168"
                             : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(lds_addr));
            else if constexpr (OP == DIV_SCALE_F32)
                asm volatile("v_div_scale_f32 %0, vcc, %8, %9, %8\n\tv_div_scale_f32 %1, vcc, %8, %9, %8\n\tv_div_scale_f32 %2, vcc, %8, %9, %8\n\tv_div_scale_f32 %3, vcc, %8, %9, %8\n\t"
                             "v_div_scale_f32 %4, vcc, %8, %9, %8\n\tv_div_scale_f32 %5, vcc, %8, %9, %8\n\tv_div_scale_f32 %6, vcc, %8, %9, %8\n\tv_div_scale_f32 %7, vcc, %8, %9, %8"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fa), "v"(fb) : "vcc");
            else if constexpr (OP == DIV_FMAS_F32) ROUND32_3("v_div_fmas_f32");
            else if constexpr (OP == DIV_FIXUP_F32) ROUND32_3("v_div_fixup_f32");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63u) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    const double s = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + (double)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7)
                   + (double)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7) + (double)(q0.x + q1.x + q2.x + q3.x);
    if (s == 12345.6789) sink[0] = s;
}

template <int OP>
static void run_one(uint64_t *d_cyc, double *d_sink, int waves_per_simd, double *out_cycles)
{
    const int threads = 256 * waves_per_simd;       // one block on one CU: 4 SIMDs x waves_per_simd
    uint64_t h[16];
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL((k_issue<OP>), dim3(1), dim3(threads), 0, 0, d_cyc, d_sink, 0.5);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d_cyc, sizeof(uint64_t) * (threads / 64), hipMemcpyDeviceToHost));
        uint64_t mx = 0;
        for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
        const double c = (double)mx / ((double)kIters * kPerTrip);
        best = c < best ? c : best;
    }
    *out_cycles = best;
}

template <int OP>
static void run_all(uint64_t *d_cyc, double *d_sink, double res[][2])
{
    if constexpr (OP < N_OPS) {
        run_one<OP>(d_cyc, d_sink, 1, &res[OP][0]);
        run_one<OP>(d_cyc, d_sink, 2, &res[OP][1]);
        run_all<OP + 1>(d_cyc, d_sink, res);
    }
}

int main()
{
    uint64_t *d_cyc; double *d_sink;
    CK(hipMalloc(&d_cyc, 64 * sizeof(uint64_t)));
    CK(hipMalloc(&d_sink, sizeof(double)));
    static double res[N_OPS][2];
    run_all<0>(d_cyc, d_sink, res);
    // s_memtime counts at a fixed 100 MHz on gfx9; __builtin_readcyclecounter = s_memtime.  Calibrate against v_mov_b32
    // (the guide's constants table: plain VALU = 4 cycles issue) and print both raw ticks and the ratio to v_mov_b32.
    printf("{\"unit\": \"s_memtime ticks per wave-instruction; ratio = relative to v_mov_b32 (guide: plain VALU issues in 4 cycles)\", \"ops\": {\n");
    for (int o = 0; o < N_OPS; ++o)
        printf("  \"%s\": {\"ticks_1wave\": %.4f, \"ticks_2waves_per_simd_each\": %.4f, \"ratio_1wave\": %.2f, \"ratio_2waves\": %.2f}%s\n", kNames[o], res[o][0], res[o][1],
               res[o][0] / res[MOV_B32][0], res[o][1] / res[MOV_B32][1], o + 1 < N_OPS ? "," : "");
    printf("}}\n");
    return 0;
}
