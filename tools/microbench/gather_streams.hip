// gather_streams.hip — the bench kernel's memory behaviour without its arithmetic: per wave and iteration
//   read 2 KB of "input stream" (2 x dwordx4 per lane), copy 2 x 64 random 128-B bricks of a table into LDS
//   (global_load_lds_dwordx4, as k_table_dma), write 2.75 KB of "output stream" (dwordx4 stores, 176 per wave).
// Question: how much do the streams' cache-policy bits change the brick hit rate in L2 and the time?
//   store flavour: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1      load flavour: 0 plain, 1 nt, 2 sc1
//   ./gather_streams <table_MB> <load_flavour> <store_flavour> [iters]
// Run under rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum for the fabric side.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

typedef float v4f __attribute__((ext_vector_type(4)));

template <int LF> __device__ __forceinline__ float4 ld4(const float4 *p)
{
    v4f v;
    if constexpr (LF == 0) return *p;
    else if constexpr (LF == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}
template <int SF> __device__ __forceinline__ void st4(float4 *p, float4 f)
{
    const v4f v = { f.x, f.y, f.z, f.w };
    if constexpr (SF == 0) *p = f;
    else if constexpr (SF == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    else if constexpr (SF == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

template <int LF, int SF>
__global__ __launch_bounds__(256) void k(const float4 *table, uint64_t n_lines, const float4 *in, float4 *out, size_t waves_total, int iters)
{
    extern __shared__ float4 lds[];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 *my = lds + (size_t)wave * 2 * 512;
    const size_t gwave = (size_t)blockIdx.x * 4 + wave, nwaves = (size_t)gridDim.x * 4;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const size_t slot = (gwave + (size_t)it * nwaves) % waves_total;          // this iteration's 64 "units"
        const float4 a = ld4<LF>(in + slot * 128 + lane), b = ld4<LF>(in + slot * 128 + 64 + lane);
        acc += a.x + b.y;
        for (int l = 0; l < 2; ++l) {
            const uint32_t idx = (uint32_t)(mix64((slot * 64 + lane) * 2 + l) % n_lines);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const unsigned unit = 8 * kk + (lane >> 3);
                const uint32_t sidx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(unit << 2), (int)idx);
                const float4 *src = table + (size_t)sidx * 8 + (lane & 7);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(my + l * 512 + kk * 64), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float4 r = my[lane * 8];
        acc += r.x;
        float4 *o = out + slot * 176;
        st4<SF>(o + lane, make_float4(acc, r.y, a.z, b.w));
        st4<SF>(o + 64 + lane, make_float4(r.z, acc, b.x, a.y));
        if (lane < 48) st4<SF>(o + 128 + lane, make_float4(a.w, b.z, acc, r.w));
        asm volatile("" ::: "memory");
    }
}

int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? atol(argv[1]) : 187;
    const int lf = argc > 2 ? atoi(argv[2]) : 1, sf = argc > 3 ? atoi(argv[3]) : 1, iters = argc > 4 ? atoi(argv[4]) : 64;
    const size_t bytes = mb << 20; const uint64_t n_lines = bytes / 128;
    const dim3 grid(512), block(256);
    const size_t waves_total = (size_t)grid.x * 4 * iters;                         // every iteration touches fresh stream memory
    float4 *t, *in, *out;
    CK(hipMalloc(&t, bytes)); CK(hipMemset(t, 0, bytes));
    CK(hipMalloc(&in, waves_total * 128 * sizeof(float4))); CK(hipMemset(in, 0, waves_total * 128 * sizeof(float4)));
    CK(hipMalloc(&out, waves_total * 176 * sizeof(float4)));
    const size_t lds = 4 * 2 * 8192;
    auto launch = [&]() {
#define L(LF, SF) hipLaunchKernelGGL((k<LF, SF>), grid, block, lds, 0, t, n_lines, in, out, waves_total, iters)
        switch (lf * 4 + sf) {
            case 0: L(0, 0); break; case 1: L(0, 1); break; case 2: L(0, 2); break; case 3: L(0, 3); break;
            case 4: L(1, 0); break; case 5: L(1, 1); break; case 6: L(1, 2); break; case 7: L(1, 3); break;
            case 8: L(2, 0); break; case 9: L(2, 1); break; case 10: L(2, 2); break; case 11: L(2, 3); break;
            default: printf("bad flavour\n"); exit(1);
        }
    };
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int r = 0; r < 5; ++r) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    const double units = (double)waves_total * 64;
    printf("table %zu MB  loads %s  stores %s : %.3f ms per %.0f units = %.2f G units/s (2 bricks + 32 B in + 44 B out per unit)\n", mb,
           lf == 0 ? "plain" : lf == 1 ? "nt" : "sc1", sf == 0 ? "plain" : sf == 1 ? "nt" : sf == 2 ? "sc1" : "sc0 sc1", ms, units, units / ms / 1e6);
    return 0;
}
