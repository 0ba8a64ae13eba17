// gather_hot.hip — can cache-policy bits on the COLD brick fetches keep a hot set of bricks in L2?
// Random bricks copied to LDS as in k_table_dma; a share p_hot of the lookups goes to a small hot region (hot_MB), the rest
// uniformly to the remainder of a 187 MB table.  Hot bricks are always fetched with the default policy; cold bricks with
// the policy under test (aux: 0 default, 2 nt, 16 sc1, 18 sc1 nt, 1 sc0, 3 sc0 nt).  Each copy step issues one masked
// instruction for the hot bricks of the step and one for the cold ones.
//   ./gather_hot <hot_MB> <p_hot_percent> <aux_cold> [table_MB]
// Under rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum: fabric reads per lookup = 1 - L2 hit rate.
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

template <int AUX>
__global__ __launch_bounds__(256) void k(const float4 *table, uint32_t hot_lines, uint32_t n_lines, uint32_t p_hot_1024, int iters, float *sink)
{
    extern __shared__ float4 lds[];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 *my = lds + (size_t)wave * 2 * 512;
    float acc = 0.f;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        for (int l = 0; l < 2; ++l) {
            const uint64_t r = mix64(gid * 131 + it * 2 + l);
            const bool hot = (uint32_t)(r & 1023u) < p_hot_1024;
            const uint32_t idx = hot ? (uint32_t)((r >> 10) % hot_lines) : hot_lines + (uint32_t)((r >> 10) % (n_lines - hot_lines));
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const unsigned unit = 8 * kk + (lane >> 3);
                const uint32_t sidx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(unit << 2), (int)idx);
                const float4 *src = table + (size_t)sidx * 8 + (lane & 7);
                if (sidx < hot_lines)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(my + l * 512 + kk * 64), 16, 0, 0);
                else
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(my + l * 512 + kk * 64), 16, 0, AUX);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += my[lane * 8].x;
        asm volatile("" ::: "memory");
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const size_t hot_mb = argc > 1 ? atol(argv[1]) : 3;
    const int p_hot = argc > 2 ? atoi(argv[2]) : 15, aux = argc > 3 ? atoi(argv[3]) : 0;
    const size_t mb = argc > 4 ? atol(argv[4]) : 187;
    const size_t bytes = mb << 20;
    const uint32_t n_lines = (uint32_t)(bytes / 128), hot_lines = (uint32_t)((hot_mb << 20) / 128);
    float4 *t; float *sink;
    CK(hipMalloc(&t, bytes)); CK(hipMemset(t, 0, bytes)); CK(hipMalloc(&sink, 4));
    const int iters = 64;
    const dim3 grid(512), block(256);
    const size_t lds = 4 * 2 * 8192;
    const uint32_t p1024 = (uint32_t)(p_hot * 1024 / 100);
    auto launch = [&]() {
        switch (aux) {
            case 0: hipLaunchKernelGGL((k<0>), grid, block, lds, 0, t, hot_lines, n_lines, p1024, iters, sink); break;
            case 1: hipLaunchKernelGGL((k<1>), grid, block, lds, 0, t, hot_lines, n_lines, p1024, iters, sink); break;
            case 2: hipLaunchKernelGGL((k<2>), grid, block, lds, 0, t, hot_lines, n_lines, p1024, iters, sink); break;
            case 3: hipLaunchKernelGGL((k<3>), grid, block, lds, 0, t, hot_lines, n_lines, p1024, iters, sink); break;
            case 16: hipLaunchKernelGGL((k<16>), grid, block, lds, 0, t, hot_lines, n_lines, p1024, iters, sink); break;
            case 18: hipLaunchKernelGGL((k<18>), grid, block, lds, 0, t, hot_lines, n_lines, p1024, iters, sink); break;
            default: printf("bad aux\n"); exit(1);
        }
    };
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int r = 0; r < 5; ++r) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    const double lines = (double)grid.x * block.x * iters * 2;
    printf("hot %zu MB at %d %% of the lookups, cold policy aux=%d : %.3f ms per %.0f lookups = %.1f G lookups/s\n", hot_mb, p_hot, aux, ms, lines, lines / ms / 1e6);
    return 0;
}
