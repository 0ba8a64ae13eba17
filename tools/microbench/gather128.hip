// gather128.hip — ceiling of the brick access pattern: random 128-B lines copied to LDS by
// global_load_lds_dwordx4, eight lanes per line, nothing else.   hipcc -O3 --offload-arch=gfx950
//   ./gather128 <table_MB> <waves_per_block> <blocks_per_cu> <lines_per_lane_per_iter(1|2)> [aux]
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

template <int LOOKUPS, int AUX>
__global__ void k(const float4 *table, uint64_t n_lines, int iters, float *sink)
{
    extern __shared__ float4 lds[];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 *my = lds + (size_t)wave * LOOKUPS * 512;
    float acc = 0.f;
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        for (int l = 0; l < LOOKUPS; ++l) {
            uint32_t idx = (uint32_t)(mix64(gid * 131 + it * 2 + l) % n_lines);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                unsigned unit = 8 * kk + (lane >> 3);
                uint32_t sidx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(unit << 2), (int)idx);
                const float4 *src = table + (size_t)sidx * 8 + (lane & 7);
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
    size_t mb = argc > 1 ? atol(argv[1]) : 187;
    int wpb = argc > 2 ? atoi(argv[2]) : 4, bpc = argc > 3 ? atoi(argv[3]) : 2, lookups = argc > 4 ? atoi(argv[4]) : 2, aux = argc > 5 ? atoi(argv[5]) : 0;
    size_t bytes = mb << 20; uint64_t n_lines = bytes / 128;
    float4 *t; float *sink;
    CK(hipMalloc(&t, bytes)); CK(hipMemset(t, 0, bytes)); CK(hipMalloc(&sink, 4));
    int iters = 64;
    dim3 grid(256 * bpc), block(64 * wpb);
    size_t lds = (size_t)wpb * lookups * 8192;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto launch = [&]() {
        if (aux == 1) hipLaunchKernelGGL((k<2, 1>), grid, block, lds, 0, t, n_lines, iters, sink); else if (aux == 3) hipLaunchKernelGGL((k<2, 3>), grid, block, lds, 0, t, n_lines, iters, sink); else if (aux == 16) hipLaunchKernelGGL((k<2, 16>), grid, block, lds, 0, t, n_lines, iters, sink); else if (aux == 17) hipLaunchKernelGGL((k<2, 17>), grid, block, lds, 0, t, n_lines, iters, sink); else if (aux == 18) hipLaunchKernelGGL((k<2, 18>), grid, block, lds, 0, t, n_lines, iters, sink); else if (lookups == 2) { if (aux) hipLaunchKernelGGL((k<2, 2>), grid, block, lds, 0, t, n_lines, iters, sink); else hipLaunchKernelGGL((k<2, 0>), grid, block, lds, 0, t, n_lines, iters, sink); }
        else { if (aux) hipLaunchKernelGGL((k<1, 2>), grid, block, lds, 0, t, n_lines, iters, sink); else hipLaunchKernelGGL((k<1, 0>), grid, block, lds, 0, t, n_lines, iters, sink); }
    };
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int r = 0; r < 5; ++r) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    double lines = (double)grid.x * block.x * iters * lookups;
    printf("table %zu MB  waves/block %d  blocks/CU %d  lookups/iter %d aux %d  LDS/block %zu KB : %.3f ms, %.1f G lines/s, %.2f TB/s\n",
           mb, wpb, bpc, lookups, aux, lds >> 10, ms, lines / ms / 1e6, lines * 128 / ms / 1e9);
    return 0;
}
