// gather_partial.hip — does the fabric move less than a 128-B line when a lookup asks for less?
// Random bricks copied to LDS by global_load_lds_dwordx4 as in gather128.hip, but only the first LPL of the eight
// 16-B pieces of each 128-B line are requested (LPL = 8: whole line, 6: the 96 B a packed RGB brick really holds,
// 4: 64 B, 2: 32 B).  Run under rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum (and FETCH_SIZE in a second
// pass) to see whether the L2 issues partial-line (32-B / 64-B) requests to the fabric or always fetches 128 B.
// Mode "dense": LPL = 6 with the wave's 64 lanes covering 10 bricks per instruction (60 active lanes) instead of 8
// bricks with 2 of 8 lanes masked off — the instruction-count saving a 96-B brick fetch could have.
// Mode "copy": float4 streaming copy of a known byte count, the calibration run for FETCH_SIZE / WRITE_SIZE.
//   hipcc -O3 --offload-arch=gfx950 -o gather_partial gather_partial.hip
//   ./gather_partial <table_MB> <lpl: 8|6|4|2|60 (dense 96 B)|0 (copy)> [blocks_per_cu]
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

template <int LPL>
__global__ __launch_bounds__(256) void k_partial(const float4 *table, uint64_t n_lines, int iters, float *sink)
{
    extern __shared__ float4 lds[];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 *my = lds + (size_t)wave * 2 * 512;
    float acc = 0.f;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        for (int l = 0; l < 2; ++l) {
            const uint32_t idx = (uint32_t)(mix64(gid * 131 + it * 2 + l) % n_lines);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const unsigned unit = 8 * kk + (lane >> 3);
                const uint32_t sidx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(unit << 2), (int)idx);
                const float4 *src = table + (size_t)sidx * 8 + (lane & 7);
                if ((lane & 7) < (unsigned)LPL)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(my + l * 512 + kk * 64), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += my[lane * 8].x;
        asm volatile("" ::: "memory");
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// 96 B per brick, 6 lanes per brick, 10 bricks per instruction: 64 bricks of a wave-lookup in 7 instructions
// (6 x 10 + 1 x 4 bricks) instead of 8.
__global__ __launch_bounds__(256) void k_dense96(const float4 *table, uint64_t n_lines, int iters, float *sink)
{
    extern __shared__ float4 lds[];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 *my = lds + (size_t)wave * 2 * 512;
    float acc = 0.f;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned sub = lane / 6, piece = lane % 6;                 // lanes 60..63: sub = 10 -> inactive
    for (int it = 0; it < iters; ++it) {
        for (int l = 0; l < 2; ++l) {
            const uint32_t idx = (uint32_t)(mix64(gid * 131 + it * 2 + l) % n_lines);
#pragma unroll
            for (int kk = 0; kk < 7; ++kk) {
                const unsigned unit = 10 * kk + sub;
                const uint32_t sidx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((unit & 63u) << 2), (int)idx);
                const float4 *src = table + (size_t)sidx * 8 + piece;
                if (sub < 10 && unit < 64)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(my + l * 512 + kk * 64), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += my[lane * 6].x;
        asm volatile("" ::: "memory");
    }
    if (acc == 12345.678f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_copy(const float4 *src, float4 *dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? atol(argv[1]) : 187;
    const int lpl = argc > 2 ? atoi(argv[2]) : 8, bpc = argc > 3 ? atoi(argv[3]) : 2;
    const size_t bytes = mb << 20; const uint64_t n_lines = bytes / 128;
    float4 *t; float *sink;
    CK(hipMalloc(&t, bytes)); CK(hipMemset(t, 0, bytes)); CK(hipMalloc(&sink, 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    if (lpl == 0) {
        float4 *d; CK(hipMalloc(&d, bytes));
        const size_t n = bytes / 16;
        k_copy<<<256 * 8, 256>>>(t, d, n); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a)); for (int r = 0; r < 5; ++r) k_copy<<<256 * 8, 256>>>(t, d, n); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
        printf("copy %zu MB read + %zu MB written per launch: %.3f ms, %.2f TB/s (read+write)\n", mb, mb, ms, 2.0 * bytes / ms / 1e9);
        return 0;
    }
    const int iters = 64;
    const dim3 grid(256 * bpc), block(256);
    const size_t lds = 4 * 2 * 8192;
    auto launch = [&]() {
        switch (lpl) {
            case 8: hipLaunchKernelGGL((k_partial<8>), grid, block, lds, 0, t, n_lines, iters, sink); break;
            case 6: hipLaunchKernelGGL((k_partial<6>), grid, block, lds, 0, t, n_lines, iters, sink); break;
            case 4: hipLaunchKernelGGL((k_partial<4>), grid, block, lds, 0, t, n_lines, iters, sink); break;
            case 2: hipLaunchKernelGGL((k_partial<2>), grid, block, lds, 0, t, n_lines, iters, sink); break;
            case 60: hipLaunchKernelGGL(k_dense96, grid, block, lds, 0, t, n_lines, iters, sink); break;
            default: printf("bad lpl\n"); exit(1);
        }
    };
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int r = 0; r < 5; ++r) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    const double lines = (double)grid.x * block.x * iters * 2;
    const int req = lpl == 60 ? 96 : lpl * 16;
    printf("table %zu MB  %3d B requested per 128-B line (%s)  blocks/CU %d : %.3f ms, %.1f G lines/s, %.2f TB/s requested, %.2f TB/s if whole lines move; lines per launch %.0f\n",
           mb, req, lpl == 60 ? "dense, 10 bricks per instruction" : "masked lanes", bpc, ms, lines / ms / 1e6, lines * req / ms / 1e9, lines * 128 / ms / 1e9, lines);
    return 0;
}
