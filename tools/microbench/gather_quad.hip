// gather_quad.hip — what a lane's four 16-B reads cost the CU's texture addresser, by how their addresses lie:
//   scatter: four independent random 16-B slots per lane (a slice-major RGL table: one line per slice);
//   lane64:  the four slots of one random 64-B block per lane (the bracket-major table: one line per lane, four instructions);
//   quad:    the same 64-B blocks, read by the QUAD — in instruction j the four lanes of a quad read the four slots of lane j's
//            block (one line per quad and instruction), the vectors then go back to their lane through DPP quad permutes;
//   quadraw: quad without the way back (the loads alone);
//   pair32:  32-B blocks (two slots), two instructions, lane pairs read one block per instruction — against lane32, each lane its own.
// Prints ns per wave-instruction slot and G blocks/s.  Footprint in MB picks L2-resident or not.
//   hipcc -O3 --offload-arch=gfx950 -o gather_quad gather_quad.hip ;  ./gather_quad <MB> <mode> [waves_per_simd=4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int CTRL>
__device__ __forceinline__ float dpp(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true)); }
template <int CTRL>
__device__ __forceinline__ uint32_t dppu(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ float4 dpp4(const float4 &v) { return make_float4(dpp<CTRL>(v.x), dpp<CTRL>(v.y), dpp<CTRL>(v.z), dpp<CTRL>(v.w)); }

// quad_perm controls: broadcast lane j of the quad = j * 0x55
constexpr int QB0 = 0x00, QB1 = 0x55, QB2 = 0xAA, QB3 = 0xFF;

template <int MODE>
__global__ __launch_bounds__(256) void k(const float4 *table, uint32_t n_blocks, int iters, float *sink)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 3;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const uint32_t h = mix32(gid * 2654435761u + (uint32_t)it * 40503u);
        float4 a, b, c, d;
        if constexpr (MODE == 0) {              // scatter
            const uint32_t i0 = h % (n_blocks * 4), i1 = mix32(h + 1) % (n_blocks * 4), i2 = mix32(h + 2) % (n_blocks * 4), i3 = mix32(h + 3) % (n_blocks * 4);
            a = table[i0]; b = table[i1]; c = table[i2]; d = table[i3];
        } else if constexpr (MODE == 1) {       // lane64
            const float4 *p = table + (size_t)(h % n_blocks) * 4;
            a = p[0]; b = p[1]; c = p[2]; d = p[3];
        } else if constexpr (MODE == 2 || MODE == 3) {   // quad / quadraw
            const uint32_t blk = h % n_blocks;
            const float4 *p0 = table + (size_t)dppu<QB0>(blk) * 4 + lane, *p1 = table + (size_t)dppu<QB1>(blk) * 4 + lane;
            const float4 *p2 = table + (size_t)dppu<QB2>(blk) * 4 + lane, *p3 = table + (size_t)dppu<QB3>(blk) * 4 + lane;
            const float4 r0 = *p0, r1 = *p1, r2 = *p2, r3 = *p3;       // r_j: slot `lane` of quad-lane j's block
            if constexpr (MODE == 3) { a = r0; b = r1; c = r2; d = r3; }
            else {
                // lane l wants slot k of its own block: held by lane k in r_l.  Select r_l first (the register index is the lane's own
                // number), then fetch across the quad: slot k = broadcast-from-lane-k of (that lane's r_{reader})... the reader differs per
                // lane, so: m_k = r_{(lane - k) & 3 ... } rotate form: step s = 0..3, lane l takes from lane (l + s) & 3 its register r_l.
                // A lane cannot name the reader's register; the sender selects for the reader: sender t sends r_{(t - s) & 3} under rotation s.
                auto sel = [&](uint32_t j) { return j == 0 ? r0 : (j == 1 ? r1 : (j == 2 ? r2 : r3)); };
                const float4 s0 = sel(lane), s1 = sel((lane + 3) & 3), s2 = sel((lane + 2) & 3), s3 = sel((lane + 1) & 3);
                // rotation s: reader l reads from lane (l + s) & 3 what that lane selected for reader (t - s) & 3 = l
                const float4 t0 = s0;                       // slot lane
                const float4 t1 = dpp4<0x39>(s1);           // quad_perm [1,2,3,0]: reader l <- lane l + 1 : slot (l + 1) & 3
                const float4 t2 = dpp4<0x4E>(s2);           // quad_perm [2,3,0,1]: slot (l + 2) & 3
                const float4 t3 = dpp4<0x93>(s3);           // quad_perm [3,0,1,2]: slot (l + 3) & 3
                a = t0; b = t1; c = t2; d = t3;             // (slot order rotated by the lane number: a blend would index its weights alike)
            }
        } else if constexpr (MODE == 4) {       // lane32
            const float4 *p = table + (size_t)(h % (n_blocks * 2)) * 2;
            a = p[0]; b = p[1]; c = a; d = b;
        } else {                                // pair32: lane pairs, 2 instructions
            const uint32_t blk = h % (n_blocks * 2);
            const uint32_t b0 = dppu<0xA0>(blk), b1 = dppu<0xF5>(blk);   // quad_perm [0,0,2,2] / [1,1,3,3]
            const float4 r0 = table[(size_t)b0 * 2 + (lane & 1)], r1 = table[(size_t)b1 * 2 + (lane & 1)];
            a = r0; b = r1; c = a; d = b;
        }
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? atol(argv[1]) : 1;
    const char *mode = argc > 2 ? argv[2] : "scatter";
    const int wps = argc > 3 ? atoi(argv[3]) : 4;
    const char *names[6] = { "scatter", "lane64", "quad", "quadraw", "lane32", "pair32" };
    int m = -1;
    for (int i = 0; i < 6; ++i) if (!strcmp(mode, names[i])) m = i;
    if (m < 0) { printf("mode?\n"); return 2; }
    const uint32_t n_blocks = (uint32_t)(mb * 1024 * 1024 / 64);
    float4 *table; float *sink;
    CK(hipMalloc(&table, (size_t)n_blocks * 64)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(table, 0, (size_t)n_blocks * 64));
    const int blocks = 256 * wps, iters = 400;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        switch (m) {
        case 0: k<0><<<blocks, 256>>>(table, n_blocks, iters, sink); break;
        case 1: k<1><<<blocks, 256>>>(table, n_blocks, iters, sink); break;
        case 2: k<2><<<blocks, 256>>>(table, n_blocks, iters, sink); break;
        case 3: k<3><<<blocks, 256>>>(table, n_blocks, iters, sink); break;
        case 4: k<4><<<blocks, 256>>>(table, n_blocks, iters, sink); break;
        default: k<5><<<blocks, 256>>>(table, n_blocks, iters, sink); break;
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double lanes = (double)blocks * 256 * iters, loads = (m >= 4 ? 2.0 : 4.0);
    // cycles of one CU per wave-instruction: the chip's CU-cycles over the wave-instructions issued
    const double wave_instr = lanes / 64 * loads, cu_cycles = best * 1e-3 * 2.4e9 * 256;
    printf("{\"mode\": \"%s\", \"table_MB\": %zu, \"waves_per_simd\": %d, \"ms\": %.3f, \"G_lane_blocks_per_s\": %.2f, \"cu_cycles_per_wave_load\": %.1f}\n",
           mode, mb, wps, best, lanes / best / 1e6, cu_cycles / wave_instr);
    return 0;
}
