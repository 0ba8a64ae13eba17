// cu_mask_probe.hip — does a stream created with hipExtStreamCreateWithCUMask keep kernels off the masked compute units, and which
// bit is which CU?  Every workgroup of a long-running grid records the hardware ids of the CU it runs on (HW_REG_HW_ID: CU / SH / SE,
// HW_REG_XCC_ID: the XCD); the host counts the distinct (xcc, se, sh, cu) tuples seen with all CUs enabled and with k CUs masked the way
// libmerl_hip masks them (mrlabi::create_compute_stream: k bits cleared at even spacing).
//   hipcc -O2 --offload-arch=gfx950 -o cu_mask_probe cu_mask_probe.hip && ./cu_mask_probe      (prints one JSON object)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <set>
#include <vector>

__global__ void k_where(unsigned *out, int spin)
{
    // HW_REG_HW_ID = 4 (all 32 bits), HW_REG_XCC_ID = 20 (low 4 bits)
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    // keep the CU busy for a while so that the dispatcher has to use every CU it may use
    float x = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, 1.0000001f, 0.5f);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc | (x == 0.0f ? 1u << 31 : 0u); }
}

static hipStream_t masked_stream(int device_cus, int reserved)
{
    hipStream_t s = nullptr;
    if (reserved <= 0) { (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); return s; }
    const int words = (device_cus + 31) / 32;
    std::vector<uint32_t> mask((size_t)words, 0u);
    for (int i = 0; i < device_cus; ++i) mask[(size_t)(i >> 5)] |= 1u << (i & 31);
    for (int k = 0; k < reserved; ++k) {
        const int bit = (int)(((long long)(2 * k + 1) * device_cus) / (2 * reserved));
        mask[(size_t)(bit >> 5)] &= ~(1u << (bit & 31));
    }
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask.data()) != hipSuccess) return nullptr;
    return s;
}

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 16;
    unsigned *d = nullptr;
    if (hipMalloc((void **)&d, (size_t)blocks * 8) != hipSuccess) return 1;
    std::vector<unsigned> h((size_t)blocks * 2);
    std::printf("{\"device_cus\": %d, \"blocks\": %d, \"rows\": [", cus, blocks);
    const int ks[] = { 0, 1, 8, 16, 32, 64, 128 };
    bool first = true;
    for (int k : ks) {
        hipStream_t s = masked_stream(cus, k);
        if (!s) { std::printf("%s{\"reserved\": %d, \"failed\": \"hipExtStreamCreateWithCUMask\"}", first ? "" : ", ", k); first = false; continue; }
        hipLaunchKernelGGL(k_where, dim3(blocks), dim3(256), 0, s, d, 200000);
        if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(h.data(), d, (size_t)blocks * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        std::set<unsigned long long> seen;
        std::set<unsigned> xccs;
        for (int b = 0; b < blocks; ++b) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xFu;
            const unsigned cu = (hw >> 8) & 0xFu, sh = (hw >> 12) & 0x1u, se = (hw >> 13) & 0x7u;
            seen.insert(((unsigned long long)xcc << 24) | (se << 16) | (sh << 8) | cu);
            xccs.insert(xcc);
        }
        std::printf("%s{\"reserved\": %d, \"distinct_cus_used\": %zu, \"xcds_used\": %zu}", first ? "" : ", ", k, seen.size(), xccs.size());
        first = false;
        (void)hipStreamDestroy(s);
    }
    std::printf("]}\n");
    (void)hipFree(d);
    return 0;
}
