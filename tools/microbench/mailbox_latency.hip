// mailbox_latency.hip — floor of a host -> GPU -> host round trip through a polled mailbox (no evaluation at all), for
// the one-unit call service (csrc/merl_scalar.hip):  where may the request live, and what does each placement cost?
//   A  request + sequence number in pinned host memory, polled by the GPU over PCIe           (what the service does)
//   B  request + sequence number in fine-grained DEVICE memory, written by the host through the PCIe BAR, polled locally
// The answer (11 floats + sequence number) always goes to pinned host memory.  One lane, bounded lifetime per launch.
//   hipcc -O3 --offload-arch=gfx950 -o mailbox_latency mailbox_latency.hip && ./mailbox_latency
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

struct Box { uint32_t seq; float req[9]; uint32_t pad[6]; };          // 64 B
struct Ans { float out[11]; uint32_t done; uint32_t pad[4]; };        // 64 B

__global__ void k_echo(Box *box, Ans *ans, uint64_t ticks, int words)
{
    uint32_t last = __hip_atomic_load(&ans->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
        const uint32_t q = __hip_atomic_load(&box->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (q == last) continue;
        __threadfence_system();
        float acc = 0.0f;
        for (int k = 0; k < words; ++k) acc += __uint_as_float(__hip_atomic_load((uint32_t *)&box->req[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        for (int k = 0; k < 11; ++k) __hip_atomic_store((uint32_t *)&ans->out[k], __float_as_uint(acc + (float)k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __hip_atomic_store(&ans->done, q, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        last = q;
    }
}

static uint64_t g_ticks_per_us = 100;

static double run(Box *box_host_view, Box *box_dev, Ans *ans, Ans *ans_dev, hipStream_t s, int words, int calls)
{
    using Clock = std::chrono::steady_clock;
    uint32_t seq = ans->done;
    double total = 0.0;
    int done_calls = 0;
    while (done_calls < calls) {
        hipLaunchKernelGGL(k_echo, dim3(1), dim3(1), 0, s, box_dev, ans_dev, 2000 * g_ticks_per_us, words);   // 2 ms instances
        const auto until = Clock::now() + std::chrono::microseconds(1500);
        while (Clock::now() < until && done_calls < calls) {
            for (int k = 0; k < 9; ++k) box_host_view->req[k] = (float)(seq + k);
            ++seq;
            const auto t0 = Clock::now();
            __atomic_store_n(&box_host_view->seq, seq, __ATOMIC_RELEASE);
            bool lost = false;
            while (__atomic_load_n(&ans->done, __ATOMIC_ACQUIRE) != seq) {
                __builtin_ia32_pause();
                if (Clock::now() - t0 > std::chrono::milliseconds(5)) {          // the instance expired under us: start another, do not count the call
                    hipLaunchKernelGGL(k_echo, dim3(1), dim3(1), 0, s, box_dev, ans_dev, 2000 * g_ticks_per_us, words);
                    while (__atomic_load_n(&ans->done, __ATOMIC_ACQUIRE) != seq) {
                        __builtin_ia32_pause();
                        if (Clock::now() - t0 > std::chrono::seconds(2)) { std::printf(", \"stalled\": true}\n"); std::fflush(stdout); std::_Exit(3); }
                    }
                    lost = true;
                }
            }
            if (lost) break;
            total += std::chrono::duration<double, std::micro>(Clock::now() - t0).count();
            ++done_calls;
        }
        (void)hipStreamSynchronize(s);
    }
    return total / calls;
}

int main()
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { std::fprintf(stderr, "no device\n"); return 2; }
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0) == hipSuccess && khz > 0) g_ticks_per_us = (uint64_t)khz / 1000;
    std::printf("{\"device\": \"%s\", \"isLargeBar\": %d, \"wall_clock_kHz\": %d", p.name, p.isLargeBar, khz);
    std::fflush(stdout);
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    Box *hbox = nullptr; Ans *ans = nullptr;
    (void)hipHostMalloc((void **)&hbox, sizeof(Box), hipHostMallocMapped | hipHostMallocCoherent);
    (void)hipHostMalloc((void **)&ans, sizeof(Ans), hipHostMallocMapped | hipHostMallocCoherent);
    std::memset(hbox, 0, sizeof *hbox); std::memset(ans, 0, sizeof *ans);
    Box *hbox_dev; Ans *ans_dev;
    (void)hipHostGetDevicePointer((void **)&hbox_dev, hbox, 0); (void)hipHostGetDevicePointer((void **)&ans_dev, ans, 0);
    (void)run(hbox, hbox_dev, ans, ans_dev, s, 9, 200);
    std::printf(", \"A_host_mailbox_us\": %.3f", run(hbox, hbox_dev, ans, ans_dev, s, 9, 3000)); std::fflush(stdout);
    std::printf(", \"A_host_mailbox_no_request_words_us\": %.3f", run(hbox, hbox_dev, ans, ans_dev, s, 0, 3000)); std::fflush(stdout);
    if (p.isLargeBar) {
        Box *dbox = nullptr;
        if (hipExtMallocWithFlags((void **)&dbox, sizeof(Box), hipDeviceMallocFinegrained) == hipSuccess) {
            (void)hipMemset(dbox, 0, sizeof(Box));
            (void)hipDeviceSynchronize();
            (void)run(dbox, dbox, ans, ans_dev, s, 9, 200);                   // host writes straight into device memory
            std::printf(", \"B_device_mailbox_us\": %.3f", run(dbox, dbox, ans, ans_dev, s, 9, 3000));
        } else {
            (void)hipGetLastError();
            std::printf(", \"B_device_mailbox_us\": null");
        }
    }
    std::printf("}\n");
    return 0;
}
