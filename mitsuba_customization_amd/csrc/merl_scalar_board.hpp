// merl_scalar_board.hpp — the mailbox between host threads that make ONE-unit calls (the virtual BSDF::eval / sample /
// pdf of a stock per-ray integrator) and the service kernel that answers them (merl_scalar.hip).  Plain C++: shared by
// the device code, the host protocol (merl_scalar_host.hpp) and its CPU race test (tests/scalar_service_tsan.cpp).
//
// Why: a launch + synchronize per call costs 16-19 us.  Here a caller writes its request into a slot of
// pinned, coherent host memory; a resident wave polls the slots (one lane per slot, so concurrent callers are served side
// by side), evaluates the fused unit with the batch kernels' own per-lane functions and writes the result back.  No
// launch, no stream synchronisation on the call path.
// A service kernel instance does NOT live forever: it exits after `lifetime` of wall clock (or when `stop` is raised)
// and the callers launch its successor — so a device-wide synchronisation (hipFree, hipDeviceSynchronize) waits a
// bounded time, and a host process that dies leaves nothing spinning on the GPU.
//
// Every PCIe crossing costs ~1.2 us (tools/microbench/mailbox_latency.hip), so the layout spends as few as it can:
// request and result travel in 16-byte CHUNKS — three payload words and the call's sequence number in the fourth.  A
// 16-byte aligned access is one transaction on either side, so a chunk is seen whole or not at all, and the poll that
// discovers a new sequence number has already fetched the request (no second round trip); the result needs no fence
// between its payload and its sequence number either.  A message is complete when all of its chunks carry the number.
#pragma once
#include <stdint.h>

namespace mrl {

constexpr int kScalarSlots = 128;            // two waves; callers beyond that share slots (a mutex per slot)

struct alignas(16) ScalarChunk {
    float v[3];
    uint32_t seq;                            // written last by its producer (release), read first by its consumer (acquire)
};

struct alignas(128) ScalarSlot {             // two 64-byte lines: the caller writes the first, the device the second
    ScalarChunk req[3];                      // wi | wo | u[0] u[1] material (the integer's bits in a float)
    uint32_t pad0[4];
    ScalarChunk res[4];                      // rgb | pdf wo'[0] wo'[1] | wo'[2] pdf' weight'[0] | weight'[1] weight'[2] -
};
static_assert(sizeof(ScalarSlot) == 128, "one slot = 128 bytes");

struct alignas(128) ScalarBoard {
    uint32_t active;                         // host: slots [0, active) may carry requests (grows as threads make their first call)
    uint32_t stop;                           // host: every instance exits at its next poll
    uint32_t started_gen;                    // device: generation of the instance that started last
    uint32_t exited_gen;                     // device: generation of the instance that exited last
    uint32_t pad[28];
    ScalarSlot slot[kScalarSlots];
};

} // namespace mrl
