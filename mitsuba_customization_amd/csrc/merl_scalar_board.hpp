// merl_scalar_board.hpp — the mailbox between host threads that make ONE-unit calls (the virtual BSDF::eval / sample /
// pdf of a stock per-ray integrator) and the service kernel that answers them (merl_scalar.hip).  Plain C++: shared by
// the device code, the host protocol (merl_scalar_host.hpp) and its CPU race test (tests/scalar_service_tsan.cpp).
//
// Why: a launch + synchronize per call costs 16-19 us.  Here a caller writes its request into a slot of
// pinned, coherent host memory and raises the slot's sequence number; a resident wave polls the sequence numbers (one
// lane per slot, so concurrent callers are served side by side), evaluates the fused unit with the batch kernels' own
// per-lane functions, writes the result back and publishes the sequence number it served.  No launch, no stream
// synchronisation on the call path.
// A service kernel instance does NOT live forever: it exits after `lifetime` of wall clock (or when `stop` is raised)
// and the callers launch its successor — so a device-wide synchronisation (hipFree, hipDeviceSynchronize) waits a
// bounded time, and a host process that dies leaves nothing spinning on the GPU.
#pragma once
#include <stdint.h>

namespace mrl {

constexpr int kScalarSlots = 128;            // two waves; callers beyond that share slots (a mutex per slot)

struct ScalarSlot {                          // 128 B, one cache line pair per slot: callers never share a line
    // request: the caller writes, the device reads
    float wi[3], wo[3], u[2];
    int32_t material;
    int32_t pad0;
    // result: the device writes, the caller reads.  rgb[3] pdf wo[3] pdf2 weight[3]
    float out[11];
    uint32_t done;                           // sequence number of the last request served (written last, system-scope release)
    uint32_t pad1[10];
};
static_assert(sizeof(ScalarSlot) == 128, "one slot = 128 bytes");

struct ScalarBoard {
    uint32_t seq[kScalarSlots];              // latest request per slot (the caller raises it, release): contiguous, this is what the device polls
    uint32_t stop;                           // host: every instance exits at its next poll
    uint32_t started_gen;                    // device: generation of the instance that started last
    uint32_t exited_gen;                     // device: generation of the instance that exited last
    uint32_t pad[29];
    ScalarSlot slot[kScalarSlots];
};

} // namespace mrl
