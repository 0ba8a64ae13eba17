// merl_scalar_host.hpp — host side of the one-unit call service (merl_scalar_board.hpp has the why).
//
// Header-only and free of HIP, so that the same code runs in the library (Device = the HIP launcher in merl_abi.hip)
// and under ThreadSanitizer on the CPU with a std::thread standing in for the service kernel
// (tests/scalar_service_tsan.cpp).
//
//   Device requirements:   ScalarBoard *board();      the mailbox (pinned coherent host memory in the library)
//                          bool launch(uint32_t gen); enqueue ONE service instance of that generation; instances run in
//                                                     the order they were launched, one at a time
//                          bool healthy();            false once the device reported an error (callers stop waiting)
//
// Callers ("readers") never take a shared lock: each has a gate of its own (a cache line) that it raises for the
// duration of a call.  Whoever changes what a running instance reads — the material array, the options — is a
// "writer": pause() raises `paused`, waits until every gate is down, stops the running instance and returns with
// nothing in flight; resume() lets callers in again, and the first of them launches an instance with the new state.
#pragma once
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>

#include "merl_scalar_board.hpp"

namespace mrl {

enum ScalarStatus : int { SCALAR_OK = 0, SCALAR_STALLED = -1, SCALAR_LAUNCH_FAILED = -2 };

template <typename Device>
class ScalarService {
public:
    using Clock = std::chrono::steady_clock;

    explicit ScalarService(Device *dev, std::chrono::microseconds lifetime = std::chrono::microseconds(500),
                           std::chrono::milliseconds stall = std::chrono::milliseconds(4000))
        : m_dev(dev), m_lifetime(lifetime), m_stall(stall) {}

    std::chrono::microseconds lifetime() const { return m_lifetime; }

    // ---- caller side -------------------------------------------------------------------------------------------
    // enter() ... roundtrip() ... leave(): between enter and leave no writer can be active, so the caller may look at
    // whatever pause() protects (the library validates the material id there).
    int enter()
    {
        const int slot = my_slot();
        Gate &g = m_gate[slot];
        g.owner.lock();                                        // uncontended unless more than kScalarSlots threads call
        for (;;) {
            g.in_call.store(1, std::memory_order_seq_cst);
            if (!m_paused.load(std::memory_order_seq_cst)) break;
            g.in_call.store(0, std::memory_order_release);     // a writer is at work: step back, wait, try again
            while (m_paused.load(std::memory_order_acquire)) std::this_thread::yield();
        }
        return slot;
    }
    void leave(int slot)
    {
        Gate &g = m_gate[slot];
        g.in_call.store(0, std::memory_order_release);
        g.owner.unlock();
    }
    // post the request of `slot`, keep an instance alive, wait for the answer; out[11] = rgb pdf wo pdf2 weight
    int roundtrip(int slot, int32_t material, const float wi[3], const float wo[3], const float u[2], float out[11])
    {
        ScalarBoard *b = m_dev->board();
        ScalarSlot &s = b->slot[slot];
        if ((uint32_t)slot >= __atomic_load_n(&b->active, __ATOMIC_RELAXED)) {          // first call from this slot: let the device poll it
            uint32_t seen = __atomic_load_n(&b->active, __ATOMIC_RELAXED);
            while (seen <= (uint32_t)slot && !__atomic_compare_exchange_n(&b->active, &seen, (uint32_t)slot + 1, false, __ATOMIC_RELEASE, __ATOMIC_RELAXED)) {}
        }
        const uint32_t seq = ++m_gate[slot].seq;
        float mat_bits;
        std::memcpy(&mat_bits, &material, 4);
        const float words[3][3] = { { wi[0], wi[1], wi[2] }, { wo[0], wo[1], wo[2] }, { u[0], u[1], mat_bits } };
        for (int c = 0; c < 3; ++c) {                              // payload first, the chunk's sequence number last
            std::memcpy(s.req[c].v, words[c], 12);
            __atomic_store_n(&s.req[c].seq, seq, __ATOMIC_RELEASE);
        }
        // Somebody must look after the instance chain every quarter lifetime; everybody else goes straight to waiting
        // (no shared line is written on the way).  A caller that waits longer than a round trip looks again below.
        const Clock::time_point t0 = Clock::now();
        int rc = SCALAR_OK;
        if (t0.time_since_epoch().count() >= m_next_look.load(std::memory_order_relaxed)) {
            rc = keep_alive();
            if (rc != SCALAR_OK) return rc;
        }
        for (unsigned spins = 1;; ++spins) {
            bool all = true;
            for (int c = 0; c < 4; ++c) all = all && __atomic_load_n(&s.res[c].seq, __ATOMIC_ACQUIRE) == seq;
            if (all) break;
            relax();
            if ((spins & 255u) == 0) {                         // every few microseconds: is an instance still there?
                rc = keep_alive();
                if (rc != SCALAR_OK) return rc;
                if (!m_dev->healthy() || Clock::now() - t0 > m_stall) return SCALAR_STALLED;
                if (spins > 4096u) std::this_thread::yield();  // far beyond a round trip: more callers than cores? let the others run
            }
        }
        std::memcpy(out, s.res[0].v, 12); std::memcpy(out + 3, s.res[1].v, 12); std::memcpy(out + 6, s.res[2].v, 12); std::memcpy(out + 9, s.res[3].v, 8);
        return SCALAR_OK;
    }

    // ---- writer side (one writer at a time: the library calls these under the context's lock) --------------------
    // returns false if a running instance did not stop within the stall limit (the device is then unusable)
    bool pause()
    {
        m_paused.store(1, std::memory_order_seq_cst);
        for (Gate &g : m_gate)
            while (g.in_call.load(std::memory_order_seq_cst)) std::this_thread::yield();
        std::lock_guard<std::mutex> lk(m_launch);
        ScalarBoard *b = m_dev->board();
        bool ok = true;
        if (b && m_launched != __atomic_load_n(&b->exited_gen, __ATOMIC_ACQUIRE)) {
            __atomic_store_n(&b->stop, 1u, __ATOMIC_RELEASE);
            const Clock::time_point t0 = Clock::now();
            while (__atomic_load_n(&b->exited_gen, __ATOMIC_ACQUIRE) != m_launched) {
                std::this_thread::yield();
                if (!m_dev->healthy() || Clock::now() - t0 > m_stall) { ok = false; break; }
            }
            __atomic_store_n(&b->stop, 0u, __ATOMIC_RELEASE);
        }
        return ok;
    }
    void resume()
    {
        m_next_look.store(0, std::memory_order_relaxed);       // the next caller launches an instance (with the new state) at once
        m_paused.store(0, std::memory_order_release);
    }

    uint32_t launched() { std::lock_guard<std::mutex> lk(m_launch); return m_launched; }

private:
    struct alignas(64) Gate {
        std::atomic<uint32_t> in_call{ 0 };
        std::mutex owner;
        uint32_t seq = 0;                                      // guarded by owner
    };

    static int my_slot()
    {
        static std::atomic<uint32_t> next{ 0 };
        thread_local int slot = (int)(next.fetch_add(1, std::memory_order_relaxed) % (uint32_t)kScalarSlots);
        return slot;
    }
    static void relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }

    // At most one instance running and one queued behind it.  Launch when nothing is alive, or when the running one is
    // past half its life (so that its successor is already queued when it exits: no gap in service).
    int keep_alive()
    {
        std::unique_lock<std::mutex> lk(m_launch, std::try_to_lock);
        if (!lk.owns_lock()) return SCALAR_OK;                 // somebody else is looking after it right now
        ScalarBoard *b = m_dev->board();
        const uint32_t started = __atomic_load_n(&b->started_gen, __ATOMIC_ACQUIRE);
        const uint32_t exited = __atomic_load_n(&b->exited_gen, __ATOMIC_ACQUIRE);
        const Clock::time_point now = Clock::now();
        if (started != m_seen_started) { m_seen_started = started; m_seen_at = now; }
        bool want = false;
        if (m_launched == exited) want = true;
        else if (m_launched == started && now - m_seen_at > m_lifetime / 2) want = true;
        if (want) {
            ++m_launched;
            if (!m_dev->launch(m_launched)) { --m_launched; return SCALAR_LAUNCH_FAILED; }
        }
        m_next_look.store((now + m_lifetime / 4).time_since_epoch().count(), std::memory_order_relaxed);
        return SCALAR_OK;
    }

    Device *m_dev;
    const std::chrono::microseconds m_lifetime;
    const std::chrono::milliseconds m_stall;
    Gate m_gate[kScalarSlots];
    std::atomic<uint32_t> m_paused{ 0 };
    std::atomic<Clock::rep> m_next_look{ 0 };                  // steady-clock count before which a posting caller need not call keep_alive()
    std::mutex m_launch;                                       // guards the four members below
    uint32_t m_launched = 0;                                   // generation of the newest instance handed to the device
    uint32_t m_seen_started = 0;
    Clock::time_point m_seen_at{};
};

} // namespace mrl
