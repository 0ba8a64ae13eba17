// scalar_service_tsan.cpp — ThreadSanitizer run of the one-unit call service's HOST protocol (csrc/merl_scalar_host.hpp)
// with a std::thread standing in for the service kernel (csrc/merl_scalar.hip): same mailbox, same sequence numbers,
// same bounded lifetime, same stop flag.  No GPU involved.  Built and run by tests/test_sanitize_cpu.py.
//   * T caller threads x K calls: every call answered once, with the answer to ITS request;
//   * a writer thread pauses the service again and again (what uploads, releases and option changes do) and changes the
//     "context state" the fake kernel reads — no call may observe a half-changed state, no instance may run during a pause;
//   * instances expire by lifetime and are relaunched by the callers; at most one is queued behind the running one.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../mitsuba_customization_amd/csrc/merl_scalar_host.hpp"

using namespace mrl;
using Clock = std::chrono::steady_clock;

struct FakeDevice {
    ScalarBoard *b = new ScalarBoard();
    std::chrono::microseconds lifetime{ 300 };
    // "context state" a writer changes under pause(); an instance copies it at launch, like the kernel's arguments
    int state_a = 1, state_b = -1;                       // invariant: a + b == 0
    std::atomic<int> running{ 0 }, max_queue{ 0 }, instances{ 0 }, ran_while_paused{ 0 };
    std::atomic<bool> paused_flag{ false };
    std::vector<std::thread> threads;
    std::mutex chain;                                    // instances run one at a time, in launch order (a stream)
    std::mutex threads_mu;
    std::atomic<int> queued{ 0 };

    FakeDevice() { std::memset(b, 0, sizeof *b); }
    ~FakeDevice() { for (auto &t : threads) t.join(); delete b; }
    ScalarBoard *board() { return b; }
    bool healthy() { return true; }
    bool launch(uint32_t gen)
    {
        const int a = state_a, bb = state_b;             // read between enter() and leave(): no writer is active
        const int q = ++queued;
        int seen = max_queue.load();
        while (q > seen && !max_queue.compare_exchange_weak(seen, q)) {}
        std::lock_guard<std::mutex> lk(threads_mu);
        threads.emplace_back([this, gen, a, bb] { instance(gen, a, bb); });
        return true;
    }
    void instance(uint32_t gen, int a, int bb)
    {
        std::lock_guard<std::mutex> serial(chain);       // wait for the predecessor to exit
        ++instances;
        if (paused_flag.load()) ++ran_while_paused;
        __atomic_store_n(&b->started_gen, gen, __ATOMIC_RELEASE);
        uint32_t last[kScalarSlots];
        for (int i = 0; i < kScalarSlots; ++i) last[i] = __atomic_load_n(&b->slot[i].res[3].seq, __ATOMIC_RELAXED);
        const Clock::time_point t0 = Clock::now();
        for (;;) {
            const uint32_t active = __atomic_load_n(&b->active, __ATOMIC_ACQUIRE);
            for (uint32_t i = 0; i < active; ++i) {
                ScalarSlot &s = b->slot[i];
                // a chunk = payload + sequence number, one 16-byte transaction on the device; here: acquire, then the payload
                uint32_t q[3];
                float w[3][3];
                for (int c = 0; c < 3; ++c) {
                    q[c] = __atomic_load_n(&s.req[c].seq, __ATOMIC_ACQUIRE);
                    if (q[c] != last[i] && q[c] == q[0]) std::memcpy(w[c], s.req[c].v, 12);      // only a complete, new chunk is read
                }
                if (q[0] == last[i] || q[1] != q[0] || q[2] != q[0]) continue;
                int32_t material;
                std::memcpy(&material, &w[2][2], 4);
                // the "evaluation": something the caller can verify, mixed with the state the instance was launched with
                float out[11];
                for (int k = 0; k < 3; ++k) out[k] = w[0][k] + 2.0f * w[1][k];
                out[3] = w[2][0] * 10.0f + w[2][1];
                out[4] = (float)material;
                out[5] = (float)(a + bb);                // 0 unless the state was torn
                for (int k = 6; k < 11; ++k) out[k] = (float)k;
                const float padded[12] = { out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[7], out[8], out[9], out[10], 0.0f };
                for (int c = 0; c < 4; ++c) {
                    std::memcpy(s.res[c].v, padded + 3 * c, 12);
                    __atomic_store_n(&s.res[c].seq, q[0], __ATOMIC_RELEASE);
                }
                last[i] = q[0];
            }
            if (Clock::now() - t0 > lifetime || __atomic_load_n(&b->stop, __ATOMIC_ACQUIRE)) break;
        }
        if (paused_flag.load()) ++ran_while_paused;
        --queued;
        __atomic_store_n(&b->exited_gen, gen, __ATOMIC_RELEASE);
    }
};

int main()
{
    FakeDevice dev;
    ScalarService<FakeDevice> svc(&dev, dev.lifetime, std::chrono::milliseconds(20000));
    const int T = 12, K = 3000;
    std::atomic<long> wrong{ 0 }, failed{ 0 };
    std::atomic<bool> stop{ false };
    std::thread writer([&] {
        long rounds = 0;
        while (!stop.load()) {
            if (!svc.pause()) { ++failed; break; }
            dev.paused_flag.store(true);
            // nothing may be running now; change the state in two steps a reader would catch
            dev.state_a += 1;
            std::this_thread::sleep_for(std::chrono::microseconds(50));
            dev.state_b -= 1;
            dev.paused_flag.store(false);
            svc.resume();
            ++rounds;
            std::this_thread::sleep_for(std::chrono::microseconds(400));
        }
        std::printf("writer: %ld pauses\n", rounds);
    });
    std::vector<std::thread> callers;
    for (int t = 0; t < T; ++t)
        callers.emplace_back([&, t] {
            for (int k = 0; k < K; ++k) {
                const float wi[3] = { (float)t, (float)k, 1.0f }, wo[3] = { 0.5f * (float)k, -1.0f, (float)t }, u[2] = { 0.25f, (float)(k % 7) };
                float out[11];
                const int slot = svc.enter();
                const int rc = svc.roundtrip(slot, 100 * t + (k % 100), wi, wo, u, out);
                svc.leave(slot);
                if (rc != SCALAR_OK) { ++failed; continue; }
                bool ok = out[3] == u[0] * 10.0f + u[1] && out[4] == (float)(100 * t + (k % 100)) && out[5] == 0.0f;
                for (int c = 0; c < 3; ++c) ok = ok && out[c] == wi[c] + 2.0f * wo[c];
                for (int c = 6; c < 11; ++c) ok = ok && out[c] == (float)c;
                if (!ok) ++wrong;
                if ((k & 255) == 255) std::this_thread::sleep_for(std::chrono::microseconds(700));    // let instances expire: relaunch path
            }
        });
    for (auto &th : callers) th.join();
    stop.store(true);
    writer.join();
    // one last pause: afterwards nothing runs, and nothing is relaunched because nobody calls
    if (!svc.pause()) ++failed;
    const int queued_at_end = dev.queued.load();
    svc.resume();
    std::printf("instances %d, deepest queue %d, ran while paused %d, wrong %ld, failed %ld, queued at the end %d\n",
                dev.instances.load(), dev.max_queue.load(), dev.ran_while_paused.load(), wrong.load(), failed.load(), queued_at_end);
    const bool ok = wrong.load() == 0 && failed.load() == 0 && dev.ran_while_paused.load() == 0 && dev.max_queue.load() <= 2 &&
                    dev.instances.load() > 3 && queued_at_end == 0;
    std::puts(ok ? "scalar service ok" : "scalar service FAILED");
    return ok ? 0 : 1;
}
