// driver_common.hpp — shared bits of the two plugin test drivers (binary I/O with pytest).
#pragma once
#include <dlfcn.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct Pairs {
    uint64_t n = 0;
    std::vector<float> wi, wo, u;
};

// file: uint64 n, then wi[n*3], wo[n*3], u[n*2] as f32
inline Pairs read_pairs(const std::string &path)
{
    Pairs p;
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) { std::perror(path.c_str()); std::exit(2); }
    if (std::fread(&p.n, sizeof p.n, 1, f) != 1) std::exit(2);
    p.wi.resize(3 * p.n); p.wo.resize(3 * p.n); p.u.resize(2 * p.n);
    if (std::fread(p.wi.data(), 4, 3 * p.n, f) != 3 * p.n || std::fread(p.wo.data(), 4, 3 * p.n, f) != 3 * p.n ||
        std::fread(p.u.data(), 4, 2 * p.n, f) != 2 * p.n)
        std::exit(2);
    std::fclose(f);
    return p;
}

// file: per unit rgb[3] pdf wo2[3] pdf2 weight[3]  (11 floats), first the scalar-call block (m units), then the batch block (n units)
inline void write_floats(FILE *f, const std::vector<float> &v) { std::fwrite(v.data(), 4, v.size(), f); }

// The wavefront-queue entry of a BatchedBSDF: every other slot queued, inputs and outputs in pinned device-mapped
// memory (the driver has no HIP of its own: it borrows mrl_host_alloc from the libmerl_hip.so the plugin loaded).
// Queued slots must equal the whole-array results bit for bit, other slots must stay untouched.
template <typename Wave>
inline bool check_queue_call(const Wave *wave, const Pairs &p, const std::vector<float> &rgb, const std::vector<float> &pdf,
                             const std::vector<float> &wo2, const std::vector<float> &pdf2, const std::vector<float> &wgt)
{
    void *lib = dlopen("libmerl_hip.so", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) { std::fprintf(stderr, "libmerl_hip.so is not loaded: %s\n", dlerror()); return false; }
    struct mrl_ctx;
    auto init = (int (*)(int, mrl_ctx **))dlsym(lib, "mrl_init");
    auto destroy = (int (*)(mrl_ctx *))dlsym(lib, "mrl_destroy");
    auto host_alloc = (int (*)(mrl_ctx *, size_t, void **))dlsym(lib, "mrl_host_alloc");
    auto host_free = (int (*)(mrl_ctx *, void *))dlsym(lib, "mrl_host_free");
    mrl_ctx *ctx = nullptr;
    if (!init || !destroy || !host_alloc || !host_free || init(0, &ctx) != 0) return false;
    const size_t n = p.n;
    void *raw = nullptr;
    if (host_alloc(ctx, n * 80 + 64, &raw) != 0) { destroy(ctx); return false; }
    float *base = static_cast<float *>(raw);
    float *wi = base, *wo = wi + 3 * n, *u = wo + 3 * n, *o_rgb = u + 2 * n, *o_pdf = o_rgb + 3 * n, *o_wo2 = o_pdf + n,
          *o_pdf2 = o_wo2 + 3 * n, *o_w = o_pdf2 + n;
    uint32_t *queue = reinterpret_cast<uint32_t *>(o_w + 3 * n), *count = queue + n;
    std::memcpy(wi, p.wi.data(), 12 * n); std::memcpy(wo, p.wo.data(), 12 * n); std::memcpy(u, p.u.data(), 8 * n);
    const float sentinel = -7.0f;
    for (size_t i = 0; i < 11 * n; ++i) o_rgb[i] = sentinel;
    uint32_t k = 0;
    for (size_t i = 0; i < n; i += 2) queue[k++] = (uint32_t)i;
    *count = k;
    wave->evalSampleQueue(wi, wo, u, queue, count, k, o_rgb, o_pdf, o_wo2, o_pdf2, o_w);
    wave->synchronize();
    bool ok = true;
    for (size_t i = 0; i < n && ok; ++i) {
        if (i % 2 == 0)
            ok = !std::memcmp(o_rgb + 3 * i, &rgb[3 * i], 12) && !std::memcmp(o_pdf + i, &pdf[i], 4) && !std::memcmp(o_wo2 + 3 * i, &wo2[3 * i], 12) &&
                 !std::memcmp(o_pdf2 + i, &pdf2[i], 4) && !std::memcmp(o_w + 3 * i, &wgt[3 * i], 12);
        else
            ok = o_rgb[3 * i] == sentinel && o_pdf[i] == sentinel && o_wo2[3 * i + 2] == sentinel && o_pdf2[i] == sentinel && o_w[3 * i + 1] == sentinel;
        if (!ok) std::fprintf(stderr, "queue call differs at slot %zu\n", i);
    }
    host_free(ctx, raw);
    destroy(ctx);
    return ok;
}

// Residency of plugin instances (--residency): two instances that name the same file share ONE resident table, and
// create/destroy cycles return the device's free memory to where it started (mrl_material_release in the Material's
// destructor).  make(which) builds a plugin instance over file 0 or file 1, drop() destroys it.  An instance of file 1
// (the keeper) lives throughout, so the plugin's context stays and what is measured is the release of file 0's table,
// not the teardown of the context.  Device memory is read through the libmerl_hip.so the plugin loaded (the driver
// has no HIP of its own).
template <typename Make, typename Drop>
inline int check_residency(Make make, Drop drop, int cycles)
{
    void *lib = dlopen("libmerl_hip.so", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) { std::fprintf(stderr, "libmerl_hip.so is not loaded: %s\n", dlerror()); return 20; }
    struct mrl_ctx;
    auto init = (int (*)(int, mrl_ctx **))dlsym(lib, "mrl_init");
    auto destroy = (int (*)(mrl_ctx *))dlsym(lib, "mrl_destroy");
    auto mem = (int (*)(const mrl_ctx *, size_t *, size_t *, size_t *, size_t *))dlsym(lib, "mrl_memory_info");
    mrl_ctx *probe = nullptr;
    if (!init || !destroy || !mem || init(0, &probe) != 0) return 20;
    auto free_now = [&]() { size_t f = 0; mem(probe, nullptr, nullptr, &f, nullptr); return (long long)f; };
    const long long slack = 64ll << 20;   // the runtime keeps a few freed 2-MiB blocks in its pool; a table is 178 MiB
    auto *keeper = make(1);                                 // the plugin's context + the keeper's table
    const long long with_keeper = free_now();
    auto *first = make(0);
    const long long with_one = free_now();
    const long long table = with_keeper - with_one;         // what one resident table of file 0 costs
    if (table < (16ll << 20)) { std::fprintf(stderr, "a table upload took only %lld bytes?\n", table); return 21; }
    auto *second = make(0);                                 // same file: must share the resident table
    if (with_one - free_now() > slack) { std::fprintf(stderr, "a second instance of the same file took %lld more bytes\n", with_one - free_now()); return 21; }
    drop(second);
    for (int c = 0; c < cycles; ++c) {                      // scene reloads while `first` lives: shared, nothing moves
        auto *x = make(0);
        drop(x);
        if (with_one - free_now() > slack) { std::fprintf(stderr, "free memory sank by %lld bytes in shared cycle %d\n", with_one - free_now(), c); return 22; }
    }
    drop(first);                                            // last instance of file 0 gone: its table leaves HBM
    if (with_keeper - free_now() > slack) { std::fprintf(stderr, "%lld bytes stayed resident after the last instance died\n", with_keeper - free_now()); return 23; }
    for (int c = 0; c < cycles; ++c) {                      // now every cycle uploads and releases the table
        auto *x = make(0);
        if (with_keeper - free_now() < table - slack) { std::fprintf(stderr, "cycle %d: the table is not resident while its instance lives\n", c); return 24; }
        drop(x);
        if (with_keeper - free_now() > slack) { std::fprintf(stderr, "cycle %d: %lld bytes leaked\n", c, with_keeper - free_now()); return 24; }
    }
    drop(keeper);
    destroy(probe);
    std::printf("residency ok: %lld MB per table, two instances share it, %d shared + %d upload/release cycles leave free memory flat\n",
                table >> 20, cycles, cycles);
    return 0;
}
