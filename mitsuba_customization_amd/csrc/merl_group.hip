// merl_group.hip — device groups: one host process driving several GPUs (include/merl_hip.h, "device groups").
//
// Built on the single-device C ABI: a group owns one mrl_ctx per member plus, per member, a compute stream (set as
// the context's stream), a transfer stream, two chunk-sized result buffers and the events that chain them.
// The path shards with no data-path collective; the only communication is the result gather:
//
//   step c:   member r != root     compute chunk c into buffer c%2   (compute stream; waits until send c-2 left that buffer)
//             root                 compute chunk c straight into the caller's arrays
//             transfer streams     wait for the chunk's compute, then move it into the root's arrays:
//                                  RCCL: one ncclGroupStart/End holding 5 ncclSend per peer and the matching 5 ncclRecv
//                                  on the root (point-to-point: every peer's own xGMI link to the root, not a ring);
//                                  PEER_COPY: 5 hipMemcpyPeerAsync per peer
//   step c+1: computes overlap those transfers.
//
// Sizing: BASELINE config 5 moves 44 B/unit x 875M units = 38.5 GB into the root, >= 36 ms at
// 7 x 153 GB/s, against ~6 ms of compute per member: the gather dominates end to end, which is why it is chunked
// and overlapped and why bench.py reports it beside, not inside, the compute throughput.
// librccl is loaded on demand (dlopen): a single-GPU host never needs it.
#include "../../include/merl_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace mrlabi { hipError_t create_compute_stream(int device_cus, int reserved, hipStream_t *out); }     // merl_abi.hip

#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;

    bool load()
    {
        if (handle) return true;
        // torch's bundled librccl.so (same SONAME) is reused when it is already in the process
        for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) { error = std::string("dlopen(librccl): ") + dlerror(); return false; }
#define MRL_SYM(field, sym)                                                                  \
        field = (decltype(field))dlsym(handle, sym);                                         \
        if (!field) { error = std::string("librccl lacks ") + sym; return false; }
        MRL_SYM(CommInitAll, "ncclCommInitAll")
        MRL_SYM(CommDestroy, "ncclCommDestroy")
        MRL_SYM(GroupStart, "ncclGroupStart")
        MRL_SYM(GroupEnd, "ncclGroupEnd")
        MRL_SYM(Send, "ncclSend")
        MRL_SYM(Recv, "ncclRecv")
        MRL_SYM(GetErrorString, "ncclGetErrorString")
#undef MRL_SYM
        return true;
    }
};

struct Member {
    int device = 0;
    mrl_ctx *ctx = nullptr;
    hipStream_t compute = nullptr, transfer = nullptr;
    hipEvent_t done[2] = { nullptr, nullptr };     // chunk computed into buffer s
    hipEvent_t sent[2] = { nullptr, nullptr };     // buffer s has left the device (free for chunk c+2)
    bool sent_valid[2] = { false, false };
    hipEvent_t t0 = nullptr, t1 = nullptr;         // timing of the last sharded call (compute stream / end marker)
    hipEvent_t landed = nullptr;                   // transfer stream: everything posted so far has arrived
    float *buf[2] = { nullptr, nullptr };          // 11 floats per unit: rgb[3] pdf wo[3] pdf2 weight[3], SoA per chunk
    size_t buf_units = 0;
    // mrl_group_generate_tiles
    float *gen = nullptr;                          // wi[3n] wo[3n] u[2n] mat[n]
    size_t gen_units = 0;
    bool timed = false;
    ncclComm_t comm = nullptr;
};

} // namespace

struct mrl_group {
    std::vector<Member> members;
    int transport = MRL_TRANSPORT_PEER_COPY;
    Rccl rccl;
    std::string last_error;
};

namespace {

// why the last mrl_group_init on this thread failed (the caller has no group to ask): mrl_group_last_error(NULL)
thread_local std::string t_init_error;

int gfail(mrl_group *g, int status, const std::string &msg)
{
    if (g) g->last_error = msg;
    return status;
}

#define MRL_GHIP(g, expr)                                                                    \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            (void)hipGetLastError();                                                         \
            return gfail((g), _e == hipErrorOutOfMemory ? MRL_ERR_OOM : MRL_ERR_HIP,         \
                         std::string(#expr) + ": " + hipGetErrorString(_e));                 \
        }                                                                                    \
    } while (0)

#define MRL_GNCCL(g, expr)                                                                   \
    do {                                                                                     \
        ncclResult_t _r = (expr);                                                            \
        if (_r != ncclSuccess)                                                               \
            return gfail((g), MRL_ERR_COMM, std::string(#expr) + ": " + (g)->rccl.GetErrorString(_r)); \
    } while (0)

// a member call failed: carry its context's message into the group's
int member_fail(mrl_group *g, int rank, int rc, const char *what)
{
    const char *d = mrl_last_error(g->members[(size_t)rank].ctx);
    return gfail(g, rc, std::string(what) + " on member " + std::to_string(rank) + ": " + ((d && *d) ? d : mrl_strerror(rc)));
}

void free_member(Member &m)
{
    (void)hipSetDevice(m.device);
    if (m.compute) (void)hipStreamSynchronize(m.compute);
    if (m.transfer) (void)hipStreamSynchronize(m.transfer);
    for (int s = 0; s < 2; ++s) {
        if (m.buf[s]) (void)hipFree(m.buf[s]);
        if (m.done[s]) (void)hipEventDestroy(m.done[s]);
        if (m.sent[s]) (void)hipEventDestroy(m.sent[s]);
    }
    if (m.gen) (void)hipFree(m.gen);
    if (m.t0) (void)hipEventDestroy(m.t0);
    if (m.t1) (void)hipEventDestroy(m.t1);
    if (m.landed) (void)hipEventDestroy(m.landed);
    if (m.ctx) (void)mrl_destroy(m.ctx);              // before its stream goes
    if (m.compute) (void)hipStreamDestroy(m.compute);
    if (m.transfer) (void)hipStreamDestroy(m.transfer);
    m = Member();
}

int ensure_buffers(mrl_group *g, Member &m, size_t units)
{
    if (units <= m.buf_units) return MRL_OK;
    MRL_GHIP(g, hipSetDevice(m.device));
    MRL_GHIP(g, hipStreamSynchronize(m.compute));
    MRL_GHIP(g, hipStreamSynchronize(m.transfer));
    for (int s = 0; s < 2; ++s) {
        if (m.buf[s]) { (void)hipFree(m.buf[s]); m.buf[s] = nullptr; }
        m.sent_valid[s] = false;
    }
    m.buf_units = 0;
    for (int s = 0; s < 2; ++s) MRL_GHIP(g, hipMalloc((void **)&m.buf[s], units * 11 * sizeof(float)));
    m.buf_units = units;
    return MRL_OK;
}

// the five output arrays of one chunk inside a member buffer of capacity cap units
struct ChunkOut { float *rgb, *pdf, *wo, *pdf2, *weight; };
ChunkOut chunk_out(float *base, size_t cap)
{
    return { base, base + 3 * cap, base + 4 * cap, base + 7 * cap, base + 8 * cap };
}

template <typename F>
int for_each_member(mrl_group *g, const char *what, F &&call)
{
    for (size_t r = 0; r < g->members.size(); ++r) {
        const int rc = call(g->members[r], (int)r);
        if (rc != MRL_OK) return member_fail(g, (int)r, rc, what);
    }
    return MRL_OK;
}

// the same upload on every member; ids must agree (they do: members see the same sequence of uploads and releases)
template <typename F>
int replicated_material(mrl_group *g, const char *what, int *out_id, F &&call)
{
    if (!g || !out_id) return MRL_ERR_INVALID;
    int first = -1;
    for (size_t r = 0; r < g->members.size(); ++r) {
        int id = -1;
        const int rc = call(g->members[r].ctx, &id);
        if (rc != MRL_OK) {
            for (size_t q = 0; q < r; ++q) (void)mrl_material_release(g->members[q].ctx, first);     // all or nothing
            return member_fail(g, (int)r, rc, what);
        }
        if (r == 0) first = id;
        else if (id != first) {
            // undo what this call has placed — `first` on the members before r, `id` on member r — so that the tables do not
            // stay resident and the members' id sequences are what they were
            for (size_t q = 0; q < r; ++q) (void)mrl_material_release(g->members[q].ctx, first);
            (void)mrl_material_release(g->members[r].ctx, id);
            return gfail(g, MRL_ERR_MATERIAL, "members disagree on the material id (a member context was used directly)");
        }
    }
    *out_id = first;
    return MRL_OK;
}

// host arrays of n units split into tiles, one host thread per member; call(ctx, lo, hi) runs the member's tile
template <typename F>
int host_split(mrl_group *g, size_t n, const char *what, F &&call)
{
    const int G = (int)g->members.size();
    std::vector<int> rcs((size_t)G, MRL_OK);
    std::vector<std::thread> pool;
    for (int r = 0; r < G; ++r) {
        size_t lo, hi;
        mrl_tile_bounds(n, G, r, &lo, &hi);
        if (hi <= lo) continue;
        pool.emplace_back([=, &rcs, &call]() {
            mrl_ctx *c = g->members[(size_t)r].ctx;
            rcs[(size_t)r] = call(c, lo, hi);
            if (rcs[(size_t)r] == MRL_OK) rcs[(size_t)r] = mrl_synchronize(c);          // device-accessible (pinned) arrays are async
        });
    }
    for (auto &t : pool) t.join();
    for (int r = 0; r < G; ++r)
        if (rcs[(size_t)r] != MRL_OK) return member_fail(g, r, rcs[(size_t)r], what);
    return MRL_OK;
}

} // namespace

extern "C" {

void mrl_tile_bounds(size_t n_total, int world, int rank, size_t *lo, size_t *hi)
{
    size_t a = 0, b = 0;
    if (world >= 1 && rank >= 0 && rank < world && n_total > 0) {
        const size_t per = (n_total + (size_t)world - 1) / (size_t)world;
        a = std::min(n_total, (size_t)rank * per);
        b = std::min(n_total, a + per);
    }
    if (lo) *lo = a;
    if (hi) *hi = b;
}

size_t mrl_chunk_steps(size_t n_total, int world, size_t chunk_units)
{
    if (world < 1 || n_total == 0 || chunk_units == 0) return 0;
    const size_t per = (n_total + (size_t)world - 1) / (size_t)world;
    return (per + chunk_units - 1) / chunk_units;
}

void mrl_chunk_bounds(size_t n_total, int world, int rank, size_t chunk_units, size_t step, size_t *lo, size_t *hi)
{
    size_t tlo, thi;
    mrl_tile_bounds(n_total, world, rank, &tlo, &thi);
    size_t a = thi, b = thi;
    if (chunk_units > 0 && step < (thi - tlo + chunk_units - 1) / chunk_units) {
        a = tlo + step * chunk_units;
        b = std::min(thi, a + chunk_units);
    }
    if (lo) *lo = a;
    if (hi) *hi = b;
}

// The schedule of one sharded call as a list of operations, in issue order — pure arithmetic, no device.  The call itself
// (sharded() below) walks exactly this list, so that the buffer / offset / dependency bookkeeping of the pipeline can be
// checked on a CPU against a stub transport (tests/test_group_cpu.py) and a multi-GPU run only has to validate the
// transport itself.  Per step c: one COMPUTE per member with units left (the root into the caller's arrays, a peer
// into its chunk buffer c % 2, after the transfer that last read that buffer — step c - 2 — has left it), then one
// TRANSFER per such peer out of that buffer into the root's arrays at the chunk's global offset.
size_t mrl_group_plan(size_t n_total, int world, size_t chunk_units, int root, mrl_plan_op *ops, size_t max_ops)
{
    if (world < 1 || root < 0 || root >= world || chunk_units == 0 || n_total == 0) return 0;
    const size_t steps = mrl_chunk_steps(n_total, world, chunk_units);
    size_t n_ops = 0;
    auto emit = [&](const mrl_plan_op &op) { if (ops && n_ops < max_ops) ops[n_ops] = op; ++n_ops; };
    for (size_t c = 0; c < steps; ++c) {
        const int s = (int)(c & 1);
        for (int pass = 0; pass < 2; ++pass)                  // every compute of the step, then its transfers
            for (int r = 0; r < world; ++r) {
                if (pass == 1 && r == root) continue;
                size_t tlo, thi, a, b;
                mrl_tile_bounds(n_total, world, r, &tlo, &thi);
                mrl_chunk_bounds(n_total, world, r, chunk_units, c, &a, &b);
                if (b <= a) continue;
                mrl_plan_op op;
                op.kind = pass == 0 ? MRL_PLAN_COMPUTE : MRL_PLAN_TRANSFER;
                op.step = c; op.member = r;
                op.buffer = r == root ? -1 : s;
                op.first = a; op.count = b - a; op.tile_offset = a - tlo;
                op.after_transfer_of_step = (pass == 0 && r != root && c >= 2) ? (long long)(c - 2) : -1;
                emit(op);
            }
    }
    return n_ops;
}

int mrl_group_init(int n_devices, const int *device_ids, int transport, mrl_group **out)
{
    if (!out) return MRL_ERR_INVALID;
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64 || !device_ids) return MRL_ERR_INVALID;
    if (transport < MRL_TRANSPORT_AUTO || transport > MRL_TRANSPORT_PEER_COPY) return MRL_ERR_INVALID;
    bool distinct = true;
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j) distinct = distinct && device_ids[i] != device_ids[j];
    if (transport == MRL_TRANSPORT_RCCL && !distinct) return MRL_ERR_INVALID;       // RCCL refuses two ranks on one GPU
    if (transport == MRL_TRANSPORT_AUTO) transport = (distinct && n_devices > 1) ? MRL_TRANSPORT_RCCL : MRL_TRANSPORT_PEER_COPY;
    mrl_group *g = new (std::nothrow) mrl_group();
    if (!g) return MRL_ERR_OOM;
    g->transport = transport;
    g->members.resize((size_t)n_devices);
    int rc = MRL_OK;
    for (int r = 0; r < n_devices && rc == MRL_OK; ++r) {
        Member &m = g->members[(size_t)r];
        m.device = device_ids[r];
        rc = mrl_init(m.device, &m.ctx);
        if (rc != MRL_OK) break;
        hipError_t e = hipSetDevice(m.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m.compute, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m.transfer, hipStreamNonBlocking);
        for (int s = 0; s < 2 && e == hipSuccess; ++s) {
            e = hipEventCreateWithFlags(&m.done[s], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&m.sent[s], hipEventDisableTiming);
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m.landed, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreate(&m.t0);
        if (e == hipSuccess) e = hipEventCreate(&m.t1);
        if (e != hipSuccess) { (void)hipGetLastError(); rc = MRL_ERR_HIP; break; }
        rc = mrl_set_stream(m.ctx, (void *)m.compute);
    }
    if (rc == MRL_OK && distinct && n_devices > 1) {
        // direct peer mappings where the topology offers them (xGMI inside a node); copies work without, slower
        for (int a = 0; a < n_devices; ++a)
            for (int b = 0; b < n_devices; ++b) {
                if (a == b) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, device_ids[a], device_ids[b]) == hipSuccess && can) {
                    (void)hipSetDevice(device_ids[a]);
                    const hipError_t e = hipDeviceEnablePeerAccess(device_ids[b], 0);
                    if (e != hipSuccess) (void)hipGetLastError();                   // already enabled is fine
                }
            }
    }
    if (rc == MRL_OK && transport == MRL_TRANSPORT_RCCL) {
        if (!g->rccl.load()) { g->last_error = g->rccl.error; rc = MRL_ERR_COMM; }
        else {
            std::vector<ncclComm_t> comms((size_t)n_devices);
            const ncclResult_t nr = g->rccl.CommInitAll(comms.data(), n_devices, device_ids);
            if (nr != ncclSuccess) { g->last_error = std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(nr); rc = MRL_ERR_COMM; }
            else for (int r = 0; r < n_devices; ++r) g->members[(size_t)r].comm = comms[(size_t)r];
        }
    }
    if (rc != MRL_OK) {
        t_init_error = g->last_error.empty() ? std::string("member context: ") + mrl_strerror(rc) : g->last_error;
        mrl_group_destroy(g);
        return rc;
    }
    *out = g;
    return MRL_OK;
}

int mrl_group_destroy(mrl_group *g)
{
    if (!g) return MRL_OK;
    for (Member &m : g->members) {
        if (m.compute) { (void)hipSetDevice(m.device); (void)hipStreamSynchronize(m.compute); }
        if (m.transfer) { (void)hipSetDevice(m.device); (void)hipStreamSynchronize(m.transfer); }
    }
    for (Member &m : g->members)
        if (m.comm && g->rccl.CommDestroy) { (void)g->rccl.CommDestroy(m.comm); m.comm = nullptr; }
    for (Member &m : g->members) free_member(m);
    delete g;
    return MRL_OK;
}

int mrl_group_size(const mrl_group *g) { return g ? (int)g->members.size() : MRL_ERR_INVALID; }
int mrl_group_transport(const mrl_group *g) { return g ? g->transport : MRL_ERR_INVALID; }
const char *mrl_group_last_error(const mrl_group *g) { return g ? g->last_error.c_str() : t_init_error.c_str(); }

int mrl_group_context(mrl_group *g, int rank, mrl_ctx **out)
{
    if (!g || !out || rank < 0 || (size_t)rank >= g->members.size()) return MRL_ERR_INVALID;
    *out = g->members[(size_t)rank].ctx;
    return MRL_OK;
}

int mrl_group_set_option(mrl_group *g, int option, int value)
{
    if (!g) return MRL_ERR_INVALID;
    if (option == MRL_OPT_RESERVED_CUS) {
        // every member's COMPUTE stream gets the CU mask (the group owns those streams); the transfer streams — where RCCL's
        // send / receive kernels run — stay unrestricted, so the reserved CUs are theirs while the persistent grids run
        return for_each_member(g, "mrl_set_option(reserved CUs)", [&](Member &m, int) {
            int rc = mrl_set_option(m.ctx, option, value);                       // validates the value, sizes the grids
            if (rc != MRL_OK) return rc;
            int name_cus = 0;
            rc = mrl_device_info(m.ctx, nullptr, 0, &name_cus, nullptr);
            if (rc != MRL_OK) return rc;
            if (hipSetDevice(m.device) != hipSuccess || hipStreamSynchronize(m.compute) != hipSuccess) { (void)hipGetLastError(); return (int)MRL_ERR_HIP; }
            hipStream_t fresh = nullptr;
            if (mrlabi::create_compute_stream(name_cus + value, value, &fresh) != hipSuccess)      // (mrl_device_info reports the CUs left to the grids) { (void)hipGetLastError(); return (int)MRL_ERR_HIP; }
            (void)hipStreamDestroy(m.compute);
            m.compute = fresh;
            return mrl_set_stream(m.ctx, (void *)m.compute);
        });
    }
    return for_each_member(g, "mrl_set_option", [&](Member &m, int) { return mrl_set_option(m.ctx, option, value); });
}

int mrl_group_material_load_merl(mrl_group *g, const char *path, int *out_id)
{
    return replicated_material(g, "mrl_material_load_merl", out_id, [&](mrl_ctx *c, int *id) { return mrl_material_load_merl(c, path, id); });
}
int mrl_group_material_upload_f64(mrl_group *g, const double *planar_rgb, int *out_id)
{
    return replicated_material(g, "mrl_material_upload_f64", out_id, [&](mrl_ctx *c, int *id) { return mrl_material_upload_f64(c, planar_rgb, id); });
}
int mrl_group_material_upload_table(mrl_group *g, const double *planar_rgb, const int dims[3], const double scale[3], int *out_id)
{
    return replicated_material(g, "mrl_material_upload_table", out_id,
                               [&](mrl_ctx *c, int *id) { return mrl_material_upload_table(c, planar_rgb, dims, scale, id); });
}
int mrl_group_material_ggx(mrl_group *g, float alpha, const float eta[3], const float k[3], int *out_id)
{
    return replicated_material(g, "mrl_material_ggx", out_id, [&](mrl_ctx *c, int *id) { return mrl_material_ggx(c, alpha, eta, k, id); });
}
int mrl_group_material_upload_rgl_spectral(mrl_group *g, const mrl_rgl_spectral_fields *fields, int *out_id)
{
    return replicated_material(g, "mrl_material_upload_rgl_spectral", out_id, [&](mrl_ctx *c, int *id) { return mrl_material_upload_rgl_spectral(c, fields, id); });
}
int mrl_group_material_upload_rgl(mrl_group *g, const mrl_rgl_fields *fields, int *out_id)
{
    return replicated_material(g, "mrl_material_upload_rgl", out_id, [&](mrl_ctx *c, int *id) { return mrl_material_upload_rgl(c, fields, id); });
}
int mrl_group_material_load_rgl(mrl_group *g, const char *path, int *out_id)
{
    return replicated_material(g, "mrl_material_load_rgl", out_id, [&](mrl_ctx *c, int *id) { return mrl_material_load_rgl(c, path, id); });
}
int mrl_group_material_release(mrl_group *g, int id)
{
    if (!g) return MRL_ERR_INVALID;
    return for_each_member(g, "mrl_material_release", [&](Member &m, int) { return mrl_material_release(m.ctx, id); });
}

int mrl_group_generate_tiles(mrl_group *g, uint64_t seed, uint64_t first_index, size_t n_total, int n_materials, mrl_tile_inputs *tiles_out)
{
    if (!g || !tiles_out || n_materials < 0) return gfail(g, MRL_ERR_INVALID, "bad argument");
    const int G = (int)g->members.size();
    for (int r = 0; r < G; ++r) {
        Member &m = g->members[(size_t)r];
        size_t lo, hi;
        mrl_tile_bounds(n_total, G, r, &lo, &hi);
        const size_t n = hi - lo;
        MRL_GHIP(g, hipSetDevice(m.device));
        if (n > m.gen_units) {
            MRL_GHIP(g, hipStreamSynchronize(m.compute));
            if (m.gen) { (void)hipFree(m.gen); m.gen = nullptr; m.gen_units = 0; }
            MRL_GHIP(g, hipMalloc((void **)&m.gen, n * 9 * sizeof(float)));
            m.gen_units = n;
        }
        const size_t cap = m.gen_units;
        float *wi = m.gen, *wo = wi + 3 * cap, *u = wo + 3 * cap;
        int32_t *mat = (int32_t *)(u + 2 * cap);
        tiles_out[r] = { wi, wo, u, n_materials > 0 ? mat : nullptr };
        if (n == 0) continue;
        int rc = mrl_generate_pairs(m.ctx, seed, first_index + lo, n, wi, wo, u);
        if (rc == MRL_OK && n_materials > 0) rc = mrl_generate_materials(m.ctx, seed, first_index + lo, n, n_materials, mat);
        if (rc != MRL_OK) return member_fail(g, r, rc, "mrl_generate_pairs");
    }
    return MRL_OK;
}

// which: 0 the fused unit (five result arrays, 44 B per unit across the links), 1 eval only (rgb, 12 B per unit)
static int sharded(mrl_group *g, int which, const mrl_tile_inputs *tiles, int32_t single_id, size_t n_total, size_t chunk_units, int root,
                   float *out_rgb, float *out_pdf, float *out_wo, float *out_pdf2, float *out_weight)
{
    const int n_arrays = which == 0 ? 5 : 1;
    if (!g) return MRL_ERR_INVALID;
    const int G = (int)g->members.size();
    if (!tiles || root < 0 || root >= G || chunk_units == 0) return gfail(g, MRL_ERR_INVALID, "bad argument");
    if (n_total == 0) return MRL_OK;
    if (!out_rgb || (which == 0 && (!out_pdf || !out_wo || !out_pdf2 || !out_weight))) return gfail(g, MRL_ERR_INVALID, "null array argument");
    const size_t per = (n_total + (size_t)G - 1) / (size_t)G;
    const size_t cap = std::min(chunk_units, per);
    Member &R = g->members[(size_t)root];
    for (int r = 0; r < G; ++r) {
        Member &m = g->members[(size_t)r];
        size_t lo, hi;
        mrl_tile_bounds(n_total, G, r, &lo, &hi);
        m.timed = hi > lo;
        if (hi > lo && (!tiles[r].wi || !tiles[r].wo || (which == 0 && !tiles[r].u))) return gfail(g, MRL_ERR_INVALID, "null tile inputs on member " + std::to_string(r));
        if (r != root && hi > lo) { const int rc = ensure_buffers(g, m, cap); if (rc != MRL_OK) return rc; }
        if (m.timed) { MRL_GHIP(g, hipSetDevice(m.device)); MRL_GHIP(g, hipEventRecord(m.t0, m.compute)); }
    }
    // the schedule: mrl_group_plan's operations in issue order (computes of a step, then its transfers).  sent_valid[]
    // survives the call: the first two steps of the NEXT call wait for this call's last transfers out of the same buffers.
    std::vector<mrl_plan_op> plan(mrl_group_plan(n_total, G, chunk_units, root, nullptr, 0));
    (void)mrl_group_plan(n_total, G, chunk_units, root, plan.data(), plan.size());
    const bool rccl = g->transport == MRL_TRANSPORT_RCCL;
    size_t at = 0;
    while (at < plan.size()) {
        const size_t c = plan[at].step;
        // 1. every member's compute of chunk c
        for (; at < plan.size() && plan[at].step == c && plan[at].kind == MRL_PLAN_COMPUTE; ++at) {
            const mrl_plan_op &op = plan[at];
            const int r = op.member;
            Member &m = g->members[(size_t)r];
            const size_t a = op.first, off = op.tile_offset, n = op.count;
            const mrl_tile_inputs &in = tiles[r];
            ChunkOut o;
            if (op.buffer < 0) o = { out_rgb + 3 * a, which == 0 ? out_pdf + a : nullptr, which == 0 ? out_wo + 3 * a : nullptr, which == 0 ? out_pdf2 + a : nullptr,
                                     which == 0 ? out_weight + 3 * a : nullptr };
            else {
                if (n > m.buf_units) return gfail(g, MRL_ERR_INVALID, "plan: a chunk exceeds the member's buffer");
                o = chunk_out(m.buf[op.buffer], m.buf_units);
                if (m.sent_valid[op.buffer]) { MRL_GHIP(g, hipSetDevice(m.device)); MRL_GHIP(g, hipStreamWaitEvent(m.compute, m.sent[op.buffer], 0)); }
            }
            const int rc = which == 0 ? mrl_eval_sample_batch(m.ctx, in.wi + 3 * off, in.wo + 3 * off, in.u + 2 * off, in.mat ? in.mat + off : nullptr,
                                                              single_id, n, o.rgb, o.pdf, o.wo, o.pdf2, o.weight)
                                      : mrl_eval_batch(m.ctx, in.wi + 3 * off, in.wo + 3 * off, in.mat ? in.mat + off : nullptr, single_id, n, o.rgb);
            if (rc != MRL_OK) return member_fail(g, r, rc, which == 0 ? "mrl_eval_sample_batch" : "mrl_eval_batch");
            if (op.buffer >= 0) {
                MRL_GHIP(g, hipSetDevice(m.device));
                MRL_GHIP(g, hipEventRecord(m.done[op.buffer], m.compute));
                MRL_GHIP(g, hipStreamWaitEvent(m.transfer, m.done[op.buffer], 0));
            }
        }
        // 2. the chunk's transfers to the root, behind the computes, on the transfer streams
        const size_t first_transfer = at;
        while (at < plan.size() && plan[at].step == c && plan[at].kind == MRL_PLAN_TRANSFER) ++at;
        if (first_transfer == at) continue;
        if (rccl) MRL_GNCCL(g, g->rccl.GroupStart());
        int posted = MRL_OK;                                  // a failure inside the group must still close it
        for (size_t t = first_transfer; t < at && posted == MRL_OK; ++t) {
            const mrl_plan_op &op = plan[t];
            const int r = op.member;
            Member &m = g->members[(size_t)r];
            const size_t a = op.first, n = op.count;
            const ChunkOut src = chunk_out(m.buf[op.buffer], m.buf_units);
            const float *from[5] = { src.rgb, src.pdf, src.wo, src.pdf2, src.weight };
            float *to[5] = { out_rgb + 3 * a, which == 0 ? out_pdf + a : nullptr, which == 0 ? out_wo + 3 * a : nullptr, which == 0 ? out_pdf2 + a : nullptr,
                             which == 0 ? out_weight + 3 * a : nullptr };
            const size_t count[5] = { 3 * n, n, 3 * n, n, 3 * n };
            for (int k = 0; k < n_arrays && posted == MRL_OK; ++k) {
                if (rccl) {
                    ncclResult_t nr = g->rccl.Send(from[k], count[k], ncclFloat, root, m.comm, m.transfer);
                    if (nr == ncclSuccess) nr = g->rccl.Recv(to[k], count[k], ncclFloat, r, R.comm, R.transfer);
                    if (nr != ncclSuccess) posted = gfail(g, MRL_ERR_COMM, std::string("ncclSend/ncclRecv: ") + g->rccl.GetErrorString(nr));
                } else {
                    hipError_t e = hipSetDevice(m.device);
                    if (e == hipSuccess) e = hipMemcpyPeerAsync(to[k], R.device, from[k], m.device, count[k] * sizeof(float), m.transfer);
                    if (e != hipSuccess) { (void)hipGetLastError(); posted = gfail(g, MRL_ERR_HIP, std::string("hipMemcpyPeerAsync: ") + hipGetErrorString(e)); }
                }
            }
        }
        if (rccl) {
            const ncclResult_t nr = g->rccl.GroupEnd();
            if (nr != ncclSuccess && posted == MRL_OK) posted = gfail(g, MRL_ERR_COMM, std::string("ncclGroupEnd: ") + g->rccl.GetErrorString(nr));
        }
        if (posted != MRL_OK) return posted;
        for (size_t t = first_transfer; t < at; ++t) {
            Member &m = g->members[(size_t)plan[t].member];
            MRL_GHIP(g, hipSetDevice(m.device));
            MRL_GHIP(g, hipEventRecord(m.sent[plan[t].buffer], m.transfer));
            m.sent_valid[plan[t].buffer] = true;
        }
    }
    // 3. order the root's context stream after everything that writes the caller's arrays, and close the timers
    //    (every record / wait is issued with the stream's own device current)
    const bool root_receives = g->transport == MRL_TRANSPORT_RCCL;
    for (int r = 0; r < G; ++r) {
        Member &m = g->members[(size_t)r];
        if (r != root || root_receives) {
            MRL_GHIP(g, hipSetDevice(m.device));
            MRL_GHIP(g, hipEventRecord(m.landed, m.transfer));
        }
    }
    MRL_GHIP(g, hipSetDevice(R.device));
    for (int r = 0; r < G; ++r)
        if (r != root || root_receives) MRL_GHIP(g, hipStreamWaitEvent(R.compute, g->members[(size_t)r].landed, 0));
    for (int r = 0; r < G; ++r) {
        Member &m = g->members[(size_t)r];
        if (!m.timed) continue;
        MRL_GHIP(g, hipSetDevice(m.device));
        if (r != root) MRL_GHIP(g, hipStreamWaitEvent(m.compute, m.landed, 0));         // a member's time includes its sends
        MRL_GHIP(g, hipEventRecord(m.t1, m.compute));
    }
    return MRL_OK;
}

int mrl_group_eval_sample_sharded(mrl_group *g, const mrl_tile_inputs *tiles, int32_t single_id, size_t n_total, size_t chunk_units, int root,
                                  float *out_rgb, float *out_pdf, float *out_wo, float *out_pdf2, float *out_weight)
{
    return sharded(g, 0, tiles, single_id, n_total, chunk_units, root, out_rgb, out_pdf, out_wo, out_pdf2, out_weight);
}

int mrl_group_eval_sharded(mrl_group *g, const mrl_tile_inputs *tiles, int32_t single_id, size_t n_total, size_t chunk_units, int root, float *out_rgb)
{
    return sharded(g, 1, tiles, single_id, n_total, chunk_units, root, out_rgb, nullptr, nullptr, nullptr, nullptr);
}

int mrl_group_eval_sample_batch(mrl_group *g, const float *wi, const float *wo, const float *u, const int32_t *mat, int32_t single_id, size_t n,
                                float *out_rgb, float *out_pdf, float *out_wo, float *out_pdf2, float *out_weight)
{
    if (!g) return MRL_ERR_INVALID;
    if (n == 0) return MRL_OK;
    if (!wi || !wo || !u || !out_rgb || !out_pdf || !out_wo || !out_pdf2 || !out_weight) return gfail(g, MRL_ERR_INVALID, "null array argument");
    return host_split(g, n, "mrl_eval_sample_batch", [=](mrl_ctx *c, size_t lo, size_t hi) {
        return mrl_eval_sample_batch(c, wi + 3 * lo, wo + 3 * lo, u + 2 * lo, mat ? mat + lo : nullptr, single_id, hi - lo,
                                     out_rgb + 3 * lo, out_pdf + lo, out_wo + 3 * lo, out_pdf2 + lo, out_weight + 3 * lo);
    });
}

int mrl_group_eval_batch(mrl_group *g, const float *wi, const float *wo, const int32_t *mat, int32_t single_id, size_t n, float *out_rgb)
{
    if (!g) return MRL_ERR_INVALID;
    if (n == 0) return MRL_OK;
    if (!wi || !wo || !out_rgb) return gfail(g, MRL_ERR_INVALID, "null array argument");
    return host_split(g, n, "mrl_eval_batch", [=](mrl_ctx *c, size_t lo, size_t hi) {
        return mrl_eval_batch(c, wi + 3 * lo, wo + 3 * lo, mat ? mat + lo : nullptr, single_id, hi - lo, out_rgb + 3 * lo);
    });
}

int mrl_group_pdf_batch(mrl_group *g, const float *wi, const float *wo, const int32_t *mat, int32_t single_id, size_t n, float *out_pdf)
{
    if (!g) return MRL_ERR_INVALID;
    if (n == 0) return MRL_OK;
    if (!wi || !wo || !out_pdf) return gfail(g, MRL_ERR_INVALID, "null array argument");
    return host_split(g, n, "mrl_pdf_batch", [=](mrl_ctx *c, size_t lo, size_t hi) {
        return mrl_pdf_batch(c, wi + 3 * lo, wo + 3 * lo, mat ? mat + lo : nullptr, single_id, hi - lo, out_pdf + lo);
    });
}

int mrl_group_eval_pdf_batch(mrl_group *g, const float *wi, const float *wo, const int32_t *mat, int32_t single_id, size_t n,
                             float *out_rgb, float *out_pdf)
{
    if (!g) return MRL_ERR_INVALID;
    if (n == 0) return MRL_OK;
    if (!wi || !wo || !out_rgb || !out_pdf) return gfail(g, MRL_ERR_INVALID, "null array argument");
    return host_split(g, n, "mrl_eval_pdf_batch", [=](mrl_ctx *c, size_t lo, size_t hi) {
        return mrl_eval_pdf_batch(c, wi + 3 * lo, wo + 3 * lo, mat ? mat + lo : nullptr, single_id, hi - lo, out_rgb + 3 * lo, out_pdf + lo);
    });
}

int mrl_group_sample_batch(mrl_group *g, const float *wi, const float *u, const int32_t *mat, int32_t single_id, size_t n,
                           float *out_wo, float *out_pdf, float *out_weight)
{
    if (!g) return MRL_ERR_INVALID;
    if (n == 0) return MRL_OK;
    if (!wi || !u || !out_wo || !out_pdf || !out_weight) return gfail(g, MRL_ERR_INVALID, "null array argument");
    return host_split(g, n, "mrl_sample_batch", [=](mrl_ctx *c, size_t lo, size_t hi) {
        return mrl_sample_batch(c, wi + 3 * lo, u + 2 * lo, mat ? mat + lo : nullptr, single_id, hi - lo, out_wo + 3 * lo, out_pdf + lo, out_weight + 3 * lo);
    });
}

// One payload from every peer to the root over the group's transport (or the one named), each link on its own: timed with
// events on the transfer stream and compared bit for bit on the host.  What mrl_group_eval_sample_sharded's gather does,
// reduced to the transport — run it BEFORE the pipeline on a machine whose links have never carried this traffic.
int mrl_group_link_test(mrl_group *g, size_t bytes, int transport, int root, mrl_link_report *out)
{
    if (!g || !out) return MRL_ERR_INVALID;
    const int G = (int)g->members.size();
    if (root < 0 || root >= G || bytes < 4 || bytes > ((size_t)1 << 31)) return gfail(g, MRL_ERR_INVALID, "bad argument");
    if (transport == MRL_TRANSPORT_AUTO) transport = g->transport;
    if (transport == MRL_TRANSPORT_RCCL && g->transport != MRL_TRANSPORT_RCCL)
        return gfail(g, MRL_ERR_COMM, "the group holds no RCCL communicators (it was initialised for device copies)");
    const size_t n = bytes / sizeof(float);
    Member &R = g->members[(size_t)root];
    std::vector<float> h_src(n), h_dst(n);
    for (int r = 0; r < G; ++r) {
        out[r].peer = r; out[r].ok = 1; out[r].ms = 0.0f; out[r].GBps = 0.0f; out[r].bytes = bytes; out[r].mismatches = 0;
        if (r == root) continue;
        Member &m = g->members[(size_t)r];
        float *src = nullptr, *dst = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        int status = MRL_OK;
        auto hip = [&](hipError_t e, const char *what) {
            if (e != hipSuccess && status == MRL_OK) { (void)hipGetLastError(); status = gfail(g, MRL_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
            return e == hipSuccess;
        };
        for (size_t i = 0; i < n; ++i) { const uint32_t v = (uint32_t)(i * 2654435761u) ^ (uint32_t)(r * 0x9E3779B9u); std::memcpy(&h_src[i], &v, 4); }
        hip(hipSetDevice(m.device), "hipSetDevice") && hip(hipMalloc((void **)&src, n * sizeof(float)), "hipMalloc(src)") &&
            hip(hipMemcpy(src, h_src.data(), n * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy(src)");
        // the events belong to the device whose stream they are recorded on (an event cannot be recorded on another device's
        // stream: hipErrorInvalidHandle): the root's for RCCL — ncclRecv ends the transfer there —, the peer's for a device copy
        if (status == MRL_OK) hip(hipSetDevice(transport == MRL_TRANSPORT_RCCL ? R.device : m.device), "hipSetDevice") &&
            hip(hipEventCreate(&e0), "hipEventCreate") && hip(hipEventCreate(&e1), "hipEventCreate");
        if (status == MRL_OK) hip(hipSetDevice(R.device), "hipSetDevice") && hip(hipMalloc((void **)&dst, n * sizeof(float)), "hipMalloc(dst)") &&
            hip(hipMemset(dst, 0, n * sizeof(float)), "hipMemset(dst)") && hip(hipDeviceSynchronize(), "hipDeviceSynchronize");
        if (status == MRL_OK) {
            // timed on the stream that ends the transfer: the root's for RCCL (ncclRecv), the peer's for a device copy
            hipStream_t timed = transport == MRL_TRANSPORT_RCCL ? R.transfer : m.transfer;
            hip(hipSetDevice(transport == MRL_TRANSPORT_RCCL ? R.device : m.device), "hipSetDevice");
            hip(hipEventRecord(e0, timed), "hipEventRecord");
            if (transport == MRL_TRANSPORT_RCCL) {
                ncclResult_t nr = g->rccl.GroupStart();
                if (nr == ncclSuccess) nr = g->rccl.Send(src, n, ncclFloat, root, m.comm, m.transfer);
                if (nr == ncclSuccess) nr = g->rccl.Recv(dst, n, ncclFloat, r, R.comm, R.transfer);
                const ncclResult_t ne = g->rccl.GroupEnd();
                if (nr == ncclSuccess) nr = ne;
                if (nr != ncclSuccess) status = gfail(g, MRL_ERR_COMM, std::string("ncclSend/ncclRecv: ") + g->rccl.GetErrorString(nr));
            } else {
                hip(hipMemcpyPeerAsync(dst, R.device, src, m.device, n * sizeof(float), m.transfer), "hipMemcpyPeerAsync");
            }
            if (status == MRL_OK) {
                hip(hipEventRecord(e1, timed), "hipEventRecord");
                hip(hipSetDevice(m.device), "hipSetDevice") && hip(hipStreamSynchronize(m.transfer), "hipStreamSynchronize(peer)");
                hip(hipSetDevice(R.device), "hipSetDevice") && hip(hipStreamSynchronize(R.transfer), "hipStreamSynchronize(root)");
                float ms = 0.0f;
                if (status == MRL_OK && hip(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime")) {
                    out[r].ms = ms;
                    out[r].GBps = ms > 0.0f ? (float)((double)bytes / (double)ms / 1e6) : 0.0f;
                }
                if (status == MRL_OK && hip(hipMemcpy(h_dst.data(), dst, n * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy(dst)")) {
                    size_t bad = 0;
                    for (size_t i = 0; i < n; ++i) bad += std::memcmp(&h_dst[i], &h_src[i], 4) != 0;
                    out[r].mismatches = bad;
                    if (bad) status = gfail(g, MRL_ERR_COMM, "link test: " + std::to_string(bad) + " of " + std::to_string(n) + " words from member " +
                                                            std::to_string(r) + " arrived wrong");
                }
            }
        }
        if (src) { (void)hipSetDevice(m.device); (void)hipFree(src); }
        if (dst) { (void)hipSetDevice(R.device); (void)hipFree(dst); }
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (status != MRL_OK) { out[r].ok = 0; return status; }
    }
    return MRL_OK;
}

int mrl_group_synchronize(mrl_group *g)
{
    if (!g) return MRL_ERR_INVALID;
    for (Member &m : g->members) {
        MRL_GHIP(g, hipSetDevice(m.device));
        MRL_GHIP(g, hipStreamSynchronize(m.transfer));
        MRL_GHIP(g, hipStreamSynchronize(m.compute));
    }
    return MRL_OK;
}

int mrl_group_last_timing(mrl_group *g, float *ms_out)
{
    if (!g || !ms_out) return MRL_ERR_INVALID;
    for (size_t r = 0; r < g->members.size(); ++r) {
        Member &m = g->members[r];
        ms_out[r] = 0.0f;
        if (!m.timed) continue;
        MRL_GHIP(g, hipSetDevice(m.device));
        MRL_GHIP(g, hipEventSynchronize(m.t1));
        MRL_GHIP(g, hipEventElapsedTime(&ms_out[r], m.t0, m.t1));
    }
    return MRL_OK;
}

} // extern "C"
