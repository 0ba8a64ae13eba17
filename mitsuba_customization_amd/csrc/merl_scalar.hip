// merl_scalar.hip — the service kernel behind mrl_scalar_eval_sample (one-unit calls without a launch per call).
//
// One lane per mailbox slot (merl_scalar_board.hpp).  A lane polls its slot in pinned host memory with three 16-byte
// system-scope loads — the whole request, each chunk carrying the call's sequence number; when all three carry a new
// number the lane evaluates the fused unit — eval(wi, wo), pdf(wi, wo), sample(wi, u) — with the SAME per-lane functions
// the batch kernels use (merl_table_fast.hpp / merl_ggx_fast.hpp: k_table's and k_ggx's arithmetic, so a scalar call
// returns what a batch call returns) and writes the eleven floats back as four 16-byte stores, each carrying the number.
// Requests of concurrent callers sit in different lanes and are served side by side.  (Tried: the slots dealt over eight
// waves instead of two, so that callers do not wait for each other's evaluation — slower at every thread count, 0.84 vs
// 0.59 us per call amortised over 16 threads: the PCIe transactions of the polls, not the evaluation, are what callers
// queue behind, and more polling waves mean more of them.  Also tried: chunk c of four neighbouring slots in one 64-byte
// line, so that a quad's four accesses could cross PCIe as one transaction — no gain either, 0.76 us: the callers then
// share cache lines on the host.)
// Every wave reaches the exit: the loop ends after `lifetime_ticks` of the 100 MHz wall clock, when the host raises
// `stop`, or after a fixed number of polls, whichever comes first; the host launches the successor.
#include "merl_kernels.hpp"
#include "merl_scalar_board.hpp"
#include "merl_table_fast.hpp"
#include "merl_ggx_fast.hpp"

namespace mrl {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t sys_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// the three request chunks of a slot: 16-byte loads that bypass every cache (sc0 sc1), all in flight together
__device__ __forceinline__ void load_request(const ScalarSlot *s, v4f &a, v4f &b, v4f &c)
{
    asm volatile("global_load_dwordx4 %0, %3, off sc0 sc1\n\t"
                 "global_load_dwordx4 %1, %3, off offset:16 sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %3, off offset:32 sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(s) : "memory");
}
// the four result chunks: 16-byte write-through stores; each arrives whole, so no fence separates payload and number
__device__ __forceinline__ void store_result(ScalarSlot *s, v4f r0, v4f r1, v4f r2, v4f r3)
{
    asm volatile("global_store_dwordx4 %0, %1, off offset:64 sc0 sc1\n\t"
                 "global_store_dwordx4 %0, %2, off offset:80 sc0 sc1\n\t"
                 "global_store_dwordx4 %0, %3, off offset:96 sc0 sc1\n\t"
                 "global_store_dwordx4 %0, %4, off offset:112 sc0 sc1"
                 :: "v"(s), "v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory");
}

// what a request asks for (bits 28-29 of the material word; 0 = both halves = the fused unit)
constexpr int kWantEvalOnly = 1, kWantSampleOnly = 2;

template <int LOOKUP, int LAYOUT>
__device__ __forceinline__ void table_unit(const MaterialDev &m, const Options &o, int want, float wix, float wiy, float wiz,
                                           float wox, float woy, float woz, float u0, float u1, float out[11])
{
    // k_table<MODE_EVAL_SAMPLE>'s lane, verbatim; a half the caller did not ask for is skipped (its outputs are zero)
    const fast::Vec3 in = fast::normalize_f32(wix, wiy, wiz);
    if (want != kWantSampleOnly) {
        fast::unit_eval<LOOKUP, LAYOUT, true>(m, o, in, wix, wiy, wiz, wox, woy, woz, out);
        float pdf = (wiz > 0.0f && woz > 0.0f) ? woz * kInvPiF : 0.0f;
        if (o.sampling && pdf > 0.0f) pdf = (float)fast::table_pdf(m, in, fast::normalize_f32(wox, woy, woz), woz, o.sampling);
        out[3] = pdf;
    }
    if (want != kWantEvalOnly) fast::unit_sample<LOOKUP, LAYOUT, true>(m, o, in, wix, wiy, wiz, u0, u1, out + 4, out[7], out + 8);
}

__device__ __forceinline__ void ggx_unit(const MaterialDev &m, int want, float wix, float wiy, float wiz, float wox, float woy, float woz,
                                         float u0, float u1, float out[11])
{
    // k_ggx<MODE_EVAL_SAMPLE>'s lane, verbatim
    const fast::GgxConsts g(m);
    const fast::Vec3 in = fast::normalize_f32(wix, wiy, wiz);
    if (want != kWantSampleOnly) {
        const fast::Vec3 o = fast::normalize_f32(wox, woy, woz);
        double v[3], p;
        fast::ggx_eval_pdf(g, in, o, v, p);
        const bool valid = (wiz > 0.0f) && (woz > 0.0f);
        const double poison = fast::cos_or_nan(wix, wiy, wiz, wox, woy, 1.0f);
        out[0] = valid ? (float)(v[0] * poison) : 0.0f; out[1] = valid ? (float)(v[1] * poison) : 0.0f; out[2] = valid ? (float)(v[2] * poison) : 0.0f;
        out[3] = valid ? (float)(p * poison) : 0.0f;
    }
    if (want != kWantEvalOnly) {
        fast::ggx_sample(g, in, u0, u1, out + 4, out[7], out + 8);
        if (!(wiz > 0.0f)) { out[4] = out[5] = out[6] = 0.0f; out[7] = 0.0f; out[8] = out[9] = out[10] = 0.0f; }
    }
}

__global__ __launch_bounds__(kScalarSlots) void k_scalar_service(ScalarArgs a)
{
    ScalarBoard *b = a.board;
    const unsigned i = threadIdx.x;
    if (i == 0) __hip_atomic_store(&b->started_gen, a.gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    ScalarSlot *s = &b->slot[i];
    uint32_t last = sys_load(&s->res[3].seq);                 // what this slot's caller has already been answered
    const uint64_t t0 = wall_clock64();
    for (uint32_t it = 0; it < a.max_polls; ++it) {
        const uint32_t active = sys_load(&b->active);         // slots beyond it have never carried a request: do not poll them
        if (i < active) {
            v4f ra, rb, rc;
            load_request(s, ra, rb, rc);
            const uint32_t q = __float_as_uint(ra.w);
            if (q != last && __float_as_uint(rb.w) == q && __float_as_uint(rc.w) == q) {      // a new request, all of it
                float wix = ra.x, wiy = ra.y, wiz = ra.z;
                const float wox = rb.x, woy = rb.y, woz = rb.z, u0 = rc.x, u1 = rc.y;
                const uint32_t word = __float_as_uint(rc.z);           // material id, and in bits 28-29 which half is wanted
                const int id = (int)(word & 0x0FFFFFFFu), want = (int)((word >> 28) & 3u);
                bool known = id >= 0 && id < a.n_materials;
                MaterialDev m = a.materials[known ? id : 0];
                known = known && kind_is_rgb_path(m.kind);
                if (!known) { m = a.safe; wiz = 0.0f; }       // the host refuses such ids before they get here; zeros if one does
                float out[11] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
                if (m.kind == KIND_GGX) {
                    ggx_unit(m, want, wix, wiy, wiz, wox, woy, woz, u0, u1, out);
                } else if (a.opts.lookup) {
                    if (m.layout == LAYOUT_BRICK) table_unit<1, LAYOUT_BRICK>(m, a.opts, want, wix, wiy, wiz, wox, woy, woz, u0, u1, out);
                    else table_unit<1, LAYOUT_ROWS>(m, a.opts, want, wix, wiy, wiz, wox, woy, woz, u0, u1, out);
                } else {
                    if (m.layout == LAYOUT_BRICK) table_unit<0, LAYOUT_BRICK>(m, a.opts, want, wix, wiy, wiz, wox, woy, woz, u0, u1, out);
                    else table_unit<0, LAYOUT_ROWS>(m, a.opts, want, wix, wiy, wiz, wox, woy, woz, u0, u1, out);
                }
                const float qf = __uint_as_float(q);
                const v4f r0 = { out[0], out[1], out[2], qf }, r1 = { out[3], out[4], out[5], qf }, r2 = { out[6], out[7], out[8], qf },
                          r3 = { out[9], out[10], 0.0f, qf };
                store_result(s, r0, r1, r2, r3);
                last = q;
            }
        }
        if (wall_clock64() - t0 > a.lifetime_ticks) break;
        if ((it & 7u) == 7u && sys_load(&b->stop)) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the last answers have left before the instance reports its exit
    __syncthreads();
    if (i == 0) __hip_atomic_store(&b->exited_gen, a.gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

} // namespace

hipError_t launch_scalar_service(const ScalarArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL(k_scalar_service, dim3(1), dim3(kScalarSlots), 0, stream, a);
    return hipGetLastError();
}

} // namespace mrl
