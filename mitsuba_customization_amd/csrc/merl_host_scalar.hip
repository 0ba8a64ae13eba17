// merl_host_scalar.hip — one-unit calls evaluated on the calling CPU thread (include/merl_hip.h, mrl_host_*).
//
// SURVEY.md §8b "what calls it (2)": the scalar virtual BSDF::eval / sample / pdf of a stock per-ray integrator is one
// pair per call from every render thread — a PCIe round trip per pair (the one-unit call service, merl_scalar.hip:
// 4.8-6.7 us) can never approach what a core does in 0.2-0.4 us.  This file is that core path: the SAME per-unit
// functions the kernels run (merl_device.hpp / merl_table_fast.hpp are __host__ __device__), compiled for the host,
// over a host image of the table that holds the device's own Float texel values (rows layout, 24 MB for a MERL table).
// Nothing here is a fallback for the batch calls: those stay on the GPU, a context still needs a gfx950 device, and a
// host image is made from a RESIDENT table.  Lock-free and allocation-free on the call path: the image is immutable.
// No code of oracle/ is used, linked or mirrored here — the oracle restates BRDFRead's rotations; this is the product's
// atan2 formulation.
#include "../../include/merl_hip.h"

#include <hip/hip_runtime.h>

#include "merl_host_table.hpp"
#include "merl_table_fast.hpp"
#include "merl_ggx_fast.hpp"      // sincos_2pi (table importance sampling)

// the per-unit functions are inlined into these three entry points and pick up their target features: hardware FMA for
// the ~200 fused multiply-adds of a unit (a libm call each otherwise); the rest of the library keeps the baseline ISA
#if !defined(__HIP_DEVICE_COMPILE__)
#define MRL_HOST_FAST __attribute__((target("avx2,fma")))
#else
#define MRL_HOST_FAST
#endif

namespace {

using mrl::fast::Vec3;

template <int LOOKUP>
MRL_HOST_FAST inline void host_eval_pdf(const mrl_host_table *t, const float wi[3], const float wo[3], float rgb[3], float *pdf)
{
    const Vec3 in = mrl::fast::normalize_f32(wi[0], wi[1], wi[2]);
    mrl::fast::unit_eval<LOOKUP, mrl::LAYOUT_ROWS, true>(t->m, t->opts, in, wi[0], wi[1], wi[2], wo[0], wo[1], wo[2], rgb);
    if (pdf) {
        float p = (wi[2] > 0.0f && wo[2] > 0.0f) ? wo[2] * mrl::kInvPiF : 0.0f;
        if (t->opts.sampling && p > 0.0f) p = (float)mrl::fast::table_pdf(t->m, in, mrl::fast::normalize_f32(wo[0], wo[1], wo[2]), wo[2], t->opts.sampling);
        *pdf = p;
    }
}

template <int LOOKUP>
MRL_HOST_FAST inline void host_sample(const mrl_host_table *t, const float wi[3], const float u[2], float wo[3], float *pdf, float weight[3])
{
    const Vec3 in = mrl::fast::normalize_f32(wi[0], wi[1], wi[2]);
    mrl::fast::unit_sample<LOOKUP, mrl::LAYOUT_ROWS, true>(t->m, t->opts, in, wi[0], wi[1], wi[2], u[0], u[1], wo, *pdf, weight);
}

// the adaptive-parameterisation material: the kernels' per-unit functions (merl_rgl.hpp) over the host copy of the image
MRL_HOST_FAST inline void host_rgl_eval_pdf(const mrl_host_table *t, const float wi[3], const float wo[3], float rgb[3], float *pdf)
{
    float p;
    if (pdf) mrl::rgl::eval_pdf<true, true>(t->rgl, wi[0], wi[1], wi[2], wo[0], wo[1], wo[2], rgb, p);
    else mrl::rgl::eval_pdf<true, false>(t->rgl, wi[0], wi[1], wi[2], wo[0], wo[1], wo[2], rgb, p);
    if (pdf) *pdf = p;
}

} // namespace

extern "C" {

MRL_HOST_FAST int mrl_host_eval_pdf(const mrl_host_table *t, const float wi[3], const float wo[3], float out_rgb[3], float *out_pdf)
{
    if (!t || !wi || !wo || !out_rgb) return MRL_ERR_INVALID;
    if (t->m.kind == mrl::KIND_RGL_SPECTRAL) return MRL_ERR_MATERIAL;
    if (t->m.kind == mrl::KIND_RGL) { host_rgl_eval_pdf(t, wi, wo, out_rgb, out_pdf); return MRL_OK; }
    if (t->opts.lookup) host_eval_pdf<1>(t, wi, wo, out_rgb, out_pdf);
    else host_eval_pdf<0>(t, wi, wo, out_rgb, out_pdf);
    return MRL_OK;
}

MRL_HOST_FAST int mrl_host_sample(const mrl_host_table *t, const float wi[3], const float u[2], float out_wo[3], float *out_pdf, float out_weight[3])
{
    if (!t || !wi || !u || !out_wo || !out_pdf || !out_weight) return MRL_ERR_INVALID;
    if (t->m.kind == mrl::KIND_RGL_SPECTRAL) return MRL_ERR_MATERIAL;
    if (t->m.kind == mrl::KIND_RGL) { mrl::rgl::sample(t->rgl, wi[0], wi[1], wi[2], u[0], u[1], out_wo, *out_pdf, out_weight); return MRL_OK; }
    if (t->opts.lookup) host_sample<1>(t, wi, u, out_wo, out_pdf, out_weight);
    else host_sample<0>(t, wi, u, out_wo, out_pdf, out_weight);
    return MRL_OK;
}

MRL_HOST_FAST int mrl_host_eval_sample(const mrl_host_table *t, const float wi[3], const float wo[3], const float u[2], float out[11])
{
    if (!t || !wi || !wo || !u || !out) return MRL_ERR_INVALID;
    if (t->m.kind == mrl::KIND_RGL_SPECTRAL) return MRL_ERR_MATERIAL;
    if (t->m.kind == mrl::KIND_RGL) {
        host_rgl_eval_pdf(t, wi, wo, out, out + 3);
        mrl::rgl::sample(t->rgl, wi[0], wi[1], wi[2], u[0], u[1], out + 4, out[7], out + 8);
        return MRL_OK;
    }
    if (t->opts.lookup) { host_eval_pdf<1>(t, wi, wo, out, out + 3); host_sample<1>(t, wi, u, out + 4, out + 7, out + 8); }
    else { host_eval_pdf<0>(t, wi, wo, out, out + 3); host_sample<0>(t, wi, u, out + 4, out + 7, out + 8); }
    return MRL_OK;
}

// a spectral RGL material: W values at the wavelengths wl[0 .. W) (NULL: the file's own nodes, W = their number)
MRL_HOST_FAST int mrl_host_eval_pdf_spectral(const mrl_host_table *t, const float wi[3], const float wo[3], const float *wl, int W, float *out_values, float *out_pdf)
{
    if (!t || !wi || !wo || !out_values || W < 1) return MRL_ERR_INVALID;
    if (t->m.kind != mrl::KIND_RGL_SPECTRAL) return MRL_ERR_MATERIAL;
    if (!wl && W != t->rgl.n_wl) return MRL_ERR_INVALID;
    float p;
    if (out_pdf) { mrl::rgl::eval_pdf_spectral<true, true>(t->rgl, wi[0], wi[1], wi[2], wo[0], wo[1], wo[2], wl, W, out_values, p); *out_pdf = p; }
    else mrl::rgl::eval_pdf_spectral<true, false>(t->rgl, wi[0], wi[1], wi[2], wo[0], wo[1], wo[2], wl, W, out_values, p);
    return MRL_OK;
}

MRL_HOST_FAST int mrl_host_sample_spectral(const mrl_host_table *t, const float wi[3], const float u[2], const float *wl, int W, float out_wo[3], float *out_pdf,
                                           float *out_weight)
{
    if (!t || !wi || !u || !out_wo || !out_pdf || !out_weight || W < 1) return MRL_ERR_INVALID;
    if (t->m.kind != mrl::KIND_RGL_SPECTRAL) return MRL_ERR_MATERIAL;
    if (!wl && W != t->rgl.n_wl) return MRL_ERR_INVALID;
    mrl::rgl::sample_spectral(t->rgl, wi[0], wi[1], wi[2], u[0], u[1], wl, W, out_wo, *out_pdf, out_weight);
    return MRL_OK;
}

int mrl_host_table_retain(mrl_host_table *t)
{
    if (!t) return MRL_ERR_INVALID;
    t->refs.fetch_add(1, std::memory_order_relaxed);
    return MRL_OK;
}

int mrl_host_table_release(mrl_host_table *t)
{
    if (!t) return MRL_ERR_INVALID;
    if (t->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) delete t;
    return MRL_OK;
}

int mrl_host_table_info(const mrl_host_table *t, int dims[3], int *param, int *lookup, int *sampling, size_t *bytes)
{
    if (!t) return MRL_ERR_INVALID;
    if (dims) { dims[0] = t->m.n_th; dims[1] = t->m.n_td; dims[2] = t->m.n_pd; }
    if (param) *param = t->m.param;
    if (lookup) *lookup = t->opts.lookup;
    if (sampling) *sampling = t->opts.sampling;
    if (bytes) *bytes = t->rows.size() * sizeof(float4) + t->marginal.size() * sizeof(double) + t->rgl_image.size() * sizeof(float);
    return MRL_OK;
}

} // extern "C"
