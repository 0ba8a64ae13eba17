// merl_rgl_spectral.hip — spectral RGL materials ("optional spectral channels, RGL/.bsdf"): the calls.
// A spectral file of the RGL material database holds "spectra" [n_phi][n_theta][n_wavelengths][res][res] over a "wavelengths" grid
// where the *_rgb.bsdf variant holds "rgb"; upstream Mitsuba 3's `measured` plugin, in its spectral variants, evaluates it with the
// ray's wavelengths as a third interpolated parameter.  So do these entry points: W values per unit at the wavelengths the caller
// passes per unit (wavelengths [n][W] — what hero-wavelength rendering carries per ray), or at the file's own nodes (wavelengths ==
// NULL, W = the number of nodes); pdf and the sampled direction are wavelength-free.
// PARITY UNPINNED: no spectral file, no upstream source exists offline (oracle/rgl_oracle.c, rgl_eval_pdf_spectral, is the checker).
#include "merl_ctx.hpp"

using namespace mrlabi;

namespace {

struct SpectralCall {
    int mode;                                        // 0 eval, 2 sample, 3 eval + sample, 4 eval + pdf
    const float *wi, *wo, *u, *wl;
    int W;
    int32_t id;
    size_t n;
    float *out_values, *out_pdf, *out_wo, *out_pdf2, *out_weight;
};

int run_spectral(mrl_ctx *ctx, const SpectralCall &c)
{
    if (!ctx) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    const bool has_eval = c.mode == 0 || c.mode == 3 || c.mode == 4, has_pdf = c.mode == 3 || c.mode == 4, has_sample = c.mode == 2 || c.mode == 3;
    if (c.n == 0) return MRL_OK;
    if (!c.wi || (has_eval && (!c.wo || !c.out_values)) || (has_pdf && !c.out_pdf) || (has_sample && (!c.u || !c.out_wo || !c.out_pdf2 || !c.out_weight)))
        return fail(ctx, MRL_ERR_INVALID, "null array argument");
    if (c.id < 0 || (size_t)c.id >= ctx->materials.size() || ctx->materials[(size_t)c.id].released) return fail(ctx, MRL_ERR_MATERIAL, "unknown material id");
    const MaterialHost &mh = ctx->materials[(size_t)c.id];
    if (mh.dev.kind != mrl::KIND_RGL_SPECTRAL) return fail(ctx, MRL_ERR_MATERIAL, "the spectral entry points evaluate spectral RGL materials (mrl_material_upload_rgl_spectral)");
    if (c.W < 1 || c.W > 4096) return fail(ctx, MRL_ERR_INVALID, "1..4096 wavelengths per unit");
    if (!c.wl && c.W != mh.rgl.n_wl)
        return fail(ctx, MRL_ERR_INVALID, "without a wavelength array the values are those at the file's " + std::to_string(mh.rgl.n_wl) + " wavelength nodes");
    MRL_HIP(ctx, hipSetDevice(ctx->device));
    const int kind = common_kind({ c.wi, has_eval ? c.wo : nullptr, has_sample ? c.u : nullptr, c.wl, has_eval ? c.out_values : nullptr, has_pdf ? c.out_pdf : nullptr,
                                   has_sample ? c.out_wo : nullptr, has_sample ? c.out_pdf2 : nullptr, has_sample ? c.out_weight : nullptr });
    if (kind < 0) return fail(ctx, MRL_ERR_POINTER_MIX, "host and device pointers mixed in one call");
    auto launch = [&](const float *wi, const float *wo, const float *u, const float *wl, size_t n, float *values, float *pdf, float *wo2, float *pdf2, float *w) {
        mrl::BatchArgs a;
        std::memset(&a, 0, sizeof a);
        a.wi = wi; a.wo = wo; a.u = u; a.n = n;
        a.out_rgb = values; a.out_pdf = pdf; a.out_wo = wo2; a.out_pdf2 = pdf2; a.out_weight = w;
        a.opts = ctx->opts;
        return mrl::launch_rgl_spectral(c.mode, a, mh.rgl, wl, c.W, ctx->rgl_search, ctx->compute_units, ctx->stream);
    };
    if (kind == 1) {
        MRL_HIP(ctx, launch(c.wi, c.wo, c.u, c.wl, c.n, c.out_values, c.out_pdf, c.out_wo, c.out_pdf2, c.out_weight));
        return MRL_OK;
    }
    // host arrays: staged through HBM in chunks (a renderer that holds spectral rays on the host hands over a few million at a time)
    const size_t W = (size_t)c.W;
    const size_t unit_floats = 3 + 3 + 2 + W + W + 1 + 3 + 1 + W;
    const size_t chunk = std::min(c.n, std::max<size_t>(1, std::min(ctx->host_chunk, ((size_t)256 << 20) / (unit_floats * 4))));
    float *d = nullptr;
    MRL_ALLOC(ctx, hipMalloc((void **)&d, chunk * unit_floats * sizeof(float)));
    float *d_wi = d, *d_wo = d_wi + 3 * chunk, *d_u = d_wo + 3 * chunk, *d_wl = d_u + 2 * chunk, *d_val = d_wl + W * chunk, *d_pdf = d_val + W * chunk,
          *d_wo2 = d_pdf + chunk, *d_pdf2 = d_wo2 + 3 * chunk, *d_w = d_pdf2 + chunk;
    hipError_t e = hipSuccess;
    for (size_t off = 0; off < c.n && e == hipSuccess; off += chunk) {
        const size_t m = std::min(chunk, c.n - off);
        e = hipMemcpyAsync(d_wi, c.wi + 3 * off, 12 * m, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && has_eval) e = hipMemcpyAsync(d_wo, c.wo + 3 * off, 12 * m, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && has_sample) e = hipMemcpyAsync(d_u, c.u + 2 * off, 8 * m, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && c.wl) e = hipMemcpyAsync(d_wl, c.wl + W * off, 4 * W * m, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = launch(d_wi, d_wo, d_u, c.wl ? d_wl : nullptr, m, d_val, d_pdf, d_wo2, d_pdf2, d_w);
        if (e == hipSuccess && has_eval) e = hipMemcpyAsync(c.out_values + W * off, d_val, 4 * W * m, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && has_pdf) e = hipMemcpyAsync(c.out_pdf + off, d_pdf, 4 * m, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && has_sample) {
            e = hipMemcpyAsync(c.out_wo + 3 * off, d_wo2, 12 * m, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(c.out_pdf2 + off, d_pdf2, 4 * m, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(c.out_weight + W * off, d_w, 4 * W * m, hipMemcpyDeviceToHost, ctx->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(d);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(ctx, MRL_ERR_HIP, std::string("spectral call: ") + hipGetErrorString(e)); }
    return MRL_OK;
}

} // namespace

extern "C" {

int mrl_eval_spectral_batch(mrl_ctx *ctx, const float *wi, const float *wo, const float *wavelengths, int n_wavelengths, int32_t id, size_t n, float *out_values)
{
    return run_spectral(ctx, { 0, wi, wo, nullptr, wavelengths, n_wavelengths, id, n, out_values, nullptr, nullptr, nullptr, nullptr });
}
int mrl_eval_pdf_spectral_batch(mrl_ctx *ctx, const float *wi, const float *wo, const float *wavelengths, int n_wavelengths, int32_t id, size_t n,
                                float *out_values, float *out_pdf)
{
    return run_spectral(ctx, { 4, wi, wo, nullptr, wavelengths, n_wavelengths, id, n, out_values, out_pdf, nullptr, nullptr, nullptr });
}
int mrl_sample_spectral_batch(mrl_ctx *ctx, const float *wi, const float *u, const float *wavelengths, int n_wavelengths, int32_t id, size_t n,
                              float *out_wo, float *out_pdf, float *out_weight)
{
    return run_spectral(ctx, { 2, wi, nullptr, u, wavelengths, n_wavelengths, id, n, nullptr, nullptr, out_wo, out_pdf, out_weight });
}
int mrl_eval_sample_spectral_batch(mrl_ctx *ctx, const float *wi, const float *wo, const float *u, const float *wavelengths, int n_wavelengths, int32_t id,
                                   size_t n, float *out_values, float *out_pdf, float *out_wo, float *out_pdf2, float *out_weight)
{
    return run_spectral(ctx, { 3, wi, wo, u, wavelengths, n_wavelengths, id, n, out_values, out_pdf, out_wo, out_pdf2, out_weight });
}

int mrl_material_wavelengths(mrl_ctx *ctx, int id, int *n_wavelengths, float *out, size_t max_floats)
{
    if (!ctx) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (!n_wavelengths) return fail(ctx, MRL_ERR_INVALID, "null argument");
    if (id < 0 || (size_t)id >= ctx->materials.size() || ctx->materials[(size_t)id].released) return fail(ctx, MRL_ERR_MATERIAL, "unknown material id");
    const MaterialHost &mh = ctx->materials[(size_t)id];
    if (mh.dev.kind != mrl::KIND_RGL_SPECTRAL) return fail(ctx, MRL_ERR_MATERIAL, "not a spectral material");
    *n_wavelengths = mh.rgl.n_wl;
    if (out) {
        if (max_floats < (size_t)mh.rgl.n_wl) return fail(ctx, MRL_ERR_INVALID, "output array too small");
        MRL_HIP(ctx, hipSetDevice(ctx->device));
        MRL_HIP(ctx, hipMemcpy(out, mh.rgl.wavelengths, (size_t)mh.rgl.n_wl * sizeof(float), hipMemcpyDeviceToHost));
    }
    return MRL_OK;
}

} // extern "C"
