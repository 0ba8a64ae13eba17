// merl_image_cache.hip — mrl_material_save_image / _load_image (include/merl_hip.h; SURVEY.md §8f item 4, second half)
#include "merl_ctx.hpp"

using namespace mrlabi;

// ---- on-disk cache of a material's device image (second half) ----------------------------------------
// What is resident for a material — the texel image in its device layout, the sampling marginal, the conditional sampling rows; for an
// RGL material the cell-brick image with its running integrals — written as it is, so that a later process makes the material
// resident with one read and one copy: no parse, no re-layout kernel, no quadrature / prefix-scan kernels, no host normalisation.
// A file is untrusted input: every size is recomputed from the header's shapes (never taken from the file), an RGL descriptor is
// rebuilt from the shapes, and the payload carries a checksum.  What the payload's VALUES say is data (a table), not structure.
namespace {

using mrl::ImageHeader;
using mrl::kImageMagic;
using mrl::image_checksum;
using mrl::rgl_shapes_of;

// device bytes of a table material's texel image, from its descriptor
size_t texel_image_bytes(const mrl::MaterialDev &d)
{
    const size_t plane = (size_t)d.n_th * d.n_td * d.n_pd;
    if (d.kind == mrl::KIND_TABLE_NCH) return plane * mrl::nch_brick_float4s(d.n_ch) * sizeof(float4);
    return (d.layout == mrl::LAYOUT_BRICK ? plane * 8 : (size_t)(d.n_th + 1) * (d.n_td + 1) * (d.n_pd + 1)) * sizeof(float4);
}

size_t rows_image_bytes(const mrl::MaterialDev &d) { return (size_t)(d.n_th + 1) * (d.n_td + 1) * (d.n_pd + 1) * sizeof(float4); }

} // namespace

extern "C" {

int mrl_material_save_image(mrl_ctx *ctx, int id, const char *path)
{
    if (!ctx) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (!path) return fail(ctx, MRL_ERR_INVALID, "null path");
    if (id < 0 || (size_t)id >= ctx->materials.size() || ctx->materials[(size_t)id].released) return fail(ctx, MRL_ERR_MATERIAL, "unknown material id");
    const MaterialHost &mh = ctx->materials[(size_t)id];
    const mrl::MaterialDev &d = mh.dev;
    if (d.kind == mrl::KIND_GGX) return fail(ctx, MRL_ERR_MATERIAL, "an analytic material has no image to cache");
    MRL_HIP(ctx, hipSetDevice(ctx->device));
    ImageHeader h;
    std::memset(&h, 0, sizeof h);
    std::memcpy(h.magic, kImageMagic, 8);
    h.header_bytes = (uint32_t)sizeof h; h.kind = (uint32_t)d.kind; h.layout = (uint32_t)d.layout; h.n_ch = (uint32_t)d.n_ch; h.param = (uint32_t)d.param;
    // (the conditional rows are stamped with the options they were integrated under — at upload —, not with today's)
    h.lookup = (uint32_t)mh.rows_lookup; h.node = (uint32_t)mh.rows_node; h.n_ti = (uint32_t)d.n_ti;
    const bool rgl_kind = d.kind == mrl::KIND_RGL || d.kind == mrl::KIND_RGL_SPECTRAL;
    h.negative = rgl_kind ? 0u : (uint32_t)ctx->opts.negative;
    h.dims[0] = d.n_th; h.dims[1] = d.n_td; h.dims[2] = d.n_pd;
    if (rgl_kind) {
        const mrl::RglDev &r = mh.rgl;
        const int32_t shape[8] = { r.n_phi, r.n_theta, r.nx, r.ny, r.ndf_nx, r.ndf_ny, r.sigma_nx, r.sigma_ny };
        std::memcpy(h.rgl_shape, shape, sizeof shape);
        h.rgl_flags[0] = r.jacobian; h.rgl_flags[1] = r.n_wl;
        mrl::RglLayout l;
        h.texel_bytes = mrl::rgl_plan_layout(rgl_shapes_of(shape, r.jacobian, r.n_wl), l) * sizeof(float);
    } else {
        // RGB tables travel in the compact rows form whatever the context's layout (a brick image is 7.8 x larger than the rows image
        // and reads slower than the source file parses); n-channel tables have one layout
        const bool rgb = d.kind != mrl::KIND_TABLE_NCH;
        if (rgb) h.layout = (uint32_t)mrl::LAYOUT_ROWS;
        h.texel_bytes = rgb ? rows_image_bytes(d) : texel_image_bytes(d);
        h.sampling_doubles = 3 * (uint64_t)d.n_th + 2;
        h.sampling2d_doubles = mh.d_sampling2d ? (uint64_t)d.n_ti * (2 * (uint64_t)d.n_th + 1) : 0;
    }
    std::vector<char> payload;
    try { payload.resize((size_t)h.texel_bytes + (size_t)(h.sampling_doubles + h.sampling2d_doubles) * sizeof(double)); }
    catch (const std::bad_alloc &) { return fail(ctx, MRL_ERR_OOM, "image buffer"); }
    MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!rgl_kind && d.kind != mrl::KIND_TABLE_NCH && d.layout == mrl::LAYOUT_BRICK) {
        float4 *d_rows = nullptr;
        MRL_ALLOC(ctx, hipMalloc((void **)&d_rows, (size_t)h.texel_bytes));
        hipError_t e = mrl::launch_bricks_to_rows(mh.d_texels, h.dims, d.param, d_rows, ctx->compute_units, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = hipMemcpy(payload.data(), d_rows, (size_t)h.texel_bytes, hipMemcpyDeviceToHost);
        (void)hipFree(d_rows);
        MRL_HIP(ctx, e);
    } else {
        MRL_HIP(ctx, hipMemcpy(payload.data(), mh.d_texels, (size_t)h.texel_bytes, hipMemcpyDeviceToHost));
    }
    if (h.sampling_doubles) MRL_HIP(ctx, hipMemcpy(payload.data() + h.texel_bytes, mh.d_sampling, (size_t)h.sampling_doubles * sizeof(double), hipMemcpyDeviceToHost));
    if (h.sampling2d_doubles)
        MRL_HIP(ctx, hipMemcpy(payload.data() + h.texel_bytes + h.sampling_doubles * sizeof(double), mh.d_sampling2d, (size_t)h.sampling2d_doubles * sizeof(double), hipMemcpyDeviceToHost));
    h.checksum = image_checksum(payload.data(), payload.size(), mrl::kImageChecksumSeed);
    // written under a private name and renamed into place: a reader never sees half a file
    const std::string tmp = std::string(path) + ".tmp." + std::to_string((long long)::getpid());
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return fail(ctx, MRL_ERR_IO, std::string("cannot create ") + tmp);
    const bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 && (payload.empty() || std::fwrite(payload.data(), 1, payload.size(), f) == payload.size());
    const bool closed = std::fclose(f) == 0;
    if (!ok || !closed || std::rename(tmp.c_str(), path) != 0) { (void)std::remove(tmp.c_str()); return fail(ctx, MRL_ERR_IO, std::string("cannot write ") + path); }
    return MRL_OK;
}

int mrl_material_load_image(mrl_ctx *ctx, const char *path, int *out_id)
{
    if (!ctx) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (!path || !out_id) return fail(ctx, MRL_ERR_INVALID, "null argument");
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(ctx, MRL_ERR_IO, std::string("cannot open ") + path);
    ImageHeader h;
    auto refuse = [&](const std::string &why) { std::fclose(f); return fail(ctx, MRL_ERR_FORMAT, why + " (" + path + ")"); };
    if (std::fread(&h, sizeof h, 1, f) != 1) return refuse("not a material image of this library version");
    if (std::fseek(f, 0, SEEK_END) != 0) return refuse("seek failed");
    const long long file_bytes = (long long)std::ftell(f);
    // everything the header implies, computed from its shapes (merl_image_file.hpp: the part that is fuzzed on the CPU)
    mrl::ImagePlan plan;
    if (const char *why = mrl::image_plan(h, (unsigned long long)file_bytes, ctx->opts.lookup, ctx->opts.node, ctx->opts.negative, plan)) return refuse(why);
    if (std::fseek(f, (long)sizeof h, SEEK_SET) != 0) return refuse("seek failed");
    const bool is_rgl = plan.is_rgl, is_nch = plan.is_nch;
    const uint64_t texel_bytes = plan.texel_bytes, sampling_doubles = plan.sampling_doubles, sampling2d_doubles = plan.sampling2d_doubles;
    const size_t payload_bytes = plan.payload_bytes;
    mrl::RglFields shapes = plan.shapes;
    const mrl::RglLayout layout = plan.layout;
    mrl::MaterialDev d;
    std::memset(&d, 0, sizeof d);
    d.kind = (int)h.kind;
    d.n_th = plan.dims[0]; d.n_td = plan.dims[1]; d.n_pd = plan.dims[2];
    d.n_ch = plan.n_ch; d.param = plan.param;
    if (!is_rgl) {
        d.layout = is_nch ? mrl::LAYOUT_BRICK : ctx->table_layout;         // an RGB table becomes what this context holds
        d.row_td = d.n_pd + 1; d.row_th = (d.n_td + 1) * (d.n_pd + 1);
    }
    std::vector<char> payload;
    try { payload.resize(payload_bytes); } catch (const std::bad_alloc &) { std::fclose(f); return fail(ctx, MRL_ERR_OOM, "image buffer"); }
    if (payload_bytes && std::fread(payload.data(), 1, payload_bytes, f) != payload_bytes) return refuse("short read");
    std::fclose(f);
    if (image_checksum(payload.data(), payload.size(), mrl::kImageChecksumSeed) != h.checksum) return fail(ctx, MRL_ERR_FORMAT, std::string("checksum mismatch (") + path + ")");
    // the checksum is unkeyed: what a foreign writer could have put there is checked for being evaluable
    if (const char *why = mrl::image_content_check(plan, payload.data())) return fail(ctx, MRL_ERR_FORMAT, std::string(why) + " (" + path + ")");
    MRL_HIP(ctx, hipSetDevice(ctx->device));
    MaterialHost m;
    const bool expand = !is_rgl && !is_nch && d.layout == mrl::LAYOUT_BRICK;      // rows on disk, bricks on this context
    const size_t image_bytes = is_rgl ? ((size_t)texel_bytes + 255) / 256 * 256 : (expand ? texel_image_bytes(d) : (size_t)texel_bytes);
    m.bytes = image_bytes + (is_rgl ? sizeof(mrl::RglDev) : 0) + (size_t)(sampling_doubles + sampling2d_doubles) * sizeof(double);
    int rc = budget_check(ctx, m.bytes + (expand ? (size_t)texel_bytes : 0));
    if (rc != MRL_OK) return rc;
    hipError_t e;
    if (is_rgl || is_nch) e = hipMalloc((void **)&m.d_texels, image_bytes + (is_rgl ? sizeof(mrl::RglDev) : 0));
    else e = table_alloc(ctx, image_bytes, &m.d_texels, &m.in_arena);
    bool oom = e == hipErrorOutOfMemory;
    if (e == hipSuccess && expand) {
        float4 *d_rows = nullptr;
        e = hipMalloc((void **)&d_rows, (size_t)texel_bytes);
        oom = oom || e == hipErrorOutOfMemory;
        if (e == hipSuccess) e = hipMemcpyAsync(d_rows, payload.data(), (size_t)texel_bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = mrl::launch_rows_to_bricks(d_rows, h.dims, m.d_texels, ctx->compute_units, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (d_rows) (void)hipFree(d_rows);
    } else if (e == hipSuccess) {
        e = hipMemcpy(m.d_texels, payload.data(), (size_t)texel_bytes, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && is_rgl) {
        // isotropy and the stored part of the azimuth follow from the phi_i grid, which is the image's first n_phi floats
        shapes.phi_i = (const float *)payload.data();
        m.rgl = mrl::rgl_descriptor(shapes, layout, (const float *)m.d_texels);
        if (m.rgl.reduction != 1 && m.rgl.reduction != 2 && m.rgl.reduction != 4) e = hipErrorInvalidValue;
        else e = hipMemcpy((char *)m.d_texels + image_bytes, &m.rgl, sizeof m.rgl, hipMemcpyHostToDevice);
        d.rgl = (const char *)m.d_texels + image_bytes;
    }
    if (e == hipSuccess && sampling_doubles) {
        e = hipMalloc((void **)&m.d_sampling, (size_t)sampling_doubles * sizeof(double));
        oom = oom || e == hipErrorOutOfMemory;
        if (e == hipSuccess) e = hipMemcpy(m.d_sampling, payload.data() + texel_bytes, (size_t)sampling_doubles * sizeof(double), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && sampling2d_doubles) {
        e = hipMalloc((void **)&m.d_sampling2d, (size_t)sampling2d_doubles * sizeof(double));
        oom = oom || e == hipErrorOutOfMemory;
        if (e == hipSuccess) e = hipMemcpy(m.d_sampling2d, payload.data() + texel_bytes + sampling_doubles * sizeof(double), (size_t)sampling2d_doubles * sizeof(double), hipMemcpyHostToDevice);
    }
    auto drop = [&]() {
        if (is_rgl || is_nch) { if (m.d_texels) (void)hipFree(m.d_texels); } else table_free(ctx, m.d_texels, m.in_arena);
        if (m.d_sampling) (void)hipFree(m.d_sampling);
        if (m.d_sampling2d) (void)hipFree(m.d_sampling2d);
    };
    if (e != hipSuccess) {
        (void)hipGetLastError();
        drop();
        return fail(ctx, oom ? MRL_ERR_OOM : (e == hipErrorInvalidValue ? MRL_ERR_FORMAT : MRL_ERR_HIP), std::string("image upload: ") + hipGetErrorString(e));
    }
    d.texels = m.d_texels; d.sampling = m.d_sampling; d.sampling2d = m.d_sampling2d; d.n_ti = sampling2d_doubles ? mrl::kSamplingIncidentBins : 0;
    m.dev = d;
    m.rows_lookup = (int)h.lookup; m.rows_node = (int)h.node;
    rc = place_material(ctx, m, out_id);
    if (rc != MRL_OK) { drop(); return rc; }
    return MRL_OK;
}

} // extern "C"
