// merl_materials.hip — material constructors and what else is per material: MERL / customized_measurement / n-channel tables
// (parse, upload, re-layout kernel, sampling marginals), GGX, RGL, release, host images, descriptions.  (include/merl_hip.h)
#include "merl_ctx.hpp"

namespace mrlabi {


// Row marginal for table importance sampling (definition: oracle/merl_oracle.h):
// s[n+1] = sin^2(theta_i), cdf[n+1], c[n]; computed on the host in f64, in the file's loop order.
std::vector<double> build_sampling(const double *planar, int n_th, int n_td, int n_pd, const double scale[3], int param)
{
    const size_t plane = (size_t)n_th * n_td * n_pd;
    std::vector<double> D((size_t)n_th), out(3 * (size_t)n_th + 2);
    double *s = out.data(), *cdf = s + (n_th + 1), *c = cdf + (n_th + 1);
    double mean = 0.0;
    for (int i = 0; i < n_th; ++i) {
        double acc = 0.0;
        const double *row = planar + (size_t)i * n_td * n_pd;
        for (size_t k = 0; k < (size_t)n_td * n_pd; ++k) {
            const double r = std::max(row[k] * scale[0], 0.0), g = std::max(row[k + plane] * scale[1], 0.0), b = std::max(row[k + 2 * plane] * scale[2], 0.0);
            acc += 0.2126 * r + 0.7152 * g + 0.0722 * b;
        }
        D[(size_t)i] = acc / ((double)n_td * (double)n_pd);
        mean += D[(size_t)i];
    }
    mean /= (double)n_th;
    if (param != mrl::PARAM_HALF_DIFF) mean = 0.0;                    // the rows are not theta_h: flat lobe (oracle/merl_oracle.h)
    for (int i = 0; i < n_th; ++i) D[(size_t)i] = mean > 0.0 ? D[(size_t)i] + 0.01 * mean : 1.0;
    const double kHalfPi = 3.14159265358979323846 / 2.0;
    for (int i = 0; i <= n_th; ++i) {
        const double r = (double)i / (double)n_th, sn = std::sin(r * r * kHalfPi);
        s[i] = i == n_th ? 1.0 : sn * sn;
    }
    double Z = 0.0;
    for (int i = 0; i < n_th; ++i) Z += D[(size_t)i] * (s[i + 1] - s[i]);
    double run = 0.0;
    for (int i = 0; i < n_th; ++i) {
        cdf[i] = run / Z;
        run += D[(size_t)i] * (s[i + 1] - s[i]);
        c[i] = D[(size_t)i] / (3.14159265358979323846 * Z);
    }
    cdf[n_th] = 1.0;
    return out;
}



// planar f64 (file layout) -> padded, texel-interleaved RGBA f32 in HBM.
// Row layout [n_th+1][n_td+1][n_pd+1]: the extra theta rows repeat the last row (clamp), the
// extra phi texel repeats texel 0 (phi_d is periodic with period pi), so the kernel's "+1"
// neighbours never need a clamp or a wrap.  Scales applied and negatives clamped here, once.
int upload_table(mrl_ctx *ctx, const double *planar, const int dims[3], const double scale[3], int kind, int *out_id)
{
    if (!ctx) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (!planar || !dims || !scale || !out_id) return fail(ctx, MRL_ERR_INVALID, "null argument");
    const int n_th = dims[0], n_td = dims[1], n_pd = dims[2];
    if (n_th < 1 || n_td < 1 || n_pd < 1 || (long long)n_th * n_td * n_pd > (1LL << 28))
        return fail(ctx, MRL_ERR_INVALID, "table dims out of range");
    MRL_HIP(ctx, hipSetDevice(ctx->device));
    const size_t H = n_th + 1, D = n_td + 1, P = n_pd + 1;
    const size_t plane = (size_t)n_th * n_td * n_pd;
    const int layout = ctx->table_layout;
    const int param = kind == mrl::KIND_MERL ? mrl::PARAM_HALF_DIFF : ctx->table_param;      // a MERL file is what it is
    const size_t out_texels = layout == mrl::LAYOUT_BRICK ? plane * 8 : H * D * P;
    const size_t sampling_doubles = 3 * (size_t)n_th + 2;
    MaterialHost m;
    m.bytes = out_texels * sizeof(float4) + sampling_doubles * sizeof(double);
    // budget first: the resident image plus the transient planar copy the re-layout kernel reads
    // (a table that will land in the already-allocated arena needs no free device memory of its own)
    const size_t table_bytes = out_texels * sizeof(float4);
    int rc = budget_check(ctx, m.bytes + 3 * plane * sizeof(double), arena_has_room(ctx, table_bytes) ? table_bytes : 0);
    if (rc != MRL_OK) return rc;
    // the file payload goes to the device as it is; a kernel scales, clamps and re-lays it out
    double *d_planar = nullptr;
    MRL_ALLOC(ctx, hipMalloc((void **)&d_planar, 3 * plane * sizeof(double)));
    hipError_t e = table_alloc(ctx, out_texels * sizeof(float4), &m.d_texels, &m.in_arena);
    const bool oom = e == hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMemcpyAsync(d_planar, planar, 3 * plane * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = mrl::launch_build_table(d_planar, dims, scale, layout, param, ctx->opts.negative == mrl::NEGATIVE_CLAMP, m.d_texels, ctx->compute_units, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_planar);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        table_free(ctx, m.d_texels, m.in_arena);
        return fail(ctx, oom ? MRL_ERR_OOM : MRL_ERR_HIP, std::string("table upload: ") + hipGetErrorString(e));
    }
    {
        const std::vector<double> sampling = build_sampling(planar, n_th, n_td, n_pd, scale, param);
        e = hipMalloc((void **)&m.d_sampling, sampling.size() * sizeof(double));
        const bool oom2 = e == hipErrorOutOfMemory;
        if (e == hipSuccess) e = hipMemcpy(m.d_sampling, sampling.data(), sampling.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            table_free(ctx, m.d_texels, m.in_arena);
            if (m.d_sampling) (void)hipFree(m.d_sampling);
            return fail(ctx, oom2 ? MRL_ERR_OOM : MRL_ERR_HIP, std::string("sampling table upload: ") + hipGetErrorString(e));
        }
    }
    std::memset(&m.dev, 0, sizeof m.dev);
    m.dev.kind = kind;
    m.dev.sampling = m.d_sampling;
    m.dev.n_th = n_th; m.dev.n_td = n_td; m.dev.n_pd = n_pd;
    m.dev.row_td = (int)P;
    m.dev.row_th = (int)(D * P);
    m.dev.texels = m.d_texels;
    m.dev.layout = layout;
    m.dev.n_ch = 3;
    m.dev.param = param;
    {
        // the conditional sampling table, from the table that has just become resident (quadrature + prefix scan on the device)
        const int n_ti = mrl::kSamplingIncidentBins;
        double *d_work = nullptr;
        e = hipMalloc((void **)&m.d_sampling2d, (size_t)n_ti * (2 * (size_t)n_th + 1) * sizeof(double));
        const bool oom3 = e == hipErrorOutOfMemory;
        if (e == hipSuccess) e = hipMalloc((void **)&d_work, (size_t)n_ti * (size_t)n_th * sizeof(double));
        if (e == hipSuccess) e = mrl::launch_build_sampling2d(m.dev, ctx->opts, n_ti, m.d_sampling2d, d_work, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (d_work) (void)hipFree(d_work);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            table_free(ctx, m.d_texels, m.in_arena); (void)hipFree(m.d_sampling);
            if (m.d_sampling2d) (void)hipFree(m.d_sampling2d);
            return fail(ctx, oom3 ? MRL_ERR_OOM : MRL_ERR_HIP, std::string("conditional sampling table: ") + hipGetErrorString(e));
        }
        m.dev.sampling2d = m.d_sampling2d;
        m.dev.n_ti = n_ti;
        m.rows_lookup = ctx->opts.lookup; m.rows_node = ctx->opts.node;
        m.bytes += (size_t)n_ti * (2 * (size_t)n_th + 1) * sizeof(double);
    }
    rc = place_material(ctx, m, out_id);
    if (rc != MRL_OK) { table_free(ctx, m.d_texels, m.in_arena); (void)hipFree(m.d_sampling); (void)hipFree(m.d_sampling2d); return rc; }
    return MRL_OK;
}

// a1: MERL .binary reader: int32 dims[3], then 3*n planar doubles.  A customized_measurement
// table may carry its payload as f32 instead (the file length says which); MERL files are f64 only.
int read_table_file(mrl_ctx *ctx, const char *path, bool require_merl, std::vector<double> &data, int dims[3])
{
    if (!path) return fail(ctx, MRL_ERR_INVALID, "null path");
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(ctx, MRL_ERR_IO, std::string("cannot open ") + path);
    int32_t d[3];
    if (std::fread(d, sizeof(int32_t), 3, f) != 3) { std::fclose(f); return fail(ctx, MRL_ERR_FORMAT, "short header"); }
    if (d[0] <= 0 || d[1] <= 0 || d[2] <= 0) { std::fclose(f); return fail(ctx, MRL_ERR_FORMAT, "non-positive dims"); }
    long long n = (long long)d[0] * d[1] * d[2];
    if (require_merl && n != 90LL * 90 * 180) { std::fclose(f); return fail(ctx, MRL_ERR_FORMAT, "dims do not match the MERL grid (90*90*360/2)"); }
    if (n > (1LL << 28)) { std::fclose(f); return fail(ctx, MRL_ERR_FORMAT, "table too large"); }
    bool f32_payload = false;
    if (!require_merl && std::fseek(f, 0, SEEK_END) == 0) {
        const long long bytes = (long long)std::ftell(f);
        f32_payload = bytes == 12 + 3 * n * 4;
        if (std::fseek(f, 12, SEEK_SET) != 0) { std::fclose(f); return fail(ctx, MRL_ERR_IO, "seek failed"); }
    }
    try { data.resize(3 * (size_t)n); } catch (const std::bad_alloc &) { std::fclose(f); return fail(ctx, MRL_ERR_OOM, "table buffer"); }
    size_t got;
    if (f32_payload) {
        std::vector<float> narrow;
        try { narrow.resize(data.size()); } catch (const std::bad_alloc &) { std::fclose(f); return fail(ctx, MRL_ERR_OOM, "table buffer"); }
        got = std::fread(narrow.data(), sizeof(float), narrow.size(), f);
        for (size_t i = 0; i < got; ++i) data[i] = (double)narrow[i];
    } else {
        got = std::fread(data.data(), sizeof(double), data.size(), f);
    }
    std::fclose(f);
    if (got != data.size()) return fail(ctx, MRL_ERR_FORMAT, "truncated table payload");
    if (require_merl) { dims[0] = kMerlDims[0]; dims[1] = kMerlDims[1]; dims[2] = kMerlDims[2]; }
    else { dims[0] = d[0]; dims[1] = d[1]; dims[2] = d[2]; }
    return MRL_OK;
}

// staging area of host-pointer calls: `units` units of `unit_bytes` each (80 B for RGB, 56 + 8 C for C channels)


// n-channel row marginal for table importance sampling: as build_sampling, with the plain mean over the channels in
// place of the RGB luminance (oracle: orc_build_sampling_nch)
std::vector<double> build_sampling_nch(const double *planar, int n_th, int n_td, int n_pd, int n_ch, const double *scale, int param)
{
    const size_t plane = (size_t)n_th * n_td * n_pd;
    std::vector<double> D((size_t)n_th), out(3 * (size_t)n_th + 2);
    double *s = out.data(), *cdf = s + (n_th + 1), *c = cdf + (n_th + 1);
    double mean = 0.0;
    for (int i = 0; i < n_th; ++i) {
        double acc = 0.0;
        const double *row = planar + (size_t)i * n_td * n_pd;
        for (size_t k = 0; k < (size_t)n_td * n_pd; ++k) {
            double sum = 0.0;
            for (int ch = 0; ch < n_ch; ++ch) sum += std::max(row[k + (size_t)ch * plane] * scale[ch], 0.0);
            acc += sum / (double)n_ch;
        }
        D[(size_t)i] = acc / ((double)n_td * (double)n_pd);
        mean += D[(size_t)i];
    }
    mean /= (double)n_th;
    if (param != mrl::PARAM_HALF_DIFF) mean = 0.0;                    // the rows are not theta_h: flat lobe (oracle/merl_oracle.h)
    for (int i = 0; i < n_th; ++i) D[(size_t)i] = mean > 0.0 ? D[(size_t)i] + 0.01 * mean : 1.0;
    const double kHalfPi = 3.14159265358979323846 / 2.0;
    for (int i = 0; i <= n_th; ++i) {
        const double r = (double)i / (double)n_th, sn = std::sin(r * r * kHalfPi);
        s[i] = i == n_th ? 1.0 : sn * sn;
    }
    double Z = 0.0;
    for (int i = 0; i < n_th; ++i) Z += D[(size_t)i] * (s[i + 1] - s[i]);
    double run = 0.0;
    for (int i = 0; i < n_th; ++i) {
        cdf[i] = run / Z;
        run += D[(size_t)i] * (s[i + 1] - s[i]);
        c[i] = D[(size_t)i] / (3.14159265358979323846 * Z);
    }
    cdf[n_th] = 1.0;
    return out;
}

// planar f64, n_ch planes -> n-channel bricks in HBM (merl_nch.hip).  n_ch == 3 is the RGB path (packed 96-B bricks).
int upload_table_nch(mrl_ctx *ctx, const double *planar, const int dims[3], int n_ch, const double *scale, int *out_id)
{
    if (!ctx) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (!planar || !dims || !out_id) return fail(ctx, MRL_ERR_INVALID, "null argument");
    if (n_ch < 1 || n_ch > mrl::kMaxChannels) return fail(ctx, MRL_ERR_INVALID, "channel count must be 1.." + std::to_string(mrl::kMaxChannels));
    std::vector<double> ones((size_t)n_ch, 1.0);
    if (!scale) scale = ones.data();
    if (n_ch == 3) return upload_table(ctx, planar, dims, scale, mrl::KIND_TABLE, out_id);
    const int n_th = dims[0], n_td = dims[1], n_pd = dims[2];
    if (n_th < 1 || n_td < 1 || n_pd < 1 || (long long)n_th * n_td * n_pd > (1LL << 28))
        return fail(ctx, MRL_ERR_INVALID, "table dims out of range");
    MRL_HIP(ctx, hipSetDevice(ctx->device));
    const size_t plane = (size_t)n_th * n_td * n_pd;
    const size_t out_f4 = plane * mrl::nch_brick_float4s(n_ch);
    const size_t sampling_doubles = 3 * (size_t)n_th + 2;
    MaterialHost m;
    m.bytes = out_f4 * sizeof(float4) + sampling_doubles * sizeof(double);
    const size_t planar_bytes = ((size_t)n_ch * plane + (size_t)n_ch) * sizeof(double);       // payload + the channel scales
    int rc = budget_check(ctx, m.bytes + planar_bytes);
    if (rc != MRL_OK) return rc;
    double *d_planar = nullptr;
    MRL_ALLOC(ctx, hipMalloc((void **)&d_planar, planar_bytes));
    double *d_scale = d_planar + (size_t)n_ch * plane;
    hipError_t e = hipMalloc((void **)&m.d_texels, out_f4 * sizeof(float4));
    const bool oom = e == hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMemcpyAsync(d_planar, planar, (size_t)n_ch * plane * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_scale, scale, (size_t)n_ch * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = mrl::launch_build_table_nch(d_planar, d_scale, dims, n_ch, ctx->table_param, ctx->opts.negative == mrl::NEGATIVE_CLAMP, m.d_texels, ctx->compute_units, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_planar);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (m.d_texels) (void)hipFree(m.d_texels);
        return fail(ctx, oom ? MRL_ERR_OOM : MRL_ERR_HIP, std::string("n-channel table upload: ") + hipGetErrorString(e));
    }
    const std::vector<double> sampling = build_sampling_nch(planar, n_th, n_td, n_pd, n_ch, scale, ctx->table_param);
    e = hipMalloc((void **)&m.d_sampling, sampling.size() * sizeof(double));
    const bool oom2 = e == hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMemcpy(m.d_sampling, sampling.data(), sampling.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(m.d_texels);
        if (m.d_sampling) (void)hipFree(m.d_sampling);
        return fail(ctx, oom2 ? MRL_ERR_OOM : MRL_ERR_HIP, std::string("sampling table upload: ") + hipGetErrorString(e));
    }
    std::memset(&m.dev, 0, sizeof m.dev);
    m.dev.kind = mrl::KIND_TABLE_NCH;
    m.dev.sampling = m.d_sampling;
    m.dev.n_th = n_th; m.dev.n_td = n_td; m.dev.n_pd = n_pd;
    m.dev.texels = m.d_texels;
    m.dev.layout = mrl::LAYOUT_BRICK;
    m.dev.n_ch = n_ch;
    m.dev.param = ctx->table_param;
    rc = place_material(ctx, m, out_id);
    if (rc != MRL_OK) { (void)hipFree(m.d_texels); (void)hipFree(m.d_sampling); return rc; }
    return MRL_OK;
}

// customized_measurement file with n_ch planes: int32 dims[3], then planar values as f64 or f32 (told apart by the file length)
int read_table_file_nch(mrl_ctx *ctx, const char *path, int n_ch, std::vector<double> &data, int dims[3])
{
    if (!path) return fail(ctx, MRL_ERR_INVALID, "null path");
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(ctx, MRL_ERR_IO, std::string("cannot open ") + path);
    int32_t d[3];
    if (std::fread(d, sizeof(int32_t), 3, f) != 3) { std::fclose(f); return fail(ctx, MRL_ERR_FORMAT, "short header"); }
    if (d[0] <= 0 || d[1] <= 0 || d[2] <= 0) { std::fclose(f); return fail(ctx, MRL_ERR_FORMAT, "non-positive dims"); }
    const long long n = (long long)d[0] * d[1] * d[2];
    if (n > (1LL << 28)) { std::fclose(f); return fail(ctx, MRL_ERR_FORMAT, "table too large"); }
    if (std::fseek(f, 0, SEEK_END) != 0) { std::fclose(f); return fail(ctx, MRL_ERR_IO, "seek failed"); }
    const long long bytes = (long long)std::ftell(f);
    const bool f64_payload = bytes == 12 + (long long)n_ch * n * 8, f32_payload = bytes == 12 + (long long)n_ch * n * 4;
    if (!f64_payload && !f32_payload) { std::fclose(f); return fail(ctx, MRL_ERR_FORMAT, "file length matches neither an f64 nor an f32 payload of " + std::to_string(n_ch) + " channels"); }
    if (std::fseek(f, 12, SEEK_SET) != 0) { std::fclose(f); return fail(ctx, MRL_ERR_IO, "seek failed"); }
    try { data.resize((size_t)n_ch * (size_t)n); } catch (const std::bad_alloc &) { std::fclose(f); return fail(ctx, MRL_ERR_OOM, "table buffer"); }
    size_t got;
    if (f32_payload) {
        std::vector<float> narrow;
        try { narrow.resize(data.size()); } catch (const std::bad_alloc &) { std::fclose(f); return fail(ctx, MRL_ERR_OOM, "table buffer"); }
        got = std::fread(narrow.data(), sizeof(float), narrow.size(), f);
        for (size_t i = 0; i < got; ++i) data[i] = (double)narrow[i];
    } else {
        got = std::fread(data.data(), sizeof(double), data.size(), f);
    }
    std::fclose(f);
    if (got != data.size()) return fail(ctx, MRL_ERR_FORMAT, "truncated table payload");
    dims[0] = d[0]; dims[1] = d[1]; dims[2] = d[2];
    return MRL_OK;
}

} // namespace mrlabi
using namespace mrlabi;

extern "C" {

int mrl_material_load_merl(mrl_ctx *ctx, const char *path, int *out_id)
{
    if (!ctx || !out_id) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    std::vector<double> data; int dims[3];
    int rc = read_table_file(ctx, path, true, data, dims);
    if (rc != MRL_OK) return rc;
    return upload_table(ctx, data.data(), dims, kMerlScale, mrl::KIND_MERL, out_id);
}

int mrl_material_upload_f64(mrl_ctx *ctx, const double *planar_rgb, int *out_id)
{
    return upload_table(ctx, planar_rgb, kMerlDims, kMerlScale, mrl::KIND_MERL, out_id);
}

int mrl_material_upload_table(mrl_ctx *ctx, const double *planar_rgb, const int dims[3], const double scale[3], int *out_id)
{
    return upload_table(ctx, planar_rgb, dims, scale, mrl::KIND_TABLE, out_id);
}

int mrl_material_load_table(mrl_ctx *ctx, const char *path, const double scale[3], int *out_id)
{
    if (!ctx || !out_id || !scale) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    std::vector<double> data; int dims[3];
    int rc = read_table_file(ctx, path, false, data, dims);
    if (rc != MRL_OK) return rc;
    return upload_table(ctx, data.data(), dims, scale, mrl::KIND_TABLE, out_id);
}

int mrl_material_ggx(mrl_ctx *ctx, float alpha, const float eta[3], const float k[3], int *out_id)
{
    if (!ctx || !eta || !k || !out_id) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (!(alpha > 0.0f)) return fail(ctx, MRL_ERR_INVALID, "alpha must be positive");
    MRL_HIP(ctx, hipSetDevice(ctx->device));
    MaterialHost m;
    std::memset(&m.dev, 0, sizeof m.dev);
    m.dev.kind = mrl::KIND_GGX;
    m.dev.n_ch = 3;
    m.dev.alpha = (double)alpha;
    for (int c = 0; c < 3; ++c) { m.dev.eta[c] = (double)eta[c]; m.dev.k[c] = (double)k[c]; }
    return place_material(ctx, m, out_id);
}

// The adaptive-parameterisation measured BSDF (RGL *.bsdf fields): the host normalises the two distributions and forms
// their running integrals (f64, once), the image goes to HBM as one allocation.  PARITY UNPINNED (merl_rgl.hpp).
static int upload_rgl_fields(mrl_ctx *ctx, const mrl_rgl_fields *f, int n_wl, const float *wavelengths, const float *values, int *out_id)
{
    if (!ctx) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (!f || !out_id) return fail(ctx, MRL_ERR_INVALID, "null argument");
    mrl::RglFields h;
    h.n_phi = f->n_phi; h.n_theta = f->n_theta; h.phi_i = f->phi_i; h.theta_i = f->theta_i;
    for (int k = 0; k < 2; ++k) { h.res_ndf[k] = f->res_ndf[k]; h.res_sigma[k] = f->res_sigma[k]; h.res[k] = f->res[k]; }
    h.ndf = f->ndf; h.sigma = f->sigma; h.vndf = f->vndf; h.luminance = f->luminance; h.rgb = values;
    h.jacobian = f->jacobian;
    h.n_wl = n_wl; h.wavelengths = wavelengths;
    if (const char *why = mrl::rgl_check_fields(h)) return fail(ctx, MRL_ERR_INVALID, std::string("RGL fields: ") + why);
    MRL_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<float> blob;
    mrl::RglLayout layout;
    try { layout = mrl::rgl_build_image(h, blob); } catch (const std::bad_alloc &) { return fail(ctx, MRL_ERR_OOM, "RGL image"); }
    MaterialHost m;
    const size_t image_bytes = (blob.size() * sizeof(float) + 255) / 256 * 256;          // the descriptor sits behind the image
    m.bytes = image_bytes + sizeof(mrl::RglDev);
    int rc = budget_check(ctx, m.bytes);
    if (rc != MRL_OK) return rc;
    hipError_t e = hipMalloc((void **)&m.d_texels, m.bytes);
    const bool oom = e == hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMemcpy(m.d_texels, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        m.rgl = mrl::rgl_descriptor(h, layout, (const float *)m.d_texels);
        e = hipMemcpy((char *)m.d_texels + image_bytes, &m.rgl, sizeof m.rgl, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (m.d_texels) (void)hipFree(m.d_texels);
        return fail(ctx, oom ? MRL_ERR_OOM : MRL_ERR_HIP, std::string("RGL upload: ") + hipGetErrorString(e));
    }
    std::memset(&m.dev, 0, sizeof m.dev);
    m.dev.kind = n_wl > 0 ? mrl::KIND_RGL_SPECTRAL : mrl::KIND_RGL;
    m.dev.rgl = (const char *)m.d_texels + image_bytes;
    m.dev.n_ch = n_wl > 0 ? n_wl : 3;
    m.dev.n_th = h.n_phi; m.dev.n_td = h.n_theta; m.dev.n_pd = h.res[0];     // what mrl_material_info reports
    rc = place_material(ctx, m, out_id);
    if (rc != MRL_OK) { (void)hipFree(m.d_texels); return rc; }
    return MRL_OK;
}

int mrl_material_upload_rgl(mrl_ctx *ctx, const mrl_rgl_fields *f, int *out_id)
{
    return upload_rgl_fields(ctx, f, 0, nullptr, f ? f->rgb : nullptr, out_id);
}

// a spectral file: "spectra" over "wavelengths" instead of "rgb" (merl_rgl_spectral.hip holds the calls)
int mrl_material_upload_rgl_spectral(mrl_ctx *ctx, const mrl_rgl_spectral_fields *f, int *out_id)
{
    if (!ctx) return MRL_ERR_INVALID;
    if (!f || f->n_wavelengths < 1 || !f->wavelengths || !f->spectra) { MRL_GUARD(ctx); return fail(ctx, MRL_ERR_INVALID, "a spectral RGL material needs spectra over at least one wavelength"); }
    return upload_rgl_fields(ctx, &f->base, f->n_wavelengths, f->wavelengths, f->spectra, out_id);
}

int mrl_material_release(mrl_ctx *ctx, int id)
{
    if (!ctx) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (id < 0 || (size_t)id >= ctx->materials.size() || ctx->materials[(size_t)id].released)
        return fail(ctx, MRL_ERR_MATERIAL, "unknown material id");
    MRL_HIP(ctx, hipSetDevice(ctx->device));
    const ScalarPause quiet(ctx);                                // a service instance may be reading the table
    int rc = ensure_dummy(ctx);
    if (rc != MRL_OK) return rc;
    MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));             // launches in flight may still read the table
    MaterialHost &m = ctx->materials[(size_t)id];
    const MaterialHost before = m;
    m.dev = tombstone_dev(ctx);
    m.released = true;
    rc = sync_material_array(ctx);                               // the device array must stop naming the table first
    if (rc != MRL_OK) { m = before; return rc; }
    table_free(ctx, before.d_texels, before.in_arena);
    if (before.d_sampling) (void)hipFree(before.d_sampling);
    if (before.d_sampling2d) (void)hipFree(before.d_sampling2d);
    m.d_texels = nullptr; m.d_sampling = nullptr; m.d_sampling2d = nullptr;
    ctx->material_bytes -= before.bytes;
    m.bytes = 0;
    return MRL_OK;
}

// The host image of a resident RGB table for one-unit calls on the CPU (merl_host_scalar.hip): the device's own Float texel
// values, re-read from HBM into the rows layout — from a rows-layout table as it is, from bricks by taking corner 0 of
// every cell (= the texel itself) and re-creating the padding rows — plus the sampling marginal and a snapshot of the
// context's lookup options.  One D2H copy of the table (24 MB rows / 187 MB bricks for MERL): about 15 ms, once per image.
int mrl_material_host_table(mrl_ctx *ctx, int id, mrl_host_table **out)
{
    if (!ctx) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (!out) return fail(ctx, MRL_ERR_INVALID, "null argument");
    *out = nullptr;
    if (id < 0 || (size_t)id >= ctx->materials.size() || ctx->materials[(size_t)id].released) return fail(ctx, MRL_ERR_MATERIAL, "unknown material id");
    const MaterialHost &mh = ctx->materials[(size_t)id];
    if (mh.dev.kind != mrl::KIND_MERL && mh.dev.kind != mrl::KIND_TABLE && mh.dev.kind != mrl::KIND_RGL && mh.dev.kind != mrl::KIND_RGL_SPECTRAL)
        return fail(ctx, MRL_ERR_MATERIAL, "host images exist for three-channel table materials and RGL materials");
    if (!__builtin_cpu_supports("fma") || !__builtin_cpu_supports("avx2"))
        return fail(ctx, MRL_ERR_INVALID, "the host one-unit path needs a CPU with FMA and AVX2");
    MRL_HIP(ctx, hipSetDevice(ctx->device));
    if (mh.dev.kind == mrl::KIND_RGL || mh.dev.kind == mrl::KIND_RGL_SPECTRAL) {      // the image is position independent: copy it, move the descriptor's pointers
        MRL_HIP(ctx, hipSetDevice(ctx->device));
        MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));            // before anything is allocated: an early return leaks nothing
        mrl_host_table *t = nullptr;
        try {
            t = new mrl_host_table;
            t->rgl_image.resize(mh.bytes / sizeof(float));
        } catch (const std::bad_alloc &) { delete t; return fail(ctx, MRL_ERR_OOM, "host image"); }
        const hipError_t e = hipMemcpy(t->rgl_image.data(), mh.d_texels, mh.bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { (void)hipGetLastError(); delete t; return fail(ctx, MRL_ERR_HIP, std::string("host image: ") + hipGetErrorString(e)); }
        t->rgl = mh.rgl;
        const char *from = (const char *)mh.d_texels, *to = (const char *)t->rgl_image.data();
        t->rgl.rebase(from, to);
        t->m = mh.dev;
        t->opts = ctx->opts;
        *out = t;
        return MRL_OK;
    }
    const int n_th = mh.dev.n_th, n_td = mh.dev.n_td, n_pd = mh.dev.n_pd;
    const size_t H = n_th + 1, D = n_td + 1, P = n_pd + 1, cells = (size_t)n_th * n_td * n_pd;
    MRL_HIP(ctx, hipStreamSynchronize(ctx->stream));                // before anything is allocated: an early return leaks nothing
    mrl_host_table *t = nullptr;
    try {
        t = new mrl_host_table;
        t->rows.resize(H * D * P);
        t->marginal.resize(3 * (size_t)n_th + 2);
        std::vector<float4> bricks;
        hipError_t e = hipSuccess;
        if (mh.dev.layout == mrl::LAYOUT_ROWS) {
            e = hipMemcpy(t->rows.data(), mh.d_texels, t->rows.size() * sizeof(float4), hipMemcpyDeviceToHost);
        } else {
            bricks.resize(cells * 8);
            e = hipMemcpy(bricks.data(), mh.d_texels, bricks.size() * sizeof(float4), hipMemcpyDeviceToHost);
        }
        if (e == hipSuccess) e = hipMemcpy(t->marginal.data(), mh.d_sampling, t->marginal.size() * sizeof(double), hipMemcpyDeviceToHost);
        if (e == hipSuccess && mh.d_sampling2d) {
            t->marginal2d.resize((size_t)mh.dev.n_ti * (2 * (size_t)n_th + 1));
            e = hipMemcpy(t->marginal2d.data(), mh.d_sampling2d, t->marginal2d.size() * sizeof(double), hipMemcpyDeviceToHost);
        }
        if (e != hipSuccess) { (void)hipGetLastError(); delete t; return fail(ctx, MRL_ERR_HIP, std::string("host image: ") + hipGetErrorString(e)); }
        if (mh.dev.layout != mrl::LAYOUT_ROWS) {
            const bool periodic = mrl::param_phi_periodic(mh.dev.param);
            for (size_t ih = 0; ih < H; ++ih)
                for (size_t idd = 0; idd < D; ++idd)
                    for (size_t ip = 0; ip < P; ++ip) {
                        const size_t sh = ih < (size_t)n_th ? ih : n_th - 1, sd = idd < (size_t)n_td ? idd : n_td - 1;
                        const size_t sp = ip == (size_t)n_pd ? (periodic ? 0 : n_pd - 1) : ip;
                        const float4 q = bricks[((sh * n_td + sd) * n_pd + sp) * 8];        // corner 0: x y z = the cell's own texel
                        t->rows[(ih * D + idd) * P + ip] = make_float4(q.x, q.y, q.z, 0.0f);
                    }
        }
    } catch (const std::bad_alloc &) {
        delete t;
        return fail(ctx, MRL_ERR_OOM, "host image");
    }
    t->m = mh.dev;
    t->m.texels = t->rows.data();
    t->m.sampling = t->marginal.data();
    t->m.sampling2d = t->marginal2d.empty() ? nullptr : t->marginal2d.data();
    t->m.layout = mrl::LAYOUT_ROWS;
    t->m.row_td = (int)P;
    t->m.row_th = (int)(D * P);
    t->opts = ctx->opts;
    *out = t;
    return MRL_OK;
}

// the conditional sampling table of an RGB table material, as the device built it: n_ti rows of (n_th + 1 cdf | n_th c)
int mrl_material_sampling2d(mrl_ctx *ctx, int id, int *n_ti, int *n_th, double *out, size_t max_doubles)
{
    if (!ctx) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (id < 0 || (size_t)id >= ctx->materials.size() || ctx->materials[(size_t)id].released) return fail(ctx, MRL_ERR_MATERIAL, "unknown material id");
    const MaterialHost &mh = ctx->materials[(size_t)id];
    if (!mh.d_sampling2d) return fail(ctx, MRL_ERR_MATERIAL, "the material has no conditional sampling table (RGB table materials do)");
    const size_t need = (size_t)mh.dev.n_ti * (2 * (size_t)mh.dev.n_th + 1);
    if (n_ti) *n_ti = mh.dev.n_ti;
    if (n_th) *n_th = mh.dev.n_th;
    if (!out) return MRL_OK;
    if (max_doubles < need) return fail(ctx, MRL_ERR_INVALID, "buffer too small");
    MRL_HIP(ctx, hipSetDevice(ctx->device));
    MRL_HIP(ctx, hipMemcpy(out, mh.d_sampling2d, need * sizeof(double), hipMemcpyDeviceToHost));
    return MRL_OK;
}

int mrl_material_info(const mrl_ctx *ctx, int id, int *kind, int dims[3])
{
    if (!ctx) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (id < 0 || (size_t)id >= ctx->materials.size() || ctx->materials[(size_t)id].released) return MRL_ERR_MATERIAL;
    const mrl::MaterialDev &d = ctx->materials[(size_t)id].dev;
    if (kind) *kind = d.kind;
    if (dims) { dims[0] = d.n_th; dims[1] = d.n_td; dims[2] = d.n_pd; }
    return MRL_OK;
}

/* ---- n-channel tables ---- */
int mrl_material_upload_table_nch(mrl_ctx *ctx, const double *planar, const int dims[3], int n_channels, const double *scale, int *out_id)
{
    return upload_table_nch(ctx, planar, dims, n_channels, scale, out_id);
}

// the same upload with the parameterisation named in the call (the context's MRL_OPT_TABLE_PARAM is left as it was)
int mrl_material_upload_table_param(mrl_ctx *ctx, const double *planar, const int dims[3], int n_channels, const double *scale, int param, int *out_id)
{
    if (!ctx) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);                                    // recursive: the option cannot be seen half-way by another thread's upload
    if (param < mrl::PARAM_HALF_DIFF || param > mrl::PARAM_STANDARD_FULL) return fail(ctx, MRL_ERR_INVALID, "unknown parameterisation");
    const int before = ctx->table_param;
    ctx->table_param = param;
    const int rc = upload_table_nch(ctx, planar, dims, n_channels, scale, out_id);
    ctx->table_param = before;
    return rc;
}

int mrl_material_load_table_nch(mrl_ctx *ctx, const char *path, int n_channels, const double *scale, int *out_id)
{
    if (!ctx || !out_id) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (n_channels < 1 || n_channels > mrl::kMaxChannels) return fail(ctx, MRL_ERR_INVALID, "channel count must be 1.." + std::to_string(mrl::kMaxChannels));
    std::vector<double> data; int dims[3];
    int rc = read_table_file_nch(ctx, path, n_channels, data, dims);
    if (rc != MRL_OK) return rc;
    return upload_table_nch(ctx, data.data(), dims, n_channels, scale, out_id);
}

int mrl_material_channels(const mrl_ctx *ctx, int id, int *n_channels)
{
    if (!ctx || !n_channels) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (id < 0 || (size_t)id >= ctx->materials.size() || ctx->materials[(size_t)id].released) return MRL_ERR_MATERIAL;
    *n_channels = ctx->materials[(size_t)id].dev.n_ch;
    return MRL_OK;
}

int mrl_material_param(const mrl_ctx *ctx, int id, int *param)
{
    if (!ctx || !param) return MRL_ERR_INVALID;
    MRL_GUARD(ctx);
    if (id < 0 || (size_t)id >= ctx->materials.size() || ctx->materials[(size_t)id].released) return MRL_ERR_MATERIAL;
    const mrl::MaterialDev &d = ctx->materials[(size_t)id].dev;
    if (d.kind == mrl::KIND_GGX) return MRL_ERR_MATERIAL;
    *param = d.param;
    return MRL_OK;
}

} // extern "C"
