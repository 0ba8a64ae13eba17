// merl_tensor_file.hip — reader for the "tensor_file" container (host code only; no kernel here).
//
// This is the container the RGL material database ships its measured BSDFs in (*.bsdf) and that upstream
// Mitsuba 3's `measured` plugin reads; SURVEY.md §8f item 3 lists it among the generalised measurement formats.
// Layout, restated from the public format description (the reference ships no file of this kind, no reader and no
// test for one — PARITY UNPINNED; nothing below was checked against a real RGL file in this container):
//     bytes 0..11   "tensor_file\0"
//     u8 major, u8 minor            version 1.0
//     u32 n_fields
//     per field:    u16 name_length, name (no terminator), u16 ndim, u8 dtype, u64 offset (from the file start),
//                   u64 shape[ndim]
//     payloads at their offsets, C order, little endian.
// dtype: 1..8 integers (1-2: 1 byte, 3-4: 2 bytes, 5-6: 4 bytes, 7-8: 8 bytes; writers disagree on which of a pair is
// the signed one, so integers are exposed by width only), 9 f16, 10 f32, 11 f64.
// Two consumers: mrl_material_load_rgl hands the RGL fields (phi_i, theta_i, ndf, sigma, vndf, luminance, rgb, jacobian) to
// the adaptive-parameterisation material (merl_rgl.hip), and mrl_material_load_tensor_table feeds the table hot path
// with a customized_measurement table stored in this container: a float field of shape
// [channels, n_0, n_1, n_2] (default name "table"), optionally with a "scale" field [channels] and a "parameterization"
// field (one integer, enum mrl_param: which three angles the axes are — half/diff by default).
#include "../../include/merl_hip.h"

#include <cstdio>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

struct Field {
    std::string name;
    int dtype = 0;
    std::vector<uint64_t> shape;
    uint64_t offset = 0, count = 0;
};

size_t dtype_size(int dtype)
{
    switch (dtype) {
        case 1: case 2: return 1;
        case 3: case 4: case 9: return 2;
        case 5: case 6: case 10: return 4;
        case 7: case 8: case 11: return 8;
    }
    return 0;
}

} // namespace

struct mrl_tensor_file {
    std::vector<unsigned char> bytes;
    std::vector<Field> fields;
    std::string error;
};

namespace {

thread_local std::string t_open_error;

template <typename T> bool take(const std::vector<unsigned char> &b, size_t &at, T &out)
{
    if (at + sizeof(T) > b.size()) return false;
    std::memcpy(&out, b.data() + at, sizeof(T));
    at += sizeof(T);
    return true;
}

int parse(mrl_tensor_file *f)
{
    const std::vector<unsigned char> &b = f->bytes;
    if (b.size() < 18 || std::memcmp(b.data(), "tensor_file", 12) != 0) { f->error = "not a tensor_file container (bad magic)"; return MRL_ERR_FORMAT; }
    size_t at = 12;
    uint8_t major = 0, minor = 0;
    uint32_t n_fields = 0;
    if (!take(b, at, major) || !take(b, at, minor) || !take(b, at, n_fields)) { f->error = "short header"; return MRL_ERR_FORMAT; }
    if (major != 1 || minor != 0) { f->error = "unsupported tensor_file version " + std::to_string(major) + "." + std::to_string(minor); return MRL_ERR_FORMAT; }
    if (n_fields > 4096) { f->error = "implausible field count"; return MRL_ERR_FORMAT; }
    for (uint32_t i = 0; i < n_fields; ++i) {
        Field fd;
        uint16_t name_len = 0, ndim = 0;
        uint8_t dtype = 0;
        if (!take(b, at, name_len) || at + name_len > b.size()) { f->error = "truncated field table"; return MRL_ERR_FORMAT; }
        fd.name.assign((const char *)b.data() + at, name_len);
        at += name_len;
        if (!take(b, at, ndim) || !take(b, at, dtype) || !take(b, at, fd.offset)) { f->error = "truncated field table"; return MRL_ERR_FORMAT; }
        if (ndim > 16) { f->error = "field \"" + fd.name + "\": implausible rank"; return MRL_ERR_FORMAT; }
        fd.dtype = dtype;
        const size_t width = dtype_size(dtype);
        if (width == 0) { f->error = "field \"" + fd.name + "\": unknown dtype " + std::to_string(dtype); return MRL_ERR_FORMAT; }
        fd.count = 1;
        for (uint16_t d = 0; d < ndim; ++d) {
            uint64_t extent = 0;
            if (!take(b, at, extent)) { f->error = "truncated field table"; return MRL_ERR_FORMAT; }
            if (extent != 0 && fd.count > (uint64_t)b.size() / extent) { f->error = "field \"" + fd.name + "\": shape exceeds the file"; return MRL_ERR_FORMAT; }
            fd.count *= extent;
            fd.shape.push_back(extent);
        }
        if (fd.offset > b.size() || fd.count * width > b.size() - fd.offset) { f->error = "field \"" + fd.name + "\": payload outside the file"; return MRL_ERR_FORMAT; }
        f->fields.push_back(std::move(fd));
    }
    return MRL_OK;
}

double half_to_double(uint16_t h)
{
    const int sign = h >> 15, exp = (h >> 10) & 31, man = h & 1023;
    double v;
    if (exp == 0) v = man * 5.9604644775390625e-08;                      // 2^-24
    else if (exp == 31) v = man ? __builtin_nan("") : __builtin_inf();
    else v = (1.0 + man / 1024.0) * __builtin_ldexp(1.0, exp - 15);
    return sign ? -v : v;
}

} // namespace

extern "C" {

int mrl_tensor_file_open(const char *path, mrl_tensor_file **out)
{
    if (!path || !out) return MRL_ERR_INVALID;
    *out = nullptr;
    FILE *fp = std::fopen(path, "rb");
    if (!fp) { t_open_error = std::string("cannot open ") + path; return MRL_ERR_IO; }
    mrl_tensor_file *f = new (std::nothrow) mrl_tensor_file();
    if (!f) { std::fclose(fp); return MRL_ERR_OOM; }
    int rc = MRL_OK;
    if (std::fseek(fp, 0, SEEK_END) != 0) rc = MRL_ERR_IO;
    const long size = rc == MRL_OK ? std::ftell(fp) : -1;
    if (size < 0 || std::fseek(fp, 0, SEEK_SET) != 0) rc = MRL_ERR_IO;
    if (rc == MRL_OK) {
        try { f->bytes.resize((size_t)size); } catch (const std::bad_alloc &) { rc = MRL_ERR_OOM; }
    }
    if (rc == MRL_OK && size > 0 && std::fread(f->bytes.data(), 1, (size_t)size, fp) != (size_t)size) rc = MRL_ERR_IO;
    std::fclose(fp);
    if (rc == MRL_OK) rc = parse(f);
    if (rc != MRL_OK) { t_open_error = f->error.empty() ? std::string("cannot read ") + path : f->error; delete f; return rc; }
    *out = f;
    return MRL_OK;
}

int mrl_tensor_file_close(mrl_tensor_file *f) { delete f; return MRL_OK; }

const char *mrl_tensor_file_last_error(const mrl_tensor_file *f) { return f ? f->error.c_str() : t_open_error.c_str(); }

int mrl_tensor_file_field_count(const mrl_tensor_file *f) { return f ? (int)f->fields.size() : MRL_ERR_INVALID; }

int mrl_tensor_file_find(const mrl_tensor_file *f, const char *name)
{
    if (!f || !name) return MRL_ERR_INVALID;
    for (size_t i = 0; i < f->fields.size(); ++i)
        if (f->fields[i].name == name) return (int)i;
    return MRL_ERR_FORMAT;
}

int mrl_tensor_file_field_info(const mrl_tensor_file *f, int index, const char **name, int *dtype, int *ndim, const uint64_t **shape)
{
    if (!f || index < 0 || (size_t)index >= f->fields.size()) return MRL_ERR_INVALID;
    const Field &fd = f->fields[(size_t)index];
    if (name) *name = fd.name.c_str();
    if (dtype) *dtype = fd.dtype;
    if (ndim) *ndim = (int)fd.shape.size();
    if (shape) *shape = fd.shape.data();
    return MRL_OK;
}

const void *mrl_tensor_file_field_data(const mrl_tensor_file *f, int index, size_t *bytes)
{
    if (!f || index < 0 || (size_t)index >= f->fields.size()) return nullptr;
    const Field &fd = f->fields[(size_t)index];
    if (bytes) *bytes = (size_t)fd.count * dtype_size(fd.dtype);
    return f->bytes.data() + fd.offset;
}

int mrl_tensor_file_read_f64(const mrl_tensor_file *f, int index, double *out, size_t capacity)
{
    if (!f || index < 0 || (size_t)index >= f->fields.size() || (!out && capacity)) return MRL_ERR_INVALID;
    const Field &fd = f->fields[(size_t)index];
    if (fd.count > capacity) return MRL_ERR_INVALID;
    const unsigned char *p = f->bytes.data() + fd.offset;
    for (uint64_t i = 0; i < fd.count; ++i) {
        switch (fd.dtype) {
            case 9:  { uint16_t v; std::memcpy(&v, p + 2 * i, 2); out[i] = half_to_double(v); break; }
            case 10: { float v; std::memcpy(&v, p + 4 * i, 4); out[i] = (double)v; break; }
            case 11: { std::memcpy(&out[i], p + 8 * i, 8); break; }
            default: return MRL_ERR_FORMAT;                              // integer fields: use mrl_tensor_file_field_data
        }
    }
    return MRL_OK;
}

int mrl_material_load_tensor_table(mrl_ctx *ctx, const char *path, const char *field, int *out_id, int *out_channels)
{
    if (!ctx || !path || !out_id) return MRL_ERR_INVALID;
    mrl_tensor_file *f = nullptr;
    int rc = mrl_tensor_file_open(path, &f);
    if (rc != MRL_OK) return rc;
    t_open_error.clear();                                   // from here on a non-empty text is this call's own refusal
    const char *want = field ? field : "table";
    const int at = mrl_tensor_file_find(f, want);
    if (at < 0) { t_open_error = std::string("no field \"") + want + "\" in " + path; mrl_tensor_file_close(f); return MRL_ERR_FORMAT; }
    const Field &fd = f->fields[(size_t)at];
    if (fd.shape.size() != 4 || (fd.dtype != 10 && fd.dtype != 11) || fd.shape[0] < 1 || fd.shape[0] > 32 ||
        fd.shape[1] < 1 || fd.shape[2] < 1 || fd.shape[3] < 1 || fd.shape[1] * fd.shape[2] * fd.shape[3] > (1ull << 28)) {
        t_open_error = std::string("field \"") + want + "\" is not a float table of shape [channels <= 32, n_theta_h, n_theta_d, n_phi_d]";
        mrl_tensor_file_close(f);
        return MRL_ERR_FORMAT;
    }
    const int n_ch = (int)fd.shape[0];
    const int dims[3] = { (int)fd.shape[1], (int)fd.shape[2], (int)fd.shape[3] };
    std::vector<double> data, scale((size_t)n_ch, 1.0);
    try { data.resize((size_t)fd.count); } catch (const std::bad_alloc &) { mrl_tensor_file_close(f); return MRL_ERR_OOM; }
    rc = mrl_tensor_file_read_f64(f, at, data.data(), data.size());
    const int sc = mrl_tensor_file_find(f, "scale");
    if (rc == MRL_OK && sc >= 0) {
        const Field &sf = f->fields[(size_t)sc];
        if (sf.count != (uint64_t)n_ch) { t_open_error = "field \"scale\" must hold one factor per channel"; rc = MRL_ERR_FORMAT; }
        else rc = mrl_tensor_file_read_f64(f, sc, scale.data(), scale.size());
    }
    // optional: the file names its own parameterisation (one integer of any integer or float type)
    int param = -1;
    const int pf = mrl_tensor_file_find(f, "parameterization");
    if (rc == MRL_OK && pf >= 0) {
        const Field &q = f->fields[(size_t)pf];
        const unsigned char *b = f->bytes.data() + q.offset;
        long long v = -1;
        double dv = 0.0;
        if (q.count != 1) v = -1;
        else if (q.dtype == 1 || q.dtype == 2) v = q.dtype == 1 ? (long long)*b : (long long)*(const signed char *)b;
        else if (q.dtype == 3 || q.dtype == 4) { int16_t t; std::memcpy(&t, b, 2); v = q.dtype == 3 ? (long long)(uint16_t)t : (long long)t; }
        else if (q.dtype == 5 || q.dtype == 6) { int32_t t; std::memcpy(&t, b, 4); v = q.dtype == 5 ? (long long)(uint32_t)t : (long long)t; }
        else if (q.dtype == 7 || q.dtype == 8) { int64_t t; std::memcpy(&t, b, 8); v = (long long)t; }
        // a float field: range-check BEFORE converting (NaN, inf or |dv| >= 2^63 make the cast undefined behaviour)
        else if (mrl_tensor_file_read_f64(f, pf, &dv, 1) == MRL_OK && dv >= 0.0 && dv <= 2.0 && dv == std::floor(dv)) v = (long long)dv;
        if (v < MRL_PARAM_HALF_DIFF || v > MRL_PARAM_STANDARD_FULL) { t_open_error = "field \"parameterization\" must hold one integer 0..2 (enum mrl_param)"; rc = MRL_ERR_FORMAT; }
        else param = (int)v;
    }
    mrl_tensor_file_close(f);
    if (rc != MRL_OK) return rc;
    rc = param >= 0 ? mrl_material_upload_table_param(ctx, data.data(), dims, n_ch, scale.data(), param, out_id)
                    : mrl_material_upload_table_nch(ctx, data.data(), dims, n_ch, scale.data(), out_id);
    if (rc == MRL_OK && out_channels) *out_channels = n_ch;
    return rc;
}

int mrl_material_load_rgl(mrl_ctx *ctx, const char *path, int *out_id)
{
    if (!ctx || !path || !out_id) return MRL_ERR_INVALID;
    mrl_tensor_file *f = nullptr;
    int rc = mrl_tensor_file_open(path, &f);
    if (rc != MRL_OK) return rc;
    t_open_error.clear();                                   // from here on a non-empty text is this call's own refusal
    auto refuse = [&](const std::string &why) { t_open_error = why + " (" + path + ")"; mrl_tensor_file_close(f); return (int)MRL_ERR_FORMAT; };
    // a Float field of the given rank; returns nullptr (and the reason in `why`) otherwise
    std::string why;
    auto floats = [&](const char *name, size_t rank) -> const Field * {
        const int at = mrl_tensor_file_find(f, name);
        if (at < 0) { why = std::string("no field \"") + name + "\""; return nullptr; }
        const Field &fd = f->fields[(size_t)at];
        if (fd.dtype != 10 || fd.shape.size() != rank) { why = std::string("field \"") + name + "\" must be float32 of rank " + std::to_string(rank); return nullptr; }
        if (fd.offset % 4 != 0) { why = std::string("field \"") + name + "\" is not 4-byte aligned"; return nullptr; }
        for (uint64_t d : fd.shape) if (d < 1 || d > 8192) { why = std::string("field \"") + name + "\": every axis must have 1..8192 entries"; return nullptr; }
        return &fd;
    };
    // a spectral file holds "spectra" [n_phi, n_theta, n_wavelengths, res, res] + "wavelengths" instead of "rgb"
    const bool spectral = mrl_tensor_file_find(f, "rgb") < 0 && mrl_tensor_file_find(f, "spectra") >= 0;
    const Field *phi = floats("phi_i", 1);
    const Field *theta = phi ? floats("theta_i", 1) : nullptr;
    const Field *ndf = theta ? floats("ndf", 2) : nullptr;
    const Field *sigma = ndf ? floats("sigma", 2) : nullptr;
    const Field *vndf = sigma ? floats("vndf", 4) : nullptr;
    const Field *lum = vndf ? floats("luminance", 4) : nullptr;
    const Field *rgb = lum ? floats(spectral ? "spectra" : "rgb", 5) : nullptr;
    const Field *wavelengths = (rgb && spectral) ? floats("wavelengths", 1) : nullptr;
    if (!rgb || (spectral && !wavelengths)) return refuse(why);
    const uint64_t n_phi = phi->shape[0], n_theta = theta->shape[0], n_values = spectral ? wavelengths->shape[0] : 3;
    if (vndf->shape[0] != n_phi || vndf->shape[1] != n_theta || lum->shape != vndf->shape || rgb->shape[0] != n_phi || rgb->shape[1] != n_theta ||
        rgb->shape[2] != n_values || rgb->shape[3] != vndf->shape[2] || rgb->shape[4] != vndf->shape[3])
        return refuse(spectral ? "vndf / luminance must be [n_phi, n_theta, res, res] and spectra [n_phi, n_theta, n_wavelengths, res, res]"
                               : "vndf / luminance must be [n_phi, n_theta, res, res] and rgb [n_phi, n_theta, 3, res, res]");
    int jacobian = 0;
    const int jf = mrl_tensor_file_find(f, "jacobian");
    if (jf >= 0) {
        const Field &q = f->fields[(size_t)jf];
        if (q.count != 1 || dtype_size(q.dtype) != 1) return refuse("field \"jacobian\" must hold one byte");
        jacobian = f->bytes[q.offset] != 0;
    }
    auto data = [&](const Field *fd) { return (const float *)(const void *)(f->bytes.data() + fd->offset); };
    mrl_rgl_fields r;
    r.n_phi = (int)n_phi; r.n_theta = (int)n_theta;
    r.phi_i = data(phi); r.theta_i = data(theta);
    r.res_ndf[0] = (int)ndf->shape[1]; r.res_ndf[1] = (int)ndf->shape[0];
    r.res_sigma[0] = (int)sigma->shape[1]; r.res_sigma[1] = (int)sigma->shape[0];
    r.res[0] = (int)vndf->shape[3]; r.res[1] = (int)vndf->shape[2];
    r.ndf = data(ndf); r.sigma = data(sigma); r.vndf = data(vndf); r.luminance = data(lum); r.rgb = data(rgb);
    r.jacobian = jacobian;
    if (spectral) {
        mrl_rgl_spectral_fields sp;
        sp.base = r;
        sp.base.rgb = nullptr;
        sp.n_wavelengths = (int)n_values; sp.wavelengths = data(wavelengths); sp.spectra = data(rgb);
        rc = mrl_material_upload_rgl_spectral(ctx, &sp, out_id);
    } else {
        rc = mrl_material_upload_rgl(ctx, &r, out_id);
    }
    mrl_tensor_file_close(f);
    return rc;
}

} // extern "C"
