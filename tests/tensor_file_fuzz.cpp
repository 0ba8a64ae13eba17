// tensor_file_fuzz.cpp — the tensor_file container reader (csrc/merl_tensor_file.hip, host code) under
// AddressSanitizer + UBSan: a well-formed file, then thousands of corrupted copies (byte flips, truncations, grown
// counts, moved offsets).  Every open must either fail with a status or yield fields whose payloads lie inside the
// file; nothing may read out of bounds.  Built and run by tests/test_sanitize_cpu.py (g++ -x c++ on the .hip source).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/merl_hip.h"

// the only context call the reader makes; records what it was handed
static int g_uploads = 0;
extern "C" int mrl_material_upload_table_nch(mrl_ctx *, const double *planar, const int dims[3], int n_channels, const double *scale, int *out_id)
{
    double sum = 0.0;
    const size_t n = (size_t)dims[0] * dims[1] * dims[2] * (size_t)n_channels;
    for (size_t i = 0; i < n; ++i) sum += planar[i];              // touches every value: ASAN sees an undersized buffer
    for (int c = 0; c < n_channels; ++c) sum += scale[c];
    *out_id = sum == 12345.0 ? 1 : 0;
    ++g_uploads;
    return MRL_OK;
}

// a file that names its parameterisation goes through the explicit-parameterisation upload
static int g_param_uploads = 0;
extern "C" int mrl_material_upload_table_param(mrl_ctx *c, const double *planar, const int dims[3], int n_channels, const double *scale, int param, int *out_id)
{
    if (param < 0 || param > 2) std::abort();                     // the reader validates the field before it gets here
    ++g_param_uploads;
    return mrl_material_upload_table_nch(c, planar, dims, n_channels, scale, out_id);
}

// the RGL loader's only context call: every array is read to the extent its shape claims
static int g_rgl_uploads = 0;
extern "C" int mrl_material_upload_rgl(mrl_ctx *, const mrl_rgl_fields *f, int *out_id)
{
    double sum = 0.0;
    auto eat = [&](const float *p, size_t n) { for (size_t i = 0; i < n; ++i) sum += p[i]; };
    const size_t slices = (size_t)f->n_phi * (size_t)f->n_theta, per = (size_t)f->res[0] * (size_t)f->res[1];
    eat(f->phi_i, (size_t)f->n_phi); eat(f->theta_i, (size_t)f->n_theta);
    eat(f->ndf, (size_t)f->res_ndf[0] * f->res_ndf[1]); eat(f->sigma, (size_t)f->res_sigma[0] * f->res_sigma[1]);
    eat(f->vndf, slices * per); eat(f->luminance, slices * per); eat(f->rgb, slices * per * 3);
    *out_id = sum == 12345.0 ? 1 : 0;
    ++g_rgl_uploads;
    return MRL_OK;
}

// ... and of a spectral file: spectra [n_phi][n_theta][n_wavelengths][res][res] over wavelengths [n_wavelengths]
static int g_spectral_uploads = 0;
extern "C" int mrl_material_upload_rgl_spectral(mrl_ctx *, const mrl_rgl_spectral_fields *sp, int *out_id)
{
    const mrl_rgl_fields *f = &sp->base;
    double sum = 0.0;
    auto eat = [&](const float *p, size_t n) { for (size_t i = 0; i < n; ++i) sum += p[i]; };
    const size_t slices = (size_t)f->n_phi * (size_t)f->n_theta, per = (size_t)f->res[0] * (size_t)f->res[1];
    if (f->rgb || sp->n_wavelengths < 1) std::abort();
    eat(f->phi_i, (size_t)f->n_phi); eat(f->theta_i, (size_t)f->n_theta);
    eat(f->ndf, (size_t)f->res_ndf[0] * f->res_ndf[1]); eat(f->sigma, (size_t)f->res_sigma[0] * f->res_sigma[1]);
    eat(f->vndf, slices * per); eat(f->luminance, slices * per);
    eat(sp->wavelengths, (size_t)sp->n_wavelengths); eat(sp->spectra, slices * per * (size_t)sp->n_wavelengths);
    *out_id = sum == 12345.0 ? 1 : 0;
    ++g_spectral_uploads;
    return MRL_OK;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static void put(std::vector<unsigned char> &b, const void *p, size_t n) { const unsigned char *c = (const unsigned char *)p; b.insert(b.end(), c, c + n); }

// param_as_f64 != nullptr: the "parameterization" field is a float64 scalar holding *param_as_f64 (files may say so)
static std::vector<unsigned char> good_file(const double *param_as_f64 = nullptr)
{
    std::vector<unsigned char> b;
    put(b, "tensor_file", 12);
    const uint8_t ver[2] = { 1, 0 }; put(b, ver, 2);
    const uint32_t nf = 4; put(b, &nf, 4);
    struct F { const char *name; uint16_t ndim; uint8_t dtype; std::vector<uint64_t> shape; size_t bytes; };
    const F fs[4] = { { "table", 4, 10, { 2, 3, 2, 4 }, 2 * 3 * 2 * 4 * 4 }, { "scale", 1, 11, { 2 }, 16 }, { "description", 1, 1, { 9 }, 9 },
                      { "parameterization", 0, (uint8_t)(param_as_f64 ? 11 : 1), {}, (size_t)(param_as_f64 ? 8 : 1) } };
    size_t head = b.size();
    for (const F &f : fs) head += 2 + std::strlen(f.name) + 2 + 1 + 8 + 8 * f.ndim;
    uint64_t off = (head + 7) / 8 * 8;
    std::vector<uint64_t> offs;
    for (const F &f : fs) { offs.push_back(off); off = (off + f.bytes + 7) / 8 * 8; }
    for (int i = 0; i < 4; ++i) {
        const F &f = fs[i];
        const uint16_t nl = (uint16_t)std::strlen(f.name); put(b, &nl, 2); put(b, f.name, nl);
        put(b, &f.ndim, 2); put(b, &f.dtype, 1); put(b, &offs[i], 8);
        for (uint64_t e : f.shape) put(b, &e, 8);
    }
    for (int i = 0; i < 4; ++i) {
        b.resize(offs[i], 0);
        if (i == 3 && param_as_f64) { put(b, param_as_f64, 8); continue; }
        for (size_t k = 0; k < fs[i].bytes; ++k) b.push_back(i == 3 ? (unsigned char)1 : (unsigned char)(k * 7 + i));    // parameterization = 1
    }
    return b;
}

// a well-formed file with the RGL field names: n_phi = 1, n_theta = 2, 3 x 2 warps, 2 x 2 ndf, 3 x 2 sigma
static std::vector<unsigned char> good_rgl_file(bool spectral = false)
{
    std::vector<unsigned char> b;
    put(b, "tensor_file", 12);
    const uint8_t ver[2] = { 1, 0 }; put(b, ver, 2);
    struct F { const char *name; uint8_t dtype; std::vector<uint64_t> shape; };
    const std::vector<F> fs = { { "phi_i", 10, { 1 } }, { "theta_i", 10, { 2 } }, { "ndf", 10, { 2, 2 } }, { "sigma", 10, { 2, 3 } },
                                { "vndf", 10, { 1, 2, 2, 3 } }, { "luminance", 10, { 1, 2, 2, 3 } },
                                spectral ? F{ "spectra", 10, { 1, 2, 5, 2, 3 } } : F{ "rgb", 10, { 1, 2, 3, 2, 3 } },
                                spectral ? F{ "wavelengths", 10, { 5 } } : F{ "jacobian", 1, { 1 } } };
    const uint32_t nf = (uint32_t)fs.size(); put(b, &nf, 4);
    size_t head = b.size();
    for (const F &f : fs) head += 2 + std::strlen(f.name) + 2 + 1 + 8 + 8 * f.shape.size();
    uint64_t off = (head + 7) / 8 * 8;
    std::vector<uint64_t> offs, sizes;
    for (const F &f : fs) {
        uint64_t n = f.dtype == 10 ? 4 : 1;
        for (uint64_t e : f.shape) n *= e;
        offs.push_back(off); sizes.push_back(n); off = (off + n + 7) / 8 * 8;
    }
    for (size_t i = 0; i < fs.size(); ++i) {
        const F &f = fs[i];
        const uint16_t nl = (uint16_t)std::strlen(f.name), nd = (uint16_t)f.shape.size(); put(b, &nl, 2); put(b, f.name, nl);
        put(b, &nd, 2); put(b, &f.dtype, 1); put(b, &offs[i], 8);
        for (uint64_t e : f.shape) put(b, &e, 8);
    }
    for (size_t i = 0; i < fs.size(); ++i) {
        b.resize(offs[i], 0);
        if (fs[i].dtype != 10) { b.push_back(1); continue; }
        for (uint64_t k = 0; k < sizes[i] / 4; ++k) { const float v = 0.25f + 0.125f * (float)k + (float)i; put(b, &v, 4); }
    }
    return b;
}

static int probe(const std::string &path, const std::vector<unsigned char> &bytes)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return -100;
    if (!bytes.empty()) std::fwrite(bytes.data(), 1, bytes.size(), f);
    std::fclose(f);
    mrl_tensor_file *t = nullptr;
    int rc = mrl_tensor_file_open(path.c_str(), &t);
    if (rc != MRL_OK) { (void)mrl_tensor_file_last_error(nullptr); return rc; }
    const int n = mrl_tensor_file_field_count(t);
    for (int i = 0; i < n; ++i) {
        const char *name; int dtype, ndim; const uint64_t *shape;
        if (mrl_tensor_file_field_info(t, i, &name, &dtype, &ndim, &shape) != MRL_OK) return -101;
        size_t nbytes = 0;
        const unsigned char *p = (const unsigned char *)mrl_tensor_file_field_data(t, i, &nbytes);
        unsigned acc = 0;
        for (size_t k = 0; k < nbytes; ++k) acc += p[k];               // every payload byte must be readable
        uint64_t count = 1;
        for (int d = 0; d < ndim; ++d) count *= shape[d];
        if (dtype >= 9 && count < (1u << 20)) {
            std::vector<double> out(count ? count : 1);
            (void)mrl_tensor_file_read_f64(t, i, out.data(), count);
            (void)mrl_tensor_file_read_f64(t, i, out.data(), count ? count - 1 : 0);       // too small a buffer: must refuse
        }
        (void)mrl_tensor_file_find(t, name);
        (void)acc;
    }
    (void)mrl_tensor_file_field_info(t, n, nullptr, nullptr, nullptr, nullptr);
    (void)mrl_tensor_file_field_data(t, -1, nullptr);
    mrl_tensor_file_close(t);
    int id = -1, ch = 0;
    (void)mrl_material_load_tensor_table((mrl_ctx *)0x1, path.c_str(), nullptr, &id, &ch);  // the fake context is never dereferenced
    (void)mrl_material_load_rgl((mrl_ctx *)0x1, path.c_str(), &id);
    return rc;
}

int main(int argc, char **argv)
{
    const std::string path = argc > 1 ? argv[1] : "/tmp/tensor_fuzz.bsdf";
    const std::vector<unsigned char> good = good_file();
    if (probe(path, good) != MRL_OK || g_uploads != 1 || g_param_uploads != 1) { std::fprintf(stderr, "the well-formed file was rejected\n"); return 1; }
    int opened = 0, refused = 0;
    for (int round = 0; round < 6000; ++round) {
        std::vector<unsigned char> b = good;
        const int kind = (int)(rnd() % 5);
        if (kind == 0) {                                               // flip 1..4 bytes of the header / field table
            for (int k = 0, m = 1 + (int)(rnd() % 4); k < m; ++k) b[rnd() % 150] ^= (unsigned char)(1u << (rnd() % 8));
        } else if (kind == 1) {                                        // truncate anywhere
            b.resize(rnd() % b.size());
        } else if (kind == 2) {                                        // overwrite 8 bytes of the field table with a huge or random value
            const uint64_t v = (rnd() % 2) ? ~0ull >> (rnd() % 40) : rnd();
            std::memcpy(&b[18 + rnd() % 130], &v, 8);
        } else if (kind == 3) {                                        // random byte anywhere
            b[rnd() % b.size()] = (unsigned char)rnd();
        } else {                                                       // grow the field count
            const uint32_t nf = 3 + (uint32_t)(rnd() % 70000); std::memcpy(&b[14], &nf, 4);
        }
        const int rc = probe(path, b);
        if (rc == MRL_OK) ++opened; else ++refused;
    }
    // the RGL loader: a well-formed file is handed on once, corrupted copies never make the stub read past a payload
    {
        const std::vector<unsigned char> rgl = good_rgl_file();
        if (probe(path, rgl) != MRL_OK || g_rgl_uploads != 1) { std::fprintf(stderr, "the well-formed RGL file was rejected: %s\n", mrl_tensor_file_last_error(nullptr)); return 1; }
        for (int round = 0; round < 4000; ++round) {
            std::vector<unsigned char> b = rgl;
            const int kind = (int)(rnd() % 4);
            if (kind == 0) { for (int k = 0, m = 1 + (int)(rnd() % 4); k < m; ++k) b[rnd() % 330] ^= (unsigned char)(1u << (rnd() % 8)); }
            else if (kind == 1) b.resize(rnd() % b.size());
            else if (kind == 2) { const uint64_t v = (rnd() % 2) ? ~0ull >> (rnd() % 40) : rnd() % 64; std::memcpy(&b[18 + rnd() % 300], &v, 8); }
            else b[rnd() % b.size()] = (unsigned char)rnd();
            const int rc = probe(path, b);
            if (rc == MRL_OK) ++opened; else ++refused;
        }
        std::printf("rgl files handed on: %d\n", g_rgl_uploads);
        // the spectral variant of the file ("spectra" + "wavelengths" instead of "rgb")
        const std::vector<unsigned char> spec = good_rgl_file(true);
        if (probe(path, spec) != MRL_OK || g_spectral_uploads != 1) { std::fprintf(stderr, "the well-formed spectral RGL file was rejected: %s\n", mrl_tensor_file_last_error(nullptr)); return 1; }
        for (int round = 0; round < 4000; ++round) {
            std::vector<unsigned char> b = spec;
            const int kind = (int)(rnd() % 4);
            if (kind == 0) { for (int k = 0, m = 1 + (int)(rnd() % 4); k < m; ++k) b[rnd() % 340] ^= (unsigned char)(1u << (rnd() % 8)); }
            else if (kind == 1) b.resize(rnd() % b.size());
            else if (kind == 2) { const uint64_t v = (rnd() % 2) ? ~0ull >> (rnd() % 40) : rnd() % 64; std::memcpy(&b[18 + rnd() % 300], &v, 8); }
            else b[rnd() % b.size()] = (unsigned char)rnd();
            const int rc = probe(path, b);
            if (rc == MRL_OK) ++opened; else ++refused;
        }
        std::printf("spectral rgl files handed on: %d\n", g_spectral_uploads);
    }
    // a float-typed "parameterization": only 0.0, 1.0, 2.0 are values of enum mrl_param; NaN, infinities and magnitudes
    // beyond the integer range must be refused BEFORE any float -> integer conversion (-fsanitize=float-cast-overflow)
    {
        const double inf = 1.0 / 0.0, nan = inf - inf;
        const double cases[] = { 0.0, 1.0, 2.0, 1.5, 3.0, -1.0, 1e300, -1e300, 9.3e18, inf, -inf, nan, 4.9e-324 };
        for (double v : cases) {
            const int before = g_param_uploads;
            (void)probe(path, good_file(&v));
            const bool accepted = g_param_uploads == before + 1, valid = v == 0.0 || v == 1.0 || v == 2.0;
            if (accepted != valid) { std::fprintf(stderr, "parameterization = %g: %s\n", v, accepted ? "accepted" : "refused"); return 1; }
        }
    }
    std::remove(path.c_str());
    std::printf("tensor fuzz ok: %d corrupted files opened consistently, %d refused, %d table uploads\n", opened, refused, g_uploads);
    return refused > 0 ? 0 : 1;
}
