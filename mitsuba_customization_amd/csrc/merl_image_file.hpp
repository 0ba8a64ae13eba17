// merl_image_file.hpp — the on-disk image of a material (mrl_material_save_image / _load_image): header, checksum, and the plan of
// what a header's shapes imply.  Pure host C++ (no HIP type, no allocation proportional to anything a file claims): the part of the
// loader that reads untrusted bytes, kept apart so that it is fuzzed under AddressSanitizer / UBSan on the CPU
// (tests/image_file_fuzz.cpp).  Every size is COMPUTED here from the shapes; the loader then demands that the header's own size
// fields and the file's length agree with them.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace mrl {

// ---- shapes of an RGL material and the layout of its cell-brick image (merl_rgl.hip builds the image, merl_rgl.hpp reads it) ----
struct RglFields {                   // host arrays, as the file holds them (x fastest; res[] = { nx, ny })
    int n_phi, n_theta;
    const float *phi_i, *theta_i;
    int res_ndf[2], res_sigma[2], res[2];
    const float *ndf, *sigma, *vndf, *luminance, *rgb;
    int jacobian;
    int n_wl;                        // 0: an RGB file (rgb [n_phi][n_theta][3][ny][nx]); else a spectral one: `rgb` holds "spectra"
    const float *wavelengths;        // [n_phi][n_theta][n_wl][ny][nx] over this ascending grid
};
struct RglLayout { size_t phi, theta, wavelengths, cells[5], margq[5], rowh[5]; };   // float offsets into the image (ndf, sigma, vndf, luminance, rgb)
inline int rgl_value_channels(const RglFields &f) { return f.n_wl > 0 ? f.n_wl : 3; }

inline const char *rgl_check_shapes(const RglFields &f)
{
    if (f.n_phi < 1 || f.n_theta < 1 || f.n_phi > 4096 || f.n_theta > 4096) return "phi_i / theta_i: 1..4096 nodes each";
    for (int k = 0; k < 2; ++k)
        if (f.res_ndf[k] < 2 || f.res_sigma[k] < 2 || f.res[k] < 2 || f.res_ndf[k] > 8192 || f.res_sigma[k] > 8192 || f.res[k] > 8192)
            return "every table needs 2..8192 nodes per axis";
    if (f.n_wl < 0 || f.n_wl > 4096) return "wavelengths: at most 4096 nodes";
    const size_t slices = (size_t)f.n_phi * (size_t)f.n_theta, per = (size_t)f.res[0] * (size_t)f.res[1];
    if (slices * per * (size_t)rgl_value_channels(f) > ((size_t)1 << 28)) return "tables too large (more than 2^28 values)";
    // the image stores a slice's cells once per parameter bracket it bounds (up to 4 x): 32-bit float4 offsets must still reach
    const size_t pb = f.n_phi > 1 ? (size_t)f.n_phi - 1 : 1, tb = f.n_theta > 1 ? (size_t)f.n_theta - 1 : 1, in_bracket = (f.n_phi > 1 ? 2 : 1) * (f.n_theta > 1 ? 2 : 1);
    const size_t widest = in_bracket * (size_t)rgl_value_channels(f) > 2 * (size_t)(f.n_phi > 1 ? 2 : 1) + in_bracket ? in_bracket * (size_t)rgl_value_channels(f) : 2 * (size_t)(f.n_phi > 1 ? 2 : 1) + in_bracket;
    if (pb * tb * per * widest > ((size_t)1 << 30)) return "tables too large (more than 2^30 cell vectors in the image)";
    return nullptr;
}

struct WarpOffsets { size_t cells = 0, margq = 0, rowh = 0; };
// Everything is stored per parameter BRACKET (merl_rgl.hpp, WarpDev): brackets along theta / phi, the 1 / 2 / 4 slices of a bracket
// side by side, one copy of a slice per bracket it bounds
inline size_t rgl_theta_brackets(int n_theta) { return n_theta > 1 ? (size_t)n_theta - 1 : 1; }
inline size_t rgl_phi_brackets(int n_phi) { return n_phi > 1 ? (size_t)n_phi - 1 : 1; }
inline size_t rgl_bracket_slices(int n_phi, int n_theta) { return (size_t)(n_phi > 1 ? 2 : 1) * (size_t)(n_theta > 1 ? 2 : 1); }
inline size_t rgl_brackets(int n_phi, int n_theta) { return rgl_phi_brackets(n_phi) * rgl_theta_brackets(n_theta); }
// float4s per cell of a DISTRIBUTION's record: the running integrals left of the cell (one float4 per phi node of the bracket), the
// corner values (one per slice), the totals of the cell's two node rows (per phi node): 64 B isotropic, 128 B — one line — anisotropic
inline size_t rgl_record_float4s(int n_phi, int n_theta) { return 2 * (size_t)(n_phi > 1 ? 2 : 1) + rgl_bracket_slices(n_phi, n_theta); }
// float4s per cell row of a distribution's ROW HEADER: five blocks of four entries (one float4 per phi node each: 64 B isotropic, 128 B —
// one line — anisotropic per block) — the row's totals with the conditional integrals at the three columns the column search's first two
// halvings test, then per quarter of the row it can be in by then the three columns of its next two halvings
inline size_t rgl_row_header_float4s(int n_phi) { return 20 * (size_t)(n_phi > 1 ? 2 : 1); }
// where one function's tables go: `at` is the running size of the image in floats (every table starts on a 128-byte boundary, a cache
// line: records and the value vectors of a cell then never straddle one more line than their size asks for)
inline WarpOffsets plan_warp(size_t &at, int nx, int ny, int n_phi, int n_theta, int n_ch, bool distribution)
{
    const size_t cells = (size_t)(nx - 1) * (size_t)(ny - 1), brackets = rgl_brackets(n_phi, n_theta);
    auto grow = [&](size_t floats) { const size_t off = (at + 31) / 32 * 32; at = off + floats; return off; };
    WarpOffsets off;
    if (distribution) {
        off.cells = grow(cells * 4 * brackets * rgl_record_float4s(n_phi, n_theta));
        off.margq = grow((size_t)(ny - 1) * 4 * brackets);
        off.rowh = grow((size_t)(ny - 1) * 4 * brackets * rgl_row_header_float4s(n_phi));
    } else {
        off.cells = grow(cells * 4 * (size_t)n_ch * brackets * rgl_bracket_slices(n_phi, n_theta));
    }
    return off;
}

// the image's layout from the shapes alone; returns the image's size in floats
inline size_t rgl_plan_layout(const RglFields &f, RglLayout &l)
{
    size_t at = (size_t)f.n_phi + (size_t)f.n_theta + (size_t)f.n_wl;
    l.phi = 0; l.theta = (size_t)f.n_phi; l.wavelengths = (size_t)f.n_phi + (size_t)f.n_theta;
    auto put = [&](int which, const int res[2], int n_phi, int n_theta, int n_ch, bool distribution) {
        const WarpOffsets o = plan_warp(at, res[0], res[1], n_phi, n_theta, n_ch, distribution);
        l.cells[which] = o.cells; l.margq[which] = o.margq; l.rowh[which] = o.rowh;
    };
    put(0, f.res_ndf, 1, 1, 1, false);
    put(1, f.res_sigma, 1, 1, 1, false);
    put(2, f.res, f.n_phi, f.n_theta, 1, true);
    put(3, f.res, f.n_phi, f.n_theta, 1, true);
    put(4, f.res, f.n_phi, f.n_theta, rgl_value_channels(f), false);
    return at;
}

// ---- n-channel bricks: float4s per cell ----
inline size_t nch_brick_float4s(int n_ch) { return n_ch == 1 ? 2 : n_ch == 2 ? 4 : 8 * (size_t)((n_ch + 3) / 4); }

// ---- the file ----
struct ImageHeader {
    char magic[8];                       // "MRLIMG\4\0" (2: RGL search tables in the bracket form; 3: the cells too; 4: one record per cell of a distribution; 5: + row headers; 6: with their quarter blocks)
    uint32_t header_bytes, kind, layout, n_ch, param, lookup, node, n_ti;
    int32_t dims[3];
    int32_t rgl_shape[8];                // n_phi n_theta res_x res_y res_ndf_x res_ndf_y res_sigma_x res_sigma_y
    int32_t rgl_flags[2];                // jacobian, wavelength nodes (0: an RGB file)
    uint32_t negative;                   // MRL_OPT_NEGATIVE the table was built under: 0 = negative values were clamped to 0, 1 / 2 = the image holds them
    uint64_t texel_bytes, sampling_doubles, sampling2d_doubles, checksum;
};
constexpr char kImageMagic[8] = { 'M', 'R', 'L', 'I', 'M', 'G', 6, 0 };
constexpr uint64_t kImageChecksumSeed = 0xCBF29CE484222325ull;
// the kinds and layouts an image can name (values of mrl::Kind / mrl::Layout / mrl::Param, repeated here so that this header needs no HIP)
constexpr uint32_t kImgKindMerl = 0, kImgKindTable = 1, kImgKindNch = 4, kImgKindRgl = 5, kImgKindRglSpectral = 6, kImgLayoutRows = 0, kImgLayoutBrick = 1, kImgParamLast = 2;
constexpr int kImgMaxChannels = 32, kImgIncidentBins = 32;

inline uint64_t image_checksum(const void *p, size_t bytes, uint64_t h)
{
    const unsigned char *b = (const unsigned char *)p;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) { uint64_t w; std::memcpy(&w, b + i, 8); h = (h ^ w) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; }
    for (; i < bytes; ++i) { h = (h ^ b[i]) * 0x100000001B3ull; }
    return h;
}

inline RglFields rgl_shapes_of(const int32_t s[8], int jacobian, int n_wl = 0)
{
    RglFields f;
    std::memset(&f, 0, sizeof f);
    f.n_phi = s[0]; f.n_theta = s[1]; f.res[0] = s[2]; f.res[1] = s[3]; f.res_ndf[0] = s[4]; f.res_ndf[1] = s[5]; f.res_sigma[0] = s[6]; f.res_sigma[1] = s[7];
    f.jacobian = jacobian;
    f.n_wl = n_wl;
    return f;
}

// What a header implies.  RGB table images hold the padded rows form whatever layout they came from.
struct ImagePlan {
    bool is_rgl = false, is_nch = false, has_rows2d = false;
    int dims[3] = { 0, 0, 0 }, n_ch = 0, param = 0;
    RglFields shapes;
    RglLayout layout;
    uint64_t texel_bytes = 0, sampling_doubles = 0, sampling2d_doubles = 0;
    size_t payload_bytes = 0;
};

// nullptr and a filled plan, or why the header cannot be an image this library wrote.  file_bytes: the file's length.
// ctx_lookup / ctx_node: the loading context's options (conditional sampling rows are tied to them); ctx_negative: its MRL_OPT_NEGATIVE
// (an image of clamped values serves a clamping context only, an image of raw values the two others).
inline const char *image_plan(const ImageHeader &h, unsigned long long file_bytes, int ctx_lookup, int ctx_node, int ctx_negative, ImagePlan &p)
{
    if (std::memcmp(h.magic, kImageMagic, 8) != 0 || h.header_bytes != sizeof(ImageHeader)) return "not a material image of this library version";
    p = ImagePlan();
    p.is_rgl = h.kind == kImgKindRgl || h.kind == kImgKindRglSpectral; p.is_nch = h.kind == kImgKindNch;
    if (!p.is_rgl && !p.is_nch && h.kind != kImgKindMerl && h.kind != kImgKindTable) return "unknown material kind";
    if (p.is_rgl) {
        if ((h.kind == kImgKindRglSpectral) != (h.rgl_flags[1] > 0)) return "kind and wavelength count disagree";
        if (h.rgl_flags[1] < 0 || h.rgl_flags[1] > 4096) return "bad wavelength count";
        p.shapes = rgl_shapes_of(h.rgl_shape, h.rgl_flags[0] != 0, h.rgl_flags[1]);
        if (const char *why = rgl_check_shapes(p.shapes)) return why;
        p.texel_bytes = (uint64_t)rgl_plan_layout(p.shapes, p.layout) * sizeof(float);
        p.n_ch = rgl_value_channels(p.shapes); p.dims[0] = p.shapes.n_phi; p.dims[1] = p.shapes.n_theta; p.dims[2] = p.shapes.res[0];
    } else {
        for (int k = 0; k < 3; ++k)
            if (h.dims[k] < 1 || h.dims[k] > (1 << 28)) return "table dims out of range";           // each one first: the product below must not overflow
        if ((uint64_t)h.dims[0] * (uint64_t)h.dims[1] > ((uint64_t)1 << 28) || (uint64_t)h.dims[0] * (uint64_t)h.dims[1] * (uint64_t)h.dims[2] > ((uint64_t)1 << 28))
            return "table dims out of range";
        if (h.layout > kImgLayoutBrick || h.param > kImgParamLast) return "bad layout / parameterisation";
        if (h.negative > 2 || (h.negative == 0) != (ctx_negative == 0)) return "the image's table was built under another MRL_OPT_NEGATIVE (clamped against raw values)";
        if (p.is_nch ? (h.n_ch < 1 || h.n_ch > (uint32_t)kImgMaxChannels || h.layout != kImgLayoutBrick) : (h.n_ch != 3 || h.layout != kImgLayoutRows))
            return "bad channel count / layout for the kind (RGB tables are stored in the rows form)";
        if (h.kind == kImgKindMerl && h.param != 0) return "a MERL table is in half / difference angles";
        for (int k = 0; k < 3; ++k) p.dims[k] = h.dims[k];
        p.n_ch = (int)h.n_ch; p.param = (int)h.param;
        const uint64_t plane = (uint64_t)h.dims[0] * (uint64_t)h.dims[1] * (uint64_t)h.dims[2];
        p.texel_bytes = p.is_nch ? plane * nch_brick_float4s(p.n_ch) * 16 : (uint64_t)(h.dims[0] + 1) * (uint64_t)(h.dims[1] + 1) * (uint64_t)(h.dims[2] + 1) * 16;
        p.sampling_doubles = 3 * (uint64_t)h.dims[0] + 2;
        p.has_rows2d = !p.is_nch && h.sampling2d_doubles != 0;
        if (p.has_rows2d) {
            if (h.n_ti != (uint32_t)kImgIncidentBins) return "bad incident-bin count";
            // the conditional rows were integrated through the table's lookup: under other lookup options they are another table
            if ((int)h.lookup != ctx_lookup || (int)h.node != ctx_node) return "the image's conditional sampling rows were built under other lookup / node options";
            p.sampling2d_doubles = (uint64_t)kImgIncidentBins * (2 * (uint64_t)h.dims[0] + 1);
        }
    }
    if (h.texel_bytes != p.texel_bytes || h.sampling_doubles != p.sampling_doubles || h.sampling2d_doubles != p.sampling2d_doubles) return "sizes do not follow from the shapes";
    const uint64_t payload = p.texel_bytes + (p.sampling_doubles + p.sampling2d_doubles) * 8;
    if (payload > ((uint64_t)1 << 40)) return "image too large";
    p.payload_bytes = (size_t)payload;
    if (file_bytes != sizeof(ImageHeader) + payload) return "file length does not match the header";
    return nullptr;
}

// What the checksum cannot say (it is unkeyed: it catches corruption, not a foreign writer): is the CONTENT something the kernels can
// evaluate?  Every value finite; an RGL image's parameter grids (and wavelength grid) strictly ascending, its two distributions and their
// running integrals non-negative; a table image's sampling rows monotone cdfs in [0, 1] with non-negative densities.  Searches are
// index-bounded either way (memory-safe); this keeps a crafted or foreign image from evaluating to NaN / garbage with MRL_OK.
// payload: plan.payload_bytes bytes.  nullptr, or what is wrong.
inline const char *image_content_check(const ImagePlan &p, const void *payload)
{
    auto finite = [](double v) { return v == v && v - v == 0.0; };
    const float *f = (const float *)payload;
    const size_t n_floats = (size_t)(p.texel_bytes / 4);
    for (size_t i = 0; i < n_floats; ++i) if (!finite((double)f[i])) return "non-finite value in the image";
    if (p.is_rgl) {
        const RglFields &s = p.shapes;
        auto ascending = [&](size_t at, int n) { for (int i = 1; i < n; ++i) if (!(f[at + i] > f[at + i - 1])) return false; return true; };
        if (!ascending(p.layout.phi, s.n_phi) || !ascending(p.layout.theta, s.n_theta) || (s.n_wl > 0 && !ascending(p.layout.wavelengths, s.n_wl)))
            return "phi_i / theta_i / wavelengths must be strictly ascending";
        const size_t cells = (size_t)(s.res[0] - 1) * (size_t)(s.res[1] - 1);
        const size_t brackets = rgl_brackets(s.n_phi, s.n_theta);
        for (int w = 2; w <= 3; ++w) {                       // vndf, luminance: densities and their integrals (the records, the marginals)
            const size_t spans[3][2] = { { p.layout.cells[w], cells * 4 * brackets * rgl_record_float4s(s.n_phi, s.n_theta) },
                                         { p.layout.margq[w], (size_t)(s.res[1] - 1) * 4 * brackets },
                                         { p.layout.rowh[w], (size_t)(s.res[1] - 1) * 4 * brackets * rgl_row_header_float4s(s.n_phi) } };
            for (const auto &sp : spans)
                for (size_t i = 0; i < sp[1]; ++i) if (f[sp[0] + i] < 0.0f) return "negative value in a distribution of the image";
        }
    } else {
        const double *d = (const double *)((const char *)payload + p.texel_bytes);
        const size_t n_th = (size_t)p.dims[0];
        for (size_t i = 0; i < (size_t)(p.sampling_doubles + p.sampling2d_doubles); ++i) if (!finite(d[i])) return "non-finite value in the image's sampling rows";
        auto row_ok = [&](const double *cdf, const double *c) {
            for (size_t i = 0; i <= n_th; ++i) if (cdf[i] < 0.0 || cdf[i] > 1.0 + 1e-9 || (i && cdf[i] < cdf[i - 1])) return false;
            for (size_t i = 0; i < n_th; ++i) if (c[i] < 0.0) return false;
            return true;
        };
        if (p.sampling_doubles && !row_ok(d + (n_th + 1), d + 2 * (n_th + 1))) return "the image's sampling marginal is not a distribution";
        const double *r = d + p.sampling_doubles;
        for (int i = 0; p.sampling2d_doubles && i < kImgIncidentBins; ++i, r += 2 * n_th + 1)
            if (!row_ok(r, r + (n_th + 1))) return "a conditional sampling row of the image is not a distribution";
    }
    return nullptr;
}

} // namespace mrl
