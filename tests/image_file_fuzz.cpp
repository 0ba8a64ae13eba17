// image_file_fuzz.cpp — the header side of the on-disk material image (csrc/merl_image_file.hpp, pure host C++) under
// AddressSanitizer + UBSan: well-formed headers of every kind, then a million corrupted copies (byte flips, random words, huge and
// negative shapes) against right and wrong file lengths.  Whenever image_plan() accepts a header, what it planned must be
// consistent: payload + header == file length, every table of an RGL image inside the image, sizes that follow from the shapes.
// Built and run by tests/test_sanitize_cpu.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../mitsuba_customization_amd/csrc/merl_image_file.hpp"

using namespace mrl;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static ImageHeader table_header(uint32_t kind, int a, int b, int c, uint32_t n_ch, bool rows2d)
{
    ImageHeader h;
    std::memset(&h, 0, sizeof h);
    std::memcpy(h.magic, kImageMagic, 8);
    h.header_bytes = sizeof h; h.kind = kind; h.n_ch = n_ch; h.lookup = 1; h.node = 0;
    h.layout = kind == kImgKindNch ? kImgLayoutBrick : kImgLayoutRows;
    h.dims[0] = a; h.dims[1] = b; h.dims[2] = c;
    const uint64_t plane = (uint64_t)a * b * c;
    h.texel_bytes = kind == kImgKindNch ? plane * nch_brick_float4s((int)n_ch) * 16 : (uint64_t)(a + 1) * (b + 1) * (c + 1) * 16;
    h.sampling_doubles = 3 * (uint64_t)a + 2;
    if (rows2d) { h.n_ti = kImgIncidentBins; h.sampling2d_doubles = (uint64_t)kImgIncidentBins * (2 * (uint64_t)a + 1); }
    return h;
}

static ImageHeader rgl_header(int n_phi, int n_theta, int res, int res_ndf, int res_sigma)
{
    ImageHeader h;
    std::memset(&h, 0, sizeof h);
    std::memcpy(h.magic, kImageMagic, 8);
    h.header_bytes = sizeof h; h.kind = kImgKindRgl; h.n_ch = 3;
    const int32_t s[8] = { n_phi, n_theta, res, res, res_ndf, res_ndf, res_sigma, res_sigma };
    std::memcpy(h.rgl_shape, s, sizeof s);
    h.rgl_flags[0] = 1;
    RglLayout l;
    h.texel_bytes = (uint64_t)rgl_plan_layout(rgl_shapes_of(s, 1), l) * 4;
    return h;
}

static unsigned long long length_of(const ImageHeader &h) { return sizeof h + h.texel_bytes + (h.sampling_doubles + h.sampling2d_doubles) * 8; }

// an accepted plan must hold together
static bool consistent(const ImageHeader &h, unsigned long long file_bytes, const ImagePlan &p)
{
    if (sizeof(ImageHeader) + (unsigned long long)p.payload_bytes != file_bytes) return false;
    if (p.payload_bytes != p.texel_bytes + (p.sampling_doubles + p.sampling2d_doubles) * 8) return false;
    if (p.texel_bytes != h.texel_bytes) return false;
    if (p.is_rgl) {
        const RglFields &f = p.shapes;
        // everything is stored per parameter bracket: a distribution's cells as records (integrals left of the cell, corner values per
        // slice, row totals), the measured values as the bracket's 1 / 2 / 4 slices side by side; tables start on 128-B boundaries
        const uint64_t floats = p.texel_bytes / 4;
        const uint64_t brackets = (uint64_t)(f.n_phi > 1 ? f.n_phi - 1 : 1) * (f.n_theta > 1 ? f.n_theta - 1 : 1);
        const uint64_t in_bracket = (uint64_t)(f.n_phi > 1 ? 2 : 1) * (f.n_theta > 1 ? 2 : 1), record = 2 * (uint64_t)(f.n_phi > 1 ? 2 : 1) + in_bracket;
        const uint64_t per_c = (uint64_t)(f.res[0] - 1) * (f.res[1] - 1);
        const uint64_t cells[5] = { (uint64_t)(f.res_ndf[0] - 1) * (f.res_ndf[1] - 1), (uint64_t)(f.res_sigma[0] - 1) * (f.res_sigma[1] - 1),
                                    per_c * brackets * record, per_c * brackets * record, per_c * brackets * in_bracket * 3 };
        for (int w = 0; w < 5; ++w) {
            if (p.layout.cells[w] % 32 != 0 || p.layout.cells[w] + cells[w] * 4 > floats) return false;
            if (w == 2 || w == 3)
                if (p.layout.margq[w] % 32 != 0 || p.layout.margq[w] + (uint64_t)(f.res[1] - 1) * brackets * 4 > floats ||
                    p.layout.rowh[w] % 32 != 0 || p.layout.rowh[w] + (uint64_t)(f.res[1] - 1) * brackets * 80 * (f.n_phi > 1 ? 2 : 1) > floats) return false;
        }
        if (p.layout.theta + (uint64_t)f.n_theta > floats) return false;
    } else {
        if (p.dims[0] < 1 || p.dims[1] < 1 || p.dims[2] < 1 || (uint64_t)p.dims[0] * p.dims[1] * p.dims[2] > ((uint64_t)1 << 28)) return false;
        if (p.sampling_doubles != 3 * (uint64_t)p.dims[0] + 2) return false;
    }
    return true;
}

int main()
{
    const ImageHeader good[] = { table_header(kImgKindMerl, 90, 90, 180, 3, true), table_header(kImgKindTable, 6, 5, 8, 3, true),
                                 table_header(kImgKindTable, 7, 3, 9, 3, false), table_header(kImgKindNch, 10, 8, 12, 5, false),
                                 table_header(kImgKindNch, 4, 4, 4, 1, false), rgl_header(1, 8, 32, 128, 64), rgl_header(5, 3, 6, 6, 4) };
    const int n_good = (int)(sizeof good / sizeof good[0]);
    ImagePlan p;
    for (int i = 0; i < n_good; ++i) {
        const char *why = image_plan(good[i], length_of(good[i]), 1, 0, 0, p);
        if (why || !consistent(good[i], length_of(good[i]), p)) { std::fprintf(stderr, "well-formed header %d refused: %s\n", i, why ? why : "inconsistent plan"); return 1; }
        if (!image_plan(good[i], length_of(good[i]) - 1, 1, 0, 0, p) || !image_plan(good[i], length_of(good[i]) + 1, 1, 0, 0, p)) { std::fprintf(stderr, "wrong length accepted\n"); return 1; }
    }
    if (!image_plan(good[0], length_of(good[0]), 0, 0, 0, p) || !image_plan(good[0], length_of(good[0]), 1, 1, 0, p)) { std::fprintf(stderr, "foreign lookup options accepted\n"); return 1; }
    long accepted = 0, refused = 0, checked = 0, content_ok = 0;
    for (long round = 0; round < 1000000; ++round) {
        ImageHeader h = good[rnd() % n_good];
        unsigned char *b = (unsigned char *)&h;
        const int kind = (int)(rnd() % 5);
        if (kind == 0) { for (int k = 0, m = 1 + (int)(rnd() % 4); k < m; ++k) b[rnd() % sizeof h] ^= (unsigned char)(1u << (rnd() % 8)); }
        else if (kind == 1) { const uint32_t v = (rnd() % 2) ? (uint32_t)(0xFFFFFFFFu >> (rnd() % 31)) : (uint32_t)rnd(); std::memcpy(b + 8 + 4 * (rnd() % 21), &v, 4); }
        else if (kind == 2) { const int32_t v = (int32_t)((rnd() % 3 == 0) ? -(int32_t)(rnd() % 100000) : (int32_t)(rnd() % 70000)); std::memcpy(b + 40 + 4 * (rnd() % 13), &v, 4); }
        else if (kind == 3) { const uint64_t v = (rnd() % 2) ? ~0ull >> (rnd() % 60) : rnd(); std::memcpy(b + sizeof h - 32 + 8 * (rnd() % 4), &v, 8); }
        else { b[rnd() % sizeof h] = (unsigned char)rnd(); }
        // the length the (possibly corrupted) header itself claims — the case that must be decided on the shapes — or a random one
        const unsigned long long len = (rnd() % 4) ? length_of(h) : rnd() % (1ull << 36);
        const char *why = image_plan(h, len, 1, 0, 0, p);
        if (!why) {
            ++accepted;
            if (!consistent(h, len, p)) { std::fprintf(stderr, "round %ld: an inconsistent plan was accepted\n", round); return 1; }
            // the content check behind the checksum reads exactly the planned payload, whatever bytes it holds (small plans only:
            // a MERL-sized payload per round would make this a memory benchmark)
            if (p.payload_bytes <= ((size_t)1 << 16) && (round & 3) == 0) {
                std::vector<unsigned char> payload(p.payload_bytes ? p.payload_bytes : 1);
                const int fill = (int)(rnd() % 4);
                for (size_t k = 0; k < p.payload_bytes; ++k) payload[k] = fill == 0 ? 0 : (fill == 1 ? 0xFF : (unsigned char)rnd());
                if (fill == 3) for (size_t k = 0; k + 4 <= p.payload_bytes; k += 4) { const float v = (float)(k / 4 + 1); std::memcpy(&payload[k], &v, 4); }
                const char *bad = image_content_check(p, payload.data());
                checked += 1; content_ok += bad ? 0 : 1;
            }
        } else ++refused;
    }
    std::printf("image header fuzz ok: %ld accepted, %ld refused; content check ran on %ld payloads (%ld passed)\n", accepted, refused, checked, content_ok);
    return 0;
}
