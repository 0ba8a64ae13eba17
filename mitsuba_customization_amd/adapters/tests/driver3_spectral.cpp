// driver3_spectral.cpp — stands in for Mitsuba 3's PluginManager + an integrator in the scalar_spectral variant: every ray carries four
// wavelengths (SurfaceInteraction::wavelengths), the BSDF answers with four values.
//   driver3_spectral <measured.so> <file_spec.bsdf> <pairs.bin> <out.bin> <n_scalar>
//   pairs.bin: uint64 n, wi[n][3] wo[n][3] u[n][2], then wavelengths[n][4];  out.bin: per unit values[4] pdf wo'[3] pdf' weight'[4] (13 floats),
//   first the scalar-call block (n_scalar units), then the batch block (n units)
#include <dlfcn.h>

#include <cstring>
#include <iostream>

#include <mitsuba/mitsuba3.h>

#include "../common/batched_bsdf.hpp"
#include "driver_common.hpp"

using namespace mitsuba;
using SpectralBSDF = BSDF<float, Spectrum4f>;
typedef void *(*CreateFn)(const Properties &);

int main(int argc, char **argv)
{
    if (argc < 6) { std::cerr << "usage\n"; return 2; }
    void *h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { std::cerr << "dlopen: " << dlerror() << "\n"; return 3; }
    auto create = (CreateFn)dlsym(h, "plugin_create_scalar_spectral");
    if (!create) { std::cerr << "plugin lacks plugin_create_scalar_spectral\n"; return 3; }
    Properties props("bsdf");
    props.set_string("filename", argv[2]);
    SpectralBSDF *bsdf = nullptr;
    try {
        bsdf = static_cast<SpectralBSDF *>(create(props));
    } catch (const std::exception &e) {
        std::cerr << "constructor threw: " << e.what() << "\n";
        return 5;
    }
    std::cout << bsdf->to_string() << "\n";
    Pairs p;
    std::vector<float> wl;
    {
        FILE *f = std::fopen(argv[3], "rb");
        if (!f || std::fread(&p.n, 8, 1, f) != 1) return 2;
        p.wi.resize(3 * p.n); p.wo.resize(3 * p.n); p.u.resize(2 * p.n); wl.resize(4 * p.n);
        if (std::fread(p.wi.data(), 4, 3 * p.n, f) != 3 * p.n || std::fread(p.wo.data(), 4, 3 * p.n, f) != 3 * p.n ||
            std::fread(p.u.data(), 4, 2 * p.n, f) != 2 * p.n || std::fread(wl.data(), 4, 4 * p.n, f) != 4 * p.n) return 2;
        std::fclose(f);
    }
    const size_t m = std::min<size_t>(p.n, (size_t)atoll(argv[5]));
    std::vector<float> scalar(13 * m), batch(13 * p.n);
    BSDFContext ctx;
    SpectralBSDF::SurfaceInteraction3f si;
    for (size_t i = 0; i < m; ++i) {
        si.wi = Vector3f(p.wi[3 * i], p.wi[3 * i + 1], p.wi[3 * i + 2]);
        si.wavelengths = Spectrum4f(wl[4 * i], wl[4 * i + 1], wl[4 * i + 2], wl[4 * i + 3]);
        Vector3f wo(p.wo[3 * i], p.wo[3 * i + 1], p.wo[3 * i + 2]);
        Spectrum4f f = bsdf->eval(ctx, si, wo, true);
        float pdf = bsdf->pdf(ctx, si, wo, true);
        auto fp = bsdf->eval_pdf(ctx, si, wo, true);
        if (fp.first[0] != f[0] || fp.first[3] != f[3] || fp.second != pdf) return 7;
        auto sw = bsdf->sample(ctx, si, 0.5f, Point2f(p.u[2 * i], p.u[2 * i + 1]), true);
        float *o = &scalar[13 * i];
        for (int k = 0; k < 4; ++k) { o[k] = f[k]; o[9 + k] = sw.second[k]; }
        o[4] = pdf; o[5] = sw.first.wo.x(); o[6] = sw.first.wo.y(); o[7] = sw.first.wo.z(); o[8] = sw.first.pdf;
    }
    const BatchedBSDF *wave = dynamic_cast<const BatchedBSDF *>(bsdf);
    if (!wave) return 9;
    std::vector<float> val(4 * p.n), pdf(p.n), wo2(3 * p.n), pdf2(p.n), wgt(4 * p.n);
    wave->evalSampleSpectralBatch(p.wi.data(), p.wo.data(), p.u.data(), wl.data(), 4, p.n, val.data(), pdf.data(), wo2.data(), pdf2.data(), wgt.data());
    wave->synchronize();
    for (size_t i = 0; i < p.n; ++i) {
        float *o = &batch[13 * i];
        for (int k = 0; k < 4; ++k) { o[k] = val[4 * i + k]; o[9 + k] = wgt[4 * i + k]; }
        o[4] = pdf[i]; o[5] = wo2[3 * i]; o[6] = wo2[3 * i + 1]; o[7] = wo2[3 * i + 2]; o[8] = pdf2[i];
    }
    FILE *f = std::fopen(argv[4], "wb");
    if (!f) return 2;
    write_floats(f, scalar); write_floats(f, batch);
    std::fclose(f);
    delete bsdf;
    std::cout << "driver3_spectral ok: " << m << " scalar, " << p.n << " batched units\n";
    return 0;
}
