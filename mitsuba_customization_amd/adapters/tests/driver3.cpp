// driver3.cpp — stands in for Mitsuba 3's PluginManager + an integrator in the scalar_rgb variant.
//   driver3 <plugin.so> <table.binary> <pairs.bin> <out.bin> <n_scalar> [interpolation] [scale_r scale_g scale_b]
//   driver3 --expect-no-device <plugin.so> <table.binary>
#include <dlfcn.h>

#include <cstring>
#include <iostream>

#include <mitsuba/mitsuba3.h>

#include "../common/batched_bsdf.hpp"
#include "driver_common.hpp"

using namespace mitsuba;
using ScalarBSDF = BSDF<float, Color3f>;
typedef void *(*CreateFn)(const Properties &);
typedef const char *(*NameFn)();

int main(int argc, char **argv)
{
    bool expect_no_device = argc > 1 && std::strcmp(argv[1], "--expect-no-device") == 0;
    if (expect_no_device) { --argc; ++argv; }
    if (argc < 3) { std::cerr << "usage\n"; return 2; }
    void *h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { std::cerr << "dlopen: " << dlerror() << "\n"; return 3; }
    auto create = (CreateFn)dlsym(h, "plugin_create_scalar_rgb");
    auto name = (NameFn)dlsym(h, "plugin_name");
    auto descr = (NameFn)dlsym(h, "plugin_descr");
    if (!create || !name || !descr) { std::cerr << "plugin lacks plugin_name/plugin_descr/plugin_create_scalar_rgb\n"; return 3; }
    std::cout << "plugin: " << name() << " — " << descr() << "\n";

    Properties props("bsdf");
    props.set_string("filename", argv[2]);
    if (expect_no_device) {
        try {
            create(props);
        } catch (const std::exception &e) {
            std::cout << "constructor threw: " << e.what() << "\n";
            return std::strstr(e.what(), "no CPU fallback") ? 0 : 4;
        }
        return 4;
    }
    if (argc >= 5 && std::strcmp(argv[3], "--residency") == 0) {
        const int cycles = argc > 5 ? atoi(argv[5]) : 50;
        try {
            return check_residency([&](int which) {
                                       Properties q("bsdf");
                                       q.set_string("filename", which ? argv[4] : argv[2]);
                                       return static_cast<ScalarBSDF *>(create(q));
                                   },
                                   [](ScalarBSDF *b) { delete b; }, cycles);
        } catch (const std::exception &e) { std::cerr << "residency: " << e.what() << "\n"; return 25; }
    }
    if (argc < 6) { std::cerr << "usage\n"; return 2; }
    if (argc > 6) props.set_string("interpolation", argv[6]);
    if (argc > 9) { props.set_float("scale_r", atof(argv[7])); props.set_float("scale_g", atof(argv[8])); props.set_float("scale_b", atof(argv[9])); }

    if (argc > 10) props.set_string("sampling", argv[10]);
    if (argc > 11) props.set_string("parameterization", argv[11]);
    // where the scalar virtual calls evaluate: MERL_DRIVER_SCALAR = cpu | gpu (unset: the plugin's default, cpu)
    if (const char *sc = std::getenv("MERL_DRIVER_SCALAR")) props.set_string("scalar", sc);
    // further string properties: MERL_DRIVER_PROPS = name=value[,name=value ...]  (e.g. cosine_factor=omitted,negative_values=keep)
    if (const char *more = std::getenv("MERL_DRIVER_PROPS")) {
        std::string all(more);
        size_t at = 0;
        while (at < all.size()) {
            size_t end = all.find(',', at);
            if (end == std::string::npos) end = all.size();
            const std::string item = all.substr(at, end - at);
            const size_t eq = item.find('=');
            if (eq != std::string::npos) props.set_string(item.substr(0, eq), item.substr(eq + 1));
            at = end + 1;
        }
    }
    ScalarBSDF *bsdf = nullptr;
    try {
        bsdf = static_cast<ScalarBSDF *>(create(props));
    } catch (const std::exception &e) {
        std::cerr << "constructor threw: " << e.what() << "\n";
        return 5;
    }
    std::cout << bsdf->to_string() << "\n";
    if (bsdf->component_count() != 1 || bsdf->flags() != (BSDFFlags::GlossyReflection | BSDFFlags::FrontSide)) return 6;

    Pairs p = read_pairs(argv[3]);
    const size_t m = std::min<size_t>(p.n, (size_t)atoll(argv[5]));
    std::vector<float> scalar(11 * m), batch(11 * p.n);
    BSDFContext ctx;
    ScalarBSDF::SurfaceInteraction3f si;
    for (size_t i = 0; i < m; ++i) {
        si.wi = Vector3f(p.wi[3 * i], p.wi[3 * i + 1], p.wi[3 * i + 2]);
        Vector3f wo(p.wo[3 * i], p.wo[3 * i + 1], p.wo[3 * i + 2]);
        Color3f f = bsdf->eval(ctx, si, wo, true);
        float pdf = bsdf->pdf(ctx, si, wo, true);
        auto fp = bsdf->eval_pdf(ctx, si, wo, true);
        if (fp.first[0] != f[0] || fp.first[2] != f[2] || fp.second != pdf) return 7;       // eval_pdf == (eval, pdf)
        auto sw = bsdf->sample(ctx, si, 0.5f, Point2f(p.u[2 * i], p.u[2 * i + 1]), true);
        float *o = &scalar[11 * i];
        o[0] = f[0]; o[1] = f[1]; o[2] = f[2]; o[3] = pdf;
        o[4] = sw.first.wo.x(); o[5] = sw.first.wo.y(); o[6] = sw.first.wo.z(); o[7] = sw.first.pdf;
        o[8] = sw.second[0]; o[9] = sw.second[1]; o[10] = sw.second[2];
        if (sw.first.pdf > 0 && (sw.first.eta != 1.f || sw.first.sampled_component != 0 || sw.first.sampled_type != +BSDFFlags::GlossyReflection)) return 7;
    }
    {   // masked lobe / inactive lane -> zero
        si.wi = Vector3f(0.f, 0.6f, 0.8f);
        BSDFContext off; off.type_mask = +BSDFFlags::DiffuseReflection;
        if (bsdf->eval(off, si, Vector3f(0.6f, 0.f, 0.8f), true)[0] != 0.f) return 8;
        if (bsdf->eval(ctx, si, Vector3f(0.6f, 0.f, 0.8f), false)[0] != 0.f) return 8;
    }
    const BatchedBSDF *wave = dynamic_cast<const BatchedBSDF *>(bsdf);
    if (!wave) return 9;
    std::vector<float> rgb(3 * p.n), pdf(p.n), wo2(3 * p.n), pdf2(p.n), wgt(3 * p.n);
    wave->evalSampleBatch(p.wi.data(), p.wo.data(), p.u.data(), p.n, rgb.data(), pdf.data(), wo2.data(), pdf2.data(), wgt.data());
    wave->synchronize();
    if (!check_queue_call(wave, p, rgb, pdf, wo2, pdf2, wgt)) return 11;
    for (size_t i = 0; i < p.n; ++i) {
        float *o = &batch[11 * i];
        o[0] = rgb[3 * i]; o[1] = rgb[3 * i + 1]; o[2] = rgb[3 * i + 2]; o[3] = pdf[i];
        o[4] = wo2[3 * i]; o[5] = wo2[3 * i + 1]; o[6] = wo2[3 * i + 2]; o[7] = pdf2[i];
        o[8] = wgt[3 * i]; o[9] = wgt[3 * i + 1]; o[10] = wgt[3 * i + 2];
    }
    FILE *f = std::fopen(argv[4], "wb");
    if (!f) return 2;
    write_floats(f, scalar); write_floats(f, batch);
    std::fclose(f);
    delete bsdf;
    std::cout << "driver3 ok: " << m << " scalar, " << p.n << " batched units\n";
    return 0;
}
