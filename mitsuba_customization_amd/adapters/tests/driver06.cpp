// driver06.cpp — stands in for Mitsuba 0.6's PluginManager + an integrator: dlopen()s a plugin,
// resolves CreateInstance / GetDescription, builds the BSDF from Properties and calls it the way
// MIPathTracer does (scalar virtual calls) and the way a wavefront host would (BatchedBSDF).
//   driver06 <plugin.so> <table.binary> <pairs.bin> <out.bin> <n_scalar> [interpolation] [scaleR scaleG scaleB]
//   driver06 --expect-no-device <plugin.so> <table.binary>
//   driver06 <plugin.so> <table.binary> --residency <other_table.binary> [cycles]
#include <dlfcn.h>

#include <chrono>
#include <cstring>
#include <iostream>
#include <thread>

#include <mitsuba/mitsuba.h>

#include "../common/batched_bsdf.hpp"
#include "driver_common.hpp"

using namespace mitsuba;
typedef void *(*CreateInstanceFn)(const Properties &);
typedef const char *(*GetDescriptionFn)();

int main(int argc, char **argv)
{
    bool expect_no_device = argc > 1 && std::strcmp(argv[1], "--expect-no-device") == 0;
    if (expect_no_device) { --argc; ++argv; }
    if (argc < 3) { std::cerr << "usage\n"; return 2; }
    void *h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { std::cerr << "dlopen: " << dlerror() << "\n"; return 3; }
    auto create = (CreateInstanceFn)dlsym(h, "CreateInstance");
    auto descr = (GetDescriptionFn)dlsym(h, "GetDescription");
    if (!create || !descr) { std::cerr << "plugin lacks CreateInstance/GetDescription\n"; return 3; }
    std::cout << "plugin: " << descr() << "\n";

    Properties props("bsdf");
    props.setString("filename", argv[2]);
    if (expect_no_device) {
        try {
            create(props);
        } catch (const std::exception &e) {
            std::cout << "constructor threw: " << e.what() << "\n";
            return std::strstr(e.what(), "no CPU fallback") ? 0 : 4;
        }
        std::cerr << "constructor succeeded without a GPU?\n";
        return 4;
    }
    if (argc >= 4 && std::strcmp(argv[3], "--residency") == 0) {
        if (argc < 5) { std::cerr << "usage\n"; return 2; }
        const int cycles = argc > 5 ? atoi(argv[5]) : 50;
        try {
            return check_residency([&](int which) {
                                       Properties q("bsdf");
                                       q.setString("filename", which ? argv[4] : argv[2]);
                                       BSDF *b = static_cast<BSDF *>(create(q)); b->incRef(); b->configure(); return b;
                                   },
                                   [](BSDF *b) { b->decRef(); }, cycles);
        } catch (const std::exception &e) { std::cerr << "residency: " << e.what() << "\n"; return 25; }
    }
    if (argc < 6) { std::cerr << "usage\n"; return 2; }
    if (argc > 6) props.setString("interpolation", argv[6]);
    if (argc > 9) { props.setFloat("scaleR", (Float)atof(argv[7])); props.setFloat("scaleG", (Float)atof(argv[8])); props.setFloat("scaleB", (Float)atof(argv[9])); }

    if (argc > 10) props.setString("sampling", argv[10]);
    if (argc > 11) props.setString("parameterization", argv[11]);
    // where the scalar virtual calls evaluate: MERL_DRIVER_SCALAR = cpu | gpu (unset: the plugin's default, cpu)
    if (const char *sc = std::getenv("MERL_DRIVER_SCALAR")) props.setString("scalar", sc);
    // further string properties: MERL_DRIVER_PROPS = name=value[,name=value ...]  (e.g. cosine_factor=omitted,negative_values=keep)
    if (const char *more = std::getenv("MERL_DRIVER_PROPS")) {
        std::string all(more);
        size_t at = 0;
        while (at < all.size()) {
            size_t end = all.find(',', at);
            if (end == std::string::npos) end = all.size();
            const std::string item = all.substr(at, end - at);
            const size_t eq = item.find('=');
            if (eq != std::string::npos) props.setString(item.substr(0, eq), item.substr(eq + 1));
            at = end + 1;
        }
    }
    BSDF *bsdf = nullptr;
    try {
        bsdf = static_cast<BSDF *>(create(props));
        bsdf->incRef();
        bsdf->configure();
    } catch (const std::exception &e) {
        std::cerr << "constructor threw: " << e.what() << "\n";
        return 5;
    }
    std::cout << bsdf->toString() << "\n";
    if (bsdf->getComponentCount() != 1 || !(bsdf->getType() & BSDF::EGlossyReflection) || !(bsdf->getType() & BSDF::EFrontSide)) return 6;

    Pairs p = read_pairs(argv[3]);
    const size_t m = std::min<size_t>(p.n, (size_t)atoll(argv[5]));
    std::vector<float> scalar(11 * m), batch(11 * p.n);
    Intersection its;
    const auto t_single = std::chrono::steady_clock::now();
    for (size_t i = 0; i < m; ++i) {
        its.wi = Vector(p.wi[3 * i], p.wi[3 * i + 1], p.wi[3 * i + 2]);
        Vector wo(p.wo[3 * i], p.wo[3 * i + 1], p.wo[3 * i + 2]);
        BSDFSamplingRecord q(its, wo);
        Spectrum f = bsdf->eval(q, ESolidAngle);
        Float pdf = bsdf->pdf(q, ESolidAngle);
        BSDFSamplingRecord s(its);
        Float spdf;
        Spectrum w = bsdf->sample(s, spdf, Point2(p.u[2 * i], p.u[2 * i + 1]));
        float *o = &scalar[11 * i];
        o[0] = f[0]; o[1] = f[1]; o[2] = f[2]; o[3] = pdf;
        o[4] = s.wo.x; o[5] = s.wo.y; o[6] = s.wo.z; o[7] = spdf; o[8] = w[0]; o[9] = w[1]; o[10] = w[2];
        if (spdf > 0 && (s.eta != 1.0f || s.sampledComponent != 0 || s.sampledType != BSDF::EGlossyReflection)) return 7;
        // the two sample() overloads agree
        BSDFSamplingRecord s2(its);
        Spectrum w2 = bsdf->sample(s2, Point2(p.u[2 * i], p.u[2 * i + 1]));
        if (w2[0] != w[0] || s2.wo.x != s.wo.x) return 7;
    }
    const double us_single = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_single).count() / (4.0 * m);
    // guards: wrong measure / masked lobe -> zero, without touching the GPU result
    {
        its.wi = Vector(0.f, 0.6f, 0.8f);
        BSDFSamplingRecord q(its, Vector(0.6f, 0.f, 0.8f));
        if (!bsdf->eval(q, EDiscrete).isZero() || bsdf->pdf(q, EDiscrete) != 0.f) return 8;
        q.typeMask = BSDF::EDiffuseReflection;
        if (!bsdf->eval(q, ESolidAngle).isZero()) return 8;
    }
    // the renderer calls a const BSDF from all of its render threads at once
    {
        const unsigned T = 16;
        const auto t_threads = std::chrono::steady_clock::now();
        std::vector<std::vector<float>> per_thread(T);
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < T; ++t)
            pool.emplace_back([&, t]() {
                Intersection its_t;
                for (size_t i = t; i < m; i += T) {
                    its_t.wi = Vector(p.wi[3 * i], p.wi[3 * i + 1], p.wi[3 * i + 2]);
                    BSDFSamplingRecord q(its_t, Vector(p.wo[3 * i], p.wo[3 * i + 1], p.wo[3 * i + 2]));
                    Spectrum f = bsdf->eval(q, ESolidAngle);
                    BSDFSamplingRecord s(its_t);
                    Float spdf;
                    Spectrum w = bsdf->sample(s, spdf, Point2(p.u[2 * i], p.u[2 * i + 1]));
                    per_thread[t].insert(per_thread[t].end(), { f[0], f[1], f[2], s.wo.x, s.wo.y, s.wo.z, spdf, w[0], w[1], w[2] });
                }
            });
        for (auto &th : pool) th.join();
        const double us_threads = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_threads).count() / (2.0 * m);
        std::cout << "scalar calls: " << us_single << " us/call from one thread, " << us_threads << " us/call amortised over " << T
                  << " threads (scalar=" << (std::getenv("MERL_DRIVER_SCALAR") ? std::getenv("MERL_DRIVER_SCALAR") : "cpu") << ")\n";
        for (unsigned t = 0; t < T; ++t) {
            size_t k = 0;
            for (size_t i = t; i < m; i += T, ++k) {
                const float *o = &scalar[11 * i], *g = &per_thread[t][10 * k];
                const float want[10] = { o[0], o[1], o[2], o[4], o[5], o[6], o[7], o[8], o[9], o[10] };
                if (std::memcmp(g, want, sizeof want) != 0) { std::cerr << "threaded scalar call differs at unit " << i << "\n"; return 10; }
            }
        }
    }
    // network rendering: serialize() on the master, the unserialising constructor on a worker -> same answers
    {
        typedef void *(*UnserializeFn)(Stream *, InstanceManager *);
        auto unserialize = (UnserializeFn)dlsym(h, "UnserializeInstance");
        if (!unserialize) { std::cerr << "plugin lacks UnserializeInstance\n"; return 12; }
        Stream stream;
        bsdf->serialize(&stream, nullptr);
        BSDF *copy = nullptr;
        try { copy = static_cast<BSDF *>(unserialize(&stream, nullptr)); } catch (const std::exception &e) { std::cerr << "unserialize threw: " << e.what() << "\n"; return 12; }
        copy->incRef();
        if (stream.getPos() != stream.getSize() || copy->getType() != bsdf->getType() || copy->getComponentCount() != 1) {
            std::cerr << "unserialised copy: stream not consumed or type differs: " << copy->toString() << "\n";      // (its material id is a new one)
            return 12;
        }
        for (size_t i = 0; i < std::min<size_t>(m, 50); ++i) {
            its.wi = Vector(p.wi[3 * i], p.wi[3 * i + 1], p.wi[3 * i + 2]);
            BSDFSamplingRecord q(its, Vector(p.wo[3 * i], p.wo[3 * i + 1], p.wo[3 * i + 2]));
            const Spectrum f = copy->eval(q, ESolidAngle);
            const float got[3] = { f[0], f[1], f[2] };
            if (std::memcmp(got, &scalar[11 * i], sizeof got) != 0) { std::cerr << "unserialised copy differs at unit " << i << "\n"; return 12; }
        }
        copy->decRef();
        Stream garbage;
        garbage.writeInt(1); garbage.writeString("x.binary"); garbage.writeInt(0); garbage.writeInt(7);     // truncated + bad option
        bool threw = false;
        try { unserialize(&garbage, nullptr); } catch (const std::exception &) { threw = true; }
        if (!threw) return 12;
    }
    const BatchedBSDF *wave = dynamic_cast<const BatchedBSDF *>(bsdf);
    if (!wave) { std::cerr << "plugin is not a BatchedBSDF\n"; return 9; }
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
    bsdf->decRef();
    std::cout << "driver06 ok: " << m << " scalar, " << p.n << " batched units\n";
    return 0;
}
