// driver_common.hpp — shared bits of the two plugin test drivers (binary I/O with pytest).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

struct Pairs {
    uint64_t n = 0;
    std::vector<float> wi, wo, u;
};

// file: uint64 n, then wi[n*3], wo[n*3], u[n*2] as f32
inline Pairs read_pairs(const std::string &path)
{
    Pairs p;
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) { std::perror(path.c_str()); std::exit(2); }
    if (std::fread(&p.n, sizeof p.n, 1, f) != 1) std::exit(2);
    p.wi.resize(3 * p.n); p.wo.resize(3 * p.n); p.u.resize(2 * p.n);
    if (std::fread(p.wi.data(), 4, 3 * p.n, f) != 3 * p.n || std::fread(p.wo.data(), 4, 3 * p.n, f) != 3 * p.n ||
        std::fread(p.u.data(), 4, 2 * p.n, f) != 2 * p.n)
        std::exit(2);
    std::fclose(f);
    return p;
}

// file: per unit rgb[3] pdf wo2[3] pdf2 weight[3]  (11 floats), first the scalar-call block (m units), then the batch block (n units)
inline void write_floats(FILE *f, const std::vector<float> &v) { std::fwrite(v.data(), 4, v.size(), f); }
