// merl_host_table.hpp — the host-side image of one table (or RGL) material: what the one-unit calls on the CPU
// (merl_host_scalar.hip) evaluate.  Built by mrl_material_host_table (merl_abi.hip) from the resident device table,
// immutable afterwards, reference-counted; a snapshot of the context's lookup options travels with it.
#pragma once
#include <atomic>
#include <vector>

#include "merl_device.hpp"
#include "merl_rgl.hpp"

struct mrl_host_table {
    std::atomic<int> refs{ 1 };
    mrl::MaterialDev m;                 // texels -> rows.data() (LAYOUT_ROWS), sampling -> marginal.data()
    mrl::Options opts;                  // lookup / node / disk map / sampling of the context when the image was taken
    std::vector<float4> rows;           // [(n_th+1)][(n_td+1)][(n_pd+1)] RGBA f32: the device's own texel values
    std::vector<double> marginal;       // s | cdf | c (table importance sampling)
    std::vector<double> marginal2d;     // the conditional rows P(theta_h | theta_i), as the device built them
    // KIND_RGL (m.kind says so): the device image as it is (grids, five functions, running integrals) and its descriptor
    std::vector<float> rgl_image;
    mrl::RglDev rgl{};
};
