// merl_kernels.hpp — launch interface between the C-ABI layer and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "merl_device.hpp"

namespace mrl {

// Kernel arguments of one batched call; pointers are device-accessible.
struct BatchArgs {
    const float *wi, *wo, *u;
    const int32_t *mat;              // nullptr: single
    size_t n;
    float *out_rgb, *out_pdf, *out_wo, *out_pdf2, *out_weight;
    MaterialDev single;              // by value -> SGPRs (single-material launches)
    const MaterialDev *materials;    // device array (mixed-material launches)
    int n_materials;
    Options opts;
};

// mode: 0 eval, 1 pdf, 2 sample, 3 eval+sample
// variant: MRL_OPT_KERNEL (0 generic, 1 tuned table path, 2 tuned + non-temporal streams)
// layout: the context-wide table layout (every table of a context has the same one)
// has_ggx: the context holds at least one analytic (GGX) material, so a mixed batch may contain such lanes
hipError_t launch_batch(int mode, const BatchArgs &a, bool multi, int variant, int layout, bool has_ggx, int compute_units, hipStream_t stream);
hipError_t launch_generate_pairs(uint64_t seed, uint64_t first, size_t n, float *wi, float *wo, float *u,
                                 int compute_units, hipStream_t stream);
hipError_t launch_generate_materials(uint64_t seed, uint64_t first, size_t n, int n_materials, int32_t *mat,
                                     int compute_units, hipStream_t stream);

} // namespace mrl
