// merl_kernels.hpp — launch interface between the C-ABI layer and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <vector>

#include "merl_device.hpp"
#include "merl_image_file.hpp"       // RglFields / RglLayout, rgl_plan_layout, nch_brick_float4s: shapes -> sizes, pure host

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
    // what a unit evaluates when its material id names nothing this call can evaluate (out of range, released,
    // another kind such as an n-channel table): a 1 x 1 x 1 table of zeros in valid device memory.  Its outputs are
    // forced to zero anyway, but its lookups run (the code is branch-free) and must touch memory that exists — an
    // n-channel table's cells are narrower than the 128-B bricks these kernels read.
    MaterialDev safe;
    int any_standard;                // some table this launch may meet is in a standard parameterisation (MaterialDev::param != 0)
    int block_map;                   // k_table_dma: 0 interleaved grid-stride tiles, 1 one contiguous eighth of the batch per XCD
    Options opts;
    // queue launches: a queue of unit indices (a caller's wavefront queue, or one kind's queue built by k_partition_kinds)
    const uint32_t *idx;
    const uint32_t *idx_count;               // its length, in device memory
};

// mode: 0 eval, 1 pdf, 2 sample, 3 eval+sample, 4 eval+pdf
// variant: MRL_OPT_KERNEL (0 generic, 1 tuned table path, 2 + non-temporal streams, 3 + LDS-DMA brick fetch; 4 is handled by the caller)
// layout: the context-wide table layout (every table of a context has the same one)
// has_ggx / has_table: the context holds at least one analytic (GGX) / one table material, i.e. what a mixed batch may contain
hipError_t launch_batch(int mode, const BatchArgs &a, bool multi, int variant, int layout, bool has_ggx, bool has_table, int compute_units, hipStream_t stream);
// kind-partitioned mixed batches (MRL_OPT_KERNEL >= 4): build the two queues, then run one of them
void partition_geometry(size_t n, int compute_units, uint32_t *segments, uint32_t *seg_len);
// work: 4*segments + 2 uint32 (counts, offsets, totals[2] at work + 4*segments)
hipError_t launch_partition_kinds(const int32_t *mat, size_t n, const MaterialDev *materials, int n_materials,
                                  uint32_t *queue_table, uint32_t *queue_ggx, uint32_t *work,
                                  uint32_t segments, uint32_t seg_len, hipStream_t stream);
hipError_t launch_batch_queue(int mode, const BatchArgs &a, bool ggx_queue, int compute_units, hipStream_t stream);
// a caller's wavefront queue (mrl_*_queue): units a.idx[0 .. min(*a.idx_count, a.n)), a.n = the queue's capacity
hipError_t launch_batch_indexed(int mode, const BatchArgs &a, bool multi, int layout, bool has_ggx, bool has_table,
                                int compute_units, hipStream_t stream);
// per-material compaction for wavefront callers (mrl_partition_by_material): stable partition of [0, n) by material id
constexpr int kMaxPartitionMaterials = 2048;
void material_partition_geometry(size_t n, int compute_units, uint32_t *chunks, uint32_t *chunk_len);
// work: chunks*K + K uint32; queue: n uint32; offsets: K + 1; counts: K (all device)
hipError_t launch_partition_materials(const int32_t *mat, size_t n, int K, uint32_t *queue, uint32_t *offsets, uint32_t *counts,
                                      uint32_t *work, uint32_t chunks, uint32_t chunk_len, int compute_units, hipStream_t stream);
// a1: planar f64 table (device copy of the file payload) -> padded rows or bricks; clamp: negative values become 0 (MRL_OPT_NEGATIVE = 0)
hipError_t launch_build_table(const double *d_planar, const int dims[3], const double scale[3], int layout, int param, int clamp, float4 *d_out,
                              int compute_units, hipStream_t stream);
// one RGB table layout to the other (the on-disk image cache stores the rows form): padded rows <-> bricks, same Float values
hipError_t launch_rows_to_bricks(const float4 *d_rows, const int dims[3], float4 *d_bricks, int compute_units, hipStream_t stream);
hipError_t launch_bricks_to_rows(const float4 *d_bricks, const int dims[3], int param, float4 *d_rows, int compute_units, hipStream_t stream);
// the conditional sampling table P(theta_h | theta_i) of a resident RGB table (MRL_OPT_SAMPLING = 2): a quadrature kernel
// and a prefix-scan kernel; d_rows: n_ti x (2 n_th + 1) doubles, d_work: n_ti x n_th doubles
constexpr int kSamplingIncidentBins = 32;
hipError_t launch_build_sampling2d(const MaterialDev &m, const Options &opts, int n_ti, double *d_rows, double *d_work, hipStream_t stream);
// ---- n-channel tables (merl_nch.hip): a.out_rgb / a.out_weight hold n x n_ch values ----
constexpr int kMaxChannels = 32;
// nch_brick_float4s(n_ch): float4s per cell: 2 (1 ch), 4 (2 ch), 8 * ceil(n_ch / 4)  (merl_image_file.hpp)
// mode: 0 eval, 2 sample, 3 eval+sample, 4 eval+pdf (pdf alone: the RGB pdf kernel serves every table kind)
hipError_t launch_batch_nch(int mode, const BatchArgs &a, bool multi, int n_ch, int compute_units, hipStream_t stream);
hipError_t launch_build_table_nch(const double *d_planar, const double *d_scale, const int dims[3], int n_ch, int param, int clamp, float4 *d_out,
                                  int compute_units, hipStream_t stream);
// ---- the adaptive-parameterisation measured BSDF (merl_rgl.hip; RGL *.bsdf) ----
struct RglDev;
int rgl_reduction(const RglFields &f);                                   // 1, 2, 4: the part of the azimuth an anisotropic file stores
const char *rgl_check_fields(const RglFields &f);                        // nullptr, or what is wrong
RglLayout rgl_build_image(const RglFields &f, std::vector<float> &blob); // normalised tables + running integrals, host f64
RglDev rgl_descriptor(const RglFields &f, const RglLayout &l, const float *base);
// r != nullptr: a single-material launch; r == nullptr: a batch with material ids (a.mat) — the units whose id names an RGL material
// are evaluated and written, every other unit is left as it is.  indexed: walk the queue a.idx / a.idx_count.
// search (MRL_OPT_RGL_SEARCH): 0 = a single-material launch reads the distributions' search tables from a copy in LDS when they fit
// a CU's LDS, 1 = always from memory (the results are the same bits)
hipError_t launch_rgl(int mode, const BatchArgs &a, const RglDev *r, bool indexed, int search, int compute_units, hipStream_t stream);
// a spectral RGL material: a.out_rgb / a.out_weight hold n x W values at the per-unit wavelengths wl [n][W] (nullptr: the file's own
// wavelength nodes, W = their number); single material, whole arrays
hipError_t launch_rgl_spectral(int mode, const BatchArgs &a, const RglDev &r, const float *wl, int W, int search, int compute_units, hipStream_t stream);
// ---- one-unit calls (merl_scalar.hip): a bounded-lifetime service kernel answers requests posted in pinned host memory ----
struct ScalarBoard;
struct ScalarArgs {
    const MaterialDev *materials;    // the context's material array (the host stops the service before it changes)
    int n_materials;
    MaterialDev safe;
    Options opts;
    ScalarBoard *board;              // pinned, coherent host memory (device-visible address)
    uint32_t gen;                    // this instance's generation: written to board->started_gen / exited_gen
    uint32_t max_polls;              // hard bound on the poll loop, whatever the clock says
    uint64_t lifetime_ticks;         // of wall_clock64() (100 MHz)
};
hipError_t launch_scalar_service(const ScalarArgs &a, hipStream_t stream);
hipError_t launch_generate_pairs(uint64_t seed, uint64_t first, size_t n, float *wi, float *wo, float *u,
                                 int compute_units, hipStream_t stream);
hipError_t launch_generate_materials(uint64_t seed, uint64_t first, size_t n, int n_materials, int32_t *mat,
                                     int compute_units, hipStream_t stream);

} // namespace mrl
