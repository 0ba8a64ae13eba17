/*
 * merl_hip.h — C ABI of libmerl_hip.so: the MI355X (gfx950) implementation of the
 * MERL / customized_measurement BSDF eval()/sample()/pdf() hot path.
 *
 * This is the drop-in boundary.  The reference's own interface for the path is
 * the Mitsuba BSDF plugin class — /root/reference/README.md:1 names the plugins ("Merl,
 * customized_measurment brdf pluggin for Mitsuba 0.6 Mitsuba 3.0"); the sources that would
 * carry file:line (mitsuba/src/bsdfs/<plugin>.cpp, mitsuba3/src/bsdfs/<plugin>.cpp) are empty gitlinks in the
 * snapshot, so each entry point below cites the public upstream method it stands in for:
 *
 *   mrl_material_load_merl / _upload_f64   plugin constructor (Properties "filename" -> table load)
 *   mrl_material_upload_table / _load_table  customized_measurement constructor (free dims/scales; MRL_OPT_TABLE_PARAM
 *                                          = the plugin's "parameterization" property, mrl_material_param reads it back)
 *   mrl_material_ggx                       upstream roughconductor constructor (BASELINE config 3)
 *   mrl_eval_batch                         BSDF::eval(bRec, ESolidAngle)   / M3 BSDF::eval
 *   mrl_pdf_batch                          BSDF::pdf(bRec, ESolidAngle)    / M3 BSDF::pdf
 *   mrl_sample_batch                       BSDF::sample(bRec, pdf, sample) / M3 BSDF::sample
 *   mrl_eval_pdf_batch                     M3 BSDF::eval_pdf (one table lookup for both; what a light-sampling
 *                                          integrator calls for its MIS weight)
 *   mrl_eval_sample_batch                  the fused eval + pdf + sample unit (BASELINE metric)
 *   mrl_scalar_eval_sample                 ONE virtual BSDF::eval / sample / pdf call of a per-ray integrator
 *   mrl_*_queue                            the same calls over a wavefront integrator's material queue
 *                                          (the caller side of the path)
 *   mrl_material_release / mrl_memory_info plugin destructor (the host drops its last ref<BSDF>) / no counterpart
 *   mrl_material_*_nch, mrl_*_batch_nch,   customized_measurement tables with 1..32 channels (the
 *   mrl_*_queue_nch                        reference's own table format is unknown, Appendix B item 7)
 *   mrl_tensor_file_*,                     the "tensor_file" container that upstream Mitsuba 3's `measured` plugin reads
 *   mrl_material_load_tensor_table         (src/bsdfs/measured.cpp + src/core/tensor.cpp upstream; not in the snapshot)
 *   mrl_group_*                            no counterpart in the reference (it has no multi-device path): the
 *                                          data-parallel outer loop of SURVEY.md §8b/§8e, `mrl_init(n_devices, device_ids, …)`
 *
 * Conventions: every function returns 0 on success or a negative mrl_status; no exception
 * crosses the boundary.  All arrays are f32, direction arrays are xyzxyz… (n x 3), sample
 * arrays uvuv… (n x 2), directions live in the local shading frame (z = normal).  Pointers
 * may be device pointers (any allocation of the same HIP runtime, e.g. a torch tensor's
 * data_ptr) or host pointers; all pointers of one call must be of the same kind.  Device
 * pointer calls are asynchronous on the context's stream; host pointer calls return when the
 * outputs are written.  The caller owns every buffer; the context owns tables and streams.
 * A context is thread-safe: every entry point takes the context's lock, so calls from several host threads serialise
 * (device-pointer calls only enqueue — microseconds; host-array calls hold the lock for their duration).  What stays
 * the caller's business is ordering: mrl_set_stream / mrl_timer_* / mrl_last_error describe "the last call", whoever
 * made it, and a material must not be released while another thread still passes its id.
 *
 * There is NO CPU fallback: without a usable gfx950 device mrl_init fails.
 */
#ifndef MERL_HIP_H
#define MERL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mrl_ctx mrl_ctx;

enum mrl_status {
    MRL_OK = 0,
    MRL_ERR_INVALID = -1,      /* bad argument */
    MRL_ERR_HIP = -2,          /* a HIP runtime call failed (see mrl_last_error) */
    MRL_ERR_IO = -3,           /* file could not be opened / read */
    MRL_ERR_FORMAT = -4,       /* file is not a MERL / table file of the expected dims */
    MRL_ERR_OOM = -5,
    MRL_ERR_MATERIAL = -6,     /* unknown material id */
    MRL_ERR_POINTER_MIX = -7,  /* host and device pointers mixed in one call */
    MRL_ERR_NO_DEVICE = -8,    /* no gfx950 device: there is no CPU fallback */
    MRL_ERR_COMM = -9          /* an RCCL call failed or librccl could not be loaded (see mrl_group_last_error) */
};

/* Which three angles index a customized_measurement table ("dims/parameterisation"); axis order
 * in the file stays (0, 1, 2) = the order listed here, axis 2 fastest.
 *   HALF_DIFF      theta_h (sqrt-warped), theta_d, phi_d mod pi — the MERL / Rusinkiewicz form (default);
 *   STANDARD       theta_i, theta_o, |phi_o - phi_i| in [0, pi] — an isotropic, bilaterally symmetric gonioreflectometer grid;
 *                  every axis linear in its angle, the azimuth axis clamped (0 and pi are its two ends);
 *   STANDARD_FULL  theta_i, theta_o, (phi_o - phi_i) mod 2 pi — isotropic without the mirror symmetry; azimuth periodic.
 * x = angle / range * n along each axis; nearest lookups truncate, trilinear lookups follow MRL_OPT_NODE, exactly as for
 * MERL tables.  Table importance sampling (MRL_OPT_SAMPLING = 1) has no theta_h rows to learn from on the standard forms:
 * its half-vector lobe is flat there (p_h = cos(theta_h) / pi) — valid, not variance-reducing. */
enum mrl_param { MRL_PARAM_HALF_DIFF = 0, MRL_PARAM_STANDARD = 1, MRL_PARAM_STANDARD_FULL = 2 };

enum mrl_negative { MRL_NEGATIVE_CLAMP = 0, MRL_NEGATIVE_KEEP = 1, MRL_NEGATIVE_RENORMALISE = 2 };
enum mrl_option {
    MRL_OPT_LOOKUP = 0,        /* 0 nearest (BRDFRead), 1 trilinear (default) */
    MRL_OPT_NODE = 1,          /* trilinear node position: 0 integer coordinate (default), 1 texel centre */
    MRL_OPT_DISK_MAP = 2,      /* concentric disk flavour: 0 Mitsuba 0.6 (default), 1 Mitsuba 3 */
    MRL_OPT_KERNEL = 3,        /* implementation variant of the kernels: 0 generic, 1 tuned math,
                                  2 + non-temporal streams, 3 (default) + cooperative LDS-DMA brick fetch,
                                  4 = 3 + ballot/prefix partition of a batch that mixes table and analytic
                                  materials into one dense queue per kind (pays only when most units are analytic).
                                  Every variant passes the same parity tests; they differ in speed only.  Values
                                  outside 0..4 are rejected with MRL_ERR_INVALID. */
    MRL_OPT_HOST_CHUNK = 4,    /* units per staging chunk for host-pointer calls (the pipelined path uses at most 2^20) */
    MRL_OPT_SAMPLING = 6,      /* (2 = the conditional table P(theta_h | theta_i) with the cosine lobe at weight 1/8: see
                                  mrl_material_sampling2d)
                                  sample()/pdf() strategy of table materials: 0 cosine hemisphere (default, the upstream
                                  convention), 1 table importance sampling: one-sample mixture of the cosine lobe and a
                                  half-vector lobe read off the table's theta_h rows */
    MRL_OPT_TABLE_LAYOUT = 5,  /* HBM layout of the context's tables, settable only while it holds no table:
                                  0 padded rows (24 MB per MERL table),
                                  1 bricks (default): one 128-B line per cell holds its 8 corners (187 MB per MERL table).
                                  Bricks are 2.3x faster for trilinear lookups, rows 1.45x faster for nearest lookups. */
    MRL_OPT_HOST_THREADS = 8,  /* host-pointer calls on plain (pageable) arrays: threads that copy between the caller's arrays
                                  and two pinned, device-mapped chunk buffers while the kernel of the previous / next chunk
                                  reads and writes those buffers over PCIe (default 4, counting the calling thread;
                                  0 = the staged hipMemcpy path of round 1, ~3x slower) */
    MRL_OPT_BLOCK_MAP = 9,     /* how the LDS-DMA kernel's workgroups walk a batch: 0 interleaved (block b takes tiles b, b + G, ...),
                                  1 XCD-contiguous: the workgroups of one XCD (its own 4 MB L2) walk one contiguous eighth of
                                  the batch, so neighbouring units — pixels and scanlines of a render — share one L2 instead
                                  of being fetched by all eight.  Same results either way; see DESIGN.md §6 for which is faster when */
    MRL_OPT_TABLE_PARAM = 10,  /* parameterisation of the customized_measurement tables uploaded FROM NOW ON (enum mrl_param below;
                                  recorded per material, so one context can hold tables of all three; MERL files are always
                                  half/diff).  Same kernels, same HBM layouts: only the three lookup angles differ. */
    MRL_OPT_TABLE_ARENA_MB = 11, /* one device allocation of this many MiB that the RGB tables uploaded from now on are placed in back
                                  to back (2 MiB aligned) while it has room, instead of one allocation per table; a table
                                  that does not fit gets its own.  Released tables return their space when the arena
                                  empties.  Settable while no table lives in it; 0 frees it.  The arena is asked for as
                                  physically contiguous memory (plain device memory if the driver refuses).  Measured effect
                                  on the 100-table launch: DESIGN.md §6 (address translation bounds that launch). */
    MRL_OPT_RGL_SEARCH = 12,   /* where a single-material launch on an RGL material reads the two distributions' running integrals
                                  (conditional / marginal, what sample()'s searches walk): 0 (default) a copy in the CU's LDS when
                                  they fit (129 KB for the database's isotropic 8 x 32 x 32 shape; the marginal rows alone for
                                  sample() on larger files), 1 always the image in memory.  Same results bit for bit; the option
                                  exists so that both paths can be measured and tested. */
    MRL_OPT_COSINE_FACTOR = 13, /* SURVEY.md Appendix B 4 — does the plugin's eval() multiply the BRDF by cos(theta_o)?  0 (default,
                                  upstream Mitsuba's convention): eval() = f cos(theta_o); 1: eval() = f alone — from eval() and from
                                  the eval() inside sample()'s weight (weight == eval / pdf stays true).  Table materials (MERL,
                                  customized_measurement, n-channel); GGX and RGL materials follow their upstream plugins, whose
                                  convention is known.  May be changed at any time (it is applied per call). */
    MRL_OPT_NEGATIVE = 14,     /* SURVEY.md Appendix B 2 — what a negative stored value (MERL's marker for a sample that was not
                                  measured) does to a lookup (enum mrl_negative): 0 MRL_NEGATIVE_CLAMP (default) it counts as 0;
                                  1 MRL_NEGATIVE_KEEP it is used as stored (what BRDFRead itself does — it only prints a warning):
                                  eval() can come out negative; 2 MRL_NEGATIVE_RENORMALISE it is left out: a trilinear lookup blends
                                  the valid corners only and divides by their weight, per channel (0 when no corner is valid), a
                                  nearest lookup returns 0.  Clamping happens when a table's image is built, so 0 <-> {1, 2} can only
                                  be switched while the context holds no table (1 <-> 2 at any time); the sampling marginals
                                  (MRL_OPT_SAMPLING) are built from clamped values under every setting.  With 2 the batch calls run the
                                  generic kernel for nearest lookups and rows-layout tables and the LDS-DMA kernel for trilinear lookups
                                  on bricks, whatever MRL_OPT_KERNEL says. */
    MRL_OPT_RESERVED_CUS = 15, /* compute units the batch kernels leave alone (0 .. 16; default 0).  The kernels are
                                  persistent grids sized to fill every CU; a communication library whose transfers are KERNELS (RCCL's
                                  ncclSend / ncclRecv) then finds no CU to run on until a grid drains.  With k > 0 the context's stream
                                  becomes a stream with a CU mask (hipExtStreamCreateWithCUMask: k CUs, spread evenly, are excluded)
                                  and the grids are sized for the rest.  mrl_group_set_option applies it to every member's compute
                                  stream; the transfer streams stay unrestricted.  Results do not depend on it.  Verified with hardware
                                  ids: k = 1, 8, 16 idle exactly k CUs; larger masks are dropped by the driver, hence the range
                                  (profiles/r04_cu_mask_probe.json).  Cost on one device: DESIGN.md §7 (profiles/r04_reserved_cus.json). */
    MRL_OPT_MEMORY_LIMIT_MB = 7 /* budget for the context's resident material data (tables + sampling marginals), in MiB;
                                  0 (default) = no budget, the device's free memory is the limit.  An upload that would
                                  exceed the budget — or the device — fails with MRL_ERR_OOM and leaves the context as it
                                  was.  Capacity for scale: one MERL table is 186.6 MB as bricks (24.0 MB as rows), so a
                                  288 GB MI355X holds about 1,500 brick tables (11,000 as rows). */
};

enum mrl_material_kind { MRL_KIND_MERL = 0, MRL_KIND_TABLE = 1, MRL_KIND_GGX = 2, MRL_KIND_RELEASED = 3 /* tombstone, never reported */,
                         MRL_KIND_TABLE_NCH = 4 /* n-channel table: evaluated by the *_nch entry points only */,
                         MRL_KIND_RGL = 5 /* adaptive-parameterisation measured BSDF (mrl_material_upload_rgl) */,
                         MRL_KIND_RGL_SPECTRAL = 6 /* the same from a spectral file: evaluated by the mrl_*_spectral_batch entry points */ };

/* ---- context ---- */
int mrl_init(int device_id, mrl_ctx **out);
int mrl_destroy(mrl_ctx *ctx);
const char *mrl_strerror(int status);
/* "sources <12 hex digits>": a hash over the library's source files, fixed at build time.  Committed counter measurements
 * (profiles/traffic.json) carry it; bench.py marks them stale when the library it runs reports another one. */
const char *mrl_build_info(void);
const char *mrl_last_error(const mrl_ctx *ctx);
int mrl_set_option(mrl_ctx *ctx, int option, int value);
int mrl_get_option(const mrl_ctx *ctx, int option, int *value);
/* launch on a caller-owned hipStream_t (e.g. torch's current stream); NULL is HIP's default
 * (null) stream, which is what torch uses unless a stream context is active */
int mrl_set_stream(mrl_ctx *ctx, void *hip_stream);
/* back to the context's own non-blocking stream (the state after mrl_init) */
int mrl_reset_stream(mrl_ctx *ctx);
int mrl_synchronize(mrl_ctx *ctx);
int mrl_device_info(const mrl_ctx *ctx, char *name, size_t name_len, int *compute_units, size_t *total_mem);

/* ---- materials (immutable after creation; ids are dense, starting at 0) ---- */
int mrl_material_load_merl(mrl_ctx *ctx, const char *path, int *out_id);
/* planar R,G,B doubles in MERL order, 3 x 90*90*180 raw file values (scales applied inside) */
int mrl_material_upload_f64(mrl_ctx *ctx, const double *planar_rgb, int *out_id);
/* customized_measurement: MERL parameterisation with free dims and channel scales */
int mrl_material_upload_table(mrl_ctx *ctx, const double *planar_rgb, const int dims[3],
                              const double scale[3], int *out_id);
/* file: int32 dims[3], then planar R,G,B values in MERL order, as f64 (like a MERL file) or f32 (told apart by the
 * file length) */
int mrl_material_load_table(mrl_ctx *ctx, const char *path, const double scale[3], int *out_id);
int mrl_material_ggx(mrl_ctx *ctx, float alpha, const float eta[3], const float k[3], int *out_id);
/* The adaptive-parameterisation measured BSDF of the RGL material database's *.bsdf files (Dupuy & Jakob 2018; what upstream
 * Mitsuba 3's stock `measured` plugin evaluates — its eval / sample / pdf are the interface this replaces; the reference
 * snapshot holds neither that plugin nor a file: PARITY UNPINNED, restated from the published model, oracle/rgl_oracle.c).
 * Arrays are the file's Float fields as they are, x (the last axis) fastest; res[] = { nodes along x, nodes along y }:
 *     phi_i [n_phi], theta_i [n_theta]         incident-direction grids (strictly ascending; n_phi <= 2 means isotropic; an anisotropic
 *                                              file spans the whole azimuth, or [-pi, 0] / [-pi, -pi/2] for a sample with a point
 *                                              symmetry / two mirror planes: pairs are mapped into the stored part by the signs of wi)
 *     ndf [res_ndf[1]][res_ndf[0]], sigma [res_sigma[1]][res_sigma[0]]
 *     vndf, luminance [n_phi][n_theta][res[1]][res[0]]          rgb [n_phi][n_theta][3][res[1]][res[0]]
 * The library normalises vndf / luminance per slice and builds their running integrals once (host, f64), then keeps one
 * image in HBM.  Every entry point of the RGB family evaluates it (eval, pdf, sample, eval_sample, eval_pdf; whole-array, host-array
 * and queue calls; single_id or inside a batch with material ids next to tables and analytic materials — its units are then
 * evaluated by a second launch of the same call, through a descriptor kept behind the image): eval returns f * cos(theta_o);
 * sample() draws from the file's own luminance / vndf warps whatever MRL_OPT_SAMPLING says, and reports eval / pdf AT the Float
 * direction it returns.  The n-channel entry points render its id as zeros; one-unit mrl_scalar_* calls do not take it
 * (MRL_ERR_MATERIAL); mrl_material_host_table does (one-unit calls on the CPU); device groups replicate it like a table.  Spectral files ("spectra" +
 * "wavelengths" instead of "rgb") become MRL_KIND_RGL_SPECTRAL materials with entry points of their own: see mrl_rgl_spectral_fields below. */
typedef struct mrl_rgl_fields {
    int n_phi, n_theta;
    const float *phi_i, *theta_i;
    int res_ndf[2], res_sigma[2], res[2];
    const float *ndf, *sigma, *vndf, *luminance, *rgb;
    int jacobian;                    /* the file's "jacobian" flag: multiply the spectrum by ndf / (4 sigma) */
} mrl_rgl_fields;
int mrl_material_upload_rgl(mrl_ctx *ctx, const mrl_rgl_fields *fields, int *out_id);
/* the same from a tensor_file container holding the fields under their RGL names (phi_i, theta_i, ndf, sigma, vndf,
 * luminance, rgb, jacobian); why a file was refused: mrl_tensor_file_last_error(NULL) or mrl_last_error(ctx) */
int mrl_material_load_rgl(mrl_ctx *ctx, const char *path, int *out_id);
/* Spectral RGL files ("optional spectral channels"): "spectra" [n_phi][n_theta][n_wavelengths][res[1]][res[0]] over
 * the strictly ascending grid "wavelengths" [n_wavelengths] where the *_rgb.bsdf variant holds "rgb" (base.rgb is ignored).  Upstream's
 * `measured` evaluates such a file, in its spectral variants, with the ray's wavelengths as a THIRD interpolated parameter (linear
 * between the file's nodes, clamped outside them); so do the mrl_*_spectral_batch calls: W values per unit at the wavelengths the caller
 * passes PER UNIT — wavelengths [n][W], what hero-wavelength rendering carries per ray — or, with wavelengths == NULL, at the file's own
 * nodes (W must then be n_wavelengths: the n-channel form of the material, channel = node).  out_values / out_weight are [n][W];
 * pdf and the sampled direction do not depend on the wavelength; sample() reports eval / pdf AT the Float direction it returns, as for
 * RGB files.  Whole arrays, one material (no material-id array), host or device pointers (host arrays are staged in chunks).
 * mrl_pdf_batch serves the kind too (the pdf is wavelength-free), mrl_material_load_rgl reads either variant,
 * mrl_material_save_image / _load_image and mrl_material_host_table take it; the RGB entry points answer MRL_ERR_MATERIAL for a
 * single_id of this kind and render it as zeros inside a batch with material ids.  PARITY UNPINNED (no spectral file exists offline;
 * checker: oracle/rgl_oracle.c, rgl_eval_pdf_spectral / rgl_sample_spectral). */
typedef struct mrl_rgl_spectral_fields {
    mrl_rgl_fields base;             /* phi_i .. luminance, jacobian as for an RGB file; base.rgb unused */
    int n_wavelengths;
    const float *wavelengths;        /* [n_wavelengths], strictly ascending */
    const float *spectra;            /* [n_phi][n_theta][n_wavelengths][res[1]][res[0]] */
} mrl_rgl_spectral_fields;
int mrl_material_upload_rgl_spectral(mrl_ctx *ctx, const mrl_rgl_spectral_fields *fields, int *out_id);
/* the file's wavelength grid: *n_wavelengths always; the nodes into out when it is not NULL (max_floats >= *n_wavelengths) */
int mrl_material_wavelengths(mrl_ctx *ctx, int id, int *n_wavelengths, float *out, size_t max_floats);
int mrl_eval_spectral_batch(mrl_ctx *ctx, const float *wi, const float *wo, const float *wavelengths /* [n][W] or NULL */, int n_wavelengths /* W */,
                            int32_t id, size_t n, float *out_values /* [n][W] */);
int mrl_eval_pdf_spectral_batch(mrl_ctx *ctx, const float *wi, const float *wo, const float *wavelengths, int n_wavelengths, int32_t id, size_t n,
                                float *out_values, float *out_pdf);
int mrl_sample_spectral_batch(mrl_ctx *ctx, const float *wi, const float *u, const float *wavelengths, int n_wavelengths, int32_t id, size_t n,
                              float *out_wo, float *out_pdf, float *out_weight /* [n][W] */);
int mrl_eval_sample_spectral_batch(mrl_ctx *ctx, const float *wi, const float *wo, const float *u, const float *wavelengths, int n_wavelengths, int32_t id,
                                   size_t n, float *out_values, float *out_pdf, float *out_wo, float *out_pdf2, float *out_weight);
/* On-disk cache of a material's DEVICE image: what is resident for the material — the texels as Float (RGB tables
 * in the compact rows form, 24 MB for a MERL table, whatever the context's layout: a brick context expands them on the device), the
 * sampling marginal, the conditional sampling rows; for an RGL material the bracket-major image with its cell records — written so that
 * another process makes the material resident with one read and one copy: no parse, no re-layout of the f64 payload, no quadrature /
 * prefix-scan kernels, no host normalisation.  5.5 ms instead of 12.4 per MERL table.  Table, n-channel and RGL materials;
 * not analytic ones.  An image is tied to this library version and (when it holds conditional sampling rows) to the lookup / node options
 * it was built under: anything else is refused with MRL_ERR_FORMAT, as are truncated or altered files (every size is recomputed from
 * the header's shapes, the payload carries a checksum); the context is then as it was. */
int mrl_material_save_image(mrl_ctx *ctx, int id, const char *path);
int mrl_material_load_image(mrl_ctx *ctx, const char *path, int *out_id);
/* number of material SLOTS (live + released); ids are slot indices */
int mrl_material_count(const mrl_ctx *ctx);
int mrl_material_info(const mrl_ctx *ctx, int id, int *kind, int dims[3]);
int mrl_material_param(const mrl_ctx *ctx, int id, int *param);
/* The conditional sampling table P(theta_h | theta_i) of an RGB table material (MRL_OPT_SAMPLING = 2), as the device built
 * it at upload (a quadrature kernel through the resident table's own trilinear lookup + a prefix-scan kernel): n_ti incident
 * bins uniform in cos(theta_i), each a row of n_th + 1 cdf values followed by n_th densities c.  out == NULL: sizes only. */
int mrl_material_sampling2d(mrl_ctx *ctx, int id, int *n_ti, int *n_th, double *out, size_t max_doubles);     /* enum mrl_param of a table material (GGX: MRL_ERR_MATERIAL) */
/* Frees a material's device memory (plugin destructor).  Waits for the context's stream first.  The slot becomes a
 * tombstone: batch and queue calls treat its id like an unknown id (every output zero), single_id calls and
 * mrl_material_info return MRL_ERR_MATERIAL.  A later upload may reuse the slot (lowest free slot first), exactly like
 * a file descriptor — do not keep ids of released materials in material-id arrays. */
int mrl_material_release(mrl_ctx *ctx, int id);
/* Resident bytes of this context: material data (tables + sampling marginals), scratch the context holds (staging,
 * partition work areas), and the device's free / total memory as the runtime reports them.  Any pointer may be NULL. */
int mrl_memory_info(const mrl_ctx *ctx, size_t *material_bytes, size_t *workspace_bytes, size_t *device_free, size_t *device_total);

/* ---- batched hot path.  mat == NULL: every unit uses single_id ---- */
int mrl_eval_batch(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id,
                   size_t n, float *out_rgb);
int mrl_pdf_batch(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id,
                  size_t n, float *out_pdf);
int mrl_sample_batch(mrl_ctx *ctx, const float *wi, const float *u, const int32_t *mat, int32_t single_id,
                     size_t n, float *out_wo, float *out_pdf, float *out_weight);
/* eval and pdf of the same pairs in one launch */
int mrl_eval_pdf_batch(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id,
                       size_t n, float *out_rgb, float *out_pdf);
/* the benchmarked unit: eval(wi,wo) rgb, pdf(wi,wo), sample(wi,u) -> (wo', pdf', weight') */
int mrl_eval_sample_batch(mrl_ctx *ctx, const float *wi, const float *wo, const float *u,
                          const int32_t *mat, int32_t single_id, size_t n,
                          float *out_rgb, float *out_pdf, float *out_wo, float *out_pdf2, float *out_weight);

/* ---- one-unit calls: the virtual BSDF::eval / sample / pdf of a stock per-ray integrator ----
 * The same fused unit for ONE (wi, wo, u), for callers that cannot batch.  out[11] = rgb[3] pdf wo'[3] pdf' weight'[3];
 * the numbers are those mrl_eval_sample_batch returns for the unit (the service runs the batch kernels' own per-lane
 * functions).  No launch and no stream synchronisation on the call path: the caller writes its request into a mailbox in
 * pinned host memory and a resident wave — one lane per calling thread, so concurrent callers are answered side by
 * side — writes the result back.  Service instances have a bounded lifetime (500 us) and are relaunched by the callers,
 * so a device-wide synchronisation elsewhere in the process waits a bounded time and nothing spins once calls stop.
 * Thread-safe and lock-free between callers; uploads, releases and option changes wait for calls in flight.
 * Table materials with three channels and GGX materials; other ids are MRL_ERR_MATERIAL. */
int mrl_scalar_eval_sample(mrl_ctx *ctx, int32_t material, const float wi[3], const float wo[3], const float u[2], float out[11]);
/* the two halves on their own — what one virtual eval() / pdf() / eval_pdf(), resp. one sample(), needs: one table lookup
 * instead of two on the device */
int mrl_scalar_eval_pdf(mrl_ctx *ctx, int32_t material, const float wi[3], const float wo[3], float out_rgb[3], float *out_pdf);
int mrl_scalar_sample(mrl_ctx *ctx, int32_t material, const float wi[3], const float u[2], float out_wo[3], float *out_pdf, float out_weight[3]);

/* ---- one-unit calls on the calling CPU thread (SURVEY.md §8b "what calls it (2)": scalar BSDF::eval / sample / pdf call
 * the CPU core directly, no shim hop to the device).  Replaces, for the per-ray virtual calls of a stock integrator, what the
 * reference's scalar plugin does in its eval()/sample()/pdf() bodies (reference sources absent: README.md:1 names the plugins).
 * mrl_material_host_table takes a host image of a RESIDENT three-channel table — the device's own Float texel values, copied
 * back from HBM once (24 MB for a MERL table), with the sampling marginal and a snapshot of the context's lookup options
 * (MRL_OPT_LOOKUP / NODE / DISK_MAP / SAMPLING as they are at that moment) — and the mrl_host_* calls evaluate one unit on
 * it with the kernels' own per-unit functions compiled for the host (one formulation, two targets; the two differ only in
 * the hardware reciprocal seeds, i.e. by ~1e-15 before rounding: sampled directions and cosine pdfs are bit-identical to
 * the batch calls', values and weights equal to 1 ulp of Float).  The image is immutable: the calls are lock-free, allocation-
 * free and safe from any number of threads; it is reference-counted and outlives the material and the context it was taken
 * from.  This is NOT a fallback for the batch / queue calls (they have none: no device, no context) — it is where ONE-unit
 * calls belong: 0.2-0.4 us on a core against 4.8-6.7 us through the device's one-unit call service (mrl_scalar_*).
 * Needs a host CPU with FMA + AVX2 (MRL_ERR_INVALID otherwise).  GGX and n-channel materials: MRL_ERR_MATERIAL. ---- */
typedef struct mrl_host_table mrl_host_table;
int mrl_material_host_table(mrl_ctx *ctx, int material, mrl_host_table **out);
int mrl_host_table_retain(mrl_host_table *table);
int mrl_host_table_release(mrl_host_table *table);
/* dims, enum mrl_param, lookup mode, sampling strategy and host bytes of an image (any pointer may be NULL) */
int mrl_host_table_info(const mrl_host_table *table, int dims[3], int *param, int *lookup, int *sampling, size_t *bytes);
/* eval(wi, wo) -> rgb (cosine included) and, when out_pdf is not NULL, pdf(wi, wo): one lookup */
int mrl_host_eval_pdf(const mrl_host_table *table, const float wi[3], const float wo[3], float out_rgb[3], float *out_pdf);
/* sample(wi, u) -> wo', pdf', weight' = eval(wi, wo') / pdf' in Float */
int mrl_host_sample(const mrl_host_table *table, const float wi[3], const float u[2], float out_wo[3], float *out_pdf, float out_weight[3]);
/* the fused unit: out[11] = rgb[3] pdf wo'[3] pdf' weight'[3], as mrl_scalar_eval_sample lays it out */
int mrl_host_eval_sample(const mrl_host_table *table, const float wi[3], const float wo[3], const float u[2], float out[11]);
/* a spectral RGL material (MRL_KIND_RGL_SPECTRAL): W values at wavelengths[0 .. W) (NULL: the file's own nodes); the three calls
 * above answer MRL_ERR_MATERIAL for it, these two for every other kind */
int mrl_host_eval_pdf_spectral(const mrl_host_table *table, const float wi[3], const float wo[3], const float *wavelengths, int n_wavelengths,
                               float *out_values, float *out_pdf /* may be NULL */);
int mrl_host_sample_spectral(const mrl_host_table *table, const float wi[3], const float u[2], const float *wavelengths, int n_wavelengths,
                             float out_wo[3], float *out_pdf, float *out_weight);

/* ---- n-channel tables: customized_measurement beyond RGB (monochrome, RGB + alpha, spectral bins; SURVEY.md §8f
 * item 3).  Same MERL parameterisation, same transform and trilinear blend; a texel has n_channels values, 1..32.
 * planar: n_channels planes in MERL order; scale: n_channels factors (NULL = all 1).  Bricks only (the table-layout
 * option does not apply); HBM per cell: 32 B (1 channel), 64 B (2), 128 B x ceil(n_channels / 4) (4..32).
 * n_channels == 3 is the RGB path: the material becomes an MRL_KIND_TABLE and every *_nch call with n_channels == 3
 * forwards to its RGB twin.  In a *_nch batch, ids of materials with another channel count (or analytic, released,
 * unknown ids) give zeros, like unknown ids do in the RGB calls; a single_id of the wrong width is MRL_ERR_MATERIAL.
 * out_values / out_weight: n x n_channels floats, channels of a unit adjacent.  pdf is channel-free: mrl_pdf_batch
 * serves every table kind.  sample()/pdf() follow MRL_OPT_SAMPLING; the row marginal of an n-channel table weighs the
 * channels equally (the RGB path uses luminance).  Host or device pointers, like the RGB calls. ---- */
int mrl_material_upload_table_nch(mrl_ctx *ctx, const double *planar, const int dims[3], int n_channels,
                                  const double *scale, int *out_id);
/* the same with the parameterisation (enum mrl_param) named in the call instead of taken from MRL_OPT_TABLE_PARAM */
int mrl_material_upload_table_param(mrl_ctx *ctx, const double *planar, const int dims[3], int n_channels,
                                    const double *scale, int param, int *out_id);
/* file: int32 dims[3], then n_channels planes as f64 or f32 (told apart by the file length) */
int mrl_material_load_table_nch(mrl_ctx *ctx, const char *path, int n_channels, const double *scale, int *out_id);
int mrl_material_channels(const mrl_ctx *ctx, int id, int *n_channels);
int mrl_eval_batch_nch(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id,
                       size_t n, int n_channels, float *out_values);
int mrl_sample_batch_nch(mrl_ctx *ctx, const float *wi, const float *u, const int32_t *mat, int32_t single_id,
                         size_t n, int n_channels, float *out_wo, float *out_pdf, float *out_weight);
int mrl_eval_pdf_batch_nch(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id,
                           size_t n, int n_channels, float *out_values, float *out_pdf);
int mrl_eval_sample_batch_nch(mrl_ctx *ctx, const float *wi, const float *wo, const float *u,
                              const int32_t *mat, int32_t single_id, size_t n, int n_channels,
                              float *out_values, float *out_pdf, float *out_wo, float *out_pdf2, float *out_weight);

/* the same over a wavefront queue (see "wavefront queues" below): slots queue[0 .. min(*queue_count, capacity)); device pointers only */
int mrl_eval_queue_nch(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id,
                       const uint32_t *queue, const uint32_t *queue_count, size_t capacity, int n_channels, float *out_values);
int mrl_sample_queue_nch(mrl_ctx *ctx, const float *wi, const float *u, const int32_t *mat, int32_t single_id,
                         const uint32_t *queue, const uint32_t *queue_count, size_t capacity, int n_channels,
                         float *out_wo, float *out_pdf, float *out_weight);
int mrl_eval_pdf_queue_nch(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id,
                           const uint32_t *queue, const uint32_t *queue_count, size_t capacity, int n_channels,
                           float *out_values, float *out_pdf);
int mrl_eval_sample_queue_nch(mrl_ctx *ctx, const float *wi, const float *wo, const float *u,
                              const int32_t *mat, int32_t single_id,
                              const uint32_t *queue, const uint32_t *queue_count, size_t capacity, int n_channels,
                              float *out_values, float *out_pdf, float *out_wo, float *out_pdf2, float *out_weight);

/* ---- "tensor_file" container (the RGL material database's *.bsdf files, read by upstream Mitsuba 3's `measured`
 * plugin).  LOADER ONLY: fields are listed and copied out; the adaptive parameterisation an RGL
 * *.bsdf describes is not evaluated by this library.  A customized_measurement table may be stored in the container
 * (float field [channels, n_theta_h, n_theta_d, n_phi_d] + optional "scale" [channels]) and loaded as a material.
 * dtype codes: 1-8 integers (1-2: 1 byte, 3-4: 2, 5-6: 4, 7-8: 8 bytes), 9 f16, 10 f32, 11 f64.  Host code only. ---- */
typedef struct mrl_tensor_file mrl_tensor_file;
int mrl_tensor_file_open(const char *path, mrl_tensor_file **out);
int mrl_tensor_file_close(mrl_tensor_file *f);
const char *mrl_tensor_file_last_error(const mrl_tensor_file *f);   /* f == NULL: why the last open / load of this thread failed */
int mrl_tensor_file_field_count(const mrl_tensor_file *f);
int mrl_tensor_file_find(const mrl_tensor_file *f, const char *name);             /* index, or < 0 */
int mrl_tensor_file_field_info(const mrl_tensor_file *f, int index, const char **name, int *dtype, int *ndim, const uint64_t **shape);
const void *mrl_tensor_file_field_data(const mrl_tensor_file *f, int index, size_t *bytes);   /* raw payload, valid until close */
int mrl_tensor_file_read_f64(const mrl_tensor_file *f, int index, double *out, size_t capacity);   /* float fields, converted */
/* field == NULL: "table".  The material is an RGB table for 3 channels, an n-channel table otherwise.  Optional fields of
 * the container: "scale" (one factor per channel) and "parameterization" (one integer, enum mrl_param: the file says which
 * angles index it; without the field MRL_OPT_TABLE_PARAM decides). */
int mrl_material_load_tensor_table(mrl_ctx *ctx, const char *path, const char *field, int *out_id, int *out_channels);

/* ---- wavefront queues.  A wavefront path tracer keeps its path state in arrays
 * indexed by path slot and a queue of the slots that hit this BSDF.  These calls process the units
 * queue[0 .. min(*queue_count, capacity)): inputs are read from, and outputs written to, the slots
 * the queue names; every other slot is left untouched.  queue_count lives in DEVICE memory (the
 * kernel that built the queue wrote it), so no host round trip separates queue building from the
 * BSDF call.  Device(-accessible) pointers only; asynchronous on the context's stream.  The caller
 * guarantees that every queued index addresses a valid slot of the arrays. ---- */
/* Per-material compaction (wavefront ballot/prefix, no atomics): a stable partition of the slots [0, n) by
 * material id.  queue_out[n] receives the slot indices grouped by material, ascending inside each group;
 * offsets_out[mrl_material_count() + 1]: group m is queue_out[offsets[m] .. offsets[m + 1]);
 * counts_out[mrl_material_count()]: the group sizes — pass queue_out + offsets[m] (host-known only after a read
 * back) or simply the whole layout plus &counts_out[m] to the calls below.  Slots whose id names no material are
 * dropped.  Device pointers only; asynchronous.  n == 0 leaves every count at zero. */
int mrl_partition_by_material(mrl_ctx *ctx, const int32_t *mat, size_t n,
                              uint32_t *queue_out, uint32_t *offsets_out, uint32_t *counts_out);
int mrl_eval_queue(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id,
                   const uint32_t *queue, const uint32_t *queue_count, size_t capacity, float *out_rgb);
int mrl_pdf_queue(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id,
                  const uint32_t *queue, const uint32_t *queue_count, size_t capacity, float *out_pdf);
int mrl_eval_pdf_queue(mrl_ctx *ctx, const float *wi, const float *wo, const int32_t *mat, int32_t single_id,
                       const uint32_t *queue, const uint32_t *queue_count, size_t capacity,
                       float *out_rgb, float *out_pdf);
int mrl_sample_queue(mrl_ctx *ctx, const float *wi, const float *u, const int32_t *mat, int32_t single_id,
                     const uint32_t *queue, const uint32_t *queue_count, size_t capacity,
                     float *out_wo, float *out_pdf, float *out_weight);
int mrl_eval_sample_queue(mrl_ctx *ctx, const float *wi, const float *wo, const float *u,
                          const int32_t *mat, int32_t single_id,
                          const uint32_t *queue, const uint32_t *queue_count, size_t capacity,
                          float *out_rgb, float *out_pdf, float *out_wo, float *out_pdf2, float *out_weight);

/* ---- synthetic inputs, generated in place on the device; device pointers only ---- */
int mrl_generate_pairs(mrl_ctx *ctx, uint64_t seed, uint64_t first_index, size_t n,
                       float *wi, float *wo, float *u);
int mrl_generate_materials(mrl_ctx *ctx, uint64_t seed, uint64_t first_index, size_t n,
                           int n_materials, int32_t *mat);

/* ---- device memory + timing helpers for hosts that have no allocator of their own ---- */
int mrl_device_alloc(mrl_ctx *ctx, size_t bytes, void **out);
int mrl_device_free(mrl_ctx *ctx, void *ptr);
int mrl_copy_to_device(mrl_ctx *ctx, void *dst_device, const void *src_host, size_t bytes);
int mrl_copy_to_host(mrl_ctx *ctx, void *dst_host, const void *src_device, size_t bytes);
/* pinned host memory the device can dereference: a batch call on such pointers runs zero-copy
 * (this is what the plugin adapters' scalar eval()/sample()/pdf() calls use for their 1-unit batches) */
int mrl_host_alloc(mrl_ctx *ctx, size_t bytes, void **out);
int mrl_host_free(mrl_ctx *ctx, void *ptr);
/* hipEvent pair recorded on the stream the kernels are launched on */
int mrl_timer_start(mrl_ctx *ctx);
int mrl_timer_stop(mrl_ctx *ctx, float *elapsed_ms);   /* records, synchronises, returns ms */

/* ---- device groups: one host process, several GPUs (SURVEY.md §8b `mrl_init(n_devices, device_ids, ...)`, §8e) ----
 * The path shards with no data-path collective: member r owns the contiguous unit tile
 *     [r * ceil(N / G), min(N, (r + 1) * ceil(N / G)))        (mrl_tile_bounds)
 * keeps every material table replicated, and computes its tile alone on its own stream.  The only communication is
 * the delivery of per-tile RESULTS to a root device: peers send their chunk straight to the root (grouped
 * ncclSend / ncclRecv over RCCL: each peer's own xGMI link, not a ring), cut into chunks so that the send of chunk k
 * overlaps the compute of chunk k + 1.  A group is thread-compatible (one caller at a time; its member contexts are thread-safe).  A renderer host stays in
 * C++ and reaches every GPU of the node through these calls; it replaces nothing in the reference (the reference
 * has no multi-device path), it is the data-parallel outer loop around mrl_eval_sample_batch. */
typedef struct mrl_group mrl_group;

enum mrl_group_transport {
    MRL_TRANSPORT_AUTO = 0,        /* RCCL when the group has >= 2 distinct devices, else device copies */
    MRL_TRANSPORT_RCCL = 1,        /* grouped ncclSend/ncclRecv on a side stream per device (librccl is loaded on demand) */
    MRL_TRANSPORT_PEER_COPY = 2    /* hipMemcpyPeerAsync on a side stream per device: no RCCL; also valid when device_ids
                                      repeats a device (rehearsal of the whole pipeline on a box with fewer GPUs) */
};

/* per-member device pointers to the tile's inputs, indexed from the tile's first unit; mat may be NULL */
typedef struct mrl_tile_inputs {
    const float *wi, *wo, *u;
    const int32_t *mat;
} mrl_tile_inputs;

int mrl_group_init(int n_devices, const int *device_ids, int transport, mrl_group **out);
int mrl_group_destroy(mrl_group *g);
int mrl_group_size(const mrl_group *g);
int mrl_group_transport(const mrl_group *g);                       /* the transport in use (never AUTO) */
const char *mrl_group_last_error(const mrl_group *g);            /* g == NULL: why the last mrl_group_init of this thread failed */
/* member `rank`'s context (owned by the group): device memory helpers, generators, single-device calls */
int mrl_group_context(mrl_group *g, int rank, mrl_ctx **out);
/* replicated state: applied to every member; material ids are the same on every member */
int mrl_group_set_option(mrl_group *g, int option, int value);
int mrl_group_material_load_merl(mrl_group *g, const char *path, int *out_id);
int mrl_group_material_upload_f64(mrl_group *g, const double *planar_rgb, int *out_id);
int mrl_group_material_upload_table(mrl_group *g, const double *planar_rgb, const int dims[3], const double scale[3], int *out_id);
int mrl_group_material_ggx(mrl_group *g, float alpha, const float eta[3], const float k[3], int *out_id);
int mrl_group_material_upload_rgl(mrl_group *g, const mrl_rgl_fields *fields, int *out_id);      /* one image per member, like a table */
int mrl_group_material_load_rgl(mrl_group *g, const char *path, int *out_id);                      /* RGB or spectral file */
int mrl_group_material_upload_rgl_spectral(mrl_group *g, const mrl_rgl_spectral_fields *fields, int *out_id);
int mrl_group_material_release(mrl_group *g, int id);
/* tile / chunk arithmetic (pure functions; usable without a device) */
void mrl_tile_bounds(size_t n_total, int world, int rank, size_t *lo, size_t *hi);
/* chunk `step` of member `rank`'s tile: units [*lo, *hi) in global numbering (empty once the tile is exhausted) */
void mrl_chunk_bounds(size_t n_total, int world, int rank, size_t chunk_units, size_t step, size_t *lo, size_t *hi);
size_t mrl_chunk_steps(size_t n_total, int world, size_t chunk_units);
/* The schedule of one sharded call as data — pure arithmetic like the three functions above, no device: the operations
 * mrl_group_eval_sample_sharded / mrl_group_eval_sharded issue, in issue order.  Per step: one COMPUTE per member that still has
 * units (buffer -1: the root writes the caller's arrays; 0 / 1: a peer writes its chunk buffer, after the transfer of step
 * after_transfer_of_step — the last reader of that buffer — has left it, -1: none in this call), then one TRANSFER per such
 * peer out of that buffer to the root's arrays at [first, first + count).  Returns the number of operations (also when
 * ops is NULL or max_ops is too small: call twice).  The library's own pipeline walks this list. */
enum mrl_plan_kind { MRL_PLAN_COMPUTE = 0, MRL_PLAN_TRANSFER = 1 };
typedef struct mrl_plan_op {
    int kind;                       /* enum mrl_plan_kind */
    int member;
    int buffer;                     /* -1: the caller's arrays on the root; 0 / 1: the member's chunk buffer */
    size_t step;
    size_t first, count;            /* units [first, first + count) of the batch */
    size_t tile_offset;             /* first - the member's tile start: where its inputs sit in mrl_tile_inputs */
    long long after_transfer_of_step;
} mrl_plan_op;
size_t mrl_group_plan(size_t n_total, int world, size_t chunk_units, int root, mrl_plan_op *ops, size_t max_ops);
/* One payload of `bytes` from every peer to the root, each link on its own, over the named transport (AUTO: the group's):
 * timed with events, compared bit for bit on the host.  out: n_devices entries (the root's own entry stays ok = 1, 0 bytes/s).
 * The first contact of a machine's links with this traffic should be this call, not the pipeline. */
typedef struct mrl_link_report {
    int peer, ok;
    size_t bytes, mismatches;
    float ms, GBps;
} mrl_link_report;
int mrl_group_link_test(mrl_group *g, size_t bytes, int transport, int root, mrl_link_report *out);
/* Synthetic inputs generated in place on every member for its own tile (a pure function of the unit
 * index, so the union over members equals mrl_generate_pairs over [first_index, first_index + n_total)).
 * n_materials > 0 also fills mat with ids in [0, n_materials).  The buffers belong to the group and stay valid until
 * the next mrl_group_generate_tiles or mrl_group_destroy.  tiles_out: n_devices entries. */
int mrl_group_generate_tiles(mrl_group *g, uint64_t seed, uint64_t first_index, size_t n_total, int n_materials,
                             mrl_tile_inputs *tiles_out);
/* The sharded fused unit: every member computes eval+sample over its tile (inputs: tiles[rank], device pointers on that
 * member's device) in chunks of chunk_units; results land in the root member's device arrays (n_total units each, global
 * unit order).  Asynchronous: returns when everything is enqueued; the root member's context stream is ordered after the
 * last transfer (mrl_group_synchronize waits for every member).  With one member this IS mrl_eval_sample_batch. */
int mrl_group_eval_sample_sharded(mrl_group *g, const mrl_tile_inputs *tiles, int32_t single_id, size_t n_total,
                                  size_t chunk_units, int root,
                                  float *out_rgb, float *out_pdf, float *out_wo, float *out_pdf2, float *out_weight);
/* eval only: the same pipeline with one result array — 12 B per unit across the links instead of 44 (10.5 GB
 * instead of 38.5 GB into the root for 10^9 units on 8 GPUs).  tiles[r].u may be NULL. */
int mrl_group_eval_sharded(mrl_group *g, const mrl_tile_inputs *tiles, int32_t single_id, size_t n_total,
                           size_t chunk_units, int root, float *out_rgb);
/* Host arrays of n units (what a CPU renderer holds): tiles are staged to the members and back concurrently, one
 * host thread per member; no device-to-device traffic at all.  Returns when the outputs are written. */
int mrl_group_eval_sample_batch(mrl_group *g, const float *wi, const float *wo, const float *u, const int32_t *mat,
                                int32_t single_id, size_t n, float *out_rgb, float *out_pdf, float *out_wo,
                                float *out_pdf2, float *out_weight);
/* the other batch calls over host arrays, split the same way (BSDF::eval / pdf / eval_pdf / sample on every GPU of the node) */
int mrl_group_eval_batch(mrl_group *g, const float *wi, const float *wo, const int32_t *mat, int32_t single_id, size_t n, float *out_rgb);
int mrl_group_pdf_batch(mrl_group *g, const float *wi, const float *wo, const int32_t *mat, int32_t single_id, size_t n, float *out_pdf);
int mrl_group_eval_pdf_batch(mrl_group *g, const float *wi, const float *wo, const int32_t *mat, int32_t single_id, size_t n,
                             float *out_rgb, float *out_pdf);
int mrl_group_sample_batch(mrl_group *g, const float *wi, const float *u, const int32_t *mat, int32_t single_id, size_t n,
                           float *out_wo, float *out_pdf, float *out_weight);
int mrl_group_synchronize(mrl_group *g);
/* device time of the last mrl_group_eval_sample_sharded per member (its first launch to its last launch or send),
 * valid after mrl_group_synchronize: ms_out[n_devices] */
int mrl_group_last_timing(mrl_group *g, float *ms_out);

#ifdef __cplusplus
}
#endif
#endif
