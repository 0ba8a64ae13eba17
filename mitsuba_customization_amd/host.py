"""Python host over the C ABI (include/merl_hip.h) — plumbing for tests, bench and sharding.

torch is used for what the task allows it for: device memory (tensors), streams and
torch.distributed.  Every compute call lands in libmerl_hip.so; there is no Python or CPU
evaluation path, and a missing library or GPU raises instead of falling back.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MRL_LIB_PATH") or os.path.join(_PKG, "lib", "libmerl_hip.so")   # env override: A/B builds

OPT_LOOKUP, OPT_NODE, OPT_DISK_MAP, OPT_KERNEL, OPT_HOST_CHUNK, OPT_TABLE_LAYOUT, OPT_SAMPLING, OPT_MEMORY_LIMIT_MB, OPT_HOST_THREADS, OPT_BLOCK_MAP = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
OPT_TABLE_PARAM = 10
OPT_TABLE_ARENA_MB = 11
OPT_RGL_SEARCH = 12           # 0: RGL search tables from LDS when they fit (default), 1: always from memory
OPT_COSINE_FACTOR = 13        # 0: eval() = f cos(theta_o) (default), 1: eval() = f            
OPT_RESERVED_CUS = 15          # compute units the batch kernels leave to communication kernels (CU-masked stream)
OPT_NEGATIVE = 14             # negative stored values: 0 clamp (default), 1 keep, 2 skip and renormalise   
NEGATIVE_CLAMP, NEGATIVE_KEEP, NEGATIVE_RENORMALISE = 0, 1, 2
PARAM_HALF_DIFF, PARAM_STANDARD, PARAM_STANDARD_FULL = 0, 1, 2          # enum mrl_param
SAMPLING_COSINE, SAMPLING_TABLE, SAMPLING_TABLE_2D = 0, 1, 2
LAYOUT_ROWS, LAYOUT_BRICK = 0, 1
LOOKUP_NEAREST, LOOKUP_TRILINEAR = 0, 1
KIND_MERL, KIND_TABLE, KIND_GGX = 0, 1, 2
KIND_TABLE_NCH = 4
KIND_RGL = 5
KIND_RGL_SPECTRAL = 6
ERR_INVALID, ERR_HIP, ERR_IO, ERR_FORMAT, ERR_OOM, ERR_MATERIAL, ERR_POINTER_MIX, ERR_NO_DEVICE = -1, -2, -3, -4, -5, -6, -7, -8

# every symbol include/merl_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = (
    "mrl_init", "mrl_destroy", "mrl_strerror", "mrl_build_info", "mrl_last_error", "mrl_set_option", "mrl_get_option",
    "mrl_set_stream", "mrl_reset_stream", "mrl_synchronize", "mrl_device_info",
    "mrl_material_load_merl", "mrl_material_upload_f64", "mrl_material_upload_table", "mrl_material_load_table",
    "mrl_scalar_eval_sample", "mrl_scalar_eval_pdf", "mrl_scalar_sample", "mrl_material_ggx", "mrl_material_count", "mrl_material_info", "mrl_material_release", "mrl_memory_info",
    "mrl_eval_batch", "mrl_pdf_batch", "mrl_sample_batch", "mrl_eval_pdf_batch", "mrl_eval_sample_batch",
    "mrl_partition_by_material", "mrl_eval_queue", "mrl_pdf_queue", "mrl_eval_pdf_queue", "mrl_sample_queue", "mrl_eval_sample_queue",
    "mrl_generate_pairs", "mrl_generate_materials",
    "mrl_device_alloc", "mrl_device_free", "mrl_copy_to_device", "mrl_copy_to_host", "mrl_host_alloc", "mrl_host_free",
    "mrl_timer_start", "mrl_timer_stop",
    "mrl_material_upload_table_nch", "mrl_material_upload_table_param", "mrl_material_load_table_nch", "mrl_material_channels", "mrl_material_param",
    "mrl_eval_batch_nch", "mrl_sample_batch_nch", "mrl_eval_pdf_batch_nch", "mrl_eval_sample_batch_nch",
    "mrl_eval_queue_nch", "mrl_sample_queue_nch", "mrl_eval_pdf_queue_nch", "mrl_eval_sample_queue_nch",
    "mrl_tensor_file_open", "mrl_tensor_file_close", "mrl_tensor_file_last_error", "mrl_tensor_file_field_count", "mrl_tensor_file_find",
    "mrl_tensor_file_field_info", "mrl_tensor_file_field_data", "mrl_tensor_file_read_f64", "mrl_material_load_tensor_table",
    "mrl_material_upload_rgl", "mrl_material_load_rgl", "mrl_material_save_image", "mrl_material_load_image",
    "mrl_group_init", "mrl_group_destroy", "mrl_group_size", "mrl_group_transport", "mrl_group_last_error", "mrl_group_context",
    "mrl_group_set_option", "mrl_group_material_load_merl", "mrl_group_material_upload_f64", "mrl_group_material_upload_table",
    "mrl_group_material_ggx", "mrl_group_material_upload_rgl", "mrl_group_material_load_rgl", "mrl_group_material_release", "mrl_tile_bounds", "mrl_chunk_bounds", "mrl_chunk_steps",
    "mrl_group_generate_tiles", "mrl_group_eval_sample_sharded", "mrl_group_eval_sharded", "mrl_group_eval_sample_batch", "mrl_group_synchronize",
    "mrl_group_eval_batch", "mrl_group_pdf_batch", "mrl_group_eval_pdf_batch", "mrl_group_sample_batch",
    "mrl_group_last_timing", "mrl_group_plan", "mrl_group_link_test",
    "mrl_material_sampling2d", "mrl_material_host_table", "mrl_host_table_retain", "mrl_host_table_release", "mrl_host_table_info",
    "mrl_host_eval_pdf", "mrl_host_sample", "mrl_host_eval_sample",
    "mrl_material_upload_rgl_spectral", "mrl_material_wavelengths", "mrl_eval_spectral_batch", "mrl_eval_pdf_spectral_batch",
    "mrl_sample_spectral_batch", "mrl_eval_sample_spectral_batch", "mrl_host_eval_pdf_spectral", "mrl_host_sample_spectral",
    "mrl_group_material_upload_rgl_spectral",
)
TRANSPORT_AUTO, TRANSPORT_RCCL, TRANSPORT_PEER_COPY = 0, 1, 2
ERR_COMM = -9


class HostTable:
    """mrl_host_table: immutable, reference-counted; outlives the context it was taken from."""

    def __init__(self, lib, handle):
        self._lib, self._h = lib, handle

    def close(self):
        if self._h:
            self._lib.mrl_host_table_release(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def info(self):
        dims = (C.c_int * 3)(); prm = C.c_int(); lk = C.c_int(); smp = C.c_int(); nbytes = C.c_size_t()
        self._lib.mrl_host_table_info(self._h, dims, C.byref(prm), C.byref(lk), C.byref(smp), C.byref(nbytes))
        return {"dims": tuple(dims), "param": prm.value, "lookup": lk.value, "sampling": smp.value, "bytes": nbytes.value}

    def eval_sample(self, wi, wo, u) -> np.ndarray:
        """ONE fused unit on this thread: rgb[3] pdf wo'[3] pdf' weight'[3]."""
        a = (C.c_float * 3)(*[float(x) for x in wi]); b = (C.c_float * 3)(*[float(x) for x in wo]); c = (C.c_float * 2)(*[float(x) for x in u])
        out = (C.c_float * 11)()
        rc = self._lib.mrl_host_eval_sample(self._h, a, b, c, out)
        if rc != 0:
            raise MerlHipError(rc, "mrl_host_eval_sample")
        return np.frombuffer(out, dtype=np.float32).copy()

    def eval_sample_spectral(self, wi, wo, u, wavelengths) -> tuple:
        """ONE unit of a spectral RGL material on this thread at `wavelengths` [W]: (values [W], pdf, wo' [3], pdf', weight' [W])."""
        a = (C.c_float * 3)(*[float(x) for x in wi]); b = (C.c_float * 3)(*[float(x) for x in wo]); c = (C.c_float * 2)(*[float(x) for x in u])
        W = len(wavelengths)
        wl = (C.c_float * W)(*[float(x) for x in wavelengths])
        val = (C.c_float * W)(); pdf = C.c_float(); wo2 = (C.c_float * 3)(); pdf2 = C.c_float(); w = (C.c_float * W)()
        rc = self._lib.mrl_host_eval_pdf_spectral(self._h, a, b, wl, W, val, C.byref(pdf))
        if rc == 0:
            rc = self._lib.mrl_host_sample_spectral(self._h, a, c, wl, W, wo2, C.byref(pdf2), w)
        if rc != 0:
            raise MerlHipError(rc, "mrl_host_*_spectral")
        return (np.array(val[:], np.float32), np.float32(pdf.value), np.array(wo2[:], np.float32), np.float32(pdf2.value), np.array(w[:], np.float32))


class RglFields(C.Structure):
    """struct mrl_rgl_fields"""
    _fields_ = [("n_phi", C.c_int), ("n_theta", C.c_int), ("phi_i", C.POINTER(C.c_float)), ("theta_i", C.POINTER(C.c_float)),
                ("res_ndf", C.c_int * 2), ("res_sigma", C.c_int * 2), ("res", C.c_int * 2),
                ("ndf", C.POINTER(C.c_float)), ("sigma", C.POINTER(C.c_float)), ("vndf", C.POINTER(C.c_float)),
                ("luminance", C.POINTER(C.c_float)), ("rgb", C.POINTER(C.c_float)), ("jacobian", C.c_int)]


class RglSpectralFields(C.Structure):
    """struct mrl_rgl_spectral_fields"""
    _fields_ = [("base", RglFields), ("n_wavelengths", C.c_int), ("wavelengths", C.POINTER(C.c_float)), ("spectra", C.POINTER(C.c_float))]


def rgl_fields_struct(fields: dict):
    """(struct mrl_rgl_fields — or mrl_rgl_spectral_fields for a dict with "spectra" + "wavelengths" instead of "rgb" —, the arrays it
    points into) from a dict of RGL field arrays."""
    spectral = "spectra" in fields and "rgb" not in fields
    names = ("phi_i", "theta_i", "ndf", "sigma", "vndf", "luminance") + (("spectra", "wavelengths") if spectral else ("rgb",))
    a = {k: np.ascontiguousarray(fields[k], np.float32) for k in names}
    vn = a["vndf"].shape
    values = a["spectra" if spectral else "rgb"]
    n_values = a["wavelengths"].shape[0] if spectral else 3
    if len(vn) != 4 or a["luminance"].shape != vn or values.shape != (vn[0], vn[1], n_values, vn[2], vn[3]) or a["ndf"].ndim != 2 or a["sigma"].ndim != 2 \
            or a["phi_i"].shape != (vn[0],) or a["theta_i"].shape != (vn[1],) or (spectral and a["wavelengths"].ndim != 1):
        raise ValueError("RGL fields: vndf / luminance [n_phi, n_theta, res, res], rgb [n_phi, n_theta, 3, res, res] "
                         "(or spectra [n_phi, n_theta, n_wavelengths, res, res] + wavelengths), ndf / sigma 2-D")
    fp = C.POINTER(C.c_float)
    p = lambda k: a[k].ctypes.data_as(fp)
    r = RglFields(vn[0], vn[1], p("phi_i"), p("theta_i"), (C.c_int * 2)(a["ndf"].shape[1], a["ndf"].shape[0]),
                  (C.c_int * 2)(a["sigma"].shape[1], a["sigma"].shape[0]), (C.c_int * 2)(vn[3], vn[2]),
                  p("ndf"), p("sigma"), p("vndf"), p("luminance"), None if spectral else p("rgb"), int(np.asarray(fields.get("jacobian", 1)).reshape(-1)[0]))
    if spectral:
        return RglSpectralFields(r, int(n_values), p("wavelengths"), p("spectra")), a
    return r, a


class TileInputs(C.Structure):
    """mrl_tile_inputs: per-member device pointers to a tile's inputs."""
    _fields_ = [("wi", C.c_void_p), ("wo", C.c_void_p), ("u", C.c_void_p), ("mat", C.c_void_p)]


class MerlHipError(RuntimeError):
    def __init__(self, status: int, what: str, detail: str = ""):
        super().__init__(f"{what}: status {status} ({detail})" if detail else f"{what}: status {status}")
        self.status = status


_lib = None


def load_library(path: Optional[str] = None):
    """dlopen libmerl_hip.so.  torch is imported first so that its bundled libamdhip64.so.7 is
    the one HIP runtime in the process (same SONAME -> the loader reuses it)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p) and path is None and not os.environ.get("MRL_LIB_PATH"):
        # a checkout without build outputs: compile in-tree (hipcc cross-compiles gfx950 anywhere)
        try:
            from . import build as _build
            _build.build_lib()
        except Exception as e:
            raise RuntimeError(f"{p} is missing and could not be built ({e}); there is no CPU fallback for the hot path") from e
    if not os.path.exists(p):
        raise RuntimeError(f"{p} is missing: build it with `python -m mitsuba_customization_amd.build` "
                           "(there is no CPU fallback for the hot path)")
    try:
        import torch  # noqa: F401  (one HIP runtime per process)
    except Exception:  # pragma: no cover - torch is always present in this image
        pass
    L = C.CDLL(p)
    vp, i32p, fp = C.c_void_p, C.POINTER(C.c_int32), C.c_void_p  # arrays travel as raw addresses
    L.mrl_init.argtypes = [C.c_int, C.POINTER(vp)]
    L.mrl_destroy.argtypes = [vp]
    L.mrl_strerror.argtypes = [C.c_int]; L.mrl_strerror.restype = C.c_char_p
    L.mrl_last_error.argtypes = [vp]; L.mrl_last_error.restype = C.c_char_p
    L.mrl_set_option.argtypes = [vp, C.c_int, C.c_int]
    L.mrl_get_option.argtypes = [vp, C.c_int, C.POINTER(C.c_int)]
    L.mrl_set_stream.argtypes = [vp, vp]
    L.mrl_reset_stream.argtypes = [vp]
    L.mrl_synchronize.argtypes = [vp]
    L.mrl_device_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.mrl_material_load_merl.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
    L.mrl_material_upload_f64.argtypes = [vp, vp, C.POINTER(C.c_int)]
    L.mrl_material_upload_table.argtypes = [vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.mrl_material_load_table.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.mrl_material_ggx.argtypes = [vp, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.mrl_material_count.argtypes = [vp]
    L.mrl_material_info.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mrl_material_release.argtypes = [vp, C.c_int]
    L.mrl_memory_info.argtypes = [vp] + [C.POINTER(C.c_size_t)] * 4
    L.mrl_eval_batch.argtypes = [vp, fp, fp, vp, C.c_int32, C.c_size_t, fp]
    L.mrl_pdf_batch.argtypes = [vp, fp, fp, vp, C.c_int32, C.c_size_t, fp]
    L.mrl_partition_by_material.argtypes = [vp, vp, C.c_size_t, vp, vp, vp]
    L.mrl_eval_pdf_batch.argtypes = [vp, fp, fp, vp, C.c_int32, C.c_size_t, fp, fp]
    L.mrl_eval_pdf_queue.argtypes = [vp, fp, fp, vp, C.c_int32, vp, vp, C.c_size_t, fp, fp]
    L.mrl_sample_batch.argtypes = [vp, fp, fp, vp, C.c_int32, C.c_size_t, fp, fp, fp]
    L.mrl_eval_sample_batch.argtypes = [vp, fp, fp, fp, vp, C.c_int32, C.c_size_t, fp, fp, fp, fp, fp]
    L.mrl_eval_queue.argtypes = [vp, fp, fp, vp, C.c_int32, vp, vp, C.c_size_t, fp]
    L.mrl_pdf_queue.argtypes = [vp, fp, fp, vp, C.c_int32, vp, vp, C.c_size_t, fp]
    L.mrl_sample_queue.argtypes = [vp, fp, fp, vp, C.c_int32, vp, vp, C.c_size_t, fp, fp, fp]
    L.mrl_eval_sample_queue.argtypes = [vp, fp, fp, fp, vp, C.c_int32, vp, vp, C.c_size_t, fp, fp, fp, fp, fp]
    L.mrl_generate_pairs.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_size_t, fp, fp, fp]
    L.mrl_generate_materials.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_size_t, C.c_int, vp]
    L.mrl_device_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.mrl_device_free.argtypes = [vp, vp]
    L.mrl_copy_to_device.argtypes = [vp, vp, vp, C.c_size_t]
    L.mrl_copy_to_host.argtypes = [vp, vp, vp, C.c_size_t]
    L.mrl_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.mrl_host_free.argtypes = [vp, vp]
    L.mrl_timer_start.argtypes = [vp]
    L.mrl_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    L.mrl_material_upload_table_nch.argtypes = [vp, vp, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.mrl_material_upload_table_param.argtypes = [vp, vp, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
    L.mrl_material_load_table_nch.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.mrl_material_channels.argtypes = [vp, C.c_int, C.POINTER(C.c_int)]
    L.mrl_scalar_eval_sample.argtypes = [vp, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.mrl_scalar_eval_pdf.argtypes = [vp, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.mrl_scalar_sample.argtypes = [vp, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.mrl_material_param.argtypes = [vp, C.c_int, C.POINTER(C.c_int)]
    L.mrl_material_host_table.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.mrl_host_table_retain.argtypes = [vp]
    L.mrl_host_table_release.argtypes = [vp]
    L.mrl_host_table_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.mrl_host_eval_pdf.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.mrl_host_sample.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.mrl_host_eval_sample.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.mrl_eval_batch_nch.argtypes = [vp, fp, fp, vp, C.c_int32, C.c_size_t, C.c_int, fp]
    L.mrl_sample_batch_nch.argtypes = [vp, fp, fp, vp, C.c_int32, C.c_size_t, C.c_int, fp, fp, fp]
    L.mrl_eval_pdf_batch_nch.argtypes = [vp, fp, fp, vp, C.c_int32, C.c_size_t, C.c_int, fp, fp]
    L.mrl_eval_sample_batch_nch.argtypes = [vp, fp, fp, fp, vp, C.c_int32, C.c_size_t, C.c_int, fp, fp, fp, fp, fp]
    L.mrl_eval_queue_nch.argtypes = [vp, fp, fp, vp, C.c_int32, vp, vp, C.c_size_t, C.c_int, fp]
    L.mrl_sample_queue_nch.argtypes = [vp, fp, fp, vp, C.c_int32, vp, vp, C.c_size_t, C.c_int, fp, fp, fp]
    L.mrl_eval_pdf_queue_nch.argtypes = [vp, fp, fp, vp, C.c_int32, vp, vp, C.c_size_t, C.c_int, fp, fp]
    L.mrl_eval_sample_queue_nch.argtypes = [vp, fp, fp, fp, vp, C.c_int32, vp, vp, C.c_size_t, C.c_int, fp, fp, fp, fp, fp]
    L.mrl_tensor_file_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.mrl_tensor_file_close.argtypes = [vp]
    L.mrl_tensor_file_last_error.argtypes = [vp]; L.mrl_tensor_file_last_error.restype = C.c_char_p
    L.mrl_tensor_file_field_count.argtypes = [vp]
    L.mrl_tensor_file_find.argtypes = [vp, C.c_char_p]
    L.mrl_tensor_file_field_info.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_uint64))]
    L.mrl_tensor_file_field_data.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t)]; L.mrl_tensor_file_field_data.restype = vp
    L.mrl_tensor_file_read_f64.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.mrl_material_load_tensor_table.argtypes = [vp, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mrl_material_upload_rgl.argtypes = [vp, C.POINTER(RglFields), C.POINTER(C.c_int)]
    L.mrl_material_upload_rgl_spectral.argtypes = [vp, C.POINTER(RglSpectralFields), C.POINTER(C.c_int)]
    L.mrl_material_wavelengths.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_size_t]
    L.mrl_eval_spectral_batch.argtypes = [vp, fp, fp, fp, C.c_int, C.c_int32, C.c_size_t, fp]
    L.mrl_eval_pdf_spectral_batch.argtypes = [vp, fp, fp, fp, C.c_int, C.c_int32, C.c_size_t, fp, fp]
    L.mrl_sample_spectral_batch.argtypes = [vp, fp, fp, fp, C.c_int, C.c_int32, C.c_size_t, fp, fp, fp]
    L.mrl_eval_sample_spectral_batch.argtypes = [vp, fp, fp, fp, fp, C.c_int, C.c_int32, C.c_size_t, fp, fp, fp, fp, fp]
    cfp = C.POINTER(C.c_float)
    L.mrl_host_eval_pdf_spectral.argtypes = [vp, cfp, cfp, cfp, C.c_int, cfp, cfp]
    L.mrl_host_sample_spectral.argtypes = [vp, cfp, cfp, cfp, C.c_int, cfp, cfp, cfp]
    L.mrl_material_load_rgl.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
    L.mrl_material_save_image.argtypes = [vp, C.c_int, C.c_char_p]
    L.mrl_material_load_image.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
    szp = C.POINTER(C.c_size_t)
    L.mrl_group_init.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.mrl_group_destroy.argtypes = [vp]
    L.mrl_group_size.argtypes = [vp]
    L.mrl_group_transport.argtypes = [vp]
    L.mrl_group_last_error.argtypes = [vp]; L.mrl_group_last_error.restype = C.c_char_p
    L.mrl_group_context.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.mrl_group_set_option.argtypes = [vp, C.c_int, C.c_int]
    L.mrl_group_material_load_merl.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
    L.mrl_group_material_upload_f64.argtypes = [vp, vp, C.POINTER(C.c_int)]
    L.mrl_group_material_upload_table.argtypes = [vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.mrl_group_material_ggx.argtypes = [vp, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.mrl_group_material_upload_rgl.argtypes = [vp, C.POINTER(RglFields), C.POINTER(C.c_int)]
    L.mrl_group_material_load_rgl.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
    L.mrl_group_material_release.argtypes = [vp, C.c_int]
    L.mrl_tile_bounds.argtypes = [C.c_size_t, C.c_int, C.c_int, szp, szp]; L.mrl_tile_bounds.restype = None
    L.mrl_chunk_bounds.argtypes = [C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_size_t, szp, szp]; L.mrl_chunk_bounds.restype = None
    L.mrl_chunk_steps.argtypes = [C.c_size_t, C.c_int, C.c_size_t]; L.mrl_chunk_steps.restype = C.c_size_t
    L.mrl_group_generate_tiles.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_size_t, C.c_int, C.POINTER(TileInputs)]
    L.mrl_group_eval_sample_sharded.argtypes = [vp, C.POINTER(TileInputs), C.c_int32, C.c_size_t, C.c_size_t, C.c_int, fp, fp, fp, fp, fp]
    L.mrl_group_eval_sharded.argtypes = [vp, C.POINTER(TileInputs), C.c_int32, C.c_size_t, C.c_size_t, C.c_int, fp]
    L.mrl_group_eval_sample_batch.argtypes = [vp, fp, fp, fp, vp, C.c_int32, C.c_size_t, fp, fp, fp, fp, fp]
    L.mrl_group_eval_batch.argtypes = [vp, fp, fp, vp, C.c_int32, C.c_size_t, fp]
    L.mrl_group_pdf_batch.argtypes = [vp, fp, fp, vp, C.c_int32, C.c_size_t, fp]
    L.mrl_group_eval_pdf_batch.argtypes = [vp, fp, fp, vp, C.c_int32, C.c_size_t, fp, fp]
    L.mrl_group_sample_batch.argtypes = [vp, fp, fp, vp, C.c_int32, C.c_size_t, fp, fp, fp]
    L.mrl_group_synchronize.argtypes = [vp]
    L.mrl_group_last_timing.argtypes = [vp, C.POINTER(C.c_float)]
    if path is None:
        _lib = L
    return L


def _is_tensor(x) -> bool:
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


def _addr(x, dtype, cols: Optional[int], n: Optional[int], name: str):
    """Raw address of a contiguous array (torch device tensor or numpy host array) after checks."""
    if x is None:
        return None
    if _is_tensor(x):
        import torch
        want = torch.float32 if dtype == np.float32 else torch.int32
        if x.dtype != want or not x.is_contiguous():
            raise ValueError(f"{name}: need a contiguous {want} tensor")
        shape = tuple(x.shape)
        ptr = x.data_ptr()
    else:
        if not isinstance(x, np.ndarray) or x.dtype != dtype or not x.flags["C_CONTIGUOUS"]:
            raise ValueError(f"{name}: need a C-contiguous numpy {np.dtype(dtype).name} array")
        shape = x.shape
        ptr = x.ctypes.data
    want_shape = (n,) if cols is None else (n, cols)
    if n is not None and shape != want_shape:
        raise ValueError(f"{name}: shape {shape}, expected {want_shape}")
    return ptr


class MerlHip:
    """One context = one GPU (one process per GPU)."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        rc = self._lib.mrl_init(device, C.byref(self._ctx))
        if rc != 0:
            raise MerlHipError(rc, "mrl_init", self._lib.mrl_strerror(rc).decode())
        self.device = device
        name = C.create_string_buffer(256); cus = C.c_int(); mem = C.c_size_t()
        self._check(self._lib.mrl_device_info(self._ctx, name, 256, C.byref(cus), C.byref(mem)), "mrl_device_info")
        self.device_name, self.compute_units, self.total_mem = name.value.decode(), cus.value, mem.value

    # ---- plumbing ----
    def _check(self, rc: int, what: str):
        if rc != 0:
            detail = self._lib.mrl_last_error(self._ctx).decode() or self._lib.mrl_strerror(rc).decode()
            raise MerlHipError(rc, what, detail)

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.mrl_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, option: int, value: int):
        self._check(self._lib.mrl_set_option(self._ctx, option, value), "mrl_set_option")

    def get_option(self, option: int) -> int:
        v = C.c_int()
        self._check(self._lib.mrl_get_option(self._ctx, option, C.byref(v)), "mrl_get_option")
        return v.value

    def use_torch_stream(self):
        """Launch on torch's current stream so torch tensors and our kernels are ordered."""
        import torch
        s = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self._lib.mrl_set_stream(self._ctx, C.c_void_p(s)), "mrl_set_stream")

    def use_own_stream(self):
        self._check(self._lib.mrl_reset_stream(self._ctx), "mrl_reset_stream")

    def synchronize(self):
        self._check(self._lib.mrl_synchronize(self._ctx), "mrl_synchronize")

    def timer_start(self):
        self._check(self._lib.mrl_timer_start(self._ctx), "mrl_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._check(self._lib.mrl_timer_stop(self._ctx, C.byref(ms)), "mrl_timer_stop")
        return ms.value

    # ---- materials ----
    def load_merl(self, path: str) -> int:
        mid = C.c_int()
        self._check(self._lib.mrl_material_load_merl(self._ctx, path.encode(), C.byref(mid)), "mrl_material_load_merl")
        return mid.value

    def upload_merl(self, planar: np.ndarray) -> int:
        p = np.ascontiguousarray(planar, dtype=np.float64)
        if p.size != 3 * 90 * 90 * 180:
            raise ValueError("upload_merl needs 3 x 90 x 90 x 180 values")
        mid = C.c_int()
        self._check(self._lib.mrl_material_upload_f64(self._ctx, p.ctypes.data, C.byref(mid)), "mrl_material_upload_f64")
        return mid.value

    def upload_table(self, planar: np.ndarray, scale: Sequence[float] = (1.0, 1.0, 1.0)) -> int:
        p = np.ascontiguousarray(planar, dtype=np.float64)
        if p.ndim != 4 or p.shape[0] != 3:
            raise ValueError("upload_table needs a (3, n_th, n_td, n_pd) array")
        dims = (C.c_int * 3)(*p.shape[1:]); sc = (C.c_double * 3)(*scale); mid = C.c_int()
        self._check(self._lib.mrl_material_upload_table(self._ctx, p.ctypes.data, dims, sc, C.byref(mid)), "mrl_material_upload_table")
        return mid.value

    def load_table(self, path: str, scale: Sequence[float] = (1.0, 1.0, 1.0)) -> int:
        sc = (C.c_double * 3)(*scale); mid = C.c_int()
        self._check(self._lib.mrl_material_load_table(self._ctx, path.encode(), sc, C.byref(mid)), "mrl_material_load_table")
        return mid.value

    def ggx(self, alpha: float, eta: Sequence[float], k: Sequence[float]) -> int:
        mid = C.c_int()
        self._check(self._lib.mrl_material_ggx(self._ctx, alpha, (C.c_float * 3)(*eta), (C.c_float * 3)(*k), C.byref(mid)), "mrl_material_ggx")
        return mid.value

    def upload_rgl(self, fields: dict) -> int:
        """The adaptive-parameterisation measured BSDF from the fields of an RGL *.bsdf file (dict of arrays: phi_i, theta_i,
        ndf, sigma, vndf, luminance, rgb[, jacobian]).  Single-material calls evaluate it (mrl_material_upload_rgl)."""
        r, keep = rgl_fields_struct(fields)
        mid = C.c_int()
        if isinstance(r, RglSpectralFields):       # "spectra" + "wavelengths": a spectral material (the *_spectral calls evaluate it)
            self._check(self._lib.mrl_material_upload_rgl_spectral(self._ctx, C.byref(r), C.byref(mid)), "mrl_material_upload_rgl_spectral")
        else:
            self._check(self._lib.mrl_material_upload_rgl(self._ctx, C.byref(r), C.byref(mid)), "mrl_material_upload_rgl")
        return mid.value

    def wavelengths(self, mid: int) -> np.ndarray:
        """the wavelength grid of a spectral RGL material"""
        n = C.c_int()
        self._check(self._lib.mrl_material_wavelengths(self._ctx, mid, C.byref(n), None, 0), "mrl_material_wavelengths")
        out = np.empty(n.value, np.float32)
        self._check(self._lib.mrl_material_wavelengths(self._ctx, mid, C.byref(n), out.ctypes.data_as(C.POINTER(C.c_float)), out.size), "mrl_material_wavelengths")
        return out

    def eval_sample_spectral(self, wi, wo, u, wavelengths, material: int, n_wavelengths: Optional[int] = None):
        """A spectral RGL material at per-unit wavelengths [n, W] (None: the file's own nodes, n_wavelengths = their number):
        (values [n, W], pdf, wo', pdf', weight' [n, W])."""
        n = int(wi.shape[0]); self._prep(wi)
        W = int(wavelengths.shape[1]) if wavelengths is not None else int(n_wavelengths)
        out = (self._empty(wi, (n, W)), self._empty(wi, (n,)), self._empty(wi, (n, 3)), self._empty(wi, (n,)), self._empty(wi, (n, W)))
        val, pdf, wo2, pdf2, w = out
        self._check(self._lib.mrl_eval_sample_spectral_batch(
            self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"), _addr(u, np.float32, 2, n, "u"),
            _addr(wavelengths, np.float32, W, n, "wavelengths"), W, material, n,
            _addr(val, np.float32, W, n, "out_values"), _addr(pdf, np.float32, None, n, "out_pdf"), _addr(wo2, np.float32, 3, n, "out_wo"),
            _addr(pdf2, np.float32, None, n, "out_pdf2"), _addr(w, np.float32, W, n, "out_weight")), "mrl_eval_sample_spectral_batch")
        return out

    def eval_spectral(self, wi, wo, wavelengths, material: int, n_wavelengths: Optional[int] = None, with_pdf: bool = False):
        n = int(wi.shape[0]); self._prep(wi)
        W = int(wavelengths.shape[1]) if wavelengths is not None else int(n_wavelengths)
        val = self._empty(wi, (n, W))
        args = (self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"), _addr(wavelengths, np.float32, W, n, "wavelengths"), W, material, n,
                _addr(val, np.float32, W, n, "out_values"))
        if with_pdf:
            pdf = self._empty(wi, (n,))
            self._check(self._lib.mrl_eval_pdf_spectral_batch(*args, _addr(pdf, np.float32, None, n, "out_pdf")), "mrl_eval_pdf_spectral_batch")
            return val, pdf
        self._check(self._lib.mrl_eval_spectral_batch(*args), "mrl_eval_spectral_batch")
        return val

    def sample_spectral(self, wi, u, wavelengths, material: int, n_wavelengths: Optional[int] = None):
        n = int(wi.shape[0]); self._prep(wi)
        W = int(wavelengths.shape[1]) if wavelengths is not None else int(n_wavelengths)
        wo2, pdf2, w = self._empty(wi, (n, 3)), self._empty(wi, (n,)), self._empty(wi, (n, W))
        self._check(self._lib.mrl_sample_spectral_batch(
            self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(u, np.float32, 2, n, "u"), _addr(wavelengths, np.float32, W, n, "wavelengths"), W, material, n,
            _addr(wo2, np.float32, 3, n, "out_wo"), _addr(pdf2, np.float32, None, n, "out_pdf"), _addr(w, np.float32, W, n, "out_weight")), "mrl_sample_spectral_batch")
        return wo2, pdf2, w

    def load_rgl(self, path: str) -> int:
        """An RGL *.bsdf file (tensor_file container with the RGL field names; the *_rgb variant)."""
        mid = C.c_int()
        rc = self._lib.mrl_material_load_rgl(self._ctx, path.encode(), C.byref(mid))
        if rc != 0:
            detail = self._lib.mrl_tensor_file_last_error(None).decode() or self._lib.mrl_last_error(self._ctx).decode()
            raise MerlHipError(rc, "mrl_material_load_rgl", detail)
        return mid.value

    def save_image(self, mid: int, path: str):
        """The material's device image to disk (mrl_material_save_image): tables, n-channel tables, RGL materials."""
        self._check(self._lib.mrl_material_save_image(self._ctx, mid, path.encode()), "mrl_material_save_image")

    def load_image(self, path: str) -> int:
        mid = C.c_int()
        self._check(self._lib.mrl_material_load_image(self._ctx, path.encode(), C.byref(mid)), "mrl_material_load_image")
        return mid.value

    def material_count(self) -> int:
        return self._lib.mrl_material_count(self._ctx)

    def material_info(self, mid: int):
        kind = C.c_int(); dims = (C.c_int * 3)()
        self._check(self._lib.mrl_material_info(self._ctx, mid, C.byref(kind), dims), "mrl_material_info")
        return kind.value, tuple(dims)

    def scalar_eval_sample(self, wi, wo, u, material: int = 0) -> np.ndarray:
        """ONE fused unit through the scalar service (mrl_scalar_eval_sample): no launch per call.  Returns 11 floats:
        rgb[3] pdf wo'[3] pdf' weight'[3] — what eval_sample returns for the unit."""
        a = (C.c_float * 3)(*[float(x) for x in wi]); b = (C.c_float * 3)(*[float(x) for x in wo]); c = (C.c_float * 2)(*[float(x) for x in u])
        out = (C.c_float * 11)()
        self._check(self._lib.mrl_scalar_eval_sample(self._ctx, int(material), a, b, c, out), "mrl_scalar_eval_sample")
        return np.frombuffer(out, dtype=np.float32).copy()

    def scalar_eval_pdf(self, wi, wo, material: int = 0):
        """ONE eval + pdf through the scalar service (no sample half): (rgb[3], pdf)."""
        a = (C.c_float * 3)(*[float(x) for x in wi]); b = (C.c_float * 3)(*[float(x) for x in wo])
        rgb = (C.c_float * 3)(); pdf = C.c_float()
        self._check(self._lib.mrl_scalar_eval_pdf(self._ctx, int(material), a, b, rgb, C.byref(pdf)), "mrl_scalar_eval_pdf")
        return np.frombuffer(rgb, dtype=np.float32).copy(), float(pdf.value)

    def scalar_sample(self, wi, u, material: int = 0):
        """ONE sample through the scalar service (no eval half): (wo'[3], pdf', weight'[3])."""
        a = (C.c_float * 3)(*[float(x) for x in wi]); c = (C.c_float * 2)(*[float(x) for x in u])
        wo = (C.c_float * 3)(); pdf = C.c_float(); w = (C.c_float * 3)()
        self._check(self._lib.mrl_scalar_sample(self._ctx, int(material), a, c, wo, C.byref(pdf), w), "mrl_scalar_sample")
        return np.frombuffer(wo, dtype=np.float32).copy(), float(pdf.value), np.frombuffer(w, dtype=np.float32).copy()

    def material_sampling2d(self, material: int = 0) -> np.ndarray:
        """The conditional sampling table P(theta_h | theta_i) as the device built it: [n_ti, 2 n_th + 1] doubles (cdf | c)."""
        n_ti, n_th = C.c_int(), C.c_int()
        self._lib.mrl_material_sampling2d.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
        self._check(self._lib.mrl_material_sampling2d(self._ctx, int(material), C.byref(n_ti), C.byref(n_th), None, 0), "mrl_material_sampling2d")
        out = np.empty((n_ti.value, 2 * n_th.value + 1), np.float64)
        self._check(self._lib.mrl_material_sampling2d(self._ctx, int(material), None, None, out.ctypes.data, out.size), "mrl_material_sampling2d")
        return out

    # ---- one-unit calls on the calling CPU thread (mrl_host_*) ----
    def host_table(self, material: int = 0) -> "HostTable":
        """A host image of a resident three-channel table (mrl_material_host_table): one-unit calls on it run on the
        calling CPU thread with the kernels' own per-unit functions compiled for the host."""
        h = C.c_void_p()
        self._check(self._lib.mrl_material_host_table(self._ctx, int(material), C.byref(h)), "mrl_material_host_table")
        return HostTable(self._lib, h)

    # ---- n-channel tables ----
    def upload_table_nch(self, planar: np.ndarray, scale: Optional[Sequence[float]] = None) -> int:
        """planar: (n_channels, n_th, n_td, n_pd) f64."""
        p = np.ascontiguousarray(planar, dtype=np.float64)
        if p.ndim != 4:
            raise ValueError("upload_table_nch needs a (n_channels, n_th, n_td, n_pd) array")
        c = int(p.shape[0])
        dims = (C.c_int * 3)(*p.shape[1:]); mid = C.c_int()
        sc = None if scale is None else (C.c_double * c)(*scale)
        self._check(self._lib.mrl_material_upload_table_nch(self._ctx, p.ctypes.data, dims, c, sc, C.byref(mid)), "mrl_material_upload_table_nch")
        return mid.value

    def upload_table_param(self, planar: np.ndarray, param: int, scale: Optional[Sequence[float]] = None) -> int:
        """planar: (n_channels, n_0, n_1, n_2) f64 indexed by the angles of `param` (PARAM_*); 3 channels = the RGB path.
        The context's OPT_TABLE_PARAM stays as it is."""
        p = np.ascontiguousarray(planar, dtype=np.float64)
        if p.ndim != 4:
            raise ValueError("upload_table_param needs a (n_channels, n_0, n_1, n_2) array")
        c = int(p.shape[0])
        dims = (C.c_int * 3)(*p.shape[1:]); mid = C.c_int()
        sc = None if scale is None else (C.c_double * c)(*scale)
        self._check(self._lib.mrl_material_upload_table_param(self._ctx, p.ctypes.data, dims, c, sc, int(param), C.byref(mid)), "mrl_material_upload_table_param")
        return mid.value

    def load_table_nch(self, path: str, n_channels: int, scale: Optional[Sequence[float]] = None) -> int:
        sc = None if scale is None else (C.c_double * n_channels)(*scale); mid = C.c_int()
        self._check(self._lib.mrl_material_load_table_nch(self._ctx, path.encode(), n_channels, sc, C.byref(mid)), "mrl_material_load_table_nch")
        return mid.value

    def material_channels(self, mid: int) -> int:
        c = C.c_int()
        self._check(self._lib.mrl_material_channels(self._ctx, mid, C.byref(c)), "mrl_material_channels")
        return c.value

    def material_param(self, mid: int) -> int:
        """enum mrl_param of a table material (the MRL_OPT_TABLE_PARAM in force when it was uploaded)."""
        c = C.c_int()
        self._check(self._lib.mrl_material_param(self._ctx, mid, C.byref(c)), "mrl_material_param")
        return c.value

    def eval_nch(self, wi, wo, n_channels: int, mat=None, material: int = 0, out=None):
        n = int(wi.shape[0]); self._prep(wi)
        out = self._empty(wi, (n, n_channels)) if out is None else out
        self._check(self._lib.mrl_eval_batch_nch(self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"),
                                                 _addr(mat, np.int32, None, n, "mat"), material, n, n_channels,
                                                 _addr(out, np.float32, n_channels, n, "out_values")), "mrl_eval_batch_nch")
        return out

    def sample_nch(self, wi, u, n_channels: int, mat=None, material: int = 0):
        n = int(wi.shape[0]); self._prep(wi)
        wo, pdf, w = self._empty(wi, (n, 3)), self._empty(wi, (n,)), self._empty(wi, (n, n_channels))
        self._check(self._lib.mrl_sample_batch_nch(self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(u, np.float32, 2, n, "u"),
                                                   _addr(mat, np.int32, None, n, "mat"), material, n, n_channels,
                                                   _addr(wo, np.float32, 3, n, "out_wo"), _addr(pdf, np.float32, None, n, "out_pdf"),
                                                   _addr(w, np.float32, n_channels, n, "out_weight")), "mrl_sample_batch_nch")
        return wo, pdf, w

    def eval_pdf_nch(self, wi, wo, n_channels: int, mat=None, material: int = 0):
        n = int(wi.shape[0]); self._prep(wi)
        val, pdf = self._empty(wi, (n, n_channels)), self._empty(wi, (n,))
        self._check(self._lib.mrl_eval_pdf_batch_nch(self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"),
                                                     _addr(mat, np.int32, None, n, "mat"), material, n, n_channels,
                                                     _addr(val, np.float32, n_channels, n, "out_values"), _addr(pdf, np.float32, None, n, "out_pdf")),
                    "mrl_eval_pdf_batch_nch")
        return val, pdf

    def eval_sample_nch(self, wi, wo, u, n_channels: int, mat=None, material: int = 0):
        """Returns (values[n, C], pdf, wo', pdf', weight'[n, C])."""
        n = int(wi.shape[0]); self._prep(wi)
        out = (self._empty(wi, (n, n_channels)), self._empty(wi, (n,)), self._empty(wi, (n, 3)), self._empty(wi, (n,)),
               self._empty(wi, (n, n_channels)))
        val, pdf, wo2, pdf2, w = out
        self._check(self._lib.mrl_eval_sample_batch_nch(
            self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"), _addr(u, np.float32, 2, n, "u"),
            _addr(mat, np.int32, None, n, "mat"), material, n, n_channels,
            _addr(val, np.float32, n_channels, n, "out_values"), _addr(pdf, np.float32, None, n, "out_pdf"),
            _addr(wo2, np.float32, 3, n, "out_wo"), _addr(pdf2, np.float32, None, n, "out_pdf2"),
            _addr(w, np.float32, n_channels, n, "out_weight")), "mrl_eval_sample_batch_nch")
        return out

    def load_tensor_table(self, path: str, field: Optional[str] = None):
        """A customized_measurement table stored in a tensor_file container.  Returns (material id, channels)."""
        mid, ch = C.c_int(), C.c_int()
        rc = self._lib.mrl_material_load_tensor_table(self._ctx, path.encode(), None if field is None else field.encode(), C.byref(mid), C.byref(ch))
        if rc != 0:
            detail = self._lib.mrl_tensor_file_last_error(None).decode() or self._lib.mrl_last_error(self._ctx).decode()
            raise MerlHipError(rc, "mrl_material_load_tensor_table", detail)
        return mid.value, ch.value

    def eval_sample_queue_nch(self, wi, wo, u, queue, count, n_channels: int, mat=None, material: int = 0, capacity=None, out=None):
        """Fused n-channel unit over a wavefront queue; unqueued slots of `out` stay as they are."""
        n = int(wi.shape[0]); q, c, cap = self._queue(wi, queue, count, capacity)
        if out is None:
            out = (self._zeros(wi, (n, n_channels)), self._zeros(wi, (n,)), self._zeros(wi, (n, 3)), self._zeros(wi, (n,)),
                   self._zeros(wi, (n, n_channels)))
        val, pdf, wo2, pdf2, w = out
        self._check(self._lib.mrl_eval_sample_queue_nch(
            self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"), _addr(u, np.float32, 2, n, "u"),
            _addr(mat, np.int32, None, n, "mat"), material, q, c, cap, n_channels,
            _addr(val, np.float32, n_channels, n, "out_values"), _addr(pdf, np.float32, None, n, "out_pdf"),
            _addr(wo2, np.float32, 3, n, "out_wo"), _addr(pdf2, np.float32, None, n, "out_pdf2"),
            _addr(w, np.float32, n_channels, n, "out_weight")), "mrl_eval_sample_queue_nch")
        return out

    def eval_queue_nch(self, wi, wo, queue, count, n_channels: int, mat=None, material: int = 0, capacity=None, out=None):
        n = int(wi.shape[0]); q, c, cap = self._queue(wi, queue, count, capacity)
        out = self._zeros(wi, (n, n_channels)) if out is None else out
        self._check(self._lib.mrl_eval_queue_nch(self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"),
                                                 _addr(mat, np.int32, None, n, "mat"), material, q, c, cap, n_channels,
                                                 _addr(out, np.float32, n_channels, n, "out_values")), "mrl_eval_queue_nch")
        return out

    def sample_queue_nch(self, wi, u, queue, count, n_channels: int, mat=None, material: int = 0, capacity=None):
        n = int(wi.shape[0]); q, c, cap = self._queue(wi, queue, count, capacity)
        wo, pdf, w = self._zeros(wi, (n, 3)), self._zeros(wi, (n,)), self._zeros(wi, (n, n_channels))
        self._check(self._lib.mrl_sample_queue_nch(self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(u, np.float32, 2, n, "u"),
                                                   _addr(mat, np.int32, None, n, "mat"), material, q, c, cap, n_channels,
                                                   _addr(wo, np.float32, 3, n, "out_wo"), _addr(pdf, np.float32, None, n, "out_pdf"),
                                                   _addr(w, np.float32, n_channels, n, "out_weight")), "mrl_sample_queue_nch")
        return wo, pdf, w

    def eval_pdf_queue_nch(self, wi, wo, queue, count, n_channels: int, mat=None, material: int = 0, capacity=None):
        n = int(wi.shape[0]); q, c, cap = self._queue(wi, queue, count, capacity)
        val, pdf = self._zeros(wi, (n, n_channels)), self._zeros(wi, (n,))
        self._check(self._lib.mrl_eval_pdf_queue_nch(self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"),
                                                     _addr(mat, np.int32, None, n, "mat"), material, q, c, cap, n_channels,
                                                     _addr(val, np.float32, n_channels, n, "out_values"), _addr(pdf, np.float32, None, n, "out_pdf")),
                    "mrl_eval_pdf_queue_nch")
        return val, pdf

    def release_material(self, mid: int):
        """Frees the material's device memory; its id becomes a tombstone (batch calls render it as zeros)."""
        self._check(self._lib.mrl_material_release(self._ctx, mid), "mrl_material_release")

    def memory_info(self) -> dict:
        v = [C.c_size_t() for _ in range(4)]
        self._check(self._lib.mrl_memory_info(self._ctx, *[C.byref(x) for x in v]), "mrl_memory_info")
        return {"table_bytes": v[0].value, "workspace_bytes": v[1].value, "device_free": v[2].value, "device_total": v[3].value}

    # ---- batches ----
    def _empty(self, like, shape):
        if _is_tensor(like):
            import torch
            return torch.empty(shape, dtype=torch.float32, device=like.device)
        return np.empty(shape, dtype=np.float32)

    def _prep(self, first):
        if _is_tensor(first):
            if not first.is_cuda:
                raise ValueError("torch tensors must live on the GPU (pass numpy arrays for host data)")
            self.use_torch_stream()

    def eval(self, wi, wo, mat=None, material: int = 0, out=None):
        n = int(wi.shape[0]); self._prep(wi)
        out = self._empty(wi, (n, 3)) if out is None else out
        self._check(self._lib.mrl_eval_batch(self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"),
                                             _addr(mat, np.int32, None, n, "mat"), material, n,
                                             _addr(out, np.float32, 3, n, "out_rgb")), "mrl_eval_batch")
        return out

    def pdf(self, wi, wo, mat=None, material: int = 0, out=None):
        n = int(wi.shape[0]); self._prep(wi)
        out = self._empty(wi, (n,)) if out is None else out
        self._check(self._lib.mrl_pdf_batch(self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"),
                                            _addr(mat, np.int32, None, n, "mat"), material, n,
                                            _addr(out, np.float32, None, n, "out_pdf")), "mrl_pdf_batch")
        return out

    def eval_pdf(self, wi, wo, mat=None, material: int = 0, out=None):
        """eval and pdf of the same pairs in one launch (Mitsuba 3's eval_pdf).  Returns (rgb, pdf)."""
        n = int(wi.shape[0]); self._prep(wi)
        rgb, pdf = out if out is not None else (self._empty(wi, (n, 3)), self._empty(wi, (n,)))
        self._check(self._lib.mrl_eval_pdf_batch(self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"),
                                                 _addr(mat, np.int32, None, n, "mat"), material, n,
                                                 _addr(rgb, np.float32, 3, n, "out_rgb"), _addr(pdf, np.float32, None, n, "out_pdf")),
                    "mrl_eval_pdf_batch")
        return rgb, pdf

    def sample(self, wi, u, mat=None, material: int = 0, out=None):
        n = int(wi.shape[0]); self._prep(wi)
        wo, pdf, w = out if out is not None else (self._empty(wi, (n, 3)), self._empty(wi, (n,)), self._empty(wi, (n, 3)))
        self._check(self._lib.mrl_sample_batch(self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(u, np.float32, 2, n, "u"),
                                               _addr(mat, np.int32, None, n, "mat"), material, n,
                                               _addr(wo, np.float32, 3, n, "out_wo"), _addr(pdf, np.float32, None, n, "out_pdf"),
                                               _addr(w, np.float32, 3, n, "out_weight")), "mrl_sample_batch")
        return wo, pdf, w

    def eval_sample(self, wi, wo, u, mat=None, material: int = 0, out=None):
        """The benchmarked unit.  Returns (rgb, pdf, wo', pdf', weight')."""
        n = int(wi.shape[0]); self._prep(wi)
        if out is None:
            out = (self._empty(wi, (n, 3)), self._empty(wi, (n,)), self._empty(wi, (n, 3)), self._empty(wi, (n,)), self._empty(wi, (n, 3)))
        rgb, pdf, wo2, pdf2, w = out
        self._check(self._lib.mrl_eval_sample_batch(
            self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"), _addr(u, np.float32, 2, n, "u"),
            _addr(mat, np.int32, None, n, "mat"), material, n,
            _addr(rgb, np.float32, 3, n, "out_rgb"), _addr(pdf, np.float32, None, n, "out_pdf"),
            _addr(wo2, np.float32, 3, n, "out_wo"), _addr(pdf2, np.float32, None, n, "out_pdf2"),
            _addr(w, np.float32, 3, n, "out_weight")), "mrl_eval_sample_batch")
        return out

    # ---- wavefront queues (device tensors only) ----
    def _zeros(self, like, shape):
        import torch
        return torch.zeros(shape, dtype=torch.float32, device=like.device)

    def _queue(self, wi, queue, count, capacity):
        """queue: int32 tensor of slot indices (non-negative; read as uint32), count: int32 tensor with one element."""
        if not (_is_tensor(wi) and _is_tensor(queue) and _is_tensor(count)):
            raise ValueError("queue calls take GPU tensors")
        self._prep(wi)
        cap = int(queue.shape[0]) if capacity is None else int(capacity)
        if cap > int(queue.shape[0]):
            raise ValueError("capacity exceeds the queue's length")
        return _addr(queue, np.int32, None, None, "queue"), _addr(count, np.int32, None, 1, "count"), cap

    def partition_by_material(self, mat):
        """Stable partition of the slots by material id.  Returns (queue, offsets, counts) as int32 GPU tensors:
        material m owns queue[offsets[m]:offsets[m + 1]] (ascending slot indices), counts[m] of them."""
        import torch
        if not (_is_tensor(mat) and mat.is_cuda):
            raise ValueError("partition_by_material takes a GPU int32 tensor")
        self._prep(mat)
        n = int(mat.shape[0]); k = self.material_count()
        queue = torch.empty((n,), dtype=torch.int32, device=mat.device)
        offsets = torch.zeros((k + 1,), dtype=torch.int32, device=mat.device)
        counts = torch.zeros((k,), dtype=torch.int32, device=mat.device)
        self._check(self._lib.mrl_partition_by_material(self._ctx, _addr(mat, np.int32, None, n, "mat"), n, _addr(queue, np.int32, None, n, "queue"),
                                                        _addr(offsets, np.int32, None, k + 1, "offsets"), _addr(counts, np.int32, None, k, "counts")),
                    "mrl_partition_by_material")
        return queue, offsets, counts

    def eval_queue(self, wi, wo, queue, count, mat=None, material: int = 0, capacity=None, out=None):
        """eval() of the slots queue[0 .. min(count, capacity)); other slots of `out` stay as they are."""
        n = int(wi.shape[0]); q, c, cap = self._queue(wi, queue, count, capacity)
        out = self._zeros(wi, (n, 3)) if out is None else out
        self._check(self._lib.mrl_eval_queue(self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"),
                                             _addr(mat, np.int32, None, n, "mat"), material, q, c, cap,
                                             _addr(out, np.float32, 3, n, "out_rgb")), "mrl_eval_queue")
        return out

    def pdf_queue(self, wi, wo, queue, count, mat=None, material: int = 0, capacity=None, out=None):
        n = int(wi.shape[0]); q, c, cap = self._queue(wi, queue, count, capacity)
        out = self._zeros(wi, (n,)) if out is None else out
        self._check(self._lib.mrl_pdf_queue(self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"),
                                            _addr(mat, np.int32, None, n, "mat"), material, q, c, cap,
                                            _addr(out, np.float32, None, n, "out_pdf")), "mrl_pdf_queue")
        return out

    def eval_pdf_queue(self, wi, wo, queue, count, mat=None, material: int = 0, capacity=None, out=None):
        n = int(wi.shape[0]); q, c, cap = self._queue(wi, queue, count, capacity)
        rgb, pdf = out if out is not None else (self._zeros(wi, (n, 3)), self._zeros(wi, (n,)))
        self._check(self._lib.mrl_eval_pdf_queue(self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"),
                                                 _addr(mat, np.int32, None, n, "mat"), material, q, c, cap,
                                                 _addr(rgb, np.float32, 3, n, "out_rgb"), _addr(pdf, np.float32, None, n, "out_pdf")),
                    "mrl_eval_pdf_queue")
        return rgb, pdf

    def sample_queue(self, wi, u, queue, count, mat=None, material: int = 0, capacity=None, out=None):
        n = int(wi.shape[0]); q, c, cap = self._queue(wi, queue, count, capacity)
        wo, pdf, w = out if out is not None else (self._zeros(wi, (n, 3)), self._zeros(wi, (n,)), self._zeros(wi, (n, 3)))
        self._check(self._lib.mrl_sample_queue(self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(u, np.float32, 2, n, "u"),
                                               _addr(mat, np.int32, None, n, "mat"), material, q, c, cap,
                                               _addr(wo, np.float32, 3, n, "out_wo"), _addr(pdf, np.float32, None, n, "out_pdf"),
                                               _addr(w, np.float32, 3, n, "out_weight")), "mrl_sample_queue")
        return wo, pdf, w

    def eval_sample_queue(self, wi, wo, u, queue, count, mat=None, material: int = 0, capacity=None, out=None):
        n = int(wi.shape[0]); q, c, cap = self._queue(wi, queue, count, capacity)
        if out is None:
            out = (self._zeros(wi, (n, 3)), self._zeros(wi, (n,)), self._zeros(wi, (n, 3)), self._zeros(wi, (n,)), self._zeros(wi, (n, 3)))
        rgb, pdf, wo2, pdf2, w = out
        self._check(self._lib.mrl_eval_sample_queue(
            self._ctx, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"), _addr(u, np.float32, 2, n, "u"),
            _addr(mat, np.int32, None, n, "mat"), material, q, c, cap,
            _addr(rgb, np.float32, 3, n, "out_rgb"), _addr(pdf, np.float32, None, n, "out_pdf"),
            _addr(wo2, np.float32, 3, n, "out_wo"), _addr(pdf2, np.float32, None, n, "out_pdf2"),
            _addr(w, np.float32, 3, n, "out_weight")), "mrl_eval_sample_queue")
        return out

    # ---- synthetic inputs (device) ----
    def generate_pairs(self, seed: int, first: int, n: int, out=None):
        import torch
        dev = torch.device("cuda", self.device)
        if out is None:
            out = (torch.empty((n, 3), dtype=torch.float32, device=dev), torch.empty((n, 3), dtype=torch.float32, device=dev),
                   torch.empty((n, 2), dtype=torch.float32, device=dev))
        wi, wo, u = out
        self.use_torch_stream()
        self._check(self._lib.mrl_generate_pairs(self._ctx, seed, first, n, _addr(wi, np.float32, 3, n, "wi"),
                                                 _addr(wo, np.float32, 3, n, "wo"), _addr(u, np.float32, 2, n, "u")), "mrl_generate_pairs")
        return out

    def generate_materials(self, seed: int, first: int, n: int, n_materials: int, out=None):
        import torch
        if out is None:
            out = torch.empty((n,), dtype=torch.int32, device=torch.device("cuda", self.device))
        self.use_torch_stream()
        self._check(self._lib.mrl_generate_materials(self._ctx, seed, first, n, n_materials, _addr(out, np.int32, None, n, "mat")),
                    "mrl_generate_materials")
        return out


TENSOR_DTYPES = {1: "int8/uint8", 2: "int8/uint8", 3: "int16/uint16", 4: "int16/uint16", 5: "int32/uint32", 6: "int32/uint32",
                 7: "int64/uint64", 8: "int64/uint64", 9: "float16", 10: "float32", 11: "float64"}


def read_tensor_file(path: str) -> dict:
    """Every field of a tensor_file container through the library's reader (host code; needs no GPU):
    {name: ndarray}; float fields come back as float64, integer fields as raw bytes reshaped to the field's shape."""
    L = load_library()
    f = C.c_void_p()
    rc = L.mrl_tensor_file_open(path.encode(), C.byref(f))
    if rc != 0:
        raise MerlHipError(rc, "mrl_tensor_file_open", L.mrl_tensor_file_last_error(None).decode())
    try:
        out = {}
        for i in range(L.mrl_tensor_file_field_count(f)):
            name, dtype, ndim, shape = C.c_char_p(), C.c_int(), C.c_int(), C.POINTER(C.c_uint64)()
            assert L.mrl_tensor_file_field_info(f, i, C.byref(name), C.byref(dtype), C.byref(ndim), C.byref(shape)) == 0
            shp = tuple(int(shape[d]) for d in range(ndim.value))
            count = int(np.prod(shp)) if shp else 1
            if dtype.value >= 9:
                a = np.empty(count, np.float64)
                assert L.mrl_tensor_file_read_f64(f, i, a.ctypes.data, count) == 0
                out[name.value.decode()] = a.reshape(shp)
            else:
                nbytes = C.c_size_t()
                ptr = L.mrl_tensor_file_field_data(f, i, C.byref(nbytes))
                raw = np.frombuffer(C.string_at(ptr, nbytes.value), dtype=np.uint8).copy()
                width = nbytes.value // max(count, 1)
                out[name.value.decode()] = raw.view({1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[width]).reshape(shp)
        return out
    finally:
        L.mrl_tensor_file_close(f)


def build_info() -> str:
    """mrl_build_info: "sources <hash>" of the loaded library."""
    L = load_library()
    L.mrl_build_info.restype = C.c_char_p
    return L.mrl_build_info().decode()


def tile_bounds(n_total: int, world: int, rank: int):
    """mrl_tile_bounds through the library (pure arithmetic: needs no GPU)."""
    lo, hi = C.c_size_t(), C.c_size_t()
    load_library().mrl_tile_bounds(n_total, world, rank, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def chunk_bounds(n_total: int, world: int, rank: int, chunk: int, step: int):
    lo, hi = C.c_size_t(), C.c_size_t()
    load_library().mrl_chunk_bounds(n_total, world, rank, chunk, step, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def chunk_steps(n_total: int, world: int, chunk: int) -> int:
    return int(load_library().mrl_chunk_steps(n_total, world, chunk))


class PlanOp(C.Structure):
    """mrl_plan_op: one operation of a sharded call's schedule (include/merl_hip.h)."""
    _fields_ = [("kind", C.c_int), ("member", C.c_int), ("buffer", C.c_int), ("step", C.c_size_t), ("first", C.c_size_t),
                ("count", C.c_size_t), ("tile_offset", C.c_size_t), ("after_transfer_of_step", C.c_longlong)]


PLAN_COMPUTE, PLAN_TRANSFER = 0, 1


def group_plan(n_total: int, world: int, chunk: int, root: int):
    """mrl_group_plan: the operations of one sharded call in issue order (pure arithmetic: needs no GPU)."""
    L = load_library()
    L.mrl_group_plan.restype = C.c_size_t
    L.mrl_group_plan.argtypes = [C.c_size_t, C.c_int, C.c_size_t, C.c_int, C.POINTER(PlanOp), C.c_size_t]
    n = L.mrl_group_plan(n_total, world, chunk, root, None, 0)
    ops = (PlanOp * max(n, 1))()
    got = L.mrl_group_plan(n_total, world, chunk, root, ops, n)
    assert got == n
    return [ops[i] for i in range(n)]


class LinkReport(C.Structure):
    """mrl_link_report: one peer -> root link of mrl_group_link_test."""
    _fields_ = [("peer", C.c_int), ("ok", C.c_int), ("bytes", C.c_size_t), ("mismatches", C.c_size_t), ("ms", C.c_float), ("GBps", C.c_float)]


class MerlGroup:
    """mrl_group: one host process, several GPUs (a device id may repeat: rehearsal with device copies)."""

    def link_test(self, nbytes: int, transport: int = TRANSPORT_AUTO, root: int = 0):
        """mrl_group_link_test: one payload from every peer to the root, timed and bit-checked; list of dicts."""
        n = int(self._lib.mrl_group_size(self._g))
        rep = (LinkReport * n)()
        self._lib.mrl_group_link_test.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(LinkReport)]
        rc = self._lib.mrl_group_link_test(self._g, nbytes, transport, root, rep)
        if rc != 0:
            raise MerlHipError(rc, "mrl_group_link_test", (self._lib.mrl_group_last_error(self._g) or b"").decode())
        return [{"peer": r.peer, "ok": bool(r.ok), "bytes": r.bytes, "mismatches": r.mismatches, "ms": r.ms, "GBps": r.GBps} for r in rep]

    def __init__(self, devices: Sequence[int], transport: int = TRANSPORT_AUTO):
        self._lib = load_library()
        self._g = C.c_void_p()
        ids = (C.c_int * len(devices))(*devices)
        rc = self._lib.mrl_group_init(len(devices), ids, transport, C.byref(self._g))
        if rc != 0:
            raise MerlHipError(rc, "mrl_group_init", self._lib.mrl_group_last_error(None).decode() or self._lib.mrl_strerror(rc).decode())
        self.devices = list(devices)
        self.size = len(devices)
        self.transport = self._lib.mrl_group_transport(self._g)

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise MerlHipError(rc, what, self._lib.mrl_group_last_error(self._g).decode() or self._lib.mrl_strerror(rc).decode())

    def close(self):
        if getattr(self, "_g", None) is not None and self._g:
            self._lib.mrl_group_destroy(self._g)
            self._g = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, option: int, value: int):
        self._check(self._lib.mrl_group_set_option(self._g, option, value), "mrl_group_set_option")

    def upload_merl(self, planar: np.ndarray) -> int:
        p = np.ascontiguousarray(planar, dtype=np.float64)
        mid = C.c_int()
        self._check(self._lib.mrl_group_material_upload_f64(self._g, p.ctypes.data, C.byref(mid)), "mrl_group_material_upload_f64")
        return mid.value

    def upload_rgl(self, fields: dict) -> int:
        r, keep = rgl_fields_struct(fields)
        mid = C.c_int()
        if isinstance(r, RglSpectralFields):
            self._check(self._lib.mrl_group_material_upload_rgl_spectral(self._g, C.byref(r), C.byref(mid)), "mrl_group_material_upload_rgl_spectral")
        else:
            self._check(self._lib.mrl_group_material_upload_rgl(self._g, C.byref(r), C.byref(mid)), "mrl_group_material_upload_rgl")
        return mid.value

    def upload_table(self, planar: np.ndarray, scale: Sequence[float] = (1.0, 1.0, 1.0)) -> int:
        p = np.ascontiguousarray(planar, dtype=np.float64)
        dims = (C.c_int * 3)(*p.shape[1:]); sc = (C.c_double * 3)(*scale); mid = C.c_int()
        self._check(self._lib.mrl_group_material_upload_table(self._g, p.ctypes.data, dims, sc, C.byref(mid)), "mrl_group_material_upload_table")
        return mid.value

    def ggx(self, alpha: float, eta: Sequence[float], k: Sequence[float]) -> int:
        mid = C.c_int()
        self._check(self._lib.mrl_group_material_ggx(self._g, alpha, (C.c_float * 3)(*eta), (C.c_float * 3)(*k), C.byref(mid)), "mrl_group_material_ggx")
        return mid.value

    def release_material(self, mid: int):
        self._check(self._lib.mrl_group_material_release(self._g, mid), "mrl_group_material_release")

    def generate_tiles(self, seed: int, first: int, n_total: int, n_materials: int = 0):
        tiles = (TileInputs * self.size)()
        self._check(self._lib.mrl_group_generate_tiles(self._g, seed, first, n_total, n_materials, tiles), "mrl_group_generate_tiles")
        return tiles

    def eval_sample_sharded(self, tiles, n_total: int, chunk: int, out, root: int = 0, material: int = 0):
        """out: five GPU tensors on the root member's device (n_total units).  Asynchronous; synchronize() waits."""
        rgb, pdf, wo2, pdf2, w = out
        self._check(self._lib.mrl_group_eval_sample_sharded(
            self._g, tiles, material, n_total, chunk, root,
            _addr(rgb, np.float32, 3, n_total, "out_rgb"), _addr(pdf, np.float32, None, n_total, "out_pdf"),
            _addr(wo2, np.float32, 3, n_total, "out_wo"), _addr(pdf2, np.float32, None, n_total, "out_pdf2"),
            _addr(w, np.float32, 3, n_total, "out_weight")), "mrl_group_eval_sample_sharded")
        return out

    def eval_sharded(self, tiles, n_total: int, chunk: int, out_rgb, root: int = 0, material: int = 0):
        """eval only: one GPU tensor (n_total x 3) on the root member's device; 12 B per unit cross the links."""
        self._check(self._lib.mrl_group_eval_sharded(self._g, tiles, material, n_total, chunk, root,
                                                     _addr(out_rgb, np.float32, 3, n_total, "out_rgb")), "mrl_group_eval_sharded")
        return out_rgb

    def eval_sample_host(self, wi, wo, u, mat=None, material: int = 0):
        """Host (numpy) arrays split over the members, staged concurrently; returns numpy outputs."""
        n = int(wi.shape[0])
        out = (np.empty((n, 3), np.float32), np.empty((n,), np.float32), np.empty((n, 3), np.float32),
               np.empty((n,), np.float32), np.empty((n, 3), np.float32))
        self._check(self._lib.mrl_group_eval_sample_batch(
            self._g, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"), _addr(u, np.float32, 2, n, "u"),
            _addr(mat, np.int32, None, n, "mat"), material, n,
            *[_addr(o, np.float32, c, n, "out") for o, c in zip(out, (3, None, 3, None, 3))]), "mrl_group_eval_sample_batch")
        return out

    def eval_host(self, wi, wo, mat=None, material: int = 0):
        n = int(wi.shape[0]); out = np.empty((n, 3), np.float32)
        self._check(self._lib.mrl_group_eval_batch(self._g, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"),
                                                   _addr(mat, np.int32, None, n, "mat"), material, n, _addr(out, np.float32, 3, n, "out_rgb")),
                    "mrl_group_eval_batch")
        return out

    def pdf_host(self, wi, wo, mat=None, material: int = 0):
        n = int(wi.shape[0]); out = np.empty((n,), np.float32)
        self._check(self._lib.mrl_group_pdf_batch(self._g, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"),
                                                  _addr(mat, np.int32, None, n, "mat"), material, n, _addr(out, np.float32, None, n, "out_pdf")),
                    "mrl_group_pdf_batch")
        return out

    def eval_pdf_host(self, wi, wo, mat=None, material: int = 0):
        n = int(wi.shape[0]); rgb, pdf = np.empty((n, 3), np.float32), np.empty((n,), np.float32)
        self._check(self._lib.mrl_group_eval_pdf_batch(self._g, _addr(wi, np.float32, 3, n, "wi"), _addr(wo, np.float32, 3, n, "wo"),
                                                       _addr(mat, np.int32, None, n, "mat"), material, n,
                                                       _addr(rgb, np.float32, 3, n, "out_rgb"), _addr(pdf, np.float32, None, n, "out_pdf")),
                    "mrl_group_eval_pdf_batch")
        return rgb, pdf

    def sample_host(self, wi, u, mat=None, material: int = 0):
        n = int(wi.shape[0]); wo, pdf, w = np.empty((n, 3), np.float32), np.empty((n,), np.float32), np.empty((n, 3), np.float32)
        self._check(self._lib.mrl_group_sample_batch(self._g, _addr(wi, np.float32, 3, n, "wi"), _addr(u, np.float32, 2, n, "u"),
                                                     _addr(mat, np.int32, None, n, "mat"), material, n,
                                                     _addr(wo, np.float32, 3, n, "out_wo"), _addr(pdf, np.float32, None, n, "out_pdf"),
                                                     _addr(w, np.float32, 3, n, "out_weight")), "mrl_group_sample_batch")
        return wo, pdf, w

    def synchronize(self):
        self._check(self._lib.mrl_group_synchronize(self._g), "mrl_group_synchronize")

    def last_timing(self):
        ms = (C.c_float * self.size)()
        self._check(self._lib.mrl_group_last_timing(self._g, ms), "mrl_group_last_timing")
        return list(ms)
