"""A small wavefront path tracer: the CALLER side of the BSDF hot path.

Mitsuba's integrators call ``bsdf->eval()/sample()`` once per path vertex; a wavefront
integrator instead advances all paths one bounce at a time and hands each material the queue of path
slots that hit it.  This module is that loop, reduced to what exercises the boundary: path state lives in
slot-indexed device arrays, each bounce builds a queue of the live slots WITHOUT a host round trip (the
queue length stays in device memory) and calls ``mrl_eval_sample_queue`` once for every material in the
scene — ``eval`` of the light direction and ``sample`` of the continuation direction in one launch.

Geometry (a sphere on a disc), camera and lights are closed-form torch expressions: plumbing, not the
product.  ``shade`` is pluggable so the tests can drive the same loop with the CPU oracle and compare
images; the shipped callers pass :class:`GpuShade`.
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass, field
from typing import Callable, Tuple

import torch

_M64 = (1 << 64) - 1


def _s64(x: int) -> int:
    """A 64-bit pattern as the signed value torch's int64 arithmetic wants."""
    x &= _M64
    return x - (1 << 64) if x >> 63 else x


def _lsr(z: torch.Tensor, k: int) -> torch.Tensor:
    return (z >> k) & ((1 << (64 - k)) - 1)


def hash_u01(counter: torch.Tensor, salt: int) -> torch.Tensor:
    """splitmix64 finaliser of (counter, salt) -> f32 uniform in [0, 1) (24 bits); wrapping int64 arithmetic."""
    z = counter + _s64((salt + 1) * 0x9E3779B97F4A7C15)
    z = (z ^ _lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    z = z ^ _lsr(z, 31)
    return _lsr(z, 40).to(torch.float32) * (1.0 / 16777216.0)


@dataclass
class Scene:
    """Unit sphere resting on a disc; material 0 on the sphere, 1 on the disc."""
    sphere_centre: Tuple[float, float, float] = (0.0, 0.0, 1.0)
    sphere_radius: float = 1.0
    disc_radius: float = 6.0
    cam_origin: Tuple[float, float, float] = (0.0, -5.0, 1.7)
    cam_target: Tuple[float, float, float] = (0.0, 0.0, 0.85)
    fov_deg: float = 34.0
    light_dir: Tuple[float, float, float] = (0.45, -0.5, 0.74)       # towards the light
    light_irradiance: Tuple[float, float, float] = (3.0, 2.9, 2.7)
    sky_zenith: Tuple[float, float, float] = (0.10, 0.16, 0.35)
    sky_horizon: Tuple[float, float, float] = (0.45, 0.50, 0.55)
    eps: float = 1e-4


@dataclass
class Stats:
    bounces: int = 0
    queued_units: int = 0
    shade_seconds: float = 0.0
    total_seconds: float = 0.0
    per_bounce: list = field(default_factory=list)


def _normalize(v: torch.Tensor) -> torch.Tensor:
    return v / torch.sqrt((v * v).sum(-1, keepdim=True))


def _vec(x, dev) -> torch.Tensor:
    return torch.tensor(x, dtype=torch.float32, device=dev)


def camera_rays(scene: Scene, width: int, height: int, sample: int, dev) -> Tuple[torch.Tensor, torch.Tensor]:
    """Pinhole camera, one jittered ray per pixel for sample index `sample`."""
    n = width * height
    pix = torch.arange(n, dtype=torch.int64, device=dev)
    px = (pix % width).to(torch.float32) + hash_u01(pix * 4096 + sample, 101)
    py = (pix // width).to(torch.float32) + hash_u01(pix * 4096 + sample, 102)
    o = _vec(scene.cam_origin, dev)
    fwd = _normalize(_vec(scene.cam_target, dev) - o)
    right = _normalize(torch.linalg.cross(fwd, _vec((0.0, 0.0, 1.0), dev)))
    up = torch.linalg.cross(right, fwd)
    half = math.tan(math.radians(scene.fov_deg) * 0.5)
    sx = (px / width * 2.0 - 1.0) * half * (width / height)
    sy = (1.0 - py / height * 2.0) * half
    d = _normalize(fwd[None, :] + sx[:, None] * right[None, :] + sy[:, None] * up[None, :])
    return o[None, :].expand(n, 3).contiguous(), d.contiguous()


def intersect(scene: Scene, o: torch.Tensor, d: torch.Tensor):
    """Nearest hit of rays (o, d) with the sphere and the disc: hit mask, distance, normal, material id."""
    c = _vec(scene.sphere_centre, o.device)
    oc = o - c
    b = (oc * d).sum(-1)
    disc = b * b - ((oc * oc).sum(-1) - scene.sphere_radius ** 2)
    sq = torch.sqrt(torch.clamp(disc, min=0.0))
    t0, t1 = -b - sq, -b + sq
    ts = torch.where(t0 > scene.eps, t0, t1)
    hit_s = (disc > 0.0) & (ts > scene.eps)
    tp = -o[:, 2] / torch.where(d[:, 2] == 0.0, torch.ones_like(d[:, 2]), d[:, 2])
    pp = o + tp[:, None] * d
    hit_p = (d[:, 2] < 0.0) & (tp > scene.eps) & ((pp[:, 0] ** 2 + pp[:, 1] ** 2) < scene.disc_radius ** 2)
    inf = torch.full_like(ts, float("inf"))
    ts = torch.where(hit_s, ts, inf)
    tp = torch.where(hit_p, tp, inf)
    sphere_first = ts < tp
    t = torch.minimum(ts, tp)
    hit = torch.isfinite(t)
    p = o + torch.where(hit, t, torch.zeros_like(t))[:, None] * d
    n_s = (p - c) / scene.sphere_radius
    n_p = torch.zeros_like(p); n_p[:, 2] = 1.0
    normal = torch.where(sphere_first[:, None], n_s, n_p)
    mat = torch.where(sphere_first, 0, 1).to(torch.int32)
    return hit, t, p, normal, mat


def frame(n: torch.Tensor):
    """Branch-free orthonormal basis (s, t, n) of a unit normal."""
    sign = torch.where(n[:, 2] >= 0.0, 1.0, -1.0).to(n.dtype)
    a = -1.0 / (sign + n[:, 2])
    b = n[:, 0] * n[:, 1] * a
    s = torch.stack([1.0 + sign * n[:, 0] * n[:, 0] * a, sign * b, -sign * n[:, 0]], -1)
    t = torch.stack([b, sign + n[:, 1] * n[:, 1] * a, -n[:, 1]], -1)
    return s, t


def to_local(v, s, t, n):
    return torch.stack([(v * s).sum(-1), (v * t).sum(-1), (v * n).sum(-1)], -1).contiguous()


def to_world(v, s, t, n):
    return v[:, 0:1] * s + v[:, 1:2] * t + v[:, 2:3] * n


def sky(scene: Scene, d: torch.Tensor) -> torch.Tensor:
    k = torch.clamp(d[:, 2:3], 0.0, 1.0)
    up = (1.0 - k) * _vec(scene.sky_horizon, d.device)[None, :] + k * _vec(scene.sky_zenith, d.device)[None, :]
    return torch.where(d[:, 2:3] >= 0.0, up, torch.full_like(up, 0.02))


def build_queue(active: torch.Tensor):
    """Live slots first, length on the device: no host synchronisation between queue building and the BSDF call."""
    order = torch.argsort((~active).to(torch.int8), stable=True).to(torch.int32)
    count = active.sum().to(torch.int32).reshape(1)
    return order, count


class GpuShade:
    """shade() through libmerl_hip's queue entry point (the shipped path)."""

    def __init__(self, gpu):
        self.gpu = gpu
        self._out = None

    def __call__(self, wi, wo, u, mat, queue, count):
        n = wi.shape[0]
        if self._out is None or self._out[0].shape[0] != n:
            z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=wi.device)
            self._out = (z(n, 3), z(n), z(n, 3), z(n), z(n, 3))
        return self.gpu.eval_sample_queue(wi, wo, u, queue, count, mat=mat, out=self._out)


def render(shade: Callable, width: int, height: int, spp: int = 4, max_depth: int = 4, scene: Scene = None,
           device: str = "cuda:0"):
    """Path-trace the scene.  shade(wi, wo, u, mat, queue, count) -> (rgb, pdf, wo', pdf', weight') over slot arrays;
    only the queued slots of its outputs are read.  Returns (image[h, w, 3] as a float32 tensor, Stats)."""
    scene = scene or Scene()
    dev = torch.device(device)
    n = width * height
    light = _normalize(_vec(scene.light_dir, dev))[None, :]
    irradiance = _vec(scene.light_irradiance, dev)[None, :]
    image = torch.zeros((n, 3), dtype=torch.float32, device=dev)
    slot = torch.arange(n, dtype=torch.int64, device=dev)
    stats = Stats()
    t_all = time.perf_counter()
    for s in range(spp):
        o, d = camera_rays(scene, width, height, s, dev)
        beta = torch.ones((n, 3), dtype=torch.float32, device=dev)
        radiance = torch.zeros((n, 3), dtype=torch.float32, device=dev)
        active = torch.ones(n, dtype=torch.bool, device=dev)
        for depth in range(max_depth):
            hit, _, p, normal, mat = intersect(scene, o, d)
            escaped = active & ~hit
            radiance = radiance + torch.where(escaped[:, None], beta * sky(scene, d), torch.zeros_like(beta))
            active = active & hit
            fs, ft = frame(normal)
            wi = to_local(-d, fs, ft, normal)
            wl = to_local(light.expand(n, 3), fs, ft, normal)
            origin = p + scene.eps * normal
            shadowed, _, _, _, _ = intersect(scene, origin, light.expand(n, 3).contiguous())
            counter = (slot * 64 + s) * 64 + depth
            u = torch.stack([hash_u01(counter, 7), hash_u01(counter, 8)], -1).contiguous()
            queue, count = build_queue(active)

            torch.cuda.synchronize(dev) if dev.type == "cuda" else None
            t0 = time.perf_counter()
            rgb, _, wo2, pdf2, weight = shade(wi, wl, u, mat.contiguous(), queue, count)
            torch.cuda.synchronize(dev) if dev.type == "cuda" else None
            dt = time.perf_counter() - t0
            live = int(count.item())
            stats.bounces += 1; stats.queued_units += live; stats.shade_seconds += dt
            stats.per_bounce.append((s, depth, live, dt))

            lit = active & ~shadowed
            radiance = radiance + torch.where(lit[:, None], beta * rgb * irradiance, torch.zeros_like(beta))
            active = active & (pdf2 > 0.0)
            beta = torch.where(active[:, None], beta * weight, torch.zeros_like(beta))
            d = torch.where(active[:, None], _normalize(to_world(wo2, fs, ft, normal)), d).contiguous()
            o = origin
        image += radiance
    image /= float(spp)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    stats.total_seconds = time.perf_counter() - t_all
    return image.reshape(height, width, 3), stats


def write_png(path: str, image: torch.Tensor, exposure: float = 1.0):
    """8-bit sRGB PNG with nothing but zlib (no imaging library in the image)."""
    import struct
    import zlib
    x = torch.clamp(image * exposure, 0.0, 1.0)
    x = torch.where(x <= 0.0031308, 12.92 * x, 1.055 * torch.pow(x, 1.0 / 2.4) - 0.055)
    data = (x * 255.0 + 0.5).to(torch.uint8).cpu().numpy()
    h, w, _ = data.shape
    raw = b"".join(b"\x00" + data[y].tobytes() for y in range(h))

    def chunk(tag, payload):
        return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))
