"""In-tree build of the native pieces (hipcc cross-compiles gfx950 without a GPU).

    python -m mitsuba_customization_amd.build [--force] [--asm]

Outputs (git-ignored, but shipped to the GPU box by gpurun):
    mitsuba_customization_amd/lib/libmerl_hip.so     C-ABI library + gfx950 kernels
    mitsuba_customization_amd/lib/merl.so, customized_measurement.so, ...   plugin adapters (if present)
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libmerl_hip.so")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
HIP_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function"]


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def lib_sources():
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [os.path.join(ROOT, "include", "merl_hip.h")]
    return srcs, deps


def source_hash() -> str:
    """12 hex digits over the library's sources (kernels, C ABI, header): the library reports it (mrl_build_info) and
    committed counter measurements carry it, so that bench.py can tell when profiles/traffic.json was measured on other code."""
    import hashlib
    h = hashlib.sha256()
    for path in lib_sources()[1]:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def _compile_objects(srcs, objdir: str, flags) -> list:
    """One object per translation unit, compiled side by side (hipcc takes ~10-60 s per file: the kernels dominate)."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(objdir, exist_ok=True)
    objs = [os.path.join(objdir, os.path.basename(s) + ".o") for s in srcs]

    def one(pair):
        src, obj = pair
        subprocess.check_call([HIPCC] + flags + ["-c", "-o", obj, src], cwd=PKG)
        return obj
    with ThreadPoolExecutor(max_workers=min(len(srcs), max(1, (os.cpu_count() or 2) - 1))) as pool:
        return list(pool.map(one, zip(srcs, objs)))


def build_lib(force: bool = False, asm: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    srcs, deps = lib_sources()
    if force or _newer(LIB, deps):
        # several ranks of one job may get here at once in a checkout without build outputs: one of them
        # compiles (to a private name, renamed into place when complete), the others wait on the lock
        import fcntl
        with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if force or _newer(LIB, deps):
                tmp = f"{LIB}.{os.getpid()}.tmp"
                objdir = os.path.join(LIBDIR, f"obj.{os.getpid()}")
                try:
                    objs = _compile_objects(srcs, objdir, HIP_FLAGS + [f'-DMRL_SOURCE_HASH="{source_hash()}"'])
                    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs, cwd=PKG)
                    os.replace(tmp, LIB)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
                    shutil.rmtree(objdir, ignore_errors=True)
    if asm:
        out = os.path.join(LIBDIR, "asm")
        os.makedirs(out, exist_ok=True)
        for s in srcs:
            if "kernels" not in os.path.basename(s):
                continue
            subprocess.check_call([HIPCC] + HIP_FLAGS + ["--cuda-device-only", "-S", "-o",
                                  os.path.join(out, os.path.basename(s) + ".s"), s], cwd=PKG)
    return LIB


def build_adapters(force: bool = False):
    """Plugin adapters + their C++ test drivers (see adapters/Makefile-less recipe in adapters/build.py)."""
    ad = os.path.join(PKG, "adapters", "build.py")
    if os.path.exists(ad):
        from .adapters import build as adapters_build
        return adapters_build.build_all(force=force)
    return []


def build_all(force: bool = False, asm: bool = False):
    out = [build_lib(force=force, asm=asm)]
    out += build_adapters(force=force)
    return out


if __name__ == "__main__":
    print("\n".join(build_all(force="--force" in sys.argv, asm="--asm" in sys.argv)))
