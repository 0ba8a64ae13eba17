"""Builds the plugin adapters and their test drivers with g++ (host C++ only — every compute call
goes through libmerl_hip.so).  Outputs under mitsuba_customization_amd/lib/:

    plugins06/merl.so, plugins06/customized_measurement.so     Mitsuba 0.6 plugins (CreateInstance / GetDescription)
    plugins3/merl.so,  plugins3/customized_measurement.so      Mitsuba 3 plugins (plugin_name / plugin_descr / ...)
    plugins3/measured.so                                       Mitsuba 3's stock RGL plugin name over the library's RGL material
    driver06, driver3                                          stand-ins for the hosts' plugin managers (tests)

Against a real Mitsuba tree: add -DMERL_USE_REAL_MITSUBA and the tree's include dirs instead of mirror/.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIBDIR = os.path.join(PKG, "lib")
CXX = shutil.which("g++") or "g++"
CXXFLAGS = ["-O2", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter", "-fvisibility=hidden"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_all(force: bool = False):
    outs = []
    common = glob.glob(os.path.join(HERE, "common", "*.hpp")) + [os.path.join(os.path.dirname(PKG), "include", "merl_hip.h")]
    for host, mirror_hdr in (("mitsuba06", "mitsuba.h"), ("mitsuba3", "mitsuba3.h")):
        src_dir = os.path.join(HERE, host)
        out_dir = os.path.join(LIBDIR, "plugins06" if host == "mitsuba06" else "plugins3")
        os.makedirs(out_dir, exist_ok=True)
        inc = ["-I", os.path.join(src_dir, "mirror")]
        deps = common + glob.glob(os.path.join(src_dir, "*.hpp")) + [os.path.join(src_dir, "mirror", "mitsuba", mirror_hdr)]
        for name in ("merl", "customized_measurement") + (("measured",) if host == "mitsuba3" else ()):
            src = os.path.join(src_dir, name + ".cpp")
            out = os.path.join(out_dir, name + ".so")
            if force or _stale(out, deps + [src]):
                # $ORIGIN/.. = mitsuba_customization_amd/lib, where libmerl_hip.so lives
                subprocess.check_call([CXX] + CXXFLAGS + inc + ["-shared", "-o", out, src, "-L", LIBDIR, "-lmerl_hip",
                                                                  "-Wl,-rpath,$ORIGIN/..", "-lpthread"])
            outs.append(out)
        drv_src = os.path.join(HERE, "tests", "driver06.cpp" if host == "mitsuba06" else "driver3.cpp")
        drv = os.path.join(LIBDIR, "driver06" if host == "mitsuba06" else "driver3")
        if force or _stale(drv, deps + [drv_src, os.path.join(HERE, "tests", "driver_common.hpp")]):
            subprocess.check_call([CXX] + [f for f in CXXFLAGS if f != "-fvisibility=hidden"] + inc + ["-o", drv, drv_src, "-ldl", "-lpthread"])
        outs.append(drv)
        if host == "mitsuba3":                      # the scalar_spectral variant's driver (an RGL *_spec.bsdf file through `measured`)
            sp_src = os.path.join(HERE, "tests", "driver3_spectral.cpp")
            sp = os.path.join(LIBDIR, "driver3_spectral")
            if force or _stale(sp, deps + [sp_src, os.path.join(HERE, "tests", "driver_common.hpp")]):
                subprocess.check_call([CXX] + [f for f in CXXFLAGS if f != "-fvisibility=hidden"] + inc + ["-o", sp, sp_src, "-ldl", "-lpthread"])
            outs.append(sp)
    # the C ABI from plain C99 (examples/abi_example.c): also proves include/merl_hip.h is a C header
    ex_src = os.path.join(os.path.dirname(PKG), "examples", "abi_example.c")
    ex = os.path.join(LIBDIR, "abi_example")
    if os.path.exists(ex_src) and (force or _stale(ex, [ex_src, os.path.join(os.path.dirname(PKG), "include", "merl_hip.h")])):
        subprocess.check_call([shutil.which("gcc") or "gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-pedantic",
                               "-I", os.path.join(os.path.dirname(PKG), "include"), "-o", ex, ex_src,
                               "-L", LIBDIR, "-lmerl_hip", "-Wl,-rpath,$ORIGIN", "-lm"])
    if os.path.exists(ex):
        outs.append(ex)
    # the native multi-GPU host (examples/group_host.cpp): plain C++ over the device-group entry points
    gh_src = os.path.join(os.path.dirname(PKG), "examples", "group_host.cpp")
    gh = os.path.join(LIBDIR, "group_host")
    if os.path.exists(gh_src) and (force or _stale(gh, [gh_src, os.path.join(os.path.dirname(PKG), "include", "merl_hip.h")])):
        subprocess.check_call([CXX, "-std=c++17", "-O2", "-Wall", "-Wextra", "-I", os.path.join(os.path.dirname(PKG), "include"),
                               "-o", gh, gh_src, "-L", LIBDIR, "-lmerl_hip", "-Wl,-rpath,$ORIGIN"])
    if os.path.exists(gh):
        outs.append(gh)
    # a host that makes one-unit calls from many threads (examples/scalar_host.cpp): the scalar service end to end
    sh_src = os.path.join(os.path.dirname(PKG), "examples", "scalar_host.cpp")
    sh = os.path.join(LIBDIR, "scalar_host")
    if os.path.exists(sh_src) and (force or _stale(sh, [sh_src, os.path.join(os.path.dirname(PKG), "include", "merl_hip.h")])):
        subprocess.check_call([CXX, "-std=c++17", "-O2", "-Wall", "-Wextra", "-I", os.path.join(os.path.dirname(PKG), "include"),
                               "-o", sh, sh_src, "-L", LIBDIR, "-lmerl_hip", "-Wl,-rpath,$ORIGIN", "-lpthread"])
    if os.path.exists(sh):
        outs.append(sh)
    return outs


if __name__ == "__main__":
    print("\n".join(build_all(force=True)))
