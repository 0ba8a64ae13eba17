#!/usr/bin/env python3
"""How many memory round trips a wave of each kernel makes: compiles a .hip translation unit to gfx950 assembly and counts, per kernel,
the vector loads and the GROUPS of loads that are followed by an `s_waitcnt vmcnt` before the next load is issued — each group is one
round trip the wave waits out (static count over the whole kernel: staging loops and cold paths included, so compare builds, not kernels).
A table read written next to its use inside a conditional block becomes a group of its own: round 4's RGL eval had 30 groups for 31
loads before its lookups were split into reads and sums, 14 after.
    python tools/isa_round_trips.py mitsuba_customization_amd/csrc/merl_rgl.hip [name-filter] > table.json"""
import json
import os
import re
import subprocess
import sys
import tempfile


def round_trips(asm_text):
    """{demangled kernel name: {"loads": n, "round_trips": n}} of one assembly file"""
    out = {}
    names = re.findall(r"\.amdhsa_kernel (\S+)", asm_text)
    demangled = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines() if names else []
    for n, d in zip(names, demangled):
        i = asm_text.find("\n" + n + ":")
        j = asm_text.find("s_endpgm", i)
        lines = [l.strip() for l in asm_text[i:j].splitlines()]
        loads = [k for k, l in enumerate(lines) if l.startswith(("global_load", "buffer_load", "flat_load", "scratch_load"))]
        waits = [k for k, l in enumerate(lines) if l.startswith("s_waitcnt") and "vmcnt" in l]
        groups = sum(1 for a, b in zip(loads, loads[1:] + [len(lines)]) if any(a < w < b for w in waits))
        d = d.replace("mrl::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        out[d] = {"loads": len(loads), "round_trips": groups}
    return out


def compile_to_asm(src, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        s = os.path.join(tmp, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", s, src, *extra],
                              stderr=subprocess.DEVNULL)
        return open(s).read()


if __name__ == "__main__":
    table = round_trips(compile_to_asm(sys.argv[1]))
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    print(json.dumps({k: v for k, v in table.items() if flt in k}, indent=1))
