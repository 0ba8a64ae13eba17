#!/usr/bin/env python3
"""Static instruction-class table of the shipped kernels, read off the ISA dump (lib/asm/*.s).

    python -m mitsuba_customization_amd.build --asm      # regenerates lib/asm/merl_kernels.hip.s
    python tools/isa_table.py [--out profiles/r03_isa_table.json] [--match k_table_dma] [--md]

One lane = one unit, so "instructions per wave-iteration" == "VALU instructions per unit" for the
per-lane classes.  Two counts per kernel:
  * `function`: every instruction of the kernel's text (both sides of wave-uniform option branches);
  * `hot`: the instructions on the default option path (cosine sampling, integer nodes) — blocks that are
    only reachable through a branch on the table-importance-sampling option are left out.  The hot
    count is found structurally: the basic blocks that contain the `bin_of` binary search (s_cbranch on
    v_cmp_le_f64 inside a loop of <= 12 instructions) are what the option guards; rather than guess, the
    tool reports `function` and, when the kernel has one, the longest straight-line loop body `loop`.
The executed count per wave (the number that matters) comes from the SQ counters (tools/pmc_valu.sh):
SQ_INSTS_VALU / SQ_WAVES / iterations.  SURVEY.md §8d "Flops" row: lane-ops/unit against 157.3 TFLOPS f32
(78.6 f64) = 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz [x 2 flops for FMA].
"""
from __future__ import annotations

import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = os.path.join(ROOT, "mitsuba_customization_amd", "lib", "asm", "merl_kernels.hip.s")

F64 = re.compile(r"^v_\w*_f64|^v_cvt_\w*f64|^v_cvt_f64")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_")


def classify(op: str) -> list[str]:
    """classes an opcode belongs to (an opcode can be in several: e.g. v_rsq_f64 is valu, f64, transcendental)"""
    c = []
    if op.startswith("v_"):
        c.append("valu")
        if F64.match(op):
            c.append("valu_f64")
        if TRANS.match(op):
            c.append("transcendental")
        if op.startswith("v_cvt"):
            c.append("cvt")
        if op.startswith("v_cndmask"):
            c.append("cndmask")
        if op.startswith(("v_mov", "v_accvgpr")):
            c.append("mov")
        if op.startswith("v_cmp"):
            c.append("cmp")
        if op.startswith("v_fma") or op.startswith("v_fmac") or op.startswith("v_pk_fma"):
            c.append("fma")
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            c.append("lane_xchg")
    elif op.startswith("s_"):
        c.append("salu")
        if op.startswith("s_waitcnt"):
            c.append("waitcnt")
        if op.startswith(("s_cbranch", "s_branch")):
            c.append("branch")
        if op.startswith("s_load") or op.startswith("s_buffer_load"):
            c.append("smem")
    elif op.startswith("ds_"):
        c.append("lds")
        if "bpermute" in op or "permute" in op or "swizzle" in op:
            c.append("lds_xchg")
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        c.append("vmem")
        if "lds" in op:
            c.append("vmem_lds_dma")
        elif "load" in op:
            c.append("vmem_load")
        elif "store" in op:
            c.append("vmem_store")
    return c or ["other"]


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), text=True,
                             capture_output=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def parse(path: str):
    """-> {symbol: {"ops": [(label_or_None, opcode)], "meta": {...}}}"""
    funcs = {}
    cur = None
    meta_re = re.compile(r"^\s*[;.]\s*\.?(sgpr_count|vgpr_count|NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|"
                         r"agpr_count|NumAgprs|TotalNumVgprs|codeLenInByte)[:=]?\s*:?\s*(\d+)")
    with open(path) as f:
        for line in f:
            s = line.strip()
            if s.startswith(".globl"):
                continue
            m = re.match(r"^(_Z\w+):\s*(;.*)?$", s)
            if m and "k_" in m.group(1):
                cur = m.group(1)
                funcs[cur] = {"ops": [], "meta": {}, "labels": {}}
                continue
            if cur is None:
                continue
            if s.startswith(".Lfunc_end"):
                # metadata comments follow until the next function
                continue
            mm = re.match(r"^;\s*(NumVgprs|NumAgprs|TotalNumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte):\s*(\d+)", s)
            if mm:
                funcs[cur]["meta"][mm.group(1)] = int(mm.group(2))
                continue
            lab = re.match(r"^(\.LBB\d+_\d+):", s)
            if lab:
                funcs[cur]["labels"][lab.group(1)] = len(funcs[cur]["ops"])
                continue
            if not s or s.startswith((";", ".", "//")):
                continue
            parts = s.split(None, 1)
            op = parts[0]
            if not re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", op):
                continue
            funcs[cur]["ops"].append((op, parts[1] if len(parts) > 1 else ""))
    return funcs


def count(ops):
    c = collections.Counter()
    for op, _ in ops:
        for k in classify(op):
            c[k] += 1
    c["total"] = len(ops)
    return dict(sorted(c.items()))


def main_loop(func):
    """the outermost backward branch's body: ops[label_pos .. branch_pos] of the LAST backward branch whose
    span is the largest — the grid-stride / tile loop of the kernel"""
    best = None
    for i, (op, arg) in enumerate(func["ops"]):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = arg.split()[0].rstrip(",") if arg else ""
            pos = func["labels"].get(tgt)
            if pos is not None and pos <= i:
                if best is None or i - pos > best[1] - best[0]:
                    best = (pos, i + 1)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", default=ASM)
    ap.add_argument("--match", default="k_table_dma,k_ggx,k_table_nch,k_measured")
    ap.add_argument("--out", default="")
    ap.add_argument("--md", action="store_true", help="print a markdown table")
    a = ap.parse_args()
    paths = [a.asm]
    extra = [p for p in sorted(os.listdir(os.path.dirname(a.asm))) if p.endswith(".s") and os.path.join(os.path.dirname(a.asm), p) != a.asm]
    paths += [os.path.join(os.path.dirname(a.asm), p) for p in extra]
    keys = [k for k in a.match.split(",") if k]
    rows = []
    for p in paths:
        funcs = parse(p)
        names = demangle(list(funcs))
        for sym, f in funcs.items():
            if not any(k in sym for k in keys):
                continue
            whole = count(f["ops"])
            loop = main_loop(f)
            row = {"file": os.path.basename(p), "symbol": sym, "kernel": names[sym].replace("mrl::(anonymous namespace)::", "").replace("(mrl::BatchArgs)", ""),
                   "meta": f["meta"], "function": whole}
            if loop:
                row["loop"] = count(f["ops"][loop[0]:loop[1]])
            rows.append(row)
    git = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    doc = {"git": git, "source": [os.path.relpath(p, ROOT) for p in paths],
           "note": "static counts per wave-iteration (= per unit per lane); `loop` = body of the kernel's tile loop, both sides of "
                   "wave-uniform option branches included; executed counts: profiles/*valu_pmc*.json",
           "peak": {"valu_f32_tflops": 157.3, "valu_f64_tflops": 78.6}, "kernels": rows}
    if a.out:
        with open(a.out, "w") as f:
            json.dump(doc, f, indent=1)
    if a.md or not a.out:
        cols = ["valu", "valu_f64", "transcendental", "cvt", "cndmask", "mov", "cmp", "lds", "lds_xchg", "vmem", "salu"]
        print("| kernel | VGPR | " + " | ".join(cols) + " |")
        print("|---|---|" + "---|" * len(cols))
        for r in rows:
            c = r.get("loop", r["function"])
            print(f"| `{r['kernel']}` | {r['meta'].get('NumVgprs', '?')} | " + " | ".join(str(c.get(k, 0)) for k in cols) + " |")


if __name__ == "__main__":
    main()
