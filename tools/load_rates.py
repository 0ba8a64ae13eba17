#!/usr/bin/env python3
"""Row a1 (table load + HBM re-layout): seconds per MERL table through mrl_material_load_merl (file -> bricks) and
mrl_material_upload_f64 (host array -> bricks), and the release that undoes it.   python tools/load_rates.py"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mitsuba_customization_amd import host, synth

tab = synth.make_table("ggx_tab", 0)
res = {}
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "t.binary")
    synth.write_merl_binary(path, tab)
    for layout, name in ((1, "bricks"), (0, "rows")):
        with host.MerlHip(0) as g:
            g.set_option(host.OPT_TABLE_LAYOUT, layout)
            g.release_material(g.load_merl(path))                  # warm: page cache, first hipMalloc
            t0 = time.perf_counter()
            ids = [g.load_merl(path) for _ in range(20)]
            t_file = (time.perf_counter() - t0) / 20
            t0 = time.perf_counter()
            for i in ids:
                g.release_material(i)
            t_rel = (time.perf_counter() - t0) / 20
            t0 = time.perf_counter()
            ids = [g.upload_merl(tab) for _ in range(20)]
            t_up = (time.perf_counter() - t0) / 20
            res[name] = {"load_file_ms": round(t_file * 1e3, 2), "upload_array_ms": round(t_up * 1e3, 2), "release_ms": round(t_rel * 1e3, 2),
                         "file_MBps": round(35.0 / t_file, 0), "tables_per_s": round(1 / t_file, 1)}
print(json.dumps(res, indent=1))
