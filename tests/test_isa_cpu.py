"""The property round 4's RGL speed-up rests on, checked on the compiled code (hipcc cross-compiles without a GPU): a lookup's reads
are issued together, i.e. the kernels make few memory round trips per wave.  Which loads the compiler can hoist depends on how the
source spells the conditional reads (merl_rgl.hpp, fetch_raw: defined values for absent slices) — a change there, or in the
compiler, that puts every read back next to its use shows up here as 2-3 times the round trips, long before anyone reads a profile."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc missing")
def test_rgl_kernels_issue_a_lookups_reads_together():
    import isa_round_trips as irt
    table = irt.round_trips(irt.compile_to_asm(os.path.join(ROOT, "mitsuba_customization_amd", "csrc", "merl_rgl.hip")))
    # static counts over the whole kernel (grid staging included); before the reads / sums split: eval 30 of 31 loads, fused LDS 68,
    # fused batch-with-ids 92
    bounds = {"k_rgl<0, false, false, 5>": 18, "k_rgl<0, false, false, 15>": 18, "k_rgl<1, false, false, 15>": 17, "k_rgl<4, false, false, 15>": 19,
              "k_rgl_lds<3, false, false, 5>": 34, "k_rgl<3, false, true, 0>": 44, "k_rgl<0, false, true, 0>": 20}
    for name, bound in bounds.items():
        assert name in table, (name, sorted(table)[:5])
        got = table[name]
        assert got["round_trips"] <= bound, (name, got, bound)
        assert got["loads"] >= 2 * got["round_trips"], (name, got)      # reads batched: at least two per round trip on average
