"""The resource figures DESIGN.md and bench.py's `steer_kernels.occupancy` quote are read from the built code objects
(tools/kernel_resources.py: AMDGPU metadata of reak_amd/librkh.so); this pins the ones the design rests on.  No GPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def res():
    import kernel_resources as kr

    from reak_amd import lib

    lib.build()
    return kr.kernel_resources()


def test_throughput_steer_kernels_hold_two_waves_per_simd_without_scratch(res):
    """The two-lanes-per-edge mapping is designed around two waves per SIMD (<= 256 registers per lane) and eight waves
    per CU (<= 20 KB of LDS per wave); neither form may spill or use a private segment for the chains it is launched
    for in the benchmark (<= 6 joints)."""
    import kernel_resources as kr

    for n in (1, 2, 3, 4, 6):
        for k in (f"rkh::propagate_pair_step_kernel<{n}>", f"rkh::propagate_pair_kernel<{n}>"):
            d = res[k]
            assert d["vgpr_count"] <= 256 and d["agpr_count"] == 0, (k, d)
            assert d["vgpr_spill_count"] == 0 and d["private_segment_fixed_size"] == 0, (k, d)
            assert kr.waves_per_simd(d) == 2
            assert d["group_segment_fixed_size"] * 8 <= 160 * 1024, (k, d)
    assert res["rkh::propagate_pair_step_kernel<6>"]["group_segment_fixed_size"] == 20272


def test_latency_kernels_of_scenes_without_vertex_sets_do_not_spill(res):
    """The one-wave-per-edge steer kernel (small rounds, single problems) and the quasi-static edge walk (graph planners)
    as instantiated for scenes without vertex-set shapes: no spilled registers, (next to) no private segment -- the
    support-map query's run-time-indexed simplex arrays are what needs one, and only the `true` instantiations carry it."""
    for k in ("rkh::propagate_kernel<6, 64, false, false>", "rkh::propagate_kernel<6, 64, false, true>"):  # one / two waves per edge
        d = res[k]
        assert d["vgpr_spill_count"] == 0 and d["private_segment_fixed_size"] <= 64, d
    for g in (32, 64):
        d = res[f"rkh::edge_points_kernel<6, false, {g}>"]
        assert d["vgpr_count"] <= 256 and d["vgpr_spill_count"] == 0 and d["private_segment_fixed_size"] == 0, d
    d = res["rkh::edge_points_kernel<12, false, 32>"]
    assert d["vgpr_count"] <= 256 and d["vgpr_spill_count"] == 0 and d["private_segment_fixed_size"] == 0, d


def test_committed_resource_table_matches_the_build(res):
    """profiles/r03_kernel_resources.txt is the table the documents cite: its rows for the steer and NN kernels must be
    what the current sources compile to (regenerate with `python tools/kernel_resources.py --out ...`)."""
    path = os.path.join(ROOT, "profiles", "r03_kernel_resources.txt")
    rows = {}
    for line in open(path):
        if line.startswith("#") or line.startswith("kernel") or not line.strip():
            continue
        name, rest = line[:78].strip(), line[78:].split()
        rows[name] = [int(x) for x in rest]
    checked = 0
    for k, d in res.items():
        if not any(t in k for t in ("propagate_pair", "propagate_kernel<6", "nn1_mirror_kernel", "edge_points_kernel<6")):
            continue
        got = [d["vgpr_count"], d["agpr_count"], d["vgpr_spill_count"], d["sgpr_count"], d["private_segment_fixed_size"],
               d["group_segment_fixed_size"]]
        assert rows[k[:78]][:6] == got, (k, rows[k[:78]], got)
        checked += 1
    assert checked >= 10
