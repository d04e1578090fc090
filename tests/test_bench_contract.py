"""bench.py's roofline arithmetic on the committed counters (no GPU): the object must be a fraction of the roof the
kernel is under, carry the counters and the build id it was taken on, and refuse nothing silently."""
import json
import os
import sys

import util

sys.path.insert(0, util.ROOT)
import bench  # noqa: E402


def committed():
    with open(os.path.join(util.ROOT, "profiles", "pmc_house_1080p_8b.json")) as f:
        return json.load(f)


def house_counts():
    """The counting oracle (liboracle_ops.so) on a small frame of the bench workload: f32 / u32 operations per path."""
    import oracle
    import rsoderh_raytracing_amd as R
    sc = R.Scene.load_toml(util.scene_path("house"))
    _, c = oracle.render(util.oracle_scene(sc), util.oracle_env(util.small_env()), sc.camera_uniform().view(oracle.CAMERA), 96, 54, 0, 1, 8,
                         flags=oracle.FLAG_ANYHIT_SHADOW, fast="ops")
    c["leaf_boxes"] = int((sc.bvh_nodes["primitives_len"] > 0).sum())
    return c


def test_roofline_object_is_a_fraction_of_the_valu_roof():
    c = committed()
    pmc = {k: c[k] for k in ("counters", "resolve", "kernel", "build_id", "launch_ms_under_pmc")}
    pmc["source"] = "committed"
    paths = 1920 * 1080 * 256
    counts = house_counts()
    ro = bench.roofline_object(pmc, c["build_id"], 114.0, paths, c["algorithmic_bytes_per_path"], 1920 * 1080, 256, counts)
    assert ro["bound"] == "valu" and ro["unit"] == "T lane-operations/s"
    # frac: ALGORITHMIC operations (counted by the oracle) against the peak; utilisation: the kernel's own retired lane-instructions
    assert 0.0 < ro["frac"] <= ro["utilisation"] <= 1.0
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12 and abs(ro["peak"] - 78.64) < 0.01
    assert abs(ro["overhead"] - (1.0 - ro["frac"] / ro["utilisation"])) < 1e-12 and 0.0 <= ro["overhead"] < 1.0
    w = ro["work"]
    assert 2000 < w["f32_ops_per_path"] < 5000 and 100 < w["u32_ops_per_path"] < 1000  # house, 8 bounces: ~3,100 + ~340
    assert abs(ro["achieved"] * 1e12 * 0.114 / paths - (w["f32_ops_per_path"] + w["u32_ops_per_path"])) < 1e-6
    assert 8.0 < w["reference_walk_nodes_per_ray"] < w["flat_loop_leaf_boxes_per_ray"] == 20  # the flat loop tests every leaf box
    v = ro["valu"]
    assert abs(ro["utilisation"] - v["issue_frac"] * 0.5 * v["lanes_active_per_instruction"] / 32.0) < 1e-9
    assert 0.0 < v["wall"]["frac"] <= 1.0
    assert "rt_render_pool_kernel" in ro["kernel"] and ro["counters_build_id"] == c["build_id"]
    for k in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "WRITE_SIZE", "TCC_EA0_RDREQ_128B_sum"):
        assert ro["counters"][k] > 0, k
    h = ro["hbm"]
    assert ro["traffic"] == h["fetch_bytes_per_launch"] + h["write_bytes_per_launch"] > 0
    chk = h["resolve_kernel_check"]  # the resolve kernel streams a known byte count: the read counters agree within 1 %
    assert abs(chk["counted_bytes"] / chk["known_bytes"] - 1.0) < 0.01
    assert 0.0 < h["l2_hit_rate"] < 1.0 and 0.0 < ro["lds"]["array_busy_frac"] < 1.0
    assert ro["algorithmic"]["frac_of_hbm_peak"] > 1.0  # ... which is exactly why HBM is not the roof that is reported


def test_without_counters_the_object_still_carries_the_algorithmic_fraction():
    ro = bench.roofline_object(None, "0123456789abcdef", 114.0, 1.0e6, 3450.0, 1000, 256)
    assert ro["frac"] is None and ro["achieved"] is None and ro["utilisation"] is None and "no PMC counters" in ro["note"]
    assert ro["algorithmic"]["bytes_per_path"] == 3450.0
    ro = bench.roofline_object(None, "0123456789abcdef", 114.0, 1920 * 1080 * 256, 3450.0, 1000, 256, house_counts())
    assert 0.0 < ro["frac"] < 1.0 and ro["utilisation"] is None and ro["overhead"] is None  # the numerator needs no counters, the utilisation does


def test_the_counting_oracle_computes_the_same_image():
    import numpy as np
    import oracle
    import rsoderh_raytracing_amd as R
    sc = R.Scene.load_toml(util.scene_path("default"))
    args = (util.oracle_scene(sc), util.oracle_env(util.small_env()), sc.camera_uniform().view(oracle.CAMERA), 48, 32, 0, 3, 10)
    a, sa = oracle.render(*args)
    b, sb = oracle.render(*args, fast="ops")
    assert np.array_equal(util.bits(a), util.bits(b))
    assert sa["f32_ops"] == 0 and sb["f32_ops"] > 1000 * sb["paths"] and sb["int_ops"] > 50 * sb["paths"]
    assert all(sa[k] == sb[k] for k in sa if k not in ("f32_ops", "int_ops"))
    # a known case: a camera ray that escapes at once costs the ray generation, one unpruned walk that misses the root box and the escape
    # (no shading): far fewer operations than a path that bounces
    assert sb["f32_ops"] / sb["paths"] > 500


def test_committed_counters_belong_to_the_committed_kernel_sources():
    """profiles/pmc_house_1080p_8b.json is bench.py's fallback when rocprofv3 cannot run; it is only attached when it was
    taken on the library being timed, so it must be refreshed whenever a kernel source changes."""
    import pytest
    from rsoderh_raytracing_amd import _build
    if committed()["build_id"] != _build.source_id():  # not an error of the code under test: bench.py measures live and refuses the stale file
        pytest.skip("profiles/pmc_house_1080p_8b.json is stale (kernel sources changed since): re-run tools/profile.sh r02_house on the GPU box")


def test_bench_line_carries_the_other_configs_and_the_multi_gpu_proof():
    """The keys round 4 added to the bench line: the source must emit `extra_configs` at N = 1 and `frame_equals_1gpu` /
    `multi_gpu` at N > 1, and the extra configs are the BASELINE.json ones the headline does not carry, both general-BVH scenes included."""
    labels = [c[0] for c in bench.EXTRA_CONFIGS]
    assert len(labels) == 5 and any("default.toml 1280x720 64 spp" in l for l in labels) and any("cube.toml 1280x720 128 spp" in l for l in labels)
    assert any("suzanne.toml 1280x720 128 spp" in l for l in labels) and any("suzanne grid 4x4 1280x720 32 spp" in l for l in labels)
    assert any("single-sample calls" in l for l in labels)
    with open(os.path.join(util.ROOT, "bench.py")) as f:
        src = f.read()
    for key in ('result["extra_configs"]', 'result["frame_equals_1gpu"]', 'result["multi_gpu"]', '"per_rank"', '"pre_flight"', '"exchange"', 'comm_set_mode', '"ms_per_frame"', '"mrays_s"', '"frac"'):
        assert key in src, key
    assert bench.frame_proof([[[1.0, 2.0, 3.0, 1.0]]], [[[1.0, 2.0, 3.0, 1.0]]]) == {"frame_equals_1gpu": True}
    bad = bench.frame_proof([[[1.0, 2.0, 3.0, 1.0]]], [[[1.0, 2.5, 3.0, 1.0]]])
    assert bad["frame_equals_1gpu"] is False and bad["differing_pixels"] == 1 and bad["first_differing_pixel_xy"] == [0, 0]
    assert bench.gather_rank_stats(0, 1, 12.5, 0.0) == [{"rank": 0, "trace_ms": 12.5, "reduce_ms": 0.0}]


def _profile_pairs(prefix="r04_"):
    import glob
    for csv_path in sorted(glob.glob(os.path.join(util.ROOT, "profiles", prefix + "*_kernel_stats.csv"))):
        yield csv_path, csv_path.replace("_kernel_stats.csv", "_rocprofv3_summary.txt")


def test_committed_kernel_stats_agree_with_their_summaries():
    """Round 2 and round 3 each committed a kernel_stats.csv of another run than the summary beside it (a trace directory that accumulated runs).
    From round 4 on: every profiles/r04_*_kernel_stats.csv names the library it was taken on (first line, written by tools/keep_profile.sh), its
    summary names the same one, and the dominant kernel's average duration is the same in both within 3 %."""
    import csv
    import re
    pairs = list(_profile_pairs())
    assert pairs, "no profiles/r04_*_kernel_stats.csv committed"
    for csv_path, summary_path in pairs:
        with open(csv_path) as f:
            first = f.readline()
            assert first.startswith("# rsrt_build_id "), csv_path
            csv_id = first.split()[2]
            rows = list(csv.DictReader(f))
        top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        assert "rt_render_pool_kernel" in top["Name"], (csv_path, top["Name"])
        with open(summary_path) as f:
            text = f.read()
        m = re.search(r"library build ([0-9a-f]{16})", text)
        assert m and m.group(1) == csv_id, (summary_path, csv_id, m and m.group(1))
        line = next(l for l in text.splitlines() if l.startswith(top["Name"][:64]) and " avg " in l)
        avg = float(re.search(r"avg\s+([0-9.]+) ns", line).group(1))
        assert abs(avg / float(top["AverageNs"]) - 1.0) < 0.03, (csv_path, avg, top["AverageNs"])
        bench_line = json.loads([l for l in open(csv_path.replace("_kernel_stats.csv", "_bench.json")) if l.startswith("{")][-1])
        assert bench_line["roofline"]["build_id"] == csv_id


def test_committed_r04_profiles_were_taken_on_the_committed_kernel_sources():
    import pytest
    from rsoderh_raytracing_amd import _build
    stale = []
    for csv_path, _ in _profile_pairs():
        with open(csv_path) as f:
            if f.readline().split()[2] != _build.source_id():
                stale.append(os.path.basename(csv_path))
    if stale:  # (as for the PMC file above: bench.py measures live; a stale profile is a to-do, not a defect of the code under test)
        pytest.skip("taken on other kernel sources than the committed ones: %s — re-run tools/r04_profiles.sh on the GPU box" % ", ".join(stale))


def test_overhead_ledger_adds_up_to_the_counters():
    """profiles/r04_house_ledger.json (tools/ledger.py): the causes, the unattributed rest included, add up to the hardware counters'
    lane-instructions per ray, at least four fifths of them are attributed, and bench.py hangs the ledger on the roofline object."""
    with open(os.path.join(util.ROOT, "profiles", "r04_house_ledger.json")) as f:
        lg = json.load(f)
    total = lg["counter_total_lane_instructions_per_ray"]
    causes = lg["ledger_lane_instructions_per_ray"]
    assert abs(sum(causes.values()) / total - 1.0) < 1e-9 and 1500 < total < 1800
    rest = next(v for k, v in causes.items() if k.startswith("not attributed"))
    assert 0.0 <= rest / total < 0.2
    assert abs(total / lg["counter_total_of_the_256spp_bench_run"] - 1.0) < 0.02  # the 64-spp ablation frame and the 256-spp bench frame agree
    a = lg["ablations"]
    assert a["product"]["lane_instructions_per_ray"] > a["b93b1a04be"]["lane_instructions_per_ray"] > a["402bff4434"]["lane_instructions_per_ray"]
    ro = bench.roofline_object(None, "0123456789abcdef", 98.0, 1920 * 1080 * 256, 3450.0, 1920 * 1080, 256, house_counts())
    assert abs(sum(ro["overhead_ledger"]["share"].values()) - 1.0) < 1e-9 and "declared done" in ro["overhead_ledger"]["reading"]
