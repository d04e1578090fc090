"""Condenses a tools/profile.sh output directory into the text summary committed under profiles/."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


def load(name):
    with open(os.path.join(out, name)) as f:
        lines = [l for l in f.read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


print("== bench line under rocprofv3 --kernel-trace --stats (3 frames, no counters) ==")
try:
    b = load("bench_under_rocprof.json")
    print("library build %s (rsrt_build_id of the library this run timed)" % b["roofline"]["build_id"])
    print(json.dumps({k: b[k] for k in ("metric", "value", "unit", "ms_per_step", "kernel_ms_per_frame", "config")}))
except Exception as e:  # noqa: BLE001
    print("no bench line:", e)

print("\n== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for r in rows("trace/**/*kernel_stats.csv"):
    print("%-64s calls %6s  total %14s ns  avg %16s ns  %8s %%" % (r.get("Name", "")[:64], r.get("Calls"), r.get("TotalDurationNs"),
                                                                   r.get("AverageNs"), r.get("Percentage")))
print("\n== per-dispatch resources (kernel trace) ==")
seen = set()
for r in rows("trace/**/*kernel_trace.csv"):
    n = r.get("Kernel_Name", "")
    if n in seen:
        continue
    seen.add(n)
    print("%-64s wg %s  VGPR %s accumVGPR %s SGPR %s LDS %s scratch %s" % (
        n[:64], r.get("Workgroup_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size")))

print("\n== bench line on its own (5 frames), PMC counters collected by bench.py in separate rocprofv3 --pmc passes ==")
try:
    b = load("bench.json")
    ro = b.pop("roofline")
    cpu = b.pop("cpu_baseline", None)
    print(json.dumps(b))
    c = ro.pop("counters", None) or {}
    print("\nroofline object (counters listed below):")
    print(json.dumps(ro, indent=1))
    print("\ncounters of %s, one frame per pass (source: %s; library build %s):" % (ro.get("kernel"), ro.get("counters_source"), ro.get("counters_build_id")))
    for k in sorted(c):
        print("    %-28s %20.0f" % (k, c[k]))
    if cpu:
        print("\ncpu_baseline:")
        print(json.dumps(cpu, indent=1))
except Exception as e:  # noqa: BLE001
    print("no bench line:", e)
