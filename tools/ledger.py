"""The overhead ledger of the flat kernel: where the VALU lane-instructions the kernel retires per ray go.

Static part (here, no GPU): rsrt_api.hip is compiled once more with -DRT_LEDGER — the product code plus pairs of s_nop that mark the borders of code
regions (rt_device.h RT_MARK) — and the house kernel rt_render_pool_kernel<1,1024,192,2> is disassembled; for every marker the VALU instructions
reachable from it before the next marker are counted (a walk over the control-flow graph: loops count once, both sides of a branch count), and
sorted into kinds: the expansion of IEEE divisions, square roots and exact reciprocals (what the oracle counts as ONE operation each), and the rest.
Dynamic part: the instrumented build counts the LANES that pass every marker on the bench frame (rt_math.h RT_MARK, rsrt_get_region_counters;
tools/simd_efficiency.py writes them).  region instructions x region lanes = lane-instructions; their sum is set against the hardware counters'
own SQ_INSTS_VALU x lanes of the same frame.  Within a region both sides of a branch are counted (the marks sit at the top of every big
conditional block, so what is left is small: early-outs, the NaN-exact selects); paths next to no lane takes end a region (RT_MARK_COLD).

    python tools/ledger.py gpurun_out/stage_shares_house.json profiles/r04_house_bench.json > profiles/r04_house_ledger.txt   (also writes the .json)
"""
import collections, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_Z21rt_render_pool_kernelILi1ELj1024ELj192ELi2EEv12RenderParams"

REGIONS = {  # marker -> name
    (11, 0): "census + compaction", (11, 1): "stage dispatch (list entry, tag word)", (11, 2): "GEN: hand-out + camera ray", (11, 3): "GEN -> fused trace",
    (11, 4): "trace: entry (+ GEN's two-level cull)", (11, 5): "trace: one leaf box", (11, 6): "trace: masks after the box loop", (11, 7): "trace: one trip of the triangle loop (two records)",
    (11, 8): "trace: between the primitive loops", (11, 9): "trace: one plane", (11, 10): "trace: one sphere", (11, 13): "trace: exit + what the slot is left as",
    (11, 12): "TRACE: ray from the hot columns", (11, 14): "GEN: tail", (11, 15): "stage end (hand-over)", (11, 11): "kernel end (counters; once per wave)",
    (12, 0): "MISS: memory requests, cold loads", (12, 1): "MISS: one fallback sphere", (12, 2): "MISS: one fallback plane", (12, 3): "MISS: escape (sky, pdf, MIS) + store",
    (12, 4): "SHADE: memory requests, barycentrics", (12, 5): "SHADE: NEE term (bsdf eval + pdf)", (12, 6): "SHADE: bsdf sample: frame, lobe draw", (12, 7): "FINISH",
    (12, 8): "SHADE: normal of a triangle hit", (12, 9): "SHADE: normal of a sphere hit", (12, 10): "SHADE: normal of a plane hit",
    (12, 11): "SHADE: material, environment sample, emission, NEE cosine", (12, 12): "SHADE: diffuse lobe (cosine hemisphere)", (12, 13): "SHADE: specular lobe (GGX VNDF)",
    (12, 14): "SHADE: eval + pdf of the sampled direction", (12, 15): "SHADE: throughput, termination, state out",
}


LOOP_BODIES = {(11, 5), (11, 7), (11, 9), (11, 10), (12, 1), (12, 2)}  # marks at the top of a loop body: their region is the cycle through the mark


def disassemble():
    with tempfile.TemporaryDirectory() as d:
        co = os.path.join(d, "ledger.co")
        src = os.path.join(ROOT, "rsoderh-raytracing_amd", "csrc", "hip", "rsrt_api.hip")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fno-vectorize",
                        "-I", os.path.join(ROOT, "include"), "-DRT_LEDGER", "--cuda-device-only", "--no-gpu-bundle-output", "-c", "-o", co, src], check=True, capture_output=True)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout
    blocks = re.split(r"\n(?=[0-9a-f]+ <)", dis)
    for b in blocks:
        m = re.match(r"[0-9a-f]+ <(\S+)>:", b)
        if m and m.group(1) == KERNEL:
            ins = []
            for ln in b.splitlines()[1:]:
                m2 = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*(.*)$", ln)
                if m2:
                    ins.append((int(m2.group(3), 16), m2.group(1), m2.group(2), len(m2.group(4).split()) * 4))
            return ins
    raise SystemExit("kernel not found")


def kind_of(mn, window):
    """Instruction kinds for the ledger.  `window`: mnemonics of the neighbourhood (the division sequence is recognised by its bookends)."""
    if mn in ("v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32"):
        return "div"
    if mn == "v_sqrt_f32":
        return "sqrt"
    if mn == "v_rcp_f32":
        return "rcp"
    return "other"


def main():
    ins = disassemble()
    addr_index = {a: i for i, (a, _, _, _) in enumerate(ins)}
    # markers: s_nop 11|12 followed by s_nop k
    marker_at = {}
    for i in range(len(ins) - 1):
        if ins[i][1] == "s_nop" and ins[i + 1][1] == "s_nop" and ins[i][2] in ("11", "12", "13"):
            marker_at[i] = (int(ins[i][2]), int(ins[i + 1][2]))  # (13, 0): a cold path's entry

    def successors(i):
        a, mn, ops, size = ins[i]
        nxt = i + 1 if i + 1 < len(ins) else None
        if mn == "s_endpgm":
            return []
        if mn == "s_branch" or mn.startswith("s_cbranch"):
            off = int(ops.split()[-1])
            if off >= 32768:
                off -= 65536
            tgt = addr_index.get(a + size + 4 * off)
            out = [] if tgt is None else [tgt]
            if mn != "s_branch" and nxt is not None:
                out.append(nxt)
            return out
        return [nxt] if nxt is not None else []

    # predecessors, for "can this instruction come back to the marker" (a loop body's region is the cycle through its marker: what lies behind the
    # loop's exit runs once, not once per trip, and belongs to the region of the marker it leads to)
    preds = collections.defaultdict(list)
    for i in range(len(ins)):
        for j in successors(i):
            if j is not None:
                preds[j].append(i)

    def walk(start_set, step, stop_at_markers=True):
        seen, todo = set(), list(start_set)
        while todo:
            i = todo.pop()
            if i is None or i in seen or i >= len(ins) or i < 0:
                continue
            if stop_at_markers and i in marker_at:
                continue
            seen.add(i)
            todo.extend(step(i))
        return seen

    def tally(idx):
        counts = collections.Counter()
        for i in idx:
            mn = ins[i][1]
            if mn.startswith("v_"):
                counts["valu"] += 1
                counts[kind_of(mn, None)] += 1
            elif mn.startswith("ds_"):
                counts["lds"] += 1
            elif mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
                counts["vmem"] += 1
            elif mn.startswith("s_"):
                counts["salu"] += 1
        return counts

    regions = collections.defaultdict(list)  # marker -> [per occurrence {kind: count}]
    spill = collections.defaultdict(collections.Counter)  # a loop marker's exit path, booked to the marker it leads to
    for start, key in sorted(marker_at.items()):
        if key[0] == 13:
            continue
        fwd = walk([start + 2], successors)
        if key in LOOP_BODIES:  # a loop body: keep what can come back to the marker
            back = walk([start - 1] + preds[start], lambda i: preds[i])
            body = fwd & back
            rest = fwd - body
            # the exit path: to the first marker in program order behind it
            nxt = sorted(m for m in marker_at if m > max(rest)) if rest else []
            if rest and nxt and marker_at[nxt[0]][0] != 13:
                spill[marker_at[nxt[0]]].update(tally(rest))
            regions[key].append(dict(tally(body)))
        else:
            regions[key].append(dict(tally(fwd)))
    for key, extra in spill.items():
        if regions[key]:
            for k, v in extra.items():
                regions[key][0][k] = regions[key][0].get(k, 0) + v
    # dynamic part
    shares = json.load(open(sys.argv[1]))
    c = shares["counters"]
    rays = shares["rays"]
    bench = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    ro = bench["roofline"]
    pmc_per_ray = ro["counters"]["SQ_INSTS_VALU"] * ro["valu"]["lanes_active_per_instruction"] / (bench["config"]["rays_per_frame"])
    algo_per_ray = (ro["work"]["f32_ops_per_path"] + ro["work"]["u32_ops_per_path"]) / ro["work"]["rays_per_path"]
    lanes = shares["region_lanes"]  # lanes that passed mark k (bank 11: k, bank 12: 16 + k)
    weights = {key: lanes[key[1] + (16 if key[0] == 12 else 0)] for key in regions}
    rows, total = [], collections.Counter()
    for key in sorted(regions, key=lambda k: (k[0], k[1])):
        occ = regions[key]
        avg = {k: sum(o.get(k, 0) for o in occ) / len(occ) for k in ("valu", "div", "sqrt", "rcp", "lds", "vmem", "salu")}
        w = weights.get(key, 0.0)
        per_ray = avg["valu"] * w / rays
        rows.append({"region": REGIONS.get(key, str(key)), "copies_in_the_code": len(occ), "valu_instructions": avg["valu"], "lane_trips_per_ray": w / rays,
                     "lane_instructions_per_ray": per_ray, "of_which": {k: avg[k] * w / rays for k in ("div", "sqrt", "rcp")},
                     "lds_instructions": avg["lds"], "vmem_instructions": avg["vmem"]})
        total["valu"] += per_ray
        for k in ("div", "sqrt", "rcp"):
            total[k] += avg[k] * w / rays
    # ---- the ledger by cause.  What can be MEASURED is measured: the totals and the ablations are hardware counters (SQ_INSTS_VALU x lanes active per
    # instruction, one frame per rocprofv3 --pmc pass), the algorithmic operations and the reference walk's node visits are the oracle's counts, the box
    # and scheduler trips are the instrumented build's counters; only the instructions per box (26: the product's unrolled loop, 6 sub 6 mul 8 min / max
    # 1 cmp 2 select 2 or + 1 address) and per scheduler trip (census + compaction + hand-over: the fenced regions above) are read off the disassembly.
    abl = {}
    for tag in ("product", "b93b1a04be", "402bff4434"):
        f = os.path.join(ROOT, "gpurun_out", "pmc_scene_ledger_%s.json" % tag)
        if os.path.exists(f):
            b = json.loads([l for l in open(f) if l.startswith("{")][-1])
            cc = b["roofline"]["counters"]
            abl[tag] = {"lane_instructions_per_ray": cc["SQ_INSTS_VALU"] * cc["SQ_THREAD_CYCLES_VALU"] / cc["SQ_ACTIVE_INST_VALU"] / b["config"]["rays_per_frame"],
                        "ms_per_frame": b["ms_per_frame"], "build_id": b["roofline"]["build_id"]}
    by_name = {r["region"]: r for r in rows}
    box_per_ray = lanes[5] / rays
    ref_nodes = ro["work"]["reference_walk_nodes_per_ray"]
    box_valu = 26.0
    sched_valu = by_name["census + compaction"]["valu_instructions"] + by_name["stage end (hand-over)"]["valu_instructions"]
    sched = sched_valu * lanes[0] / rays
    total_c = abl["product"]["lane_instructions_per_ray"] if "product" in abl else pmc_per_ray
    ledger = collections.OrderedDict()
    ledger["algorithmic: the f32 + u32 operations of the reference's integrator (oracle's counting build)"] = algo_per_ray
    if len(abl) == 3:
        ledger["IEEE division and square root beyond the hardware's 2.5-ulp forms (ablation: -fno-hip-fp32-correctly-rounded-divide-sqrt)"] = abl["product"]["lane_instructions_per_ray"] - abl["b93b1a04be"]["lane_instructions_per_ray"]
        ledger["the exact reciprocal rt_rcp beyond a bare v_rcp_f32 (ablation: the same + -DRT_FAST_RCP)"] = abl["b93b1a04be"]["lane_instructions_per_ray"] - abl["402bff4434"]["lane_instructions_per_ray"]
    ledger["leaf boxes the flat loop tests beyond the nodes the reference's walk visits (%.2f against %.2f a ray x %.0f instructions)" % (box_per_ray, ref_nodes, box_valu)] = max(box_per_ray - ref_nodes, 0.0) * box_valu
    ledger["scheduler: census + compaction + hand-over (%.0f VALU a trip x %.3f lane-trips a ray; 12 %% of the WAVE TIME: mostly scalar and LDS work)" % (sched_valu, lanes[0] / rays)] = sched
    rest = total_c - sum(ledger.values())
    ledger["not attributed: address arithmetic of the LDS / arena columns, selects and moves, loop control and mask walking (ctz / and), type dispatch, detmath's range reduction and selects beyond its counted polynomial, RNG-to-float conversions"] = rest
    loops = collections.OrderedDict()
    for name, mark, valu in (("leaf boxes", 5, box_valu), ("triangle-pair trips", 7, by_name["trace: one trip of the triangle loop (two records)"]["valu_instructions"]),
                             ("planes", 9, by_name["trace: one plane"]["valu_instructions"]), ("spheres", 10, by_name["trace: one sphere"]["valu_instructions"])):
        loops[name] = {"lane_trips_per_ray": lanes[mark] / rays, "valu_per_trip": valu, "lane_instructions_per_ray": lanes[mark] / rays * valu}
    out = {"kernel": KERNEL, "workload": shares["workload"], "instrumented_build_id": shares["build_id"], "bench_build_id": ro["build_id"],
           "counter_total_lane_instructions_per_ray": total_c, "counter_total_of_the_256spp_bench_run": pmc_per_ray, "ablations": abl,
           "ledger_lane_instructions_per_ray": ledger, "ledger_share": {k: v / total_c for k, v in ledger.items()},
           "trace_loops": loops, "stage_lanes_per_ray": {n: lanes[i] / rays for n, i in (("GEN", 2), ("TRACE", 12), ("MISS", 16), ("SHADE", 20), ("FINISH", 23))},
           "how": "totals and ablations: SQ_INSTS_VALU x SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU of one 64-spp house frame per rocprofv3 --pmc pass (tools/r04_ledger.sh); algorithmic operations "
                  "and reference node visits: liboracle_ops.so; trips: lanes that passed the region marks in librsrt_instr.so (rsrt_get_region_counters); instructions per loop trip: "
                  "disassembly of a -DRT_LEDGER build (product code + s_nop marks; loop bodies and the census are fenced by control flow, the straight-line regions of the "
                  "shading stages are not — the compiler moves code across the marks — and are therefore NOT used)"}
    dst = os.path.join(ROOT, "profiles", "r04_house_ledger.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("overhead ledger of %s\n%s\n" % (KERNEL, shares["workload"]))
    print("retired VALU lane-instructions per ray (hardware counters): %.1f   (the 256-spp bench run: %.1f)\n" % (total_c, pmc_per_ray))
    for k, v in ledger.items():
        print("  %8.1f  %5.1f %%  %s" % (v, 100.0 * v / total_c, k))
    print("\nablation builds (one 64-spp frame each):")
    for tag, a_ in abl.items():
        print("  %-12s %8.1f lane-instructions a ray, %.2f ms/frame, build %s" % (tag, a_["lane_instructions_per_ray"], a_["ms_per_frame"], a_["build_id"]))
    print("\nTRACE's loops (trips: instrumented build; instructions a trip: disassembly):")
    for k, v in loops.items():
        print("  %-22s %7.3f lane-trips a ray x %5.0f VALU = %7.1f lane-instructions a ray" % (k, v["lane_trips_per_ray"], v["valu_per_trip"], v["lane_instructions_per_ray"]))
    print("\nlanes a ray that enter each stage: " + "  ".join("%s %.3f" % kv for kv in out["stage_lanes_per_ray"].items()))


if __name__ == "__main__":
    main()
