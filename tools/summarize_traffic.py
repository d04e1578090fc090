"""Writes profiles/hbm_traffic_house_1080p_8b.json from a tools/profile.sh output directory: the
fabric-side bytes per launch of the dominant kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
separate passes, KiB units), which bench.py reports as roofline.traffic."""
import csv, glob, json, os, sys
out, dst = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(out, 'pmc*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rt_render_pool_kernel' in r['Kernel_Name']:
            acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
avg = {k: sum(v) / len(v) for k, v in acc.items()}
res = {k: avg[k] * 1024 for k in ('FETCH_SIZE', 'WRITE_SIZE') if k in avg}
issue = None
if all(k in avg for k in ('SQ_INSTS_VALU', 'SQ_THREAD_CYCLES_VALU', 'SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'GRBM_GUI_ACTIVE')):
    cycles = avg['GRBM_GUI_ACTIVE'] / 8  # the counter sums the 8 XCDs
    issue = {"valu_wave_instructions_per_launch": avg['SQ_INSTS_VALU'],
             "valu_instructions_per_simd_cycle": avg['SQ_INSTS_VALU'] / (cycles * 256 * 4), "valu_pipe_peak_per_simd_cycle": 0.5,
             "lanes_active_per_valu_instruction": avg['SQ_THREAD_CYCLES_VALU'] / avg['SQ_ACTIVE_INST_VALU'] * 4 / 4 if avg['SQ_ACTIVE_INST_VALU'] else None,
             "valu_active_fraction_of_wave_cycles": avg['SQ_ACTIVE_INST_VALU'] / avg['SQ_WAVE_CYCLES'],
             "wait_any_fraction": avg['SQ_WAIT_ANY'] / avg['SQ_WAVE_CYCLES'], "wait_inst_any_fraction": avg['SQ_WAIT_INST_ANY'] / avg['SQ_WAVE_CYCLES']}
json.dump({"kernel": "rt_render_pool_kernel", "fetch_bytes_per_launch": res.get('FETCH_SIZE'), "write_bytes_per_launch": res.get('WRITE_SIZE'),
           "note": "FETCH_SIZE/WRITE_SIZE are KiB at the L2's fabric side (Infinity-Cache hits included); the gfx950 x2 correction of "
                   "FETCH_SIZE applies to wide coalesced streams only, these reads are 16-byte gathers, so it is NOT applied",
           "issue": issue, "source": os.path.basename(out.rstrip('/'))}, open(dst, 'w'), indent=1)
print(open(dst).read())
