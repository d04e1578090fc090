"""Writes profiles/hbm_traffic_house_1080p_8b.json from a tools/profile.sh output directory: the
fabric-side bytes per launch of the dominant kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
separate passes, KiB units), which bench.py reports as roofline.traffic."""
import csv, glob, json, os, sys
out, dst = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(out, 'pmc*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rt_render_pool_kernel' in r['Kernel_Name'] and r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
res = {k: sum(v) / len(v) * 1024 for k, v in acc.items()}
json.dump({"kernel": "rt_render_pool_kernel", "fetch_bytes_per_launch": res.get('FETCH_SIZE'), "write_bytes_per_launch": res.get('WRITE_SIZE'),
           "note": "FETCH_SIZE/WRITE_SIZE are KiB at the L2's fabric side (Infinity-Cache hits included); the gfx950 x2 correction of "
                   "FETCH_SIZE applies to wide coalesced streams only, these reads are 16-byte gathers, so it is NOT applied",
           "source": os.path.basename(out.rstrip('/'))}, open(dst, 'w'), indent=1)
print(open(dst).read())
