#!/bin/bash
# round 3, GPU call 1: GPU test-suite, env-packing A/B (time + fabric bytes), config table
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_call1
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest exit $?" >> $O/gpu_tests.txt
tail -5 $O/gpu_tests.txt
grep -q "pytest exit 0" $O/gpu_tests.txt || exit 1
timeout -k 10 400 bash tools/lib_ab.sh 64 > $O/env_ab.txt 2>&1; cat $O/env_ab.txt
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_packed.json 2> $O/bench_packed.err; tail -3 $O/bench_packed.err
for lib in $R/rsoderh-raytracing_amd/librsrt_exp_*.so; do
  RSRT_LIB=$lib timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_unpacked.json 2> $O/bench_unpacked.err; tail -3 $O/bench_unpacked.err
done
python - <<'PY'
import json,os
O=os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'gpurun_out/r03_call1')
for n in ('packed','unpacked'):
    try:
        j=json.load(open(os.path.join(O,'bench_%s.json'%n)))
        h=j['roofline'].get('hbm') or {}
        print(n, 'ms/frame %.2f'%j['ms_per_step'], 'fetch GB %.1f'%(h.get('fetch_bytes_per_launch',0)/1e9), 'write GB %.1f'%(h.get('write_bytes_per_launch',0)/1e9), 'l2 hit', h.get('l2_hit_rate'), 'frac', j['roofline']['frac'])
    except Exception as e: print(n,'failed',e)
PY
timeout -k 10 400 python tools/config_table.py > $O/config_table.txt 2>&1; cat $O/config_table.txt; cp $R/gpurun_out/config_table.json $O/ 2>/dev/null
