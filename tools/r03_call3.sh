#!/bin/bash
# round 3, GPU call 3: counters of the wide walk against the fixed-order walk; interactive overlap A/B; the tests the lanes touch
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_call3
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 100 -k "compose or caller_streams or golden_images or progressive or passes" > $O/lane_tests.txt 2>&1; echo "pytest exit $?" >> $O/lane_tests.txt
tail -4 $O/lane_tests.txt
grep -q "pytest exit 0" $O/lane_tests.txt || exit 1
timeout -k 10 200 python tools/interactive_ab.py 1 16 > $O/interactive_ab.txt 2>&1; cat $O/interactive_ab.txt
for t in 3 4; do
  for sc in suzanne grid; do
    if [ $sc = grid ]; then A="--scene /tmp/rsrt_scenes/suzanne_grid_4.toml --width 1280 --height 720 --spp 32 --bounces 10"; python -c "import sys; sys.path.insert(0,'tools'); import make_big_scene; make_big_scene.make(4)"; else A="--scene suzanne --width 1280 --height 720 --spp 128 --bounces 10"; fi
    RSRT_TRAVERSAL=$t timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $A > $O/bench_${sc}_t$t.json 2> $O/bench_${sc}_t$t.err || tail -3 $O/bench_${sc}_t$t.err
  done
done
python - <<'PY'
import json,os
O=os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'gpurun_out/r03_call3')
for sc in ('suzanne','grid'):
    for t in (3,4):
        try:
            j=json.load(open(os.path.join(O,'bench_%s_t%d.json'%(sc,t)))); r=j['roofline']; v=r['valu']; c=r['counters']
            rays=j['config']['rays_per_frame']
            print('%s trav %d: %.2f ms  util %.3f  issue %.3f lanes %.1f  wait_any %.2f  VALU wave-instr/ray %.1f  LDS instr/ray %.2f  VMEM rd/ray %.2f  salu/valu %.2f  l2hit %s' % (sc,t,j['ms_per_step'],r['utilisation'],v['issue_frac'],v['lanes_active_per_instruction'],v['wait_any_frac_of_wave_cycles'],c['SQ_INSTS_VALU']/rays,c.get('SQ_INSTS_LDS',0)/rays,c.get('SQ_INSTS_VMEM_RD',0)/rays,v['salu_per_valu'],(r.get('hbm') or {}).get('l2_hit_rate')))
        except Exception as e: print(sc,t,'failed',e)
PY
