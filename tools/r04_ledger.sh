#!/bin/bash
# The overhead ledger's ablations: the house frame's retired VALU lane-instructions with the product library, with a build whose
# divisions and square roots are the hardware's 2.5-ulp forms (-fno-hip-fp32-correctly-rounded-divide-sqrt) and with one whose exact reciprocal is
# the bare v_rcp_f32 too (-DRT_FAST_RCP).  The differences are what the IEEE forms cost.  (The images of the two builds are NOT the product's.)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_ledger; mkdir -p $O
cd $R
for v in product b93b1a04be 402bff4434; do
  if [ $v = product ]; then unset RSRT_LIB; else export RSRT_LIB=$R/rsoderh-raytracing_amd/librsrt_exp_$v.so; fi
  timeout -k 10 280 bash tools/pmc_scene.sh ledger_$v house 1920 1080 64 8 "" > $O/$v.txt 2>&1 || { tail $O/$v.txt; exit 1; }
  grep "ms/frame\|SQ_INSTS_VALU\|SQ_THREAD_CYCLES_VALU\|SQ_ACTIVE_INST_VALU" $O/$v.txt
done
unset RSRT_LIB
timeout -k 10 300 python tools/simd_efficiency.py 4 64 > $O/simd_house.txt 2>&1; tail -3 $O/simd_house.txt; cp gpurun_out/stage_shares_house.json $O/
