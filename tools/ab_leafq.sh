#!/bin/bash
# scratch: A/B of RT_LEAFQ builds on the GPU box
R=${GRAFT_REPO_ROOT:-/root/repo}; P=$R/rsoderh-raytracing_amd
for q in 2 3 4; do
  cp $P/librsrt_q$q.so $P/librsrt.so; cp $P/librsrt_instr_q$q.so $P/librsrt_instr.so
  echo "=== LEAFQ $q"
  timeout -k 10 300 python $R/tools/ab_kernels.py 2 32 2>&1 | grep -E "False|round [12]"
  for b in 24 64; do RSRT_TRACE_BUDGET=$b timeout -k 10 200 python $R/tools/simd_efficiency.py 2 16 2>&1 | grep -E "variant|TRACE|descend loop|leaf loop"; done
done
