#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; P=$R/rsoderh-raytracing_amd
for c in $(ls $P/librsrt_c*.so | sed 's/.*librsrt_c//; s/.so//'); do
  cp $P/librsrt_c$c.so $P/librsrt.so; cp $P/librsrt_instr_c$c.so $P/librsrt_instr.so
  echo "=== cfg $c"
  timeout -k 10 300 python $R/tools/ab_kernels.py 2 32 2>&1 | grep -E "False|round 2"
  timeout -k 10 200 python $R/tools/simd_efficiency.py 2 16 2>&1 | grep -E "TRACE|descend loop|leaf loop"
done
