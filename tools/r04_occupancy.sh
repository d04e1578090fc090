#!/bin/bash
# Is the cooperative walk short of WAVES?  256-thread workgroups, every node read from global memory (RSRT_HYBRID=0) so that the comparison is
# between like and like: 128-slot pools (16 waves a CU), 96-slot pools (still 16: the registers cap it), 96-slot pools compiled for five waves
# a SIMD (20 waves a CU, 96 registers, ~37 spilled).  The product's choice (one 1024-thread workgroup, nodes staged in LDS) beside them.
O=gpurun_out/r04_occupancy; mkdir -p $O; rm -f $O/*.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$R/rsoderh-raytracing_amd
for pass in 1 2; do
  for v in product hybrid0_pool128 hybrid0_pool96 hybrid0_pool96_wps5 lds_pool96; do
    unset RSRT_LIB RSRT_HYBRID
    case $v in
      hybrid0_pool128) export RSRT_HYBRID=0;;
      hybrid0_pool96) export RSRT_HYBRID=0 RSRT_LIB=$P/librsrt_exp_8da96e05a2.so;;
      hybrid0_pool96_wps5) export RSRT_HYBRID=0 RSRT_LIB=$P/librsrt_exp_819bc00f23.so;;
      lds_pool96) export RSRT_LIB=$P/librsrt_exp_8da96e05a2.so;;
    esac
    echo "== $v" >> $O/ab.txt
    timeout -k 10 200 python tools/scene_time.py suzanne 1280 720 128 10 2>&1 | tail -1 >> $O/ab.txt &&
    timeout -k 10 200 python tools/scene_time.py grid4 1280 720 32 10 2>&1 | tail -1 >> $O/ab.txt || { tail $O/ab.txt; exit 1; }
  done
done
cat $O/ab.txt | cut -c1-200
