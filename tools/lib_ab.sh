#!/bin/bash
# bash tools/lib_ab.sh [spp]   (GPU box) — the BASELINE frame (house 1080p, 8 bounces, spp samples) timed with the product library and
# with every experiment build librsrt_exp_*.so lying in the package directory (tools/house_knobs.py each, three rounds; the build id
# printed first names the flags of each).  Twice round the list, so that drift of the box shows.
R=${GRAFT_REPO_ROOT:-/root/repo}
SPP=${1:-64}
for pass in 1 2; do
  for lib in $R/rsoderh-raytracing_amd/librsrt.so $R/rsoderh-raytracing_amd/librsrt_exp_*.so; do
    [ -f "$lib" ] || continue
    RSRT_LIB=$lib python3 $R/tools/house_knobs.py "" $SPP | awk -v p=$pass '/^library/ {id=$2} /^round [12]/ {print "pass " p, id, $0}'
  done
done
