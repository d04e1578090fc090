#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU code: librsrt_host.so (scene / OBJ / HDR loaders, BVH builder, alias table, PNG / PFM
# writers) and the oracle, built into /tmp/rsrt_san and driven by the CPU tests that exercise them.  (GPU sanitizers are not available on this
# pool: the kernels are covered by the parity tests instead.)      bash tools/sanitize_host.sh [pytest args | --fuzz [trials] [seed]]
set -e
ARGS=("$@")
R=$(cd "$(dirname "$0")/.." && pwd)
O=/tmp/rsrt_san; mkdir -p $O
SAN="-O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined"
g++ $SAN -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wextra -I$R/include -o $O/librsrt_host.so \
    $R/rsoderh-raytracing_amd/csrc/host/bvh_build.cpp $R/rsoderh-raytracing_amd/csrc/host/preprocess.cpp \
    $R/rsoderh-raytracing_amd/csrc/host/scene_load.cpp $R/rsoderh-raytracing_amd/csrc/host/image_io.cpp
FMA=$(grep -q -m1 ' fma ' /proc/cpuinfo && echo -mfma || true)
for v in "liboracle.so" "liboracle_fast.so" "liboracle_ops.so -DORC_COUNT_OPS"; do
  read -r name def <<< "$v"
  g++ $SAN -std=c++17 -fPIC -shared -fopenmp -ffp-contract=off -fno-fast-math $FMA $def -I$R/include -o $O/$name $R/oracle/rt_oracle.cpp
done
cp $R/oracle/liboracle_libm.so $O/ 2>/dev/null || true   # (the sensitivity build is not a parity reference: as it is)
cd $R
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export RSRT_HOST_LIB=$O/librsrt_host.so ORACLE_LIB_DIR=$O OMP_NUM_THREADS=4
if [ "${ARGS[0]}" = "--fuzz" ]; then  # the loaders' mutation fuzz under the sanitizers instead of the tests
  python tools/fuzz_loaders.py ${ARGS[1]:-40000} ${ARGS[2]:-1}
  exit $?
fi
if [ ${#ARGS[@]} -eq 0 ]; then ARGS=(tests/test_host_preprocess.py tests/test_golden.py tests/test_oracle_kat.py tests/test_display.py tests/test_independent_geometry.py tests/test_independent_shading.py tests/test_box_containment.py); fi
python -m pytest -q -x -m "not gpu" -p no:cacheprovider "${ARGS[@]}"

# Part 2: the HOST pass of librsrt.so (rsrt_api.hip's upload-time code: wide-tree collapse, flat leaves, visiting ranks, partition arithmetic,
# argument checks) under AddressSanitizer — clang's, so in a process of its own; the device code objects are built as always (-fno-gpu-sanitize)
# and never run here.  Skipped with SAN_SKIP_HIP=1 (the build takes a minute).
if [ -z "$SAN_SKIP_HIP" ]; then
  unset LD_PRELOAD RSRT_HOST_LIB ORACLE_LIB_DIR UBSAN_OPTIONS
  RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fno-vectorize \
      -fsanitize=address -fno-gpu-sanitize -shared-libsan -I $R/include -DRSRT_BUILD_ID='"asan-host"' -o $O/librsrt.so $R/rsoderh-raytracing_amd/csrc/hip/rsrt_api.hip
  LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 RSRT_LIB=$O/librsrt.so python -m pytest -q -x -m "not gpu" -p no:cacheprovider "${ARGS[@]}"
fi
