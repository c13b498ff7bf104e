#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the HOST half of liblmx.so (yaml reader/writer, bank and cache code, trainer host
# stages, merge / cluster chain, group bookkeeping), on the CPU, through the "not gpu" test suite.  The device code objects are the
# regular ones (the sanitizers do not apply to gfx950 without xnack+, and GPU ASan is not available on this pool).
#   scripts/sanitize_host.sh [pytest args]        -> build/asan/liblmx.so, reports under build/asan/*.log.*, exit code of pytest
#   SANITIZE=undefined BUILD_ONLY=1 scripts/sanitize_host.sh   -> build/ubsan/liblmx.so only: the UBSan-only library also runs on a GPU box
#       (LD_PRELOAD=libclang_rt.ubsan_standalone-x86_64.so LMX_SO_PATH=build/ubsan/liblmx.so pytest -m gpu); ROCm's ASan runtime intercepts
#       the HSA allocator and needs xnack+, so the address half stays on the CPU
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
san=${SANITIZE:-address,undefined}
out=$root/build/asan
[ "$san" = "undefined" ] && out=$root/build/ubsan
mkdir -p $out
cd $root/linemod_pose_estimation_amd/csrc
make -s                                            # the regular objects: lmx_kernels.o and lmx_f2.o are linked as they are
flags="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=$san -fno-omit-frame-pointer -Wno-option-ignored -I../../include -I."
for f in lmx_yaml lmx_hostcopy; do /opt/rocm/bin/hipcc $flags -c -o $out/$f.o $f.cpp; done
host="lmx_bank lmx_ctx lmx_enqueue lmx_collect lmx_cluster lmx_cache lmx_debug lmx_train lmx_group"   # the Makefile's host objects
for f in $host; do /opt/rocm/bin/hipcc $flags -x hip -c -o $out/$f.o $f.cpp; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=$san -shared-libsan -Wno-option-ignored -o $out/liblmx.so lmx_kernels.o lmx_f2.o \
  $(for f in $host; do echo $out/$f.o; done) $out/lmx_yaml.o $out/lmx_hostcopy.o -ldl -lpthread
[ -n "$BUILD_ONLY" ] && exit 0
asan=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
[ "$san" = "undefined" ] && asan=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1)
cd $root
rm -f $out/asan.log.* $out/ubsan.log.*
set +e
LD_PRELOAD=$asan ASAN_OPTIONS=detect_leaks=0:log_path=$out/asan.log UBSAN_OPTIONS=print_stacktrace=1:log_path=$out/ubsan.log LMX_SO_PATH=$out/liblmx.so \
  python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
rc=$?
n=$(ls $out/asan.log.* $out/ubsan.log.* 2>/dev/null | wc -l)
echo "sanitizer reports: $n (under $out)"
[ $n -eq 0 ] || { grep -h "runtime error\|ERROR: AddressSanitizer" $out/*.log.* | sort | uniq -c | head -20; exit 1; }
exit $rc
