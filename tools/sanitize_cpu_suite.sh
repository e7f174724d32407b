#!/bin/bash
# ASan + UBSan over the HOST code of libgandiva_amd.so under the CPU test suite (round 6, verdict item 8; rounds 1-3 did the same
# by hand).  The library is rebuilt with -fsanitize=address,undefined into a scratch directory and swapped in through
# GANDIVA_AMD_LIB; device code is not under the sanitizers (no GPU ASan on this pool).  Writes profiles/<round>_sanitizers.txt.
#      ROUND=r06 bash tools/sanitize_cpu_suite.sh
set -e
R=$(cd $(dirname $0)/.. && pwd)
ROUND=${ROUND:-r06}
B=/tmp/gdv_asan_build
rm -rf $B; mkdir -p $B
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined"
make -C $R/gandiva_amd/csrc BUILD=$B OUT=$B/libgandiva_amd.so \
  CXXFLAGS="-O1 -g -fPIC -std=c++17 -Wall -Wno-unused-function -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $SAN" \
  LINK=g++ LDFLAGS="-shared -L/opt/rocm/lib -lamdhip64 -lhiprtc -ldl -lpthread -Wl,-rpath,/opt/rocm/lib $SAN" -j8 > $B/build.log 2>&1 || { tail -20 $B/build.log; exit 1; }
RT="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
cd $R
TESTS="tests/test_api.py tests/test_planner_cpu.py tests/test_proto_build.py tests/test_kernel_identity.py tests/test_tier0.py tests/test_sharded_call.py tests/test_device_pool.py tests/test_filter_project.py tests/test_shard.py tests/test_jni_flat.py tests/test_registry_tail.py tests/test_fuzz_trees.py tests/test_decimal.py"
GDV_EVIDENCE_PENDING=1 GANDIVA_AMD_LIB=$B/libgandiva_amd.so LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  timeout 3000 python -m pytest $TESTS -q -x -m "not gpu" -p no:cacheprovider > $B/pytest.log 2>&1 || true
{ echo "# $ROUND: ASan + UBSan (g++ $(g++ -dumpversion)) over the host code of libgandiva_amd.so — gdv_node / registry / libtag / proto /"
  echo "# regex / planner / runtime / pool / tier0 / engine / c_api — under the CPU tests below (library rebuilt with -fsanitize=address,undefined,"
  echo "# loaded through GANDIVA_AMD_LIB; detect_leaks=0: the process is Python).  tools/sanitize_cpu_suite.sh"
  echo "# tests: $TESTS"
  tail -3 $B/pytest.log
  echo "# sanitizer reports in the log: $(grep -c -E 'ERROR: AddressSanitizer|runtime error:' $B/pytest.log || true)"
  grep -E 'ERROR: AddressSanitizer|runtime error:' $B/pytest.log | sort | uniq -c | head -20
} > $R/profiles/${ROUND}_sanitizers.txt
cat $R/profiles/${ROUND}_sanitizers.txt
