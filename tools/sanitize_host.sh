#!/bin/bash
# Sanitizers over the host side of libdcsg_hip (no GPU needed; GPU sanitizers are not available on the pool).
#   bash tools/sanitize_host.sh            AddressSanitizer + UBSan: parsers, optimiser, table expansion, writers on the
#                                          fixture scenes and ~1300 mutated inputs (tools/sanitize_host_calls.py)
#   bash tools/sanitize_host.sh --thread   ThreadSanitizer: concurrent JIT, the background build of dcsg_load_scene with an
#                                          early return, the writers' threads and 2-8 range writers on one file pair
#                                          (tools/tsan_host.cpp); report -> profiles/r04_tsan_report.txt
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python3 "$ROOT/designcsg_amd/build.py" > /dev/null       # embedded_cl.inc and friends
SRC="dcsg_api.cpp scene.cpp jit.cpp specialise.cpp"
COMMON="-std=c++17 -O1 -g -ffp-contract=off -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I../_build -I."
if [ "$1" = "--thread" ]; then
    OUT=/tmp/dcsg_tsan
    rm -rf $OUT; mkdir -p $OUT/tmp
    cd "$ROOT/designcsg_amd/csrc"
    g++ $COMMON -fPIC -shared -fsanitize=thread $SRC -o $OUT/libdcsg_hip.so -L/opt/rocm/lib -lamdhip64 -ldl -lpthread -Wl,-rpath,/opt/rocm/lib
    # the kernels' bitcode sits next to the product library: the JIT looks for it beside the library it is in
    cp "$ROOT"/designcsg_amd/_build/*.bc $OUT/
    g++ -std=c++17 -O1 -g -fsanitize=thread "$ROOT/tools/tsan_host.cpp" -o $OUT/tsan_host -L$OUT -ldcsg_hip -lpthread -Wl,-rpath,$OUT
    cd "$ROOT"
    REPORT=profiles/r04_tsan_report.txt
    set +e
    # no suppressions: libamd_comgr is entered by one thread at a time (jit.cpp: run_action_), which is what its globals need
    TSAN_OPTIONS="halt_on_error=0 exitcode=66" $OUT/tsan_host "$ROOT" $OUT/tmp > $OUT/out.txt 2> $OUT/err.txt
    RC=$?
    set -e
    { echo "# ThreadSanitizer run of tools/tsan_host.cpp against the host side of libdcsg_hip (g++ -fsanitize=thread, $(g++ --version | head -1))";
      echo "# exit code $RC (0 = no failed call and no report; 66 = the sanitizer reported)";
      echo "## stdout"; cat $OUT/out.txt; echo "## stderr (sanitizer reports)"; cat $OUT/err.txt; } > $REPORT
    cat $REPORT
    exit $RC
fi
OUT=/tmp/dcsg_asan
mkdir -p $OUT
cd "$ROOT/designcsg_amd/csrc"
g++ $COMMON -fPIC -shared -fsanitize=address,undefined $SRC \
    -o $OUT/libdcsg_hip.so -L/opt/rocm/lib -lamdhip64 -ldl -lpthread -Wl,-rpath,/opt/rocm/lib
cd "$ROOT"
DCSG_ASAN_LIB=$OUT/libdcsg_hip.so ASAN_OPTIONS=detect_leaks=0 \
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) python3 tools/sanitize_host_calls.py
