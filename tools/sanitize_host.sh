#!/bin/bash
# AddressSanitizer + UBSan over the host side of libdcsg_hip (no GPU needed; GPU sanitizers are not available on
# the pool).  bash tools/sanitize_host.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/dcsg_asan
mkdir -p $OUT
python3 "$ROOT/designcsg_amd/build.py" > /dev/null       # embedded_cl.inc and friends
cd "$ROOT/designcsg_amd/csrc"
g++ -std=c++17 -O1 -g -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer \
    -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I../_build -I. dcsg_api.cpp scene.cpp jit.cpp specialise.cpp \
    -o $OUT/libdcsg_hip.so -L/opt/rocm/lib -lamdhip64 -ldl -lpthread -Wl,-rpath,/opt/rocm/lib
cd "$ROOT"
DCSG_ASAN_LIB=$OUT/libdcsg_hip.so ASAN_OPTIONS=detect_leaks=0 \
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) python3 tools/sanitize_host_calls.py
