#!/bin/bash
# Host-side AddressSanitizer + UBSan run of the C oracle and the C++ host
# (GPU sanitizers are not available on this pool; device code is not touched).
# The sanitized libraries replace the normal ones for the duration of the run.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/asan
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g"
PRE="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
cp oracle/librq_oracle.so build/asan/oracle_orig.so
cp raft_amd/libraft_host.so build/asan/host_orig.so
restore() { cp build/asan/oracle_orig.so oracle/librq_oracle.so; cp build/asan/host_orig.so raft_amd/libraft_host.so; }
trap restore EXIT
gcc $SAN -std=c11 -fPIC -shared -pthread oracle/raft_oracle.c -o oracle/librq_oracle.so
g++ $SAN -std=c++17 -fPIC -shared -pthread raft_amd/csrc/host/raft_host.cpp raft_amd/csrc/host/raft_host_resident.cpp raft_amd/csrc/host/raft_host_capi.cpp \
    -Lraft_amd -lrq -Wl,-rpath,'$ORIGIN' -o raft_amd/libraft_host.so
LD_PRELOAD="$PRE${LD_PRELOAD:+ $LD_PRELOAD}" ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests/test_oracle_kats.py tests/test_host_cpp.py tests/test_wide_layout.py -x -q -m "not gpu"
