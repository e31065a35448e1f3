#!/bin/bash
# AddressSanitizer pass over the HOST side of libpermonhip.so (set-up producers, planners, converters: everything tests/test_host_logic.py and tests/test_sa_host.py reach without a GPU).
# CPU only: device code is NOT instrumented (-fno-gpu-sanitize; GPU ASan / xnack+ is not available on the pool).  Builds into /tmp/asanlib, swaps the library
# in for the run and restores the plain one.  tests/test_abi.py is left out: it links the plain-C examples against the library, which then needs the ASan runtime.
set -e
cd "$(dirname "$0")/.."
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -n 1)
mkdir -p /tmp/asanlib
SRCS=$(sed -n 's/^SRCS *= *//p' permon_amd/csrc/Makefile)
( cd permon_amd/csrc && for f in $SRCS; do echo ${f%.hip}; done | xargs -P 8 -I{} /opt/rocm/bin/hipcc -O1 -g -fno-omit-frame-pointer -fsanitize=address -fno-gpu-sanitize -Wno-unused-value -std=c++17 -fPIC \
    --offload-arch=gfx950 -ffp-contract=off -I../../include -I. -c {}.hip -o /tmp/asanlib/{}.o )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -shared-libsan -shared -fPIC -o /tmp/asanlib/libpermonhip.so /tmp/asanlib/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
cp permon_amd/libpermonhip.so /tmp/asanlib/libpermonhip.plain.so
cp /tmp/asanlib/libpermonhip.so permon_amd/libpermonhip.so
rc=0
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 python -m pytest tests/test_host_logic.py tests/test_sa_host.py -x -q -m "not gpu" -p no:cacheprovider || rc=$?
cp /tmp/asanlib/libpermonhip.plain.so permon_amd/libpermonhip.so
exit $rc
