#!/bin/bash
# AddressSanitizer + UBSan pass over the CPU oracle (test infrastructure) through its own golden tests.  CPU only (GPU ASan is not available on the pool).
# Builds the instrumented libraries over oracle/liborc*.so, runs the oracle tests with the runtime preloaded, then rebuilds the plain libraries.
set -e
cd "$(dirname "$0")/.."
SAN="-O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined"
make -C oracle clean >/dev/null
make -C oracle CFLAGS="$SAN -fPIC -Wall -Wextra -std=c11 -D_POSIX_C_SOURCE=200809L -D_XOPEN_SOURCE=700 -ffp-contract=off" >/dev/null
rc=0
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
  python -m pytest tests/test_oracle_golden.py tests/test_feti_goldens.py tests/test_dist_gloo.py -x -q -m "not gpu" -p no:cacheprovider || rc=$?
make -C oracle clean >/dev/null
make -C oracle >/dev/null
exit $rc
