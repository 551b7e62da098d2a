#!/bin/bash
# CPU-only: rebuild the two checkers under AddressSanitizer + UBSan and run their CPU tests (GPU ASan is not
# available on the pool, so the sanitizers cover the test infrastructure, not the kernels).
set -euo pipefail
cd "$(dirname "$0")/.."
TMP=$(mktemp -d)
cp oracle/libecsimd_oracle.so "$TMP/oracle.so"; cp oracle/libecsimd_ossl.so "$TMP/ossl.so" 2>/dev/null || true
restore() { cp "$TMP/oracle.so" oracle/libecsimd_oracle.so; [ -f "$TMP/ossl.so" ] && cp "$TMP/ossl.so" oracle/libecsimd_ossl.so; rm -rf "$TMP"; }
trap restore EXIT
FLAGS="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared"
gcc $FLAGS -o oracle/libecsimd_oracle.so oracle/ecsimd_oracle.c -lpthread
[ -f /usr/include/openssl/ec.h ] && gcc $FLAGS -o oracle/libecsimd_ossl.so oracle/ossl_check.c -lcrypto -lpthread
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 \
  python -m pytest tests/test_oracle.py tests/test_oracle_properties.py tests/test_openssl_crosscheck.py "tests/test_bench_contract.py::test_config1_ops8_times_the_reference_and_hashes_its_outputs" -x -q -m "not gpu"
