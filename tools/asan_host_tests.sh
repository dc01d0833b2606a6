#!/bin/bash
# Memory-safety pass over the host-side parsers (they read untrusted network payloads): builds the C++ host library and the
# C oracle with AddressSanitizer + UBSan into a scratch directory and runs the CPU tests that drive them (scenario,
# differential and fuzz tests of HTTP/2, Kafka, socket lines, packer) against those builds.  CPU only.
set -e
cd "$(dirname "$0")/.."
OUT=${OUT:-/tmp/sg_asan}; mkdir -p "$OUT"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"
g++ $SAN -std=c++17 -fPIC -shared -Wall -Wextra -pthread -I include -o "$OUT/libsgdatastore.so" alaz_amd/csrc/host/*.cpp -ldl -lz
gcc $SAN -std=gnu99 -fPIC -shared -ffp-contract=off -Wno-format-truncation -o "$OUT/libsgoracle.so" oracle/sg_oracle.c oracle/sockline.c oracle/http2.c oracle/kafka.c -lm -lz -ldl
export SG_HOST_LIB_PATH="$OUT/libsgdatastore.so" SG_ORACLE_LIB_PATH="$OUT/libsgoracle.so"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1
python -m pytest tests/test_http2.py tests/test_kafka.py tests/test_sockline.py tests/test_sockline_proc.py tests/test_host.py tests/test_oracle_golden.py -x -q -p no:cacheprovider "$@"
python tools/fuzz_host_parsers.py ${FUZZ_ITERS:-20000} ${FUZZ_SEED:-1}
