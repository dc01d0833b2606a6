#!/bin/bash
# ThreadSanitizer pass over the C++ host side: tests/micro/host_threads.cpp + all host sources in one executable.
set -e
cd "$(dirname "$0")/.."
OUT=${OUT:-/tmp/sg_tsan}; mkdir -p "$OUT"
g++ -fsanitize=thread -g -O1 -std=c++17 -pthread -I include -o "$OUT/host_threads" tests/micro/host_threads.cpp alaz_amd/csrc/host/*.cpp -ldl -lz
TSAN_OPTIONS="halt_on_error=1" "$OUT/host_threads"
