#!/bin/bash
# Host-side sanitizer run (CPU only; GPU AddressSanitizer is not available on this pool): builds the library with
# AddressSanitizer + UndefinedBehaviorSanitizer on its HOST side only (-Xarch_host: the device code is the usual one) into
# build/libbtf_host_asan.so, then runs btf_host_selftest() - the host-made tables, LDS layouts, elimination orders and
# chunk maps over a grid of shapes - from a driver built with the same sanitizers.     scripts/asan_host.sh
set -e
cd "$(dirname "$0")/.."
mkdir -p build
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
XSAN="-Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -g"
BTF_LIB_PATH=$PWD/build/libbtf_host_asan.so BTF_BUILD_DEFS="$XSAN" BTF_LINK_FLAGS="-fsanitize=address,undefined" \
  python -c "from functionalmf_amd import _native; _native.build(force=True)" 2>&1 | grep -v "warning" | tail -5
cat > build/asan_driver.cpp <<'CPP'
#include <cstdio>
extern "C" int btf_host_selftest(void);
int main() { const int rc = btf_host_selftest(); std::printf("btf_host_selftest: %d\n", rc); return rc != 0; }
CPP
CLANG=/opt/rocm/lib/llvm/bin/clang++
$CLANG $SAN build/asan_driver.cpp -o build/asan_driver -L build -lbtf_host_asan -Wl,-rpath,$PWD/build -Wl,-rpath,/opt/rocm/lib
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 build/asan_driver
