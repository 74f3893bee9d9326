#!/bin/bash
# Host side of the library under AddressSanitizer (CPU build only; GPU sanitizers are not available on the pool).
# Builds an ASan variant of libzkhip.so (host code of every translation unit; the gfx950 code objects are unchanged) in a
# scratch copy of the package and runs the host-only test files against it.  Usage: bash tools/sanitize/asan_host_tests.sh
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=${TMPDIR:-/tmp}/zkhip_asan
CL=/opt/rocm/lib/llvm/bin/clang++
RT=$(find /opt/rocm/lib/llvm -name 'libclang_rt.asan-x86_64.so' | head -1)
rm -rf $W && mkdir -p $W/repo/zecale_amd && cd $W
for f in msm ntt qap zkhip_api witness; do
  hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -DZK_MUL_INLINE=1 -fPIC -fsanitize=address -shared-libsan -Wno-option-ignored -c $ROOT/zecale_amd/csrc/$f.hip -o $f.o &
done
for f in aggregator pipeline witness_tape multi_device; do
  $CL -O1 -g -std=c++17 -fPIC -pthread -fsanitize=address -shared-libsan -c $ROOT/zecale_amd/csrc/$f.cpp -o $f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -fsanitize=address -shared-libsan -o repo/zecale_amd/libzkhip.so msm.o ntt.o qap.o zkhip_api.o witness.o aggregator.o pipeline.o witness_tape.o multi_device.o
cp $ROOT/zecale_amd/*.py repo/zecale_amd/
for d in tests oracle include bench.py; do ln -s $ROOT/$d repo/$d; done
cd repo
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$RT python -m pytest tests/test_verify_host.py tests/test_keypair_file.py tests/test_encoding.py \
  tests/test_aggregator_host.py tests/test_abi.py -q -p no:cacheprovider
