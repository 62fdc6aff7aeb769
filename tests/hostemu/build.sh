#!/bin/bash
# Builds the CPU-emulated test double of the library from the product sources (TEST INFRASTRUCTURE ONLY).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$HERE/_build"
OUT="$HERE/_build/libvkfft_hostemu.so"
SRCS="$ROOT/vkfft_amd/csrc/api.cpp $ROOT/vkfft_amd/csrc/planner.cpp $ROOT/vkfft_amd/csrc/kernels.hip $ROOT/vkfft_amd/csrc/kernels_mixed.hip $ROOT/vkfft_amd/csrc/mixed_table.inc $HERE/hostemu_runtime.cpp"
newest=$(ls -t $SRCS $ROOT/vkfft_amd/csrc/*.h $ROOT/include/vkFFT.h $HERE/hostemu_runtime.h | head -1)
if [ -f "$OUT" ] && [ "$OUT" -nt "$newest" ]; then exit 0; fi
CXX=${CXX:-g++}
FLAGS="-O2 -std=c++17 -fPIC -DVKFFT_HOSTEMU -I$HERE -I$ROOT/include -I$ROOT/vkfft_amd/csrc -I/opt/rocm/include -fvisibility=hidden -Wno-unused-result -Wno-attributes"
pids=()
$CXX $FLAGS -c $ROOT/vkfft_amd/csrc/api.cpp -o $HERE/_build/api.o & pids+=($!)
$CXX $FLAGS -c $ROOT/vkfft_amd/csrc/planner.cpp -o $HERE/_build/planner.o & pids+=($!)
$CXX $FLAGS -x c++ -c $ROOT/vkfft_amd/csrc/kernels.hip -o $HERE/_build/kernels.o & pids+=($!)
$CXX $FLAGS -x c++ -c $ROOT/vkfft_amd/csrc/kernels_mixed.hip -o $HERE/_build/kernels_mixed.o & pids+=($!)
$CXX $FLAGS -c $HERE/hostemu_runtime.cpp -o $HERE/_build/rt.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
$CXX -shared -fPIC $HERE/_build/api.o $HERE/_build/planner.o $HERE/_build/kernels.o $HERE/_build/kernels_mixed.o $HERE/_build/rt.o -o "$OUT"
echo "built $OUT"
