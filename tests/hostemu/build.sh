#!/bin/bash
# Builds the CPU-emulated test double of the library from the product sources (TEST INFRASTRUCTURE ONLY).
# Every translation unit is recompiled only when it or a header / generated table is newer than its object.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$HERE/_build"
OUT="$HERE/_build/libvkfft_hostemu.so"
CSRC="$ROOT/vkfft_amd/csrc"
CXX=${CXX:-g++}
FLAGS="-O2 -std=c++17 -fPIC -DVKFFT_HOSTEMU -I$HERE -I$ROOT/include -I$CSRC -I/opt/rocm/include -fvisibility=hidden -Wno-unused-result -Wno-attributes"
newest_hdr=$(ls -t $CSRC/*.h $CSRC/*.inc $ROOT/include/vkFFT.h $HERE/hostemu_runtime.h | head -1)
pids=(); objs=(); rebuilt=0
for src in $CSRC/api.cpp $CSRC/planner.cpp $CSRC/kernels.hip $CSRC/kernels_pow2.hip $CSRC/kernels_blue_r2r.hip $CSRC/kernels_fused.hip $CSRC/kernels_mixfused.hip $CSRC/kernels_aux.hip $CSRC/kernels_mixed_*.hip $CSRC/kernels_mixconv_*.hip $CSRC/kernels_opfft_*.hip $HERE/hostemu_runtime.cpp; do
	obj="$HERE/_build/$(basename "${src%.*}").o"
	objs+=("$obj")
	if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$newest_hdr" -nt "$obj" ]; then
		$CXX $FLAGS -x c++ -c "$src" -o "$obj" & pids+=($!)
		rebuilt=1
	fi
done
for p in "${pids[@]}"; do wait $p; done
if [ $rebuilt = 1 ] || [ ! -f "$OUT" ]; then
	$CXX -shared -fPIC "${objs[@]}" -o "$OUT"
	echo "built $OUT"
fi
