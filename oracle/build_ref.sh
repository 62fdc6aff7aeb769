#!/bin/bash
# Builds oracle/_ref/ from the reference sources where they lie (read-only /root/reference).
# Test infrastructure only; outputs are git-ignored but travel to the GPU box with gpurun.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=${VKFFT_REFERENCE_DIR:-/root/reference}
[ -f "$REF/vkFFT/vkFFT.h" ] || { echo "reference not present at $REF; keeping prebuilt oracle/_ref" ; exit 0; }
mkdir -p "$HERE/_ref"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -O2 -std=c++17 -fPIC -shared -DVKFFT_BACKEND=2 -Wno-everything -I"$REF/vkFFT" \
   "$HERE/ref_vkfft_wrapper.cpp" -o "$HERE/_ref/libvkfft_ref.so" -lhiprtc
$HIPCC -O2 -std=c++17 -DVKFFT_BACKEND=2 -DREF_MAIN -Wno-everything -I"$REF/vkFFT" \
   "$HERE/ref_vkfft_wrapper.cpp" -o "$HERE/_ref/vkfft_ref_bench" -lhiprtc
# drop-in proof: the reference's own benchmark caller, compiled UNCHANGED against this repository's header and library
ROOT="$(cd "$HERE/.." && pwd)"
if [ -f "$ROOT/vkfft_amd/lib/libvkfft_mi355x.so" ]; then
  BS="$REF/benchmark_scripts/vkFFT_scripts"
  $HIPCC -O2 -std=c++17 -DVKFFT_BACKEND=2 -Wno-everything -I"$ROOT/include" -I"$BS/include" \
     "$BS/src/sample_0_benchmark_VkFFT_single.cpp" "$BS/src/utils_VkFFT.cpp" "$HERE/dropin_main.cpp" \
     -L"$ROOT/vkfft_amd/lib" -lvkfft_mi355x -Wl,-rpath,'$ORIGIN/../../vkfft_amd/lib' -o "$HERE/_ref/dropin_sample0"
fi
echo "built $HERE/_ref/libvkfft_ref.so, vkfft_ref_bench and dropin_sample0"
