// placeholder until the specialised power-of-two kernels land
#pragma once
#include "engine.h"
namespace vkfft_mi355x {
inline int launch_pow2(const PassPlan&, const PassParams&, hipStream_t) { return 4039; }
inline bool pow2_row_available(uint32_t, bool, uint32_t*, uint32_t*, size_t*, int*) { return false; }
}
