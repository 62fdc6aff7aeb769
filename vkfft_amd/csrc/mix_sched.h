// Compile-time radix schedule shared by the mixed-radix kernel families (kernel_mixed.h, kernel_opfft.h).
#pragma once
#include "common.h"

namespace vkfft_mi355x {

template <int R0, int R1, int R2, int R3, int R4> struct MixSched {
	static constexpr int rad[5] = {R0, R1, R2, R3, R4};
	static constexpr int NS = (R0 > 1) + (R1 > 1) + (R2 > 1) + (R3 > 1) + (R4 > 1);
	static constexpr int N = R0 * R1 * R2 * R3 * R4;
	__host__ __device__ static constexpr int S(int si) { return si == 0 ? 1 : si == 1 ? R0 : si == 2 ? R0 * R1 : si == 3 ? R0 * R1 * R2 : R0 * R1 * R2 * R3; }
	__host__ __device__ static constexpr int lutOff(int si) { // complex elements before stage si's twiddle run
		int off = 0;
		for (int j = 1; j < si; j++) off += (rad[j] - 1) * S(j);
		return off;
	}
};

__host__ __device__ constexpr uint32_t mix_slot(uint32_t a) { return a + (a >> 4); }

} // namespace vkfft_mi355x
