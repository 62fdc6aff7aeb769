// Planning side of kernel_mixrad.h (shared with the planner, no device code): which cofactors the Rader-stage kernels serve, which of the two kernels
// takes a row of N = M * P points and how many rows a workgroup of the prime's instance takes.
#pragma once
#include "common.h"

namespace vkfft_mi355x {

// cofactors served (one butterfly each inside an instance: cofactors with a prime factor of 11 or more are left to Bluestein)
__host__ __device__ constexpr bool mixrad_cofactor_ok(uint32_t m) {
	return (m >= 2 && m <= 10) || m == 12 || m == 14 || m == 15 || m == 16 || m == 18 || m == 20 || m == 21 || m == 24 || m == 25 || m == 27 || m == 28 || m == 30 || m == 32;
}
constexpr uint32_t kMixradLongest = 4096; // longest row (fp32 only)
// 1 = mixrad_small_kernel: complex rows, cofactor <= 10 and <= the thread groups of the prime's instance (every group owns one sub-sequence: the LDS of the
//     prime's own Rader kernel, its occupancy);  2 = mixrad_kernel: any served cofactor, real transforms between the generic maps — a tile of whole rows in
//     LDS next to the groups' buffers, sub-sequences in rounds; instantiated for the primes that leave room for a cofactor of 12 (12 P <= 4096);  0 = neither
__host__ __device__ constexpr int mixrad_mode(uint32_t P, uint32_t FPW, uint32_t M, bool ops) {
	if (!mixrad_cofactor_ok(M) || M * P > kMixradLongest) return 0;
	if (!ops && M <= 10 && M <= FPW) return 1;
	return 12 * P <= kMixradLongest ? 2 : 0;
}
// LDS elements of the big kernel's row region: two rounds of the thread groups at full occupation or twice the longest row the prime serves, whichever is more
// (2020 = 20 * 101 alone in a workgroup leaves 12 of the 32 thread groups and two thirds of the threads of the column step idle: 1.1 TB/s)
__host__ __device__ constexpr uint32_t mixrad_row_elems(uint32_t P, uint32_t FPW) {
	const uint32_t cap = 2 * kMixradLongest, longest = 64u * P < cap ? 64u * P : cap;
	return FPW * P > longest ? FPW * P : longest;
}
// rows per workgroup.  ops: the result leaves through a second row region, so half the capacity per region
__host__ __device__ constexpr uint32_t mixrad_rows(int mode, uint32_t P, uint32_t FPW, uint32_t M, bool ops) {
	if (mode == 1) return FPW / M;
	const uint32_t cap = mixrad_row_elems(P, FPW) >> (ops ? 1 : 0);
	return cap / (M * P) > 0 ? cap / (M * P) : 1u;
}
__host__ __device__ constexpr bool mixrad_fits(int mode, uint32_t P, uint32_t FPW, uint32_t M, bool ops) { return mode == 1 || (mode == 2 && (mixrad_row_elems(P, FPW) >> (ops ? 1 : 0)) >= M * P); }

} // namespace vkfft_mi355x
