// Planning side of kernel_mixrad.h (shared with the planner, no device code): which cofactors the Rader-stage kernel serves and how many rows of
// N = M * P points a workgroup of the prime's instance takes.
#pragma once
#include "common.h"

namespace vkfft_mi355x {

// cofactors served (one butterfly each inside every instance: cofactors with a prime factor of 11 or more are left to Bluestein — 80 KB of code per instance as it is)
__host__ __device__ constexpr bool mixrad_cofactor_ok(uint32_t m) {
	return (m >= 2 && m <= 10) || m == 12 || m == 14 || m == 15 || m == 16 || m == 18 || m == 20 || m == 21 || m == 24 || m == 25 || m == 27 || m == 28 || m == 30 || m == 32;
}


// LDS elements of the tile's rows: one round of the thread groups at full occupation (FPW sub-sequences) or the longest row the prime serves, whichever is
// more; the large-cofactor instances (MHI: 151 VGPRs, two workgroups per CU whatever the LDS) take twice that, so that two or three long rows share a
// workgroup (2020 = 20 * 101 alone leaves 12 of the 32 thread groups and two thirds of the threads of the column step idle: 1.1 TB/s)
__host__ __device__ constexpr uint32_t mixrad_row_elems(uint32_t P, uint32_t FPW, bool dp, bool hi) {
	const uint32_t cap = (dp ? 2048u : 4096u) * (hi ? 2u : 1u), longest = 32u * (hi ? 2u : 1u) * P < cap ? 32u * (hi ? 2u : 1u) * P : cap;
	return FPW * P > longest ? FPW * P : longest;
}
// ops: a real transform between the generic maps (OPS form): the result leaves through a second row region, so half the capacity per region
__host__ __device__ constexpr uint32_t mixrad_rows(uint32_t P, uint32_t FPW, bool dp, uint32_t N, uint32_t M, bool ops) {
	return (mixrad_row_elems(P, FPW, dp, M > 10) >> (ops ? 1 : 0)) / N > 0 ? (mixrad_row_elems(P, FPW, dp, M > 10) >> (ops ? 1 : 0)) / N : 1u;
}
__host__ __device__ constexpr bool mixrad_fits(uint32_t P, uint32_t FPW, bool dp, uint32_t N, uint32_t M, bool ops) { return (mixrad_row_elems(P, FPW, dp, M > 10) >> (ops ? 1 : 0)) >= N; }

} // namespace vkfft_mi355x
