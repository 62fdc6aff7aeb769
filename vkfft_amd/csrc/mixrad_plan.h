// Planning side of kernel_mixrad.h (shared with the planner, no device code): which cofactors the Rader-stage kernel serves, how a cofactor splits into the
// two column steps, how many rows a workgroup takes and how much LDS that needs.
#pragma once
#include "common.h"

namespace vkfft_mi355x {

constexpr uint32_t kMixradLongest = 4096; // longest row, complex points (fp32; fp64: half of it)
// Radices of the column steps.  In registers (one butterfly per thread): 2 ... 10 and 12 — the kernel's register count is that of its largest butterfly, and with these
// it stays at the level of the convolution stages (64-80 VGPRs; measured on the 37-point instance: radix 13 94, 14 102, 15 126, 17 130, 23 159, 31 193 — round 5's
// single-step butterflies up to 32 put the tiled kernel at 141-164 VGPRs, one to three wavefronts per SIMD).  Any ODD radix up to 55 as the direct sum with
// re-read inputs (kernel_mixrad.h mixrad_col_direct): 11, 13, the primes 17 ... 31 (... 53), and 15, 25, 35, 49 where a cofactor has no other split — always the
// last step (its outputs must not land on its inputs: they leave to memory, or, for a real transform, to a second set of buffers).
__host__ __device__ constexpr bool mixrad_radix_reg(uint32_t r) { return (r >= 2 && r <= 10) || r == 12; }
__host__ __device__ constexpr bool mixrad_radix_direct(uint32_t r) { return (r & 1u) && r >= 11 && r <= 55; }
// cofactor M = A * B: A = radix of the first column step (1: there is none), B = radix of the second.  Preference: one register butterfly; two of them, the most
// balanced pair; a register butterfly and the smallest direct radix.  false: no such split (a prime factor above 53)
__host__ __device__ constexpr bool mixrad_split(uint32_t M, uint32_t& A, uint32_t& B) {
	if (mixrad_radix_reg(M)) { A = 1; B = M; return true; }
	uint32_t bestA = 0, bestB = 0;
	for (uint32_t a = 2; a <= 12 && a * a <= M; a++) {
		if (M % a) continue;
		if (mixrad_radix_reg(a) && mixrad_radix_reg(M / a)) { bestA = a; bestB = M / a; } // (the last hit has the largest a <= sqrt M: the most balanced)
	}
	if (bestA) { A = bestA; B = bestB; return true; }
	for (uint32_t a = 12; a >= 1; a--) { // the largest register radix leaves the smallest direct one
		if (M % a || !(a == 1 || mixrad_radix_reg(a))) continue;
		if (mixrad_radix_direct(M / a)) { A = a; B = M / a; return true; }
	}
	return false;
}

// LDS of a workgroup (elements of cx<T>): R * M sub-sequence buffers of SP elements followed by
// the tables: stage twiddles of the prime's convolution (lutN), its kernel spectrum (P - 1), the two-level column twiddle (64 + ceil(N / 64)), the roots of the
// cofactor (M), and the two generator permutations as 16-bit indices
__host__ __device__ constexpr uint32_t mixrad_table_elems(uint32_t P, uint32_t M, uint32_t lutN) { return lutN + (P - 1u) + 64u + (M * P + 63u) / 64u + M; }
// (two sets of buffers: a real transform whose last column step is a direct sum — that step may not write over its inputs, and the post-map reads LDS)
__host__ __device__ constexpr bool mixrad_two_sets(uint32_t M, uint32_t A) { return M > 1u && A != 0u && !mixrad_radix_reg(M / A); }
__host__ __device__ constexpr uint64_t mixrad_lds_bytes(uint32_t P, uint32_t SP, uint32_t lutN, uint32_t M, uint32_t R, uint32_t elemBytes, bool twoSets) {
	return ((uint64_t)(R * M) * SP * (twoSets ? 2u : 1u) + mixrad_table_elems(P, M, lutN)) * elemBytes + ((2u * (P - 1u) * 2u + 15u) & ~15u);
}
// rows per workgroup: the smallest count that keeps 90 % of the thread groups busy in the convolution rounds (jobs = R * M sub-sequences over FPW groups), else the
// best one, within the LDS budget (several workgroups per CU: the kernel lives on overlapping the two memory trips of one workgroup with the LDS phases of the others)
__host__ __device__ constexpr uint32_t mixrad_rows(uint32_t P, uint32_t SP, uint32_t lutN, uint32_t FPW, uint32_t M, uint32_t elemBytes, uint64_t budgetBytes, bool twoSets) {
	uint32_t best = 1; uint64_t bestU = 0;
	for (uint32_t R = 1; R <= 64u; R++) {
		if (R > 1u && mixrad_lds_bytes(P, SP, lutN, M, R, elemBytes, twoSets) > budgetBytes) break;
		const uint64_t jobs = (uint64_t)R * M, rounds = (jobs + FPW - 1u) / FPW;
		const uint64_t u = jobs * 1000u / (rounds * FPW);
		if (u > bestU) { bestU = u; best = R; }
		if (u >= 900u) break;
	}
	return best;
}

} // namespace vkfft_mi355x
