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

// LDS slot of FFT element a for one exchange: a + (a >> SH) (SH = 0: no padding).  Which padding is conflict-free depends on the
// radix that writes and the run length of the Stockham scatter (t-s)*R + s + k*S: stride 16 needs a + a/16, strides 5, 10, 13
// are conflict-free unpadded and BROKEN by it, runs of 13 need it again...  MixPad picks the shift per exchange at compile time
// by counting bank conflicts of the actual access pattern (rocprofv3: SQ_LDS_BANK_CONFLICT was 1.2-2.8x SQ_ACTIVE_INST_LDS with
// the fixed a + a/16 of the first version of these kernels).
template <int SH> __host__ __device__ constexpr uint32_t mix_slot(uint32_t a) { return SH > 0 ? a + (a >> SH) : a; }

// ES = bytes per complex element (8 / 16): the LDS serves 128 bytes per cycle = 16 / 8 lanes over 32 / 16 element-wide banks
template <typename SCH, int TPF, int ES> struct MixPad {
	static constexpr int G = 128 / ES, BANKS = 256 / ES;
	__host__ __device__ static constexpr uint32_t slot(int sh, uint32_t a) { return sh > 0 ? a + (a >> sh) : a; }
	// extra cycles of one G-lane group whose lane l touches element at(l) (lanes l >= nl idle)
	template <typename F> __host__ __device__ static constexpr int group_cost(int sh, int nl, F at) {
		int cnt[32] = {};
		int worst = 0;
		for (int l = 0; l < nl; l++) { const int bnk = (int)(slot(sh, at(l)) % (uint32_t)BANKS); cnt[bnk]++; if (cnt[bnk] > worst) worst = cnt[bnk]; }
		return worst > 1 ? worst - 1 : 0;
	}
	// exchange e = written by stage e, read by stage e + 1
	__host__ __device__ static constexpr int cost(int e, int sh) {
		const int N = SCH::N, R = SCH::rad[e], NB = N / R, S = SCH::S(e), R2 = SCH::rad[e + 1], NB2 = N / R2;
		int total = 0;
		const int lanes = TPF < NB ? TPF : NB;
		for (int g0 = 0; g0 < lanes && g0 < 4 * G; g0 += G) {
			const int nl = lanes - g0 < G ? lanes - g0 : G;
			for (int k = 0; k < R; k++)
				total += group_cost(sh, nl, [=](int l) { const uint32_t t = (uint32_t)(g0 + l), s2 = t % (uint32_t)S; return (t - s2) * (uint32_t)R + s2 + (uint32_t)(k * S); });
		}
		const int lanes2 = TPF < NB2 ? TPF : NB2;
		for (int g0 = 0; g0 < lanes2 && g0 < 4 * G; g0 += G) {
			const int nl = lanes2 - g0 < G ? lanes2 - g0 : G;
			for (int i = 0; i < R2; i++) total += group_cost(sh, nl, [=](int l) { return (uint32_t)(g0 + l + i * NB2); });
		}
		return total;
	}
	struct Table { int sh[5]; int minShift; };
	__host__ __device__ static constexpr Table make() { // evaluated once per kernel instance
		Table t = {{0, 0, 0, 0, 0}, 0};
		for (int e = 0; e + 1 < SCH::NS; e++) {
			int best = 0, bestCost = cost(e, 0);
			for (int sh = 5; sh >= 3; sh--) { const int c = cost(e, sh); if (c < bestCost) { bestCost = c; best = sh; } }
			t.sh[e] = best;
			if (best > 0 && (t.minShift == 0 || best < t.minShift)) t.minShift = best; // the smallest shift sizes the buffer
		}
		return t;
	}
	static constexpr Table tab = make();
	__host__ __device__ static constexpr int shift(int e) { return (e < 0 || e + 1 >= SCH::NS) ? 0 : tab.sh[e]; }
	__host__ __device__ static constexpr int min_shift() { return tab.minShift; }
	__host__ __device__ static constexpr int elems() { return (int)slot(min_shift(), (uint32_t)SCH::N) + 1; } // LDS elements per FFT
};

} // namespace vkfft_mi355x
