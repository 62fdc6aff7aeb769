// Instantiations and registry of the fused Four-Step kernels for non-power-of-two two-factor lengths (kernel_mix_fused.h): own translation unit (build time).
#include "engine.h"
#include "kernel_mix_fused.h"
#include <cstdlib>
#include <cstdio>
#include <algorithm>
#include <atomic>

namespace vkfft_mi355x {

constexpr int mixf_min(int a, int b) { return a < b ? a : b; }
// (a0..a3: radices of the first factor n0 — the strided columns of the input —, threads per transform, columns per tile; the same for the second factor; cap on the workgroups per CU)
#define VKFFT_MXFB(T, dp, a0, a1, a2, a3, tpfa, tca, b0, b1, b2, b3, tpfb, tcb, cap, blue) \
	{ (uint64_t)((a0) * (a1) * (a2) * (a3)) * (uint64_t)((b0) * (b1) * (b2) * (b3)), (a0) * (a1) * (a2) * (a3), (b0) * (b1) * (b2) * (b3), dp, {a0, a1, a2, a3, 1}, {b0, b1, b2, b3, 1}, tpfa, tca, tpfb, tcb, (tpfa) * (tca), \
	  mixf_min(cap, mixf_wg_per_cu<T, MixSched<a0, a1, a2, a3, 1>, tpfa, tca, MixSched<b0, b1, b2, b3, 1>, tcb>()), blue, \
	  &mix_fused_launch<T, MixSched<a0, a1, a2, a3, 1>, tpfa, tca, MixSched<b0, b1, b2, b3, 1>, tpfb, tcb, 2, blue>, \
	  (const void*)&mix_fused_kernel<T, MixSched<a0, a1, a2, a3, 1>, tpfa, tca, MixSched<b0, b1, b2, b3, 1>, tpfb, tcb, 2, blue> }
#define VKFFT_MXF(T, dp, a0, a1, a2, a3, tpfa, tca, b0, b1, b2, b3, tpfb, tcb, cap) VKFFT_MXFB(T, dp, a0, a1, a2, a3, tpfa, tca, b0, b1, b2, b3, tpfb, tcb, cap, 0)
// padded lengths of the two-launch chirp-z plan (kernel_mix_fused.h MixFusedOps): the instance with the hooks
#define VKFFT_MXB(a0, a1, a2, a3, tpfa, tca, b0, b1, b2, b3, tpfb, tcb, cap) VKFFT_MXFB(float, false, a0, a1, a2, a3, tpfa, tca, b0, b1, b2, b3, tpfb, tcb, cap, 1)

// first entry of a length is the default; VKFFT_MI355X_MXFV=k selects the k-th shape of a length (tuning)
static const MixFusedVariant kMixFusedVariants[] = {
	// powers of three (BASELINE config 3: 3^10 ... 3^15)
	VKFFT_MXF(float, false, 9, 9, 3, 1, 27, 16, 9, 9, 3, 1, 27, 16, 4),   // 3^10 = 243 x 243
	VKFFT_MXF(float, false, 9, 9, 3, 1, 27, 24, 9, 9, 9, 1, 81, 8, 4),    // 3^11 = 243 x 729
	VKFFT_MXF(float, false, 9, 9, 9, 1, 81, 8, 9, 9, 9, 1, 81, 8, 4),     // 3^12 = 729 x 729
	VKFFT_MXF(float, false, 9, 9, 9, 1, 27, 16, 9, 9, 9, 1, 27, 16, 4),   // ... 16-column tiles (128-byte segments), one workgroup of 432 threads per CU
	// powers of five (5^6 ... 5^9)
	VKFFT_MXF(float, false, 5, 5, 5, 1, 25, 16, 5, 5, 5, 1, 25, 16, 4),   // 5^6 = 125 x 125
	// (625 points as four radix-5 stages: with two radix-25 stages the instance takes 206-256 registers and spills — the table look-ups of 24 twiddles in flight beside a 25-point butterfly)
	VKFFT_MXF(float, false, 5, 5, 5, 1, 25, 16, 5, 5, 5, 5, 25, 16, 4),   // 5^7 = 125 x 625
	VKFFT_MXF(float, false, 5, 5, 5, 5, 25, 16, 5, 5, 5, 5, 25, 16, 4),   // 5^8 = 625 x 625
	// powers of seven (7^6, 7^7)
	VKFFT_MXF(float, false, 7, 7, 7, 1, 49, 16, 7, 7, 7, 1, 49, 16, 4),   // 7^6 = 343 x 343
	VKFFT_MXF(float, false, 7, 7, 7, 1, 49, 16, 7, 7, 7, 7, 98, 8, 4),    // 7^7 = 343 x 2401
	// powers of eleven and thirteen (11^4 ... 11^5, 13^4 ... 13^5)
	VKFFT_MXF(float, false, 11, 11, 1, 1, 11, 32, 11, 11, 1, 1, 11, 32, 4), // 11^4 = 121 x 121
	VKFFT_MXF(float, false, 11, 11, 1, 1, 11, 88, 11, 11, 11, 1, 121, 8, 4), // 11^5 = 121 x 1331
	VKFFT_MXF(float, false, 13, 13, 1, 1, 13, 32, 13, 13, 1, 1, 13, 32, 4), // 13^4 = 169 x 169
	VKFFT_MXF(float, false, 13, 13, 1, 1, 10, 68, 13, 13, 13, 1, 85, 8, 4), // 13^5 = 169 x 2197
	// ---- padded lengths M >= 2N - 1 of the chirp-z plan, ascending: the planner takes the smallest that fits.  Powers of two from 2^15 to 2^22 (any prime up to 2^21 has
	// one), and the 7-smooth lengths right above 2N - 1 of the primes BASELINE config 3 names (15319 -> 30720, 21269 -> 43008, 524309 -> 1049760, 2000083 -> 4014080)
	VKFFT_MXB(10, 16, 1, 1, 16, 16, 12, 16, 1, 1, 16, 16, 4),   // 30720 = 160 x 192
	VKFFT_MXB(8, 16, 1, 1, 16, 16, 16, 16, 1, 1, 16, 16, 4),    // 2^15 = 128 x 256
	VKFFT_MXB(12, 16, 1, 1, 16, 16, 14, 16, 1, 1, 16, 16, 4),   // 43008 = 192 x 224
	VKFFT_MXB(16, 16, 1, 1, 16, 16, 16, 16, 1, 1, 16, 16, 4),   // 2^16 = 256 x 256
	VKFFT_MXB(16, 16, 1, 1, 16, 32, 8, 8, 8, 1, 32, 16, 4),     // 2^17 = 256 x 512
	VKFFT_MXB(8, 8, 8, 1, 32, 16, 8, 8, 8, 1, 32, 16, 4),       // 2^18 = 512 x 512
	VKFFT_MXB(8, 8, 8, 1, 32, 16, 16, 8, 8, 1, 64, 8, 4),       // 2^19 = 512 x 1024
	VKFFT_MXB(16, 8, 8, 1, 64, 8, 16, 8, 8, 1, 64, 8, 4),       // 2^20 = 1024 x 1024
	VKFFT_MXB(12, 9, 9, 1, 54, 8, 12, 10, 9, 1, 54, 8, 4),      // 1049760 = 972 x 1080
	VKFFT_MXB(16, 8, 8, 1, 128, 8, 16, 16, 8, 1, 128, 8, 4),    // 2^21 = 1024 x 2048
	VKFFT_MXB(8, 5, 7, 7, 98, 8, 16, 16, 8, 1, 98, 8, 4),       // 4014080 = 1960 x 2048
	VKFFT_MXB(16, 16, 8, 1, 128, 8, 16, 16, 8, 1, 128, 8, 4),   // 2^22 = 2048 x 2048
};
// mix_fused_lookup(n | kMixFusedBlueQuery, ...) asks for the smallest chirp-z instance of n points or more (its length = *n0 * *n1)
constexpr uint64_t kMixFusedBlueQuery = 1ull << 63;
constexpr int kNumMixFusedVariants = (int)(sizeof(kMixFusedVariants) / sizeof(kMixFusedVariants[0]));

bool mix_fused_lookup(uint64_t n, bool dp, int* variant, int* n0, int* n1, int radA[5], int radB[5], int* tca, int* tcb, int* threads, int* wgPerCu) {
	int want = 0;
	if (const char* e = getenv("VKFFT_MI355X_MXFV")) want = atoi(e);
	int seen = 0, found = -1;
	if (n & kMixFusedBlueQuery) {
		const uint64_t minLen = n & ~kMixFusedBlueQuery;
		for (int i = 0; i < kNumMixFusedVariants; i++) {
			const MixFusedVariant& v = kMixFusedVariants[i];
			if (!v.blue || v.dp != dp || v.n < minLen) continue;
			if (found < 0 || v.n < kMixFusedVariants[found].n) found = i;
		}
	} else
	for (int i = 0; i < kNumMixFusedVariants; i++) {
		const MixFusedVariant& v = kMixFusedVariants[i];
		if (v.n != n || v.dp != dp || v.blue) continue;
		if (found < 0) found = i;
		if (seen == want) { found = i; break; }
		seen++;
	}
	if (found < 0) return false;
	const MixFusedVariant& v = kMixFusedVariants[found];
	*variant = found; *n0 = v.n0; *n1 = v.n1; *tca = v.tca; *tcb = v.tcb; *threads = v.threads; *wgPerCu = v.wgPerCu;
	for (int k = 0; k < 5; k++) { radA[k] = v.radA[k]; radB[k] = v.radB[k]; }
	return true;
}

int launch_mix_fused(const PassPlan& pp, const FusedParams& prm, hipStream_t stream) {
	if (pp.variant < 0 || pp.variant >= kNumMixFusedVariants) return 4039;
	const MixFusedVariant& v = kMixFusedVariants[pp.variant];
	// persistent grid: what the chip holds at once (the ticket queue needs no co-residency: any grid is correct)
	constexpr int kMaxDev = 32;
	static std::atomic<int> occ[kMaxDev][kNumMixFusedVariants];
	int dev = 0, n = 0;
#if !defined(VKFFT_HOSTEMU)
	if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
#endif
	const bool cached = dev < kMaxDev;
	if (cached) n = occ[dev][pp.variant].load(std::memory_order_relaxed);
	if (!n) {
#if defined(VKFFT_HOSTEMU)
		n = 1;
#else
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, v.fn, v.threads, 0) != hipSuccess || n < 1) n = 1;
#endif
		if (cached) occ[dev][pp.variant].store(n, std::memory_order_relaxed);
	}
	const uint64_t tickets = (uint64_t)(prm.C + prm.D * prm.Q) * prm.tpc;
	uint64_t grid = (uint64_t)pow2_num_cus() * (pp.fusedWgPerCu > 0 ? (uint32_t)pp.fusedWgPerCu : (uint32_t)std::min(n, v.wgPerCu));
	if (grid > tickets) grid = tickets;
	if (grid == 0) return 0;
	MixFusedOps ops = {};
	if (v.blue) { // chirp-z hooks: the planner left them in the pass descriptor's unused fields (emit_mix_fused_blue); the arena is where the stage twiddles are
		const char* const ar = (const char*)prm.lutA - pp.lutOff;
		ops.chirp = pp.aux3Off != (size_t)-1 ? ar + pp.aux3Off : nullptr;
		ops.bhat = pp.aux2Off != (size_t)-1 ? ar + pp.aux2Off : nullptr;
		ops.blueN = pp.prm.opN;
		ops.preBlue = pp.prm.preOp == OP_BLUESTEIN_PRE; ops.postMul = pp.prm.postOp == OP_MUL_LUT; ops.postBlue = pp.prm.postOp == OP_BLUESTEIN_POST;
		ops.bsSwapIn = pp.prm.bluesteinSwapIn; ops.bsSwapOut = pp.prm.bluesteinSwapOut;
	}
	v.launch(prm, ops, dim3((uint32_t)grid), stream);
	return hipGetLastError() == hipSuccess ? 0 : 4039;
}

} // namespace vkfft_mi355x
