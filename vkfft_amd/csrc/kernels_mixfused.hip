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
#define VKFFT_MXFB(T, dp, a0, a1, a2, a3, tpfa, tca, b0, b1, b2, b3, tpfb, tcb, cap, blue) VKFFT_MXFM(T, dp, a0, a1, a2, a3, tpfa, tca, b0, b1, b2, b3, tpfb, tcb, cap, blue, 2, 4)
#define VKFFT_MXFM(T, dp, a0, a1, a2, a3, tpfa, tca, b0, b1, b2, b3, tpfb, tcb, cap, blue, mode, wgc) VKFFT_MXFP(T, dp, a0, a1, a2, a3, tpfa, tca, b0, b1, b2, b3, tpfb, tcb, cap, blue, mode, wgc, 1)
#define VKFFT_MXFP(T, dp, a0, a1, a2, a3, tpfa, tca, b0, b1, b2, b3, tpfb, tcb, cap, blue, mode, wgc, pipe) \
	{ (uint64_t)((a0) * (a1) * (a2) * (a3)) * (uint64_t)((b0) * (b1) * (b2) * (b3)), (a0) * (a1) * (a2) * (a3), (b0) * (b1) * (b2) * (b3), dp, {a0, a1, a2, a3, 1}, {b0, b1, b2, b3, 1}, tpfa, tca, tpfb, tcb, (tpfa) * (tca), \
	  mixf_min(cap, mixf_wg_per_cu<T, MixSched<a0, a1, a2, a3, 1>, tpfa, tca, MixSched<b0, b1, b2, b3, 1>, tcb>()), blue, \
	  &mix_fused_launch<T, MixSched<a0, a1, a2, a3, 1>, tpfa, tca, MixSched<b0, b1, b2, b3, 1>, tpfb, tcb, mode, blue, wgc, pipe>, \
	  (const void*)&mix_fused_kernel<T, MixSched<a0, a1, a2, a3, 1>, tpfa, tca, MixSched<b0, b1, b2, b3, 1>, tpfb, tcb, mode, blue, wgc, pipe> }
#define VKFFT_MXF(T, dp, a0, a1, a2, a3, tpfa, tca, b0, b1, b2, b3, tpfb, tcb, cap) VKFFT_MXFB(T, dp, a0, a1, a2, a3, tpfa, tca, b0, b1, b2, b3, tpfb, tcb, cap, 0)
// padded lengths of the two-launch chirp-z plan (kernel_mix_fused.h MixFusedOps): the instance with the hooks
#define VKFFT_MXB(a0, a1, a2, a3, tpfa, tca, b0, b1, b2, b3, tpfb, tcb, cap) VKFFT_MXFB(float, false, a0, a1, a2, a3, tpfa, tca, b0, b1, b2, b3, tpfb, tcb, cap, 1)

// first entry of a length is the default; VKFFT_MI355X_MXFV=k selects the k-th shape of a length (tuning)
static const MixFusedVariant kMixFusedVariants[] = {
	// Shapes: ONE butterfly per thread and stage wherever the tile width allows it (threads per column = points / smallest radix) — a thread that owns P butterflies
	// keeps P x radix points live across the exchange barrier, and the instances of the first version with P = 4 ... 7 took 190-256 registers (one or two wavefronts per
	// SIMD, up to 1 KiB of scratch) where the same factor with P = 1 takes 61-112.
	// Shapes from the device sweeps (profiles/r06_mix_fused_*): the smaller factor second (its B tile — the segments of the stores to HBM and of the ring loads — is
	// then 24 ... 88 columns wide where the larger factor's would be 8); 32-column tiles where both factors are short; never two butterflies per thread and stage
	// (the shapes that doubled the B tile that way lost 20-40 %).
	// MODE 10: non-temporal loads, PLAIN stores on the HBM side (against non-temporal stores: 5^6 + 4 %, 5^8 + 5 %, 7^6 + 3.5 %, 11^6 + 12 %, the rest within 1 %)
	// last argument: workgroups per CU the launch and the planner's lag reckon with — 1 where the registers hold one workgroup per CU anyway (650 threads x 110-128
	// registers): with 3 the ring's lag was sized for workgroups that never become resident, and 3^12 lost 10 % to the longer fill and drain of its queues
#define VKFFT_MXF2(a0, a1, a2, a3, tpfa, tca, b0, b1, b2, b3, tpfb, tcb, cap) VKFFT_MXFM(float, false, a0, a1, a2, a3, tpfa, tca, b0, b1, b2, b3, tpfb, tcb, cap, 0, 10, 4)
	// powers of three (BASELINE config 3: 3^10 ... 3^12)
	VKFFT_MXF2(9, 9, 3, 1, 27, 32, 9, 9, 3, 1, 27, 32, 4),     // 3^10 = 243 x 243
	VKFFT_MXFP(float, false, 9, 9, 9, 1, 81, 8, 9, 9, 3, 1, 27, 24, 2, 0, 10, 2, 0), // 3^11 = 729 x 243 — one tile at a time, two workgroups per CU (+ 6 / + 13 % over the pipelined form)
	VKFFT_MXF2(9, 9, 9, 1, 81, 8, 9, 9, 9, 1, 81, 8, 1),       // 3^12 = 729 x 729
	// powers of five (5^7, 5^8; 5^6 = 15625, 7^5 and 11^4 run as ONE pass of the long mixed-radix rows, mixed_table_6.inc: 2.9-3.3 against 1.9-2.4 TB/s fused)
	VKFFT_MXFP(float, false, 5, 5, 5, 5, 125, 8, 5, 5, 5, 1, 25, 40, 2, 0, 10, 2, 0), // 5^7 = 625 x 125 — one tile at a time, two workgroups per CU (+ 6 / + 13 % over the pipelined form)
	VKFFT_MXF2(5, 5, 5, 5, 125, 8, 5, 5, 5, 5, 125, 8, 4),     // 5^8 = 625 x 625
	// powers of seven (7^6)
	VKFFT_MXF2(7, 7, 7, 1, 49, 16, 7, 7, 7, 1, 49, 16, 4),     // 7^6 = 343 x 343
	// powers of eleven and thirteen (11^4 ... 11^6, 13^4)
	VKFFT_MXF2(11, 11, 11, 1, 121, 8, 11, 11, 1, 1, 11, 88, 4), // 11^5 = 1331 x 121
	VKFFT_MXF2(11, 11, 11, 1, 121, 8, 11, 11, 11, 1, 121, 8, 4), // 11^6 = 1331 x 1331
	VKFFT_MXF2(13, 13, 1, 1, 13, 32, 13, 13, 1, 1, 13, 32, 4), // 13^4 = 169 x 169
	// (7^7 = 343 x 2401, 13^5 = 169 x 2197: a 2401- or 2197-point column needs four or two butterflies per thread at 8 columns per tile — 250-330 bytes of scratch,
	// 0.81 TB/s against 1.22 / 1.42 of the separate passes on the device: not instantiated)
	// ---- padded lengths M >= 2N - 1 of the chirp-z plan, ascending: the planner takes the smallest that fits.  Powers of two from 2^15 to 2^20 and the 7-smooth
	// lengths right above 2N - 1 of the primes BASELINE config 3 names (15319 -> 30720, 21269 -> 43008, 524309 -> 1049760); radices up to 8 (12 in the last)
	VKFFT_MXB(8, 5, 4, 1, 32, 16, 8, 8, 3, 1, 32, 16, 4),     // 30720 = 160 x 192
	VKFFT_MXB(8, 8, 3, 1, 32, 16, 8, 7, 4, 1, 32, 16, 4),     // 43008 = 192 x 224
	VKFFT_MXB(8, 8, 4, 1, 32, 16, 8, 8, 4, 1, 32, 16, 4),     // 2^16 = 256 x 256
	VKFFT_MXB(8, 8, 4, 1, 32, 32, 8, 8, 8, 1, 64, 16, 4),     // 2^17 = 256 x 512
	VKFFT_MXB(8, 8, 8, 1, 64, 16, 8, 8, 8, 1, 64, 16, 4),     // 2^18 = 512 x 512
	VKFFT_MXB(8, 8, 8, 1, 64, 16, 8, 8, 4, 4, 128, 8, 4),     // 2^19 = 512 x 1024
	VKFFT_MXB(8, 8, 4, 4, 128, 8, 8, 8, 4, 4, 128, 8, 4),     // 2^20 = 1024 x 1024
	VKFFT_MXB(12, 9, 9, 1, 108, 8, 12, 10, 9, 1, 108, 8, 4),  // 1049760 = 972 x 1080
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
