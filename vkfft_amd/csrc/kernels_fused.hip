// Instantiations and registry of the fused Four-Step kernels (kernel_pow2_fused.h): own translation unit (build time).
#include "engine.h"
#include "kernel_pow2_fused.h"
#include "kernel_pow2_fused_pipe.h"
#include "kernel_pow2_fused_pk.h"
#include "kernel_pow2_fused_pkh.h"
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <atomic>
#include <string>

namespace vkfft_mi355x {

constexpr int fused_min(int a, int b) { return a < b ? a : b; }
// cap: workgroups per CU at most (measured: more of them than the ring's lag tolerates only add dependency stalls)
#define VKFFT_FU0(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, mode, twl, cpt, lean) VKFFT_FUX(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, mode, twl, cpt, lean, 8)
#define VKFFT_FUX(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, mode, twl, cpt, lean, cap) \
	{ (a0) + (a1) + (a2) + (b0) + (b1) + (b2), dp, mode, (a0) + (a1) + (a2), (b0) + (b1) + (b2), {a0, a1, a2, 0}, {b0, b1, b2, 0}, tca, tcb, \
	  ((1 << ((a0) + (a1) + (a2))) >> Pow2Sched<a0, a1, a2, 0>::LOGE) * (tca) / (cpt), fused_min(cap, pow2_fused_wg_per_cu<T, Pow2Sched<a0, a1, a2, 0>, tca, Pow2Sched<b0, b1, b2, 0>, tcb, twl, cpt, lean>()), \
	  &pow2_fused_launch<T, Pow2Sched<a0, a1, a2, 0>, tca, Pow2Sched<b0, b1, b2, 0>, tcb, mode, twl, cpt, lean>, \
	  (const void*)&pow2_fused_kernel<T, Pow2Sched<a0, a1, a2, 0>, tca, Pow2Sched<b0, b1, b2, 0>, tcb, mode, twl, cpt, lean> }
#define VKFFT_FU1(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, mode, twl, cpt) VKFFT_FU0(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, mode, twl, cpt, 0)
// mode 2 = product (non-temporal hint on the streamed side); the development build (-DVKFFT_MI355X_DEV) adds mode 6 = the same with per-phase cycle sums
#if defined(VKFFT_MI355X_DEV)
#define VKFFT_FUC(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, twl, cpt) \
	VKFFT_FU1(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, 2, twl, cpt), VKFFT_FU1(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, 6, twl, cpt)
#else
#define VKFFT_FUC(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, twl, cpt) VKFFT_FU1(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, 2, twl, cpt)
#endif
// register-lean form (kernel_pow2_lean.h): two columns per thread, real / imaginary planes exchanged one after the other
#if defined(VKFFT_MI355X_DEV)
#define VKFFT_FULC(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, twl, cap) VKFFT_FUX(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, 2, twl, 2, 1, cap), VKFFT_FUX(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, 6, twl, 2, 1, cap)
#else
#define VKFFT_FULC(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, twl, cap) VKFFT_FUX(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, 2, twl, 2, 1, cap)
#endif
#define VKFFT_FUL(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, twl) VKFFT_FULC(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, twl, 8)
// software-pipelined form (kernel_pow2_fused_pipe.h): register-lean stages, the other tile's memory traffic in flight while one tile computes
#define VKFFT_FUP(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, twl, wgc) \
	{ (a0) + (a1) + (a2) + (b0) + (b1) + (b2), dp, 2, (a0) + (a1) + (a2), (b0) + (b1) + (b2), {a0, a1, a2, 0}, {b0, b1, b2, 0}, tca, tcb, \
	  ((1 << ((a0) + (a1) + (a2))) >> Pow2Sched<a0, a1, a2, 0>::LOGE) * (tca) / 2, pow2_fused_pipe_wg_per_cu<T, Pow2Sched<a0, a1, a2, 0>, tca, Pow2Sched<b0, b1, b2, 0>, tcb, twl, wgc>(), \
	  &pow2_fused_pipe_launch<T, Pow2Sched<a0, a1, a2, 0>, tca, Pow2Sched<b0, b1, b2, 0>, tcb, 2, twl, wgc>, \
	  (const void*)&pow2_fused_pipe_kernel<T, Pow2Sched<a0, a1, a2, 0>, tca, Pow2Sched<b0, b1, b2, 0>, tcb, 2, twl, wgc>, "pow2_fused_pipe_kernel" }
// the same on packed pairs (kernel_pow2_fused_pk.h, round 5): half the vector instructions, 150 registers
#if defined(VKFFT_MI355X_DEV)
#define VKFFT_FUK(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, twl, wgc) VKFFT_FUKM(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, twl, wgc, 2), VKFFT_FUKM(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, twl, wgc, 6)
#define VKFFT_FUH(splita, splitb, twl) VKFFT_FUHM(splita, splitb, twl, 2), VKFFT_FUHM(splita, splitb, twl, 6)
#else
#define VKFFT_FUK(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, twl, wgc) VKFFT_FUKM(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, twl, wgc, 2)
#define VKFFT_FUH(splita, splitb, twl) VKFFT_FUHM(splita, splitb, twl, 2)
#endif
#define VKFFT_FUKM(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, twl, wgc, mode) \
	{ (a0) + (a1) + (a2) + (b0) + (b1) + (b2), dp, mode, (a0) + (a1) + (a2), (b0) + (b1) + (b2), {a0, a1, a2, 0}, {b0, b1, b2, 0}, tca, tcb, \
	  ((1 << ((a0) + (a1) + (a2))) >> Pow2Sched<a0, a1, a2, 0>::LOGE) * (tca) / 2, pow2_fused_pk_wg_per_cu<T, Pow2Sched<a0, a1, a2, 0>, tca, Pow2Sched<b0, b1, b2, 0>, tcb, twl, wgc>(), \
	  &pow2_fused_pk_launch<T, Pow2Sched<a0, a1, a2, 0>, tca, Pow2Sched<b0, b1, b2, 0>, tcb, mode, twl, wgc>, \
	  (const void*)&pow2_fused_pk_kernel<T, Pow2Sched<a0, a1, a2, 0>, tca, Pow2Sched<b0, b1, b2, 0>, tcb, mode, twl, wgc>, "pow2_fused_pk_kernel" }
// tiles of two halves on packed pairs (kernel_pow2_fused_pkh.h, round 5): 2^21 / 2^22, a 2048-point factor as two interleaved 1024-point halves + one radix-2 layer
#define VKFFT_FUHM(splita, splitb, twl, mode) VKFFT_FUHL(4, 3, 3, splita, splitb, twl, mode)
// h0 + h1 + h2 = log2 of a half's length (10: 512 threads, one workgroup per CU; 9: 256 threads, two)
#define VKFFT_FUHL(h0, h1, h2, splita, splitb, twl, mode) \
	{ 2 * ((h0) + (h1) + (h2)) + (splita) + (splitb), false, mode, (h0) + (h1) + (h2) + (splita), (h0) + (h1) + (h2) + (splitb), {h0, h1, h2, splita}, {h0, h1, h2, splitb}, (splita) ? 16 : 32, (splitb) ? 16 : 32, \
	  ((1 << ((h0) + (h1) + (h2))) >> Pow2Sched<h0, h1, h2, 0>::LOGE) * 8, (h0) + (h1) + (h2) >= 10 ? 1 : 2, \
	  &pow2_fused_pkh_launch<float, Pow2Sched<h0, h1, h2, 0>, splita, splitb, mode, twl>, (const void*)&pow2_fused_pkh_kernel<float, Pow2Sched<h0, h1, h2, 0>, splita, splitb, mode, twl>, "pow2_fused_pkh_kernel" }
#define VKFFT_FUT(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, twl) VKFFT_FUC(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, twl, 1)
#define VKFFT_FU(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb) VKFFT_FUC(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, 1, 1)
#define VKFFT_FU2(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb) VKFFT_FUC(T, dp, a0, a1, a2, tca, b0, b1, b2, tcb, 1, 2) /* two columns per thread */

// first entry of each (log2 N, dp, mode) is the default; VKFFT_MI355X_FUV<log2n>=k selects the k-th shape (tuning)
static const Pow2FusedVariant kPow2FusedVariants[] = {
	// fp32.  First entry of a size = what ships: the packed-pair software-pipelined form (kernel_pow2_fused_pk.h, round 5); second entry = the round-4 pipelined form it
	// was measured against (VKFFT_MI355X_FUV<k>=1).  The round-2 / round-3 shapes (two columns per thread without pipelining, the plane-split form at two workgroups per
	// CU, 8-column 2048-point tiles) lost every comparison since and are no longer instantiated.
	// 2^15 = 128 x 256 (only with VKFFT_MI355X_ROW15=0: 2^15 ships as ONE pass of the packed row kernel, 4.0-4.3 against 3.5 TB/s)
	VKFFT_FUK(float, false, 4, 3, 0, 32, 4, 4, 0, 16, 1, 4),
	VKFFT_FUP(float, false, 4, 3, 0, 32, 4, 4, 0, 16, 1, 4),
	// 2^16 = 256 x 256
	VKFFT_FUK(float, false, 4, 4, 0, 32, 4, 4, 0, 32, 1, 2),
	VKFFT_FUP(float, false, 4, 4, 0, 32, 4, 4, 0, 32, 1, 2),
	// 2^17 = 256 x 512
	VKFFT_FUK(float, false, 4, 4, 0, 32, 4, 3, 2, 16, 1, 2),
	VKFFT_FUP(float, false, 4, 4, 0, 32, 4, 3, 2, 16, 1, 2),
	// 2^18 = 512 x 512
	VKFFT_FUK(float, false, 4, 3, 2, 16, 4, 3, 2, 16, 1, 2),
	VKFFT_FUP(float, false, 4, 3, 2, 16, 4, 3, 2, 16, 1, 2),
	// 2^19 = 512 x 1024, 2^20 = 1024 x 1024: 128 KiB tiles, one workgroup per CU (512 threads x 256 registers: one tile computing + one tile in flight)
	VKFFT_FUK(float, false, 4, 3, 2, 32, 4, 3, 3, 16, 1, 1),
	VKFFT_FUP(float, false, 4, 3, 2, 32, 4, 3, 3, 16, 1, 1),
	// (2^19 = 512 x 1024 as tiles of two 512-point halves — VKFFT_FUHL(4, 3, 2, 0, 1, 1, 2): 256 threads, two workgroups per CU — measured 2.07-2.16 TB/s against 3.09: not instantiated)
	// 2^20 = 1024 x 1024 as tiles of two 16-column halves (kernel_pow2_fused_pkh.h): 256-byte segments on both HBM sides, three half-tiles in registers: 3.15 against 3.07 TB/s
	// for the 16-column packed kernel (index 1), 2.90 for the round-4 pipelined form (index 2)
	VKFFT_FUH(0, 0, 1),
	VKFFT_FUK(float, false, 4, 3, 3, 16, 4, 3, 3, 16, 1, 1),
	VKFFT_FUP(float, false, 4, 3, 3, 16, 4, 3, 3, 16, 1, 1),
	// 2^21 = 2048 x 1024 (the other orientation measured the same), 2^22 = 2048 x 2048: tiles of two halves, software-pipelined at half-tile granularity (round 5);
	// second entry = the round-4 shape (register-lean 2048-point tiles 16 columns wide, 1024 threads, no pipelining)
	VKFFT_FUH(1, 0, 1),
	VKFFT_FUL(float, false, 4, 3, 3, 32, 4, 4, 3, 16, 0),
	VKFFT_FUH(1, 1, 1),
	VKFFT_FUL(float, false, 4, 4, 3, 16, 4, 4, 3, 16, 0),
	// fp64: 16-byte elements, 8-16 columns = 128-256-byte segments.  2^14 = 128 x 128, 2^15 = 128 x 256, 2^16 = 256 x 256, 2^17 = 256 x 512
	VKFFT_FU(double, true, 4, 3, 0, 16, 4, 3, 0, 16),
	VKFFT_FU(double, true, 4, 3, 0, 16, 4, 4, 0, 8),
	VKFFT_FU(double, true, 4, 4, 0, 16, 4, 4, 0, 16),
	VKFFT_FUT(double, true, 4, 4, 0, 16, 4, 3, 2, 8, 0), // (stage twiddles through L2: the LDS copy would cost the second workgroup per CU)
	// 2^18 = 512 x 512, 2^19 = 512 x 1024, 2^20 = 1024 x 1024: 128 KiB tiles, one workgroup per CU (2^18 with 64 KiB tiles and two workgroups
	// per CU measured 2.07 TB/s against 2.80 — its ring of 4 MiB chunks does not fit the cache budget with the lag two workgroups per CU want)
	VKFFT_FUT(double, true, 4, 3, 2, 16, 4, 3, 2, 16, 0),
	VKFFT_FUT(double, true, 4, 3, 2, 16, 4, 3, 3, 8, 0),
	VKFFT_FUT(double, true, 4, 3, 3, 8, 4, 3, 3, 8, 0),
};
constexpr int kNumPow2FusedVariants = (int)(sizeof(kPow2FusedVariants) / sizeof(kPow2FusedVariants[0]));

bool pow2_fused_lookup(uint32_t log2n, bool dp, int mode, int* variant, int* la, int* lb, int bitsA[4], int bitsB[4], int* tca, int* tcb, int* threads, int* wgPerCu) {
	int want = 0;
	char name[64];
	snprintf(name, sizeof(name), "VKFFT_MI355X_FUV%u", log2n);
	if (const char* e = getenv(name)) want = atoi(e);
	int seen = 0, found = -1;
	for (int i = 0; i < kNumPow2FusedVariants; i++) {
		const Pow2FusedVariant& v = kPow2FusedVariants[i];
		if (v.log2n != (int)log2n || v.dp != dp || v.mode != mode) continue;
		if (found < 0) found = i;
		if (seen == want) { found = i; break; }
		seen++;
	}
	if (found < 0) return false;
	const Pow2FusedVariant& v = kPow2FusedVariants[found];
	*variant = found; *la = v.la; *lb = v.lb; *tca = v.tca; *tcb = v.tcb; *threads = v.threads; *wgPerCu = v.wgPerCu;
	for (int k = 0; k < 4; k++) { bitsA[k] = v.bitsA[k]; bitsB[k] = v.bitsB[k]; }
	return true;
}

const char* pow2_fused_kernel_name(int variant) {
	if (variant < 0 || variant >= kNumPow2FusedVariants || !kPow2FusedVariants[variant].name) return "pow2_fused_kernel";
	return kPow2FusedVariants[variant].name;
}

int launch_pow2_fused(const PassPlan& pp, const FusedParams& prm, hipStream_t stream) {
	if (pp.variant < 0 || pp.variant >= kNumPow2FusedVariants) return 4039;
	const Pow2FusedVariant& v = kPow2FusedVariants[pp.variant];
	// persistent grid: what the chip holds at once (the ticket queue needs no co-residency: any grid is correct)
	// (cached per device and variant; racing first calls compute the same value, so a relaxed atomic is enough)
	constexpr int kMaxDev = 32;
	static std::atomic<int> occ[kMaxDev][kNumPow2FusedVariants];
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
	const uint64_t tickets = (uint64_t)(prm.C + prm.D * prm.Q) << (prm.logG + prm.logTiles);
	uint64_t grid = (uint64_t)pow2_num_cus() * (pp.fusedWgPerCu > 0 ? (uint32_t)pp.fusedWgPerCu : (uint32_t)std::min(n, v.wgPerCu));
	if (grid > tickets) grid = tickets;
	if (grid == 0) return 0;
#if !defined(VKFFT_HOSTEMU)
	if ((v.mode & 4) && getenv("VKFFT_MI355X_FUSED_PROFILE")) { // development: per-phase cycle sums (blocking)
		static unsigned long long* dbuf = nullptr;
		if (!dbuf) (void)hipMalloc(&dbuf, 8192 * 24 * sizeof(unsigned long long));
		FusedParams q = prm; q.prof = dbuf;
		if (grid > 8192) grid = 8192;
		v.launch(q, dim3((uint32_t)grid), stream);
		(void)hipStreamSynchronize(stream);
		std::vector<unsigned long long> h(grid * 12);
		(void)hipMemcpy(h.data(), dbuf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
		double sum[12] = {};
		for (uint64_t w = 0; w < grid; w++) for (int i = 0; i < 12; i++) sum[i] += (double)h[w * 12 + i];
		const double nt = sum[7] > 0 ? sum[7] : 1;
		if (v.name && (std::string(v.name) == "pow2_fused_pk_kernel" || std::string(v.name) == "pow2_fused_pkh_kernel")) // (the packed kernels' own phase cut)
			fprintf(stderr, "[fused profile %s] grid %llu tickets/wg %.1f | cycles per ticket: S1 %.0f  decode+Breq %.0f  A-landed %.0f  waitA %.0f  A-stages %.0f  A-twiddle %.0f  A-turn+stores %.0f  stores-issued %.0f  drain %.0f  S3 %.0f  B-phase %.0f | total %.0f\n",
			        v.name, (unsigned long long)grid, nt / grid, sum[0] / nt, sum[3] / nt, sum[1] / nt, sum[5] / nt, sum[8] / nt, sum[9] / nt, sum[10] / nt, sum[2] / nt, sum[6] / nt, sum[11] / nt, sum[4] / nt,
			        (sum[0] + sum[1] + sum[2] + sum[3] + sum[4] + sum[5] + sum[6] + sum[8] + sum[9] + sum[10] + sum[11]) / nt);
		else
		fprintf(stderr, "[fused profile] grid %llu tickets/wg %.1f | cycles per ticket: S1 %.0f  A-load %.0f  A-stages %.0f A-twiddle %.0f A-transpose %.0f A-stores %.0f  B-load %.0f  B-compute %.0f  waitA %.0f waitB %.0f | total %.0f\n",
		        (unsigned long long)grid, nt / grid, sum[0] / nt, sum[1] / nt, sum[8] / nt, sum[9] / nt, sum[10] / nt, sum[2] / nt, sum[3] / nt, sum[4] / nt, sum[5] / nt, sum[6] / nt, (sum[0] + sum[1] + sum[2] + sum[3] + sum[4] + sum[5] + sum[6] + sum[8] + sum[9] + sum[10]) / nt);
		return 0;
	}
#endif
	v.launch(prm, dim3((uint32_t)grid), stream);
	return hipGetLastError() == hipSuccess ? 0 : 4039;
}

} // namespace vkfft_mi355x
