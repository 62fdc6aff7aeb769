// Instantiations, registries and launchers of the hand-specialised power-of-two kernels (kernel_pow2.h, kernel_pow2_lean.h): own translation
// unit, so that the other units that use the kernel templates (fused Four-Step, Bluestein-wrapped real transforms) do not compile them again.
#include "engine.h"
#include "kernel_pow2.h"
#include "kernel_pow2_pk.h"
#include <cstdio>
#include <cstdlib>

namespace vkfft_mi355x {

// ---- registry --------------------------------------------------------------------------------------------

template <typename T, typename SCH, int FPW> void pow2_row_launch(const PassParams& prm, dim3 grid, hipStream_t s) {
	constexpr int threads = ((1 << SCH::LOGN) >> SCH::LOGE) * FPW;
	hipLaunchKernelGGL((pow2_row_kernel<T, SCH, FPW>), grid, dim3(threads), 0, s, prm);
}

template <typename T, typename SCH, int TC> void pow2_col_launch(const PassParams& prm, dim3 grid, hipStream_t s) {
	constexpr int threads = ((1 << SCH::LOGN) >> SCH::LOGE) * TC;
	if (prm.bigSpan) hipLaunchKernelGGL((pow2_col_kernel<T, SCH, TC, true>), grid, dim3(threads), 0, s, prm);
	else hipLaunchKernelGGL((pow2_col_kernel<T, SCH, TC, false>), grid, dim3(threads), 0, s, prm);
}

#define VKFFT_P2C(T, dp, b0, b1, b2, b3, tc) \
	{ (b0) + (b1) + (b2) + (b3), dp, {b0, b1, b2, b3}, tc, (((1 << ((b0) + (b1) + (b2) + (b3))) >> Pow2Sched<b0, b1, b2, b3>::LOGE) * (tc)), &pow2_col_launch<T, Pow2Sched<b0, b1, b2, b3>, tc> }

#define VKFFT_P2(T, dp, b0, b1, b2, b3, fpw) \
	{ (b0) + (b1) + (b2) + (b3), dp, {b0, b1, b2, b3}, fpw, (((1 << ((b0) + (b1) + (b2) + (b3))) >> Pow2Sched<b0, b1, b2, b3>::LOGE) * (fpw)), &pow2_row_launch<T, Pow2Sched<b0, b1, b2, b3>, fpw> }

// register-lean rows (kernel_pow2_lean.h): 32 points per thread, one row per workgroup, wpe waves per SIMD, twiddles twg at a time
#define VKFFT_P2L(T, dp, b0, b1, b2, b3, wpe, twg, pfn) \
	{ (b0) + (b1) + (b2) + (b3), dp, {b0, b1, b2, b3}, 1, ((1 << ((b0) + (b1) + (b2) + (b3))) >> Pow2Sched<b0, b1, b2, b3>::LOGE), &pow2_row_lean_launch<T, Pow2Sched<b0, b1, b2, b3>, wpe, twg, pfn>, "pow2_row_lean_kernel" }

// the same on packed (x, y) register pairs (kernel_pow2_pk.h, round 5)
#define VKFFT_P2K(T, dp, b0, b1, b2, b3, wpe, twg) \
	{ (b0) + (b1) + (b2) + (b3), dp, {b0, b1, b2, b3}, 1, ((1 << ((b0) + (b1) + (b2) + (b3))) >> Pow2Sched<b0, b1, b2, b3>::LOGE), &pow2_row_lean_pk_launch<T, Pow2Sched<b0, b1, b2, b3>, wpe, twg>, "pow2_row_lean_pk_kernel" }

// ... several rows per workgroup (2^10 ... 2^12)
#define VKFFT_P2KF(T, dp, b0, b1, b2, b3, wpe, twg, fpw) \
	{ (b0) + (b1) + (b2) + (b3), dp, {b0, b1, b2, b3}, fpw, ((1 << ((b0) + (b1) + (b2) + (b3))) >> Pow2Sched<b0, b1, b2, b3>::LOGE) * (fpw), &pow2_row_lean_pk_launch<T, Pow2Sched<b0, b1, b2, b3>, wpe, twg, fpw>, "pow2_row_lean_pk_kernel" }

// 2^15 as 2^14 pairs of neighbouring samples, persistent and software-pipelined over the rows (bits: the pair transform's three stages + the last layer's run of the table)
#define VKFFT_P2R(T, dp, b0, b1, b2, twg) \
	{ (b0) + (b1) + (b2) + 1, dp, {b0, b1, b2, 1}, 1, ((1 << ((b0) + (b1) + (b2))) >> Pow2Sched<b0, b1, b2, 0>::LOGE), &pow2_row_pairs_launch<T, Pow2Sched<b0, b1, b2, 0>, twg>, "pow2_row_pairs_kernel", true }

// first entry of each (log2n, dp) is the default; VKFFT_MI355X_P2V<log2n>=k selects the k-th (tuning)
static const Pow2Variant kPow2Variants[] = {
	// fp32
	VKFFT_P2(float, false, 2, 0, 0, 0, 64),
	VKFFT_P2(float, false, 3, 0, 0, 0, 64),
	VKFFT_P2(float, false, 4, 0, 0, 0, 64),
	VKFFT_P2(float, false, 3, 2, 0, 0, 32),
	VKFFT_P2(float, false, 3, 3, 0, 0, 32),
	VKFFT_P2(float, false, 4, 3, 0, 0, 16),
	VKFFT_P2(float, false, 4, 4, 0, 0, 8),
	// round 5: 2^9 / 2^10 on the packed (x, y) rows of kernel_pow2_pk.h, several rows per workgroup (A/B on three boxes, profiles/r05_ab_small_sizes_*: 2^9 5.72 -> 6.09 /
	// 5.94 -> 5.98, 2^10 5.42 -> 6.07 / 5.58 -> 5.92 TB/s paired).  2^11 / 2^12: their packed form (16 points per thread, 68-80 VGPRs, six to seven
	// waves per SIMD) won on one box (5.24 -> 5.69 / 5.67) and tied or lost by 1 % on two (5.44 / 5.48 -> 5.41 / 5.43; 5.33 / 5.36 -> 5.27 / 5.31); 2^8 stays: 6.07
	// against 5.99.  Index 1 of 2^9 / 2^10 = the kernel that shipped before; the other shapes of rounds 1-4 that lost their comparisons are no longer instantiated
	VKFFT_P2KF(float, false, 5, 4, 0, 0, 4, 16, 16), VKFFT_P2(float, false, 5, 4, 0, 0, 8),
	VKFFT_P2KF(float, false, 5, 5, 0, 0, 4, 16, 8), VKFFT_P2(float, false, 5, 5, 0, 0, 8),
	// (last measurement of the round, a fourth box, the reference in the same lease — profiles/r05_ab_2p11_2p12_packed_rows_fourth_box.jsonl: 2^11 5.20-5.23 -> 5.62-5.70, 2^12
	// 5.16-5.23 -> 5.56-5.60, the reference 5.66 / 5.44: never more than 1 % behind, 8 % ahead on two boxes of four — the packed form is the default now, the round-1 kernel index 1)
	VKFFT_P2KF(float, false, 4, 4, 3, 0, 5, 16, 4), VKFFT_P2(float, false, 5, 5, 1, 0, 2),
	VKFFT_P2KF(float, false, 4, 4, 4, 0, 5, 16, 1), VKFFT_P2(float, false, 4, 4, 4, 0, 1),
	// 2^13: packed register-lean rows, four 256-thread workgroups per CU; index 1 the round-1..3 kernel (two 67 KiB workgroups per CU), index 2 the round-4 lean kernel
	VKFFT_P2K(float, false, 5, 4, 4, 0, 4, 16), VKFFT_P2(float, false, 5, 4, 4, 0, 1), VKFFT_P2L(float, false, 5, 4, 4, 0, 4, 16, 0),
	// 2^14: two 512-thread workgroups per CU; index 1 the round-1..3 kernel (one 135 KiB workgroup per CU), index 2 the round-4 lean kernel
	VKFFT_P2K(float, false, 5, 5, 4, 0, 4, 16), VKFFT_P2(float, false, 5, 5, 4, 0, 1), VKFFT_P2L(float, false, 5, 5, 4, 0, 4, 16, 0),
	// 2^15 in ONE pass (VKFFT_MI355X_ROW15=0: the fused two-pass kernel): as 2^14 pairs of neighbouring samples in persistent 512-thread workgroups, half of the next
	// row in flight (pow2_row_pairs_kernel, round 5: 4.44-4.48 against 4.33-4.35 TB/s); index 1: one row in 1024 threads x 32 points on packed pairs (also what a
	// zero-padded row takes: the pairs kernel has no masks), index 2 the round-4 lean kernel
	VKFFT_P2R(float, false, 5, 5, 4, 8), VKFFT_P2K(float, false, 5, 5, 5, 0, 4, 16), VKFFT_P2L(float, false, 5, 5, 5, 0, 4, 16, 0),
	// fp64
	VKFFT_P2(double, true, 2, 0, 0, 0, 64),
	VKFFT_P2(double, true, 3, 0, 0, 0, 64),
	VKFFT_P2(double, true, 4, 0, 0, 0, 64),
	VKFFT_P2(double, true, 3, 2, 0, 0, 32),
	VKFFT_P2(double, true, 3, 3, 0, 0, 32),
	VKFFT_P2(double, true, 3, 2, 2, 0, 16),
	VKFFT_P2(double, true, 3, 3, 2, 0, 8),
	VKFFT_P2(double, true, 3, 3, 3, 0, 4),
	VKFFT_P2(double, true, 3, 3, 2, 2, 2), VKFFT_P2(double, true, 4, 3, 3, 0, 2),
	VKFFT_P2(double, true, 3, 3, 3, 2, 1), VKFFT_P2(double, true, 4, 4, 3, 0, 1),
	VKFFT_P2(double, true, 3, 3, 3, 3, 1), VKFFT_P2(double, true, 4, 4, 4, 0, 1),
	VKFFT_P2(double, true, 4, 3, 3, 3, 1),
};
constexpr int kNumPow2Variants = (int)(sizeof(kPow2Variants) / sizeof(kPow2Variants[0]));

// column kernels: first entry of each (log2n, dp) is the default; VKFFT_MI355X_P2C<log2n>=k selects the k-th
static const Pow2Variant kPow2ColVariants[] = {
	VKFFT_P2C(float, false, 1, 0, 0, 0, 64), VKFFT_P2C(float, false, 2, 0, 0, 0, 64), VKFFT_P2C(float, false, 3, 0, 0, 0, 64), // thin axes (a depth of 2, 4, 8): one butterfly per thread
	VKFFT_P2C(float, false, 2, 2, 0, 0, 32), VKFFT_P2C(float, false, 2, 2, 0, 0, 16),
	VKFFT_P2C(float, false, 3, 2, 0, 0, 32), VKFFT_P2C(float, false, 3, 2, 0, 0, 16),
	VKFFT_P2C(float, false, 3, 3, 0, 0, 32), VKFFT_P2C(float, false, 3, 3, 0, 0, 16),
	VKFFT_P2C(float, false, 4, 3, 0, 0, 32), VKFFT_P2C(float, false, 4, 3, 0, 0, 16), VKFFT_P2C(float, false, 3, 2, 2, 0, 32),
	VKFFT_P2C(float, false, 4, 4, 0, 0, 32), VKFFT_P2C(float, false, 4, 4, 0, 0, 16), VKFFT_P2C(float, false, 3, 3, 2, 0, 32), VKFFT_P2C(float, false, 3, 3, 2, 0, 16), VKFFT_P2C(float, false, 5, 3, 0, 0, 32),
	VKFFT_P2C(float, false, 5, 4, 0, 0, 32), VKFFT_P2C(float, false, 4, 3, 2, 0, 16), VKFFT_P2C(float, false, 4, 3, 2, 0, 32), VKFFT_P2C(float, false, 3, 3, 3, 0, 16), VKFFT_P2C(float, false, 5, 4, 0, 0, 16),
	VKFFT_P2C(float, false, 5, 5, 0, 0, 16), VKFFT_P2C(float, false, 4, 3, 3, 0, 16), VKFFT_P2C(float, false, 4, 3, 3, 0, 8),
	VKFFT_P2C(double, true, 1, 0, 0, 0, 32), VKFFT_P2C(double, true, 2, 0, 0, 0, 32), VKFFT_P2C(double, true, 3, 0, 0, 0, 32),
	VKFFT_P2C(double, true, 2, 2, 0, 0, 16),
	VKFFT_P2C(double, true, 3, 2, 0, 0, 16),
	VKFFT_P2C(double, true, 3, 3, 0, 0, 16),
	VKFFT_P2C(double, true, 3, 2, 2, 0, 16), VKFFT_P2C(double, true, 4, 3, 0, 0, 16),
	VKFFT_P2C(double, true, 3, 3, 2, 0, 16), VKFFT_P2C(double, true, 4, 4, 0, 0, 16),
	VKFFT_P2C(double, true, 3, 3, 3, 0, 8), VKFFT_P2C(double, true, 4, 3, 2, 0, 8), VKFFT_P2C(double, true, 3, 3, 3, 0, 16),
	VKFFT_P2C(double, true, 4, 3, 3, 0, 8),
};
constexpr int kNumPow2ColVariants = (int)(sizeof(kPow2ColVariants) / sizeof(kPow2ColVariants[0]));

// fused Bluestein on a power-of-two padded length: one entry per (log2 M, dp)
template <typename T, typename SCH, int FPW> void pow2_blue_launch(const PassParams& prm, dim3 grid, hipStream_t s) {
	constexpr int threads = ((1 << SCH::LOGN) >> SCH::LOGE) * FPW;
	const unsigned resident = pow2_num_cus() * 8u; // persistent: the workgroups stride over the row tiles
	hipLaunchKernelGGL((pow2_blue_kernel<T, SCH, FPW>), dim3(grid.x < resident ? grid.x : resident), dim3(threads), 0, s, prm);
}
#define VKFFT_P2B(T, dp, b0, b1, b2, b3, fpw) \
	{ (b0) + (b1) + (b2) + (b3), dp, {b0, b1, b2, b3}, fpw, (((1 << ((b0) + (b1) + (b2) + (b3))) >> Pow2Sched<b0, b1, b2, b3>::LOGE) * (fpw)), &pow2_blue_launch<T, Pow2Sched<b0, b1, b2, b3>, fpw> }
static const Pow2Variant kPow2BlueVariants[] = {
	VKFFT_P2B(float, false, 3, 3, 0, 0, 32),
	VKFFT_P2B(float, false, 4, 3, 0, 0, 16),
	VKFFT_P2B(float, false, 4, 4, 0, 0, 16),
	VKFFT_P2B(float, false, 4, 3, 2, 0, 8),
	VKFFT_P2B(float, false, 4, 3, 3, 0, 4),
	VKFFT_P2B(float, false, 4, 4, 3, 0, 2),
	VKFFT_P2B(float, false, 4, 4, 4, 0, 1),
	VKFFT_P2B(float, false, 4, 3, 3, 3, 1),
	VKFFT_P2B(float, false, 4, 4, 3, 3, 1),
	VKFFT_P2B(double, true, 3, 3, 0, 0, 32),
	VKFFT_P2B(double, true, 3, 2, 2, 0, 16),
	VKFFT_P2B(double, true, 3, 3, 2, 0, 8),
	VKFFT_P2B(double, true, 3, 3, 3, 0, 4),
	VKFFT_P2B(double, true, 3, 3, 2, 2, 2),
	VKFFT_P2B(double, true, 3, 3, 3, 2, 1),
	VKFFT_P2B(double, true, 3, 3, 3, 3, 1),
	VKFFT_P2B(double, true, 4, 3, 3, 3, 1),
};
constexpr int kNumPow2BlueVariants = (int)(sizeof(kPow2BlueVariants) / sizeof(kPow2BlueVariants[0]));

// multi-pass Bluestein column kernels: one entry per (log2 L, dp, mode); Pow2Variant::fpw holds the tile width
template <typename T, typename SCH, int TC, int MODE> void pow2_col_blue_launch(const PassParams& prm, dim3 grid, hipStream_t s) {
	constexpr int threads = ((1 << SCH::LOGN) >> SCH::LOGE) * TC;
	hipLaunchKernelGGL((pow2_col_blue_kernel<T, SCH, TC, MODE>), grid, dim3(threads), 0, s, prm);
}
struct Pow2ColBlueVariant { Pow2Variant v; int mode; };
#define VKFFT_P2CB1(T, dp, b0, b1, b2, b3, tc, mode) \
	{ { (b0) + (b1) + (b2) + (b3), dp, {b0, b1, b2, b3}, tc, (((1 << ((b0) + (b1) + (b2) + (b3))) >> Pow2Sched<b0, b1, b2, b3>::LOGE) * (tc)), &pow2_col_blue_launch<T, Pow2Sched<b0, b1, b2, b3>, tc, mode> }, mode }
#define VKFFT_P2CB(T, dp, b0, b1, b2, b3, tc) VKFFT_P2CB1(T, dp, b0, b1, b2, b3, tc, 1), VKFFT_P2CB1(T, dp, b0, b1, b2, b3, tc, 2), VKFFT_P2CB1(T, dp, b0, b1, b2, b3, tc, 3), VKFFT_P2CB1(T, dp, b0, b1, b2, b3, tc, 4), \
	VKFFT_P2CB1(T, dp, b0, b1, b2, b3, tc, 5), VKFFT_P2CB1(T, dp, b0, b1, b2, b3, tc, 6), VKFFT_P2CB1(T, dp, b0, b1, b2, b3, tc, 8)
static const Pow2ColBlueVariant kPow2ColBlueVariants[] = {
	VKFFT_P2CB(float, false, 3, 3, 0, 0, 32),
	VKFFT_P2CB(float, false, 4, 3, 0, 0, 32),
	VKFFT_P2CB(float, false, 4, 4, 0, 0, 32),
	VKFFT_P2CB(float, false, 4, 3, 2, 0, 16),
	VKFFT_P2CB(float, false, 4, 3, 3, 0, 16),
	VKFFT_P2CB1(float, false, 4, 4, 3, 0, 8, 5), // one-pass column Bluestein on 2048 padded points (147 KiB tile)
	VKFFT_P2CB1(float, false, 4, 3, 3, 0, 8, 7), // merged matrix convolution along 1024 points: 8-column tiles, 512 threads (three coordinate systems in registers)
	VKFFT_P2CB1(double, true, 3, 3, 3, 0, 8, 7), // ... 512 points in double precision
	VKFFT_P2CB(double, true, 3, 3, 0, 0, 16),
	VKFFT_P2CB(double, true, 3, 2, 2, 0, 16),
	VKFFT_P2CB(double, true, 3, 3, 2, 0, 16),
	VKFFT_P2CB(double, true, 3, 3, 3, 0, 8),
	VKFFT_P2CB(double, true, 4, 3, 3, 0, 8),
};
constexpr int kNumPow2ColBlueVariants = (int)(sizeof(kPow2ColBlueVariants) / sizeof(kPow2ColBlueVariants[0]));

int launch_pow2_col_blue(const PassPlan& pp, const PassParams& prm, hipStream_t stream) {
	const uint64_t grid64 = (uint64_t)prm.tilesPerG0 * prm.dim[1].count * prm.dim[2].count;
	if (grid64 == 0) return 0;
	if (grid64 > 0x7fffffffull || pp.variant < 0 || pp.variant >= kNumPow2ColBlueVariants) return 4039;
	kPow2ColBlueVariants[pp.variant].v.launch(prm, dim3((uint32_t)grid64), stream);
	return hipGetLastError() == hipSuccess ? 0 : 4039;
}

int launch_pow2_blue(const PassPlan& pp, const PassParams& prm, hipStream_t stream) {
	const uint64_t grid64 = (uint64_t)prm.tilesPerG0 * prm.dim[1].count * prm.dim[2].count;
	if (grid64 == 0) return 0;
	if (grid64 > 0x7fffffffull || pp.variant < 0 || pp.variant >= kNumPow2BlueVariants) return 4039;
	kPow2BlueVariants[pp.variant].launch(prm, dim3((uint32_t)grid64), stream);
	return hipGetLastError() == hipSuccess ? 0 : 4039;
}

const char* pow2_row_kernel_name(int variant) {
	if (variant < 0 || variant >= kNumPow2Variants || !kPow2Variants[variant].name) return "pow2_row_kernel";
	return kPow2Variants[variant].name;
}

int launch_pow2(const PassPlan& pp, const PassParams& prm, hipStream_t stream) {
	const uint64_t grid64 = (uint64_t)prm.tilesPerG0 * (prm.colMerge ? 1u : prm.dim[1].count) * prm.dim[2].count;
	if (grid64 == 0) return 0;
	const bool col = pp.kernel == KERNEL_POW2_COL;
	if (grid64 > 0x7fffffffull || pp.variant < 0 || pp.variant >= (col ? kNumPow2ColVariants : kNumPow2Variants)) return 4039;
	(col ? kPow2ColVariants : kPow2Variants)[pp.variant].launch(prm, dim3((uint32_t)grid64), stream);
	return hipGetLastError() == hipSuccess ? 0 : 4039;
}


static bool pow2_lookup(const Pow2Variant* tab, int ntab, const char* envPrefix, uint32_t log2n, bool dp, int* variant, int bits[4], int* fpw, int* threads, bool padded = false) {
	int want = 0;
	char name[64];
	snprintf(name, sizeof(name), "%s%u", envPrefix, log2n);
	if (const char* e = getenv(name)) want = atoi(e);
	int seen = 0, found = -1;
	for (int i = 0; i < ntab; i++) {
		if (tab[i].log2n != (int)log2n || tab[i].dp != dp) continue;
		if (padded && tab[i].noPadMasks) { seen++; continue; }
		if (found < 0) found = i;
		if (seen == want) { found = i; break; }
		seen++;
	}
	if (found < 0) return false;
	*variant = found;
	for (int k = 0; k < 4; k++) bits[k] = tab[found].bits[k];
	*fpw = tab[found].fpw; *threads = tab[found].threads;
	return true;
}
bool pow2_row_lookup(uint32_t log2n, bool dp, int* variant, int bits[4], int* fpw, int* threads, bool padded) {
	return pow2_lookup(kPow2Variants, kNumPow2Variants, "VKFFT_MI355X_P2V", log2n, dp, variant, bits, fpw, threads, padded);
}
bool pow2_col_lookup(uint32_t log2n, bool dp, int* variant, int bits[4], int* tc, int* threads) {
	return pow2_lookup(kPow2ColVariants, kNumPow2ColVariants, "VKFFT_MI355X_P2C", log2n, dp, variant, bits, tc, threads);
}

bool pow2_col_blue_lookup(uint32_t log2l, bool dp, int mode, int* variant, int bits[4], int* tc, int* threads) {
	for (int i = 0; i < kNumPow2ColBlueVariants; i++) {
		const Pow2ColBlueVariant& e = kPow2ColBlueVariants[i];
		if (e.v.log2n != (int)log2l || e.v.dp != dp || e.mode != mode) continue;
		*variant = i;
		for (int k = 0; k < 4; k++) bits[k] = e.v.bits[k];
		*tc = e.v.fpw; *threads = e.v.threads;
		return true;
	}
	return false;
}
bool pow2_blue_lookup(uint32_t log2m, bool dp, int* variant, int bits[4], int* fpw, int* threads) {
	return pow2_lookup(kPow2BlueVariants, kNumPow2BlueVariants, "VKFFT_MI355X_P2B", log2m, dp, variant, bits, fpw, threads);
}


} // namespace vkfft_mi355x
