// Cyclic convolution through a mixed-radix transform pair in ONE kernel: rows and strided axes whose length has a prime factor above 13.
//   BLUESTEIN (RADER = 0): x[j] conj(chirp[j]), zero padded to M -> FFT_M -> * FFT(chirp)/M -> inverse FFT_M -> * conj(chirp[k]), k < n, for ANY
//     13-smooth padded length M >= 2n-1 (the power-of-two form is pow2_blue_kernel; the reference picks its padded length from the same kind of
//     list, vkFFT_Scheduler.h:2406-2578, and runs the two transforms as separate kernels, vkFFT_Bluestein.h:32,201).
//   RADER (RADER = 1): a prime length p with 13-smooth p-1.  With g a primitive root, X[g^-q] = x[0] + sum_a x[g^a] w^(g^(a-q)): a cyclic convolution
//     of length L = p-1 EXACTLY — no chirp, no padding, half the points of the Bluestein form or less (reference: the FFT-Rader stage of
//     vkFFT_RaderKernels.h:1278, generator tables of vkFFT_RecursiveFFTGenerators.h:1021-1048).
// Structure: the compile-time radix schedule and the single padded LDS exchange buffer of kernel_mixed.h; the transform's first stage takes its inputs
// from a functor and its last stage hands its outputs to one.  A second LDS row per transform carries the row between the phases: (Rader) the row as it
// arrives, read through the generator permutation; the spectrum times the kernel spectrum, written in natural order by the forward transform's last
// stage and read by the inverse transform's first stage; (Rader) the result, scattered through the inverse permutation so that the global store is as
// coalesced as the load.  The inverse transform is the forward one between two re/im swaps.
#pragma once
#include "engine.h"
#include "butterflies.h"
#include "memops.h"
#include "mix_sched.h"
#include "mix_stage.h"
#include "kernel_generic.h"
#include "kernel_mixrad.h"
#include "kernel_tmaps.h"

namespace vkfft_mi355x {

// lut = stage twiddles of SCH; aux2 = FFT of the convolution kernel / L, natural order; Bluestein: aux = chirp (opN entries), opN = n;
// Rader: rader = uint32 g^a mod p (a < L) followed by g^-k mod p (k < L).
// COL = 0: FPW unit-stride rows per workgroup, TPF threads each.  COL = 1: a tile of FPW neighbouring columns of a strided axis, lanes along the
// columns (every global access is an FPW-element segment), element j of column c at j*inStrideJ + c*dim[0].inStride.
// OPS = 1 (Rader rows only): the row enters through the interpreter's gather-load and leaves through its gather-store (kernel_generic.h pre_gather /
// post_store): real transforms whose complex length is a Rader prime (R2C of 2 x 37 points, DCT-II of 61 ...).  The kernel spectrum then comes from aux3
// (aux / aux2 belong to the real transform's own tables), the swaps are the pass's swapIn / swapOut and post_store applies the scale.
template <typename T, typename SCH, int TPF, int FPW, int RADER, int COL, int OPS = 0>
__global__ void __launch_bounds__(TPF * FPW) mixconv_kernel(const PassParams p) {
	static_assert(OPS == 0 || (RADER != 0 && COL == 0), "pre / post maps: Rader rows");
	constexpr int L = SCH::N, NT = TPF * FPW;
	constexpr int LS = COL ? FPW + 1 : 1;                 // LDS pitch between consecutive elements of one transform
	// ONE buffer per transform: the exchange buffer of the stages, which between the phases carries the row in natural order (Rader: the row as it
	// arrives; the spectrum times the kernel spectrum; Rader: the result before the coalesced store)
	constexpr int EXPF = SCH::NS > 1 ? (COL ? L : MixPad<SCH, TPF, (int)sizeof(cx<T>)>::elems()) : 1;
	constexpr int SPN = L + (RADER ? 1 : 0);
	constexpr int SP = COL ? (EXPF > SPN ? EXPF : SPN) : ((EXPF > SPN ? EXPF : SPN) | 1); // rows of a tile: odd pitch
	constexpr int EXN = COL ? SP * LS : FPW * SP;
	constexpr bool waveOnly = COL ? NT <= 64 : ((TPF <= 64) && (64 % TPF == 0)); // a transform never straddles wavefronts
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	static_assert((size_t)(EXN + FPW) * sizeof(cx<T>) <= 160 * 1024, "LDS");
	__shared__ cx<T> lds[EXN];
	cx<T>* const rows = lds;
	__shared__ cx<T> sDc[FPW];
	const uint32_t tid = threadIdx.x;
	const uint32_t f = COL ? tid % FPW : tid / TPF, tau = COL ? tid / FPW : tid % TPF;
	uint32_t wg = p.reverseTiles ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
	const uint32_t tile = wg % p.tilesPerG0;
	wg /= p.tilesPerG0;
	const bool merge = COL && p.colMerge != 0;
	const uint32_t g1 = merge ? 0u : wg % p.dim[1].count, g2 = merge ? wg : wg / p.dim[1].count;
	const uint32_t rowMult = (OPS != 0 && p.pairRows) ? 2u : 1u; // two real rows per transform (kernel_generic.h)
	const uint32_t f0 = tile * FPW * rowMult;
	const GBuf glut = make_gbuf(p.lut), gbh = make_gbuf(OPS ? p.aux3 : p.aux2);
	// element j of this thread's transform: byte offset laneIn + j*sJin (rows: sJin = ES); guarded by `valid` at every use
	const uint32_t sJin = COL ? (uint32_t)p.inStrideJ * ES : ES, sJout = COL ? (uint32_t)p.outStrideJ * ES : ES;
	bool valid;
	uint32_t laneIn, laneOut;
	int64_t baseIn = (int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride, baseOut = (int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride;
	if (merge) {
		// column g = f0 + f of the merged index space: (g % c0) along dim[0], (g / c0) along dim[1]; the tile's first dim[1] index goes into the (uniform) base
		const uint32_t c0 = p.dim[0].count, q0 = f0 / c0, g = f0 + f, q = g / c0, r = g - q * c0;
		valid = q < p.dim[1].count;
		baseIn += (int64_t)q0 * p.dim[1].inStride; baseOut += (int64_t)q0 * p.dim[1].outStride;
		laneIn = (uint32_t)((int64_t)r * p.dim[0].inStride + (int64_t)(q - q0) * p.dim[1].inStride) * ES;
		laneOut = (uint32_t)((int64_t)r * p.dim[0].outStride + (int64_t)(q - q0) * p.dim[1].outStride) * ES;
	} else {
		valid = f0 + f < p.dim[0].count;
		baseIn += (int64_t)f0 * p.dim[0].inStride; baseOut += (int64_t)f0 * p.dim[0].outStride;
		laneIn = (f * (uint32_t)p.dim[0].inStride) * ES; laneOut = (f * (uint32_t)p.dim[0].outStride) * ES;
	}
	const GBuf gin = make_gbuf((const cx<T>*)p.in + baseIn);
	const GBuf gout = make_gbuf((cx<T>*)p.out + baseOut);
	cx<T>* const ex = COL ? lds + f : lds + f * SP;
	cx<T>* const row = ex;
	const bool swI = OPS ? p.swapIn != 0 : p.bluesteinSwapIn != 0, swO = OPS ? p.swapOut != 0 : p.bluesteinSwapOut != 0;
	const T sc = OPS ? (T)1 : (T)p.scale;
	auto fsync = [&]() { if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC(); };
	auto fromRow = [&](uint32_t t, uint32_t c) -> cx<T> { return row[(t + c) * LS]; };

	if constexpr (RADER) {
		constexpr uint32_t n = (uint32_t)L + 1u;
		const uint32_t* const gp = (const uint32_t*)p.rader;
		const uint32_t rowsHere = p.dim[0].count - f0 < (uint32_t)FPW * rowMult ? p.dim[0].count - f0 : (uint32_t)FPW * rowMult;
		const bool denseIn = !COL && p.dim[0].inStride == (int64_t)n, denseOut = !COL && p.dim[0].outStride == (int64_t)n;
		// ---- the row as it lies in memory -> LDS (dense rows: the tile is one contiguous run)
		if constexpr (OPS != 0) {
			const int64_t inB = (int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride + (int64_t)f0 * p.dim[0].inStride;
			if (p.tmPreFlags & kTmOn) {
				// table-driven pre-map (kernel_tmaps.h): thread group f fills its own row, every point's loads in flight at once
				const uint32_t rA = f * rowMult;
				const uint32_t pitch = (uint32_t)p.dim[0].inStride * p.inElemBytes;
				const TmSide<T> ts = tm_side<T>(p.tmPre, n, make_gbuf((const char*)p.in + inB * (int64_t)p.inElemBytes), rA < rowsHere ? rA * pitch : kGbInvalid,
				                                (rowMult == 2u && rA + 1u < rowsHere) ? (rA + 1u) * pitch : kGbInvalid);
				constexpr int PJ = ((int)n + TPF - 1) / TPF;
				auto fillRow = [&](auto twoTag) { // (the number of terms outside the loop: a branch per point would put a wait between the points' loads)
#pragma unroll
					for (int b = 0; b < PJ; b++) {
						const uint32_t j = tau + (uint32_t)(b * TPF);
						if ((b + 1) * TPF <= (int)n || j < n) {
							const cx<T> v = tm_pre<T, decltype(twoTag)::value>(ts, TmGlobal<T>{ts.data}, tau, (uint32_t)(b * TPF));
							row[j] = swI ? cswap(v) : v;
						}
					}
				};
				if (p.tmPreFlags & kTmTwo) fillRow(std::true_type{}); else fillRow(std::false_type{});
			} else {
				// (the operation hoisted out of the loop: ops_rows_in, kernel_generic.h)
				FastDiv divN; divN.d = n; divN.rcp = 1.0f / (float)n;
				dispatch_pre_op(p.preOp, [&](auto opc) { ops_rows_in<T>(p, opc, divN, rows, (uint32_t)SP, (uint32_t)FPW * n, rowsHere, inB, f0 * p.opStride0 + g1 * p.opStride1); });
			}
		} else if (denseIn) {
			for (uint32_t e = tid; e < rowsHere * n; e += (uint32_t)NT) {
				const uint32_t j = e % n;
				const cx<T> v = gb_load<T>(gin, (j - p.padInL < p.padInN) ? kGbInvalid : e * ES, 0); // (zero padding: the padded range is not read, vkFFT_Zeropad.h:28)
				rows[(e / n) * SP + j] = swI ? cswap(v) : v;
			}
		} else {
			for (uint32_t j = tau; j < n; j += (uint32_t)TPF) {
				const cx<T> v = gb_load<T>(gin, (valid && !(j - p.padInL < p.padInN)) ? laneIn + j * sJin : kGbInvalid, 0);
				row[j * LS] = swI ? cswap(v) : v;
			}
		}
		VKFFT_SYNC();
		const cx<T> x0 = row[0];
		// ---- forward transform of x[g^a]; spectrum * FFT(w^(g^-q)) / L, + x0 on the zero frequency (= x0 added to every output)
		mc_stage<T, SCH, 0, TPF, LS, !COL, true, true>(ex, glut, tau, waveOnly, [&](uint32_t t, uint32_t c) -> cx<T> { return row[gp[t + c] * LS]; },
		                                   [&](uint32_t t, uint32_t c, cx<T> v) {
			                                   const uint32_t k = t + c;
			                                   cx<T> w = cmul(v, gb_load<T>(gbh, t * ES, c * ES));
			                                   if (k == 0u) { sDc[f] = cadd(x0, v); w = cadd(w, x0); } // X[0] = x0 + sum of the others
			                                   row[k * LS] = cswap(w);
		                                   });
		fsync();
		// ---- inverse transform; result q belongs to output index g^-q
		mc_stage<T, SCH, 0, TPF, LS, !COL, true, true>(ex, glut, tau, waveOnly, fromRow, [&](uint32_t t, uint32_t c, cx<T> v) { row[gp[(uint32_t)L + t + c] * LS] = cswap(v); });
		VKFFT_SYNC();
		auto fin = [&](cx<T> v) { if (swO) v = cswap(v); if (sc != (T)1) v = cscale(v, sc); return v; };
		if constexpr (OPS != 0) {
			const int64_t outB = (int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride + (int64_t)f0 * p.dim[0].outStride;
			if (p.tmPostFlags & kTmOn) {
				const uint32_t rA = f * rowMult, flags = p.tmPostFlags;
				const uint32_t pitch = (uint32_t)p.dim[0].outStride * p.outElemBytes;
				const bool split = (flags & kTmSplit) != 0u;
				const TmSide<T> to = tm_side<T>(p.tmPost, split ? n / 2u + 1u : n, make_gbuf((char*)p.out + outB * (int64_t)p.outElemBytes), rA < rowsHere ? rA * pitch : kGbInvalid,
				                                (rowMult == 2u && rA + 1u < rowsHere) ? (rA + 1u) * pitch : kGbInvalid);
				auto rd = [&](uint32_t a) -> cx<T> { const cx<T> v = a == 0u ? sDc[f] : row[a]; return swO ? cswap(v) : v; };
				if (split) tm_post_split<T, (int)n, TPF>(to, TmGlobal<T>{to.data}, flags, tau, rd); else tm_post_rows<T, (int)n, TPF>(to, TmGlobal<T>{to.data}, flags, tau, rd);
			} else
			dispatch_post_op(p.postOp, [&](auto opc) { ops_rows_out<T>(p, opc, rows, sDc, (uint32_t)SP, (uint32_t)FPW, rowsHere, outB, f0 * p.opStride0 + g1 * p.opStride1, n); });
		} else if (denseOut) {
			for (uint32_t e = tid; e < rowsHere * n; e += (uint32_t)NT) {
				const uint32_t r = e / n, j = e % n;
				gb_store<T>(gout, (j - p.padOutL < p.padOutN) ? kGbInvalid : e * ES, 0, fin(j == 0u ? sDc[r] : rows[r * SP + j]));
			}
		} else {
			for (uint32_t j = tau; j < n; j += (uint32_t)TPF) gb_store<T>(gout, (valid && !(j - p.padOutL < p.padOutN)) ? laneOut + j * sJout : kGbInvalid, 0, fin(j == 0u ? sDc[f] : row[j * LS]));
		}
	} else {
		const uint32_t n = p.opN;
		const GBuf gch = make_gbuf(p.aux);
		mc_stage<T, SCH, 0, TPF, LS, !COL, false, true>(ex, glut, tau, waveOnly,
		                                   [&](uint32_t t, uint32_t c) -> cx<T> {
			                                   const bool in = t + c < n && !(t + c - p.padInL < p.padInN); // the rest is the zero padding: nothing is read (the caller's padded range included)
			                                   cx<T> v = gb_load<T>(gin, in && valid ? laneIn + t * sJin : kGbInvalid, c * sJin);
			                                   if (swI) v = cswap(v);
			                                   return cmulc(v, gb_load<T>(gch, in ? t * ES : kGbInvalid, c * ES));
		                                   },
		                                   [&](uint32_t t, uint32_t c, cx<T> v) { row[(t + c) * LS] = cswap(cmul(v, gb_load<T>(gbh, t * ES, c * ES))); });
		fsync();
		mc_stage<T, SCH, 0, TPF, LS, !COL, true, false>(ex, glut, tau, waveOnly, fromRow, [&](uint32_t t, uint32_t c, cx<T> v) {
			if (t + c < n && !(t + c - p.padOutL < p.padOutN)) {
				cx<T> y = cmulc(cswap(v), gb_load<T>(gch, t * ES, c * ES));
				if (swO) y = cswap(y);
				if (sc != (T)1) y = cscale(y, sc);
				gb_store<T>(gout, valid ? laneOut + t * sJout : kGbInvalid, c * sJout, y);
			}
		});
	}
}

// ---- registry ---------------------------------------------------------------------------------------------------
struct MixConvVariant {
	int l; bool dp; int rader; int col; int rad[5]; int tpf; int fpw;
	void (*launch)(const PassParams&, dim3, hipStream_t);
	void (*launchOps)(const PassParams&, dim3, hipStream_t); // Rader rows: the form with the interpreter's pre / post maps (nullptr otherwise)
	void (*launchRad)(const PassParams&, dim3, hipStream_t); // Rader rows: the prime as a stage of a composite length M * P (kernel_mixrad.h; nullptr otherwise)
	int radSP, radLutN, radGroups, radGroupsDense;                         // ... its buffer pitch, stage-twiddle count and thread groups of the wave-aligned layout (0: none); the planner sizes the tile with them
};
template <typename T, typename SCH, int TPF, int FPW, int RADER, int COL, int OPS> void mixconv_launch(const PassParams& prm, dim3 grid, hipStream_t s) {
	hipLaunchKernelGGL((mixconv_kernel<T, SCH, TPF, FPW, RADER, COL, OPS>), grid, dim3(TPF * FPW), 0, s, prm);
}
template <typename T, typename SCH, int TPF, int FPW, int RADER, int COL> constexpr auto mixconv_ops_ptr() -> void (*)(const PassParams&, dim3, hipStream_t) {
	if constexpr (RADER != 0 && COL == 0) return &mixconv_launch<T, SCH, TPF, FPW, RADER, COL, 1>;
	else return nullptr;
}
#define VKFFT_MC(T, dp, rader, col, r0, r1, r2, r3, r4, tpf, fpw) \
	{ (r0) * (r1) * (r2) * (r3) * (r4), dp, rader, col, {r0, r1, r2, r3, r4}, tpf, fpw, &mixconv_launch<T, MixSched<r0, r1, r2, r3, r4>, tpf, fpw, rader, col, 0>, \
	  mixconv_ops_ptr<T, MixSched<r0, r1, r2, r3, r4>, tpf, fpw, rader, col>(), mixrad_ptr<T, MixSched<r0, r1, r2, r3, r4>, tpf, fpw, rader, col>(), \
	  mixrad_sp<T, MixSched<r0, r1, r2, r3, r4>, tpf, rader, col>(), mixrad_lutn<T, MixSched<r0, r1, r2, r3, r4>, tpf, rader, col>(), mixrad_groups<T, MixSched<r0, r1, r2, r3, r4>, tpf, fpw, rader, col>(), mixrad_groups_dense<T, MixSched<r0, r1, r2, r3, r4>, tpf, fpw, rader, col>() },

} // namespace vkfft_mi355x
