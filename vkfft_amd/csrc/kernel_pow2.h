// Hand-specialised power-of-two kernels and their registries (design notes: kernel_pow2_core.h).
#pragma once
#include "kernel_pow2_core.h"
#include "kernel_pow2_lean.h"

namespace vkfft_mi355x {

template <typename T, typename SCH, int FPW>
__global__ void __launch_bounds__(((1 << SCH::LOGN) >> SCH::LOGE) * FPW) pow2_row_kernel(const PassParams p) {
	constexpr int LOGN = SCH::LOGN, N = 1 << LOGN, LOGE = SCH::LOGE, E = 1 << LOGE, TPF = N / E;
	constexpr int LDSPF = SCH::NS > 1 ? N + (N >> LOGE) : 1;
	constexpr bool waveOnly = TPF <= 64; // an FFT never straddles wavefronts
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	__shared__ cx<T> lds[FPW * LDSPF];
	const uint32_t tid = threadIdx.x;
	const uint32_t f = tid / TPF, tau = tid % TPF;
	uint32_t wg = p.reverseTiles ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
	const uint32_t tile = wg % p.tilesPerG0;
	wg /= p.tilesPerG0;
	const uint32_t g1 = wg % p.dim[1].count, g2 = wg / p.dim[1].count;
	const uint32_t f0 = tile * FPW;
	const bool valid = f0 + f < p.dim[0].count;
	// wave-uniform tile bases + one 32-bit lane offset per side (memops.h)
	const GBuf gin = make_gbuf((const cx<T>*)p.in + ((int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride + (int64_t)f0 * p.dim[0].inStride));
	const GBuf gout = make_gbuf((cx<T>*)p.out + ((int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride + (int64_t)f0 * p.dim[0].outStride));
	const GBuf glut = make_gbuf(p.lut);
	const uint32_t laneIn = valid ? (f * (uint32_t)p.dim[0].inStride + tau) * ES : kGbInvalid;
	const uint32_t laneOut = valid ? (f * (uint32_t)p.dim[0].outStride + tau) * ES : kGbInvalid;
	cx<T> v[E];
	// Short rows (one to eight threads per FFT): a thread's own elements are TPF * 8 bytes apart and a wave's lanes N * 8 bytes, so that a
	// load instruction touches up to 64 different cache lines (measured 1.4 TB/s at N = 16).  When the FPW rows of the tile are dense in
	// memory they are moved as ONE contiguous run, 16 bytes per lane, and turned through LDS (odd pitch: conflict-free 8-byte accesses).
	constexpr bool STAGED = TPF <= 8;
	constexpr int SP = N + 1, NT = TPF * FPW, PER = 16 / (int)ES; // staging pitch; elements per 16-byte access
	__shared__ cx<T> stage[STAGED ? FPW * SP : 1];
	const bool denseIn = STAGED && p.dim[0].inStride == (int64_t)N && !p.padInN, denseOut = STAGED && p.dim[0].outStride == (int64_t)N && !p.padOutN;
	const uint32_t rowsHere = p.dim[0].count - f0 < (uint32_t)FPW ? p.dim[0].count - f0 : (uint32_t)FPW;
	if (denseIn) {
#pragma unroll
		for (int i = 0; i < E / PER; i++) {
			const uint32_t e0 = (tid + (uint32_t)i * NT) * PER, row = e0 / N, col = e0 % N; // PER consecutive points of one row (N is even)
			const uint32_t off = row < rowsHere ? e0 * ES : kGbInvalid;
			if constexpr (PER == 2) {
				cx<T> a, b;
				gb_load2_x<T, 0>(gin, off, 0, a, b);
				stage[row * SP + col] = a; stage[row * SP + col + 1] = b;
			} else stage[row * SP + col] = gb_load<T>(gin, off, 0);
		}
		VKFFT_SYNC();
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = stage[f * SP + tau + m * TPF];
	} else if (p.padInN) { // zero padding: points of the padded range get an out-of-range offset (they read as zero and are not fetched)
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = gb_load<T>(gin, (tau + (uint32_t)(m * TPF) - p.padInL < p.padInN) ? kGbInvalid : laneIn, (uint32_t)(m * TPF) * ES);
	} else {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = gb_load<T>(gin, laneIn, (uint32_t)(m * TPF) * ES);
	}
	if (p.swapIn) {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cswap(v[m]);
	}
	pow2_stages<T, SCH, 0, TPF, 0, TwGlobal<T>>(v, lds + f * LDSPF, TwGlobal<T>{glut}, tau, waveOnly);
	if (p.swapOut) {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cswap(v[m]);
	}
	const T sc = (T)p.scale;
	if (sc != (T)1) {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cscale(v[m], sc);
	}
	if (denseOut) {
		if (denseIn) VKFFT_SYNC(); // every thread has taken its inputs out of the staging tile
#pragma unroll
		for (int m = 0; m < E; m++) stage[f * SP + tau + m * TPF] = v[m];
		VKFFT_SYNC();
#pragma unroll
		for (int i = 0; i < E / PER; i++) {
			const uint32_t e0 = (tid + (uint32_t)i * NT) * PER, row = e0 / N, col = e0 % N;
			const uint32_t off = row < rowsHere ? e0 * ES : kGbInvalid;
			if constexpr (PER == 2) gb_store2_x<T, 0>(gout, off, stage[row * SP + col], stage[row * SP + col + 1]);
			else gb_store<T>(gout, off, 0, stage[row * SP + col]);
		}
	} else if (p.padOutN) { // (the padded range of the output is not written)
#pragma unroll
		for (int m = 0; m < E; m++) gb_store<T>(gout, (tau + (uint32_t)(m * TPF) - p.padOutL < p.padOutN) ? kGbInvalid : laneOut, (uint32_t)(m * TPF) * ES, v[m]);
	} else {
#pragma unroll
		for (int m = 0; m < E; m++) gb_store<T>(gout, laneOut, (uint32_t)(m * TPF) * ES, v[m]);
	}
}

// ---- fused Bluestein (chirp-z) rows of length n <= M/2 on a power-of-two padded length M ---------------------------
// x[j] conj(chirp[j]) zero-padded to M -> FFT_M -> * FFT(chirp)/M -> inverse FFT_M (swap identity) -> * conj(chirp[k]), k < n:
// the single-kernel form of the reference's vkFFT_Bluestein.h:32,201.  Built on the register-resident Stockham core above: the
// data never leaves registers between the two transforms, and because register m of a thread holds point tau + m*TPF on the
// way in, at the spectrum and on the way out, the chirp and FFT(chirp) values a thread needs are the same for every row —
// the workgroup is persistent (grid-stride over the row tiles) and keeps them in registers.
template <typename T, typename SCH, int FPW>
__global__ void __launch_bounds__(((1 << SCH::LOGN) >> SCH::LOGE) * FPW) pow2_blue_kernel(const PassParams p) {
	constexpr int LOGN = SCH::LOGN, M = 1 << LOGN, LOGE = SCH::LOGE, E = 1 << LOGE, TPF = M / E, EH = E / 2;
	constexpr int LDSPF = SCH::NS > 1 ? M + (M >> LOGE) : 1;
	constexpr bool waveOnly = TPF <= 64;
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	__shared__ cx<T> lds[FPW * LDSPF];
	const uint32_t tid = threadIdx.x;
	const uint32_t f = tid / TPF, tau = tid % TPF;
	const uint32_t n = p.opN;
	const GBuf glut = make_gbuf(p.lut), gch = make_gbuf(p.aux), gbh = make_gbuf(p.aux2);
	// (the tables of a thread's points stay in registers across the tiles — except in the 16384-point shape, 1024 threads at 128 registers, where they spilled:
	// 508 bytes of scratch per lane in round 5; there they are read again for every tile)
	constexpr bool TABREG = TPF * FPW < 1024;
	cx<T> ch[EH], bh[E];
	if constexpr (TABREG) {
#pragma unroll
		for (int m = 0; m < EH; m++) { const uint32_t pos = tau + m * TPF; ch[m] = gb_load<T>(gch, pos < n ? pos * ES : kGbInvalid, 0); }
#pragma unroll
		for (int m = 0; m < E; m++) bh[m] = gb_load<T>(gbh, (tau + m * TPF) * ES, 0);
	}
	auto chv = [&](int m) -> cx<T> { if constexpr (TABREG) return ch[m]; else { const uint32_t pos = tau + m * TPF; return gb_load<T>(gch, pos < n ? pos * ES : kGbInvalid, 0); } };
	auto bhv = [&](int m) -> cx<T> { if constexpr (TABREG) return bh[m]; else return gb_load<T>(gbh, (tau + m * TPF) * ES, 0); };
	// which of this thread's points are read / written: inside the sequence and outside the caller's zero-padded range (vkFFT_Zeropad.h:28).  The points do not
	// depend on the tile, so the tests are made once, one bit per point — as compares inside the loop the four range operands cost the 8192-point
	// instance 60 scalar-register spills (2049 ... 4096-point rows 11 % slower, profiles/r04b_sample1000_*)
	uint32_t rdMask = 0, wrMask = 0;
#pragma unroll
	for (int m = 0; m < EH; m++) {
		const uint32_t pos = tau + m * TPF;
		if (pos < n && !(pos - p.padInL < p.padInN)) rdMask |= 1u << m;
		if (pos < n && !(pos - p.padOutL < p.padOutN)) wrMask |= 1u << m;
	}
	const uint32_t tiles = p.tilesPerG0 * p.dim[1].count * p.dim[2].count;
	const T sc = (T)p.scale;
	for (uint32_t wgi = blockIdx.x; wgi < tiles; wgi += gridDim.x) {
		uint32_t wg = p.reverseTiles ? tiles - 1u - wgi : wgi;
		const uint32_t tile = wg % p.tilesPerG0;
		wg /= p.tilesPerG0;
		const uint32_t g1 = wg % p.dim[1].count, g2 = wg / p.dim[1].count;
		const uint32_t f0 = tile * FPW;
		const bool valid = f0 + f < p.dim[0].count;
		const GBuf gin = make_gbuf((const cx<T>*)p.in + ((int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride + (int64_t)f0 * p.dim[0].inStride));
		const GBuf gout = make_gbuf((cx<T>*)p.out + ((int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride + (int64_t)f0 * p.dim[0].outStride));
		const uint32_t laneIn = valid ? (f * (uint32_t)p.dim[0].inStride + tau) * ES : kGbInvalid;
		const uint32_t laneOut = valid ? (f * (uint32_t)p.dim[0].outStride + tau) * ES : kGbInvalid;
		cx<T> v[E];
#pragma unroll
		for (int m = 0; m < EH; m++) { // points >= n are the zero padding (n <= M/2: they include every m >= E/2)
			cx<T> x = gb_load<T>(gin, ((rdMask >> m) & 1u) ? laneIn : kGbInvalid, (uint32_t)(m * TPF) * ES);
			if (p.bluesteinSwapIn) x = cswap(x);
			v[m] = cmulc(x, chv(m));
		}
#pragma unroll
		for (int m = EH; m < E; m++) v[m] = cx<T>{(T)0, (T)0};
		pow2_stages<T, SCH, 0, TPF, 0, TwGlobal<T>>(v, lds + f * LDSPF, TwGlobal<T>{glut}, tau, waveOnly);
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cswap(cmul(v[m], bhv(m)));
		if constexpr (SCH::NS > 1) { if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC(); } // the exchange buffer is reused
		pow2_stages<T, SCH, 0, TPF, 0, TwGlobal<T>>(v, lds + f * LDSPF, TwGlobal<T>{glut}, tau, waveOnly);
#pragma unroll
		for (int m = 0; m < EH; m++) {
			cx<T> x = cmulc(cswap(v[m]), chv(m));
			if (p.bluesteinSwapOut) x = cswap(x);
			if (sc != (T)1) x = cscale(x, sc);
			gb_store<T>(gout, ((wrMask >> m) & 1u) ? laneOut : kGbInvalid, (uint32_t)(m * TPF) * ES, x);
		}
		if constexpr (SCH::NS > 1) { if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC(); }
	}
}

// Four-Step twiddle w^(k*col), k = tau + m*TPF, of a thread's E points (reference vkFFT_4step.h:31): exponent
// e_m = e_0 + m*D with D = TPF*col.  Instead of one table look-up (2 gathers of the two-level LUT) per element,
// 2*sqrt(E)-1 look-ups feed a two-factor product.
template <typename T, int LOGE, int TPF>
__device__ inline void pow2_col_twiddle(cx<T>* v, const PassParams& p, const uint32_t tau, const uint32_t colIdx) {
	constexpr int E = 1 << LOGE;
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	const GBuf gtab = make_gbuf(p.aux);
	const uint32_t loMask = (1u << p.fsLoBits) - 1u;
	const uint32_t hiBase = (loMask + 1u) * ES;
	auto tw = [&](uint32_t e) { return cmul(gb_load<T>(gtab, (e & loMask) * ES, 0), gb_load<T>(gtab, (e >> p.fsLoBits) * ES, hiBase)); };
	// m = (j << LOB) + i:  w^(e_0 + m*D) = A[j] * B[i],  A[j] = w^((tau + (j<<LOB)*TPF)*col),  B[i] = w^(i*TPF*col)
	constexpr int HIB = (LOGE + 1) / 2, LOB = LOGE - HIB;
	cx<T> A[1 << HIB], B[1 << LOB];
#pragma unroll
	for (int j = 0; j < (1 << HIB); j++) A[j] = tw((tau + (uint32_t)((j << LOB) * TPF)) * colIdx);
	B[0] = cx<T>{(T)1, (T)0};
#pragma unroll
	for (int i = 1; i < (1 << LOB); i++) B[i] = tw((uint32_t)(i * TPF) * colIdx);
	cx<T> wm[E];
#pragma unroll
	for (int m = 0; m < E; m++) wm[m] = (m & ((1 << LOB) - 1)) ? cmul(A[m >> LOB], B[m & ((1 << LOB) - 1)]) : A[m >> LOB];
#pragma unroll
	for (int m = 0; m < E; m++) v[m] = cmul(v[m], wm[m]);
}

// ---- strided-tile ("column") kernel: Four-Step passes and the non-unit-stride axes of 2D/3D transforms ----
// A workgroup transforms TC neighbouring columns; lanes run across the columns so that every global
// access is a TC*sizeof(complex) contiguous segment (256 B for TC=32 fp32).  Same register-resident
// Stockham core as the row kernel; the LDS exchange is [element][column] with an odd pitch, conflict-free
// in both directions.  Optional fused epilogue: Four-Step twiddle (two-level LUT) and a transposed store
// (each column written out as one contiguous run) for the first Four-Step pass.
// BIG: tiles whose rows are so far apart that the tile spans 2 GiB or more (the z axis of a 1024^3 volume on one GPU: 8 MiB x 1024 rows): 64-bit
// per-lane addresses instead of one buffer resource per tile (the reference switches its index type the same way, vkFFT_InitializeApp.h:1190-1221)
template <typename T, typename SCH, int TC, bool BIG = false>
__global__ void __launch_bounds__(((1 << SCH::LOGN) >> SCH::LOGE) * TC) pow2_col_kernel(const PassParams p) {
	constexpr int LOGN = SCH::LOGN, L = 1 << LOGN, LOGE = SCH::LOGE, E = 1 << LOGE, TPF = L / E;
	constexpr int TCP = TC + 1, NT = TPF * TC;
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	__shared__ cx<T> lds[L * TCP];
	const uint32_t tid = threadIdx.x;
	const uint32_t c = tid % TC, tau = tid / TC;
	uint32_t wg = p.reverseTiles ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
	const uint32_t tile = wg % p.tilesPerG0;
	wg /= p.tilesPerG0;
	// plain column passes of small systems: the tile index may run over dim[0] x dim[1] (PassParams::colMerge: a companion axis of 16 columns would leave
	// every 32-column tile half empty)
	const bool merge = !BIG && p.colMerge != 0;
	const uint32_t g1 = merge ? 0u : wg % p.dim[1].count, g2 = merge ? wg : wg / p.dim[1].count;
	const uint32_t col0 = tile * TC;
	bool valid = col0 + c < p.dim[0].count;
	int64_t inB = (int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride, outB = (int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride;
	uint32_t cIn = c * (uint32_t)p.dim[0].inStride, cOut = c * (uint32_t)p.dim[0].outStride; // this lane's column, in elements from the tile base
	if (merge) {
		const uint32_t c0 = p.dim[0].count, q0 = col0 / c0, g = col0 + c, q = g / c0, r = g - q * c0;
		valid = q < p.dim[1].count;
		inB += (int64_t)q0 * p.dim[1].inStride; outB += (int64_t)q0 * p.dim[1].outStride;
		cIn = (uint32_t)((int64_t)r * p.dim[0].inStride + (int64_t)(q - q0) * p.dim[1].inStride);
		cOut = (uint32_t)((int64_t)r * p.dim[0].outStride + (int64_t)(q - q0) * p.dim[1].outStride);
	} else { inB += (int64_t)col0 * p.dim[0].inStride; outB += (int64_t)col0 * p.dim[0].outStride; }
	// wave-uniform tile bases + one 32-bit lane offset per side, uniform step between a thread's elements (memops.h)
	const GBuf gin = make_gbuf((const cx<T>*)p.in + inB);
	const GBuf gout = make_gbuf((cx<T>*)p.out + outB);
	const GBuf glut = make_gbuf(p.lut);
	const uint32_t laneIn = valid ? (tau * (uint32_t)p.inStrideJ + cIn) * ES : kGbInvalid;
	const uint32_t stepIn = (uint32_t)(TPF * (uint32_t)p.inStrideJ) * ES;
	cx<T> v[E];
	if constexpr (BIG) {
		const cx<T>* pin = (const cx<T>*)p.in + (inB + (int64_t)tau * p.inStrideJ + (int64_t)c * p.dim[0].inStride);
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = (valid && !(tau + (uint32_t)(m * TPF) - p.padInL < p.padInN)) ? pin[(int64_t)(m * TPF) * p.inStrideJ] : cx<T>{(T)0, (T)0};
	} else if (p.padInN) { // zero padding along this axis: rows of the padded range are not fetched
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = gb_load<T>(gin, (tau + (uint32_t)(m * TPF) - p.padInL < p.padInN) ? kGbInvalid : laneIn, m * stepIn);
	} else {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = gb_load<T>(gin, laneIn, m * stepIn);
	}
	if (p.swapIn) {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cswap(v[m]);
	}
	pow2_stages<T, SCH, 0, TPF, TCP, TwGlobal<T>>(v, lds + c, TwGlobal<T>{glut}, tau, false);
	if (p.swapOut) {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cswap(v[m]);
	}
	if (p.postOp == OP_TWIDDLE_4STEP) {
		uint32_t colIdx, rr;
		if (p.fsColFromDim1) colIdx = g1; else p.fsColDiv.divmod(col0 + c, colIdx, rr);
		pow2_col_twiddle<T, LOGE, TPF>(v, p, tau, colIdx);
	}
	const T sc = (T)p.scale;
	if (sc != (T)1) {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cscale(v[m], sc);
	}
	if (p.colModeOut) {
		const uint32_t laneOut = valid ? (tau * (uint32_t)p.outStrideJ + cOut) * ES : kGbInvalid;
		const uint32_t stepOut = (uint32_t)(TPF * (uint32_t)p.outStrideJ) * ES;
		if constexpr (BIG) {
			cx<T>* pout = (cx<T>*)p.out + (outB + (int64_t)tau * p.outStrideJ + (int64_t)c * p.dim[0].outStride);
#pragma unroll
			for (int m = 0; m < E; m++) if (valid && !(tau + (uint32_t)(m * TPF) - p.padOutL < p.padOutN)) pout[(int64_t)(m * TPF) * p.outStrideJ] = v[m];
		} else if (p.padOutN) {
#pragma unroll
			for (int m = 0; m < E; m++) gb_store<T>(gout, (tau + (uint32_t)(m * TPF) - p.padOutL < p.padOutN) ? kGbInvalid : laneOut, m * stepOut, v[m]);
		} else {
#pragma unroll
			for (int m = 0; m < E; m++) gb_store<T>(gout, laneOut, m * stepOut, v[m]);
		}
	} else {
		// transposed store: column c becomes the contiguous run out[c*dim0.outStride + k*outStrideJ], lanes along k
		if constexpr (SCH::NS > 1) VKFFT_SYNC(); // the last exchange's reads are complete
#pragma unroll
		for (int m = 0; m < E; m++) lds[(tau + m * TPF) * TCP + c] = v[m];
		VKFFT_SYNC();
		const uint32_t nvalid = p.dim[0].count - col0 < (uint32_t)TC ? p.dim[0].count - col0 : (uint32_t)TC;
#pragma unroll
		for (int i = 0; i < E; i++) {
			const uint32_t idx = tid + i * NT;
			const uint32_t k = idx % L, cc = idx / L;
			if constexpr (BIG) { if (cc < nvalid) ((cx<T>*)p.out)[outB + (int64_t)cc * p.dim[0].outStride + (int64_t)k * p.outStrideJ] = lds[k * TCP + cc]; }
			else {
				const uint32_t off = cc < nvalid ? (cc * (uint32_t)p.dim[0].outStride + k * (uint32_t)p.outStrideJ) * ES : kGbInvalid;
				gb_store<T>(gout, off, 0, lds[k * TCP + cc]);
			}
		}
	}
}

// ---- multi-pass Bluestein on a power-of-two padded length M = n0*n1 (both factors column-kernel lengths) ---------------
// Three passes instead of the five of "Four-Step FFT_M, multiply, Four-Step inverse FFT_M" (cf. the reference's merged
// convolution kernel, vkFFT_Convolution.h:125): with the transposed scratch layout T[m][k0] of the first Four-Step pass, the
// second pass of the forward transform and the first pass of the inverse act on the SAME columns, so
//   MODE 1: x[n] conj(chirp[n]) (zero for n >= N) -> column FFT over j0 (n = m + j0*n1) -> twiddle -> T[m][k0]      (N in, M out)
//   MODE 2: column FFT over m -> * FFT(chirp)[k0 + n0*k1]/M -> inverse column FFT (swap identity) -> T[m][k0] in place (M, M)
//   MODE 3: rows T[m][.] -> conj twiddle -> inverse FFT over k0 -> * conj(chirp[n]), n = m + j0*n1 < N -> y[n]        (M in, N out)
// Three factors M = n0*n1*n2 (rows up to 2^29 points): five passes — MODE 1, the plain middle Four-Step pass (pow2_col_kernel),
// MODE 2 on the innermost factor, MODE 4 = the middle pass run backwards (conj twiddle, inverse column FFT, in place), MODE 3.
// Same register-resident column core as pow2_col_kernel; MODE 3 loads its tile transposed through LDS (rows are contiguous
// in T, lanes must run along m for the output side).
template <typename T, typename SCH, int TC, int MODE>
__global__ void __launch_bounds__(((1 << SCH::LOGN) >> SCH::LOGE) * TC) pow2_col_blue_kernel(const PassParams p) {
	constexpr int LOGN = SCH::LOGN, L = 1 << LOGN, LOGE = SCH::LOGE, E = 1 << LOGE, TPF = L / E;
	constexpr int TCP = TC + 1, NT = TPF * TC;
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	__shared__ cx<T> lds[L * TCP];
	const uint32_t tid = threadIdx.x;
	const uint32_t c = tid % TC, tau = tid / TC;
	uint32_t wg = p.reverseTiles ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
	const uint32_t tile = wg % p.tilesPerG0;
	wg /= p.tilesPerG0;
	const uint32_t g1 = wg % p.dim[1].count, g2 = wg / p.dim[1].count;
	const uint32_t col0 = tile * TC;
	const bool valid = col0 + c < p.dim[0].count;
	const uint32_t nvalid = p.dim[0].count - col0 < (uint32_t)TC ? p.dim[0].count - col0 : (uint32_t)TC;
	const int64_t inB = (int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride + (int64_t)col0 * p.dim[0].inStride;
	const int64_t outB = (int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride + (int64_t)col0 * p.dim[0].outStride;
	const GBuf gin = make_gbuf((const cx<T>*)p.in + inB);
	const GBuf gout = make_gbuf((cx<T>*)p.out + outB);
	const GBuf glut = make_gbuf(p.lut);
	const T sc = (T)p.scale;
	cx<T> v[E];
	if constexpr (MODE == 1) {
		// element j0 = tau + m*TPF of column (col0 + c) is point n = (col0 + c) + j0 * inStrideJ of the length-opN row
		const GBuf gch = make_gbuf(p.aux3);
		const uint32_t stride = (uint32_t)p.inStrideJ;
#pragma unroll
		for (int m = 0; m < E; m++) {
			const uint32_t n = col0 + c + (tau + m * TPF) * stride;
			const bool in = valid && n < p.opN;
			cx<T> x = gb_load<T>(gin, in ? (c + (tau + m * TPF) * stride) * ES : kGbInvalid, 0);
			if (p.bluesteinSwapIn) x = cswap(x);
			v[m] = cmulc(x, gb_load<T>(gch, in ? n * ES : kGbInvalid, 0));
		}
		pow2_stages<T, SCH, 0, TPF, TCP, TwGlobal<T>>(v, lds + c, TwGlobal<T>{glut}, tau, false);
		pow2_col_twiddle<T, LOGE, TPF>(v, p, tau, col0 + c);
		// transposed store: column c becomes the contiguous run T[col0 + c][.]
		if constexpr (SCH::NS > 1) VKFFT_SYNC();
#pragma unroll
		for (int m = 0; m < E; m++) lds[(tau + m * TPF) * TCP + c] = v[m];
		VKFFT_SYNC();
#pragma unroll
		for (int i = 0; i < E; i++) {
			const uint32_t idx = tid + i * NT;
			const uint32_t k = idx % L, cc = idx / L;
			gb_store<T>(gout, cc < nvalid ? (cc * (uint32_t)p.dim[0].outStride + k) * ES : kGbInvalid, 0, lds[k * TCP + cc]);
		}
	} else if constexpr (MODE == 2) {
		const uint32_t lane = valid ? (tau * (uint32_t)p.inStrideJ + c) * ES : kGbInvalid;
		const uint32_t step = (uint32_t)(TPF * (uint32_t)p.inStrideJ) * ES;
		// FFT(chirp)/M at the natural spectrum index of output k of column (g0, g1): k*opStrideJ + g0*opStride0 + g1*opStride1
		const GBuf gbh = make_gbuf(p.aux2);
		const uint32_t bhLane = valid ? (tau * p.opStrideJ + (col0 + c) * p.opStride0 + g1 * p.opStride1) * ES : kGbInvalid;
		const uint32_t bhStep = (uint32_t)TPF * p.opStrideJ * ES;
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = gb_load<T>(gin, lane, m * step);
		pow2_stages<T, SCH, 0, TPF, TCP, TwGlobal<T>>(v, lds + c, TwGlobal<T>{glut}, tau, false);
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cswap(cmul(v[m], gb_load<T>(gbh, bhLane, m * bhStep)));
		if constexpr (SCH::NS > 1) VKFFT_SYNC(); // the exchange buffer is reused
		pow2_stages<T, SCH, 0, TPF, TCP, TwGlobal<T>>(v, lds + c, TwGlobal<T>{glut}, tau, false);
#pragma unroll
		for (int m = 0; m < E; m++) gb_store<T>(gout, lane, m * step, cswap(v[m]));
	} else if constexpr (MODE == 5) {
		// ONE-pass Bluestein along a strided axis (prime x prime planes, the reference's sample 7): the column tile's rows j < opN are the
		// sequence, rows up to L = M are its zero padding; chirp multiply, FFT_M, * FFT(chirp)/M, inverse FFT_M (swap identity), chirp multiply —
		// everything in registers and one LDS tile, lanes along the TC neighbouring columns on both sides (pow2_blue_kernel is the row form)
		const GBuf gch = make_gbuf(p.aux), gbh = make_gbuf(p.aux2);
		const uint32_t n = p.opN;
		const uint32_t laneIn = valid ? (tau * (uint32_t)p.inStrideJ + c * (uint32_t)p.dim[0].inStride) * ES : kGbInvalid;
		const uint32_t stepIn = (uint32_t)(TPF * (uint32_t)p.inStrideJ) * ES;
#pragma unroll
		for (int m = 0; m < E / 2; m++) { // opN <= M/2: rows of the upper half are padding
			const uint32_t pos = tau + m * TPF;
			cx<T> x = gb_load<T>(gin, (pos < n && !(pos - p.padInL < p.padInN)) ? laneIn : kGbInvalid, m * stepIn);
			if (p.bluesteinSwapIn) x = cswap(x);
			v[m] = cmulc(x, gb_load<T>(gch, pos < n ? pos * ES : kGbInvalid, 0));
		}
#pragma unroll
		for (int m = E / 2; m < E; m++) v[m] = cx<T>{(T)0, (T)0};
		pow2_stages<T, SCH, 0, TPF, TCP, TwGlobal<T>>(v, lds + c, TwGlobal<T>{glut}, tau, false);
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cswap(cmul(v[m], gb_load<T>(gbh, (tau + m * TPF) * ES, 0)));
		if constexpr (SCH::NS > 1) VKFFT_SYNC(); // the exchange buffer is reused
		pow2_stages<T, SCH, 0, TPF, TCP, TwGlobal<T>>(v, lds + c, TwGlobal<T>{glut}, tau, false);
		const uint32_t laneOut = valid ? (tau * (uint32_t)p.outStrideJ + c * (uint32_t)p.dim[0].outStride) * ES : kGbInvalid;
		const uint32_t stepOut = (uint32_t)(TPF * (uint32_t)p.outStrideJ) * ES;
#pragma unroll
		for (int m = 0; m < E / 2; m++) {
			const uint32_t pos = tau + m * TPF;
			cx<T> y = cmulc(cswap(v[m]), gb_load<T>(gch, pos < n ? pos * ES : kGbInvalid, 0));
			if (p.bluesteinSwapOut) y = cswap(y);
			if (sc != (T)1) y = cscale(y, sc);
			gb_store<T>(gout, (pos < n && !(pos - p.padOutL < p.padOutN)) ? laneOut : kGbInvalid, m * stepOut, y);
		}
	} else if constexpr (MODE == 6 || MODE == 7) { // (7: the narrow-tile instance of 1024 points whose 512 threads have the registers for a kernel matrix)
		// Merged convolution along this (strided) axis — the reference's convolution-merged last axis (vkFFT_Convolution.h:125-447, vkFFT_RunApp.h:235-345):
		// column FFT of every coordinate system -> per frequency the kernel matrix times the vector of coordinates -> inverse column FFT of every
		// result (swap identity), in place.  One trip through memory instead of three (last forward pass, element-wise product, first inverse pass).
		constexpr int MAXM = 3;
		const uint32_t mm = p.convM, cf = p.convCf;
		const uint32_t lane = valid ? (tau * (uint32_t)p.inStrideJ + c * (uint32_t)p.dim[0].inStride) * ES : kGbInvalid;
		const uint32_t step = (uint32_t)(TPF * (uint32_t)p.inStrideJ) * ES;
		const GBuf gker = make_gbuf((const cx<T>*)p.aux2 + ((int64_t)g1 * p.convKerStride1 + (int64_t)g2 * p.convKerStride2 + (int64_t)col0));
		const uint32_t klane = valid ? (tau * (uint32_t)p.convKerStrideJ + c) * ES : kGbInvalid, kstep = (uint32_t)(TPF * (uint32_t)p.convKerStrideJ) * ES; // (unit stride along the tile)
		cx<T> w[MAXM][E];
#pragma unroll
		for (int l = 0; l < MAXM; l++) {
			if ((uint32_t)l < cf) {
				const GBuf gl = make_gbuf((const cx<T>*)p.in + (inB + (int64_t)l * p.convSysStride));
#pragma unroll
				for (int m = 0; m < E; m++) w[l][m] = gb_load<T>(gl, (tau + (uint32_t)(m * TPF) - p.padInL < p.padInN) ? kGbInvalid : lane, m * step);
				if constexpr (SCH::NS > 1) { if (l > 0) VKFFT_SYNC(); } // the exchange buffer is reused
				pow2_stages<T, SCH, 0, TPF, TCP, TwGlobal<T>>(w[l], lds + c, TwGlobal<T>{glut}, tau, false);
				if (p.convConj == 1) {
#pragma unroll
					for (int m = 0; m < E; m++) w[l][m] = cconj(w[l][m]);
				}
			}
		}
		const bool kconj = p.convConj == 2;
#pragma unroll
		for (int j = 0; j < MAXM; j++) {
			if ((uint32_t)j < cf) {
				cx<T> acc[E];
				if (mm <= 1) {
					const uint32_t ks = (uint32_t)((int64_t)j * p.convKerSysStride) * ES; // (the kernel systems lie within the 2 GiB span of the resource: planner)
#pragma unroll
					for (int m = 0; m < E; m++) { cx<T> k = gb_load<T>(gker, klane, m * kstep + ks); if (kconj) k = cconj(k); acc[m] = cmul(k, w[j][m]); }
				} else {
#pragma unroll
					for (int m = 0; m < E; m++) acc[m] = cx<T>{(T)0, (T)0};
#pragma unroll
					for (int l = 0; l < MAXM; l++) {
						if ((uint32_t)l < mm) {
							const uint32_t ks = (uint32_t)((int64_t)conv_kernel_index((uint32_t)j, (uint32_t)l, mm, p.convSymmetric != 0) * p.convKerSysStride) * ES;
#pragma unroll
							for (int m = 0; m < E; m++) { cx<T> k = gb_load<T>(gker, klane, m * kstep + ks); if (kconj) k = cconj(k); acc[m] = cadd(acc[m], cmul(k, w[l][m])); }
						}
					}
				}
#pragma unroll
				for (int m = 0; m < E; m++) acc[m] = cswap(acc[m]);
				if constexpr (SCH::NS > 1) VKFFT_SYNC();
				pow2_stages<T, SCH, 0, TPF, TCP, TwGlobal<T>>(acc, lds + c, TwGlobal<T>{glut}, tau, false);
				const GBuf go = make_gbuf((cx<T>*)p.out + (outB + (int64_t)j * p.convSysStride));
#pragma unroll
				for (int m = 0; m < E; m++) {
					cx<T> y = cswap(acc[m]);
					if (sc != (T)1) y = cscale(y, sc);
					gb_store<T>(go, (tau + (uint32_t)(m * TPF) - p.padOutL < p.padOutN) ? kGbInvalid : lane, m * step, y);
				}
			}
		}
	} else if constexpr (MODE == 8) {
		// the FIRST pass of a strided two-pass (Four-Step) transform run backwards, scratch -> data: conj twiddle w^(-k * column), inverse column FFT
		// (closes the merged convolution of a long strided axis: forward pass A, merged pass on the inner factor, this pass)
		const uint32_t laneI = valid ? (tau * (uint32_t)p.inStrideJ + c * (uint32_t)p.dim[0].inStride) * ES : kGbInvalid;
		const uint32_t stepI = (uint32_t)(TPF * (uint32_t)p.inStrideJ) * ES;
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cswap(gb_load<T>(gin, laneI, m * stepI));
		pow2_col_twiddle<T, LOGE, TPF>(v, p, tau, p.fsColFromDim1 ? g1 : col0 + c); // swap(u conj(w)) = swap(u) w
		pow2_stages<T, SCH, 0, TPF, TCP, TwGlobal<T>>(v, lds + c, TwGlobal<T>{glut}, tau, false);
		const uint32_t laneO = valid ? (tau * (uint32_t)p.outStrideJ + c * (uint32_t)p.dim[0].outStride) * ES : kGbInvalid;
		const uint32_t stepO = (uint32_t)(TPF * (uint32_t)p.outStrideJ) * ES;
#pragma unroll
		for (int m = 0; m < E; m++) {
			cx<T> y = cswap(v[m]);
			if (sc != (T)1) y = cscale(y, sc);
			gb_store<T>(gout, laneO, m * stepO, y);
		}
	} else if constexpr (MODE == 4) {
		// middle pass of a three-factor inverse run backwards, in place in the column layout: conj twiddle, inverse column FFT
		const uint32_t lane = valid ? (tau * (uint32_t)p.inStrideJ + c) * ES : kGbInvalid;
		const uint32_t step = (uint32_t)(TPF * (uint32_t)p.inStrideJ) * ES;
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cswap(gb_load<T>(gin, lane, m * step));
		uint32_t colIdx, rr;
		p.fsColDiv.divmod(col0 + c, colIdx, rr);
		pow2_col_twiddle<T, LOGE, TPF>(v, p, tau, colIdx); // swap(u conj(w)) = swap(u) w
		pow2_stages<T, SCH, 0, TPF, TCP, TwGlobal<T>>(v, lds + c, TwGlobal<T>{glut}, tau, false);
#pragma unroll
		for (int m = 0; m < E; m++) gb_store<T>(gout, lane, m * step, cswap(v[m]));
	} else {
		// rows (col0 + cc) of T are contiguous runs of L points: load them with lanes along the run, turn the tile in LDS
#pragma unroll
		for (int i = 0; i < E; i++) {
			const uint32_t idx = tid + i * NT;
			const uint32_t k = idx % L, cc = idx / L;
			lds[k * TCP + cc] = gb_load<T>(gin, cc < nvalid ? (cc * (uint32_t)p.dim[0].inStride + k) * ES : kGbInvalid, 0);
		}
		VKFFT_SYNC();
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cswap(lds[(tau + m * TPF) * TCP + c]);
		if constexpr (SCH::NS > 1) VKFFT_SYNC(); // the tile is in registers before the exchanges overwrite it
		pow2_col_twiddle<T, LOGE, TPF>(v, p, tau, col0 + c); // swap(u conj(w)) = swap(u) w: the forward table serves the inverse
		pow2_stages<T, SCH, 0, TPF, TCP, TwGlobal<T>>(v, lds + c, TwGlobal<T>{glut}, tau, false);
		const GBuf gch = make_gbuf(p.aux3);
		const uint32_t stride = (uint32_t)p.outStrideJ;
#pragma unroll
		for (int m = 0; m < E; m++) {
			const uint32_t n = col0 + c + (tau + m * TPF) * stride;
			const bool in = valid && n < p.opN;
			cx<T> y = cmulc(cswap(v[m]), gb_load<T>(gch, in ? n * ES : kGbInvalid, 0));
			if (p.bluesteinSwapOut) y = cswap(y);
			if (sc != (T)1) y = cscale(y, sc);
			gb_store<T>(gout, in ? (c + (tau + m * TPF) * stride) * ES : kGbInvalid, 0, y);
		}
	}
}

} // namespace vkfft_mi355x
