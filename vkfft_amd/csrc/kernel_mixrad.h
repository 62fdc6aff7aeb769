// Rader as a stage of a composite length, in ONE kernel: unit-stride rows of N = M * P points, P a prime with 13-smooth P - 1, M a small cofactor.
//
// The reference builds a Rader tree per prime factor and runs it as one stage among the radix stages of its generated kernel
// (vkFFT_Scheduler.h:1733-1873, 2304-2404; vkFFT_RaderKernels.h:30, 1278).  Until round 4 this library had Rader for PRIME lengths only
// (kernel_mixconv.h) and sent 2670 = 30 * 89, 3232 = 32 * 101, 889 = 7 * 127 ... whole through Bluestein on two to four times the points (0.4-0.6x the
// reference).  Here, Cooley-Tukey with n = M a + b, k = k2 + P k1:
//     X[k2 + P k1] = sum_b  W_M^(b k1) * W_N^(b k2) * Y_b[k2],     Y_b[k2] = sum_a x[M a + b] W_P^(a k2)
//   1. the row arrives in LDS sub-sequence-major (x[M a + b] at b*P + a): the M sub-sequences are contiguous runs of P points;
//   2. every sub-sequence is transformed IN PLACE by the Rader convolution of kernel_mixconv.h (gather through g^a, FFT of P - 1 points, times the kernel
//      spectrum, inverse FFT, scatter through g^-q) — the FPW thread groups of the workgroup each take one sub-sequence (of one of FPW / M rows when M is
//      small, in rounds when M exceeds FPW); the compile-time radix schedule and the padded exchange buffer are those of the prime's own instance;
//   3. a thread takes output column k2: M values (consecutive lanes read consecutive addresses), the twiddles W_N^(b k2) from a table, one M-point
//      butterfly in registers, M stores that are each coalesced along k2.
// One instance per prime (the Rader row instances of the mixconv tables), the cofactor is a run-time parameter: no instance per (M, P) pair.
#pragma once
#include "engine.h"
#include "butterflies.h"
#include "memops.h"
#include "mix_sched.h"
#include "mix_stage.h"
#include "mixrad_plan.h"
#include "kernel_generic.h"
#include "kernel_tmaps.h"

namespace vkfft_mi355x {

// M-point butterfly for the cofactors that have no dft<M> of their own: one Cooley-Tukey step A x B in registers, roots folded at compile time
template <int A, int B, typename T> __host__ __device__ inline void mixrad_dft_ab(cx<T>* v) {
	constexpr int R = A * B;
	cx<T> y[R];
#pragma unroll
	for (int n2 = 0; n2 < B; n2++) {
		cx<T> tmp[A];
#pragma unroll
		for (int n1 = 0; n1 < A; n1++) tmp[n1] = v[B * n1 + n2];
		dft<A, T>(tmp);
#pragma unroll
		for (int k1 = 0; k1 < A; k1++) {
			const int m = (n2 * k1) % R;
			if (m == 0) y[n2 * A + k1] = tmp[k1];
			else y[n2 * A + k1] = cmul(tmp[k1], cx<T>{(T)__builtin_cos(6.283185307179586476925286766559 * m / R), (T)(-__builtin_sin(6.283185307179586476925286766559 * m / R))});
		}
	}
#pragma unroll
	for (int k1 = 0; k1 < A; k1++) {
		cx<T> tmp[B];
#pragma unroll
		for (int n2 = 0; n2 < B; n2++) tmp[n2] = y[n2 * A + k1];
		dft<B, T>(tmp);
#pragma unroll
		for (int k2 = 0; k2 < B; k2++) v[k1 + A * k2] = tmp[k2];
	}
}
template <int M, typename T> __host__ __device__ inline void mixrad_dft(cx<T>* v) {
	if constexpr (M == 18) mixrad_dft_ab<2, 9, T>(v);
	else if constexpr (M == 20) mixrad_dft_ab<4, 5, T>(v);
	else if constexpr (M == 21) mixrad_dft_ab<3, 7, T>(v);
	else if constexpr (M == 24) mixrad_dft_ab<8, 3, T>(v);
	else if constexpr (M == 27) mixrad_dft_ab<3, 9, T>(v);
	else if constexpr (M == 28) mixrad_dft_ab<4, 7, T>(v);
	else if constexpr (M == 30) mixrad_dft_ab<2, 15, T>(v);
	else dft<M, T>(v); // 2 ... 10, 12, 14, 15, 16, 25, 32
}
// step 3 for a compile-time cofactor
// Source: sub-sequence b of row r at rowbuf + r * rowPitch + b * SUBS (SUBS = P: the tile of whole rows; SUBS = the buffer pitch of a thread group: every group
// holds one sub-sequence, whose bin 0 lives in dc[r * M + b] as in the prime's own Rader kernel).
// toLds != nullptr (real transforms between the generic maps): the columns go to a second row region in natural order instead of global memory
template <typename T, int M, int P, int NT, int SUBS>
__device__ inline void mixrad_columns(const cx<T>* rowbuf, const uint32_t rowPitch, const cx<T>* dc, const uint32_t N, const uint32_t rowsHere, const GBuf gout, const GBuf gtw,
                                      const uint32_t outRowBytes, const bool swO, const T sc, const uint32_t tid, cx<T>* toLds) {
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	const uint32_t total = rowsHere * (uint32_t)P;
	for (uint32_t j = tid; j < total; j += (uint32_t)NT) {
		const uint32_t r = j / (uint32_t)P, k2 = j % (uint32_t)P;
		const cx<T>* const src = rowbuf + r * rowPitch + k2;
		cx<T> y[M];
#pragma unroll
		for (int b = 0; b < M; b++) y[b] = src[b * SUBS];
		if (dc && k2 == 0u) {
#pragma unroll
			for (int b = 0; b < M; b++) y[b] = dc[r * (uint32_t)M + b];
		}
		constexpr int TWG = 8; // twiddles in flight at a time (the butterfly of a large cofactor needs the registers)
#pragma unroll
		for (int b0 = 1; b0 < M; b0 += TWG) {
			cx<T> w[TWG];
#pragma unroll
			for (int b = b0; b < b0 + TWG && b < M; b++) w[b - b0] = gb_load<T>(gtw, k2 * ES, (uint32_t)((b - 1) * P) * ES);
			if constexpr (M > 10) VKFFT_SCHED_FENCE();
#pragma unroll
			for (int b = b0; b < b0 + TWG && b < M; b++) { y[b] = cmul(y[b], w[b - b0]); if constexpr (M > 10) { VKFFT_PIN(y[b].x); VKFFT_PIN(y[b].y); } }
			if constexpr (M > 10) VKFFT_SCHED_FENCE();
		}
		mixrad_dft<M, T>(y);
		if (toLds) {
#pragma unroll
			for (int k1 = 0; k1 < M; k1++) toLds[r * N + k2 + (uint32_t)(k1 * P)] = y[k1];
			continue;
		}
		const uint32_t o = r * outRowBytes + k2 * ES;
#pragma unroll
		for (int k1 = 0; k1 < M; k1++) {
			cx<T> v = swO ? cswap(y[k1]) : y[k1];
			if (sc != (T)1) v = cscale(v, sc);
			gb_store<T>(gout, o, (uint32_t)(k1 * P) * ES, v);
		}
	}
}

// lut = stage twiddles of SCH (length P - 1); rader = uint32 g^a mod P (a < L) followed by g^-k mod P; aux2 = FFT of the Rader kernel / L (L entries)
// followed by the column twiddles W_N^(b k2), (b - 1) * P + k2, b = 1 ... M - 1; raderM = M.
// Tiles: forceT = rows per workgroup = mixrad_rows(2, ...): as many rows as the tile's LDS holds (the thread groups take their sub-sequences in rounds).
template <typename T, typename SCH, int TPF, int FPW>
__global__ void __launch_bounds__(TPF * FPW) mixrad_kernel(const PassParams p) {
	constexpr int L = SCH::N, P = L + 1, NT = TPF * FPW;
	constexpr int EXPF = SCH::NS > 1 ? MixPad<SCH, TPF, (int)sizeof(cx<T>)>::elems() : 1;
	constexpr int EX = (EXPF > L ? EXPF : L) | 1;                     // per thread group: exchange buffer of the stages = carrier of the spectrum between the two transforms
	constexpr int ROWN = (int)mixrad_row_elems(P, FPW); // rows of the tile, sub-sequence-major
	constexpr bool waveOnly = (TPF <= 64) && (64 % TPF == 0);
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	static_assert((size_t)(FPW * EX + ROWN) * sizeof(cx<T>) <= 160 * 1024, "LDS");
	__shared__ cx<T> exb[FPW * EX];
	__shared__ cx<T> rowbuf[ROWN];
	const uint32_t tid = threadIdx.x;
	const uint32_t f = tid / TPF, tau = tid % TPF;
	const uint32_t M = p.raderM, N = M * (uint32_t)P;
	// real transforms (R2C / C2R / DCT / DST whose complex length is M * P): the row enters through the generic pre-map and leaves through the generic
	// post-map (kernel_generic.h: ops_rows_in / ops_rows_out, the operation hoisted out of their loops); the kernel spectrum then comes from aux3
	FastDiv divN, divM; // (by the row length and by the cofactor: run-time values of this kernel, not the pass's own dividers, which the generic maps use)
	divN.d = N; divN.rcp = 1.0f / (float)N; divM.d = M; divM.rcp = 1.0f / (float)M;
	const bool ops = p.preOp != OP_NONE || p.postOp != OP_NONE;
	const uint32_t RW = mixrad_rows(2, P, FPW, M, ops);
	uint32_t wg = p.reverseTiles ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
	const uint32_t tile = wg % p.tilesPerG0;
	wg /= p.tilesPerG0;
	const uint32_t g1 = wg % p.dim[1].count, g2 = wg / p.dim[1].count;
	const uint32_t rowMult = (ops && p.pairRows) ? 2u : 1u; // two real rows per complex row of the tile (kernel_generic.h)
	const uint32_t f0 = tile * RW * rowMult;
	const uint32_t realRowsHere = p.dim[0].count - f0 < RW * rowMult ? p.dim[0].count - f0 : RW * rowMult;
	const uint32_t rowsHere = (realRowsHere + rowMult - 1u) / rowMult; // complex rows of the tile
	const GBuf gin = make_gbuf((const cx<T>*)p.in + ((int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride + (int64_t)f0 * p.dim[0].inStride));
	const GBuf gout = make_gbuf((cx<T>*)p.out + ((int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride + (int64_t)f0 * p.dim[0].outStride));
	const GBuf glut = make_gbuf(p.lut), gbh = make_gbuf(ops ? p.aux3 : p.aux2), gtw = make_gbuf((const cx<T>*)(ops ? p.aux3 : p.aux2) + L);
	const bool swI = p.bluesteinSwapIn != 0, swO = p.bluesteinSwapOut != 0;
	const int64_t rowIn0 = (int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride + (int64_t)f0 * p.dim[0].inStride;
	const int64_t rowOut0 = (int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride + (int64_t)f0 * p.dim[0].outStride;
	const uint32_t nat0 = f0 * p.opStride0 + g1 * p.opStride1;
	// ---- 1. rows -> LDS, sub-sequence-major
	if (ops && (p.tmPreFlags & kTmOn)) { // table-driven pre-map (kernel_tmaps.h)
		const GBuf gdi = make_gbuf((const char*)p.in + rowIn0 * (int64_t)p.inElemBytes);
		const uint32_t pitch = (uint32_t)p.dim[0].inStride * p.inElemBytes;
		auto put = [&](uint32_t r, uint32_t pos, cx<T> z) { uint32_t a, b; divM.divmod(pos, a, b); rowbuf[r * N + b * (uint32_t)P + a] = z; };
		if (p.tmPreFlags & kTmTwo) tm_rows_in<T, true>(p.tmPre, gdi, N, divN, rowsHere, realRowsHere, rowMult, pitch, p.swapIn != 0, tid, (uint32_t)NT, put);
		else tm_rows_in<T, false>(p.tmPre, gdi, N, divN, rowsHere, realRowsHere, rowMult, pitch, p.swapIn != 0, tid, (uint32_t)NT, put);
	} else if (ops) {
		dispatch_pre_op(p.preOp, [&](auto opc) { ops_rows_in<T>(p, opc, divN, rowbuf, N, rowsHere * N, realRowsHere, rowIn0, nat0, M, (uint32_t)P); });
	} else {
		const uint32_t inRowBytes = (uint32_t)p.dim[0].inStride * ES;
		for (uint32_t e = tid; e < rowsHere * N; e += (uint32_t)NT) {
			uint32_t r, n, a, b;
			divN.divmod(e, r, n);
			divM.divmod(n, a, b);
			const cx<T> v = gb_load<T>(gin, r * inRowBytes + n * ES, 0);
			rowbuf[r * N + b * (uint32_t)P + a] = swI ? cswap(v) : v;
		}
	}
	VKFFT_SYNC();
	// ---- 2. Rader convolution of every sub-sequence, in place
	{
		const uint32_t* const gp = (const uint32_t*)p.rader;
		cx<T>* const ex = exb + f * EX;
		const uint32_t jobs = rowsHere * M;
		auto fsync = [&]() { if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC(); };
		for (uint32_t job = f; job < ((jobs + (uint32_t)FPW - 1u) / (uint32_t)FPW) * (uint32_t)FPW; job += (uint32_t)FPW) { // (every group runs every round: the barriers are the workgroup's)
			const bool live = job < jobs;
			uint32_t r = 0, b = 0;
			if (live) divM.divmod(job, r, b);
			cx<T>* const seq = rowbuf + r * N + b * (uint32_t)P;
			const cx<T> x0 = seq[0];
			// forward transform of x[g^a]; spectrum * FFT(w^(g^-q)) / L, + x0 on the zero frequency (= x0 added to every output); X[0] = x0 + sum of the others
			mc_stage<T, SCH, 0, TPF, 1, true, false, true>(ex, glut, tau, waveOnly, [&](uint32_t t, uint32_t c) -> cx<T> { return seq[gp[t + c]]; },
			                                                [&](uint32_t t, uint32_t c, cx<T> v) {
				                                                const uint32_t k = t + c;
				                                                cx<T> w = cmul(v, gb_load<T>(gbh, t * ES, c * ES));
				                                                if (k == 0u) { if (live) seq[0] = cadd(x0, v); w = cadd(w, x0); }
				                                                ex[k] = cswap(w);
			                                                });
			fsync();
			// inverse transform; result q belongs to output index g^-q
			mc_stage<T, SCH, 0, TPF, 1, true, true, false>(ex, glut, tau, waveOnly, [&](uint32_t t, uint32_t c) -> cx<T> { return ex[t + c]; },
			                                                [&](uint32_t t, uint32_t c, cx<T> v) { if (live) seq[gp[(uint32_t)L + t + c]] = cswap(v); });
			fsync(); // the exchange buffer is free for the next round
		}
	}
	VKFFT_SYNC();
	// ---- 3. column twiddle, M-point butterfly, coalesced stores
	const uint32_t outRowBytes = (uint32_t)p.dim[0].outStride * ES;
	const T sc = (T)p.scale;
	cx<T>* const natural = ops ? rowbuf + ROWN / 2 : nullptr; // (the rows of an OPS tile fill at most half the region: mixrad_rows)
#define VKFFT_MIXRAD_CASE(m) case m: mixrad_columns<T, m, P, NT, P>(rowbuf, N, (const cx<T>*)nullptr, N, rowsHere, gout, gtw, outRowBytes, swO, sc, tid, natural); break;
	switch (M) {
	VKFFT_MIXRAD_CASE(2) VKFFT_MIXRAD_CASE(3) VKFFT_MIXRAD_CASE(4) VKFFT_MIXRAD_CASE(5) VKFFT_MIXRAD_CASE(6) VKFFT_MIXRAD_CASE(7) VKFFT_MIXRAD_CASE(8)
	VKFFT_MIXRAD_CASE(9) VKFFT_MIXRAD_CASE(10) VKFFT_MIXRAD_CASE(12) VKFFT_MIXRAD_CASE(14) VKFFT_MIXRAD_CASE(15) VKFFT_MIXRAD_CASE(16) VKFFT_MIXRAD_CASE(18)
	VKFFT_MIXRAD_CASE(20) VKFFT_MIXRAD_CASE(21) VKFFT_MIXRAD_CASE(24) VKFFT_MIXRAD_CASE(25) VKFFT_MIXRAD_CASE(27) VKFFT_MIXRAD_CASE(28) VKFFT_MIXRAD_CASE(30)
	VKFFT_MIXRAD_CASE(32)
	default: break;
	}
#undef VKFFT_MIXRAD_CASE
	if (ops) {
		VKFFT_SYNC();
		if (p.tmPostFlags & kTmOn) {
			const bool swOut = p.swapOut != 0;
			tm_rows_out<T>(p.tmPost, make_gbuf((char*)p.out + rowOut0 * (int64_t)p.outElemBytes), N, rowsHere, realRowsHere, rowMult, (uint32_t)p.dim[0].outStride * p.outElemBytes, p.tmPostFlags,
			               tid, (uint32_t)NT, [&](uint32_t r, uint32_t a) -> cx<T> { const cx<T> v = natural[r * N + a]; return swOut ? cswap(v) : v; });
		} else
		dispatch_post_op(p.postOp, [&](auto opc) { ops_rows_out<T>(p, opc, natural, (const cx<T>*)nullptr, N, RW, realRowsHere, rowOut0, nat0, N); });
	}
}
// ---- cofactors up to 10 that fit the thread groups of the prime's instance, complex rows: every thread group owns ONE sub-sequence in ONE buffer — exchange
// buffer of the stages, carrier of the spectrum and the sub-sequence itself, exactly as a row of the prime's own Rader kernel (kernel_mixconv.h) — so the LDS
// and the occupancy are those of that kernel (a separate tile of rows halves them: 74 = 2 * 37 ran at 2.8 TB/s against 4.3 for the prime itself).
// forceT = rows per workgroup = FPW / M.
template <typename T, typename SCH, int TPF, int FPW>
__global__ void __launch_bounds__(TPF * FPW) mixrad_small_kernel(const PassParams p) {
	constexpr int L = SCH::N, P = L + 1, NT = TPF * FPW;
	constexpr int EXPF = SCH::NS > 1 ? MixPad<SCH, TPF, (int)sizeof(cx<T>)>::elems() : 1;
	constexpr int SP = (EXPF > P ? EXPF : P) | 1;
	constexpr bool waveOnly = (TPF <= 64) && (64 % TPF == 0);
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	__shared__ cx<T> lds[FPW * SP];
	__shared__ cx<T> sDc[FPW];
	const uint32_t tid = threadIdx.x;
	const uint32_t f = tid / TPF, tau = tid % TPF;
	const uint32_t M = p.raderM, N = M * (uint32_t)P;
	FastDiv divN, divM;
	divN.d = N; divN.rcp = 1.0f / (float)N; divM.d = M; divM.rcp = 1.0f / (float)M;
	const uint32_t RW = (uint32_t)FPW / M;
	uint32_t wg = p.reverseTiles ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
	const uint32_t tile = wg % p.tilesPerG0;
	wg /= p.tilesPerG0;
	const uint32_t g1 = wg % p.dim[1].count, g2 = wg / p.dim[1].count;
	const uint32_t f0 = tile * RW;
	const uint32_t rowsHere = p.dim[0].count - f0 < RW ? p.dim[0].count - f0 : RW;
	const GBuf gin = make_gbuf((const cx<T>*)p.in + ((int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride + (int64_t)f0 * p.dim[0].inStride));
	const GBuf gout = make_gbuf((cx<T>*)p.out + ((int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride + (int64_t)f0 * p.dim[0].outStride));
	const GBuf glut = make_gbuf(p.lut), gbh = make_gbuf(p.aux2), gtw = make_gbuf((const cx<T>*)p.aux2 + L);
	const bool swI = p.bluesteinSwapIn != 0, swO = p.bluesteinSwapOut != 0;
	// ---- 1. rows -> the groups' buffers: element M a + b of row r is element a of group r * M + b
	{
		const uint32_t inRowBytes = (uint32_t)p.dim[0].inStride * ES;
		for (uint32_t e = tid; e < rowsHere * N; e += (uint32_t)NT) {
			uint32_t r, n, a, b;
			divN.divmod(e, r, n);
			divM.divmod(n, a, b);
			const cx<T> v = gb_load<T>(gin, r * inRowBytes + n * ES, 0);
			lds[(r * M + b) * (uint32_t)SP + a] = swI ? cswap(v) : v;
		}
	}
	VKFFT_SYNC();
	// ---- 2. the Rader convolution of the group's sub-sequence, in its buffer (the flow of mixconv_kernel<RADER = 1>)
	{
		const uint32_t* const gp = (const uint32_t*)p.rader;
		cx<T>* const row = lds + f * SP;
		const bool live = f < rowsHere * M;
		const cx<T> x0 = row[0];
		auto fsync = [&]() { if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC(); };
		mc_stage<T, SCH, 0, TPF, 1, true, true, true>(row, glut, tau, waveOnly, [&](uint32_t t, uint32_t c) -> cx<T> { return live ? row[gp[t + c]] : cx<T>{(T)0, (T)0}; },
		                                               [&](uint32_t t, uint32_t c, cx<T> v) {
			                                               const uint32_t k = t + c;
			                                               cx<T> w = cmul(v, gb_load<T>(gbh, t * ES, c * ES));
			                                               if (k == 0u) { sDc[f] = cadd(x0, v); w = cadd(w, x0); } // X[0] = x0 + sum of the others
			                                               row[k] = cswap(w);
		                                               });
		fsync();
		mc_stage<T, SCH, 0, TPF, 1, true, true, true>(row, glut, tau, waveOnly, [&](uint32_t t, uint32_t c) -> cx<T> { return row[t + c]; },
		                                               [&](uint32_t t, uint32_t c, cx<T> v) { row[gp[(uint32_t)L + t + c]] = cswap(v); });
	}
	VKFFT_SYNC();
	// ---- 3. column twiddle, M-point butterfly, coalesced stores
	const uint32_t outRowBytes = (uint32_t)p.dim[0].outStride * ES;
	const T sc = (T)p.scale;
#define VKFFT_MIXRAD_CASE(m) case m: mixrad_columns<T, m, P, NT, SP>(lds, M * (uint32_t)SP, sDc, N, rowsHere, gout, gtw, outRowBytes, swO, sc, tid, (cx<T>*)nullptr); break;
	switch (M) {
	VKFFT_MIXRAD_CASE(2) VKFFT_MIXRAD_CASE(3) VKFFT_MIXRAD_CASE(4) VKFFT_MIXRAD_CASE(5) VKFFT_MIXRAD_CASE(6) VKFFT_MIXRAD_CASE(7) VKFFT_MIXRAD_CASE(8)
	VKFFT_MIXRAD_CASE(9) VKFFT_MIXRAD_CASE(10)
	default: break;
	}
#undef VKFFT_MIXRAD_CASE
}

template <typename T, typename SCH, int TPF, int FPW> void mixrad_launch(const PassParams& prm, dim3 grid, hipStream_t s) {
	constexpr int P = SCH::N + 1;
	const bool ops = prm.preOp != OP_NONE || prm.postOp != OP_NONE;
	const int mode = mixrad_mode((uint32_t)P, (uint32_t)FPW, prm.raderM, ops);
	if (mode == 1) hipLaunchKernelGGL((mixrad_small_kernel<T, SCH, TPF, FPW>), grid, dim3(TPF * FPW), 0, s, prm);
	else if constexpr (12 * P <= (int)kMixradLongest) { if (mode == 2) hipLaunchKernelGGL((mixrad_kernel<T, SCH, TPF, FPW>), grid, dim3(TPF * FPW), 0, s, prm); }
}
// the composite forms exist for the fp32 Rader ROW instances whose prime leaves room for a cofactor (2 P <= the longest row)
template <typename T, typename SCH, int TPF, int FPW, int RADER, int COL> constexpr auto mixrad_ptr() -> void (*)(const PassParams&, dim3, hipStream_t) {
	constexpr int P = SCH::N + 1;
	if constexpr (RADER != 0 && COL == 0 && sizeof(T) == 4 && 2 * P <= (int)kMixradLongest) return &mixrad_launch<T, SCH, TPF, FPW>;
	else return nullptr;
}

} // namespace vkfft_mi355x
