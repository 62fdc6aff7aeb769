// Rader as a stage of a composite length, in ONE kernel: unit-stride rows of N = M * P points, P a prime with a smooth P - 1, M = A * B any cofactor whose prime
// factors are at most 31 (complex rows, and the real transforms whose complex length is such a composite between the table-driven maps of kernel_tmaps.h).
//
// The reference builds a Rader tree per prime factor and runs it as one stage among the radix stages of its generated kernel
// (vkFFT_Scheduler.h:1733-1873, 2304-2404; vkFFT_RaderKernels.h:30, 1278).  Here, Cooley-Tukey with n = M a + b, k = k2 + P k1:
//     X[k2 + P k1] = sum_b  W_M^(b k1) * W_N^(b k2) * Y_b[k2],     Y_b[k2] = sum_a x[M a + b] W_P^(a k2)
//   1. the rows of a tile arrive in LDS sub-sequence-major: sub-sequence b of row r is buffer r * M + b (pitch SP: the padded exchange buffer of the prime's
//      convolution, exactly a row of the prime's own Rader kernel, kernel_mixconv.h); with them arrive the TABLES of the convolution — stage twiddles, kernel
//      spectrum, the two generator permutations — and of the column steps, so that between the load of the rows and the store of the results no phase waits for
//      memory (round 5 kept them in global memory: five or six dependent trips to L2 per tile on one or two workgroups per CU were the whole run time);
//   2. every buffer is transformed IN PLACE by the Rader convolution (gather through g^a, FFT of P - 1 points, times the kernel spectrum, inverse FFT, scatter
//      through g^-q): the FPW thread groups of the prime's instance take the R * M buffers in rounds;
//   3. the M-point transforms along b, P of them per row, in place as one or two column steps M = A * B (decimation in frequency: b = a' B + b',
//      k1 = k1a + A k1b): step one — radix A over a', times W_M^(b' k1a) — and step two — radix B over b' — each one butterfly per thread in registers with the
//      lanes along k2 (consecutive LDS addresses).  The twiddle W_N^(b k2) rides on the first step's loads as a product of two table entries (b k2 < N:
//      high and low six bits).  Step two stores to memory (B runs of P points per butterfly) or, for the real transforms, back in place, from where the
//      table-driven post-map takes natural index n at buffer (k1 mod A) * B + k1 / A, offset k2.
// One instance per prime (the Rader row instances of the mixconv tables): the cofactor and its split are run-time parameters.  M = 1 — a row of P points — runs here
// too (no column step: VKFFT_MI355X_MIXRAD_PRIMES=1; measured slower than kernel_mixconv.h, which stays the default for a prime's own rows).
// P * P (1369 = 37 * 37, 3721 = 61 * 61): the column transform is the same prime — the same convolution, its thread groups along the columns (element pitch SP).
#pragma once
#include "engine.h"
#include "butterflies.h"
#include "memops.h"
#include "mix_sched.h"
#include "mix_stage.h"
#include "mixrad_plan.h"
#include "kernel_tmaps.h"

namespace vkfft_mi355x {

template <typename SCH> __host__ __device__ constexpr int mixrad_lut_elems() { // entries of the prime's stage-twiddle table (planner.cpp finish_pass: stages 1 ... NS - 1)
	int n = 0;
	for (int j = 1; j < SCH::NS; j++) n += (SCH::rad[j] - 1) * SCH::S(j);
	return n;
}
template <typename T, typename SCH, int TPF, int FPW = 1> struct MixradGeom {
	static constexpr int L = SCH::N, P = L + 1;
	// Thread groups of the convolution (TPF threads each), two layouts chosen per plan (PassParams::raderAligned, planner.cpp mixrad_choose):
	//   dense    — group f = threads [f TPF, (f + 1) TPF): FPW groups, the stages of the convolution behind workgroup barriers unless TPF divides 64;
	//   aligned  — no group straddles two wavefronts (64 / TPF groups per wavefront, the last lanes idle): the stages order their LDS traffic inside the
	//              wavefront, wavefronts run ahead of each other through the rounds (first version of round 6, dense only: 42 % of the wave cycles parked at
	//              barriers and waits for TPF = 10, 11, 13, 14, 15, profiles/r06_rader_stage_sq_counters.txt; aligned 3144 = 24 * 131 1.80 -> 2.75 TB/s).
	// The aligned layout has WAVES * (64 / TPF) groups — fewer than FPW for some primes (101: 30 instead of 32), which costs a second round where the cofactor was
	// matched to FPW (3232 = 32 * 101: 2.35 -> 2.21 TB/s): hence the choice per plan.  Offered where whole groups fill four fifths of a wavefront or more.
	static constexpr int GPW = TPF <= 64 ? 64 / TPF : 0;
	// Whole wavefronts, rounded to the NEAREST count: 260 threads (13 x 20, 26 x 10) run as 256 — four wavefronts, five workgroups per CU at 86-94 registers — not as 320
	// (3144 = 24 * 131 2.75 against 2.10 TB/s, 314 3.26 against 2.51); the dense layout then has one group less (19 of 20)
	static constexpr int WAVES = (TPF * FPW + 32) / 64 > 0 ? (TPF * FPW + 32) / 64 : 1;
	static constexpr int NT = WAVES * 64;                                   // threads of a workgroup
	static constexpr int GROUPS_DENSE = NT / TPF < FPW ? NT / TPF : FPW;
	static constexpr int GROUPS_ALIGNED = (TPF <= 64 && GPW * TPF * 5 >= 64 * 4) ? WAVES * GPW : 0;
	static constexpr int EXPF = SCH::NS > 1 ? MixPad<SCH, TPF, (int)sizeof(cx<T>)>::elems() : 1;
	static constexpr int SP = (EXPF > P ? EXPF : P) | 1; // buffer pitch: odd (consecutive buffers start on different banks)
	static constexpr int LUTN = mixrad_lut_elems<SCH>();
};

// one transform of the prime's convolution: mc_stage (mix_stage.h) with the stage twiddles in LDS
// LS / PADDED: a row buffer is dense with the per-exchange padding of MixPad; a column of the tile (P * P rows) has element pitch LS = SP and no padding.
// live = false: a thread group without a job in this round — it runs along (the barriers are the workgroup's) and writes nothing
template <typename T, typename SCH, int SI, int TPF, int LS, bool PADDED, typename IN, typename OUT>
__device__ inline void mixrad_stage(cx<T>* ldsf, const cx<T>* lut, const uint32_t tau, const bool waveOnly, const bool live, const IN& in, const OUT& out) {
	constexpr int N = SCH::N, R = SCH::rad[SI], NB = N / R, PB = (NB + TPF - 1) / TPF, S = SCH::S(SI);
	constexpr bool first = SI == 0, last = SI == SCH::NS - 1;
	using PAD = MixPad<SCH, TPF, (int)sizeof(cx<T>)>;
	cx<T> x[PB][R];
#pragma unroll
	for (int b = 0; b < PB; b++) {
		const uint32_t t = tau + b * TPF;
		if ((b + 1) * TPF <= NB || t < (uint32_t)NB) {
#pragma unroll
			for (int i = 0; i < R; i++) {
				if constexpr (first) x[b][i] = in(t, (uint32_t)(i * NB));
				else x[b][i] = ldsf[mix_slot<PADDED ? PAD::shift(SI - 1) : 0>(t + i * NB) * LS];
			}
		}
	}
	// every input is in registers before the buffer is overwritten (in() reads the buffer, out() writes it)
	if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC();
#pragma unroll
	for (int b = 0; b < PB; b++) {
		const uint32_t t = tau + b * TPF;
		if ((b + 1) * TPF <= NB || t < (uint32_t)NB) {
			const uint32_t s = t % (uint32_t)S;
			if constexpr (!first) {
				constexpr int LO = SCH::lutOff(SI);
#pragma unroll
				for (int i = 1; i < R; i++) x[b][i] = cmul(x[b][i], lut[s + (uint32_t)(LO + (i - 1) * S)]);
			}
			dft<R, T>(x[b]);
			if constexpr (last) {
#pragma unroll
				for (int k = 0; k < R; k++) out(t, (uint32_t)(k * S), x[b][k]); // last stage: s = t
			} else {
				const uint32_t ob = (t - s) * (uint32_t)R + s;
#pragma unroll
				for (int k = 0; k < R; k++) if (live) ldsf[mix_slot<PADDED ? PAD::shift(SI) : 0>(ob + k * S) * LS] = x[b][k];
			}
		}
	}
	if constexpr (!last) {
		if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC();
		mixrad_stage<T, SCH, SI + 1 < SCH::NS ? SI + 1 : SI, TPF, LS, PADDED>(ldsf, lut, tau, waveOnly, live, in, out);
	}
}

// W_N^e as the product of two table entries (e = 64 hi + lo)
template <typename T> __device__ inline cx<T> mixrad_tw(const cx<T>* lo, const cx<T>* hi, uint32_t e) { return cmul(hi[e >> 6], lo[e & 63u]); }

// one column step of radix RAD over the buffers of a tile.  Items (r, o, k2): row, the index the step does not touch (o < M / RAD), column.  Input i of the butterfly
// sits in buffer r * M + o * so + i * si.  FIRST: the twiddle W_N^(b k2) on the loads, b = buffer index inside the row.  POSTTW: output k times W_M^(o k) (step one
// of two).  emit(r, o, k2, k, v) receives output k.
template <typename T, int RAD, int P, int SP, int NT, bool FIRST, bool POSTTW, typename EMIT>
__device__ inline void mixrad_col_step(const cx<T>* bufs, const uint32_t M, const uint32_t so, const uint32_t si, const uint32_t rowsHere, const cx<T>* twLo, const cx<T>* twHi,
                                       const cx<T>* wM, const uint32_t tid, const EMIT& emit) {
	const uint32_t others = M / (uint32_t)RAD, total = rowsHere * others * (uint32_t)P;
	FastDiv divO; divO.d = others; divO.rcp = 1.0f / (float)others;
	for (uint32_t j = tid; j < total; j += (uint32_t)NT) {
		const uint32_t q = j / (uint32_t)P, k2 = j - q * (uint32_t)P;
		uint32_t r, o;
		divO.divmod(q, r, o);
		const uint32_t b0 = o * so;
		const cx<T>* const src = bufs + (r * M + b0) * (uint32_t)SP + k2;
		cx<T> y[RAD];
#pragma unroll
		for (int i = 0; i < RAD; i++) y[i] = src[(uint32_t)i * si * (uint32_t)SP];
		if constexpr (FIRST) {
#pragma unroll
			for (int i = 0; i < RAD; i++) {
				const uint32_t e = (b0 + (uint32_t)i * si) * k2;
				if (i > 0 || b0 != 0u) y[i] = cmul(y[i], mixrad_tw<T>(twLo, twHi, e));
			}
		}
		dft<RAD, T>(y);
#pragma unroll
		for (int k = 0; k < RAD; k++) {
			if constexpr (POSTTW) { if (k > 0) y[k] = cmul(y[k], wM[o * (uint32_t)k]); }
			emit(r, o, k2, (uint32_t)k, y[k]);
		}
	}
}
// the same step for ANY odd radix (a run-time value: 11, 13, the direct primes 17 ... 31, 15, 25 ...): the direct sum in its mirrored form,
//     X[k], X[RAD - k] = x0 + sum_j cos(2 pi j k / RAD) (x_j + x_(RAD-j))  -+  i sum_j sin(2 pi j k / RAD) (x_j - x_(RAD-j)),     j, k = 1 ... (RAD - 1) / 2,
// with the inputs READ AGAIN FROM LDS for every pair of outputs and the roots from the cofactor's table (W_RAD^m = wM[m * M / RAD]: one broadcast read per term).
// Rolled loops, a dozen live registers, one body for every such radix — the butterflies of butterflies.h hold 2 RAD ... 4 RAD registers (DftPrime<31>: the kernel at 193
// VGPRs, two wavefronts per SIMD, for every cofactor it serves).  O(RAD^2) LDS reads per butterfly; at 31 points that is still a fifth of the LDS time of the
// prime's convolution.  The outputs must not land on the inputs (they are re-read): only the last step of a complex row (results leave to memory) takes this form.
template <typename T, int P, int SP, int NT, bool FIRST, typename EMIT>
__device__ inline void mixrad_col_direct(cx<T>* bufs, const uint32_t M, const uint32_t RAD, const uint32_t so, const uint32_t si, const uint32_t rowsHere, const cx<T>* twLo,
                                         const cx<T>* twHi, const cx<T>* wM, const uint32_t tid, const EMIT& emit) {
	const uint32_t others = M / RAD, cols = rowsHere * others * (uint32_t)P, wStep = M / RAD, H = (RAD - 1u) / 2u, stride = si * (uint32_t)SP;
	FastDiv divO; divO.d = others; divO.rcp = 1.0f / (float)others;
	if constexpr (FIRST) { // the twiddle W_N^(b k2) once, in place, over every element of the tile's columns (lanes along k2)
		const uint32_t totalE = cols * RAD;
		FastDiv divR; divR.d = RAD; divR.rcp = 1.0f / (float)RAD;
		for (uint32_t e = tid; e < totalE; e += (uint32_t)NT) {
			const uint32_t qq = e / (uint32_t)P, k2 = e - qq * (uint32_t)P;
			uint32_t q, i, r, o;
			divR.divmod(qq, q, i);
			divO.divmod(q, r, o);
			const uint32_t b = o * so + i * si;
			if (b * k2 != 0u) { cx<T>* const x = bufs + (r * M + b) * (uint32_t)SP + k2; *x = cmul(*x, mixrad_tw<T>(twLo, twHi, b * k2)); }
		}
		VKFFT_SYNC();
	}
	// the pairs of outputs of a column are dealt over KC threads (29 * 97: one column per thread left 97 of 256 threads with a whole 29-point sum each)
	uint32_t KC = (2u * (uint32_t)NT) / (cols ? cols : 1u);
	KC = KC < 1u ? 1u : KC > H ? H : KC;
	const uint32_t items = cols * KC;
	FastDiv divC; divC.d = cols; divC.rcp = 1.0f / (float)cols;
	for (uint32_t j0 = tid; j0 < items; j0 += (uint32_t)NT) {
		uint32_t kc, c;
		divC.divmod(j0, kc, c);
		const uint32_t q = c / (uint32_t)P, k2 = c - q * (uint32_t)P;
		uint32_t r, o;
		divO.divmod(q, r, o);
		const cx<T>* const col = bufs + (r * M + o * so) * (uint32_t)SP + k2;
		const cx<T> x0 = col[0];
		if (kc == 0u) {
			cx<T> sum = x0;
			for (uint32_t i = 1; i < RAD; i++) sum = cadd(sum, col[i * stride]);
			emit(r, o, k2, 0u, sum);
		}
		for (uint32_t k = 1u + kc; k <= H; k += KC) {
			cx<T> a = x0, b = {(T)0, (T)0};
			uint32_t m = 0;
#pragma unroll 2
			for (uint32_t jj = 1; jj <= H; jj++) {
				m += k; if (m >= RAD) m -= RAD;
				const cx<T> vj = col[jj * stride], vm = col[(RAD - jj) * stride], w = wM[m * wStep]; // w = cos - i sin
				a.x += w.x * (vj.x + vm.x); a.y += w.x * (vj.y + vm.y);
				b.x -= w.y * (vj.x - vm.x); b.y -= w.y * (vj.y - vm.y);
			}
			emit(r, o, k2, k, cx<T>{a.x + b.y, a.y - b.x});       // X_k = a - i b
			emit(r, o, k2, RAD - k, cx<T>{a.x - b.y, a.y + b.x}); // X_(RAD-k) = a + i b
		}
	}
}
#ifndef VKFFT_MIXRAD_RADICES
#define VKFFT_MIXRAD_RADICES(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(12) // butterflies in registers (mixrad_plan.h mixrad_radix_reg)
#endif

// lut = stage twiddles of SCH (length P - 1); rader = uint32 g^a mod P (a < L) followed by g^-k mod P; aux2 (aux3 for the real transforms) = FFT of the Rader
// kernel / L (L entries), W_N^lo (64), W_N^(64 hi) (ceil(N / 64)), W_M^e (M);  raderM = M, raderA = A, T = rows (transforms) per workgroup.
// SQ: the instance for M = P (rows of P * P points; only the primes up to 61 have one) — its column convolution costs registers that the other cofactors of the same
// prime should not pay for (37-point instance: 84 -> 100 VGPRs)
template <typename T, typename SCH, int TPF, int FPW, bool SQ>
__global__ void __launch_bounds__((MixradGeom<T, SCH, TPF, FPW>::NT)) mixrad_kernel(const PassParams p) {
	using G = MixradGeom<T, SCH, TPF, FPW>;
	constexpr int L = G::L, P = G::P, NT = G::NT, SP = G::SP, LUTN = G::LUTN;
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	VKFFT_DYN_SMEM(smem)
	const uint32_t tid = threadIdx.x;
	// thread group and place inside it (aligned: lanes beyond the last whole group of a wavefront belong to no group — they run along and write nothing; dense: the
	// threads beyond TPF * FPW likewise)
	const bool aligned = G::GROUPS_ALIGNED != 0 && p.raderAligned != 0;
	const uint32_t lane = tid & 63u;
	const uint32_t f = aligned ? (tid >> 6) * (uint32_t)G::GPW + lane / (uint32_t)TPF : tid / (uint32_t)TPF, tau = aligned ? lane % (uint32_t)TPF : tid % (uint32_t)TPF;
	const bool inGroup = aligned ? lane < (uint32_t)(G::GPW * TPF) : tid < (uint32_t)(TPF * G::GROUPS_DENSE);
	const uint32_t GROUPS = aligned ? (uint32_t)G::GROUPS_ALIGNED : (uint32_t)G::GROUPS_DENSE;
	const bool waveOnly = aligned || 64 % TPF == 0;
	const uint32_t M = p.raderM, A = p.raderA, B = A ? M / A : 1u, N = M * (uint32_t)P, R = p.T;
	const uint32_t nbuf = R * M, NH = (N + 63u) / 64u;
	cx<T>* const bufs = (cx<T>*)smem;
	const bool ops = p.preOp != OP_NONE || p.postOp != OP_NONE;
	// a real transform whose last column step is a direct sum: that step may not write over its inputs and the post-map reads LDS — a second set of buffers takes its results
	const bool twoSets = ops && mixrad_two_sets(M, A);
	cx<T>* const bufs2 = twoSets ? bufs + nbuf * (uint32_t)SP : bufs;
	cx<T>* const sLut = bufs2 + nbuf * (uint32_t)SP;
	cx<T>* const sBh = sLut + LUTN;
	cx<T>* const sTwLo = sBh + L;
	cx<T>* const sTwHi = sTwLo + 64;
	cx<T>* const sWM = sTwHi + NH;
	uint16_t* const sGp = (uint16_t*)(sWM + M);
	FastDiv divN, divM;
	divN.d = N; divN.rcp = 1.0f / (float)N; divM.d = M; divM.rcp = 1.0f / (float)M;
	// (one workgroup per tile.  A persistent form — the tables loaded once per workgroup, tiles blockIdx.x, + gridDim.x, ... — was built and measured: the tile loop
	// keeps its invariants live, 84-88 -> 128-136 VGPRs, and 3232-point rows fell from 2.44 to 1.74 TB/s; profiles/r06_rader_stage_persistent_form_*.jsonl)
	uint32_t wg = p.reverseTiles ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
	const uint32_t tile = wg % p.tilesPerG0;
	wg /= p.tilesPerG0;
	const uint32_t g1 = wg % p.dim[1].count, g2 = wg / p.dim[1].count;
	const uint32_t rowMult = (ops && p.pairRows) ? 2u : 1u; // two real rows per complex row of the tile (kernel_tmaps.h)
	const uint32_t f0 = tile * R * rowMult;
	const uint32_t realRowsHere = p.dim[0].count - f0 < R * rowMult ? p.dim[0].count - f0 : R * rowMult;
	const uint32_t rowsHere = (realRowsHere + rowMult - 1u) / rowMult; // complex rows of the tile
	const int64_t rowIn0 = (int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride + (int64_t)f0 * p.dim[0].inStride;
	const int64_t rowOut0 = (int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride + (int64_t)f0 * p.dim[0].outStride;
	// ---- 1. tables -> LDS, rows -> buffers (element M a + b of row r = element a of buffer r * M + b).  Every table entry of a thread and its first eight row
	// elements are REQUESTED before anything is written to LDS: one trip to memory for the tables and most of the tile (a loop that loads, waits and stores per
	// table and per batch of four cost six trips in a row per workgroup — half the lifetime of a workgroup on 3232-point rows)
	constexpr int KL = (LUTN + NT - 1) / NT, KT = (L + 64 + 64 + 128 + NT - 1) / NT, KG = (2 * L + NT - 1) / NT; // (NH <= 64, M <= 110)
	cx<T> tl[KL > 0 ? KL : 1], tt[KT]; uint32_t tg[KG];
	const uint32_t ntab = (uint32_t)L + 64u + NH + M; // (sBh, sTwLo, sTwHi, sWM are one run)
	{
		const GBuf glut = make_gbuf(p.lut), gtab = make_gbuf(ops ? p.aux3 : p.aux2), ggp = make_gbuf(p.rader);
#pragma unroll
		for (int k = 0; k < KL; k++) { const uint32_t i = tid + (uint32_t)(k * NT); tl[k] = gb_load<T>(glut, i < (uint32_t)LUTN ? i * ES : kGbInvalid, 0); }
#pragma unroll
		for (int k = 0; k < KT; k++) { const uint32_t i = tid + (uint32_t)(k * NT); tt[k] = gb_load<T>(gtab, i < ntab ? i * ES : kGbInvalid, 0); }
#pragma unroll
		for (int k = 0; k < KG; k++) { const uint32_t i = tid + (uint32_t)(k * NT); tg[k] = tm_load_u1(ggp, i < 2u * (uint32_t)L ? i * 4u : kGbInvalid, 0u); }
	}
	auto tables_to_lds = [&]() {
#pragma unroll
		for (int k = 0; k < KL; k++) { const uint32_t i = tid + (uint32_t)(k * NT); if (i < (uint32_t)LUTN) sLut[i] = tl[k]; }
#pragma unroll
		for (int k = 0; k < KT; k++) { const uint32_t i = tid + (uint32_t)(k * NT); if (i < ntab) sBh[i] = tt[k]; }
#pragma unroll
		for (int k = 0; k < KG; k++) { const uint32_t i = tid + (uint32_t)(k * NT); if (i < 2u * (uint32_t)L) sGp[i] = (uint16_t)tg[k]; }
	};
	if (ops) {
		tables_to_lds();
		const GBuf gdi = make_gbuf((const char*)p.in + rowIn0 * (int64_t)p.inElemBytes);
		const uint32_t pitch = (uint32_t)p.dim[0].inStride * p.inElemBytes;
		auto put = [&](uint32_t r, uint32_t pos, cx<T> z) { uint32_t a, b; divM.divmod(pos, a, b); bufs[(r * M + b) * (uint32_t)SP + a] = z; };
		if (p.tmPreFlags & kTmTwo) tm_rows_in<T, true>(p.tmPre, gdi, N, divN, rowsHere, realRowsHere, rowMult, pitch, p.swapIn != 0, tid, (uint32_t)NT, put);
		else tm_rows_in<T, false>(p.tmPre, gdi, N, divN, rowsHere, realRowsHere, rowMult, pitch, p.swapIn != 0, tid, (uint32_t)NT, put);
	} else {
		const GBuf gin = make_gbuf((const cx<T>*)p.in + rowIn0);
		const bool swI = p.bluesteinSwapIn != 0;
		const uint32_t inRowBytes = (uint32_t)p.dim[0].inStride * ES;
		constexpr int U = 8; // elements requested together
		const uint32_t total = rowsHere * N;
		for (uint32_t e0 = tid; e0 < total || e0 == tid; e0 += (uint32_t)(U * NT)) {
			cx<T> v[U]; uint32_t dst[U];
#pragma unroll
			for (int u = 0; u < U; u++) {
				const uint32_t e = e0 + (uint32_t)(u * NT);
				uint32_t r, n, a, b;
				divN.divmod(e < total ? e : 0u, r, n);
				divM.divmod(n, a, b);
				v[u] = gb_load<T>(gin, e < total ? r * inRowBytes + n * ES : kGbInvalid, 0);
				dst[u] = (r * M + b) * (uint32_t)SP + a;
			}
			if (e0 == tid) tables_to_lds(); // (first trip: the tables were requested ahead of these elements and land first)
#pragma unroll
			for (int u = 0; u < U; u++) if (e0 + (uint32_t)(u * NT) < total) bufs[dst[u]] = swI ? cswap(v[u]) : v[u];
		}
	}
	VKFFT_SYNC();
	// ---- 2. Rader convolution of every buffer, in place (the flow of mixconv_kernel<RADER = 1>)
	auto fsync = [&]() { if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC(); };
	{
		const uint32_t rounds = (nbuf + GROUPS - 1u) / GROUPS;
		for (uint32_t rd = 0; rd < rounds; rd++) {
			const uint32_t job = rd * GROUPS + f;
			const bool live = inGroup && job < nbuf;
			cx<T>* const row = bufs + (live ? job : 0u) * (uint32_t)SP;
			const cx<T> x0 = row[0];
			cx<T> dc = {(T)0, (T)0};
			// forward transform of x[g^a]; spectrum * FFT(w^(g^-q)) / L, + x0 on the zero frequency (= x0 added to every output); X[0] = x0 + sum of the others
			mixrad_stage<T, SCH, 0, TPF, 1, true>(row, (const cx<T>*)sLut, tau, waveOnly, live, [&](uint32_t t, uint32_t c) -> cx<T> { return row[sGp[t + c]]; },
			                                      [&](uint32_t t, uint32_t c, cx<T> v) {
				                                      const uint32_t k = t + c;
				                                      cx<T> w = cmul(v, sBh[k]);
				                                      if (k == 0u) { dc = cadd(x0, v); w = cadd(w, x0); }
				                                      if (live) row[k] = cswap(w);
			                                      });
			fsync();
			// inverse transform; result q belongs to output index g^-q
			mixrad_stage<T, SCH, 0, TPF, 1, true>(row, (const cx<T>*)sLut, tau, waveOnly, live, [&](uint32_t t, uint32_t c) -> cx<T> { return row[t + c]; },
			                                      [&](uint32_t t, uint32_t c, cx<T> v) { if (live) row[sGp[(uint32_t)L + t + c]] = cswap(v); });
			if (tau == 0u && live) row[0] = dc; // (slot 0 carried the zero frequency of the spectrum; the scatter writes 1 ... P - 1)
		}
	}
	VKFFT_SYNC();
	// ---- 2b. M = P (rows of P * P points): the transform along b is the same prime — the same convolution on the COLUMNS of the tile (element b of column k2 of row r
	// at buffer r * M + b, offset k2: element pitch SP), the twiddle W_N^(b k2) on the gather.  Result k1 lands in buffer k1: natural index n at buffer n / P, offset n mod P
	if constexpr (SQ) {
		const uint32_t ncol = R * (uint32_t)P, rounds = (ncol + GROUPS - 1u) / GROUPS;
		for (uint32_t rd = 0; rd < rounds; rd++) {
			const uint32_t job = rd * GROUPS + f;
			const bool live = inGroup && job < ncol;
			const uint32_t jc = live ? job : 0u, r = jc / (uint32_t)P, k2 = jc - r * (uint32_t)P;
			cx<T>* const col = bufs + r * M * (uint32_t)SP + k2;
			const cx<T> x0 = col[0];
			cx<T> dc = {(T)0, (T)0};
			mixrad_stage<T, SCH, 0, TPF, SP, false>(col, (const cx<T>*)sLut, tau, waveOnly, live,
			                                        [&](uint32_t t, uint32_t c) -> cx<T> { const uint32_t b = sGp[t + c]; return cmul(col[b * (uint32_t)SP], mixrad_tw<T>(sTwLo, sTwHi, b * k2)); },
			                                        [&](uint32_t t, uint32_t c, cx<T> v) {
				                                        const uint32_t k = t + c;
				                                        cx<T> w = cmul(v, sBh[k]);
				                                        if (k == 0u) { dc = cadd(x0, v); w = cadd(w, x0); }
				                                        if (live) col[k * (uint32_t)SP] = cswap(w);
			                                        });
			fsync();
			mixrad_stage<T, SCH, 0, TPF, SP, false>(col, (const cx<T>*)sLut, tau, waveOnly, live, [&](uint32_t t, uint32_t c) -> cx<T> { return col[(t + c) * (uint32_t)SP]; },
			                                        [&](uint32_t t, uint32_t c, cx<T> v) { if (live) col[(uint32_t)sGp[(uint32_t)L + t + c] * (uint32_t)SP] = cswap(v); });
			if (tau == 0u && live) col[0] = dc;
		}
		VKFFT_SYNC();
	}
	// ---- 3. column steps
	const bool swO = ops ? false : p.bluesteinSwapOut != 0;
	const T sc = ops ? (T)1 : (T)p.scale;
	const GBuf gout = make_gbuf((cx<T>*)p.out + rowOut0);
	const uint32_t outRowBytes = (uint32_t)p.dim[0].outStride * ES;
	const uint32_t Ae = A ? A : 1u; // (M = P: one "step" of M, already done)
	if ((SQ || M == 1u) && !ops) { // (P * P, and M = 1: a row of P points) the rows leave in natural order: one contiguous run per row
		const uint32_t total = rowsHere * N;
		for (uint32_t e = tid; e < total; e += (uint32_t)NT) {
			uint32_t r, n;
			divN.divmod(e, r, n);
			const uint32_t k1 = n / (uint32_t)P, k2 = n - k1 * (uint32_t)P;
			cx<T> v = bufs[(r * M + k1) * (uint32_t)SP + k2];
			if (swO) v = cswap(v);
			if (sc != (T)1) v = cscale(v, sc);
			gb_store<T>(gout, r * outRowBytes + n * ES, 0, v);
		}
	}
	if (!SQ && A > 1u) {
		auto inPlace = [&](uint32_t r, uint32_t o, uint32_t k2, uint32_t k, cx<T> v) { bufs[(r * M + o + k * B) * (uint32_t)SP + k2] = v; };
#define VKFFT_MIXRAD_CASE(m) case m: if constexpr (2 * m * P <= (int)kMixradLongest) mixrad_col_step<T, m, P, SP, NT, true, true>(bufs, M, 1u, B, rowsHere, sTwLo, sTwHi, sWM, tid, inPlace); break;
		switch (A) { VKFFT_MIXRAD_RADICES(VKFFT_MIXRAD_CASE) default: break; }
#undef VKFFT_MIXRAD_CASE
		VKFFT_SYNC();
	}
	if (!SQ && M > 1u) {
		auto store = [&](uint32_t r, uint32_t o, uint32_t k2, uint32_t k, cx<T> v) {
			if (ops) { bufs2[(r * M + o * B + k) * (uint32_t)SP + k2] = v; return; }
			if (swO) v = cswap(v);
			if (sc != (T)1) v = cscale(v, sc);
			gb_store<T>(gout, r * outRowBytes + (k2 + (uint32_t)P * (o + A * k)) * ES, 0, v);
		};
		// (A == 1: this is the first step and carries the twiddle; else the buffers hold step one's results)
#define VKFFT_MIXRAD_CASE(m) case m: if constexpr (m * P <= (int)kMixradLongest) { \
			if (A == 1u) mixrad_col_step<T, m, P, SP, NT, true, false>(bufs, M, B, 1u, rowsHere, sTwLo, sTwHi, sWM, tid, store); \
			else if constexpr (2 * m * P <= (int)kMixradLongest) mixrad_col_step<T, m, P, SP, NT, false, false>(bufs, M, B, 1u, rowsHere, sTwLo, sTwHi, sWM, tid, store); } break;
		switch (B) {
		VKFFT_MIXRAD_RADICES(VKFFT_MIXRAD_CASE)
		default: // an odd radix without a register butterfly here (complex rows only: mixrad_plan.h)
			if (A == 1u) mixrad_col_direct<T, P, SP, NT, true>(bufs, M, B, B, 1u, rowsHere, sTwLo, sTwHi, sWM, tid, store);
			else mixrad_col_direct<T, P, SP, NT, false>(bufs, M, B, B, 1u, rowsHere, sTwLo, sTwHi, sWM, tid, store);
			break;
		}
#undef VKFFT_MIXRAD_CASE
	}
	if (ops) {
		VKFFT_SYNC();
		const bool swOut = p.swapOut != 0;
		tm_rows_out<T>(p.tmPost, make_gbuf((char*)p.out + rowOut0 * (int64_t)p.outElemBytes), N, rowsHere, realRowsHere, rowMult, (uint32_t)p.dim[0].outStride * p.outElemBytes, p.tmPostFlags,
		               tid, (uint32_t)NT, [&](uint32_t r, uint32_t n) -> cx<T> {
			               const uint32_t k1 = n / (uint32_t)P, k2 = n - k1 * (uint32_t)P, k1b = k1 / Ae, k1a = k1 - k1b * Ae;
			               const cx<T> v = bufs2[(r * M + k1a * (M / Ae) + k1b) * (uint32_t)SP + k2];
			               return swOut ? cswap(v) : v;
		               });
	}
}

template <typename T, typename SCH, int TPF, int FPW> void mixrad_launch(const PassParams& prm, dim3 grid, hipStream_t s) {
	using G = MixradGeom<T, SCH, TPF, FPW>;
	const bool ops = prm.preOp != OP_NONE || prm.postOp != OP_NONE;
	const size_t lds = (size_t)mixrad_lds_bytes((uint32_t)G::P, (uint32_t)G::SP, (uint32_t)G::LUTN, prm.raderM, prm.T, (uint32_t)sizeof(cx<T>), ops && mixrad_two_sets(prm.raderM, prm.raderA));
	const bool sq = prm.raderA == 0u;
	const dim3 g = grid;
	if (sq) {
		if constexpr (G::P * G::P <= (int)kMixradLongest) hipLaunchKernelGGL((mixrad_kernel<T, SCH, TPF, FPW, true>), g, dim3(G::NT), lds, s, prm);
	} else hipLaunchKernelGGL((mixrad_kernel<T, SCH, TPF, FPW, false>), g, dim3(G::NT), lds, s, prm);
}
// the composite form exists for the Rader ROW instances whose prime leaves room for a cofactor (2 P <= the longest row)
template <typename T, typename SCH, int TPF, int FPW, int RADER, int COL> constexpr auto mixrad_ptr() -> void (*)(const PassParams&, dim3, hipStream_t) {
	constexpr int P = SCH::N + 1;
	if constexpr (RADER != 0 && COL == 0 && sizeof(T) == 4 && 2 * P <= (int)kMixradLongest) return &mixrad_launch<T, SCH, TPF, FPW>;
	else return nullptr;
}
template <typename T, typename SCH, int TPF, int RADER, int COL> constexpr int mixrad_sp() { if constexpr (RADER != 0 && COL == 0) return MixradGeom<T, SCH, TPF>::SP; else return 0; }
template <typename T, typename SCH, int TPF, int RADER, int COL> constexpr int mixrad_lutn() { if constexpr (RADER != 0 && COL == 0) return MixradGeom<T, SCH, TPF>::LUTN; else return 0; }
template <typename T, typename SCH, int TPF, int FPW, int RADER, int COL> constexpr int mixrad_groups() { if constexpr (RADER != 0 && COL == 0) return MixradGeom<T, SCH, TPF, FPW>::GROUPS_ALIGNED; else return 0; }
template <typename T, typename SCH, int TPF, int FPW, int RADER, int COL> constexpr int mixrad_groups_dense() { if constexpr (RADER != 0 && COL == 0) return MixradGeom<T, SCH, TPF, FPW>::GROUPS_DENSE; else return 0; }

} // namespace vkfft_mi355x
