// Hand-specialised single-pass kernels WITH a fused pre/post map: the real transforms (R2C / C2R through the even
// decomposition, DCT/DST I-IV; reference vkFFT_R2C.h:178,450, vkFFT_R2R.h:28-861) on unit-stride rows and on strided
// columns, and strided C2C of non-power-of-two length.  Same register/LDS structure as kernel_mixed.h (compile-time
// radix schedule, one LDS buffer, first stage fed from global memory, last stage stored to global memory); what is added:
//   * the first stage's inputs come from pre_gather<PRE> (every pre-map is a gather: kernel_generic.h), the last stage's
//     outputs go through post_scatter<POST> (every post-map except the R2C even split is 1:1 or 1:2 in scatter form) —
//     the R2C even split needs Z[k] and Z[H-k] together and makes one more trip through LDS;
//   * COL = true lays the workgroup across FPW neighbouring columns (lanes along the unit-stride direction) so that a
//     strided axis is read and written in FPW-element segments without a transposition.
// PRE/POST are compile-time: the DST members of a family share the DCT instance (op read from the pass descriptor, the
// switch folds to the one family body).
#pragma once
#include "engine.h"
#include "kernel_generic.h"
#include "mix_sched.h"

namespace vkfft_mi355x {

// the DST member of a DCT family at run time, everything else at compile time
template <int OP> __device__ inline uint32_t op_resolve(const uint32_t runtimeOp) {
	if constexpr (OP == OP_DCT2_PRE) return runtimeOp == OP_DST2_PRE ? OP_DST2_PRE : OP_DCT2_PRE;
	else if constexpr (OP == OP_DCT3_PRE) return runtimeOp == OP_DST3_PRE ? OP_DST3_PRE : OP_DCT3_PRE;
	else if constexpr (OP == OP_DCT4_PRE) return runtimeOp == OP_DST4_PRE ? OP_DST4_PRE : OP_DCT4_PRE;
	else if constexpr (OP == OP_DCT2H_PRE) return runtimeOp == OP_DST2H_PRE ? OP_DST2H_PRE : OP_DCT2H_PRE;
	else if constexpr (OP == OP_DCT3H_PRE) return runtimeOp == OP_DST3H_PRE ? OP_DST3H_PRE : OP_DCT3H_PRE;
	else if constexpr (OP == OP_DCT2H_POST) return runtimeOp == OP_DST2H_POST ? OP_DST2H_POST : OP_DCT2H_POST;
	else if constexpr (OP == OP_DCT3H_POST) return runtimeOp == OP_DST3H_POST ? OP_DST3H_POST : OP_DCT3H_POST;
	else if constexpr (OP == OP_DCT2_POST) return runtimeOp == OP_DST2_POST ? OP_DST2_POST : OP_DCT2_POST;
	else if constexpr (OP == OP_DCT3_POST) return runtimeOp == OP_DST3_POST ? OP_DST3_POST : OP_DCT3_POST;
	else if constexpr (OP == OP_DCT4_POST) return runtimeOp == OP_DST4_POST ? OP_DST4_POST : OP_DCT4_POST;
	else return (uint32_t)OP;
}

// Mirrored-pair wide access of the half-length DCT/DST-II/III on unit-stride rows.  With z[n] = v[2n] + i v[2n+1] and v the
// Makhoul permutation of the real row, the four consecutive reals x[4n .. 4n+3] (n < H/2) are exactly
//   (Re z[n], Im z[H-1-n], Im z[n], Re z[H-1-n]),
// and H-1-n is input (output) R-1-i of butterfly NB-1-t when n is input (output) i of butterfly t.  A thread that owns the
// butterfly pair (t, NB-1-t) of the first (last) stage therefore moves its 2R points with R 16-byte accesses and no index
// arithmetic, instead of 2R strided 4-byte accesses.  Needs an even radix and an even number of butterflies per thread.
template <typename SCH, int SI, int TPF> __host__ __device__ constexpr bool opfft_can_pair() {
	constexpr int R = SCH::rad[SI], NB = SCH::N / R;
	return (R % 2 == 0) && (NB % (2 * TPF) == 0);
}
// butterfly owned by slot b of thread tau: the plain map, or pairs (t, NB-1-t) in slots (b, b + P/2)
template <int NB, int TPF, int P, bool PAIR> __device__ inline uint32_t opfft_bfly(const uint32_t tau, const int b) {
	if constexpr (!PAIR) return tau + b * TPF;
	else return b < P / 2 ? tau + b * TPF : (uint32_t)(NB - 1) - (tau + (b - P / 2) * TPF);
}

template <typename T, typename SCH, int SI, int TPF, int PRE, int POST, bool ROW, bool TRANS>
__device__ inline void op_stage(cx<T>* ldsf, const Io32<T>& io, const GBuf glut, const uint32_t tau, const bool waveOnly, const PassParams& p,
                                const uint32_t colIdx, const uint32_t nat) {
	constexpr int N = SCH::N, R = SCH::rad[SI], NB = N / R, P = (NB + TPF - 1) / TPF, S = SCH::S(SI);
	constexpr bool first = SI == 0, last = SI == SCH::NS - 1;
	constexpr bool staged = POST == OP_R2C_EVEN_POST || POST == OP_DCT2H_POST || POST == OP_DCT1H_POST || TRANS; // the post-map (or the transposed store) gathers from LDS
	constexpr bool pairIn = ROW && first && PRE == OP_DCT2H_PRE && opfft_can_pair<SCH, SI, TPF>();
	constexpr bool pairOut = ROW && last && POST == OP_DCT3H_POST && opfft_can_pair<SCH, SI, TPF>();
	// half-length DCT/DST-IV: FFT input n is (x[2n], x[Nr-1-2n]) * twiddle and output m feeds y[2m] and y[Nr-1-2m].  The two consecutive reals
	// x[2n], x[2n+1] belong to n and to its mirror H-1-n, which is input R-1-i of butterfly NB-1-t when n is input i of butterfly t: a thread that
	// owns the butterfly pair (t, NB-1-t) moves its points with 8-byte accesses that are contiguous across the lanes (instead of two 4-byte
	// accesses of stride 2 per point).  Reference: vkFFT_R2R.h:368,861.
	constexpr bool pairIn4 = ROW && first && PRE == OP_DCT4_PRE && opfft_can_pair<SCH, SI, TPF>();
	constexpr bool pairOut4 = ROW && last && POST == OP_DCT4_POST && opfft_can_pair<SCH, SI, TPF>();
	constexpr bool PAIR = pairIn || pairOut || pairIn4 || pairOut4;
	// LDS padding per exchange: rows pick it by conflict count (MixPad); column tiles need none (lanes run along the columns)
	using PAD = MixPad<SCH, TPF, (int)sizeof(cx<T>)>;
	constexpr int PADIN = ROW ? PAD::shift(SI - 1) : 0, PADOUT = ROW ? PAD::shift(SI) : 0;
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	cx<T> x[P][R];
	if constexpr (pairIn) {
		const bool dst = p.preOp == OP_DST2H_PRE; // DST-II = DCT-II of (-1)^j x_j: the odd reals change sign
#pragma unroll
		for (int b = 0; b < P; b++) {
			const uint32_t t = opfft_bfly<NB, TPF, P, true>(tau, b);
			constexpr int mirror = P / 2;
			const int bm = b < mirror ? b + mirror : b - mirror; // slot of butterfly NB-1-t
#pragma unroll
			for (int i = 0; i < R / 2; i++) {
				Real4<T> q = gb_load_real4<T>(io.gin, io.inOff + t * (2 * ES), (uint32_t)(i * NB) * (2 * ES));
				if (dst) { q.y = -q.y; q.w = -q.w; }
				const cx<T> a = {q.x, q.z}, m = {q.w, q.y};
				x[b][i] = p.swapIn ? cswap(a) : a;
				x[bm][R - 1 - i] = p.swapIn ? cswap(m) : m;
			}
		}
	} else if constexpr (pairIn4) {
		const bool dst = p.preOp == OP_DST4_PRE; // DST-IV reads the reversed row: (x[2n], x[Nr-1-2n]) swap roles
		const GBuf gtw = make_gbuf(p.aux);
		constexpr uint32_t RS = (uint32_t)sizeof(T);
#pragma unroll
		for (int b = 0; b < P / 2; b++) {
			const uint32_t t = opfft_bfly<NB, TPF, P, true>(tau, b);
			constexpr int mirror = P / 2;
#pragma unroll
			for (int i = 0; i < R; i++) {
				const uint32_t n = t + i * NB, nm = (uint32_t)N - 1u - n; // nm = input R-1-i of butterfly NB-1-t
				const cx<T> q = gb_load<T>(io.gin, io.inOff + n * (2 * RS), 0);   // (x[2n], x[2n+1]): T pairs are read as one cx<T>-sized access
				const cx<T> qm = gb_load<T>(io.gin, io.inOff + nm * (2 * RS), 0); // (x[2nm], x[2nm+1])
				const cx<T> a = dst ? cx<T>{qm.y, q.x} : cx<T>{q.x, qm.y}, am = dst ? cx<T>{q.y, qm.x} : cx<T>{qm.x, q.y};
				const cx<T> z = cmul(a, gb_load<T>(gtw, n * ES, 0)), zm = cmul(am, gb_load<T>(gtw, nm * ES, 0));
				x[b][i] = p.swapIn ? cswap(z) : z;
				x[b + mirror][R - 1 - i] = p.swapIn ? cswap(zm) : zm;
			}
		}
	} else if constexpr (first && PRE == OP_DCT3H_PRE && ROW) {
		// half-length DCT/DST-III on unit-stride rows, pairs (k, H-k) together (H = N, Nr = 2H reals): the Hermitian spectrum entries
		// V_k = conj(c_k)(x_k - i x_{Nr-k}) and V_{H-k} come from four reals each read ONCE, and fold (as in the C2R path above) into both
		// z_k = s + i d and z_{H-k} = conj(s - i d), s = V_k + conj(V_{H-k}), d = conj(w_k)(V_k - conj(V_{H-k})); z is laid down in LDS and the
		// first stage gathers its inputs from there.  (Round 1 built V in LDS first and folded every z_n on its own: two LDS reads and one
		// table entry per point.)  Reference vkFFT_R2R.h:193 runs a full-length complex FFT instead.
		constexpr int HP = N / 2 + 1, PH = (HP + TPF - 1) / TPF;
		constexpr uint32_t Hc = (uint32_t)N, Nr = 2u * (uint32_t)N;
		const bool dst = p.preOp == OP_DST3H_PRE;
		auto X = [&](uint32_t k) -> T { return k >= Nr ? (T)0 : io.ldr(dst ? Nr - 1u - k : k); };
		const GBuf gc = make_gbuf(p.aux), gw = make_gbuf(p.aux2);
#pragma unroll
		for (int b = 0; b < PH; b++) {
			const uint32_t k = tau + b * TPF;
			if ((b + 1) * TPF <= HP || k < (uint32_t)HP) {
				const uint32_t km = Hc - k;
				const cx<T> vk = cmul(cconj(gb_load<T>(gc, k * (uint32_t)sizeof(cx<T>), 0)), cx<T>{X(k), -X(Nr - k)});
				const cx<T> vm = cmul(cconj(gb_load<T>(gc, km * (uint32_t)sizeof(cx<T>), 0)), cx<T>{X(km), -X(Nr - km)});
				const cx<T> bq = cconj(vm);
				const cx<T> d = cmul(cconj(gb_load<T>(gw, k * (uint32_t)sizeof(cx<T>), 0)), csub(vk, bq)), s2 = cadd(vk, bq);
				const cx<T> zk = {s2.x - d.y, s2.y + d.x}, zm = {s2.x + d.y, d.x - s2.y};
				ldsf[k] = p.swapIn ? cswap(zk) : zk;
				if (k != 0 && 2 * k != Hc) ldsf[km] = p.swapIn ? cswap(zm) : zm;
			}
		}
		if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC();
#pragma unroll
		for (int b = 0; b < P; b++) {
			const uint32_t t = opfft_bfly<NB, TPF, P, PAIR>(tau, b);
			if (PAIR || (b + 1) * TPF <= NB || t < (uint32_t)NB) {
#pragma unroll
				for (int i = 0; i < R; i++) x[b][i] = ldsf[t + i * NB];
			}
		}
		if constexpr (!last) { if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC(); } // z is consumed before stage 0 overwrites the buffer
	} else if constexpr (first && PRE == OP_DCT3H_PRE) {
		// half-length DCT/DST-III: the Hermitian spectrum V_k = e^{+i pi k/2N}(x_k - i x_{N-k}), k = 0..H, is built ONCE in LDS
		// (every real read once, one table entry per k), then each FFT input is the even C2R fold of V_n and V_{H-n}.  Gathering
		// straight from global memory (pre_gather) reads every real twice and keeps 4 reals + 3 table entries per point in
		// flight: the unrolled kernels spilled up to 1 KB/lane.
		const uint32_t Nr = p.opN, H = Nr >> 1;
		const bool dst = p.preOp == OP_DST3H_PRE;
		auto X = [&](uint32_t k) -> T { return k >= Nr ? (T)0 : io.ldr(dst ? Nr - 1 - k : k); };
		constexpr int PV = (N + 1 + TPF - 1) / TPF;
#pragma unroll
		for (int b = 0; b < PV; b++) {
			const uint32_t k = tau + b * TPF;
			if (k <= H) ldsf[k] = cmul(cconj(table_load<T>(p.aux, k)), cx<T>{X(k), -X(Nr - k)});
		}
		if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC();
#pragma unroll
		for (int b = 0; b < P; b++) {
			const uint32_t t = opfft_bfly<NB, TPF, P, PAIR>(tau, b);
			if (PAIR || (b + 1) * TPF <= NB || t < (uint32_t)NB) {
#pragma unroll
				for (int i = 0; i < R; i++) {
					const uint32_t n = t + i * NB;
					const cx<T> a = ldsf[n], bq = cconj(ldsf[H - n]);
					const cx<T> d = cmul(cconj(table_load<T>(p.aux2, n)), csub(a, bq)), s2 = cadd(a, bq);
					const cx<T> z = {s2.x - d.y, s2.y + d.x};
					x[b][i] = p.swapIn ? cswap(z) : z;
				}
			}
		}
		if constexpr (!last) { if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC(); } // V is consumed before stage 0 overwrites the buffer
	} else if constexpr (first && PRE == OP_C2R_EVEN_PRE && ROW) {
		// C2R even fold on unit-stride rows, pairs (k, H-k) together (H = N): X_k and X_{H-k} are loaded ONCE, one table entry, and give
		// both z_k = s + i d and z_{H-k} = conj(s - i d), s = X_k + conj(X_{H-k}), d = conj(w_k)(X_k - conj(X_{H-k})); the packed sequence z
		// is laid down in LDS and the first stage gathers its inputs from there.  (pre_gather computes every z_n on its own: each X loaded
		// twice, twice the arithmetic.)  Reference: vkFFT_R2C_even_decomposition.h:40-180.
		constexpr int HP = N / 2 + 1, PH = (HP + TPF - 1) / TPF;
		const GBuf gw = make_gbuf(p.aux);
#pragma unroll
		for (int b = 0; b < PH; b++) {
			const uint32_t k = tau + b * TPF;
			if ((b + 1) * TPF <= HP || k < (uint32_t)HP) {
				const cx<T> a = io.ldc(k), bq = cconj(io.ldc((uint32_t)N - k));
				const cx<T> w = cconj(gb_load<T>(gw, k * (uint32_t)sizeof(cx<T>), 0));
				const cx<T> sS = cadd(a, bq), d = cmul(w, csub(a, bq));
				const cx<T> zk = {sS.x - d.y, sS.y + d.x}, zm = {sS.x + d.y, d.x - sS.y};
				ldsf[k] = p.swapIn ? cswap(zk) : zk;
				if (k != 0 && 2 * k != (uint32_t)N) ldsf[(uint32_t)N - k] = p.swapIn ? cswap(zm) : zm;
			}
		}
		if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC();
#pragma unroll
		for (int b = 0; b < P; b++) {
			const uint32_t t = opfft_bfly<NB, TPF, P, PAIR>(tau, b);
			if (PAIR || (b + 1) * TPF <= NB || t < (uint32_t)NB) {
#pragma unroll
				for (int i = 0; i < R; i++) x[b][i] = ldsf[t + i * NB];
			}
		}
		if constexpr (!last) { if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC(); } // z is consumed before stage 0 overwrites the buffer
	} else {
#pragma unroll
		for (int b = 0; b < P; b++) {
			const uint32_t t = opfft_bfly<NB, TPF, P, PAIR>(tau, b);
			if (PAIR || (b + 1) * TPF <= NB || t < (uint32_t)NB) {
#pragma unroll
				for (int i = 0; i < R; i++) {
					if constexpr (first) {
						const cx<T> v = pre_gather<T>(p, io, t + i * NB, nat, op_resolve<PRE>(p.preOp));
						x[b][i] = p.swapIn ? cswap(v) : v;
					} else x[b][i] = ldsf[mix_slot<PADIN>(t + i * NB)];
				}
			}
		}
	}
	if constexpr (!first && (!last || staged)) { // all inputs are in registers before the buffer is overwritten
		if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC();
	}
#pragma unroll
	for (int b = 0; b < P; b++) {
		const uint32_t t = opfft_bfly<NB, TPF, P, PAIR>(tau, b);
		if (PAIR || (b + 1) * TPF <= NB || t < (uint32_t)NB) {
			const uint32_t s = t % (uint32_t)S;
			if constexpr (!first) {
				constexpr int LO = SCH::lutOff(SI);
#pragma unroll
				for (int i = 1; i < R; i++) x[b][i] = cmul(x[b][i], gb_load<T>(glut, s * ES, (uint32_t)(LO + (i - 1) * S) * ES));
			}
			dft<R, T>(x[b]);
			if constexpr (!pairOut && !pairOut4) {
				const uint32_t ob = (t - s) * (uint32_t)R + s;
#pragma unroll
				for (int k = 0; k < R; k++) {
					if constexpr (last) {
						const cx<T> v = p.swapOut ? cswap(x[b][k]) : x[b][k];
						if constexpr (staged) ldsf[ob + k * S] = v; // natural order, unpadded: read back along k
						else post_scatter<T>(p, io, ob + k * S, v, colIdx, nat, op_resolve<POST>(p.postOp), p.outLen);
					} else ldsf[mix_slot<PADOUT>(ob + k * S)] = x[b][k];
				}
			}
		}
	}
	if constexpr (pairOut) { // last stage: output k of butterfly t is point t + k*NB; four reals per 16-byte store
		const bool dst = p.postOp == OP_DST3H_POST; // DST-III: odd outputs change sign
		const T sc = (T)p.scale, so = dst ? -sc : sc;
#pragma unroll
		for (int b = 0; b < P; b++) {
			const uint32_t t = opfft_bfly<NB, TPF, P, true>(tau, b);
			constexpr int mirror = P / 2;
			const int bm = b < mirror ? b + mirror : b - mirror;
#pragma unroll
			for (int k = 0; k < R / 2; k++) {
				const cx<T> a = p.swapOut ? cswap(x[b][k]) : x[b][k], m = p.swapOut ? cswap(x[bm][R - 1 - k]) : x[bm][R - 1 - k];
				gb_store_real4<T>(io.gout, io.outOff + t * (2 * ES), (uint32_t)(k * NB) * (2 * ES), Real4<T>{a.x * sc, m.y * so, a.y * sc, m.x * so});
			}
		}
	}
	if constexpr (pairOut4) { // last stage: output k of butterfly t is FFT output m = t + k*NB; (y[2m], y[2m+1]) = (2 Re c_m, -2 Im c_{H-1-m})
		const bool dst = p.postOp == OP_DST4_POST; // DST-IV: the odd outputs change sign
		const T sc2 = (T)2 * (T)p.scale, so = dst ? sc2 : -sc2;
		const GBuf gtw = make_gbuf(p.aux2);
		constexpr uint32_t RS = (uint32_t)sizeof(T);
#pragma unroll
		for (int b = 0; b < P / 2; b++) {
			const uint32_t t = opfft_bfly<NB, TPF, P, true>(tau, b);
			constexpr int mirror = P / 2;
#pragma unroll
			for (int k = 0; k < R; k++) {
				const uint32_t m = t + k * NB, mm = (uint32_t)N - 1u - m;
				const cx<T> v = p.swapOut ? cswap(x[b][k]) : x[b][k], vm = p.swapOut ? cswap(x[b + mirror][R - 1 - k]) : x[b + mirror][R - 1 - k];
				const cx<T> c = cmul(v, gb_load<T>(gtw, m * ES, 0)), cm = cmul(vm, gb_load<T>(gtw, mm * ES, 0));
				gb_store<T>(io.gout, io.outOff + m * (2 * RS), 0, cx<T>{sc2 * c.x, so * cm.y});
				gb_store<T>(io.gout, io.outOff + mm * (2 * RS), 0, cx<T>{sc2 * cm.x, so * c.y});
			}
		}
	}
	if constexpr (!last) {
		if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC();
		op_stage<T, SCH, SI + 1 < SCH::NS ? SI + 1 : SI, TPF, PRE, POST, ROW, TRANS>(ldsf, io, glut, tau, waveOnly, p, colIdx, nat);
	} else if constexpr (staged && !TRANS && POST == OP_R2C_EVEN_POST && ROW) {
		// R2C even split on unit-stride rows, pairs (k, H-k) together (H = N: complex length): one pair of LDS reads and ONE table entry give
		// both X[k] = (s - i d)/2 and X[H-k] = conj(s + i d)/2, s = Z_k + conj(Z_{H-k}), d = w_k (Z_k - conj(Z_{H-k})), w_{H-k} = -conj(w_k)
		// (reference: vkFFT_R2C_even_decomposition.h:181-230 computes every output on its own).  k = 0 yields X[0] and X[H].
		if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC();
		constexpr int HP = N / 2 + 1, PH = (HP + TPF - 1) / TPF; // pairs k = 0 .. N/2
		const T hs = (T)0.5 * (T)p.scale;
		const GBuf gw = make_gbuf(p.aux);
#pragma unroll
		for (int b = 0; b < PH; b++) {
			const uint32_t k = tau + b * TPF;
			if ((b + 1) * TPF <= HP || k < (uint32_t)HP) {
				const uint32_t km = k == 0 ? 0u : (uint32_t)N - k;
				const cx<T> zk = ldsf[k], zm = cconj(ldsf[km]);
				const cx<T> w = gb_load<T>(gw, k * (uint32_t)sizeof(cx<T>), 0);
				const cx<T> sS = cadd(zk, zm), d = cmul(w, csub(zk, zm));
				io.stc(k, cx<T>{hs * (sS.x + d.y), hs * (sS.y - d.x)});
				if (2 * k != (uint32_t)N) io.stc((uint32_t)N - k, cx<T>{hs * (sS.x - d.y), -hs * (sS.y + d.x)});
			}
		}
	} else if constexpr (staged && !TRANS && POST == OP_DCT1H_POST && ROW) {
		// half-length DCT-I on unit-stride rows: y[k] = Re X[k], y[H-k] = Re X[H-k] from the pair (Z_k, Z_{H-k}) and one table entry (the R2C split above)
		if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC();
		constexpr int HP = N / 2 + 1, PH = (HP + TPF - 1) / TPF;
		const T hs = (T)0.5 * (T)p.scale;
		const GBuf gw = make_gbuf(p.aux);
#pragma unroll
		for (int b = 0; b < PH; b++) {
			const uint32_t k = tau + b * TPF;
			if ((b + 1) * TPF <= HP || k < (uint32_t)HP) {
				const uint32_t km = k == 0 ? 0u : (uint32_t)N - k;
				const cx<T> zk = ldsf[k], zm = cconj(ldsf[km]);
				const cx<T> w = gb_load<T>(gw, k * (uint32_t)sizeof(cx<T>), 0);
				const cx<T> sS = cadd(zk, zm), d = cmul(w, csub(zk, zm));
				io.str(k, hs * (sS.x + d.y));
				if (2 * k != (uint32_t)N) io.str((uint32_t)N - k, hs * (sS.x - d.y));
			}
		}
	} else if constexpr (staged && !TRANS && POST == OP_DCT2H_POST && ROW) {
		// half-length DCT/DST-II post-map on unit-stride rows, pairs (k, H-k) together (H = N): 2V_k = s - i d and 2V_{H-k} = conj(s + i d) from one
		// pair of LDS reads and one split twiddle (see the R2C split above), then y[k] = Re(c^k 2V_k), y[Nr-k] = -Im(c^k 2V_k) and the same for H-k
		// (reference vkFFT_R2R.h:784 runs a full-length complex FFT instead)
		if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC();
		constexpr int HP = N / 2 + 1, PH = (HP + TPF - 1) / TPF;
		constexpr uint32_t Hc = (uint32_t)N, Nr = 2u * (uint32_t)N;
		const bool dst = p.postOp == OP_DST2H_POST;
		const T sc = (T)p.scale;
		const GBuf gw = make_gbuf(p.aux2), gc = make_gbuf(p.aux);
		auto emit = [&](const uint32_t k, const cx<T> v2, const cx<T> c) { // k in [0, H]
			const cx<T> t = cmul(c, v2);
			io.str(dst ? Nr - 1u - k : k, sc * t.x);
			if (k >= 1u && k < Hc) io.str(dst ? k - 1u : Nr - k, -sc * t.y);
		};
#pragma unroll
		for (int b = 0; b < PH; b++) {
			const uint32_t k = tau + b * TPF;
			if ((b + 1) * TPF <= HP || k < (uint32_t)HP) {
				const uint32_t km = k == 0 ? 0u : Hc - k;
				const cx<T> zk = ldsf[k], zm = cconj(ldsf[km]);
				const cx<T> w = gb_load<T>(gw, k * (uint32_t)sizeof(cx<T>), 0);
				const cx<T> sS = cadd(zk, zm), d = cmul(w, csub(zk, zm));
				emit(k, cx<T>{sS.x + d.y, sS.y - d.x}, gb_load<T>(gc, k * (uint32_t)sizeof(cx<T>), 0));
				if (2 * k != Hc) emit(Hc - k, cx<T>{sS.x - d.y, -(sS.y + d.x)}, gb_load<T>(gc, (Hc - k) * (uint32_t)sizeof(cx<T>), 0));
			}
		}
	} else if constexpr (staged && !TRANS) {
		if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC();
		constexpr int PO = (N + 1 + TPF - 1) / TPF; // R2C even split: N + 1 outputs from the length-N complex FFT
		auto rd = [&](uint32_t a) { return ldsf[a]; };
#pragma unroll
		for (int b = 0; b < PO; b++) {
			const uint32_t k = tau + b * TPF;
			if (k < p.outLen) post_store<T>(p, io, k, colIdx, nat, rd, op_resolve<POST>(p.postOp));
		}
	}
}

template <int N, int FPW, bool COL, bool NEEDS_LDS, int ROWELEMS> __host__ __device__ constexpr int opfft_pitch() {
	if (!NEEDS_LDS) return 1;
	int pitch = COL ? N + 1 : (ROWELEMS > N + 1 ? ROWELEMS : N + 1);
	if (COL) { // lanes run along the FPW columns: column pitch = (32/FPW) * odd spreads a half-wave over all banks
		const int q = FPW >= 32 ? 1 : 32 / FPW;
		while (pitch % (2 * q) != q) pitch++;
	}
	return pitch;
}

template <typename T, typename SCH, int TPF, int FPW, bool COL, int PRE, int POST, bool TRANS>
__global__ void __launch_bounds__(TPF * FPW) opfft_kernel(const PassParams p) {
	constexpr int N = SCH::N;
	static_assert(!TRANS || (COL && (POST == OP_NONE || POST == OP_TWIDDLE_4STEP)), "transposed store: first Four-Step pass of a column tile");
	constexpr int LDSPF = opfft_pitch<N, FPW, COL, (SCH::NS > 1 || TRANS || PRE == OP_DCT3H_PRE || (PRE == OP_C2R_EVEN_PRE && !COL) || POST == OP_R2C_EVEN_POST || POST == OP_DCT2H_POST || POST == OP_DCT1H_POST), MixPad<SCH, TPF, (int)sizeof(cx<T>)>::elems()>();
	constexpr bool waveOnly = !COL && (TPF <= 64) && (64 % TPF == 0); // a row FFT that never straddles wavefronts
	__shared__ cx<T> lds[FPW * LDSPF];
	const uint32_t tid = threadIdx.x;
	const uint32_t f = COL ? tid % FPW : tid / TPF, tau = COL ? tid / FPW : tid % TPF;
	uint32_t wg = p.reverseTiles ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
	const uint32_t tile = wg % p.tilesPerG0;
	wg /= p.tilesPerG0;
	// plain strided C2C: the tile index may run over dim[0] x dim[1] (PassParams::colMerge: companion axes that are not a multiple of the tile width)
	constexpr bool canMerge = COL && !TRANS && PRE == OP_NONE && POST == OP_NONE;
	const bool merge = canMerge && p.colMerge != 0;
	const uint32_t g1 = merge ? 0u : wg % p.dim[1].count, g2 = merge ? wg : wg / p.dim[1].count;
	const uint32_t f0 = tile * FPW, g0 = f0 + f;
	bool valid = g0 < p.dim[0].count;
	int64_t inBase = (int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride, outBase = (int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride;
	int64_t inLane = (int64_t)f * p.dim[0].inStride, outLane = (int64_t)f * p.dim[0].outStride;
	if (merge) {
		const uint32_t c0 = p.dim[0].count, q0 = f0 / c0, q = g0 / c0, r = g0 - q * c0;
		valid = q < p.dim[1].count;
		inBase += (int64_t)q0 * p.dim[1].inStride; outBase += (int64_t)q0 * p.dim[1].outStride;
		inLane = (int64_t)r * p.dim[0].inStride + (int64_t)(q - q0) * p.dim[1].inStride;
		outLane = (int64_t)r * p.dim[0].outStride + (int64_t)(q - q0) * p.dim[1].outStride;
	} else { inBase += (int64_t)f0 * p.dim[0].inStride; outBase += (int64_t)f0 * p.dim[0].outStride; }
	Io32<T> io;
	io.gin = make_gbuf((const char*)p.in + inBase * (int64_t)p.inElemBytes);
	io.gout = make_gbuf((char*)p.out + outBase * (int64_t)p.outElemBytes);
	io.inOff = valid ? (uint32_t)inLane * p.inElemBytes : kGbInvalid;
	io.outOff = valid ? (uint32_t)outLane * p.outElemBytes : kGbInvalid;
	io.inSj = (uint32_t)p.inStrideJ * p.inElemBytes;
	io.outSj = (uint32_t)p.outStrideJ * p.outElemBytes;
	io.set_pad(p);
	const GBuf glut = make_gbuf(p.lut);
	uint32_t colIdx = 0;
	if constexpr (POST == OP_TWIDDLE_4STEP && !TRANS) { if (p.fsColFromDim1) colIdx = g1; else { uint32_t rr; p.fsColDiv.divmod(g0, colIdx, rr); } }
	const uint32_t nat = g0 * p.opStride0 + g1 * p.opStride1;
	cx<T>* ldsf = lds + f * LDSPF;
	op_stage<T, SCH, 0, TPF, PRE, POST, !COL, TRANS>(ldsf, io, glut, tau, waveOnly, p, colIdx, nat);
	if constexpr (TRANS) {
		// first Four-Step pass: every column leaves as ONE contiguous run (Y^T[m][k0], reference vkFFT_ReadWrite.h:1405-1424);
		// the tile is read back from LDS with lanes along the column so that the stores are contiguous, twiddle applied on the way
		VKFFT_SYNC();
		constexpr int NT = TPF * FPW, TOT = FPW * N, PT = (TOT + NT - 1) / NT;
		constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
		const uint32_t remain = p.dim[0].count - f0, nvalid = remain < (uint32_t)FPW ? remain : (uint32_t)FPW;
		const T sc = (T)p.scale;
#pragma unroll
		for (int b = 0; b < PT; b++) {
			const uint32_t idx = tid + b * NT;
			if (TOT % NT == 0 || idx < (uint32_t)TOT) {
				const uint32_t c = idx / (uint32_t)N, k = idx % (uint32_t)N;
				cx<T> v = lds[c * LDSPF + k];
				if constexpr (POST == OP_TWIDDLE_4STEP) {
					uint32_t ci = g1;
					if (!p.fsColFromDim1) { uint32_t rr; p.fsColDiv.divmod(f0 + c, ci, rr); }
					v = cmul(v, twiddle4<T>(p, k * ci));
				}
				if (sc != (T)1) v = cscale(v, sc);
				gb_store<T>(io.gout, c < nvalid ? (c * (uint32_t)p.dim[0].outStride + k * (uint32_t)p.outStrideJ) * ES : kGbInvalid, 0, v);
			}
		}
	}
}

// ---- registry ---------------------------------------------------------------------------------------------------
struct OpfftVariant {
	int n; bool dp; bool col; int pre, post; int rad[5]; int tpf; int fpw; bool trans;
	void (*launch)(const PassParams&, dim3, hipStream_t);
};
template <typename T, typename SCH, int TPF, int FPW, bool COL, int PRE, int POST, bool TRANS> void opfft_launch(const PassParams& prm, dim3 grid, hipStream_t s) {
	hipLaunchKernelGGL((opfft_kernel<T, SCH, TPF, FPW, COL, PRE, POST, TRANS>), grid, dim3(TPF * FPW), 0, s, prm);
}
#define VKFFT_OPX(T, dp, col, pre, post, r0, r1, r2, r3, r4, tpf, fpw, trans) \
	{ (r0) * (r1) * (r2) * (r3) * (r4), dp, col, pre, post, {r0, r1, r2, r3, r4}, tpf, fpw, trans, &opfft_launch<T, MixSched<r0, r1, r2, r3, r4>, tpf, fpw, col, pre, post, trans> },

} // namespace vkfft_mi355x
