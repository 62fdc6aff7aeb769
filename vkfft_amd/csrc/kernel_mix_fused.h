// Fused Four-Step for NON-power-of-two two-factor lengths (round 6): N = n0 * n1 with both factors mixed-radix single-pass lengths (3^10 = 243 x 243,
// 5^8 = 625 x 625, 7^6 = 343 x 343, 11^5 = 121 x 1331, 13^5 = 169 x 2197, ...).  The reference runs these as two or three "axis uploads" through device memory
// (vkFFT_Scheduler.h:2590-2893 axis split, vkFFT_4step.h:31, vkFFT_ReadWrite.h:1405-1476); so did this library until round 5 (emit_multipass: 1.1-1.6 TB/s
// algorithmic, 0.14-0.20 of the HBM roofline).  This kernel is the choreography of kernel_pow2_fused.h — one persistent launch, ticket queues per XCD, the
// intermediate of a chunk in a ring slot that lives in the Infinity Cache, write-through ring stores / memory-side ring loads — around the compile-time radix
// stages of mix_stage.h:
//   ticket (s, r) = tile r of pass A of chunk s  +  tile r of pass B of chunk s - D
//   A tile: TCA neighbouring columns (stride n1) of one transform: HBM -> registers (the first stage's inputs), FFT over n0 (exchange buffer: one LDS column per
//           tile column, lanes along the columns), back through the same LDS columns in natural order, read along the column with the Four-Step twiddle
//           w_N^(k0 * column) on the way, per-column contiguous 16-byte write-through stores into the ring (Y^T[column][k0])
//   B tile: TCB neighbouring k0 (stride n0 in the ring) -> registers, FFT over n1, natural-order stores X[k0 + n0 * k1] straight from the last stage
// Neither factor has to be a multiple of its tile width: the last tile of a phase is partial (lanes of the missing columns get out-of-range offsets), and the
// tickets of a transform number max(tiles of A, tiles of B) — a ticket without an A or B part only keeps the counters uniform.
#pragma once
#include "kernel_pow2_fused.h"
#include "mix_stage.h"

namespace vkfft_mi355x {

// LDS elements per tile column (cf. opfft_pitch): lanes run along the TC columns, column pitch = (32 / TC) * odd spreads a half-wave over all banks
template <int N, int TC> __host__ __device__ constexpr int mixf_pitch() {
	int pitch = N + 1;
	const int q = TC >= 32 ? 1 : 32 / TC;
	while (pitch % (2 * q) != q) pitch++;
	return pitch;
}
template <typename T, typename SA, int TCA, typename SB, int TCB> __host__ __device__ constexpr int mixf_lds_bytes() {
	constexpr int a = TCA * mixf_pitch<SA::N, TCA>(), b = TCB * mixf_pitch<SB::N, TCB>();
	return (a > b ? a : b) * (int)sizeof(cx<T>) + 64;
}
template <typename T, typename SA, int TPFA, int TCA, typename SB, int TCB> __host__ __device__ constexpr int mixf_wg_per_cu() {
	constexpr int nt = TPFA * TCA;
	int w = 163840 / mixf_lds_bytes<T, SA, TCA, SB, TCB>();
	if (w > 2048 / nt) w = 2048 / nt;
	return w > 4 ? 4 : w < 1 ? 1 : w;
}

// Hooks of the chirp-z transform (BLUE = 1 instances; reference vkFFT_Bluestein.h:32,201, multi-upload form vkFFT_Scheduler.h:2406-2578): a prime length N runs as
// TWO launches of this kernel on a padded length M = n0 * n1 >= 2N - 1 —
//   launch 1: forward FFT_M of x[n] * conj(chirp[n]) (n < N, zero beyond: those elements are never requested), every output times FFT(chirp)[k] / M on its way out
//   launch 2: inverse FFT_M, outputs k < N times conj(chirp[k]) to the caller's buffer (the rest is never stored)
// — instead of the five (three) separate passes of the Bluestein plans of round 2.  All indices are natural positions of the length-M sequence.
struct MixFusedOps {
	const void* chirp; const void* bhat; // chirp[n] = exp(+i pi n^2 / N), n < N; bhat[k] = FFT_M(chirp extended)[k] / M (planner: make_bluestein_tables)
	uint32_t blueN, preBlue, postMul, postBlue, bsSwapIn, bsSwapOut;
};

// MODE bit 1: non-temporal hint on the HBM side
template <typename T, typename SA, int TPFA, int TCA, typename SB, int TPFB, int TCB, int MODE, int BLUE>
__global__ void __launch_bounds__(TPFA * TCA) mix_fused_kernel(const FusedParams p, const MixFusedOps o) {
	constexpr int NA = SA::N, NBN = SB::N; // n0 (first factor: strided columns of the input), n1 (second factor)
	constexpr int NT = TPFA * TCA;
	static_assert(NT == TPFB * TCB, "both phases run on the same workgroup shape");
	static_assert(SA::NS > 1 && SB::NS > 1, "factors of two or more stages");
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	constexpr int AUX_SC = 16, AUX_ST = 16;     // ring: memory-side loads, write-through stores (no XCD's L2 ever holds a ring line: kernel_pow2_fused.h)
	constexpr int AUX_HBM = (MODE & 2) ? 2 : 0; // streamed side: non-temporal hint
	constexpr int PITA = mixf_pitch<NA, TCA>(), PITB = mixf_pitch<NBN, TCB>();
	constexpr int LDSN = TCA * PITA > TCB * PITB ? TCA * PITA : TCB * PITB;
	__shared__ cx<T> lds[LDSN];
	__shared__ uint32_t sTicket[2], sOkA[2], sOkB[2];
	const uint32_t tid = threadIdx.x;
	const uint32_t TPC = p.tpc, tiles = p.tiles;
	const uint32_t doneA = kFusedCtrDone, doneB = kFusedCtrDone + p.C;
	constexpr uint64_t nPts = ((uint64_t)NA * NBN + 1ull) & ~1ull; // elements per transform in a ring slot (even: every transform of the ring starts 16-byte aligned)
	constexpr uint32_t kNone = 0xffffffffu;
	const uint32_t Q = p.Q;
	uint32_t q = Q > 1 ? fused_xcc_id() % Q : 0u, tried = 0;
	uint32_t Cq = (p.C + Q - 1u - q) / Q;
	uint32_t totq = Cq ? (Cq + p.D) * TPC : 0u;
	auto depA = [&](uint32_t s) -> uint32_t { return (s < Cq && s >= p.NS) ? doneB + q + Q * (s - p.NS) : kNone; };
	auto depB = [&](uint32_t s) -> uint32_t { return (s >= p.D && s - p.D < Cq) ? doneA + q + Q * (s - p.D) : kNone; };
	uint32_t pending = kNone; // counter this workgroup still owes a bump: its last A tile's stores are in flight (thread 0 only)
	if (tid == 0) {
		const uint32_t t0 = VKFFT_ATOMIC_ADD_U32(p.ctr + kFusedCtrTicket + 32u * q, 1u), s0 = t0 / TPC;
		const uint32_t dA = t0 < totq ? depA(s0) : kNone, dB = t0 < totq ? depB(s0) : kNone;
		sTicket[0] = t0;
		sOkA[0] = dA == kNone ? 1u : (uint32_t)(VKFFT_ATOMIC_LOAD_U32(p.ctr + dA) >= TPC);
		sOkB[0] = dB == kNone ? 1u : (uint32_t)(VKFFT_ATOMIC_LOAD_U32(p.ctr + dB) >= TPC);
	}
	uint32_t it = 0;
	const uint32_t cA = tid % (uint32_t)TCA, tauA = tid / (uint32_t)TCA; // this thread's column of an A tile, its butterfly lane
	const uint32_t cB = tid % (uint32_t)TCB, tauB = tid / (uint32_t)TCB;
	cx<T>* const colA = lds + cA * PITA;
	cx<T>* const colB = lds + cB * PITB;
	for (;;) {
		VKFFT_SYNC(); // S1: ticket visible; exchange buffer free again
		const uint32_t t = sTicket[it];
		if (t >= totq) {
			// this queue is drained: help the next one, leave when every queue is (completion must not depend on where workgroups run)
			if (++tried >= Q) break;
			VKFFT_VMEM_DRAIN();
			VKFFT_SYNC();
			q = q + 1u == Q ? 0u : q + 1u;
			Cq = (p.C + Q - 1u - q) / Q;
			totq = Cq ? (Cq + p.D) * TPC : 0u;
			if (tid == 0) {
				fused_publish(p.ctr, pending);
				const uint32_t t0 = VKFFT_ATOMIC_ADD_U32(p.ctr + kFusedCtrTicket + 32u * q, 1u), s0 = t0 / TPC;
				const uint32_t dA = t0 < totq ? depA(s0) : kNone, dB = t0 < totq ? depB(s0) : kNone;
				sTicket[it] = t0;
				sOkA[it] = dA == kNone ? 1u : (uint32_t)(VKFFT_ATOMIC_LOAD_U32(p.ctr + dA) >= TPC);
				sOkB[it] = dB == kNone ? 1u : (uint32_t)(VKFFT_ATOMIC_LOAD_U32(p.ctr + dB) >= TPC);
			}
			continue;
		}
		const uint32_t okA = sOkA[it], okB = sOkB[it];
		it ^= 1u;
		VKFFT_OPAQUE_ZERO(oz); // (the tables are the same for every tile: an opaque zero in their base keeps the loads inside the loop)
		const GBuf gtw = make_gbuf((const char*)p.tw4 + oz);
		const uint32_t s = t / TPC, r = t - s * TPC;
		const uint32_t f = r / tiles, ti = r - f * tiles;
		const bool hasA = s < Cq, hasB = s >= p.D && s - p.D < Cq;
		uint32_t nextT = 0, nfA = TPC, nfB = TPC;
		if (tid == 0) nextT = VKFFT_ATOMIC_ADD_U32(p.ctr + kFusedCtrTicket + 32u * q, 1u);
		const uint32_t sB = s - p.D, chB = q + Q * sB;
		const uint32_t bB = ((p.reverse ? p.C - 1u - chB : chB) << p.logG) + f;
		const uint32_t k00 = ti * (uint32_t)TCB;
		const bool liveB = hasB && bB < p.batch && k00 < (uint32_t)NA;
		const uint32_t chA = q + Q * s; // chunk in processing order (counters, ring slot)
		{
			// ---- A: FFT over n0 of TCA neighbouring columns (stride n1), twiddle, per-column contiguous stores into the ring
			const uint32_t b = ((p.reverse ? p.C - 1u - chA : chA) << p.logG) + f;
			const uint32_t col0 = ti * (uint32_t)TCA;
			const bool live = hasA && b < p.batch && col0 < (uint32_t)NBN; // (the last chunk may be partial, a phase may have fewer tiles than the ticket count)
			const char* const sbase = (const char*)p.scratch + ((uint64_t)(((q * p.NS + s % p.NS) << p.logG) + f) * nPts) * ES;
			const GBuf gin = make_gbuf((const cx<T>*)p.in + ((int64_t)(live ? b : 0u) * p.inBatchStride + (live ? col0 : 0u)));
			const uint32_t laneIn = (live && col0 + cA < (uint32_t)NBN) ? cA * ES : kGbInvalid;
			constexpr int R0 = SA::rad[0], NB0 = NA / R0, P0 = (NB0 + TPFA - 1) / TPFA;
			cx<T> xa[P0][R0];
			cx<T> ca[BLUE ? P0 : 1][BLUE ? R0 : 1]; // (chirp-z first launch: the chirp of every input position)
			const bool preBlue = BLUE && o.preBlue;
			if (preBlue) {
				if constexpr (BLUE != 0) {
					const GBuf gch = make_gbuf((const char*)o.chirp + oz);
#pragma unroll
					for (int bb = 0; bb < P0; bb++) {
						const uint32_t t2 = tauA + (uint32_t)(bb * TPFA);
						if ((bb + 1) * TPFA <= NB0 || t2 < (uint32_t)NB0) {
#pragma unroll
							for (int i = 0; i < R0; i++) {
								const uint32_t n = (t2 + (uint32_t)(i * NB0)) * (uint32_t)NBN + col0 + cA; // natural position in the padded sequence
								const bool in = laneIn != kGbInvalid && n < o.blueN;
								xa[bb][i] = gb_load_x<T, AUX_HBM>(gin, in ? laneIn + (t2 + (uint32_t)(i * NB0)) * (uint32_t)NBN * ES : kGbInvalid, 0);
								ca[bb][i] = gb_load<T>(gch, in ? n * ES : kGbInvalid, 0);
							}
						}
					}
				}
			} else {
#pragma unroll
			for (int bb = 0; bb < P0; bb++) {
				const uint32_t t2 = tauA + (uint32_t)(bb * TPFA);
				if ((bb + 1) * TPFA <= NB0 || t2 < (uint32_t)NB0) {
#pragma unroll
					for (int i = 0; i < R0; i++) xa[bb][i] = gb_load_x<T, AUX_HBM>(gin, laneIn + t2 * (uint32_t)NBN * ES, (uint32_t)(i * NB0) * (uint32_t)NBN * ES);
				}
			}
			}
			VKFFT_VMEM_DRAIN(); // this tile's loads have landed, the previous ticket's stores are acknowledged, the next ticket is here
			if (tid == 0) {
				sTicket[it] = nextT;
				const uint32_t sN = nextT / TPC, dA = nextT < totq ? depA(sN) : kNone, dB = nextT < totq ? depB(sN) : kNone;
				if (dA != kNone) nfA = VKFFT_ATOMIC_LOAD_U32(p.ctr + dA); // consumed at the end of this iteration
				if (dB != kNone) nfB = VKFFT_ATOMIC_LOAD_U32(p.ctr + dB);
			}
			VKFFT_SYNC(); // S2
			if (tid == 0 && pending != kNone) { (void)VKFFT_ATOMIC_ADD_U32(p.ctr + pending, 1u); pending = kNone; }
			if (hasA && !okA) fused_wait(p.ctr + depA(s), TPC);
			if (live) {
				if (p.swapIn) {
#pragma unroll
					for (int bb = 0; bb < P0; bb++) {
#pragma unroll
						for (int i = 0; i < R0; i++) xa[bb][i] = cswap(xa[bb][i]);
					}
				}
				if (preBlue) {
					if constexpr (BLUE != 0) {
						const bool sw = o.bsSwapIn != 0;
#pragma unroll
						for (int bb = 0; bb < P0; bb++) {
#pragma unroll
							for (int i = 0; i < R0; i++) xa[bb][i] = cmulc(sw ? cswap(xa[bb][i]) : xa[bb][i], ca[bb][i]);
						}
					}
				}
				const GBuf glutA = make_gbuf((const char*)p.lutA + oz);
				mc_stage<T, SA, 0, TPFA, 1, false, false, true>(colA, glutA, tauA, false, McRegs<T, R0>{xa}, [&](uint32_t t2, uint32_t c2, cx<T> v) { colA[t2 + c2] = v; });
				VKFFT_SYNC();
				// the tile leaves as ONE contiguous run of the ring (its columns are neighbours there: Y^T[col0 + c][k0]); 16 bytes per lane, the twiddle on the way
				const uint32_t ncols = (uint32_t)NBN - col0 < (uint32_t)TCA ? (uint32_t)NBN - col0 : (uint32_t)TCA, lim = ncols * (uint32_t)NA;
				const GBuf gs = make_gbuf(sbase + (uint64_t)col0 * NA * ES);
				const uint32_t loMask = (1u << p.fsLoBits) - 1u, hiBase = (loMask + 1u) * ES;
				auto elem = [&](uint32_t e) -> cx<T> {
					const uint32_t c2 = e / (uint32_t)NA, k = e - c2 * (uint32_t)NA, x = k * (col0 + c2);
					const cx<T> w = cmul(gb_load<T>(gtw, (x & loMask) * ES, 0), gb_load<T>(gtw, (x >> p.fsLoBits) * ES, hiBase));
					return cmul(lds[c2 * PITA + k], w);
				};
				if constexpr (sizeof(T) == 4) {
					constexpr int PT = ((TCA * NA + 1) / 2 + NT - 1) / NT;
#pragma unroll
					for (int i = 0; i < PT; i++) {
						const uint32_t e0 = 2u * (tid + (uint32_t)(i * NT));
						if (e0 + 1u < lim) gb_store2_x<T, AUX_ST>(gs, e0 * ES, elem(e0), elem(e0 + 1u));
						else if (e0 < lim) gb_store_x<T, AUX_ST>(gs, e0 * ES, 0, elem(e0));
					}
				} else {
					constexpr int PT = (TCA * NA + NT - 1) / NT;
#pragma unroll
					for (int i = 0; i < PT; i++) {
						const uint32_t e0 = tid + (uint32_t)(i * NT);
						if (e0 < lim) gb_store_x<T, AUX_ST>(gs, e0 * ES, 0, elem(e0));
					}
				}
			}
			if (hasA) pending = doneA + chA;
		}
		if (hasB) {
			// ---- B: FFT over n1 (stride n0 in the ring) of TCB neighbouring k0, natural-order store X[k0 + n0 * k1]
			const char* const sbaseB = (const char*)p.scratch + ((uint64_t)(((q * p.NS + sB % p.NS) << p.logG) + f) * nPts) * ES;
			const GBuf gsB = make_gbuf(sbaseB + (uint64_t)(liveB ? k00 : 0u) * ES);
			const GBuf gout = make_gbuf((cx<T>*)p.out + ((int64_t)(liveB ? bB : 0u) * p.outBatchStride + (liveB ? k00 : 0u)));
			const uint32_t laneB = (liveB && k00 + cB < (uint32_t)NA) ? cB * ES : kGbInvalid;
			if (!okB) fused_wait(p.ctr + depB(s), TPC); // rare (okB was sampled one ticket ago: ordered before the loads by S1)
			constexpr int R0 = SB::rad[0], NB0 = NBN / R0, P0 = (NB0 + TPFB - 1) / TPFB;
			cx<T> xb[P0][R0];
#pragma unroll
			for (int bb = 0; bb < P0; bb++) {
				const uint32_t t2 = tauB + (uint32_t)(bb * TPFB);
				if ((bb + 1) * TPFB <= NB0 || t2 < (uint32_t)NB0) {
#pragma unroll
					for (int i = 0; i < R0; i++) xb[bb][i] = gb_load_x<T, AUX_SC>(gsB, laneB + t2 * (uint32_t)NA * ES, (uint32_t)(i * NB0) * (uint32_t)NA * ES);
				}
			}
			VKFFT_VMEM_DRAIN(); // the tile is in registers; the A part's ring stores are acknowledged
			VKFFT_SYNC();       // S3: ... in every wave; the exchange buffer is free
			if (tid == 0) {
				(void)VKFFT_ATOMIC_ADD_U32(p.ctr + doneB + chB, 1u); // release the ring slot
				fused_publish(p.ctr, pending);
			}
			if (liveB) {
				const GBuf glutB = make_gbuf((const char*)p.lutB + oz);
				const T sc = (T)p.scale;
				const bool swO = p.swapOut != 0;
				if constexpr (BLUE != 0) {
					const GBuf gbh = make_gbuf((const char*)o.bhat + oz), gch = make_gbuf((const char*)o.chirp + oz);
					const bool postMul = o.postMul != 0, postBlue = o.postBlue != 0, swB = o.bsSwapOut != 0;
					mc_stage<T, SB, 0, TPFB, 1, false, false, false>(colB, glutB, tauB, false, McRegs<T, R0>{xb}, [&](uint32_t t2, uint32_t c2, cx<T> v) {
						const uint32_t k = k00 + cB + (t2 + c2) * (uint32_t)NA; // natural position of this output
						const bool lane = laneB != kGbInvalid;
						if (swO) v = cswap(v);
						if (postMul) v = cmul(v, gb_load<T>(gbh, lane ? k * ES : kGbInvalid, 0));
						bool st = lane;
						if (postBlue) {
							st = lane && k < o.blueN;
							v = cmulc(v, gb_load<T>(gch, st ? k * ES : kGbInvalid, 0));
							if (swB) v = cswap(v);
						}
						if (sc != (T)1) v = cscale(v, sc);
						gb_store_x<T, AUX_HBM>(gout, st ? laneB + (t2 + c2) * (uint32_t)NA * ES : kGbInvalid, 0, v);
					});
				} else
				mc_stage<T, SB, 0, TPFB, 1, false, false, false>(colB, glutB, tauB, false, McRegs<T, R0>{xb}, [&](uint32_t t2, uint32_t c2, cx<T> v) {
					if (swO) v = cswap(v);
					if (sc != (T)1) v = cscale(v, sc);
					gb_store_x<T, AUX_HBM>(gout, laneB + t2 * (uint32_t)NA * ES, c2 * (uint32_t)NA * ES, v);
				});
			}
		}
		if (tid == 0) { sOkA[it] = nfA >= TPC; sOkB[it] = nfB >= TPC; }
	}
	// ---- exit: publish the last A tile, then the last workgroup out resets the counters for the next launch
	VKFFT_VMEM_DRAIN();
	VKFFT_SYNC();
	if (tid == 0) {
		fused_publish(p.ctr, pending);
		VKFFT_VMEM_DRAIN();
		sOkA[0] = VKFFT_ATOMIC_ADD_U32(p.ctr + kFusedCtrExit, 1u) == gridDim.x - 1u;
	}
	VKFFT_SYNC();
	if (sOkA[0]) {
		for (uint32_t i = tid; i < kFusedCtrDone + 2u * p.C; i += NT) p.ctr[i] = 0u;
	}
}

struct MixFusedVariant {
	uint64_t n; int n0, n1; bool dp; int radA[5], radB[5]; int tpfa, tca, tpfb, tcb, threads, wgPerCu;
	int blue; // 1: the instance carries the chirp-z hooks (padded lengths of the two-launch Bluestein plan)
	void (*launch)(const FusedParams&, const MixFusedOps&, dim3, hipStream_t);
	const void* fn;
};
template <typename T, typename SA, int TPFA, int TCA, typename SB, int TPFB, int TCB, int MODE, int BLUE> void mix_fused_launch(const FusedParams& prm, const MixFusedOps& ops, dim3 grid, hipStream_t s) {
	hipLaunchKernelGGL((mix_fused_kernel<T, SA, TPFA, TCA, SB, TPFB, TCB, MODE, BLUE>), grid, dim3(TPFA * TCA), 0, s, prm, ops);
}

} // namespace vkfft_mi355x
