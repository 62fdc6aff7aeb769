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

// entries of the two-level Four-Step table of N points (planner: lo = (ceil_log2(N) + 1) / 2 low bits)
__host__ __device__ constexpr int mixf_fs_lobits(uint64_t n) { int l = 0; while ((1ull << l) < n) l++; return (l + 1) / 2; }
__host__ __device__ constexpr int mixf_fs_entries(uint64_t n) { return (int)((1ull << mixf_fs_lobits(n)) + ((n + (1ull << mixf_fs_lobits(n)) - 1) >> mixf_fs_lobits(n))); }
// LDS elements per tile column (cf. opfft_pitch): lanes run along the TC columns, column pitch = (32 / TC) * odd spreads a half-wave over all banks
template <int N, int TC> __host__ __device__ constexpr int mixf_pitch() {
	int pitch = N + 1;
	const int q = TC >= 32 ? 1 : 32 / TC;
	while (pitch % (2 * q) != q) pitch++;
	return pitch;
}
template <typename T, typename SA, int TCA, typename SB, int TCB> __host__ __device__ constexpr int mixf_tile_bytes() {
	constexpr int a = TCA * mixf_pitch<SA::N, TCA>(), b = TCB * mixf_pitch<SB::N, TCB>();
	return (a > b ? a : b) * (int)sizeof(cx<T>) + 64;
}
// the tables move into LDS where they do not cost a workgroup per CU: stage twiddles of both factors (TWL), the two-level Four-Step table (FSL)
template <typename T, typename SA, int TPFA, int TCA, typename SB, int TCB> __host__ __device__ constexpr int mixf_wg_by_lds(int bytes) {
	int w = 163840 / bytes;
	if (w > 2048 / (TPFA * TCA)) w = 2048 / (TPFA * TCA);
	return w > 4 ? 4 : w < 1 ? 1 : w;
}
template <typename T, typename SA, int TPFA, int TCA, typename SB, int TCB> __host__ __device__ constexpr bool mixf_twl() {
	constexpr int tile = mixf_tile_bytes<T, SA, TCA, SB, TCB>(), tw = (SA::lutOff(SA::NS) + SB::lutOff(SB::NS)) * (int)sizeof(cx<T>);
	return tile + tw <= 163840 && mixf_wg_by_lds<T, SA, TPFA, TCA, SB, TCB>(tile + tw) == mixf_wg_by_lds<T, SA, TPFA, TCA, SB, TCB>(tile);
}
template <typename T, typename SA, int TPFA, int TCA, typename SB, int TCB> __host__ __device__ constexpr bool mixf_fsl() {
	constexpr int tile = mixf_tile_bytes<T, SA, TCA, SB, TCB>() + (mixf_twl<T, SA, TPFA, TCA, SB, TCB>() ? (SA::lutOff(SA::NS) + SB::lutOff(SB::NS)) * (int)sizeof(cx<T>) : 0);
	constexpr int fs = mixf_fs_entries((uint64_t)SA::N * SB::N) * (int)sizeof(cx<T>);
	return fs <= 16384 && tile + fs <= 163840 && mixf_wg_by_lds<T, SA, TPFA, TCA, SB, TCB>(tile + fs) == mixf_wg_by_lds<T, SA, TPFA, TCA, SB, TCB>(tile);
}
template <typename T, typename SA, int TPFA, int TCA, typename SB, int TCB> __host__ __device__ constexpr int mixf_lds_bytes() {
	return mixf_tile_bytes<T, SA, TCA, SB, TCB>() + (mixf_twl<T, SA, TPFA, TCA, SB, TCB>() ? (SA::lutOff(SA::NS) + SB::lutOff(SB::NS)) * (int)sizeof(cx<T>) : 0)
	     + (mixf_fsl<T, SA, TPFA, TCA, SB, TCB>() ? mixf_fs_entries((uint64_t)SA::N * SB::N) * (int)sizeof(cx<T>) : 0);
}
template <typename T, typename SA, int TPFA, int TCA, typename SB, int TCB> __host__ __device__ constexpr int mixf_wg_per_cu() {
	return mixf_wg_by_lds<T, SA, TPFA, TCA, SB, TCB>(mixf_lds_bytes<T, SA, TPFA, TCA, SB, TCB>());
}

// stage twiddles of a factor: staged in LDS once per persistent workgroup (every look-up of the first version went to L2: with two or three workgroups per CU the
// dependent trips table -> butterfly -> table were the ticket time), or through the buffer path where the LDS copy would not fit beside the tile
template <typename T> struct MfTwLds { const cx<T>* t; __device__ inline cx<T> operator()(uint32_t i) const { return t[i]; } };
template <typename T> struct MfTwGlobal { GBuf g; __device__ inline cx<T> operator()(uint32_t i) const { return gb_load<T>(g, i * (uint32_t)sizeof(cx<T>), 0); } };

// the stages of mix_stage.h for one column of a tile (exchange buffer = the column's own LDS run, natural order between the stages), inputs of the first stage in
// registers, twiddles through TW, the butterflies of a thread one after the other (scheduling fence: interleaved, P butterflies of radix 9 ... 16 with their
// twiddles in flight took 212-256 registers)
template <typename T, typename SCH, int SI, int TPF, bool SL, typename TW, typename IN, typename OUT>
__device__ inline void mf_stage(cx<T>* ldsf, const TW& tw, const uint32_t tau, const IN& in, const OUT& out) {
	constexpr int N = SCH::N, R = SCH::rad[SI], NB = N / R, P = (NB + TPF - 1) / TPF, S = SCH::S(SI);
	constexpr bool first = SI == 0, last = SI == SCH::NS - 1;
	cx<T> x[P][R];
#pragma unroll
	for (int b = 0; b < P; b++) {
		const uint32_t t = tau + (uint32_t)(b * TPF);
		if ((b + 1) * TPF <= NB || t < (uint32_t)NB) {
#pragma unroll
			for (int i = 0; i < R; i++) {
				if constexpr (first) x[b][i] = in.x[b][i];
				else x[b][i] = ldsf[t + (uint32_t)(i * NB)];
			}
		}
	}
	if constexpr ((!first && !last) || (last && SL)) VKFFT_SYNC(); // every input is in registers before the buffer is overwritten
#pragma unroll
	for (int b = 0; b < P; b++) {
		const uint32_t t = tau + (uint32_t)(b * TPF);
		if ((b + 1) * TPF <= NB || t < (uint32_t)NB) {
			const uint32_t s = t % (uint32_t)S;
			if constexpr (!first) {
				constexpr int LO = SCH::lutOff(SI);
#pragma unroll
				for (int i = 1; i < R; i++) x[b][i] = cmul(x[b][i], tw((uint32_t)(LO + (i - 1) * S) + s));
			}
			dft<R, T>(x[b]);
			if constexpr (last) {
#pragma unroll
				for (int k = 0; k < R; k++) out(t, (uint32_t)(k * S), x[b][k]);
			} else {
				const uint32_t ob = (t - s) * (uint32_t)R + s;
#pragma unroll
				for (int k = 0; k < R; k++) ldsf[ob + (uint32_t)(k * S)] = x[b][k];
			}
		}
		if (b + 1 < P) VKFFT_SCHED_FENCE();
	}
	if constexpr (!last) {
		VKFFT_SYNC();
		mf_stage<T, SCH, SI + 1 < SCH::NS ? SI + 1 : SI, TPF, SL>(ldsf, tw, tau, in, out);
	}
}
template <typename SCH> __host__ __device__ constexpr int mixf_lut_total() { return SCH::lutOff(SCH::NS); }

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
// WGC: workgroups per CU the instance's REGISTER budget is set for through __launch_bounds__ (at most four wavefronts per SIMD: 128 registers).  Measured both ways
// (profiles/r06_mix_fused_register_budgets.jsonl): budgets of 64-85 registers, which would let two workgroups of 650-1000 threads share a CU, cost 20-144 bytes of
// scratch and 10-40 % (a reload from scratch drains the wave's one wait counter); the large factors run ONE workgroup per CU, and with the budget of
// one (WGC = 1: 168 registers) 3^12 gained 11 % over the 128-register build: the compiler keeps more of a tile's requests in flight
template <typename T, typename SA, int TPFA, int TCA, typename SB, int TCB, int WGC> __host__ __device__ constexpr int mixf_wgc() {
	return WGC < mixf_wg_per_cu<T, SA, TPFA, TCA, SB, TCB>() ? WGC : mixf_wg_per_cu<T, SA, TPFA, TCA, SB, TCB>();
}
// PIPE = 1: software-pipelined tickets (the other tile's memory traffic in flight while one computes: two tiles' first-stage inputs live at a time);
// PIPE = 0: one tile at a time (12-20 registers fewer: for the shapes whose register count decides whether TWO workgroups share a CU — they overlap each other instead)
template <typename T, typename SA, int TPFA, int TCA, typename SB, int TPFB, int TCB, int MODE, int BLUE, int WGC, int PIPE = 1>
__global__ void __launch_bounds__(TPFA * TCA, (mixf_wgc<T, SA, TPFA, TCA, SB, TCB, WGC>() * TPFA * TCA + 255) / 256 > (PIPE ? 4 : 8) ? (PIPE ? 4 : 8) : (mixf_wgc<T, SA, TPFA, TCA, SB, TCB, WGC>() * TPFA * TCA + 255) / 256)
mix_fused_kernel(const FusedParams p, const MixFusedOps o) {
	constexpr int NA = SA::N, NBN = SB::N; // n0 (first factor: strided columns of the input), n1 (second factor)
	constexpr int NT = TPFA * TCA;
	static_assert(NT == TPFB * TCB, "both phases run on the same workgroup shape");
	static_assert(SA::NS > 1 && SB::NS > 1, "factors of two or more stages");
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	constexpr int AUX_SC = 16, AUX_ST = 16;     // ring: memory-side loads, write-through stores (no XCD's L2 ever holds a ring line: kernel_pow2_fused.h)
	constexpr int AUX_HBM = (MODE & 2) ? 2 : 0; // streamed side: non-temporal hint
	constexpr int AUX_HBM_ST = (MODE & 8) ? 0 : AUX_HBM; // (MODE bit 3, tuning: plain stores on the HBM side)
	// Ring layout Y^T[column][k0] with the column pitch rounded up to 16 elements: a B tile's TCB neighbouring k0 of one column are then ONE aligned 64- or 128-byte
	// segment.  With the dense pitch n0 (odd for every length here) each such segment straddled two lines of the memory side, and the kernel ran at the same
	// 1.2-2.0 TB/s with one workgroup per CU as with four: bound by ring requests, not by latency (profiles/r06_mix_fused_launch_shape_sweep.jsonl)
	constexpr uint32_t NAP = ((uint32_t)NA + 15u) & ~15u;
	constexpr int PITA = mixf_pitch<NA, TCA>(), PITB = mixf_pitch<NBN, TCB>();
	constexpr int LDSN = TCA * PITA > TCB * PITB ? TCA * PITA : TCB * PITB;
	constexpr bool TWL = mixf_twl<T, SA, TPFA, TCA, SB, TCB>(), FSL = mixf_fsl<T, SA, TPFA, TCA, SB, TCB>();
	constexpr int LUTA = mixf_lut_total<SA>(), LUTB = mixf_lut_total<SB>(), FSE = mixf_fs_entries((uint64_t)NA * NBN);
	__shared__ cx<T> lds[LDSN + (TWL ? LUTA + LUTB : 0) + (FSL ? FSE : 0)];
	__shared__ uint32_t sTicket[2], sOkA[2], sOkB[2];
	const uint32_t tid = threadIdx.x;
	cx<T>* const twA = lds + LDSN;
	cx<T>* const twB = twA + (TWL ? LUTA : 0);
	cx<T>* const fsT = twB + (TWL ? LUTB : 0);
	if constexpr (TWL) {
		for (uint32_t i = tid; i < (uint32_t)LUTA; i += NT) twA[i] = ((const cx<T>*)p.lutA)[i];
		for (uint32_t i = tid; i < (uint32_t)LUTB; i += NT) twB[i] = ((const cx<T>*)p.lutB)[i];
	}
	if constexpr (FSL) { for (uint32_t i = tid; i < (uint32_t)FSE; i += NT) fsT[i] = ((const cx<T>*)p.tw4)[i]; }
	const uint32_t TPC = p.tpc, tiles = p.tiles;
	const uint32_t doneA = kFusedCtrDone, doneB = kFusedCtrDone + p.C;
	constexpr uint64_t nPts = (uint64_t)NAP * NBN; // elements per transform in a ring slot (planner: build_mix_fused_pass computes the same)
	constexpr uint32_t kNone = 0xffffffffu;
	const uint32_t Q = p.Q;
	uint32_t q = Q > 1 ? fused_xcc_id() % Q : 0u, tried = 0;
	uint32_t Cq = (p.C + Q - 1u - q) / Q;
	uint32_t totq = Cq ? (Cq + p.D) * TPC : 0u;
	auto depA = [&](uint32_t s) -> uint32_t { return (s < Cq && s >= p.NS) ? doneB + q + Q * (s - p.NS) : kNone; };
	auto depB = [&](uint32_t s) -> uint32_t { return (s >= p.D && s - p.D < Cq) ? doneA + q + Q * (s - p.D) : kNone; };
	auto draw = [&](uint32_t slot) { // thread 0: next ticket of queue q and the state of ITS dependencies
		const uint32_t t0 = VKFFT_ATOMIC_ADD_U32(p.ctr + kFusedCtrTicket + 32u * q, 1u), s0 = t0 / TPC;
		const uint32_t dA = t0 < totq ? depA(s0) : kNone, dB = t0 < totq ? depB(s0) : kNone;
		sTicket[slot] = t0;
		sOkA[slot] = dA == kNone ? 1u : (uint32_t)(VKFFT_ATOMIC_LOAD_U32(p.ctr + dA) >= TPC);
		sOkB[slot] = dB == kNone ? 1u : (uint32_t)(VKFFT_ATOMIC_LOAD_U32(p.ctr + dB) >= TPC);
	};
	const uint32_t cA = tid % (uint32_t)TCA, tauA = tid / (uint32_t)TCA; // this thread's column of an A tile, its butterfly lane
	const uint32_t cB = tid % (uint32_t)TCB, tauB = tid / (uint32_t)TCB;
	cx<T>* const colA = lds + cA * PITA;
	cx<T>* const colB = lds + cB * PITB;
	constexpr int R0A = SA::rad[0], NB0A = NA / R0A, P0A = (NB0A + TPFA - 1) / TPFA;
	constexpr int R0B = SB::rad[0], NB0B = NBN / R0B, P0B = (NB0B + TPFB - 1) / TPFB;
	cx<T> xa[P0A][R0A];                         // the A tile while it travels: the inputs of this thread's first-stage butterflies
	cx<T> ca[BLUE ? P0A : 1][BLUE ? R0A : 1];   // (chirp-z first launch: the chirp of every input position)
	const bool preBlue = BLUE && o.preBlue;
	// the A tile of ticket tt (of the CURRENT queue), requested from HBM: no branch around the requests — lanes of a ticket without an A part get out-of-range
	// offsets, which cost no traffic
	auto requestA = [&](uint32_t tt) {
		const uint32_t s = tt / TPC, r = tt - s * TPC;
		const uint32_t f = r / tiles, ti = r - f * tiles;
		const uint32_t chA = q + Q * s;
		const uint32_t b = ((p.reverse ? p.C - 1u - chA : chA) << p.logG) + f;
		const uint32_t col0 = ti * (uint32_t)TCA;
		const bool live = tt < totq && s < Cq && b < p.batch && col0 < (uint32_t)NBN;
		const GBuf gin = make_gbuf((const cx<T>*)p.in + ((int64_t)(live ? b : 0u) * p.inBatchStride + (live ? col0 : 0u)));
		const uint32_t laneIn = (live && col0 + cA < (uint32_t)NBN) ? cA * ES : kGbInvalid;
		VKFFT_OPAQUE_ZERO(oq);
		if (preBlue) {
			if constexpr (BLUE != 0) {
				const GBuf gch = make_gbuf((const char*)o.chirp + oq);
#pragma unroll
				for (int bb = 0; bb < P0A; bb++) {
					const uint32_t t2 = tauA + (uint32_t)(bb * TPFA);
					if ((bb + 1) * TPFA <= NB0A || t2 < (uint32_t)NB0A) {
						const uint32_t n0row = t2 * (uint32_t)NBN + col0 + cA; // natural position in the padded sequence of input i: n0row + i * NB0 * n1
#pragma unroll
						for (int i = 0; i < R0A; i++) {
							// (the row step rides in the scalar offset, which the range check does not see: a lane is masked through its vector offset alone)
							const bool in = laneIn != kGbInvalid && n0row + (uint32_t)(i * NB0A) * (uint32_t)NBN < o.blueN;
							xa[bb][i] = gb_load_x<T, AUX_HBM>(gin, in ? laneIn + t2 * (uint32_t)NBN * ES : kGbInvalid, (uint32_t)(i * NB0A) * (uint32_t)NBN * ES);
							ca[bb][i] = gb_load<T>(gch, in ? n0row * ES : kGbInvalid, (uint32_t)(i * NB0A) * (uint32_t)NBN * ES);
						}
					}
				}
			}
		} else {
#pragma unroll
			for (int bb = 0; bb < P0A; bb++) {
				const uint32_t t2 = tauA + (uint32_t)(bb * TPFA);
				if ((bb + 1) * TPFA <= NB0A || t2 < (uint32_t)NB0A) {
#pragma unroll
					for (int i = 0; i < R0A; i++) xa[bb][i] = gb_load_x<T, AUX_HBM>(gin, laneIn + t2 * (uint32_t)NBN * ES, (uint32_t)(i * NB0A) * (uint32_t)NBN * ES + oq);
				}
			}
		}
	};
	if (tid == 0) draw(0);
	uint32_t it = 0;
	VKFFT_SYNC();
	if constexpr (PIPE != 0) requestA(sTicket[0]); // invariant at the head of the loop: the A tile of the ticket about to be read has been requested
	for (;;) {
		VKFFT_SYNC(); // S1: ticket visible; exchange buffer free again
		const uint32_t t = sTicket[it];
		if (t >= totq) {
			// this queue is drained: help the next one, leave when every queue is (completion must not depend on where workgroups run)
			if (++tried >= Q) break;
			VKFFT_SYNC(); // every wave has read the ticket
			q = q + 1u == Q ? 0u : q + 1u;
			Cq = (p.C + Q - 1u - q) / Q;
			totq = Cq ? (Cq + p.D) * TPC : 0u;
			if (tid == 0) draw(it);
			VKFFT_SYNC();
			if constexpr (PIPE != 0) requestA(sTicket[it]);
			continue;
		}
		const uint32_t okA = sOkA[it], okB = sOkB[it];
		it ^= 1u;
		VKFFT_OPAQUE_ZERO(oz); // (the tables are the same for every tile: an opaque zero in their base keeps the loads inside the loop)
		const GBuf gtw = make_gbuf((const char*)p.tw4 + oz);
		const uint32_t s = t / TPC, r = t - s * TPC;
		const uint32_t f = r / tiles, ti = r - f * tiles;
		const bool hasA = s < Cq, hasB = s >= p.D && s - p.D < Cq;
		// ---- request the B tile: ring -> registers, TCB neighbouring k0 (stride n0 in the ring); it travels while the A tile computes
		const uint32_t sB = s - p.D, chB = q + Q * sB;
		const uint32_t bB = ((p.reverse ? p.C - 1u - chB : chB) << p.logG) + f;
		const uint32_t k00 = ti * (uint32_t)TCB;
		const bool liveB = hasB && bB < p.batch && k00 < (uint32_t)NA;
		const uint32_t laneB = (liveB && k00 + cB < (uint32_t)NA) ? cB * ES : kGbInvalid;
		if (hasB && !okB) fused_wait(p.ctr + depB(s), TPC); // rare (the flag was sampled one ticket ago: ordered before these loads by S1)
		cx<T> xb[P0B][R0B];
		auto requestB = [&]() {
			const char* const sbaseB = (const char*)p.scratch + ((uint64_t)(((q * p.NS + (hasB ? sB % p.NS : 0u)) << p.logG) + f) * nPts) * ES;
			const GBuf gsB = make_gbuf(sbaseB + (uint64_t)(liveB ? k00 : 0u) * ES);
#pragma unroll
			for (int bb = 0; bb < P0B; bb++) {
				const uint32_t t2 = tauB + (uint32_t)(bb * TPFB);
				if ((bb + 1) * TPFB <= NB0B || t2 < (uint32_t)NB0B) {
#pragma unroll
					for (int i = 0; i < R0B; i++) xb[bb][i] = gb_load_x<T, AUX_SC>(gsB, laneB + t2 * NAP * ES, (uint32_t)(i * NB0B) * NAP * ES + oz);
				}
			}
		};
		if constexpr (PIPE != 0) requestB(); else requestA(t);
		// ---- A: FFT over n0 of TCA neighbouring columns (stride n1), twiddle, per-column contiguous stores into the ring
		const uint32_t chA = q + Q * s; // chunk in processing order (counters, ring slot)
		const uint32_t bA = ((p.reverse ? p.C - 1u - chA : chA) << p.logG) + f;
		const uint32_t col0 = ti * (uint32_t)TCA;
		const bool live = hasA && bA < p.batch && col0 < (uint32_t)NBN; // (the last chunk may be partial, a phase may have fewer tiles than the ticket count)
		// the A tile is in registers (counted wait: the B loads stay in flight)
#pragma unroll
		for (int bb = 0; bb < P0A; bb++) gb_landed<T, R0A>(xa[bb]);
		if (preBlue) {
			if constexpr (BLUE != 0) { // (the chirp leaves the registers here, ahead of the stages)
				const bool sw = o.bsSwapIn != 0;
#pragma unroll
				for (int bb = 0; bb < P0A; bb++) {
#pragma unroll
					for (int i = 0; i < R0A; i++) xa[bb][i] = cmulc(sw ? cswap(xa[bb][i]) : xa[bb][i], ca[bb][i]);
				}
			}
		}
		if (hasA && !okA) fused_wait(p.ctr + depA(s), TPC);
		if (live) {
			if (p.swapIn) {
#pragma unroll
				for (int bb = 0; bb < P0A; bb++) {
#pragma unroll
					for (int i = 0; i < R0A; i++) xa[bb][i] = cswap(xa[bb][i]);
				}
			}
			auto outA = [&](uint32_t t2, uint32_t c2, cx<T> v) { colA[t2 + c2] = v; };
			if constexpr (TWL) mf_stage<T, SA, 0, TPFA, true>(colA, MfTwLds<T>{twA}, tauA, McRegs<T, R0A>{xa}, outA);
			else mf_stage<T, SA, 0, TPFA, true>(colA, MfTwGlobal<T>{make_gbuf((const char*)p.lutA + oz)}, tauA, McRegs<T, R0A>{xa}, outA);
			VKFFT_SYNC();
			// the tile leaves as ONE contiguous run of the ring (its columns are neighbours there: Y^T[col0 + c][k0]); 16 bytes per lane, the twiddle on the way
			const char* const sbase = (const char*)p.scratch + ((uint64_t)(((q * p.NS + s % p.NS) << p.logG) + f) * nPts) * ES;
			const uint32_t ncols = (uint32_t)NBN - col0 < (uint32_t)TCA ? (uint32_t)NBN - col0 : (uint32_t)TCA;
			const GBuf gs = make_gbuf(sbase + (uint64_t)col0 * NAP * ES);
			const uint32_t loMask = (1u << p.fsLoBits) - 1u, hiBase = (loMask + 1u) * ES;
			auto elem = [&](uint32_t c2, uint32_t k) -> cx<T> {
				const uint32_t x = k * (col0 + c2);
				cx<T> w;
				if constexpr (FSL) w = cmul(fsT[x & loMask], fsT[(loMask + 1u) + (x >> p.fsLoBits)]);
				else w = cmul(gb_load<T>(gtw, (x & loMask) * ES, 0), gb_load<T>(gtw, (x >> p.fsLoBits) * ES, hiBase));
				return cmul(lds[c2 * PITA + k], w);
			};
			// (two iterations in flight: unrolled in full — up to thirteen — the compiler requested every table entry first and the instance took 250 registers)
			if constexpr (sizeof(T) == 4) {
				constexpr uint32_t NAH = ((uint32_t)NA + 1u) / 2u; // 16-byte units per column (the last one half filled when n0 is odd)
				const uint32_t lim = ncols * NAH;
#pragma unroll 2
				for (uint32_t e0 = tid; e0 < lim; e0 += (uint32_t)NT) {
					const uint32_t c2 = e0 / NAH, k = 2u * (e0 - c2 * NAH);
					if (k + 1u < (uint32_t)NA) gb_store2_x<T, AUX_ST>(gs, (c2 * NAP + k) * ES, elem(c2, k), elem(c2, k + 1u));
					else gb_store_x<T, AUX_ST>(gs, (c2 * NAP + k) * ES, 0, elem(c2, k));
				}
			} else {
				const uint32_t lim = ncols * (uint32_t)NA;
#pragma unroll 2
				for (uint32_t e0 = tid; e0 < lim; e0 += (uint32_t)NT) { const uint32_t c2 = e0 / (uint32_t)NA, k = e0 - c2 * (uint32_t)NA; gb_store_x<T, AUX_ST>(gs, (c2 * NAP + k) * ES, 0, elem(c2, k)); }
			}
		}
		if constexpr (PIPE == 0) requestB();
		VKFFT_VMEM_DRAIN();     // this wave: B tile in registers, ring stores acknowledged by the memory side
		if (tid == 0) draw(it); // next ticket + the state of its dependencies (read after S1 of the next iteration)
		VKFFT_SYNC();           // S3: ... in every wave; the exchange buffer is free
		if (tid == 0) {
			if (hasA) (void)VKFFT_ATOMIC_ADD_U32(p.ctr + doneA + chA, 1u); // the chunk's tile is in the ring
			if (hasB) (void)VKFFT_ATOMIC_ADD_U32(p.ctr + doneB + chB, 1u); // the ring slot's tile has been read
		}
		if constexpr (PIPE != 0) requestA(sTicket[it]); // the A tile of the next ticket travels while the B tile computes
		if (liveB) {
			// ---- B: FFT over n1 of TCB neighbouring k0, natural-order store X[k0 + n0 * k1] straight from the last stage
			const GBuf gout = make_gbuf((cx<T>*)p.out + ((int64_t)bB * p.outBatchStride + k00));
			const T sc = (T)p.scale;
			const bool swO = p.swapOut != 0;
			auto stagesB = [&](const auto& outB) {
				if constexpr (TWL) mf_stage<T, SB, 0, TPFB, false>(colB, MfTwLds<T>{twB}, tauB, McRegs<T, R0B>{xb}, outB);
				else mf_stage<T, SB, 0, TPFB, false>(colB, MfTwGlobal<T>{make_gbuf((const char*)p.lutB + oz)}, tauB, McRegs<T, R0B>{xb}, outB);
			};
			if constexpr (BLUE != 0) {
				const GBuf gbh = make_gbuf((const char*)o.bhat + oz), gch = make_gbuf((const char*)o.chirp + oz);
				const bool postMul = o.postMul != 0, postBlue = o.postBlue != 0, swB = o.bsSwapOut != 0;
				stagesB([&](uint32_t t2, uint32_t c2, cx<T> v) {
					const uint32_t kr = k00 + cB + t2 * (uint32_t)NA; // natural position of this output: kr + c2 * n0 (c2 a compile-time multiple: scalar offset)
					const bool lane = laneB != kGbInvalid;
					if (swO) v = cswap(v);
					if (postMul) v = cmul(v, gb_load<T>(gbh, lane ? kr * ES : kGbInvalid, c2 * (uint32_t)NA * ES));
					bool st = lane;
					if (postBlue) {
						st = lane && kr + c2 * (uint32_t)NA < o.blueN;
						v = cmulc(v, gb_load<T>(gch, st ? kr * ES : kGbInvalid, c2 * (uint32_t)NA * ES));
						if (swB) v = cswap(v);
					}
					if (sc != (T)1) v = cscale(v, sc);
					gb_store_x<T, AUX_HBM_ST>(gout, st ? laneB + t2 * (uint32_t)NA * ES : kGbInvalid, c2 * (uint32_t)NA * ES, v);
				});
			} else
			stagesB([&](uint32_t t2, uint32_t c2, cx<T> v) {
				if (swO) v = cswap(v);
				if (sc != (T)1) v = cscale(v, sc);
				gb_store_x<T, AUX_HBM_ST>(gout, laneB + t2 * (uint32_t)NA * ES, c2 * (uint32_t)NA * ES, v);
			});
		}
	}
	// ---- exit: the last workgroup out resets the counters for the next launch (every completion was published inside the loop)
	VKFFT_VMEM_DRAIN();
	VKFFT_SYNC();
	if (tid == 0) {
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
template <typename T, typename SA, int TPFA, int TCA, typename SB, int TPFB, int TCB, int MODE, int BLUE, int WGC, int PIPE = 1> void mix_fused_launch(const FusedParams& prm, const MixFusedOps& ops, dim3 grid, hipStream_t s) {
	hipLaunchKernelGGL((mix_fused_kernel<T, SA, TPFA, TCA, SB, TPFB, TCB, MODE, BLUE, WGC, PIPE>), grid, dim3(TPFA * TCA), 0, s, prm, ops);
}

} // namespace vkfft_mi355x
