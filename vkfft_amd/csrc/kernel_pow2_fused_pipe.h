// Fused Four-Step, software-pipelined form (round 4): the ticket queue, ring and coherence rules of kernel_pow2_fused.h, with the tile phases
// reordered so that a workgroup always has one tile's memory traffic in flight while the other tile computes.
//
// The per-phase cycle profile of the first form (profiles/r04_fused_phase_profile_*.txt; 2^20: 48 k cycles per ticket) shows a workgroup
// waiting 8.7 k cycles for its A tile from HBM and 4.6 k for its B tile from the ring with nothing else to do: the CU's registers were full with ONE
// tile (the compiler spent the other 190 registers of a 256-register thread on twiddles in flight and butterfly temporaries), and two workgroups
// per CU do not fit the LDS (or, in the register-lean plane-split form, pay for it in instructions).  With the register-lean stages of
// kernel_pow2_lean.h a thread needs its 64 data registers + about 60: at 256 registers per thread TWO tiles fit.  So, per ticket:
//
//     top:   [A tile of THIS ticket already in flight: requested during the previous ticket's B phase]
//            request the B tile of this ticket from the ring (its flag was sampled one ticket ago)
//            wait for the A tile only (counted: the B loads stay in flight)
//            A: stages, Four-Step twiddle, turn, write-through stores into the ring
//            drain (B tile landed long ago; the ring stores just issued are acknowledged), barrier -> publish doneA / doneB at once
//            request the A tile of the NEXT ticket from HBM
//            B: stages, natural-order stores to HBM                      <- the next A tile lands meanwhile
//
// Every dependency still points at a smaller ticket (deadlock-free for any grid), a ring slot is still written with write-through stores and read
// with memory-side loads only after its flag was seen, completions are published only after every wave's stores are acknowledged.  In-place
// transforms stay safe: the A tiles of a chunk are all read before its first B tile is written, and a prefetched A tile belongs to a later chunk.
#pragma once
#include "kernel_pow2_fused.h"

namespace vkfft_mi355x {

// workgroups per CU: what the LDS holds, at most WGC, and never more than two waves per SIMD (256 registers per thread)
template <typename T, typename SA, int TCA, typename SB, int TCB, int TWL, int WGC> constexpr int pow2_fused_pipe_wg_per_cu() {
	constexpr int w = pow2_fused_wg_per_cu<T, SA, TCA, SB, TCB, TWL, 2, 1>();
	constexpr int nt = ((1 << SA::LOGN) >> SA::LOGE) * TCA / 2;
	constexpr int r = 512 / nt > 0 ? 512 / nt : 1;
	return (w < WGC ? w : WGC) < r ? (w < WGC ? w : WGC) : r;
}

template <typename T, typename SA, int TCA, typename SB, int TCB, int MODE, int TWL, int WGC>
__global__ void __launch_bounds__(((1 << SA::LOGN) >> SA::LOGE) * TCA / 2, (pow2_fused_pipe_wg_per_cu<T, SA, TCA, SB, TCB, TWL, WGC>() * (((1 << SA::LOGN) >> SA::LOGE) * TCA / 2) + 255) / 256)
pow2_fused_pipe_kernel(const FusedParams p) {
	static_assert(sizeof(T) == 4, "two fp32 columns per thread");
	constexpr int CPT = 2;
	constexpr int LA = 1 << SA::LOGN, EA = 1 << SA::LOGE, TPFA = LA / EA;
	constexpr int LB = 1 << SB::LOGN, EB = 1 << SB::LOGE, TPFB = LB / EB;
	constexpr int NT = TPFA * TCA / CPT;
	static_assert(NT == TPFB * TCB / CPT, "both phases run on the same workgroup shape");
	static_assert(LA * TCA == LB * TCB, "both phases move the same number of points per tile");
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	constexpr int AUX_SC = 16, AUX_ST = 16;       // ring: memory-side loads, write-through stores
	constexpr int AUX_HBM = (MODE & 2) ? 2 : 0;   // streamed side: non-temporal hint
	constexpr int PLA = (int)pow2_lean_plane_elems<SA, TCA>(), PLB = (int)pow2_lean_plane_elems<SB, TCB>(), PLN = (PLA > PLB ? PLA : PLB) + ((PLA > PLB ? PLA : PLB) & 1);
	constexpr int LUTA = TWL ? SA::lutTotal() : 0, LUTB = TWL ? SB::lutTotal() : 0;
	constexpr int TWG = 8, PFN = TWL ? 0 : 8;
	__shared__ cx<T> lds[PLN / 2 + LUTA + LUTB];
	T* const plane = (T*)lds;
	__shared__ uint32_t sTicket[2], sOkA[2], sOkB[2];
	const uint32_t tid = threadIdx.x;
	cx<T>* const twA = lds + PLN / 2;
	cx<T>* const twB = twA + LUTA;
	for (uint32_t i = tid; i < (uint32_t)LUTA; i += NT) twA[i] = ((const cx<T>*)p.lutA)[i];
	for (uint32_t i = tid; i < (uint32_t)LUTB; i += NT) twB[i] = ((const cx<T>*)p.lutB)[i];
	const uint32_t logTPC = p.logG + p.logTiles, TPC = 1u << logTPC;
	const uint32_t doneA = kFusedCtrDone, doneB = kFusedCtrDone + p.C;
	const uint64_t nPts = (uint64_t)p.n0 * p.n1;
	constexpr uint32_t kNone = 0xffffffffu;
	const uint32_t Q = p.Q;
	uint32_t q = Q > 1 ? fused_xcc_id() % Q : 0u, tried = 0;
	uint32_t Cq = (p.C + Q - 1u - q) / Q;
	uint32_t totq = Cq ? (Cq + p.D) << logTPC : 0u;
	auto depA = [&](uint32_t s) -> uint32_t { return (s < Cq && s >= p.NS) ? doneB + q + Q * (s - p.NS) : kNone; };
	auto depB = [&](uint32_t s) -> uint32_t { return (s >= p.D && s - p.D < Cq) ? doneA + q + Q * (s - p.D) : kNone; };
	auto draw = [&](uint32_t slot) { // thread 0: next ticket of queue q and the state of ITS dependencies
		const uint32_t t0 = VKFFT_ATOMIC_ADD_U32(p.ctr + kFusedCtrTicket + 32u * q, 1u), s0 = t0 >> logTPC;
		const uint32_t dA = t0 < totq ? depA(s0) : kNone, dB = t0 < totq ? depB(s0) : kNone;
		sTicket[slot] = t0;
		sOkA[slot] = dA == kNone ? 1u : (uint32_t)(VKFFT_ATOMIC_LOAD_U32(p.ctr + dA) >= TPC);
		sOkB[slot] = dB == kNone ? 1u : (uint32_t)(VKFFT_ATOMIC_LOAD_U32(p.ctr + dB) >= TPC);
	};
	const uint32_t cAl = (tid % (TCA / CPT)) * CPT, tauA = tid / (TCA / CPT); // first of this thread's two adjacent columns, A tiles
	const uint32_t cBl = (tid % (TCB / CPT)) * CPT, tauB = tid / (TCB / CPT); // ... B tiles
	cx<T> v[CPT * EA], vB[CPT * EB];
	// the A tile of ticket tt (of the CURRENT queue): TCA neighbouring columns (stride n1) of transform b, requested from HBM
	auto requestA = [&](uint32_t tt) {
		const uint32_t s = tt >> logTPC, r = tt & (TPC - 1u);
		const uint32_t f = r >> p.logTiles, ti = r & ((1u << p.logTiles) - 1u);
		const uint32_t cA = q + Q * s;
		const uint32_t b = ((p.reverse ? p.C - 1u - cA : cA) << p.logG) + f;
		const bool live = s < Cq && b < p.batch;
		const GBuf gin = make_gbuf((const cx<T>*)p.in + ((int64_t)(live ? b : 0u) * p.inBatchStride + ti * TCA));
		const uint32_t laneIn = live ? (tauA * p.n1 + cAl) * ES : kGbInvalid, stepIn = (uint32_t)TPFA * p.n1 * ES;
#pragma unroll
		for (int m = 0; m < EA; m++) gb_load2_x<T, AUX_HBM>(gin, laneIn, m * stepIn, v[m], v[EA + m]);
	};
	if (tid == 0) draw(0);
	uint32_t it = 0;
	VKFFT_SYNC();
	requestA(sTicket[0]); // invariant at the head of the loop: the A tile of the ticket about to be read has been requested (no branch around the
	                      // requests anywhere: lanes of a ticket without an A or B part get out-of-range offsets, which cost no traffic)
	for (;;) {
		VKFFT_SYNC(); // S1: ticket visible; the plane is free
		const uint32_t t = sTicket[it];
		if (t >= totq) {
			// this queue is drained: help the next one, leave when every queue is (completion must not depend on where workgroups run)
			if (++tried >= Q) break;
			VKFFT_SYNC(); // every wave has read the ticket
			q = q + 1u == Q ? 0u : q + 1u;
			Cq = (p.C + Q - 1u - q) / Q;
			totq = Cq ? (Cq + p.D) << logTPC : 0u;
			if (tid == 0) draw(it);
			VKFFT_SYNC();
			requestA(sTicket[it]);
			continue;
		}
		const uint32_t okA = sOkA[it], okB = sOkB[it];
		it ^= 1u;
		VKFFT_OPAQUE_ZERO(oz);
		const GBuf gtw = make_gbuf((const char*)p.tw4 + oz);
		const uint32_t s = t >> logTPC, r = t & (TPC - 1u);
		const uint32_t f = r >> p.logTiles, ti = r & ((1u << p.logTiles) - 1u);
		const bool hasA = s < Cq, hasB = s >= p.D && s - p.D < Cq;
		// ---- request the B tile: ring -> registers, TCB neighbouring k0 (stride n0 in the ring)
		const uint32_t sB = s - p.D, cB = q + Q * sB;
		const uint32_t bB = ((p.reverse ? p.C - 1u - cB : cB) << p.logG) + f;
		const bool liveB = hasB && bB < p.batch;
		const uint32_t k00 = ti * TCB;
		const uint32_t laneB = liveB ? (tauB * p.n0 + cBl) * ES : kGbInvalid, stepB = (uint32_t)TPFB * p.n0 * ES;
		if (hasB && !okB) fused_wait(p.ctr + depB(s), TPC); // rare (the flag was sampled one ticket ago: ordered before these loads by S1)
		{
			const char* const sbaseB = (const char*)p.scratch + ((uint64_t)(((q * p.NS + (hasB ? sB % p.NS : 0u)) << p.logG) + f) * nPts) * ES;
			const GBuf gsB = make_gbuf(sbaseB + (uint64_t)k00 * ES);
#pragma unroll
			for (int m = 0; m < EB; m++) gb_load2_x<T, AUX_SC>(gsB, laneB, m * stepB, vB[m], vB[EB + m]);
		}
		// ---- A: FFT over n0 of TCA neighbouring columns, twiddle, per-column contiguous write-through stores into the ring
		const uint32_t cA = q + Q * s;
		const uint32_t bA = ((p.reverse ? p.C - 1u - cA : cA) << p.logG) + f;
		const bool live = hasA && bA < p.batch; // the last chunk may be partial: its empty tiles only keep the counters uniform
		gb_landed<T, CPT * EA>(v); // the A tile is in registers (counted wait: the B loads stay in flight)
		if (hasA && !okA) fused_wait(p.ctr + depA(s), TPC);
		if (live) {
			const uint32_t col0 = ti * TCA;
			if (p.swapIn) {
#pragma unroll
				for (int m = 0; m < CPT * EA; m++) v[m] = cswap(v[m]);
			}
			if constexpr (TWL) pow2_lean_stages<T, SA, 0, TPFA, TCA, TwLds<T>, TWG, CPT>(v, plane + cAl, TwLds<T>{twA}, tauA);
			else pow2_lean_stages<T, SA, 0, TPFA, TCA, TwGlobal<T>, TWG, CPT, PFN>(v, plane + cAl, TwGlobal<T>{make_gbuf((const char*)p.lutA + oz)}, tauA);
#pragma unroll
			for (int cc = 0; cc < CPT; cc++) pow2_fs_twiddle<T, SA::LOGE, TPFA>(v + cc * EA, gtw, p.fsLoBits, tauA, col0 + cAl + cc);
			if constexpr (SA::NS > 1) VKFFT_SYNC(); // the last exchange's reads are complete
			cx<T> rr[CPT * EA];
			pow2_lean_transpose<T, LA, EA, TPFA, TCA, NT>(v, rr, plane, tid, cAl, tauA);
			const char* const sbase = (const char*)p.scratch + ((uint64_t)(((q * p.NS + s % p.NS) << p.logG) + f) * nPts) * ES;
			const GBuf gs = make_gbuf(sbase + (uint64_t)col0 * LA * ES);
#pragma unroll
			for (int i = 0; i < CPT * EA / 2; i++) {
				const uint32_t idx = tid + i * NT;
				const uint32_t kp = idx % (LA / 2), cc = idx / (LA / 2);
				gb_store2_x<T, AUX_ST>(gs, (cc * LA + 2u * kp) * ES, rr[2 * i], rr[2 * i + 1]);
			}
		}
		VKFFT_VMEM_DRAIN(); // this wave: B tile in registers, ring stores acknowledged by the memory side
		if (tid == 0) draw(it); // next ticket + the state of its dependencies (read after S1 of the next iteration)
		VKFFT_SYNC();           // S3: ... in every wave; the plane is free
		if (tid == 0) {
			if (hasA) (void)VKFFT_ATOMIC_ADD_U32(p.ctr + doneA + cA, 1u); // the chunk's tile is in the ring
			if (hasB) (void)VKFFT_ATOMIC_ADD_U32(p.ctr + doneB + cB, 1u); // the ring slot's tile has been read
		}
		requestA(sTicket[it]); // the A tile of the next ticket travels while the B tile computes
		if (liveB) {
			// ---- B: FFT over n1 of TCB neighbouring k0, natural-order store X[k0 + n0*k1]
			if constexpr (TWL) pow2_lean_stages<T, SB, 0, TPFB, TCB, TwLds<T>, TWG, CPT>(vB, plane + cBl, TwLds<T>{twB}, tauB);
			else pow2_lean_stages<T, SB, 0, TPFB, TCB, TwGlobal<T>, TWG, CPT, PFN>(vB, plane + cBl, TwGlobal<T>{make_gbuf((const char*)p.lutB + oz)}, tauB);
			if (p.swapOut) {
#pragma unroll
				for (int m = 0; m < CPT * EB; m++) vB[m] = cswap(vB[m]);
			}
			const T sc = (T)p.scale;
			if (sc != (T)1) {
#pragma unroll
				for (int m = 0; m < CPT * EB; m++) vB[m] = cscale(vB[m], sc);
			}
			const GBuf gout = make_gbuf((cx<T>*)p.out + ((int64_t)bB * p.outBatchStride + k00));
#pragma unroll
			for (int m = 0; m < EB; m++) gb_store2_x<T, AUX_HBM>(gout, laneB + m * stepB, vB[m], vB[EB + m]);
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

template <typename T, typename SA, int TCA, typename SB, int TCB, int MODE, int TWL, int WGC> void pow2_fused_pipe_launch(const FusedParams& prm, dim3 grid, hipStream_t s) {
	constexpr int threads = ((1 << SA::LOGN) >> SA::LOGE) * TCA / 2;
	hipLaunchKernelGGL((pow2_fused_pipe_kernel<T, SA, TCA, SB, TCB, MODE, TWL, WGC>), grid, dim3(threads), 0, s, prm);
}

} // namespace vkfft_mi355x
