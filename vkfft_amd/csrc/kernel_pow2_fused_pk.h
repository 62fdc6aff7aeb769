// Fused Four-Step, software-pipelined form on PACKED PAIRS (round 5): the choreography of kernel_pow2_fused_pipe.h (ticket queues, ring, coherence rules,
// one tile's memory traffic in flight while the other tile computes) with the stages of kernel_pow2_pk.h — a thread's two adjacent columns live in
// structure-of-arrays register pairs and all butterfly arithmetic issues on the packed fp32 pipe (v_pk_add / v_pk_mul / v_pk_fma with op_sel broadcasts).
// Per thread and ticket at 2^20: 2 800 vector instructions (765 of them moves) in the round-4 form, about 1 700 here; 150 instead of 250 registers.
// The ring holds 16-byte units (Re p0, Re p1, Im p0, Im p1) of two consecutive points of a column: private to this kernel, written and read as register pairs.
#pragma once
#include "kernel_pow2_fused.h"
#include "kernel_pow2_pk.h"

namespace vkfft_mi355x {

// workgroups per CU: what the LDS holds, at most WGC, and never more than two waves per SIMD (256 registers per thread)
template <typename T, typename SA, int TCA, typename SB, int TCB, int TWL, int WGC> constexpr int pow2_fused_pk_wg_per_cu() {
	constexpr int w0 = pow2_fused_wg_per_cu<T, SA, TCA, SB, TCB, TWL, 2, 1>();
	constexpr int nt = ((1 << SA::LOGN) >> SA::LOGE) * TCA / 2;
	constexpr int pa = (int)pow2_lean_plane_elems<SA, TCA>(), pb = (int)pow2_lean_plane_elems<SB, TCB>();
	constexpr int ldsBytes = (pa > pb ? pa : pb) * (int)sizeof(T) + (TWL ? SA::lutTotal() + SB::lutTotal() + (2 << SA::LOGN) : 0) * (int)sizeof(cx<T>) + 128; // (+ the row table)
	constexpr int w = w0 < 163840 / ldsBytes ? w0 : 163840 / ldsBytes;
	constexpr int r = 512 / nt > 0 ? 512 / nt : 1;
	return (w < WGC ? w : WGC) < r ? (w < WGC ? w : WGC) : r;
}

template <typename T, typename SA, int TCA, typename SB, int TCB, int MODE, int TWL, int WGC>
__global__ void __launch_bounds__(((1 << SA::LOGN) >> SA::LOGE) * TCA / 2, (pow2_fused_pk_wg_per_cu<T, SA, TCA, SB, TCB, TWL, WGC>() * (((1 << SA::LOGN) >> SA::LOGE) * TCA / 2) + 255) / 256)
pow2_fused_pk_kernel(const FusedParams p) {
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
	constexpr int PLA = (int)pow2_lean_plane_elems<SA, TCA>(), PLB = (int)pow2_lean_plane_elems<SB, TCB>(), PLN = ((PLA > PLB ? PLA : PLB) + 3) & ~3; // (a multiple of 4 reals: what follows the plane is 16-byte aligned)
	constexpr int LUTA = TWL ? SA::lutTotal() : 0, LUTB = TWL ? SB::lutTotal() : 0;
	constexpr int ROWL = TWL ? 2 * LA : 0; // the row table of the Four-Step twiddle (pk_fs_apply) beside the stage twiddles: 16 bytes per point of the first factor
	constexpr int TWG = 8;
	__shared__ __attribute__((aligned(16))) cx<T> lds[PLN / 2 + ROWL + LUTA + LUTB];
	T* const plane = (T*)lds;
	__shared__ uint32_t sTicket[2], sOkA[2], sOkB[2];
	const uint32_t tid = threadIdx.x;
	cx<T>* const rowL = lds + PLN / 2;
	cx<T>* const twA = rowL + ROWL;
	cx<T>* const twB = twA + LUTA;
	for (uint32_t i = tid; i < (uint32_t)ROWL; i += NT) rowL[i] = ((const cx<T>*)p.rowTab)[i];
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
	pk4<T> vraw[EA]; // the A tile while it travels: (x, y) of the thread's two adjacent columns
	cxp<T> vB[EB];
	// the A tile of ticket tt (of the CURRENT queue): TCA neighbouring columns (stride n1) of transform b, requested from HBM
	auto requestA = [&](uint32_t tt) {
		const uint32_t s = tt >> logTPC, r = tt & (TPC - 1u);
		const uint32_t f = r >> p.logTiles, ti = r & ((1u << p.logTiles) - 1u);
		const uint32_t cA = q + Q * s;
		const uint32_t b = ((p.reverse ? p.C - 1u - cA : cA) << p.logG) + f;
		const bool live = s < Cq && b < p.batch;
		const GBuf gin = make_gbuf((const cx<T>*)p.in + ((int64_t)(live ? b : 0u) * p.inBatchStride + ti * TCA));
		VKFFT_OPAQUE_ZERO(oq); // (the multiples of the step are recomputed at every request: hoisted out of the persistent loop they are a scalar register each)
		const uint32_t laneIn = live ? (tauA * p.n1 + cAl) * ES : kGbInvalid, stepIn = (uint32_t)TPFA * p.n1 * ES + oq;
#pragma unroll
		for (int m = 0; m < EA; m++) vraw[m] = gb_load_aos2<T, AUX_HBM>(gin, laneIn, m * stepIn);
	};
	VKFFT_PKPROF_DECL;
	if (tid == 0) draw(0);
	uint32_t it = 0;
	VKFFT_SYNC();
	requestA(sTicket[0]); // invariant at the head of the loop: the A tile of the ticket about to be read has been requested (no branch around the
	                      // requests anywhere: lanes of a ticket without an A or B part get out-of-range offsets, which cost no traffic)
	for (;;) {
		VKFFT_SYNC(); // S1: ticket visible; the plane is free
		VKFFT_PKPROF(0);
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
		const uint32_t laneB = liveB ? (tauB * p.n0 + cBl) * ES : kGbInvalid, stepB = (uint32_t)TPFB * p.n0 * ES + oz;
		if (hasB && !okB) fused_wait(p.ctr + depB(s), TPC); // rare (the flag was sampled one ticket ago: ordered before these loads by S1)
		{
			const char* const sbaseB = (const char*)p.scratch + ((uint64_t)(((q * p.NS + (hasB ? sB % p.NS : 0u)) << p.logG) + f) * nPts) * ES;
			const GBuf gsB = make_gbuf(sbaseB + (uint64_t)k00 * ES);
#pragma unroll
			for (int m = 0; m < EB; m++) vB[m] = gb_load_soa2<T, AUX_SC>(gsB, laneB, m * stepB);
		}
		// ---- A: FFT over n0 of TCA neighbouring columns, twiddle, per-column contiguous write-through stores into the ring
		const uint32_t cA = q + Q * s;
		const uint32_t bA = ((p.reverse ? p.C - 1u - cA : cA) << p.logG) + f;
		const bool live = hasA && bA < p.batch; // the last chunk may be partial: its empty tiles only keep the counters uniform
		VKFFT_PKPROF(3); // (ticket decode + B request)
		gb_landed_raw<T, EA>(vraw); // the A tile is in registers (counted wait: the B loads stay in flight)
		VKFFT_PKPROF(1);
		if (hasA && !okA) { fused_wait(p.ctr + depA(s), TPC); VKFFT_PKPROF(5); }
		if (live) {
			const uint32_t col0 = ti * TCA;
			PkFsTw<T, SA::LOGE> fsq; // the Four-Step twiddle's table look-ups travel during the stages
			pk_fs_request<T, SA::LOGE, TPFA>(fsq, gtw, p.fsLoBits, tauA, col0 + cAl);
			cxp<T> v[EA];
#pragma unroll
			for (int m = 0; m < EA; m++) v[m] = pk_from_aos<T>(vraw[m]);
			if (p.swapIn) { // inverse = conj . forward . conj (the packed form of the swap identity of the other kernels: in place, no renaming)
#pragma unroll
				for (int m = 0; m < EA; m++) v[m].im = -v[m].im;
			}
			if constexpr (TWL) pk_lean_stages<T, SA, 0, TPFA, TCA, TwLds<T>, TWG>(v, plane + cAl, TwLds<T>{twA}, tauA);
			else pk_lean_stages<T, SA, 0, TPFA, TCA, TwGlobal<T>, TWG>(v, plane + cAl, TwGlobal<T>{make_gbuf((const char*)p.lutA + oz)}, tauA);
			VKFFT_PKPROF(8);
			if constexpr (TWL) pk_fs_apply<T, SA::LOGE, TPFA>(v, fsq, RowLds<T>{rowL}, tauA);
			else pk_fs_apply<T, SA::LOGE, TPFA>(v, fsq, RowGlobal<T>{make_gbuf((const char*)p.rowTab + oz)}, tauA);
			VKFFT_PKPROF(9);
			if constexpr (SA::NS > 1) VKFFT_SYNC(); // the last exchange's reads are complete
			cxp<T> rr[EA];
			pk_lean_transpose<T, LA, EA, TPFA, TCA, NT>(v, rr, plane, tid, cAl, tauA);
			VKFFT_PKPROF(10);
			const char* const sbase = (const char*)p.scratch + ((uint64_t)(((q * p.NS + s % p.NS) << p.logG) + f) * nPts) * ES;
			const GBuf gs = make_gbuf(sbase + (uint64_t)col0 * LA * ES);
#pragma unroll
			for (int i = 0; i < EA; i++) {
				const uint32_t idx = tid + i * NT;
				const uint32_t kp = idx % (LA / 2), cc = idx / (LA / 2);
				gb_store_soa2<T, AUX_ST>(gs, (cc * LA + 2u * kp) * ES, rr[i]);
			}
		}
		VKFFT_PKPROF(2); // (ring stores issued)
		VKFFT_VMEM_DRAIN(); // this wave: B tile in registers, ring stores acknowledged by the memory side
		VKFFT_PKPROF(6); // (drain: B tile landed, stores acknowledged)
		if (tid == 0) draw(it); // next ticket + the state of its dependencies (read after S1 of the next iteration)
		VKFFT_SYNC();           // S3: ... in every wave; the plane is free
		VKFFT_PKPROF(11);
		if (tid == 0) {
			if (hasA) (void)VKFFT_ATOMIC_ADD_U32(p.ctr + doneA + cA, 1u); // the chunk's tile is in the ring
			if (hasB) (void)VKFFT_ATOMIC_ADD_U32(p.ctr + doneB + cB, 1u); // the ring slot's tile has been read
		}
		requestA(sTicket[it]); // the A tile of the next ticket travels while the B tile computes
		if (liveB) {
			// ---- B: FFT over n1 of TCB neighbouring k0, natural-order store X[k0 + n0*k1]
			if constexpr (TWL) pk_lean_stages<T, SB, 0, TPFB, TCB, TwLds<T>, TWG>(vB, plane + cBl, TwLds<T>{twB}, tauB);
			else pk_lean_stages<T, SB, 0, TPFB, TCB, TwGlobal<T>, TWG>(vB, plane + cBl, TwGlobal<T>{make_gbuf((const char*)p.lutB + oz)}, tauB);
			const T sc = (T)p.scale, sci = p.swapOut ? -sc : sc;
			if (sc != (T)1 || p.swapOut) {
#pragma unroll
				for (int m = 0; m < EB; m++) { vB[m].re = vB[m].re * pk_splat<T>(sc); vB[m].im = vB[m].im * pk_splat<T>(sci); }
			}
			const GBuf gout = make_gbuf((cx<T>*)p.out + ((int64_t)bB * p.outBatchStride + k00));
#pragma unroll
			for (int m = 0; m < EB; m++) gb_store_aos2<T, AUX_HBM>(gout, laneB + m * stepB, vB[m]);
		}
		VKFFT_PKPROF(4);
#if !defined(VKFFT_HOSTEMU)
		if constexpr ((MODE & 4) != 0) { if (tid == 0) spc[7]++; }
#endif
	}
	VKFFT_PKPROF_FLUSH();
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

template <typename T, typename SA, int TCA, typename SB, int TCB, int MODE, int TWL, int WGC> void pow2_fused_pk_launch(const FusedParams& prm, dim3 grid, hipStream_t s) {
	constexpr int threads = ((1 << SA::LOGN) >> SA::LOGE) * TCA / 2;
	hipLaunchKernelGGL((pow2_fused_pk_kernel<T, SA, TCA, SB, TCB, MODE, TWL, WGC>), grid, dim3(threads), 0, s, prm);
}

} // namespace vkfft_mi355x
