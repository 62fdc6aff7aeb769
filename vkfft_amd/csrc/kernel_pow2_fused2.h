// Fused Four-Step, second generation: the ticket queue, ring and completion counters of kernel_pow2_fused.h with a different data path.
//
// What limited the first generation (profiles/r02_*): a workgroup serialised  load tile -> wait -> compute -> store  and a CU's
// registers were full with two such tiles, so HBM latency sat in the open twice per ticket.  Here the NEXT tile is on its way while the current one
// computes, at no register cost: tiles arrive by LDS-DMA (memops.h gb_dma16: buffer_load_dwordx4 ... lds, 64 x 16 bytes per wave-instruction
// straight into LDS).  tools/probe_dma.hip: column tiles streamed this way through two LDS buffers reach 5.8 TB/s (128-byte row segments) to
// 6.0 TB/s (256-byte) read + write on MI355X, where the same tiles through registers reach 5.2 / 5.9 and 64-byte segments 2.6.
//
// A workgroup owns two LDS buffers.  P receives the A tile (LA rows of TCA columns, row pitch n1: columns of the n0 x n1 view in HBM), Q the B
// tile (LB rows of TCB columns, row pitch n0: the chunk's intermediate in the ring).  A landed tile is read into registers once (lanes along the
// columns, conflict-free) and its buffer then serves as the exchange buffer of that tile's Stockham stages and, for A, of the transposition ahead
// of the ring stores.  Per ticket:
//
//     top:   wait DMA_A(cur), barrier  | thread 0: publish the previous A tile (its ring stores are acknowledged by now), request the next ticket
//            issue DMA_B(cur) -> Q      (ring -> LDS, in flight during the whole A phase)
//            A phase: P -> registers, FFT over n0, Four-Step twiddle, transpose through P | thread 0: take the next ticket, request ITS flags
//                     write-through ring stores
//     mid:   wait DMA_B(cur), barrier   | thread 0: release the ring slot (doneB)
//            issue DMA_A(next) -> P     (HBM -> LDS, in flight during the whole B phase)
//            B phase: Q -> registers, FFT over n1 | thread 0: take the flags of the next ticket
//                     natural-order stores to HBM
//
// Every wait is a counted s_waitcnt: the vector-memory queue of a wave is in order, so "all but the N youngest" retires exactly the DMA in question
// and leaves the stores behind it in flight.  All tables (stage twiddles of both factors, the two-level Four-Step table) are read from LDS.  What the
// compiler counts for itself — thread 0's ticket atomic and flag loads — it waits for with vmcnt(0) (the DMA is invisible to it), so wave 0 issues no
// DMA (the other waves share the tile) and thread 0 takes those results just BEFORE its wave's store batches (next ticket: ahead of the ring stores;
// flags: ahead of the HBM stores), when the wave's queue holds nothing younger than a phase.  Barriers are VKFFT_SYNC_RAW (no fence: a fence
// would drain the queue).
#pragma once
#include "kernel_pow2_fused.h"

namespace vkfft_mi355x {

// Four-Step twiddle w_N^(k * col) of a thread's E points from the two-level table in LDS (same factorisation as pow2_fs_twiddle)
template <typename T, int LOGE, int TPF, int LOBITS>
__device__ inline void pow2_fs_twiddle_lds(cx<T>* v, const cx<T>* tab, const uint32_t tau, const uint32_t colIdx) {
	constexpr int E = 1 << LOGE;
	constexpr uint32_t loMask = (1u << LOBITS) - 1u;
	auto tw = [&](uint32_t e) { return cmul(tab[e & loMask], tab[(loMask + 1u) + (e >> LOBITS)]); };
	constexpr int HIB = (LOGE + 1) / 2, LOB = LOGE - HIB;
	cx<T> A[1 << HIB], B[1 << LOB];
#pragma unroll
	for (int j = 0; j < (1 << HIB); j++) A[j] = tw((tau + (uint32_t)((j << LOB) * TPF)) * colIdx);
	B[0] = cx<T>{(T)1, (T)0};
#pragma unroll
	for (int i = 1; i < (1 << LOB); i++) B[i] = tw((uint32_t)(i * TPF) * colIdx);
#pragma unroll
	for (int m = 0; m < E; m++) v[m] = cmul(v[m], (m & ((1 << LOB) - 1)) ? cmul(A[m >> LOB], B[m & ((1 << LOB) - 1)]) : A[m >> LOB]);
}

template <typename T, typename SA, int TCA, typename SB, int TCB, int CPT> struct Fused2Shape {
	static constexpr int ES = (int)sizeof(cx<T>);
	static constexpr int LA = 1 << SA::LOGN, EA = 1 << SA::LOGE, TPFA = LA / EA, TCPA = TCA + (CPT == 2 ? 2 : 1);
	static constexpr int LB = 1 << SB::LOGN, EB = 1 << SB::LOGE, TPFB = LB / EB, TCPB = TCB + (CPT == 2 ? 2 : 1);
	static constexpr int NT = TPFA * TCA / CPT, NW = NT / 64;
	static constexpr int BUFN = LA * TCPA > LB * TCPB ? LA * TCPA : LB * TCPB; // complex elements per buffer (landing image L x TC, exchange image L x TCP)
	static constexpr int LUTA = SA::lutTotal(), LUTB = SB::lutTotal();
	static constexpr int LOGN = SA::LOGN + SB::LOGN, FSLO = (LOGN + 1) / 2, FSN = (1 << FSLO) + (1 << (LOGN - FSLO));
	// LDS-DMA: one wave-instruction = 1 KiB = RPI rows of the tile, LPR lanes per row
	// (wave 0 issues none — it holds the ticket thread, see below — the other NW - 1 waves deal the NI instructions of a tile round-robin)
	static constexpr int RBA = TCA * ES, RPIA = 1024 / RBA, LPRA = RBA / 16, NIA = LA * RBA / 1024, IPWA = (NIA + NW - 2) / (NW - 1);
	static constexpr int RBB = TCB * ES, RPIB = 1024 / RBB, LPRB = RBB / 16, NIB = LB * RBB / 1024, IPWB = (NIB + NW - 2) / (NW - 1);
	static constexpr int NRS = ES == 8 ? CPT * EA / 2 : CPT * EA; // ring stores per thread (16 bytes each)
	static constexpr int NSB = ES == 8 && CPT == 2 ? EB : CPT * EB; // HBM stores per thread
	static constexpr int ldsBytes = (2 * BUFN + LUTA + LUTB + FSN) * ES + 64;
	static_assert(NT % 64 == 0 && NT == TPFB * TCB / CPT && LA * TCA == LB * TCB, "both phases run on the same workgroup and tile size");
	static_assert(1024 % RBA == 0 && 1024 % RBB == 0 && RBA >= 16 && RBB >= 16, "row segments of 16 ... 1024 bytes");
	static_assert(NW >= 2, "a ticket wave and at least one DMA wave");
	static_assert(NRS < 64 && NSB < 64, "the counted waits fit the 6-bit vmcnt field (a fuller queue only stalls the issue)");
};

// MODE bit 1: non-temporal hint on the HBM side; bit 2: per-phase cycle sums (development); bit 3: without the FFT arithmetic (development)
// WPC: workgroups per CU the shape is built for (register budget through __launch_bounds__; the LDS footprint must allow it)
template <typename T, typename SA, int TCA, typename SB, int TCB, int MODE, int CPT, int WPC>
__global__ void __launch_bounds__((Fused2Shape<T, SA, TCA, SB, TCB, CPT>::NT), ((WPC * Fused2Shape<T, SA, TCA, SB, TCB, CPT>::NT + 255) / 256))
pow2_fused2_kernel(const FusedParams p) {
	using SH = Fused2Shape<T, SA, TCA, SB, TCB, CPT>;
	static_assert(CPT == 1 || (CPT == 2 && sizeof(T) == 4), "two columns per thread: fp32 only");
	static_assert(SH::ldsBytes * WPC <= 163840, "LDS footprint");
	constexpr int LA = SH::LA, EA = SH::EA, TPFA = SH::TPFA, TCPA = SH::TCPA, LB = SH::LB, EB = SH::EB, TPFB = SH::TPFB, TCPB = SH::TCPB, NT = SH::NT, BUFN = SH::BUFN;
	constexpr uint32_t ES = (uint32_t)SH::ES;
	constexpr int AUX_SC = 16, AUX_ST = 16, AUX_HBM = (MODE & 2) ? 2 : 0;
	constexpr uint32_t kNone = 0xffffffffu;
	__shared__ cx<T> buf[2 * BUFN + SH::LUTA + SH::LUTB + SH::FSN];
	__shared__ uint32_t sTk[2][2], sOk[2][2];
	cx<T>* const P = buf;
	cx<T>* const Qb = buf + BUFN;
	cx<T>* const twA = buf + 2 * BUFN;
	cx<T>* const twB = twA + SH::LUTA;
	cx<T>* const fsT = twB + SH::LUTB;
	const uint32_t tid = threadIdx.x, lane = tid & 63u;
#if defined(VKFFT_HOSTEMU)
	const uint32_t w = tid >> 6;
#else
	const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
	for (uint32_t i = tid; i < (uint32_t)SH::LUTA; i += NT) twA[i] = ((const cx<T>*)p.lutA)[i];
	for (uint32_t i = tid; i < (uint32_t)SH::LUTB; i += NT) twB[i] = ((const cx<T>*)p.lutB)[i];
	for (uint32_t i = tid; i < (uint32_t)SH::FSN; i += NT) fsT[i] = ((const cx<T>*)p.tw4)[i];
	const uint32_t logTPC = p.logG + p.logTiles, TPC = 1u << logTPC;
	const uint32_t doneA = kFusedCtrDone, doneB = kFusedCtrDone + p.C;
	const uint64_t nPts = (uint64_t)p.n0 * p.n1;
	const uint32_t Q = p.Q;
	auto cqOf = [&](uint32_t q) -> uint32_t { return (p.C + Q - 1u - q) / Q; };
	auto totOf = [&](uint32_t q) -> uint32_t { const uint32_t c = cqOf(q); return c ? (c + p.D) << logTPC : 0u; };
	// counters a ticket of slot s of queue q depends on: the ring slot's previous tenant read completely (A), the chunk written completely (B)
	auto depA = [&](uint32_t q, uint32_t s) -> uint32_t { return (s < cqOf(q) && s >= p.NS) ? doneB + q + Q * (s - p.NS) : kNone; };
	auto depB = [&](uint32_t q, uint32_t s) -> uint32_t { return (s >= p.D && s - p.D < cqOf(q)) ? doneA + q + Q * (s - p.D) : kNone; };
	// per-thread tile coordinates (the same for every tile)
	const uint32_t cA_ = (tid % (TCA / CPT)) * CPT, tauA = tid / (TCA / CPT);
	const uint32_t cB_ = (tid % (TCB / CPT)) * CPT, tauB = tid / (TCB / CPT);
	const uint32_t dmaVoffA = (lane / SH::LPRA) * p.n1 * ES + (lane % SH::LPRA) * 16u, dmaStepA = (uint32_t)SH::RPIA * p.n1 * ES;
	const uint32_t dmaVoffB = (lane / SH::LPRB) * p.n0 * ES + (lane % SH::LPRB) * 16u, dmaStepB = (uint32_t)SH::RPIB * p.n0 * ES;

	struct Tile { uint32_t s, f, ti, cA, cB, bA, bB; bool hasA, hasB, liveA, liveB; const char* ringA; const char* ringB; };
	auto decode = [&](uint32_t q, uint32_t t) -> Tile {
		Tile x;
		const uint32_t r = t & (TPC - 1u), Cq = cqOf(q);
		x.s = t >> logTPC; x.f = r >> p.logTiles; x.ti = r & ((1u << p.logTiles) - 1u);
		x.hasA = x.s < Cq; x.hasB = x.s >= p.D && x.s - p.D < Cq;
		x.cA = q + Q * x.s;
		x.bA = ((p.reverse ? p.C - 1u - x.cA : x.cA) << p.logG) + x.f;
		x.liveA = x.hasA && x.bA < p.batch; // the last chunk may be partial: its empty tiles only keep the counters uniform
		const uint32_t sB = x.s - p.D;
		x.cB = q + Q * sB;
		x.bB = ((p.reverse ? p.C - 1u - x.cB : x.cB) << p.logG) + x.f;
		x.liveB = x.hasB && x.bB < p.batch;
		x.ringA = (const char*)p.scratch + ((uint64_t)(((q * p.NS + x.s % p.NS) << p.logG) + x.f) * nPts) * ES;
		x.ringB = (const char*)p.scratch + ((uint64_t)(((q * p.NS + (x.hasB ? sB % p.NS : 0u)) << p.logG) + x.f) * nPts) * ES;
		return x;
	};
	auto dma_a = [&](const Tile& x) { // HBM -> P: rows j0 = 0 .. LA-1 of columns ti*TCA .. +TCA of transform bA
		const GDma g = make_gdma((const cx<T>*)p.in + ((int64_t)x.bA * p.inBatchStride + (int64_t)(x.ti * TCA)));
		if (w == 0) return;
#pragma unroll
		for (int j = 0; j < SH::IPWA; j++) { const uint32_t ins = (w - 1u) + (uint32_t)j * (SH::NW - 1); if (ins < (uint32_t)SH::NIA) gb_dma16<AUX_HBM>((char*)P + ins * 1024u, g, dmaVoffA, ins * dmaStepA); }
	};
	auto dma_b = [&](const Tile& x) { // ring -> Q: rows j1 = 0 .. LB-1 (pitch n0) of k0 = ti*TCB .. +TCB
		const GDma g = make_gdma(x.ringB + (uint64_t)(x.ti * TCB) * ES);
		if (w == 0) return;
#pragma unroll
		for (int j = 0; j < SH::IPWB; j++) { const uint32_t ins = (w - 1u) + (uint32_t)j * (SH::NW - 1); if (ins < (uint32_t)SH::NIB) gb_dma16<AUX_SC>((char*)Qb + ins * 1024u, g, dmaVoffB, ins * dmaStepB); }
	};
#if !defined(VKFFT_HOSTEMU)
	unsigned long long pc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ptk = 0;
#define VKFFT_PROF2(i) do { if constexpr ((MODE & 4) != 0) { const unsigned long long now = __builtin_readcyclecounter(); pc[i] += now - ptk; ptk = now; } } while (0)
	if constexpr ((MODE & 4) != 0) ptk = __builtin_readcyclecounter();
#else
#define VKFFT_PROF2(i) do { } while (0)
#endif

	// ---- thread 0: ticket state.  qT = the queue it draws from (its XCD's first; the others once that one is drained: completion must not depend on placement)
	uint32_t qT = Q > 1 ? fused_xcc_id() % Q : 0u, tried = 0, pending = kNone, nextT = 0, fA = TPC, fB = TPC;
	auto fetch_sync = [&]() -> uint32_t {
		for (;;) {
			const uint32_t t = VKFFT_ATOMIC_ADD_U32(p.ctr + kFusedCtrTicket + 32u * qT, 1u);
			if (t < totOf(qT)) return t;
			if (++tried >= Q) return kNone;
			qT = qT + 1u == Q ? 0u : qT + 1u;
		}
	};
	auto load_flags = [&](uint32_t q, uint32_t t) { // (thread 0) requests the dependency counters of ticket t; consumed at the next top
		fA = TPC; fB = TPC;
		if (t == kNone) return;
		const uint32_t s = t >> logTPC, dA = depA(q, s), dB = depB(q, s);
		if (dA != kNone) fA = VKFFT_ATOMIC_LOAD_U32(p.ctr + dA);
		if (dB != kNone) fB = VKFFT_ATOMIC_LOAD_U32(p.ctr + dB);
	};
	if (tid == 0) {
		const uint32_t t0 = fetch_sync();
		sTk[0][0] = t0; sTk[0][1] = qT;
		load_flags(qT, t0);
		sOk[0][0] = fA >= TPC; sOk[0][1] = fB >= TPC;
	}
	VKFFT_SYNC_RAW(); // tables and the first ticket are in LDS
	uint32_t it = 0;
#if defined(VKFFT_HOSTEMU)
	uint32_t t = sTk[0][0], q = sTk[0][1];
#else
	uint32_t t = __builtin_amdgcn_readfirstlane(sTk[0][0]), q = __builtin_amdgcn_readfirstlane(sTk[0][1]);
#endif
	bool ldA = false;      // DMA_A of the current ticket is under way
	uint32_t newerA = 0;   // vector-memory instructions this wave has issued after it (lower bound)
	if (t != kNone) {
		const Tile x = decode(q, t);
		if (x.liveA) { dma_a(x); ldA = true; }
	}
	while (t != kNone) {
		const Tile x = decode(q, t);
		// ================= top: the A tile has landed; the B tile is requested
		if (ldA && newerA == (uint32_t)SH::NSB) gb_wait_vm<SH::NSB>(); else gb_wait_vm<0>();
		VKFFT_SYNC_RAW(); // every wave's part of the A tile is in P; Q is free; the previous A tile's ring stores are acknowledged in every wave
		VKFFT_PROF2(0);
		if (tid == 0) {
			if (pending != kNone) { (void)VKFFT_ATOMIC_ADD_U32(p.ctr + pending, 1u); pending = kNone; }
			nextT = VKFFT_ATOMIC_ADD_U32(p.ctr + kFusedCtrTicket + 32u * qT, 1u); // taken in the A phase
		}
		const bool okA = sOk[it][0] != 0u, okB = sOk[it][1] != 0u;
		bool ldB = false;
		uint32_t newerB = 0;
		if (x.liveB && okB) { dma_b(x); ldB = true; }
		if (x.hasA) {
			if (x.liveA) {
				cx<T> v[CPT * EA];
#pragma unroll
				for (int m = 0; m < EA; m++) {
					if constexpr (CPT == 1) v[m] = P[(tauA + m * TPFA) * TCA + cA_];
					else { const cx2<T> u = *(const cx2<T>*)(P + (tauA + m * TPFA) * TCA + cA_); v[m] = u.a; v[EA + m] = u.b; }
				}
				VKFFT_SYNC_RAW(); // the landing image is in registers: P becomes the exchange buffer
				VKFFT_PROF2(1);
				if (p.swapIn) {
#pragma unroll
					for (int m = 0; m < CPT * EA; m++) v[m] = cswap(v[m]);
				}
				if constexpr ((MODE & 8) == 0) {
					pow2_stages<T, SA, 0, TPFA, TCPA, TwLds<T>, CPT, 1>(v, P + cA_, TwLds<T>{twA}, tauA, false);
					VKFFT_PROF2(2);
#pragma unroll
					for (int cc = 0; cc < CPT; cc++) pow2_fs_twiddle_lds<T, SA::LOGE, TPFA, SH::FSLO>(v + cc * EA, fsT, tauA, x.ti * TCA + cA_ + cc);
					VKFFT_PROF2(3);
					if constexpr (SA::NS > 1) VKFFT_SYNC_RAW(); // the last exchange's reads are complete
				}
#pragma unroll
				for (int m = 0; m < EA; m++) {
					if constexpr (CPT == 1) P[(tauA + m * TPFA) * TCPA + cA_] = v[m];
					else *(cx2<T>*)(P + (tauA + m * TPFA) * TCPA + cA_) = cx2<T>{v[m], v[EA + m]};
				}
				VKFFT_SYNC_RAW();
				VKFFT_PROF2(4);
			}
		}
		if (tid == 0) { // the next ticket (wave 0's queue holds nothing younger than the previous phase's stores here)
			uint32_t tn = nextT;
			if (tn >= totOf(qT)) { // this queue is drained: help the next one, leave when every queue is
				tn = kNone;
				if (++tried < Q) { qT = qT + 1u == Q ? 0u : qT + 1u; tn = fetch_sync(); }
			}
			sTk[it ^ 1u][0] = tn; sTk[it ^ 1u][1] = qT; // read by everybody after the barrier at "mid"
			load_flags(qT, tn);                          // taken in the B phase
		}
		VKFFT_PROF2(5);
		if (x.hasA) {
			if (!okA) { // rare: the ring slot's previous tenant has not been read completely yet
				if (tid == 0) { while (VKFFT_ATOMIC_LOAD_U32(p.ctr + depA(q, x.s)) < TPC) VKFFT_SLEEP(); }
				VKFFT_SYNC_RAW();
			}
			if (x.liveA) {
				// per-column contiguous runs into the ring, 16 bytes per lane, write-through
				const GBuf gs = make_gbuf(x.ringA + (uint64_t)(x.ti * TCA) * LA * ES);
				if constexpr (sizeof(T) == 4) {
#pragma unroll
					for (int i = 0; i < CPT * EA / 2; i++) {
						const uint32_t idx = tid + i * NT;
						const uint32_t kp = idx % (LA / 2), cc = idx / (LA / 2);
						gb_store2_x<T, AUX_ST>(gs, (cc * LA + 2u * kp) * ES, P[(2u * kp) * TCPA + cc], P[(2u * kp + 1u) * TCPA + cc]);
					}
				} else {
#pragma unroll
					for (int i = 0; i < CPT * EA; i++) {
						const uint32_t idx = tid + i * NT;
						const uint32_t k = idx % LA, cc = idx / LA;
						gb_store_x<T, AUX_ST>(gs, (cc * LA + k) * ES, 0, P[k * TCPA + cc]);
					}
				}
				newerB = (uint32_t)SH::NRS;
			}
			if (tid == 0) pending = doneA + x.cA;
		}
		VKFFT_PROF2(6);
		// ================= mid: the B tile has landed; the next A tile is requested
		if (x.liveB && !ldB) { // rare: the chunk was not complete at the top
			if (tid == 0) { while (VKFFT_ATOMIC_LOAD_U32(p.ctr + depB(q, x.s)) < TPC) VKFFT_SLEEP(); }
			VKFFT_SYNC_RAW();
			dma_b(x);
			ldB = true; newerB = 0;
		}
		if (ldB) { if (newerB == (uint32_t)SH::NRS) gb_wait_vm<SH::NRS>(); else gb_wait_vm<0>(); }
		VKFFT_SYNC_RAW(); // every wave's part of the B tile is in Q; P is free; the next ticket is visible
		VKFFT_PROF2(8);
		if (tid == 0 && x.hasB) (void)VKFFT_ATOMIC_ADD_U32(p.ctr + doneB + x.cB, 1u); // the ring slot has been read
#if defined(VKFFT_HOSTEMU)
		const uint32_t tn = sTk[it ^ 1u][0], qn = sTk[it ^ 1u][1];
#else
		const uint32_t tn = __builtin_amdgcn_readfirstlane(sTk[it ^ 1u][0]), qn = __builtin_amdgcn_readfirstlane(sTk[it ^ 1u][1]);
#endif
		ldA = false; newerA = 0;
		if (tn != kNone) {
			const Tile y = decode(qn, tn);
			if (y.liveA) { dma_a(y); ldA = true; }
		}
		cx<T> vB[CPT * EB];
		if (x.liveB) {
#pragma unroll
			for (int m = 0; m < EB; m++) {
				if constexpr (CPT == 1) vB[m] = Qb[(tauB + m * TPFB) * TCB + cB_];
				else { const cx2<T> u = *(const cx2<T>*)(Qb + (tauB + m * TPFB) * TCB + cB_); vB[m] = u.a; vB[EB + m] = u.b; }
			}
			VKFFT_SYNC_RAW(); // Q becomes the exchange buffer
			VKFFT_PROF2(9);
			if constexpr ((MODE & 8) == 0) pow2_stages<T, SB, 0, TPFB, TCPB, TwLds<T>, CPT, 1>(vB, Qb + cB_, TwLds<T>{twB}, tauB, false);
			if (p.swapOut) {
#pragma unroll
				for (int m = 0; m < CPT * EB; m++) vB[m] = cswap(vB[m]);
			}
			const T sc = (T)p.scale;
			if (sc != (T)1) {
#pragma unroll
				for (int m = 0; m < CPT * EB; m++) vB[m] = cscale(vB[m], sc);
			}
		}
		VKFFT_PROF2(10);
		if (tid == 0) { sOk[it ^ 1u][0] = fA >= TPC; sOk[it ^ 1u][1] = fB >= TPC; } // flags of the next ticket: read by everybody after the barrier at its "top"
		if (x.liveB) {
			// natural order X[k0 + n0*k1]: k1 = tauB + m*TPFB, k0 = ti*TCB + cB_
			const GBuf gout = make_gbuf((cx<T>*)p.out + ((int64_t)x.bB * p.outBatchStride + (int64_t)(x.ti * TCB)));
			const uint32_t laneB = (tauB * p.n0 + cB_) * ES, stepB = (uint32_t)TPFB * p.n0 * ES;
#pragma unroll
			for (int m = 0; m < EB; m++) {
				if constexpr (CPT == 1) gb_store_x<T, AUX_HBM>(gout, laneB, m * stepB, vB[m]);
				else gb_store2_x<T, AUX_HBM>(gout, laneB + m * stepB, vB[m], vB[EB + m]);
			}
			newerA = (uint32_t)SH::NSB;
		}
		VKFFT_PROF2(11);
#if !defined(VKFFT_HOSTEMU)
		if constexpr ((MODE & 4) != 0) pc[7] += 1000000ull; // (slot 7 also collects the time of the rare polls through VKFFT_PROF2(7): never used)
#endif
		t = tn; q = qn; it ^= 1u;
	}
#if !defined(VKFFT_HOSTEMU)
	if constexpr ((MODE & 4) != 0) { // thread 0 (the ticket thread) and thread 64 (a DMA wave) of every workgroup
		if ((tid == 0 || tid == 64) && p.prof) { for (int i = 0; i < 12; i++) p.prof[((size_t)blockIdx.x * 2 + (tid >> 6)) * 12 + i] = pc[i]; }
	}
#endif
	// ---- exit: publish the last A tile, then the last workgroup out resets the counters for the next launch
	gb_wait_vm<0>();
	VKFFT_SYNC_RAW();
	if (tid == 0) {
		if (pending != kNone) (void)VKFFT_ATOMIC_ADD_U32(p.ctr + pending, 1u);
		VKFFT_VMEM_DRAIN(); // this workgroup's counter updates have been performed
		sOk[0][0] = VKFFT_ATOMIC_ADD_U32(p.ctr + kFusedCtrExit, 1u) == gridDim.x - 1u;
	}
	VKFFT_SYNC_RAW();
	if (sOk[0][0]) {
		for (uint32_t i = tid; i < kFusedCtrDone + 2u * p.C; i += NT) p.ctr[i] = 0u;
	}
}

template <typename T, typename SA, int TCA, typename SB, int TCB, int MODE, int CPT, int WPC> void pow2_fused2_launch(const FusedParams& prm, dim3 grid, hipStream_t s) {
	hipLaunchKernelGGL((pow2_fused2_kernel<T, SA, TCA, SB, TCB, MODE, CPT, WPC>), grid, dim3(Fused2Shape<T, SA, TCA, SB, TCB, CPT>::NT), 0, s, prm);
}

} // namespace vkfft_mi355x
