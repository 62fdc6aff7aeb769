// Fused Four-Step, second generation: the ticket queue, ring and completion counters of kernel_pow2_fused.h with a different data path.
//
// What limited the first generation (profiles/r02_*): a workgroup serialised  load tile -> wait -> compute -> store  and a CU's
// registers were full with two such tiles, so HBM latency sat in the open and a CU had too few bytes in flight to keep HBM busy (a CU needs
// ~46 KB of reads in flight all the time: 10.9 B/clk at ~2 us).  Here a workgroup is NC compute threads plus ONE service wave:
//
//   * the service wave owns everything that waits on memory: it draws the tickets, samples the dependency counters, publishes the completions and
//     brings the A tiles (columns of the n0 x n1 view in HBM) into LDS by LDS-DMA (memops.h gb_dma16: buffer_load_dwordx4 ... lds, no register
//     ever holds the data) a WHOLE TICKET ahead: tile i+1 is requested as soon as tile i has landed, so every workgroup always has one tile of HBM
//     reads in flight.  Its vector-memory queue holds nothing anybody else waits for, so the in-order wait counter is no obstacle: one vmcnt(0) per
//     ticket, just ahead of the "top" barrier, retires the tile together with the ticket and the flags requested a ticket earlier.
//     tools/probe_dma.hip: column tiles streamed this way reach 5.8 TB/s (128-byte row segments) / 6.0 TB/s (256-byte) read + write on MI355X;
//   * the compute waves never wait for HBM.  A phase: landed tile (LDS) -> registers -> FFT over n0 (exchange in the same buffer) -> Four-Step twiddle
//     -> transposition through the buffer -> write-through ring stores.  B phase: the chunk's intermediate comes from the ring (Infinity Cache, short
//     latency) straight into registers, requested BEFORE the A tile's ring stores are issued and consumed after them -> FFT over n1 -> natural-order
//     stores to HBM.  Every table (stage twiddles of both factors, the two-level Four-Step table) is read from LDS;
//   * two LDS buffers alternate between "landing zone of the next A tile" and "exchange buffer of the current ticket".
//
// All waves run the same barrier sequence (the service wave shadows the barriers of the compute sections).  Barriers are VKFFT_SYNC_RAW (no fence:
// a fence would drain the service wave's DMA); the one place where a compute wave must know that its stores have been acknowledged uses a counted
// s_waitcnt (the queue of a wave is in order: "all but the N youngest").
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

// barriers pow2_stages<..., RAW = 1> executes for schedule SCH (the service wave runs the same number)
template <typename SCH> constexpr int pow2_stage_barriers() { return SCH::NS <= 1 ? 0 : (SCH::NS - 1) + (SCH::NS - 2); }
template <int N> __device__ inline void fused2_shadow_barriers() {
#pragma unroll
	for (int i = 0; i < N; i++) VKFFT_SYNC_RAW();
}

template <typename T, typename SA, int TCA, typename SB, int TCB, int CPT> struct Fused2Shape {
	static constexpr int ES = (int)sizeof(cx<T>);
	static constexpr int LA = 1 << SA::LOGN, EA = 1 << SA::LOGE, TPFA = LA / EA, TCPA = TCA + (CPT == 2 ? 2 : 1);
	static constexpr int LB = 1 << SB::LOGN, EB = 1 << SB::LOGE, TPFB = LB / EB, TCPB = TCB + (CPT == 2 ? 2 : 1);
	static constexpr int NC = TPFA * TCA / CPT, NT = NC + 64;  // compute threads; + the service wave
	static constexpr int BUFN = LA * TCPA > LB * TCPB ? LA * TCPA : LB * TCPB; // complex elements per buffer (landing image LA x TCA, exchange images L x TCP)
	static constexpr int LUTA = SA::lutTotal(), LUTB = SB::lutTotal();
	static constexpr int LOGN = SA::LOGN + SB::LOGN, FSLO = (LOGN + 1) / 2, FSN = (1 << FSLO) + (1 << (LOGN - FSLO));
	// LDS-DMA: one wave-instruction = 1 KiB = RPI rows of the tile, LPR lanes per row
	static constexpr int RBA = TCA * ES, RPIA = 1024 / RBA, LPRA = RBA / 16, NIA = LA * RBA / 1024;
	static constexpr int NSB = ES == 8 && CPT == 2 ? EB : CPT * EB; // HBM stores per compute thread and ticket
	static constexpr int ldsBytes = (2 * BUFN + LUTA + LUTB + FSN) * ES + 128;
	static_assert(NC % 64 == 0 && NC == TPFB * TCB / CPT && LA * TCA == LB * TCB, "both phases run on the same compute threads and tile size");
	static_assert(1024 % RBA == 0 && RBA >= 16, "row segments of 16 ... 1024 bytes");
	static_assert(NSB < 64, "the counted wait fits the 6-bit vmcnt field");
};

// MODE bit 1: non-temporal hint on the HBM side; development only: bit 2 per-phase cycle sums, bit 3 without the FFT arithmetic, bit 4 without the ring
// traffic, bit 5 plain instead of write-through ring stores (the last three give wrong results: timing experiments)
// WPC: workgroups per CU the shape is built for (register budget through __launch_bounds__; the LDS footprint must allow it)
template <typename T, typename SA, int TCA, typename SB, int TCB, int MODE, int CPT, int WPC>
__global__ void __launch_bounds__((Fused2Shape<T, SA, TCA, SB, TCB, CPT>::NT), ((WPC * Fused2Shape<T, SA, TCA, SB, TCB, CPT>::NT + 255) / 256))
pow2_fused2_kernel(const FusedParams p) {
	using SH = Fused2Shape<T, SA, TCA, SB, TCB, CPT>;
	static_assert(CPT == 1 || (CPT == 2 && sizeof(T) == 4), "two columns per thread: fp32 only");
	static_assert(SH::ldsBytes * WPC <= 163840, "LDS footprint");
	constexpr int LA = SH::LA, EA = SH::EA, TPFA = SH::TPFA, TCPA = SH::TCPA, EB = SH::EB, TPFB = SH::TPFB, TCPB = SH::TCPB, NC = SH::NC, BUFN = SH::BUFN;
	constexpr uint32_t ES = (uint32_t)SH::ES;
	constexpr int AUX_SC = 16, AUX_ST = (MODE & 32) ? 0 : 16, AUX_HBM = (MODE & 2) ? 2 : 0;
	constexpr bool RING = (MODE & 16) == 0;
	constexpr bool ARITH = (MODE & 8) == 0;
	constexpr uint32_t kNone = 0xffffffffu;
	__shared__ cx<T> buf[2 * BUFN + SH::LUTA + SH::LUTB + SH::FSN];
	__shared__ uint32_t sSlot[2][4]; // written by the ticket thread ahead of every "top" barrier: {next ticket, its queue, okA, okB of the current ticket}
	__shared__ uint32_t sLast;
	cx<T>* const twA = buf + 2 * BUFN;
	cx<T>* const twB = twA + SH::LUTA;
	cx<T>* const fsT = twB + SH::LUTB;
	const uint32_t tid = threadIdx.x, lane = tid & 63u;
#if defined(VKFFT_HOSTEMU)
	const bool svc = tid >= (uint32_t)NC;
#else
	const bool svc = __builtin_amdgcn_readfirstlane(tid >> 6) == (uint32_t)(NC / 64);
#endif
	const bool svc0 = tid == (uint32_t)NC; // the ticket thread
	for (uint32_t i = tid; i < (uint32_t)SH::LUTA; i += SH::NT) twA[i] = ((const cx<T>*)p.lutA)[i];
	for (uint32_t i = tid; i < (uint32_t)SH::LUTB; i += SH::NT) twB[i] = ((const cx<T>*)p.lutB)[i];
	for (uint32_t i = tid; i < (uint32_t)SH::FSN; i += SH::NT) fsT[i] = ((const cx<T>*)p.tw4)[i];
	const uint32_t logTPC = p.logG + p.logTiles, TPC = 1u << logTPC;
	const uint32_t doneA = kFusedCtrDone, doneB = kFusedCtrDone + p.C;
	const uint64_t nPts = (uint64_t)p.n0 * p.n1;
	const uint32_t Q = p.Q;
	auto cqOf = [&](uint32_t q) -> uint32_t { return (p.C + Q - 1u - q) / Q; };
	auto totOf = [&](uint32_t q) -> uint32_t { const uint32_t c = cqOf(q); return c ? (c + p.D) << logTPC : 0u; };
	// counters a ticket of slot s of queue q depends on: the ring slot's previous tenant read completely (A), the chunk written completely (B)
	auto depA = [&](uint32_t q, uint32_t s) -> uint32_t { return (s < cqOf(q) && s >= p.NS) ? doneB + q + Q * (s - p.NS) : kNone; };
	auto depB = [&](uint32_t q, uint32_t s) -> uint32_t { return (s >= p.D && s - p.D < cqOf(q)) ? doneA + q + Q * (s - p.D) : kNone; };
	// per-thread tile coordinates of the compute threads (the same for every tile)
	const uint32_t cA_ = (tid % (TCA / CPT)) * CPT, tauA = tid / (TCA / CPT);
	const uint32_t cB_ = (tid % (TCB / CPT)) * CPT, tauB = tid / (TCB / CPT);
	const uint32_t dmaVoffA = (lane / SH::LPRA) * p.n1 * ES + (lane % SH::LPRA) * 16u, dmaStepA = (uint32_t)SH::RPIA * p.n1 * ES;

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
	// (service wave) HBM -> LDS: rows j0 = 0 .. LA-1 of columns ti*TCA .. +TCA of transform bA, one KiB per instruction.  The issue of an LDS-DMA blocks
	// while the CU's memory pipeline is full (measured: ~100 cycles per instruction under load, 3.5 k cycles for a 32 KiB tile) and the service wave
	// has to be at every barrier of the compute waves on time, so a tile is requested in slices of DSL instructions, one slice after every barrier.
	GDma dmaG = make_gdma(p.in);
	char* dmaDst = (char*)buf;
	uint32_t dmaNext = (uint32_t)SH::NIA; // next instruction of the tile under way (NIA: none)
	constexpr int DSL = (SH::NIA + 5) / 6;
	auto dma_begin = [&](const Tile& x, cx<T>* dst) {
		dmaG = make_gdma((const cx<T>*)p.in + ((int64_t)x.bA * p.inBatchStride + (int64_t)(x.ti * TCA)));
		dmaDst = (char*)dst; dmaNext = 0;
	};
	auto dma_slice = [&](int n) {
		for (int k = 0; k < n && dmaNext < (uint32_t)SH::NIA; k++, dmaNext++) gb_dma16<AUX_HBM>(dmaDst + dmaNext * 1024u, dmaG, dmaVoffA, dmaNext * dmaStepA);
	};
#if !defined(VKFFT_HOSTEMU)
	unsigned long long pc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ptk = 0;
#define VKFFT_PROF2(i) do { if constexpr ((MODE & 4) != 0) { const unsigned long long now = __builtin_readcyclecounter(); pc[i] += now - ptk; ptk = now; } } while (0)
	if constexpr ((MODE & 4) != 0) ptk = __builtin_readcyclecounter();
#else
#define VKFFT_PROF2(i) do { } while (0)
#endif

	// ---- ticket thread (lane 0 of the service wave).  qT = the queue it draws from: its XCD's first (speed only), the others once that one is
	// drained (completion must not depend on placement).  At every "top" it takes the ticket requested a ticket ago (= the NEXT ticket: its A tile is
	// requested right away) and the flags requested a ticket ago (= those of the CURRENT ticket), then makes the next two requests.
	uint32_t qT = Q > 1 ? fused_xcc_id() % Q : 0u, tried = 0, pendA = kNone, pendB = kNone, tReq = kNone, qReq = 0, fA = TPC, fB = TPC;
	auto fetch_sync = [&]() -> uint32_t {
		for (;;) {
			const uint32_t t = VKFFT_ATOMIC_ADD_U32(p.ctr + kFusedCtrTicket + 32u * qT, 1u);
			if (t < totOf(qT)) return t;
			if (++tried >= Q) return kNone;
			qT = qT + 1u == Q ? 0u : qT + 1u;
		}
	};
	auto load_flags = [&](uint32_t q, uint32_t t) { // requests the dependency counters of ticket t
		fA = TPC; fB = TPC;
		if (t == kNone) return;
		const uint32_t s = t >> logTPC, dA = depA(q, s), dB = depB(q, s);
		if (dA != kNone) fA = VKFFT_ATOMIC_LOAD_U32(p.ctr + dA);
		if (dB != kNone) fB = VKFFT_ATOMIC_LOAD_U32(p.ctr + dB);
	};
	if (svc0) {
		const uint32_t t0 = fetch_sync();
		sSlot[1][0] = t0; sSlot[1][1] = qT;
		load_flags(qT, t0); // taken at the first "top"
		if (t0 != kNone) { tReq = VKFFT_ATOMIC_ADD_U32(p.ctr + kFusedCtrTicket + 32u * qT, 1u); qReq = qT; }
	}
	VKFFT_SYNC_RAW(); // tables and the first ticket are in LDS
#if defined(VKFFT_HOSTEMU)
	uint32_t t = sSlot[1][0], q = sSlot[1][1];
#else
	uint32_t t = __builtin_amdgcn_readfirstlane(sSlot[1][0]), q = __builtin_amdgcn_readfirstlane(sSlot[1][1]);
#endif
	uint32_t it = 0;
	if (svc && t != kNone) {
		const Tile x = decode(q, t);
		if (x.liveA) { dma_begin(x, buf); dma_slice(SH::NIA); } // the first A tile -> buffer 0
	}
	bool hbmTail = false; // (compute waves) the youngest vector-memory instructions of this wave are the NSB HBM stores of the previous ticket
	while (t != kNone) {
		const Tile x = decode(q, t);
		cx<T>* const P = buf + it * BUFN;             // this ticket's buffer: landing zone of its A tile, then exchange buffer of both phases
		cx<T>* const Pn = buf + (it ^ 1u) * BUFN;      // the next ticket's landing zone
		// ================= top
		if (svc) {
			dma_slice(SH::NIA); // (what is left of the tile: nothing on a full ticket)
			gb_wait_vm<0>(); // the A tile has landed, and with it everything the ticket thread asked for a ticket ago
			if (svc0) {
				uint32_t tn = tReq, qn = qReq;
				if (tn >= totOf(qn)) { // that queue is drained: help the next one, leave when every queue is
					tn = kNone;
					if (++tried < Q) { qT = qT + 1u == Q ? 0u : qT + 1u; tn = fetch_sync(); qn = qT; }
				}
				sSlot[it][0] = tn; sSlot[it][1] = qn; sSlot[it][2] = fA >= TPC; sSlot[it][3] = fB >= TPC;
				load_flags(qn, tn); // taken at the next "top", when tn is the current ticket
				tReq = kNone;
				if (tn != kNone) { tReq = VKFFT_ATOMIC_ADD_U32(p.ctr + kFusedCtrTicket + 32u * qT, 1u); qReq = qT; }
			}
		} else {
			// this wave's ring stores of the previous ticket are acknowledged, its B tile has been taken out of the ring
			if (hbmTail) gb_wait_vm<SH::NSB>(); else gb_wait_vm<0>();
		}
		VKFFT_SYNC_RAW(); // [top] the A tile is in P; Pn is free; every wave's ring stores of the previous ticket are acknowledged
		VKFFT_PROF2(0);
#if defined(VKFFT_HOSTEMU)
		const uint32_t tn = sSlot[it][0], qn = sSlot[it][1];
#else
		const uint32_t tn = __builtin_amdgcn_readfirstlane(sSlot[it][0]), qn = __builtin_amdgcn_readfirstlane(sSlot[it][1]);
#endif
		const bool okA = sSlot[it][2] != 0u, okB = sSlot[it][3] != 0u; // sampled a ticket ago: a stale "no" costs a poll, never correctness
		if (svc) {
			if (svc0) {
				if (pendA != kNone) { (void)VKFFT_ATOMIC_ADD_U32(p.ctr + pendA, 1u); pendA = kNone; }
				if (pendB != kNone) { (void)VKFFT_ATOMIC_ADD_U32(p.ctr + pendB, 1u); pendB = kNone; }
			}
			if (tn != kNone) { // the next A tile: on its way for the whole of this ticket
				const Tile y = decode(qn, tn);
				if (y.liveA) { dma_begin(y, Pn); dma_slice(DSL); }
			}
		}
		// ----------------- A phase
		cx<T> vB[CPT * EB];
		const GBuf gsB = make_gbuf(x.ringB + (uint64_t)(x.ti * TCB) * ES);
		const uint32_t laneB = (tauB * p.n0 + cB_) * ES, stepB = (uint32_t)TPFB * p.n0 * ES;
		const uint32_t laneBr = RING ? laneB : kGbInvalid;
		auto load_b = [&]() { // ring -> registers: rows j1 = tauB + m*TPFB (pitch n0) of k0 = ti*TCB + cB_
#pragma unroll
			for (int m = 0; m < EB; m++) {
				if constexpr (CPT == 1) vB[m] = gb_load_x<T, AUX_SC>(gsB, laneBr, m * stepB);
				else gb_load2_x<T, AUX_SC>(gsB, laneBr, m * stepB, vB[m], vB[EB + m]);
			}
		};
		if (x.liveA) {
			cx<T> v[CPT * EA];
			if (!svc) {
#pragma unroll
				for (int m = 0; m < EA; m++) {
					if constexpr (CPT == 1) v[m] = P[(tauA + m * TPFA) * TCA + cA_];
					else { const cx2<T> u = *(const cx2<T>*)(P + (tauA + m * TPFA) * TCA + cA_); v[m] = u.a; v[EA + m] = u.b; }
				}
			}
			VKFFT_SYNC_RAW(); // the landing image is in registers: P becomes the exchange buffer
			if (svc) dma_slice(DSL);
			VKFFT_PROF2(1);
			if (!svc) {
				if (p.swapIn) {
#pragma unroll
					for (int m = 0; m < CPT * EA; m++) v[m] = cswap(v[m]);
				}
				if constexpr (ARITH) {
					pow2_stages<T, SA, 0, TPFA, TCPA, TwLds<T>, CPT, 1>(v, P + cA_, TwLds<T>{twA}, tauA, false);
					VKFFT_PROF2(2);
#pragma unroll
					for (int cc = 0; cc < CPT; cc++) pow2_fs_twiddle_lds<T, SA::LOGE, TPFA, SH::FSLO>(v + cc * EA, fsT, tauA, x.ti * TCA + cA_ + cc);
					VKFFT_PROF2(3);
				}
			} else if constexpr (ARITH) {
#pragma unroll
				for (int i = 0; i < pow2_stage_barriers<SA>(); i++) { VKFFT_SYNC_RAW(); dma_slice(DSL); }
			}
			if constexpr (ARITH && SA::NS > 1) VKFFT_SYNC_RAW(); // the last exchange's reads are complete
			if (svc) dma_slice(DSL);
			if (!svc) {
#pragma unroll
				for (int m = 0; m < EA; m++) {
					if constexpr (CPT == 1) P[(tauA + m * TPFA) * TCPA + cA_] = v[m];
					else *(cx2<T>*)(P + (tauA + m * TPFA) * TCPA + cA_) = cx2<T>{v[m], v[EA + m]};
				}
			}
			VKFFT_SYNC_RAW();
			if (svc) dma_slice(DSL);
			VKFFT_PROF2(4);
		}
		if (!svc && x.liveB && okB) load_b(); // the B tile: requested ahead of the ring stores, consumed after them
		if (x.hasA && !okA) { // rare: the ring slot's previous tenant has not been read completely yet
			if (svc0) { while (VKFFT_ATOMIC_LOAD_U32(p.ctr + depA(q, x.s)) < TPC) VKFFT_SLEEP(); }
			VKFFT_SYNC_RAW();
		}
		if (x.liveA && !svc && RING) {
			// per-column contiguous runs into the ring, 16 bytes per lane, write-through
			const GBuf gs = make_gbuf(x.ringA + (uint64_t)(x.ti * TCA) * LA * ES);
			if constexpr (sizeof(T) == 4) {
#pragma unroll
				for (int i = 0; i < CPT * EA / 2; i++) {
					const uint32_t idx = tid + i * NC;
					const uint32_t kp = idx % (LA / 2), cc = idx / (LA / 2);
					gb_store2_x<T, AUX_ST>(gs, (cc * LA + 2u * kp) * ES, P[(2u * kp) * TCPA + cc], P[(2u * kp + 1u) * TCPA + cc]);
				}
			} else {
#pragma unroll
				for (int i = 0; i < CPT * EA; i++) {
					const uint32_t idx = tid + i * NC;
					const uint32_t k = idx % LA, cc = idx / LA;
					gb_store_x<T, AUX_ST>(gs, (cc * LA + k) * ES, 0, P[k * TCPA + cc]);
				}
			}
		}
		VKFFT_PROF2(6);
		// ----------------- B phase
		if (x.liveB && !okB) { // rare: the chunk was not complete when its flag was sampled
			if (svc0) { while (VKFFT_ATOMIC_LOAD_U32(p.ctr + depB(q, x.s)) < TPC) VKFFT_SLEEP(); }
			VKFFT_SYNC_RAW();
			if (!svc) load_b();
		}
		hbmTail = false;
		if (x.liveB) {
			VKFFT_SYNC_RAW(); // [mid] the ring stores' LDS reads are complete: P is the exchange buffer of the B phase
			if (svc) dma_slice(DSL);
			VKFFT_PROF2(8);
			if (!svc) {
				if constexpr (ARITH) pow2_stages<T, SB, 0, TPFB, TCPB, TwLds<T>, CPT, 1>(vB, P + cB_, TwLds<T>{twB}, tauB, false);
				VKFFT_PROF2(10);
				if (p.swapOut) {
#pragma unroll
					for (int m = 0; m < CPT * EB; m++) vB[m] = cswap(vB[m]);
				}
				const T sc = (T)p.scale;
				if (sc != (T)1) {
#pragma unroll
					for (int m = 0; m < CPT * EB; m++) vB[m] = cscale(vB[m], sc);
				}
				// natural order X[k0 + n0*k1]: k1 = tauB + m*TPFB, k0 = ti*TCB + cB_
				const GBuf gout = make_gbuf((cx<T>*)p.out + ((int64_t)x.bB * p.outBatchStride + (int64_t)(x.ti * TCB)));
#pragma unroll
				for (int m = 0; m < EB; m++) {
					if constexpr (CPT == 1) gb_store_x<T, AUX_HBM>(gout, laneB, m * stepB, vB[m]);
					else gb_store2_x<T, AUX_HBM>(gout, laneB + m * stepB, vB[m], vB[EB + m]);
				}
				hbmTail = true;
			} else if constexpr (ARITH) {
#pragma unroll
				for (int i = 0; i < pow2_stage_barriers<SB>(); i++) { VKFFT_SYNC_RAW(); dma_slice(DSL); }
			}
		}
		if (svc0) { // published after the next "top" (or at the exit), when every compute wave has had its stores acknowledged and its B tile in registers
			pendA = x.hasA ? doneA + x.cA : kNone;
			pendB = x.hasB ? doneB + x.cB : kNone;
		}
		VKFFT_PROF2(11);
#if !defined(VKFFT_HOSTEMU)
		if constexpr ((MODE & 4) != 0) pc[7] += 1000000ull;
#endif
		t = tn; q = qn; it ^= 1u;
	}
#if !defined(VKFFT_HOSTEMU)
	if constexpr ((MODE & 4) != 0) { // thread 0 (a compute wave) and the ticket thread of every workgroup
		if ((tid == 0 || svc0) && p.prof) { for (int i = 0; i < 12; i++) p.prof[((size_t)blockIdx.x * 2 + (svc0 ? 1 : 0)) * 12 + i] = pc[i]; }
	}
#endif
	// ---- exit: publish the last ticket, then the last workgroup out resets the counters for the next launch
	gb_wait_vm<0>();
	VKFFT_SYNC_RAW();
	if (svc0) {
		if (pendA != kNone) (void)VKFFT_ATOMIC_ADD_U32(p.ctr + pendA, 1u);
		if (pendB != kNone) (void)VKFFT_ATOMIC_ADD_U32(p.ctr + pendB, 1u);
		VKFFT_VMEM_DRAIN(); // this workgroup's counter updates have been performed
		sLast = VKFFT_ATOMIC_ADD_U32(p.ctr + kFusedCtrExit, 1u) == gridDim.x - 1u;
	}
	VKFFT_SYNC_RAW();
	if (sLast) {
		for (uint32_t i = tid; i < kFusedCtrDone + 2u * p.C; i += SH::NT) p.ctr[i] = 0u;
	}
}

template <typename T, typename SA, int TCA, typename SB, int TCB, int MODE, int CPT, int WPC> void pow2_fused2_launch(const FusedParams& prm, dim3 grid, hipStream_t s) {
	hipLaunchKernelGGL((pow2_fused2_kernel<T, SA, TCA, SB, TCB, MODE, CPT, WPC>), grid, dim3(Fused2Shape<T, SA, TCA, SB, TCB, CPT>::NT), 0, s, prm);
}

} // namespace vkfft_mi355x
