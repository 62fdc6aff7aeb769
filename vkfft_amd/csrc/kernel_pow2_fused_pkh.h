// Fused Four-Step of 2^21 / 2^22 on packed pairs with tiles of TWO HALVES (round 5).
//
// A 2048-point factor with 128-byte HBM segments is a tile of 2048 x 16 points = 256 KiB: half a CU's register file.  Round 4 held it in 1024 threads x 64 data
// registers (128 VGPRs per thread: 35-37 spilled, no second tile, load -> compute -> store in series: 2.2-2.35 TB/s).  Here the workgroup has 512 threads with
// 256 registers and the tile is two half-tiles of 1024 x 16 points that are INDEPENDENT sub-transforms until one last radix-2 layer:
//   * a 2048-point factor: decimation in time — the even rows and the odd rows of the tile are two 1024-point transforms (the packed stages of
//     kernel_pow2_pk.h, 88 registers), combined by X[k] = E[k] + w^k O[k], X[k + 1024] = E[k] - w^k O[k] in registers (the same thread holds point k of both);
//   * a 1024-point factor (2^21): the two halves are the column groups [0, 16) and [16, 32) of a 32-column tile, no combine.
// A half is 64 registers per thread.  With three such sets + the stages' temporaries a workgroup always has one half computing and one or two halves in
// flight — the software pipelining of kernel_pow2_fused_pipe.h at half-tile granularity:
//
//   top:   [A.e and A.o of THIS ticket in flight, the previous ticket's stores to HBM draining]
//          A.e stages -> (A.o landed: the queue is empty, thread 0 draws the next ticket) -> A.o stages -> combine, Four-Step twiddle,
//          turn + ring stores of the first half, request B.e, turn + ring stores of the second half
//          drain, barrier, publish doneA          request B.o from the ring and A.e of the NEXT ticket from HBM
//          B.e stages [B.o, next A.e in flight] -> B.o stages -> combine, stores to HBM        request A.o of the next ticket
// (one wait counter covers loads AND stores on this architecture: a load issued behind stores is available only when they are — the order above keeps
//  every wait short — and a reload from scratch would drain the queue: the loop must not spill.)
//
// Queues, ring, flags and coherence rules are those of kernel_pow2_fused.h; the ring unit is the 16-byte (Re p0, Re p1, Im p0, Im p1) of kernel_pow2_fused_pk.h.
// Reference shape replaced: three axis uploads of a 2^21+ sequence (vkFFT_Scheduler.h:2590-2893, vkFFT_4step.h:31).
#pragma once
#include "kernel_pow2_fused.h"
#include "kernel_pow2_pk.h"

namespace vkfft_mi355x {

// SH: the schedule of a half (L = 1024 points: 512 threads, one workgroup per CU; L = 512: 256 threads, two workgroups per CU — 256 registers per thread either way).
// SPLITA / SPLITB: the factor of that side has 2 L points (two row-interleaved halves + combine) instead of L (two column groups)
template <typename T, typename SH, int SPLITA, int SPLITB, int MODE, int TWL>
__global__ void __launch_bounds__(((1 << SH::LOGN) >> SH::LOGE) * 8, 2) pow2_fused_pkh_kernel(const FusedParams p) {
	static_assert(sizeof(T) == 4 && SH::bits[3] == 0, "two fp32 columns per thread; three stages per half");
	typedef Pow2Sched<SH::bits[0], SH::bits[1], SH::bits[2], 1> SF; // (table layout of a 2 L-point factor: SH's runs, then w_2L^k, k < L)
	constexpr int L = 1 << SH::LOGN, E = 1 << SH::LOGE, TPF = L / E, TC = 16, NT = TPF * TC / 2;
	constexpr int LA = SPLITA ? 2 * L : L, LB = SPLITB ? 2 * L : L;
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	constexpr int AUX_SC = 16, AUX_ST = 16;       // ring: memory-side loads, write-through stores
	constexpr int AUX_HBM = (MODE & 2) ? 2 : 0;   // streamed side: non-temporal hint
	constexpr int PLN = ((int)pow2_lean_plane_elems<SH, TC>() + 3) & ~3; // (a multiple of 4 reals: what follows the plane is 16-byte aligned)
	constexpr int LUTA = SPLITA ? SF::lutTotal() : SH::lutTotal(), LUTB = SPLITB ? SF::lutTotal() : SH::lutTotal(), LUTC = SH::lutTotal(); // LUTC: where w_2048^k starts
	constexpr int ROWL = TWL ? 2 * LA : 0; // the row table of the Four-Step twiddle beside the stage twiddles (16 bytes per point of the first factor)
	constexpr int TWG = 8;
	__shared__ __attribute__((aligned(16))) cx<T> lds[PLN / 2 + ROWL + (TWL ? LUTA + LUTB : 0)];
	T* const plane = (T*)lds;
	__shared__ uint32_t sTicket[2], sOkA[2], sOkB[2];
	const uint32_t tid = threadIdx.x;
	cx<T>* const rowL = lds + PLN / 2;
	cx<T>* const twA = rowL + ROWL;
	cx<T>* const twB = twA + LUTA;
	if constexpr (TWL) {
		for (uint32_t i = tid; i < (uint32_t)ROWL; i += NT) rowL[i] = ((const cx<T>*)p.rowTab)[i];
		for (uint32_t i = tid; i < (uint32_t)LUTA; i += NT) twA[i] = ((const cx<T>*)p.lutA)[i];
		for (uint32_t i = tid; i < (uint32_t)LUTB; i += NT) twB[i] = ((const cx<T>*)p.lutB)[i];
	}
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
	const uint32_t cl = (tid % (TC / 2)) * 2, tau = tid / (TC / 2); // first of this thread's two adjacent columns within a half, its first point
	pk4<T> rawE[E], rawO[E]; // the halves of an A tile while they travel
	cxp<T> bE[E], bO[E];     // the halves of a B tile (ring units: pair form already)
	// half h of the A tile of ticket tt (of the CURRENT queue), requested from HBM.  2048-point factor: rows 2i + h; else columns [16 h, 16 h + 16) of 32
	auto requestA = [&](uint32_t tt, int h, pk4<T>* raw) {
		const uint32_t s = tt >> logTPC, r = tt & (TPC - 1u);
		const uint32_t f = r >> p.logTiles, ti = r & ((1u << p.logTiles) - 1u);
		const uint32_t cA = q + Q * s;
		const uint32_t b = ((p.reverse ? p.C - 1u - cA : cA) << p.logG) + f;
		const bool live = s < Cq && b < p.batch;
		const GBuf gin = make_gbuf((const cx<T>*)p.in + ((int64_t)(live ? b : 0u) * p.inBatchStride + ti * (SPLITA ? 16 : 32)));
		const uint32_t lane = !live ? kGbInvalid : SPLITA ? ((2u * tau + (uint32_t)h) * p.n1 + cl) * ES : (tau * p.n1 + 16u * (uint32_t)h + cl) * ES;
		VKFFT_OPAQUE_ZERO(oq); // (the E multiples of the step are recomputed at every request: hoisted out of the persistent loop they are 16 scalar registers per request kind, and the overflow goes to VGPRs and to scratch)
		const uint32_t step = (uint32_t)(SPLITA ? 2 * TPF : TPF) * p.n1 * ES + oq;
#pragma unroll
		for (int m = 0; m < E; m++) raw[m] = gb_load_aos2<T, AUX_HBM>(gin, lane, m * step);
	};
	VKFFT_PKPROF_DECL;
	if (tid == 0) draw(0);
	uint32_t it = 0;
	VKFFT_SYNC();
	requestA(sTicket[0], 0, rawE); // invariant at the head of the loop: both halves of the A tile of the ticket about to be read have been requested
	requestA(sTicket[0], 1, rawO);
	uint32_t pendB = kNone; // thread 0: the ring slot whose release is still owed (published once EVERY wave has its B.o half in registers: after the next barrier)
	for (;;) {
		VKFFT_SYNC(); // S1: ticket visible; the plane is free; every wave has read both halves of the previous ticket's ring slot
		VKFFT_PKPROF(0);
		if (tid == 0 && pendB != kNone) { (void)VKFFT_ATOMIC_ADD_U32(p.ctr + pendB, 1u); pendB = kNone; }
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
			requestA(sTicket[it], 0, rawE);
			requestA(sTicket[it], 1, rawO);
			continue;
		}
		const uint32_t okA = sOkA[it], okB = sOkB[it];
		it ^= 1u;
		VKFFT_OPAQUE_ZERO(oz);
		const GBuf gtw = make_gbuf((const char*)p.tw4 + oz);
		const uint32_t s = t >> logTPC, r = t & (TPC - 1u);
		const uint32_t f = r >> p.logTiles, ti = r & ((1u << p.logTiles) - 1u);
		const bool hasA = s < Cq, hasB = s >= p.D && s - p.D < Cq;
		// ---- the B tile's halves: ring -> registers.  2048-point factor: rows (j1) 2i + h of 16 neighbouring k0; else k0 groups [16 h, 16 h + 16) of 32
		const uint32_t sB = s - p.D, cB = q + Q * sB;
		const uint32_t bB = ((p.reverse ? p.C - 1u - cB : cB) << p.logG) + f;
		const bool liveB = hasB && bB < p.batch;
		const uint32_t k00 = ti * (SPLITB ? 16 : 32);
		const char* const sbaseB = (const char*)p.scratch + ((uint64_t)(((q * p.NS + (hasB ? sB % p.NS : 0u)) << p.logG) + f) * nPts) * ES;
		auto requestB = [&](int h, cxp<T>* v) {
			const GBuf gsB = make_gbuf(sbaseB + (uint64_t)k00 * ES);
			const uint32_t lane = !liveB ? kGbInvalid : SPLITB ? ((2u * tau + (uint32_t)h) * p.n0 + cl) * ES : (tau * p.n0 + 16u * (uint32_t)h + cl) * ES;
			VKFFT_OPAQUE_ZERO(oq);
			const uint32_t step = (uint32_t)(SPLITB ? 2 * TPF : TPF) * p.n0 * ES + oq;
#pragma unroll
			for (int m = 0; m < E; m++) v[m] = gb_load_soa2<T, AUX_SC>(gsB, lane, m * step);
		};
		// ---- A: both halves through the 1024-point stages, [combine,] Four-Step twiddle, per-column contiguous write-through stores into the ring.
		// Vector-memory operations of a wave complete in the order of issue as far as s_waitcnt can tell (ONE counter for loads and stores on this
		// architecture): a load issued behind a batch of stores is available only once those stores are, and any reload from scratch drains the queue.
		// Hence the order of the requests below, and no register spilled inside the loop.
		const uint32_t cA = q + Q * s;
		const uint32_t bA = ((p.reverse ? p.C - 1u - cA : cA) << p.logG) + f;
		const bool live = hasA && bA < p.batch; // the last chunk may be partial: its empty tiles only keep the counters uniform
		VKFFT_PKPROF(3);
		gb_landed_raw<T, E>(rawE); // (counted wait: the other half stays in flight)
		VKFFT_PKPROF(1);
		if (hasA && !okA) { fused_wait(p.ctr + depA(s), TPC); VKFFT_PKPROF(5); }
		const uint32_t col0 = ti * (SPLITA ? 16 : 32);
		auto stagesA = [&](cxp<T>* v) {
			if constexpr (TWL) pk_lean_stages<T, SH, 0, TPF, TC, TwLds<T>, TWG>(v, plane + cl, TwLds<T>{twA}, tau);
			else pk_lean_stages<T, SH, 0, TPF, TC, TwGlobal<T>, TWG>(v, plane + cl, TwGlobal<T>{make_gbuf((const char*)p.lutA + oz)}, tau);
		};

		const char* const sbase = (const char*)p.scratch + ((uint64_t)(((q * p.NS + s % p.NS) << p.logG) + f) * nPts) * ES;
		const GBuf gs = make_gbuf(sbase + (uint64_t)col0 * LA * ES);
		if (live) { // (ONE region: register arrays that live across several conditional regions are copied at every merge)
			// the Four-Step twiddle's table look-ups travel during the stages (pk_fs_request); the second half of a 2048-point factor needs ONE more: w_N^(1024 j)
			PkFsTw<T, SH::LOGE> fsq;
			pk2<T> fshLo = pk2<T>{(T)1, (T)0}, fshHi = fshLo;
			cxp<T> vE[E], vO[E];
#pragma unroll
			for (int m = 0; m < E; m++) vE[m] = pk_from_aos<T>(rawE[m]);
			if (p.swapIn) { // inverse = conj . forward . conj
#pragma unroll
				for (int m = 0; m < E; m++) vE[m].im = -vE[m].im;
			}
			stagesA(vE);
			VKFFT_SYNC(); // the last exchange's reads are complete: the plane is free for the other half
			gb_landed_raw<T, E>(rawO); // everything this wave has issued is complete now (the previous ticket's stores to HBM came before this half's request)
			if (tid == 0) draw(it);    // next ticket + the state of its dependencies (read after S3): the wait for the atomic's result finds an empty queue
			pk_fs_request<T, SH::LOGE, TPF>(fsq, gtw, p.fsLoBits, tau, col0 + cl); // (behind the first half's stages: three halves + these are what the register file holds)
			if constexpr (SPLITA) {
				const uint32_t e = (uint32_t)L * (col0 + cl), loMask = (1u << p.fsLoBits) - 1u;
				const cx<T> a = gb_load<T>(gtw, (e & loMask) * ES, 0), b = gb_load<T>(gtw, (e >> p.fsLoBits) * ES, (loMask + 1u) * ES);
				fshLo = pk2<T>{a.x, a.y}; fshHi = pk2<T>{b.x, b.y};
			}
#pragma unroll
			for (int m = 0; m < E; m++) vO[m] = pk_from_aos<T>(rawO[m]);
			if (p.swapIn) {
#pragma unroll
				for (int m = 0; m < E; m++) vO[m].im = -vO[m].im;
			}
			stagesA(vO);
			if constexpr (SPLITA) { // X[k] = E[k] + w^k O[k], X[k + 1024] = E[k] - w^k O[k], k = tau + m TPF
#pragma unroll
				for (int m = 0; m < E; m++) {
					pk2<T> w;
					if constexpr (TWL) w = pk_tw<T>(TwLds<T>{twA}, tau, (uint32_t)(LUTC + m * TPF));
					else w = pk_tw<T>(TwGlobal<T>{make_gbuf((const char*)p.lutA + oz)}, tau, (uint32_t)(LUTC + m * TPF));
					const cxp<T> tw = pcmul1(vO[m], w), e = vE[m];
					vE[m] = pcadd(e, tw); vO[m] = pcsub(e, tw);
				}
			}
			VKFFT_PKPROF(8);
			auto twiddle = [&](cxp<T>* v, uint32_t tk, const pk2<T>* hiStep) {
				if constexpr (TWL) pk_fs_apply<T, SH::LOGE, TPF>(v, fsq, RowLds<T>{rowL}, tk, hiStep);
				else pk_fs_apply<T, SH::LOGE, TPF>(v, fsq, RowGlobal<T>{make_gbuf((const char*)p.rowTab + oz)}, tk, hiStep);
			};
			twiddle(vE, tau, nullptr);
			if constexpr (SPLITA) { // points k + 1024 of the same columns: every factor times w_N^(1024 j)
				const pk2<T> hs = pk_cmul_aos<T>(fshLo, fshHi);
				twiddle(vO, tau + (uint32_t)L, &hs);
			} else { // the other 16 columns: their own look-ups, requested now that the first half's are consumed (they travel during the first turn)
				pk_fs_request<T, SH::LOGE, TPF>(fsq, gtw, p.fsLoBits, tau, col0 + 16u + cl);
			}
			VKFFT_SYNC(); // the last exchange's reads are complete
			VKFFT_PKPROF(9);
			{
				cxp<T> rr[E];
				pk_lean_transpose<T, L, E, TPF, TC, NT>(vE, rr, plane, tid, cl, tau);
#pragma unroll
				for (int i = 0; i < E; i++) {
					const uint32_t idx = tid + i * NT;
					const uint32_t kp = idx % (L / 2), cc = idx / (L / 2);
					gb_store_soa2<T, AUX_ST>(gs, (cc * LA + 2u * kp) * ES, rr[i]);
				}
			}
			if (hasB && !okB) fused_wait(p.ctr + depB(s), TPC); // rare (the flag was sampled one ticket ago)
			if constexpr (!SPLITA) twiddle(vO, tau, nullptr);
			requestB(0, bE); // (behind the first half's ring stores: it travels during the second turn)
			VKFFT_SYNC(); // the plane is free again
			{
				cxp<T> rr[E];
				pk_lean_transpose<T, L, E, TPF, TC, NT>(vO, rr, plane, tid, cl, tau);
#pragma unroll
				for (int i = 0; i < E; i++) {
					const uint32_t idx = tid + i * NT;
					const uint32_t kp = idx % (L / 2), cc = idx / (L / 2);
					gb_store_soa2<T, AUX_ST>(gs, (SPLITA ? cc * LA + (uint32_t)L + 2u * kp : (cc + 16u) * LA + 2u * kp) * ES, rr[i]);
				}
			}
		} else {
			gb_landed_raw<T, E>(rawO);
			if (tid == 0) draw(it);
			if (hasB && !okB) fused_wait(p.ctr + depB(s), TPC);
			requestB(0, bE);
		}
		VKFFT_PKPROF(10); // (both turns + ring stores issued)
		VKFFT_VMEM_DRAIN(); // this wave: ring stores acknowledged by the memory side (and B.e in registers)
		VKFFT_PKPROF(6);
		VKFFT_SYNC();       // S3: ... in every wave; the next ticket is visible; the plane is free
		if (tid == 0) {
			if (hasA) (void)VKFFT_ATOMIC_ADD_U32(p.ctr + doneA + cA, 1u); // the chunk's tile is in the ring
		}
		VKFFT_PKPROF(11);
		requestB(1, bO);
		requestA(sTicket[it], 0, rawE); // the first half of the next ticket's A tile travels while the B tile computes
		if (liveB) {
			auto stagesB = [&](cxp<T>* v) {
				if constexpr (TWL) pk_lean_stages<T, SH, 0, TPF, TC, TwLds<T>, TWG>(v, plane + cl, TwLds<T>{twB}, tau);
				else pk_lean_stages<T, SH, 0, TPF, TC, TwGlobal<T>, TWG>(v, plane + cl, TwGlobal<T>{make_gbuf((const char*)p.lutB + oz)}, tau);
			};
			// natural-order store X[k0 + n0 k1]: k1 = tau + m TPF (second half of a 2048-point factor: + 1024; of a 1024-point one: the other 16 k0)
			auto storeHalf = [&](cxp<T>* v, int h) { // (scale, sign of the inverse and addresses formed at the store: computed ahead they sit in registers through the stages)
				const T sc = (T)p.scale, sci = p.swapOut ? -sc : sc;
				if (sc != (T)1 || p.swapOut) {
#pragma unroll
					for (int m = 0; m < E; m++) { v[m].re = v[m].re * pk_splat<T>(sc); v[m].im = v[m].im * pk_splat<T>(sci); }
				}
				const GBuf gout = make_gbuf((cx<T>*)p.out + ((int64_t)bB * p.outBatchStride + k00));
				const uint32_t lane = (h == 0 ? (tau * p.n0 + cl) : SPLITB ? ((tau + (uint32_t)L) * p.n0 + cl) : (tau * p.n0 + 16u + cl)) * ES;
				const uint32_t step = (uint32_t)TPF * p.n0 * ES + oz;
#pragma unroll
				for (int m = 0; m < E; m++) gb_store_aos2<T, AUX_HBM>(gout, lane + m * step, v[m]);
			};
			stagesB(bE);
			if constexpr (!SPLITB) storeHalf(bE, 0); // independent halves: the first one leaves while the second one computes
			VKFFT_SYNC();
			gb_landed_pk<T, E>(bO);
			stagesB(bO);
			if constexpr (SPLITB) {
#pragma unroll
				for (int m = 0; m < E; m++) {
					pk2<T> w;
					if constexpr (TWL) w = pk_tw<T>(TwLds<T>{twB}, tau, (uint32_t)(LUTC + m * TPF));
					else w = pk_tw<T>(TwGlobal<T>{make_gbuf((const char*)p.lutB + oz)}, tau, (uint32_t)(LUTC + m * TPF));
					const cxp<T> tw = pcmul1(bO[m], w), e = bE[m];
					bE[m] = pcadd(e, tw); bO[m] = pcsub(e, tw);
				}
				// (this exact order — both halves scaled, then both stored — is the one the register allocator fits without scratch)
				const T sc = (T)p.scale, sci = p.swapOut ? -sc : sc;
				if (sc != (T)1 || p.swapOut) {
#pragma unroll
					for (int m = 0; m < E; m++) { bE[m].re = bE[m].re * pk_splat<T>(sc); bE[m].im = bE[m].im * pk_splat<T>(sci); bO[m].re = bO[m].re * pk_splat<T>(sc); bO[m].im = bO[m].im * pk_splat<T>(sci); }
				}
				const GBuf gout = make_gbuf((cx<T>*)p.out + ((int64_t)bB * p.outBatchStride + k00));
				const uint32_t laneE = (tau * p.n0 + cl) * ES, laneO = ((tau + (uint32_t)L) * p.n0 + cl) * ES;
				const uint32_t step = (uint32_t)TPF * p.n0 * ES + oz;
#pragma unroll
				for (int m = 0; m < E; m++) gb_store_aos2<T, AUX_HBM>(gout, laneE + m * step, bE[m]);
#pragma unroll
				for (int m = 0; m < E; m++) gb_store_aos2<T, AUX_HBM>(gout, laneO + m * step, bO[m]);
			} else storeHalf(bO, 1);
		} else if (hasB) {
			gb_landed_pk<T, E>(bO); // (a tile without transform still has to have READ its ring slot before the slot is released)
		}
		VKFFT_PKPROF(4);
#if !defined(VKFFT_HOSTEMU)
		if constexpr ((MODE & 4) != 0) { if (tid == 0) spc[7]++; }
#endif
		if (hasB) pendB = doneB + cB; // the ring slot is released once BOTH halves have been read by every wave: B.o landed above, the barrier is the next S1
		requestA(sTicket[it], 1, rawO); // (behind the stores to HBM: it is waited for after the first half's stages of the next ticket)
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

template <typename T, typename SH, int SPLITA, int SPLITB, int MODE, int TWL> void pow2_fused_pkh_launch(const FusedParams& prm, dim3 grid, hipStream_t s) {
	hipLaunchKernelGGL((pow2_fused_pkh_kernel<T, SH, SPLITA, SPLITB, MODE, TWL>), grid, dim3(((1 << SH::LOGN) >> SH::LOGE) * 8), 0, s, prm);
}

} // namespace vkfft_mi355x
