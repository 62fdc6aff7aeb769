// Fused Four-Step: both passes of a two-factor power-of-two transform in ONE persistent launch, the intermediate kept
// on-die (256 MiB Infinity Cache) instead of making a round trip through HBM.
//
// The reference runs a long sequence as 2-3 "axis uploads", each a full kernel with the data set written to and read
// back from device memory in between (vkFFT_RunApp.h:114, vkFFT_4step.h:31, vkFFT_ReadWrite.h:1405-1476).  Measured on
// MI355X (tools/probe3.hip): HBM streams 6.0 TB/s read / 5.3 TB/s write, a <= 64 MiB working set that is rewritten all the
// time is served by the Infinity Cache at 27 TB/s read / 7-12 TB/s write.  So the batch is cut into chunks of about a MiB and
// the two passes of a chunk become tiles of one work queue, the intermediate living in a small ring of chunk-sized slots:
//
//   queue of slots s = 0, 1, ...;  ticket (s, r) = tile r of pass A of chunk s  +  tile r of pass B of chunk s - D
//
//   * a workgroup draws tickets with one atomic add (dynamic, so no tail imbalance and no co-residency requirement: every
//     dependency points at a SMALLER ticket, which some running workgroup already holds -> deadlock-free for any grid size
//     and dispatch order).  One counter hands out ~90 tickets/us, not enough for 64 KiB tiles: there is one queue per XCD
//     (chunks dealt round-robin), a workgroup serves the queue of the XCD it runs on (HW_REG_XCC_ID; speed only) and helps
//     the other queues when its own is drained, so completion does not depend on the placement;
//   * an A tile (columns of the n0 x n1 view: FFT over n0, Four-Step twiddle) writes its columns as contiguous runs into the
//     ring with write-through (sc1) 16-byte stores; when those are acknowledged doneA[chunk] is bumped;
//   * a B tile (FFT over n1, natural-order store) needs doneA[chunk] == tiles per chunk, reads the ring with sc1 loads (served
//     from the memory side, never from a stale per-XCD L2 line) and bumps doneB[chunk]; the A tile that reuses the ring slot
//     NS chunks later needs doneB == tiles per chunk;
//   * none of the latencies is on the critical path: the next ticket and the state of ITS dependencies are fetched while the
//     current tiles compute (a dependency found unsatisfied is polled at the tile, rare with the lag D and ring NS the planner
//     chooses from the number of tickets in flight), and an A tile's completion is published at the next wait for loads;
//   * the ring (tens of MiB) is rewritten every few microseconds and stays resident in the Infinity Cache, so HBM sees one
//     read and one write of the data set: the algorithmic minimum;
//   * the last workgroup to leave zeroes the counters for the next launch.
#pragma once
#include "kernel_pow2.h"

namespace vkfft_mi355x {

#if defined(VKFFT_HOSTEMU)
// the emulator runs one workgroup at a time: workgroup 0 drains the whole queue in ticket order, every wait is already satisfied
#define VKFFT_ATOMIC_ADD_U32(p, v) hostemu_fetch_add((p), (v))
#define VKFFT_ATOMIC_LOAD_U32(p) (*(volatile uint32_t*)(p))
#define VKFFT_SLEEP() do { } while (0)
#define VKFFT_VMEM_DRAIN() do { } while (0)
inline uint32_t hostemu_fetch_add(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
#else
#define VKFFT_ATOMIC_ADD_U32(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define VKFFT_ATOMIC_LOAD_U32(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define VKFFT_SLEEP() __builtin_amdgcn_s_sleep(2)
// every store of this wave has been acknowledged by the memory side (inline asm: the compiler cannot drop it)
#define VKFFT_VMEM_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

template <typename T, int LOGE, int TPF>
__device__ inline void pow2_fs_twiddle(cx<T>* v, const GBuf gtab, const uint32_t fsLoBits, const uint32_t tau, const uint32_t colIdx) {
	constexpr int E = 1 << LOGE;
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	const uint32_t loMask = (1u << fsLoBits) - 1u;
	const uint32_t hiBase = (loMask + 1u) * ES;
	auto tw = [&](uint32_t e) { return cmul(gb_load<T>(gtab, (e & loMask) * ES, 0), gb_load<T>(gtab, (e >> fsLoBits) * ES, hiBase)); };
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

template <typename T, typename SCH, int TPF, int TCP, int TWL, int CPT>
__device__ inline void fused_stages(cx<T>* v, cx<T>* ldsf, const cx<T>* twLds, const void* twGlobal, uint32_t oz, uint32_t tau) {
	if constexpr (TWL) pow2_stages<T, SCH, 0, TPF, TCP, TwLds<T>, CPT>(v, ldsf, TwLds<T>{twLds}, tau, false);
	else pow2_stages<T, SCH, 0, TPF, TCP, TwGlobal<T>, CPT>(v, ldsf, TwGlobal<T>{make_gbuf((const char*)twGlobal + oz)}, tau, false);
}

__device__ inline void fused_wait(uint32_t* ctr, uint32_t target) {
	if (threadIdx.x == 0) {
		while (VKFFT_ATOMIC_LOAD_U32(ctr) < target) VKFFT_SLEEP();
	}
	VKFFT_SYNC();
}

// workgroups of a fused kernel that fit one CU (LDS and wave slots), at most 4: fixes the register budget through __launch_bounds__
// LEAN: the register-lean stages of kernel_pow2_lean.h (real and imaginary parts through one real-valued plane: half the LDS per tile, 128 VGPRs)
template <typename T, typename SA, int TCA, typename SB, int TCB, int TWL, int CPT, int LEAN = 0> constexpr int pow2_fused_wg_per_cu() {
	constexpr int la = (1 << SA::LOGN) * (TCA + (CPT == 2 ? 2 : 1)), lb = (1 << SB::LOGN) * (TCB + (CPT == 2 ? 2 : 1));
	constexpr int pa = (int)pow2_lean_plane_elems<SA, TCA>(), pb = (int)pow2_lean_plane_elems<SB, TCB>();
	constexpr int ldsBytes = LEAN ? (pa > pb ? pa : pb) * (int)sizeof(T) + (TWL ? SA::lutTotal() + SB::lutTotal() : 0) * (int)sizeof(cx<T>) + 64
	                              : ((la > lb ? la : lb) + (TWL ? SA::lutTotal() + SB::lutTotal() : 0)) * (int)sizeof(cx<T>) + 64;
	constexpr int nt = ((1 << SA::LOGN) >> SA::LOGE) * TCA / CPT;
	int w = 163840 / ldsBytes;
	if (w > 2048 / nt) w = 2048 / nt;
	return w > (LEAN ? 8 : 4) ? (LEAN ? 8 : 4) : w < 1 ? 1 : w;
}

// Publishing an A tile's completion (thread 0, at a point where every wave's vector-memory operations have drained: the ring stores
// are write-through, so "acknowledged" means "at the memory side").  Write-back stores + an L2 write-back before the signal were tried
// and are wrong: the writer's L2 keeps the lines, and a later tenant of the ring slot written by another XCD is then read stale.
__device__ inline void fused_publish(uint32_t* ctr, uint32_t& pending) {
	constexpr uint32_t kNone = 0xffffffffu;
	if (pending != kNone) { (void)VKFFT_ATOMIC_ADD_U32(ctr + pending, 1u); pending = kNone; }
}

__device__ inline uint32_t fused_xcc_id() {
#if defined(VKFFT_HOSTEMU)
	return 0;
#else
	return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u; // HW_REG_XCC_ID[3:0]
#endif
}

// MODE bit 1: non-temporal hint on the HBM side; bit 2 (development build only): per-phase cycle sums (tools/prof_fused.py)
// TWL: stage twiddles staged in LDS (1) or read through the buffer path from L2 (0: where the LDS copy would cost a workgroup per CU)
// CPT: columns per thread.  2 (fp32 only): a thread keeps two adjacent columns, so that every global and ring access is 16 bytes per lane and
// every LDS exchange access 16 bytes (the fp64 kernels, whose elements are 16 bytes, measured 10-15 % above the 8-byte fp32 ones)
template <typename T, typename SA, int TCA, typename SB, int TCB, int MODE, int TWL, int CPT, int LEAN = 0>
__global__ void __launch_bounds__(((1 << SA::LOGN) >> SA::LOGE) * TCA / CPT, (pow2_fused_wg_per_cu<T, SA, TCA, SB, TCB, TWL, CPT, LEAN>() * (((1 << SA::LOGN) >> SA::LOGE) * TCA / CPT) + 255) / 256)
pow2_fused_kernel(const FusedParams p) {
	static_assert(CPT == 1 || (CPT == 2 && sizeof(T) == 4), "two columns per thread: fp32 only");
	static_assert(!LEAN || CPT == 2, "register-lean form: two columns per thread");
	constexpr int LA = 1 << SA::LOGN, EA = 1 << SA::LOGE, TPFA = LA / EA, TCPA = TCA + (CPT == 2 ? 2 : 1);
	constexpr int LB = 1 << SB::LOGN, EB = 1 << SB::LOGE, TPFB = LB / EB, TCPB = TCB + (CPT == 2 ? 2 : 1);
	constexpr int NT = TPFA * TCA / CPT;
	static_assert(NT == TPFB * TCB / CPT, "both phases run on the same workgroup shape");
	static_assert(LA * TCA == LB * TCB, "both phases move the same number of points per tile");
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	constexpr int AUX_SC = 16;                   // ring loads: agent scope, served from the memory side
	constexpr int AUX_ST = 16;                   // ring stores: write-through (no XCD's L2 ever holds a ring line)
	constexpr int AUX_HBM = (MODE & 2) ? 2 : 0;
	constexpr int AUX_HBM_LD = AUX_HBM, AUX_HBM_ST = AUX_HBM;
	constexpr int PLA = (int)pow2_lean_plane_elems<SA, TCA>(), PLB = (int)pow2_lean_plane_elems<SB, TCB>(), PLN = (PLA > PLB ? PLA : PLB) + ((PLA > PLB ? PLA : PLB) & 1);
	constexpr int LDSN = LEAN ? PLN / 2 : (LA * TCPA > LB * TCPB ? LA * TCPA : LB * TCPB); // (in complex elements; the lean plane holds PLN real ones)
	constexpr int LUTA = TWL ? SA::lutTotal() : 0, LUTB = TWL ? SB::lutTotal() : 0;
	constexpr int TWG = 8, PFN = TWL ? 0 : 8; // lean stages: stage twiddles in flight at a time; twiddles read through L2 are requested ahead of the exchange
	__shared__ cx<T> lds[LDSN + LUTA + LUTB];
	T* const plane = (T*)lds;
	__shared__ uint32_t sTicket[2], sOkA[2], sOkB[2];
	const uint32_t tid = threadIdx.x;
	// stage twiddles of both factors staged in LDS for the lifetime of the workgroup
	cx<T>* const twA = lds + LDSN;
	cx<T>* const twB = twA + LUTA;
	for (uint32_t i = tid; i < (uint32_t)LUTA; i += NT) twA[i] = ((const cx<T>*)p.lutA)[i];
	for (uint32_t i = tid; i < (uint32_t)LUTB; i += NT) twB[i] = ((const cx<T>*)p.lutB)[i];
	const uint32_t logTPC = p.logG + p.logTiles, TPC = 1u << logTPC;
	const uint32_t doneA = kFusedCtrDone, doneB = kFusedCtrDone + p.C; // counter indices
	const uint64_t nPts = (uint64_t)p.n0 * p.n1;
	constexpr uint32_t kNone = 0xffffffffu;
	const uint32_t Q = p.Q;
	uint32_t q = Q > 1 ? fused_xcc_id() % Q : 0u, tried = 0;
	uint32_t Cq = (p.C + Q - 1u - q) / Q;               // chunks q, q + Q, q + 2Q, ... of this queue
	uint32_t totq = Cq ? (Cq + p.D) << logTPC : 0u;      // tickets of this queue
	// counters a ticket of slot s depends on: the ring slot's previous tenant read completely (A), the chunk written completely (B)
	auto depA = [&](uint32_t s) -> uint32_t { return (s < Cq && s >= p.NS) ? doneB + q + Q * (s - p.NS) : kNone; };
	auto depB = [&](uint32_t s) -> uint32_t { return (s >= p.D && s - p.D < Cq) ? doneA + q + Q * (s - p.D) : kNone; };
	uint32_t pending = kNone;  // counter this workgroup still owes a bump: its last A tile's stores are in flight (thread 0 only)
	if (tid == 0) {
		const uint32_t t0 = VKFFT_ATOMIC_ADD_U32(p.ctr + kFusedCtrTicket + 32u * q, 1u), s0 = t0 >> logTPC;
		const uint32_t dA = depA(s0), dB = depB(s0);
		sTicket[0] = t0;
		sOkA[0] = dA == kNone ? 1u : (uint32_t)(VKFFT_ATOMIC_LOAD_U32(p.ctr + dA) >= TPC);
		sOkB[0] = dB == kNone ? 1u : (uint32_t)(VKFFT_ATOMIC_LOAD_U32(p.ctr + dB) >= TPC);
	}
	uint32_t it = 0; // iteration parity: the next ticket is written while slower waves may still read the current one
#if !defined(VKFFT_HOSTEMU)
	unsigned long long pc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ptk = 0; // (MODE bit 2) cycle sums: S1 wait, A loads, A compute, B wait+loads, B compute; slow waits
#define VKFFT_PROF(i) do { if constexpr ((MODE & 4) != 0) { const unsigned long long now = __builtin_readcyclecounter(); pc[i] += now - ptk; ptk = now; } } while (0)
	if constexpr ((MODE & 4) != 0) ptk = __builtin_readcyclecounter();
#else
#define VKFFT_PROF(i) do { } while (0)
#endif
	for (;;) {
		VKFFT_SYNC(); // S1: ticket visible; exchange buffer free again
		VKFFT_PROF(0);
		const uint32_t t = sTicket[it];
		if (t >= totq) {
			// this queue is drained: help the next one, leave when every queue is (completion must not depend on where workgroups run)
			if (++tried >= Q) break;
			VKFFT_VMEM_DRAIN();
			VKFFT_SYNC();
			q = q + 1u == Q ? 0u : q + 1u;
			Cq = (p.C + Q - 1u - q) / Q;
			totq = Cq ? (Cq + p.D) << logTPC : 0u;
			if (tid == 0) {
				fused_publish(p.ctr, pending);
				const uint32_t t0 = VKFFT_ATOMIC_ADD_U32(p.ctr + kFusedCtrTicket + 32u * q, 1u), s0 = t0 >> logTPC;
				const uint32_t dA = depA(s0), dB = depB(s0);
				sTicket[it] = t0;
				sOkA[it] = dA == kNone ? 1u : (uint32_t)(VKFFT_ATOMIC_LOAD_U32(p.ctr + dA) >= TPC);
				sOkB[it] = dB == kNone ? 1u : (uint32_t)(VKFFT_ATOMIC_LOAD_U32(p.ctr + dB) >= TPC);
			}
			continue;
		}
		const uint32_t okA = sOkA[it], okB = sOkB[it];
		it ^= 1u;
		// the tables are the same for every tile: an opaque zero in their base keeps the loads inside the loop (hoisted, they would
		// pin dozens of VGPRs for the lifetime of the persistent workgroup)
		VKFFT_OPAQUE_ZERO(oz);
		const GBuf gtw = make_gbuf((const char*)p.tw4 + oz);
		const uint32_t s = t >> logTPC, r = t & (TPC - 1u);
		const uint32_t f = r >> p.logTiles, ti = r & ((1u << p.logTiles) - 1u);
		const bool hasA = s < Cq, hasB = s >= p.D && s - p.D < Cq;
		uint32_t nextT = 0, nfA = TPC, nfB = TPC;
		if (tid == 0) nextT = VKFFT_ATOMIC_ADD_U32(p.ctr + kFusedCtrTicket + 32u * q, 1u);
		// ring addresses of the B part (its loads are issued in the middle of the A part, see below)
		const uint32_t sB = s - p.D, cB = q + Q * sB;
		const uint32_t bB = ((p.reverse ? p.C - 1u - cB : cB) << p.logG) + f;
		const bool liveB = hasB && bB < p.batch;
		const uint32_t cBl = (tid % (TCB / CPT)) * CPT, tauB = tid / (TCB / CPT); // first of this thread's CPT adjacent columns
		const uint32_t k00 = ti * TCB;
		const char* const sbaseB = (const char*)p.scratch + ((uint64_t)(((q * p.NS + (hasB ? sB % p.NS : 0u)) << p.logG) + f) * nPts) * ES;
		const GBuf gsB = make_gbuf(sbaseB + (uint64_t)k00 * ES);
		const uint32_t laneB = liveB ? (tauB * p.n0 + cBl) * ES : kGbInvalid, stepB = (uint32_t)TPFB * p.n0 * ES;
		cx<T> vB[CPT * EB];
		{
			// ---- A: FFT over n0 of TCA neighbouring columns (stride n1), twiddle, per-column contiguous store into the ring
			const uint32_t cA = q + Q * s;                          // chunk in processing order (counters, ring slot)
			const uint32_t b = ((p.reverse ? p.C - 1u - cA : cA) << p.logG) + f;
			const bool live = hasA && b < p.batch; // the last chunk may be partial: its empty tiles only keep the counters uniform
			const char* const sbase = (const char*)p.scratch + ((uint64_t)(((q * p.NS + s % p.NS) << p.logG) + f) * nPts) * ES;
			const uint32_t c = (tid % (TCA / CPT)) * CPT, tau = tid / (TCA / CPT); // first of this thread's CPT adjacent columns
			const uint32_t col0 = ti * TCA;
			const GBuf gin = make_gbuf((const cx<T>*)p.in + ((int64_t)(live ? b : 0u) * p.inBatchStride + col0));
			const uint32_t laneIn = live ? (tau * p.n1 + c) * ES : kGbInvalid, stepIn = (uint32_t)TPFA * p.n1 * ES;
			{
				cx<T> v[CPT * EA];
#pragma unroll
				for (int m = 0; m < EA; m++) {
					if constexpr (CPT == 1) v[m] = gb_load_x<T, AUX_HBM_LD>(gin, laneIn, m * stepIn);
					else gb_load2_x<T, AUX_HBM_LD>(gin, laneIn, m * stepIn, v[m], v[EA + m]);
				}
				VKFFT_VMEM_DRAIN(); // this tile's loads have landed, the previous ticket's stores are acknowledged, the next ticket is here
				if (tid == 0) {
					sTicket[it] = nextT;
					const uint32_t sN = nextT >> logTPC, dA = nextT < totq ? depA(sN) : kNone, dB = nextT < totq ? depB(sN) : kNone;
					if (dA != kNone) nfA = VKFFT_ATOMIC_LOAD_U32(p.ctr + dA); // consumed at the end of this iteration
					if (dB != kNone) nfB = VKFFT_ATOMIC_LOAD_U32(p.ctr + dB);
				}
				VKFFT_SYNC(); // S2
				VKFFT_PROF(1);
				if (tid == 0 && pending != kNone) { (void)VKFFT_ATOMIC_ADD_U32(p.ctr + pending, 1u); pending = kNone; }
				if (hasA && !okA) { fused_wait(p.ctr + depA(s), TPC); VKFFT_PROF(5); }
				if (live) {
					if (p.swapIn) {
#pragma unroll
						for (int m = 0; m < CPT * EA; m++) v[m] = cswap(v[m]);
					}
					if constexpr (LEAN) {
						if constexpr (TWL) pow2_lean_stages<T, SA, 0, TPFA, TCA, TwLds<T>, TWG, CPT>(v, plane + c, TwLds<T>{twA}, tau);
						else pow2_lean_stages<T, SA, 0, TPFA, TCA, TwGlobal<T>, TWG, CPT, PFN>(v, plane + c, TwGlobal<T>{make_gbuf((const char*)p.lutA + oz)}, tau);
					} else fused_stages<T, SA, TPFA, TCPA, TWL, CPT>(v, lds + c, twA, p.lutA, oz, tau);
					VKFFT_PROF(8);
#pragma unroll
					for (int cc = 0; cc < CPT; cc++) pow2_fs_twiddle<T, SA::LOGE, TPFA>(v + cc * EA, gtw, p.fsLoBits, tau, col0 + c + cc);
					VKFFT_PROF(9);
					if constexpr (SA::NS > 1) VKFFT_SYNC(); // the last exchange's reads are complete
					if constexpr (LEAN) {
						// the tile turned through the plane (real parts, then imaginary parts) into per-column contiguous order and stored from registers
						cx<T> r[CPT * EA];
						pow2_lean_transpose<T, LA, EA, TPFA, TCA, NT>(v, r, plane, tid, c, tau);
						VKFFT_PROF(10);
						const GBuf gs = make_gbuf(sbase + (uint64_t)col0 * LA * ES);
#pragma unroll
						for (int i = 0; i < CPT * EA / 2; i++) {
							const uint32_t idx = tid + i * NT;
							const uint32_t kp = idx % (LA / 2), cc = idx / (LA / 2);
							gb_store2_x<T, AUX_ST>(gs, (cc * LA + 2u * kp) * ES, r[2 * i], r[2 * i + 1]);
						}
					} else {
#pragma unroll
					for (int m = 0; m < EA; m++) {
						if constexpr (CPT == 1) lds[(tau + m * TPFA) * TCPA + c] = v[m];
						else *(cx2<T>*)(lds + (tau + m * TPFA) * TCPA + c) = cx2<T>{v[m], v[EA + m]};
					}
					VKFFT_SYNC();
					VKFFT_PROF(10);
					}
				}
			}
			if (live && !LEAN) {
				const GBuf gs = make_gbuf(sbase + (uint64_t)col0 * LA * ES);
				if constexpr (sizeof(T) == 4) {
					// two consecutive k per lane: 16-byte write-through stores (8-byte sc1 stores cost 2.7x per byte)
#pragma unroll
					for (int i = 0; i < CPT * EA / 2; i++) {
						const uint32_t idx = tid + i * NT;
						const uint32_t kp = idx % (LA / 2), cc = idx / (LA / 2);
						gb_store2_x<T, AUX_ST>(gs, (cc * LA + 2u * kp) * ES, lds[(2u * kp) * TCPA + cc], lds[(2u * kp + 1u) * TCPA + cc]);
					}
				} else {
#pragma unroll
					for (int i = 0; i < CPT * EA; i++) {
						const uint32_t idx = tid + i * NT;
						const uint32_t k = idx % LA, cc = idx / LA;
						gb_store_x<T, AUX_ST>(gs, (cc * LA + k) * ES, 0, lds[k * TCPA + cc]);
					}
				}
			}
			if (hasA) pending = doneA + cA;
		}
		VKFFT_PROF(2);
		if (hasB) {
			// ---- B: FFT over n1 (stride n0 in the ring) of TCB neighbouring k0, natural-order store X[k0 + n0*k1]
			const GBuf gout = make_gbuf((cx<T>*)p.out + ((int64_t)(liveB ? bB : 0u) * p.outBatchStride + k00));
			if (!okB) { fused_wait(p.ctr + depB(s), TPC); VKFFT_PROF(6); } // rare (okB was sampled one ticket ago: ordered before the loads by S1)
#pragma unroll
			for (int m = 0; m < EB; m++) {
				if constexpr (CPT == 1) vB[m] = gb_load_x<T, AUX_SC>(gsB, laneB, m * stepB);
				else gb_load2_x<T, AUX_SC>(gsB, laneB, m * stepB, vB[m], vB[EB + m]);
			}
			VKFFT_VMEM_DRAIN(); // the tile is in registers; the A part's ring stores are acknowledged
			VKFFT_SYNC();    // S3: ... in every wave
			VKFFT_PROF(3);
			if (tid == 0) {
				(void)VKFFT_ATOMIC_ADD_U32(p.ctr + doneB + cB, 1u); // release the ring slot
				fused_publish(p.ctr, pending);
			}
			if (liveB) {
				if constexpr (LEAN) {
					if constexpr (TWL) pow2_lean_stages<T, SB, 0, TPFB, TCB, TwLds<T>, TWG, CPT>(vB, plane + cBl, TwLds<T>{twB}, tauB);
					else pow2_lean_stages<T, SB, 0, TPFB, TCB, TwGlobal<T>, TWG, CPT, PFN>(vB, plane + cBl, TwGlobal<T>{make_gbuf((const char*)p.lutB + oz)}, tauB);
				} else fused_stages<T, SB, TPFB, TCPB, TWL, CPT>(vB, lds + cBl, twB, p.lutB, oz, tauB);
				if (p.swapOut) {
#pragma unroll
					for (int m = 0; m < CPT * EB; m++) vB[m] = cswap(vB[m]);
				}
				const T sc = (T)p.scale;
				if (sc != (T)1) {
#pragma unroll
					for (int m = 0; m < CPT * EB; m++) vB[m] = cscale(vB[m], sc);
				}
#pragma unroll
				for (int m = 0; m < EB; m++) {
					if constexpr (CPT == 1) gb_store_x<T, AUX_HBM_ST>(gout, laneB, m * stepB, vB[m]);
					else gb_store2_x<T, AUX_HBM_ST>(gout, laneB + m * stepB, vB[m], vB[EB + m]);
				}
			}
		}
		if (tid == 0) { sOkA[it] = nfA >= TPC; sOkB[it] = nfB >= TPC; }
		VKFFT_PROF(4);
#if !defined(VKFFT_HOSTEMU)
		if constexpr ((MODE & 4) != 0) pc[7]++;
#endif
	}
#if !defined(VKFFT_HOSTEMU)
	if constexpr ((MODE & 4) != 0) { if (tid == 0 && p.prof) { for (int i = 0; i < 12; i++) p.prof[(size_t)blockIdx.x * 12 + i] = pc[i]; } }
#endif
	// ---- exit: publish the last A tile, then the last workgroup out resets the counters for the next launch
	VKFFT_VMEM_DRAIN();
	VKFFT_SYNC();
	if (tid == 0) {
		fused_publish(p.ctr, pending);
		VKFFT_VMEM_DRAIN(); // this workgroup's counter updates have been performed
		sOkA[0] = VKFFT_ATOMIC_ADD_U32(p.ctr + kFusedCtrExit, 1u) == gridDim.x - 1u;
	}
	VKFFT_SYNC();
	if (sOkA[0]) {
		for (uint32_t i = tid; i < kFusedCtrDone + 2u * p.C; i += NT) p.ctr[i] = 0u;
	}
}

struct Pow2FusedVariant {
	int log2n; bool dp; int mode; int la, lb; int bitsA[4], bitsB[4]; int tca, tcb, threads, wgPerCu; // la, lb: log2 of the two factors; wgPerCu: workgroups per CU the kernel is launched with (resources, or a measured cap below them)
	void (*launch)(const FusedParams&, dim3, hipStream_t);
	const void* fn;
	const char* name = nullptr; // the __global__ function behind the entry when it is not pow2_fused_kernel (vkfftMI355XDescribePlan, bench labels)
};

template <typename T, typename SA, int TCA, typename SB, int TCB, int MODE, int TWL, int CPT, int LEAN = 0> void pow2_fused_launch(const FusedParams& prm, dim3 grid, hipStream_t s) {
	constexpr int threads = ((1 << SA::LOGN) >> SA::LOGE) * TCA / CPT;
	hipLaunchKernelGGL((pow2_fused_kernel<T, SA, TCA, SB, TCB, MODE, TWL, CPT, LEAN>), grid, dim3(threads), 0, s, prm);
}

} // namespace vkfft_mi355x
