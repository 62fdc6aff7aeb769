// Register-lean power-of-two Stockham stages: 32 complex points per thread in at most 128 VGPRs, real and imaginary parts exchanged
// through ONE real-valued LDS plane, one after the other.
//
// Why: the stages of kernel_pow2_core.h exchange interleaved complex values (8 bytes per point in LDS) and leave instruction order to the
// compiler, which — given 256 registers — keeps all 31 twiddles of a radix-32 butterfly and the temporaries of its recursion in flight
// (168 VGPRs for the 2^14 row).  That caps a CU at ONE 135 KiB / 512-thread workgroup: load, compute and store of a tile never overlap.
// Here
//   * a point's real and imaginary part travel separately through a plane of N (+ padding) floats: half the LDS, so two tiles fit a CU;
//   * the butterflies are decimation-in-frequency, in place, one radix-2 layer at a time, the scheduler fenced between layers: the live set
//     is the thread's own 2E floats plus a handful of temporaries (reference shape this replaces: the register-boost split of
//     vkFFT_RegisterBoost.h:31 and the per-thread register count of vkFFT_AxisBlockSplitter.h:266-366);
//   * stage twiddles arrive TWG at a time (16 VGPRs for 8) and each product is pinned where it is written (VKFFT_PIN: otherwise the
//     optimiser sinks the multiplies into the butterfly and all loads are in flight at once again);
//   * every LDS address is one base register + an immediate: slot(a0 + c) = slot(a0) + c + (c >> P) holds for the padded slot
//     a + (a >> P), P = log2 of the first radix, because the low P bits never carry in the index patterns of a Stockham stage.
// Measured resource use (hipcc -O3, gfx950): 2^13, 2^14, 2^15 rows of 32 points per thread: 126-128 VGPRs, no scratch, 4 waves per SIMD.
#pragma once
#include "kernel_pow2_core.h"

namespace vkfft_mi355x {

// two neighbouring real values (the same point of two adjacent columns, or two consecutive points of a column) as one 8-byte LDS access
template <typename T> struct alignas(2 * sizeof(T)) RealPair { T x, y; };

__host__ __device__ constexpr int pow2_bitrev(int k, int bits) { int r = 0; for (int i = 0; i < bits; i++) r |= ((k >> i) & 1) << (bits - 1 - i); return r; }

// radix-R decimation-in-frequency butterfly, in place, layer by layer; X[k] ends up at x[bitrev(k)]
template <int R, typename T> __device__ inline void pow2_dif_inplace(cx<T>* x) {
#pragma unroll
	for (int h = R / 2; h >= 1; h >>= 1) {
#pragma unroll
		for (int blk = 0; blk < R; blk += 2 * h) {
#pragma unroll
			for (int j = 0; j < h; j++) {
				const cx<T> a = x[blk + j], b = x[blk + j + h];
				x[blk + j] = cadd(a, b);
				const cx<T> d = csub(a, b);
				const int kk = j * (16 / h); // w_{2h}^j as a power of w_32
				if (kk == 0) x[blk + j + h] = d;
				else if (kk == 8) x[blk + j + h] = cmul_mi(d);
				else x[blk + j + h] = cmul(d, cx<T>{(T)pow2_cos32(kk), (T)(-pow2_sin32(kk))});
			}
		}
		VKFFT_SCHED_FENCE();
	}
}

// The butterflies of stage SI on the thread's E points (register m <-> point tau + m*TPF, as in pow2_stages), results in natural order.
// CPT = 2 (column kernels): two adjacent columns in v[0..E) and v[E..2E), one twiddle read serves both.
// PF > 0: the first PF twiddles of butterfly 0 were requested BEFORE the exchange that feeds this stage (pow2_lean_prefetch: their latency is
// hidden behind the exchange's barriers) and arrive in pf[].
template <int R, int TWG, int PF> __host__ __device__ constexpr int pow2_lean_chunk_end(int i0) { return (i0 == 1 && PF > 0) ? 1 + PF : (i0 + TWG < R ? i0 + TWG : R); }

template <typename T, typename SCH, int SI, int TPF, typename TW, int PF>
__device__ inline void pow2_lean_prefetch(cx<T>* pf, const TW lut, const uint32_t tau) {
	constexpr int LOGS = SCH::logS(SI), S = 1 << LOGS, LO = SCH::lutOff(SI);
	const uint32_t s = tau & (S - 1);
#pragma unroll
	for (int i = 1; i <= PF; i++) pf[i - 1] = lut.get(s, (uint32_t)(LO + (i - 1) * S));
}

template <typename T, typename SCH, int SI, int TPF, typename TW, int TWG, int CPT, int PF>
__device__ inline void pow2_lean_butterflies(cx<T>* v, const TW lut, const uint32_t tau, const cx<T>* pf) {
	constexpr int LOGE = SCH::LOGE, E = 1 << LOGE;
	constexpr int LOGR = SCH::bits[SI], R = 1 << LOGR, NB = E / R;
	constexpr int LOGS = SCH::logS(SI), S = 1 << LOGS;
	constexpr int TW_MAX = TWG > PF ? TWG : (PF > 0 ? PF : 1);
#pragma unroll
	for (int b = 0; b < NB; b++) {
		if constexpr (SI > 0) {
			const uint32_t s = (tau + b * TPF) & (S - 1);
			constexpr int LO = SCH::lutOff(SI);
#pragma unroll
			for (int i0 = 1; i0 < R;) {
				const int i1 = (b == 0) ? pow2_lean_chunk_end<R, TWG, PF>(i0) : (i0 + TWG < R ? i0 + TWG : R);
				cx<T> w[TW_MAX];
#pragma unroll
				for (int i = i0; i < i1; i++) {
					if (b == 0 && PF > 0 && i0 == 1) w[i - i0] = pf[i - 1];
					else w[i - i0] = lut.get(s, (uint32_t)(LO + (i - 1) * S));
				}
				VKFFT_SCHED_FENCE();
#pragma unroll
				for (int i = i0; i < i1; i++) {
#pragma unroll
					for (int cc = 0; cc < CPT; cc++) { cx<T>& q = v[cc * E + b + i * NB]; q = cmul(q, w[i - i0]); VKFFT_PIN(q.x); VKFFT_PIN(q.y); }
				}
				VKFFT_SCHED_FENCE();
				i0 = i1;
			}
		}
#pragma unroll
		for (int cc = 0; cc < CPT; cc++) {
			cx<T> x[R];
#pragma unroll
			for (int i = 0; i < R; i++) x[i] = v[cc * E + b + i * NB];
			pow2_dif_inplace<R, T>(x);
#pragma unroll
			for (int k = 0; k < R; k++) v[cc * E + b + k * NB] = x[pow2_bitrev(k, LOGR)];
		}
	}
}

// LDS slot of element a of a tile's plane: rows  a + (a >> P)  floats; column tiles  (a + (a >> P)) * TC + column  (TC floats per element row,
// no column padding: with 16 columns a half-wave of 8-byte accesses covers two consecutive elements, whose padded slots differ in parity — all
// 32 banks, both for the Stockham scatter and for the gather)
template <int TC, int P> __device__ inline uint32_t pow2_lean_slot(uint32_t a) { return (a + (a >> P)) * (TC ? TC : 1); }
template <int TC, int P> __host__ __device__ constexpr uint32_t pow2_lean_step(uint32_t c) { return (c + (c >> P)) * (TC ? TC : 1); }
template <typename SCH, int TC> __host__ __device__ constexpr uint32_t pow2_lean_plane_elems() { return ((1u << SCH::LOGN) + ((1u << SCH::LOGN) >> SCH::bits[0])) * (TC ? TC : 1); }

// one real-valued plane exchange of stage SI's results (part 0: real parts, 1: imaginary parts); `plane` points at the thread's FFT (rows) or at
// its first column (column tiles)
template <typename T, typename SCH, int SI, int TPF, int TC, int CPT, int PART>
__device__ inline void pow2_lean_exchange_part(cx<T>* v, T* plane, const uint32_t tau) {
	constexpr int LOGE = SCH::LOGE, E = 1 << LOGE, P = SCH::bits[0];
	constexpr int LOGR = SCH::bits[SI], R = 1 << LOGR, NB = E / R;
	constexpr int LOGS = SCH::logS(SI), S = 1 << LOGS;
	static_assert(CPT == 1 || (CPT == 2 && TC > 0 && TC % 2 == 0), "two columns per thread: column tiles of even width");
	typedef RealPair<T> pair_t;
#pragma unroll
	for (int b = 0; b < NB; b++) {
		const uint32_t t = tau + b * TPF;
		const uint32_t s = t & (S - 1);
		const uint32_t ob = ((t - s) << LOGR) + s;
		T* const wp = plane + pow2_lean_slot<TC, P>(ob);
#pragma unroll
		for (int k = 0; k < R; k++) {
			const cx<T> q0 = v[b + k * NB];
			if constexpr (CPT == 1) wp[pow2_lean_step<TC, P>(k * S)] = PART ? q0.y : q0.x;
			else { const cx<T> q1 = v[E + b + k * NB]; pair_t pr; pr.x = PART ? q0.y : q0.x; pr.y = PART ? q1.y : q1.x; *(pair_t*)(wp + pow2_lean_step<TC, P>(k * S)) = pr; }
		}
	}
	VKFFT_SYNC();
	const T* const rp = plane + pow2_lean_slot<TC, P>(tau);
#pragma unroll
	for (int m = 0; m < E; m++) {
		if constexpr (CPT == 1) { const T r = rp[pow2_lean_step<TC, P>(m * TPF)]; if (PART) v[m].y = r; else v[m].x = r; }
		else { const pair_t pr = *(const pair_t*)(rp + pow2_lean_step<TC, P>(m * TPF)); if (PART) { v[m].y = pr.x; v[E + m].y = pr.y; } else { v[m].x = pr.x; v[E + m].x = pr.y; } }
	}
}

// all stages of SCH on the thread's registers.  The plane must be free on entry (no barrier at the head); on exit the LAST exchange's reads may
// still be in flight in other waves: a caller that reuses the plane synchronises first.
// PFN: twiddles of the next stage requested ahead of each exchange (0: none)
template <typename T, typename SCH, int SI, int TPF, int TC, typename TW, int TWG, int CPT = 1, int PFN = 0, int PF = 0>
__device__ inline void pow2_lean_stages(cx<T>* v, T* plane, const TW lut, const uint32_t tau, const cx<T>* pf = nullptr) {
	pow2_lean_butterflies<T, SCH, SI, TPF, TW, TWG, CPT, PF>(v, lut, tau, pf);
	if constexpr (SI + 1 < SCH::NS) {
		constexpr int RN = 1 << SCH::bits[SI + 1];
		constexpr int PFX = PFN < RN - 1 ? PFN : RN - 1; // (a radix-R butterfly has R - 1 twiddles)
		cx<T> nx[PFX > 0 ? PFX : 1];
		if constexpr (PFX > 0) pow2_lean_prefetch<T, SCH, SI + 1, TPF, TW, PFX>(nx, lut, tau);
		pow2_lean_exchange_part<T, SCH, SI, TPF, TC, CPT, 0>(v, plane, tau);
		VKFFT_SYNC();
		pow2_lean_exchange_part<T, SCH, SI, TPF, TC, CPT, 1>(v, plane, tau);
		if constexpr (SI + 2 < SCH::NS) VKFFT_SYNC(); // another exchange will overwrite the plane
		pow2_lean_stages<T, SCH, SI + 1, TPF, TC, TW, TWG, CPT, PFN, PFX>(v, plane, lut, tau, nx);
	}
}

// Column tile in registers (v[cc*E + m] = point tau + m*TPF of column c + cc, CPT = 2 adjacent columns per thread) -> per-column contiguous order,
// through the plane laid out [column][point] with pitch L + 4 floats (the 8 column pairs x 8 consecutive points of a wave's 4-byte writes fall on
// distinct banks twice over; the 8-byte reads run along a column).  Item i of thread tid is the point pair (2kp, 2kp + 1) of column cc with
// idx = tid + i*NT, kp = idx % (L/2), cc = idx / (L/2): r[2i], r[2i+1].  The plane must be free on entry; it is free again after the caller's
// next barrier.
template <typename T, int L, int E, int TPF, int TC, int NT>
__device__ inline void pow2_lean_transpose(const cx<T>* v, cx<T>* r, T* plane, const uint32_t tid, const uint32_t c, const uint32_t tau) {
	constexpr int PT = L + 4;
	typedef RealPair<T> pair_t;
	static_assert(E * NT * 2 == L * TC && (L / 2) % 1 == 0, "two columns per thread");
#pragma unroll
	for (int part = 0; part < 2; part++) {
		if (part) VKFFT_SYNC();
		T* const w0 = plane + c * PT + tau;
#pragma unroll
		for (int m = 0; m < E; m++) { w0[m * TPF] = part ? v[m].y : v[m].x; w0[PT + m * TPF] = part ? v[E + m].y : v[E + m].x; }
		VKFFT_SYNC();
#pragma unroll
		for (int i = 0; i < E; i++) {
			const uint32_t idx = tid + i * NT;
			const uint32_t kp = idx % (L / 2), cc = idx / (L / 2);
			const pair_t pr = *(const pair_t*)(plane + cc * PT + 2u * kp);
			if (part) { r[2 * i].y = pr.x; r[2 * i + 1].y = pr.y; } else { r[2 * i].x = pr.x; r[2 * i + 1].x = pr.y; }
		}
	}
}

// ---- unit-stride rows of N = 2^13 ... 2^15 points, 32 points per thread, ONE row per workgroup (the two-tiles-per-CU form of pow2_row_kernel) ----
template <typename T, typename SCH, int WPE, int TWG, int PFN = 0>
__global__ void __launch_bounds__((1 << SCH::LOGN) >> SCH::LOGE, WPE) pow2_row_lean_kernel(const PassParams p) {
	constexpr int LOGN = SCH::LOGN, N = 1 << LOGN, LOGE = SCH::LOGE, E = 1 << LOGE, TPF = N / E;
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	__shared__ T plane[pow2_lean_plane_elems<SCH, 0>()];
	const uint32_t tau = threadIdx.x;
	uint32_t wg = p.reverseTiles ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
	const uint32_t f0 = wg % p.tilesPerG0; // one row per tile
	wg /= p.tilesPerG0;
	const uint32_t g1 = wg % p.dim[1].count, g2 = wg / p.dim[1].count;
	const GBuf gin = make_gbuf((const cx<T>*)p.in + ((int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride + (int64_t)f0 * p.dim[0].inStride));
	const GBuf gout = make_gbuf((cx<T>*)p.out + ((int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride + (int64_t)f0 * p.dim[0].outStride));
	const GBuf glut = make_gbuf(p.lut);
	const uint32_t lane = tau * ES;
	cx<T> v[E];
	if (p.padInN) { // zero padding: points of the padded range get an out-of-range offset (they read as zero and are not fetched)
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = gb_load<T>(gin, (tau + (uint32_t)(m * TPF) - p.padInL < p.padInN) ? kGbInvalid : lane, (uint32_t)(m * TPF) * ES);
	} else {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = gb_load<T>(gin, lane, (uint32_t)(m * TPF) * ES);
	}
	if (p.swapIn) {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cswap(v[m]);
	}
	pow2_lean_stages<T, SCH, 0, TPF, 0, TwGlobal<T>, TWG, 1, PFN>(v, plane, TwGlobal<T>{glut}, tau);
	if (p.swapOut) {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cswap(v[m]);
	}
	const T sc = (T)p.scale;
	if (sc != (T)1) {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cscale(v[m], sc);
	}
	if (p.padOutN) { // (the padded range of the output is not written)
#pragma unroll
		for (int m = 0; m < E; m++) gb_store<T>(gout, (tau + (uint32_t)(m * TPF) - p.padOutL < p.padOutN) ? kGbInvalid : lane, (uint32_t)(m * TPF) * ES, v[m]);
	} else {
#pragma unroll
		for (int m = 0; m < E; m++) gb_store<T>(gout, lane, (uint32_t)(m * TPF) * ES, v[m]);
	}
}

template <typename T, typename SCH, int WPE, int TWG, int PFN = 0> void pow2_row_lean_launch(const PassParams& prm, dim3 grid, hipStream_t s) {
	hipLaunchKernelGGL((pow2_row_lean_kernel<T, SCH, WPE, TWG, PFN>), grid, dim3((1 << SCH::LOGN) >> SCH::LOGE), 0, s, prm);
}

} // namespace vkfft_mi355x
