// Core of the hand-specialised power-of-two kernels (schedule, LDS slots, twiddle sources, the register-resident Stockham
// stages, registry entry type) — shared by kernel_pow2.h and kernel_blue_r2r.h.
//
// Hand-specialised power-of-two kernels for the headline path (batched unit-stride C2C, N = 2^2..2^13/14).
//
// Design (MI355X-first, not a translation of the reference's generated code):
//   * every thread keeps E = 2^LOGE points in registers; register m holds point tau + m*TPF of its FFT
//     (TPF = N/E threads per FFT), which is simultaneously
//        - the coalesced global access pattern (lane tau -> consecutive 8/16-byte elements),
//        - the input set of the thread's Stockham butterflies in EVERY stage (t + i*N/R), and
//        - the output set of the last stage,
//     so data goes HBM -> registers -> (LDS exchange between stages only) -> registers -> HBM:
//     one HBM read, one HBM write, (stages-1) LDS round trips, no LDS staging of loads/stores;
//   * stage radices up to 16 (2^B0 * 2^B1 * ...), butterflies fully unrolled in registers;
//   * twiddles come from a per-stage LUT laid out [(i-1)*S + s] so that the lanes of a wave read
//     consecutive entries (the same layout the reference's LUT uses, vkFFT_ManageLUT.h:985-1011);
//   * LDS exchange index a -> a + (a >> LOGE): conflict-free ds_write_b64 for the strided Stockham
//     scatter (t-s)*R + s + k*S and conflict-free ds_read_b64 for the gather tau + m*TPF;
//   * FFTs with TPF <= 64 live inside one wavefront: their exchanges need no s_barrier, only LDS
//     ordering within the wave (wave-synchronous exchange);
//   * inverse transforms reuse the forward code through the re/im swap identity; normalisation is a
//     multiply at the store.
#pragma once
#include "engine.h"
#include "butterflies.h"
#include "memops.h"
#include <cstdlib>

namespace vkfft_mi355x {



template <int B0, int B1, int B2, int B3> struct Pow2Sched {
	static constexpr int bits[4] = {B0, B1, B2, B3};
	static constexpr int NS = (B0 > 0) + (B1 > 0) + (B2 > 0) + (B3 > 0);
	static constexpr int LOGN = B0 + B1 + B2 + B3;
	static constexpr int LOGE = B0 > B1 ? (B0 > B2 ? (B0 > B3 ? B0 : B3) : (B2 > B3 ? B2 : B3)) : (B1 > B2 ? (B1 > B3 ? B1 : B3) : (B2 > B3 ? B2 : B3));
	__host__ __device__ static constexpr int logS(int si) { return si == 0 ? 0 : si == 1 ? B0 : si == 2 ? B0 + B1 : B0 + B1 + B2; }
	__host__ __device__ static constexpr int lutOff(int si) { // complex elements before stage si's run
		int off = 0;
		for (int j = 1; j < si; j++) off += ((1 << bits[j]) - 1) << logS(j);
		return off;
	}
	__host__ __device__ static constexpr int lutTotal() { return lutOff(NS); } // complex elements of all stage twiddle runs
};

// LDS slot of FFT element a: row kernels pad the index (a + a>>LOGE) inside the FFT's own slab; column
// kernels keep TCP = TC+1 columns per element row ([a][c], odd pitch) and ldsf already points at column c.
template <int TCP, int LOGE> __device__ inline uint32_t pow2_slot(uint32_t a) {
	if constexpr (TCP == 0) return a + (a >> LOGE);
	else return a * TCP;
}

// stage-twiddle source: global LUT through a buffer resource (one-tile kernels) or a copy of the LUT staged in LDS once per
// workgroup (persistent fused Four-Step kernel: short-latency reads need no deep prefetch, which is what drives the
// register count of the radix-32 stages, and the vector-memory queue stays free for the tile's own loads)
template <typename T> struct TwGlobal {
	GBuf lut;
	__device__ inline cx<T> get(uint32_t s, uint32_t constOff) const { return gb_load<T>(lut, s * (uint32_t)sizeof(cx<T>), constOff * (uint32_t)sizeof(cx<T>)); }
};
template <typename T> struct TwLds {
	const cx<T>* tab;
	__device__ inline cx<T> get(uint32_t s, uint32_t constOff) const { return tab[constOff + s]; }
};

// two neighbouring complex values as one 16-byte LDS access (column kernels that keep two adjacent columns per thread)
template <typename T> struct alignas(4 * sizeof(T)) cx2 { cx<T> a, b; };

// CPT = columns per thread (column kernels only): 1, or 2 adjacent columns kept in v[0..E) and v[E..2E) — 16-byte global and LDS
// accesses for fp32 data, one twiddle read serves both columns (needs an even column pitch TCP and ldsf at an even column)
// RAW = 1: barriers without the compiler's fence (VKFFT_SYNC_RAW, memops.h) — for kernels that keep LDS-DMA transfers in flight across them
template <typename T, typename SCH, int SI, int TPF, int TCP, typename TW, int CPT = 1, int RAW = 0>
__device__ inline void pow2_stages(cx<T>* v, cx<T>* ldsf, const TW lut, const uint32_t tau, const bool waveOnly) {
	constexpr int LOGE = SCH::LOGE, E = 1 << LOGE;
	constexpr int LOGR = SCH::bits[SI], R = 1 << LOGR, NB = E / R;
	constexpr int LOGS = SCH::logS(SI), S = 1 << LOGS;
	constexpr bool last = (SI == SCH::NS - 1);
	static_assert(CPT == 1 || (CPT == 2 && TCP > 0 && TCP % 2 == 0), "two columns per thread: column kernels with an even pitch");
#pragma unroll
	for (int b = 0; b < NB; b++) {
		cx<T> x[CPT][R];
#pragma unroll
		for (int cc = 0; cc < CPT; cc++) {
#pragma unroll
			for (int i = 0; i < R; i++) x[cc][i] = v[cc * E + b + i * NB];
		}
		const uint32_t t = tau + b * TPF;
		const uint32_t s = t & (S - 1);
		if constexpr (SI > 0) {
			constexpr int LO = SCH::lutOff(SI);
#pragma unroll
			for (int i = 1; i < R; i++) {
				const cx<T> w = lut.get(s, (uint32_t)(LO + (i - 1) * S));
#pragma unroll
				for (int cc = 0; cc < CPT; cc++) x[cc][i] = cmul(x[cc][i], w);
			}
		}
#pragma unroll
		for (int cc = 0; cc < CPT; cc++) dft<R, T>(x[cc]);
		if constexpr (last) {
#pragma unroll
			for (int cc = 0; cc < CPT; cc++) {
#pragma unroll
				for (int k = 0; k < R; k++) v[cc * E + b + k * NB] = x[cc][k];
			}
		} else {
#if defined(VKFFT_PROBE_NO_EXCHANGE) // tools/probe_exchange.hip only: timing of the kernel without its exchange (results meaningless)
#pragma unroll
			for (int cc = 0; cc < CPT; cc++) {
#pragma unroll
				for (int k = 0; k < R; k++) v[cc * E + b + k * NB] = x[cc][k];
			}
			continue;
#endif
			const uint32_t ob = ((t - s) << LOGR) + s;
#pragma unroll
			for (int k = 0; k < R; k++) {
				const uint32_t a = ob + k * S;
				if constexpr (CPT == 1) ldsf[pow2_slot<TCP, LOGE>(a)] = x[0][k];
				else *(cx2<T>*)(ldsf + pow2_slot<TCP, LOGE>(a)) = cx2<T>{x[0][k], x[1][k]};
			}
		}
	}
	if constexpr (!last) {
#if defined(VKFFT_PROBE_NO_EXCHANGE)
		pow2_stages<T, SCH, SI + 1 < SCH::NS ? SI + 1 : SI, TPF, TCP, TW, CPT, RAW>(v, ldsf, lut, tau, waveOnly);
		return;
#endif
		if (waveOnly) VKFFT_WAVE_SYNC(); else if constexpr (RAW) VKFFT_SYNC_RAW(); else VKFFT_SYNC();
#pragma unroll
		for (int m = 0; m < E; m++) {
			const uint32_t a = tau + m * TPF;
			if constexpr (CPT == 1) v[m] = ldsf[pow2_slot<TCP, LOGE>(a)];
			else { const cx2<T> q = *(const cx2<T>*)(ldsf + pow2_slot<TCP, LOGE>(a)); v[m] = q.a; v[E + m] = q.b; }
		}
		if constexpr (SI + 2 < SCH::NS) { // another exchange will overwrite the buffer: all reads must be done first
			if (waveOnly) VKFFT_WAVE_SYNC(); else if constexpr (RAW) VKFFT_SYNC_RAW(); else VKFFT_SYNC();
		}
		pow2_stages<T, SCH, SI + 1 < SCH::NS ? SI + 1 : SI, TPF, TCP, TW, CPT, RAW>(v, ldsf, lut, tau, waveOnly);
	}
}

inline unsigned pow2_num_cus() { // CUs of the CURRENT device (plans are made and launched with their device current)
#if defined(VKFFT_HOSTEMU)
	return 4;
#else
	static unsigned cache[64] = {};
	int dev = 0, v = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
	if (!cache[dev]) cache[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? (unsigned)v : 256u;
	return cache[dev];
#endif
}

struct Pow2Variant {
	int log2n; bool dp; int bits[4]; int fpw; int threads; // fpw: FFTs per workgroup (row) / columns per workgroup (col)
	void (*launch)(const PassParams&, dim3, hipStream_t);
	const char* name = nullptr; // the __global__ function behind the entry when it is not the family's first (vkfftMI355XDescribePlan, bench labels)
	bool noPadMasks = false;    // the kernel has no zero-padding masks: a padded pass takes the next entry of its size
};

} // namespace vkfft_mi355x
