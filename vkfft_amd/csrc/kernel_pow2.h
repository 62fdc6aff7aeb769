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

namespace vkfft_mi355x {

#if defined(VKFFT_HOSTEMU)
#define VKFFT_WAVE_SYNC() hostemu::wave_sync()
#else
// orders this wave's LDS writes before its later LDS reads without an s_barrier
#define VKFFT_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
#endif

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
};

template <typename T, typename SCH, int SI, int TPF, int LDSPF>
__device__ inline void pow2_stages(cx<T>* v, cx<T>* ldsf, const cx<T>* __restrict__ lut, const uint32_t tau, const bool waveOnly) {
	constexpr int LOGE = SCH::LOGE, E = 1 << LOGE;
	constexpr int LOGR = SCH::bits[SI], R = 1 << LOGR, NB = E / R;
	constexpr int LOGS = SCH::logS(SI), S = 1 << LOGS;
	constexpr bool last = (SI == SCH::NS - 1);
#pragma unroll
	for (int b = 0; b < NB; b++) {
		cx<T> x[R];
#pragma unroll
		for (int i = 0; i < R; i++) x[i] = v[b + i * NB];
		const uint32_t t = tau + b * TPF;
		const uint32_t s = t & (S - 1);
		if constexpr (SI > 0) {
			constexpr int LO = SCH::lutOff(SI);
			const cx<T>* w = lut + LO + s;
#pragma unroll
			for (int i = 1; i < R; i++) x[i] = cmul(x[i], w[(i - 1) * S]);
		}
		dft<R, T>(x);
		if constexpr (last) {
#pragma unroll
			for (int k = 0; k < R; k++) v[b + k * NB] = x[k];
		} else {
			const uint32_t ob = ((t - s) << LOGR) + s;
#pragma unroll
			for (int k = 0; k < R; k++) {
				const uint32_t a = ob + k * S;
				ldsf[a + (a >> LOGE)] = x[k];
			}
		}
	}
	if constexpr (!last) {
		if (waveOnly) VKFFT_WAVE_SYNC(); else __syncthreads();
#pragma unroll
		for (int m = 0; m < E; m++) {
			const uint32_t a = tau + m * TPF;
			v[m] = ldsf[a + (a >> LOGE)];
		}
		if constexpr (SI + 2 < SCH::NS) { // another exchange will overwrite the buffer: all reads must be done first
			if (waveOnly) VKFFT_WAVE_SYNC(); else __syncthreads();
		}
		pow2_stages<T, SCH, SI + 1 < SCH::NS ? SI + 1 : SI, TPF, LDSPF>(v, ldsf, lut, tau, waveOnly);
	}
}

template <typename T, typename SCH, int FPW>
__global__ void __launch_bounds__(((1 << SCH::LOGN) >> SCH::LOGE) * FPW) pow2_row_kernel(const PassParams p) {
	constexpr int LOGN = SCH::LOGN, N = 1 << LOGN, LOGE = SCH::LOGE, E = 1 << LOGE, TPF = N / E;
	constexpr int LDSPF = SCH::NS > 1 ? N + (N >> LOGE) : 1;
	constexpr bool waveOnly = TPF <= 64; // an FFT never straddles wavefronts
	__shared__ cx<T> lds[FPW * LDSPF];
	const uint32_t tid = threadIdx.x;
	const uint32_t f = tid / TPF, tau = tid % TPF;
	uint32_t wg = blockIdx.x;
	const uint32_t tile = wg % p.tilesPerG0;
	wg /= p.tilesPerG0;
	const uint32_t g1 = wg % p.dim[1].count, g2 = wg / p.dim[1].count;
	const uint32_t g0 = tile * FPW + f;
	const bool valid = g0 < p.dim[0].count;
	const cx<T>* in = (const cx<T>*)p.in + ((int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride + (int64_t)g0 * p.dim[0].inStride);
	cx<T>* out = (cx<T>*)p.out + ((int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride + (int64_t)g0 * p.dim[0].outStride);
	cx<T> v[E];
	if (valid) {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = in[tau + m * TPF];
	} else {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cx<T>{(T)0, (T)0};
	}
	if (p.swapIn) {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cswap(v[m]);
	}
	pow2_stages<T, SCH, 0, TPF, LDSPF>(v, lds + f * LDSPF, (const cx<T>*)p.lut, tau, waveOnly);
	if (p.swapOut) {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cswap(v[m]);
	}
	const T sc = (T)p.scale;
	if (sc != (T)1) {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cscale(v[m], sc);
	}
	if (valid) {
#pragma unroll
		for (int m = 0; m < E; m++) out[tau + m * TPF] = v[m];
	}
}

// ---- registry --------------------------------------------------------------------------------------------
struct Pow2Variant {
	int log2n; bool dp; int bits[4]; int fpw; int threads;
	void (*launch)(const PassParams&, dim3, hipStream_t);
};

template <typename T, typename SCH, int FPW> void pow2_row_launch(const PassParams& prm, dim3 grid, hipStream_t s) {
	constexpr int threads = ((1 << SCH::LOGN) >> SCH::LOGE) * FPW;
	hipLaunchKernelGGL((pow2_row_kernel<T, SCH, FPW>), grid, dim3(threads), 0, s, prm);
}

#define VKFFT_P2(T, dp, b0, b1, b2, b3, fpw) \
	{ (b0) + (b1) + (b2) + (b3), dp, {b0, b1, b2, b3}, fpw, (((1 << ((b0) + (b1) + (b2) + (b3))) >> Pow2Sched<b0, b1, b2, b3>::LOGE) * (fpw)), &pow2_row_launch<T, Pow2Sched<b0, b1, b2, b3>, fpw> }

// first entry of each (log2n, dp) is the default; VKFFT_MI355X_P2V<log2n>=k selects the k-th (tuning)
static const Pow2Variant kPow2Variants[] = {
	// fp32
	VKFFT_P2(float, false, 2, 0, 0, 0, 64),
	VKFFT_P2(float, false, 3, 0, 0, 0, 64),
	VKFFT_P2(float, false, 4, 0, 0, 0, 64),
	VKFFT_P2(float, false, 3, 2, 0, 0, 32),
	VKFFT_P2(float, false, 3, 3, 0, 0, 32),
	VKFFT_P2(float, false, 4, 3, 0, 0, 16), VKFFT_P2(float, false, 3, 2, 2, 0, 16),
	VKFFT_P2(float, false, 3, 3, 2, 0, 8), VKFFT_P2(float, false, 4, 4, 0, 0, 16), VKFFT_P2(float, false, 4, 4, 0, 0, 8), VKFFT_P2(float, false, 3, 3, 2, 0, 4),
	VKFFT_P2(float, false, 3, 3, 3, 0, 4), VKFFT_P2(float, false, 4, 3, 2, 0, 8), VKFFT_P2(float, false, 4, 3, 2, 0, 4), VKFFT_P2(float, false, 3, 3, 3, 0, 2),
	VKFFT_P2(float, false, 4, 3, 3, 0, 4), VKFFT_P2(float, false, 4, 3, 3, 0, 2), VKFFT_P2(float, false, 4, 4, 2, 0, 4), VKFFT_P2(float, false, 3, 3, 2, 2, 2),
	VKFFT_P2(float, false, 4, 4, 3, 0, 2), VKFFT_P2(float, false, 4, 4, 3, 0, 1), VKFFT_P2(float, false, 3, 3, 3, 2, 1), VKFFT_P2(float, false, 4, 4, 3, 0, 4),
	VKFFT_P2(float, false, 4, 4, 4, 0, 1), VKFFT_P2(float, false, 3, 3, 3, 3, 1), VKFFT_P2(float, false, 4, 4, 4, 0, 2),
	VKFFT_P2(float, false, 4, 3, 3, 3, 1), VKFFT_P2(float, false, 4, 4, 4, 1, 1),
	VKFFT_P2(float, false, 4, 4, 3, 3, 1), VKFFT_P2(float, false, 4, 4, 4, 2, 1),
	// fp64
	VKFFT_P2(double, true, 2, 0, 0, 0, 64),
	VKFFT_P2(double, true, 3, 0, 0, 0, 64),
	VKFFT_P2(double, true, 4, 0, 0, 0, 64),
	VKFFT_P2(double, true, 3, 2, 0, 0, 32),
	VKFFT_P2(double, true, 3, 3, 0, 0, 32),
	VKFFT_P2(double, true, 3, 2, 2, 0, 16),
	VKFFT_P2(double, true, 3, 3, 2, 0, 8),
	VKFFT_P2(double, true, 3, 3, 3, 0, 4),
	VKFFT_P2(double, true, 3, 3, 2, 2, 2), VKFFT_P2(double, true, 4, 3, 3, 0, 2),
	VKFFT_P2(double, true, 3, 3, 3, 2, 1), VKFFT_P2(double, true, 4, 4, 3, 0, 1),
	VKFFT_P2(double, true, 3, 3, 3, 3, 1), VKFFT_P2(double, true, 4, 4, 4, 0, 1),
	VKFFT_P2(double, true, 4, 3, 3, 3, 1),
};
constexpr int kNumPow2Variants = (int)(sizeof(kPow2Variants) / sizeof(kPow2Variants[0]));

inline int launch_pow2(const PassPlan& pp, const PassParams& prm, hipStream_t stream) {
	const uint64_t grid64 = (uint64_t)prm.tilesPerG0 * prm.dim[1].count * prm.dim[2].count;
	if (grid64 == 0) return 0;
	if (grid64 > 0x7fffffffull || pp.variant < 0 || pp.variant >= kNumPow2Variants) return 4039;
	kPow2Variants[pp.variant].launch(prm, dim3((uint32_t)grid64), stream);
	return hipGetLastError() == hipSuccess ? 0 : 4039;
}

} // namespace vkfft_mi355x
