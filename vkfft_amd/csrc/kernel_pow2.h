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

// LDS slot of FFT element a: row kernels pad the index (a + a>>LOGE) inside the FFT's own slab; column
// kernels keep TCP = TC+1 columns per element row ([a][c], odd pitch) and ldsf already points at column c.
template <int TCP, int LOGE> __device__ inline uint32_t pow2_slot(uint32_t a) {
	if constexpr (TCP == 0) return a + (a >> LOGE);
	else return a * TCP;
}

template <typename T, typename SCH, int SI, int TPF, int TCP>
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
				ldsf[pow2_slot<TCP, LOGE>(a)] = x[k];
			}
		}
	}
	if constexpr (!last) {
		if (waveOnly) VKFFT_WAVE_SYNC(); else __syncthreads();
#pragma unroll
		for (int m = 0; m < E; m++) {
			const uint32_t a = tau + m * TPF;
			v[m] = ldsf[pow2_slot<TCP, LOGE>(a)];
		}
		if constexpr (SI + 2 < SCH::NS) { // another exchange will overwrite the buffer: all reads must be done first
			if (waveOnly) VKFFT_WAVE_SYNC(); else __syncthreads();
		}
		pow2_stages<T, SCH, SI + 1 < SCH::NS ? SI + 1 : SI, TPF, TCP>(v, ldsf, lut, tau, waveOnly);
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
	pow2_stages<T, SCH, 0, TPF, 0>(v, lds + f * LDSPF, (const cx<T>*)p.lut, tau, waveOnly);
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

// ---- strided-tile ("column") kernel: Four-Step passes and the non-unit-stride axes of 2D/3D transforms ----
// A workgroup transforms TC neighbouring columns; lanes run across the columns so that every global
// access is a TC*sizeof(complex) contiguous segment (256 B for TC=32 fp32).  Same register-resident
// Stockham core as the row kernel; the LDS exchange is [element][column] with an odd pitch, conflict-free
// in both directions.  Optional fused epilogue: Four-Step twiddle (two-level LUT) and a transposed store
// (each column written out as one contiguous run) for the first Four-Step pass.
template <typename T, typename SCH, int TC>
__global__ void __launch_bounds__(((1 << SCH::LOGN) >> SCH::LOGE) * TC) pow2_col_kernel(const PassParams p) {
	constexpr int LOGN = SCH::LOGN, L = 1 << LOGN, LOGE = SCH::LOGE, E = 1 << LOGE, TPF = L / E;
	constexpr int TCP = TC + 1, NT = TPF * TC;
	__shared__ cx<T> lds[L * TCP];
	const uint32_t tid = threadIdx.x;
	const uint32_t c = tid % TC, tau = tid / TC;
	uint32_t wg = blockIdx.x;
	const uint32_t tile = wg % p.tilesPerG0;
	wg /= p.tilesPerG0;
	const uint32_t g1 = wg % p.dim[1].count, g2 = wg / p.dim[1].count;
	const uint32_t col0 = tile * TC;
	const bool valid = col0 + c < p.dim[0].count;
	const int64_t inB = (int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride + (int64_t)col0 * p.dim[0].inStride;
	const int64_t outB = (int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride + (int64_t)col0 * p.dim[0].outStride;
	const cx<T>* in = (const cx<T>*)p.in + inB + (int64_t)c * p.dim[0].inStride;
	cx<T> v[E];
	if (valid) {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = in[(int64_t)(tau + m * TPF) * p.inStrideJ];
	} else {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cx<T>{(T)0, (T)0};
	}
	if (p.swapIn) {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cswap(v[m]);
	}
	pow2_stages<T, SCH, 0, TPF, TCP>(v, lds + c, (const cx<T>*)p.lut, tau, false);
	if (p.swapOut) {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cswap(v[m]);
	}
	if (p.postOp == OP_TWIDDLE_4STEP) {
		uint32_t colIdx, rr;
		p.fsColDiv.divmod(col0 + c, colIdx, rr);
		const cx<T>* tab = (const cx<T>*)p.aux;
		const uint32_t loMask = (1u << p.fsLoBits) - 1u;
#pragma unroll
		for (int m = 0; m < E; m++) {
			const uint32_t e = (tau + m * TPF) * colIdx;
			v[m] = cmul(v[m], cmul(tab[e & loMask], tab[(loMask + 1u) + (e >> p.fsLoBits)]));
		}
	}
	const T sc = (T)p.scale;
	if (sc != (T)1) {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cscale(v[m], sc);
	}
	if (p.colModeOut) {
		if (valid) {
			cx<T>* out = (cx<T>*)p.out + outB + (int64_t)c * p.dim[0].outStride;
#pragma unroll
			for (int m = 0; m < E; m++) out[(int64_t)(tau + m * TPF) * p.outStrideJ] = v[m];
		}
	} else {
		// transposed store: column c becomes the contiguous run out[c*dim0.outStride + k*outStrideJ], lanes along k
		if constexpr (SCH::NS > 1) __syncthreads(); // the last exchange's reads are complete
#pragma unroll
		for (int m = 0; m < E; m++) lds[(tau + m * TPF) * TCP + c] = v[m];
		__syncthreads();
		const uint32_t nvalid = p.dim[0].count - col0 < (uint32_t)TC ? p.dim[0].count - col0 : (uint32_t)TC;
		cx<T>* out = (cx<T>*)p.out + outB;
#pragma unroll
		for (int i = 0; i < E; i++) {
			const uint32_t idx = tid + i * NT;
			const uint32_t k = idx % L, cc = idx / L;
			if (cc < nvalid) out[(int64_t)cc * p.dim[0].outStride + (int64_t)k * p.outStrideJ] = lds[k * TCP + cc];
		}
	}
}

// ---- registry --------------------------------------------------------------------------------------------
struct Pow2Variant {
	int log2n; bool dp; int bits[4]; int fpw; int threads; // fpw: FFTs per workgroup (row) / columns per workgroup (col)
	void (*launch)(const PassParams&, dim3, hipStream_t);
};

template <typename T, typename SCH, int FPW> void pow2_row_launch(const PassParams& prm, dim3 grid, hipStream_t s) {
	constexpr int threads = ((1 << SCH::LOGN) >> SCH::LOGE) * FPW;
	hipLaunchKernelGGL((pow2_row_kernel<T, SCH, FPW>), grid, dim3(threads), 0, s, prm);
}

template <typename T, typename SCH, int TC> void pow2_col_launch(const PassParams& prm, dim3 grid, hipStream_t s) {
	constexpr int threads = ((1 << SCH::LOGN) >> SCH::LOGE) * TC;
	hipLaunchKernelGGL((pow2_col_kernel<T, SCH, TC>), grid, dim3(threads), 0, s, prm);
}

#define VKFFT_P2C(T, dp, b0, b1, b2, b3, tc) \
	{ (b0) + (b1) + (b2) + (b3), dp, {b0, b1, b2, b3}, tc, (((1 << ((b0) + (b1) + (b2) + (b3))) >> Pow2Sched<b0, b1, b2, b3>::LOGE) * (tc)), &pow2_col_launch<T, Pow2Sched<b0, b1, b2, b3>, tc> }

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

// column kernels: first entry of each (log2n, dp) is the default; VKFFT_MI355X_P2C<log2n>=k selects the k-th
static const Pow2Variant kPow2ColVariants[] = {
	VKFFT_P2C(float, false, 2, 2, 0, 0, 32), VKFFT_P2C(float, false, 2, 2, 0, 0, 16),
	VKFFT_P2C(float, false, 3, 2, 0, 0, 32), VKFFT_P2C(float, false, 3, 2, 0, 0, 16),
	VKFFT_P2C(float, false, 3, 3, 0, 0, 32), VKFFT_P2C(float, false, 3, 3, 0, 0, 16),
	VKFFT_P2C(float, false, 4, 3, 0, 0, 32), VKFFT_P2C(float, false, 4, 3, 0, 0, 16), VKFFT_P2C(float, false, 3, 2, 2, 0, 32),
	VKFFT_P2C(float, false, 4, 4, 0, 0, 32), VKFFT_P2C(float, false, 4, 4, 0, 0, 16), VKFFT_P2C(float, false, 3, 3, 2, 0, 32), VKFFT_P2C(float, false, 3, 3, 2, 0, 16),
	VKFFT_P2C(float, false, 4, 3, 2, 0, 16), VKFFT_P2C(float, false, 4, 3, 2, 0, 32), VKFFT_P2C(float, false, 3, 3, 3, 0, 16),
	VKFFT_P2C(float, false, 4, 3, 3, 0, 16), VKFFT_P2C(float, false, 4, 3, 3, 0, 8),
	VKFFT_P2C(double, true, 2, 2, 0, 0, 16),
	VKFFT_P2C(double, true, 3, 2, 0, 0, 16),
	VKFFT_P2C(double, true, 3, 3, 0, 0, 16),
	VKFFT_P2C(double, true, 3, 2, 2, 0, 16), VKFFT_P2C(double, true, 4, 3, 0, 0, 16),
	VKFFT_P2C(double, true, 3, 3, 2, 0, 16), VKFFT_P2C(double, true, 4, 4, 0, 0, 16),
	VKFFT_P2C(double, true, 3, 3, 3, 0, 8), VKFFT_P2C(double, true, 4, 3, 2, 0, 8), VKFFT_P2C(double, true, 3, 3, 3, 0, 16),
	VKFFT_P2C(double, true, 4, 3, 3, 0, 8),
};
constexpr int kNumPow2ColVariants = (int)(sizeof(kPow2ColVariants) / sizeof(kPow2ColVariants[0]));

inline int launch_pow2(const PassPlan& pp, const PassParams& prm, hipStream_t stream) {
	const uint64_t grid64 = (uint64_t)prm.tilesPerG0 * prm.dim[1].count * prm.dim[2].count;
	if (grid64 == 0) return 0;
	const bool col = pp.kernel == KERNEL_POW2_COL;
	if (grid64 > 0x7fffffffull || pp.variant < 0 || pp.variant >= (col ? kNumPow2ColVariants : kNumPow2Variants)) return 4039;
	(col ? kPow2ColVariants : kPow2Variants)[pp.variant].launch(prm, dim3((uint32_t)grid64), stream);
	return hipGetLastError() == hipSuccess ? 0 : 4039;
}

} // namespace vkfft_mi355x
