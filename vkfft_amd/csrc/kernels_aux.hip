// Element-wise helpers of the convolution and zero-padding configurations: own translation unit.
//   * conv_pointwise_kernel — the product the reference merges into the last axis of a convolution plan
//     (appendKernelConvolution, vkFFT_CodeGen/vkFFT_KernelsLevel1/PrePostProcessing/vkFFT_Convolution.h:125-447): here a separate
//     HBM-bound pass between the forward and the inverse transform (one read of every spectrum and kernel element, one write);
//   * zero_slab_kernel — zero padding (vkFFT_KernelsLevel0/vkFFT_Zeropad.h:28): the reference does not read the padded range and
//     takes it as zero; here the range is written with zeros before the transform that would read it.
#include "engine.h"
#include "butterflies.h"
#include "memops.h"
#include "kernel_generic.h"

namespace vkfft_mi355x {

constexpr int kConvMaxMatrix = 8;

static unsigned pow2_num_cus_aux() { // CUs of the current device
#if defined(VKFFT_HOSTEMU)
	return 4;
#else
	int dev = 0, v = 0;
	if (hipGetDevice(&dev) != hipSuccess) return 256;
	return (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? (unsigned)v : 256u;
#endif
}

template <typename T> __global__ void __launch_bounds__(256) conv_pointwise_kernel(const ConvParams p) {
	const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= p.systemStride) return;
	const uint32_t b = blockIdx.y; // batch of the input
	cx<T>* const data = (cx<T>*)p.data;
	const cx<T>* const ker = (const cx<T>*)p.kernel;
	const uint32_t m = p.matrix, cf = p.coordinates;
	if (m <= 1) {
		// 1x1 convolution: every coordinate is multiplied by its own kernel component
		for (uint32_t v = 0; v < cf; v++) {
			cx<T> x = data[((uint64_t)b * cf + v) * p.systemStride + e];
			if (p.conjugate == 1) x = cconj(x);
			for (uint32_t f = p.numKernels; f-- > 0;) {
				cx<T> k = ker[((uint64_t)f * p.kernelSystems + v) * p.systemStride + e];
				if (p.conjugate == 2) k = cconj(k);
				cx<T> y = cmul(k, x);
				if (p.crossPower) { const T n = y.x * y.x + y.y * y.y; const T s = n > (T)0 ? (T)1 / sqrt(n) : (T)0; y = cscale(y, s); }
				data[(((uint64_t)f * p.batches + b) * cf + v) * p.systemStride + e] = y;
			}
		}
		return;
	}
	cx<T> x[kConvMaxMatrix];
	for (uint32_t l = 0; l < m; l++) {
		x[l] = data[((uint64_t)b * m + l) * p.systemStride + e];
		if (p.conjugate == 1) x[l] = cconj(x[l]);
	}
	for (uint32_t f = p.numKernels; f-- > 0;) { // kernel 0 last: its output replaces the input
		for (uint32_t j = 0; j < m; j++) {
			cx<T> acc = cx<T>{(T)0, (T)0};
			for (uint32_t l = 0; l < m; l++) {
				cx<T> k = ker[((uint64_t)f * p.kernelSystems + conv_kernel_index(j, l, m, p.symmetric != 0)) * p.systemStride + e];
				if (p.conjugate == 2) k = cconj(k);
				acc = cadd(acc, cmul(k, x[l]));
			}
			if (p.crossPower) { const T n = acc.x * acc.x + acc.y * acc.y; const T s = n > (T)0 ? (T)1 / sqrt(n) : (T)0; acc = cscale(acc, s); }
			data[(((uint64_t)f * p.batches + b) * m + j) * p.systemStride + e] = acc;
		}
	}
}

int launch_conv_pointwise(const ConvParams& p, bool dp, hipStream_t stream) {
	if (p.matrix > (uint32_t)kConvMaxMatrix) return 4039;
	if (p.systemStride == 0 || p.batches == 0) return 0;
	const dim3 grid((uint32_t)((p.systemStride + 255) / 256), p.batches);
	if (dp) hipLaunchKernelGGL(conv_pointwise_kernel<double>, grid, dim3(256), 0, stream, p);
	else hipLaunchKernelGGL(conv_pointwise_kernel<float>, grid, dim3(256), 0, stream, p);
	return hipGetLastError() == hipSuccess ? 0 : 4039;
}

// zeroes, in every system, the elements whose coordinate on `axis` lies in [left, right); 32-bit words, `words` per element
__global__ void __launch_bounds__(256) zero_slab_kernel(const ZeroParams p) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t n = 1;
	uint32_t ext[4];
	for (int d = 0; d < 4; d++) { ext[d] = d == (int)p.axis ? p.right - p.left : p.size[d]; n *= ext[d]; }
	if (i >= n) return;
	uint64_t r = i, off = 0;
	for (int d = 0; d < 4; d++) {
		const uint32_t c = (uint32_t)(r % ext[d]); r /= ext[d];
		off += (uint64_t)(d == (int)p.axis ? c + p.left : c) * p.stride[d];
	}
	uint32_t* w = (uint32_t*)p.base + ((uint64_t)blockIdx.y * p.systemStride + off) * p.words;
	for (uint32_t k = 0; k < p.words; k++) w[k] = 0u;
}

int launch_zero_slab(const ZeroParams& p, hipStream_t stream) {
	if (p.right <= p.left || p.systems == 0) return 0;
	uint64_t n = 1;
	for (int d = 0; d < 4; d++) n *= d == (int)p.axis ? p.right - p.left : p.size[d];
	if (n == 0) return 0;
	for (uint32_t s0 = 0; s0 < p.systems; s0 += 65535u) { // grid.y limit
		ZeroParams q = p;
		q.base = (char*)p.base + (uint64_t)s0 * p.systemStride * p.words * 4u;
		const uint32_t ns = p.systems - s0 < 65535u ? p.systems - s0 : 65535u;
		hipLaunchKernelGGL(zero_slab_kernel, dim3((uint32_t)((n + 255) / 256), ns), dim3(256), 0, stream, q);
	}
	return hipGetLastError() == hipSuccess ? 0 : 4039;
}

// out[j * outStrideJ + c * dim0.outStride + outer] = in[j * inStrideJ + c * dim0.inStride + outer] for j < L, c < dim0.count: 32 x 32 tiles through
// LDS, lanes along c on the read side and along j on the write side (the planner names the dimensions so that these are the unit-stride ones)
template <typename T> __global__ void __launch_bounds__(256) transpose_kernel(const PassParams p) {
	__shared__ cx<T> tile[32][33];
	const uint32_t tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;
	const uint32_t tilesC = (p.dim[0].count + 31u) / 32u, tilesJ = (p.L + 31u) / 32u;
	uint32_t b = blockIdx.x;
	const uint32_t tc = b % tilesC; b /= tilesC;
	const uint32_t tj = b % tilesJ; b /= tilesJ;
	const uint32_t g1 = b % p.dim[1].count, g2 = b / p.dim[1].count;
	const cx<T>* in = (const cx<T>*)p.in + ((int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride);
	cx<T>* out = (cx<T>*)p.out + ((int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride);
	const uint32_t j0 = tj * 32u, c0 = tc * 32u;
#pragma unroll
	for (uint32_t r = 0; r < 4; r++) {
		const uint32_t j = j0 + ty + 8u * r, c = c0 + tx;
		if (j < p.L && c < p.dim[0].count) tile[ty + 8u * r][tx] = in[(int64_t)j * p.inStrideJ + (int64_t)c * p.dim[0].inStride];
	}
	VKFFT_SYNC();
#pragma unroll
	for (uint32_t r = 0; r < 4; r++) {
		const uint32_t c = c0 + ty + 8u * r, j = j0 + tx;
		if (j < p.L && c < p.dim[0].count) out[(int64_t)j * p.outStrideJ + (int64_t)c * p.dim[0].outStride] = tile[tx][ty + 8u * r];
	}
}

// Pre / post map of a real transform as a pass of its own (coverage path: real transforms whose embedding length no fused kernel serves — it needs
// Bluestein along a strided axis, or more points than the fused Bluestein kernels hold — run as  map -> complex plan of the embedding length on
// dense scratch rows -> map).  The reference builds these maps into its Bluestein kernels (vkFFT_R2R.h, vkFFT_R2C.h:27, vkFFT_Scheduler.h:2271-2280);
// the maps themselves are the element-wise full-length forms of kernel_generic.h (pre_gather / post_scatter), addressed by the natural index.
//   preNat  = 1: scratch[row][n] = pre-map of the row's elements, n < L                    (row side = in, strides dim[].inStride / inStrideJ)
//   postNat = 1: FFT output a of scratch[row][.] -> post-map -> the row's output element(s)  (row side = out, strides dim[].outStride / outStrideJ)
// rows are enumerated by dim[0..2]; the scratch rows are dense: row index = g0 + count0 * (g1 + count1 * g2), pitch L
template <typename T> __global__ void __launch_bounds__(256) real_map_kernel(const PassParams p) {
	const uint32_t chunks = (p.L + 255u) / 256u;
	const uint32_t row = blockIdx.x / chunks, n = (blockIdx.x % chunks) * 256u + threadIdx.x;
	if (n >= p.L) return;
	const uint32_t g0 = row % p.dim[0].count, r1 = row / p.dim[0].count;
	const uint32_t g1 = r1 % p.dim[1].count, g2 = r1 / p.dim[1].count;
	if (p.preNat) {
		const int64_t rowBase = (int64_t)g0 * p.dim[0].inStride + (int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride;
		const Io64<T> io{p.in, nullptr, rowBase, 0, p.inStrideJ, 1};
		cx<T> v = pre_gather<T>(p, io, n, 0, p.preOp);
		if (p.swapIn) v = cswap(v);
		((cx<T>*)p.out)[(int64_t)row * p.L + n] = v;
	} else {
		const int64_t rowBase = (int64_t)g0 * p.dim[0].outStride + (int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride;
		const Io64<T> io{nullptr, p.out, 0, rowBase, 1, p.outStrideJ};
		cx<T> v = ((const cx<T>*)p.in)[(int64_t)row * p.L + n];
		if (p.swapOut) v = cswap(v);
		post_scatter<T>(p, io, n, v, 0, 0, p.postOp, p.natOutLen);
	}
}

int launch_real_map(const PassPlan& pp, const PassParams& prm, hipStream_t stream) {
	const uint64_t blocks = (uint64_t)((prm.L + 255u) / 256u) * prm.dim[0].count * prm.dim[1].count * prm.dim[2].count;
	if (blocks == 0) return 0;
	if (blocks > 0x7fffffffull) return 4039;
	if (pp.dp) hipLaunchKernelGGL(real_map_kernel<double>, dim3((uint32_t)blocks), dim3(256), 0, stream, prm);
	else hipLaunchKernelGGL(real_map_kernel<float>, dim3((uint32_t)blocks), dim3(256), 0, stream, prm);
	return hipGetLastError() == hipSuccess ? 0 : 4039;
}

int launch_transpose(const PassPlan& pp, const PassParams& prm, hipStream_t stream) {
	const uint64_t blocks = (uint64_t)((prm.dim[0].count + 31u) / 32u) * ((prm.L + 31u) / 32u) * prm.dim[1].count * prm.dim[2].count;
	if (blocks == 0) return 0;
	if (blocks > 0x7fffffffull) return 4039;
	if (pp.dp) hipLaunchKernelGGL(transpose_kernel<double>, dim3((uint32_t)blocks), dim3(256), 0, stream, prm);
	else hipLaunchKernelGGL(transpose_kernel<float>, dim3((uint32_t)blocks), dim3(256), 0, stream, prm);
	return hipGetLastError() == hipSuccess ? 0 : 4039;
}

// ---- the library's own streaming copy (extension vkfftMI355XStreamCopy): 16 bytes per lane, four transfers in flight per lane, non-temporal hint on
// both sides, persistent grid — the practical HBM ceiling the roofline fractions of bench.py are quoted against next to torch's copy kernel
struct alignas(16) Copy16 { uint32_t x, y, z, w; };
// a workgroup moves contiguous 32 KiB blocks: eight 16-byte transfers per lane in flight, lanes along the block (the access shape of the FFT kernels' tiles and of
// tools/probe3.hip's k_copy, which measured 5.5-5.7 TB/s; a grid-stride form with four scattered transfers per lane measured 4.4-4.8)
__global__ void __launch_bounds__(256) stream_copy_kernel(const Copy16* __restrict__ src, Copy16* __restrict__ dst, const uint64_t n16) {
	constexpr uint64_t BLK16 = 256 * 8; // 16-byte units per block
	const uint64_t nBlocks = n16 / BLK16;
#if defined(VKFFT_HOSTEMU)
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i];
#else
	for (uint64_t i = blockIdx.x; i < nBlocks; i += gridDim.x) {
		const GBuf gs = make_gbuf(src + i * BLK16), gd = make_gbuf(dst + i * BLK16);
		vk_u32x4 v[8];
#pragma unroll
		for (int j = 0; j < 8; j++) v[j] = __builtin_amdgcn_raw_buffer_load_b128(gs.r, threadIdx.x * 16u + j * 4096u, 0, 0);
#pragma unroll
		for (int j = 0; j < 8; j++) __builtin_amdgcn_raw_buffer_store_b128(v[j], gd.r, threadIdx.x * 16u + j * 4096u, 0, 0);
	}
	for (uint64_t i = nBlocks * BLK16 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i]; // tail
#endif
}
int launch_stream_copy(void* dst, const void* src, uint64_t bytes, hipStream_t stream) {
	if (bytes % 16 || ((uintptr_t)dst | (uintptr_t)src) % 16) return 4039;
	if (!bytes) return 0;
	const uint64_t n16 = bytes / 16;
	uint64_t grid = (uint64_t)pow2_num_cus_aux() * 8u;
	if (grid > (n16 + 2047) / 2048) grid = (n16 + 2047) / 2048;
	hipLaunchKernelGGL(stream_copy_kernel, dim3((uint32_t)grid), dim3(256), 0, stream, (const Copy16*)src, (Copy16*)dst, n16);
	return hipGetLastError() == hipSuccess ? 0 : 4039;
}

} // namespace vkfft_mi355x
