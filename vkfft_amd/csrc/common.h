// Shared host/device definitions of the MI355X FFT engine: complex type, fast integer division,
// and the per-launch parameter blocks ("pass descriptors") the planner fills and the kernels read.
//
// A *pass* is one kernel launch = one trip of the data through HBM.  It runs G independent
// sub-FFTs of length L.  The reference's equivalent is one "axis upload" (VkFFTAxis,
// vkFFT_Structs.h:1037; built by VkFFTPlanAxis, vkFFT_Plan_FFT.h:33), but where the reference
// bakes these numbers into a generated source string, here they are plain kernel arguments of
// ahead-of-time compiled kernels.
#pragma once
#include <stdint.h>

#if defined(VKFFT_HOSTEMU)
#include "hostemu_runtime.h" // tests/hostemu: CPU SIMT emulation used ONLY by the CPU test-suite
#else
#include <hip/hip_runtime.h>
#define VKFFT_DYN_SMEM(var) extern __shared__ __attribute__((aligned(16))) char var[];
#endif

namespace vkfft_mi355x {

template <typename T> struct alignas(2 * sizeof(T)) cx {
	T x, y;
};
using cf = cx<float>;
using cd = cx<double>;

template <typename T> __host__ __device__ inline cx<T> cmul(cx<T> a, cx<T> b) {
	return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
template <typename T> __host__ __device__ inline cx<T> cmulc(cx<T> a, cx<T> b) { // a * conj(b)
	return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y};
}
template <typename T> __host__ __device__ inline cx<T> cadd(cx<T> a, cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> __host__ __device__ inline cx<T> csub(cx<T> a, cx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T> __host__ __device__ inline cx<T> cswap(cx<T> a) { return {a.y, a.x}; }
template <typename T> __host__ __device__ inline cx<T> cconj(cx<T> a) { return {a.x, -a.y}; }
template <typename T> __host__ __device__ inline cx<T> cscale(cx<T> a, T s) { return {a.x * s, a.y * s}; }
// multiply by -i (forward quarter turn) / +i
template <typename T> __host__ __device__ inline cx<T> cmul_mi(cx<T> a) { return {a.y, -a.x}; }
template <typename T> __host__ __device__ inline cx<T> cmul_pi(cx<T> a) { return {-a.y, a.x}; }

// Division by a launch-invariant divisor for n < 2^24 (all in-workgroup indices are far below).
struct FastDiv {
	uint32_t d;
	float rcp;
	__host__ __device__ inline void divmod(uint32_t n, uint32_t& q, uint32_t& r) const {
		uint32_t qq = (uint32_t)((float)n * rcp);
		int32_t rr = (int32_t)(n - qq * d);
		if (rr < 0) { qq--; rr += (int32_t)d; }
		else if (rr >= (int32_t)d) { qq++; rr -= (int32_t)d; }
		q = qq; r = (uint32_t)rr;
	}
};
inline FastDiv make_fastdiv(uint32_t d) { FastDiv f; f.d = d ? d : 1; f.rcp = 1.0f / (float)f.d; return f; }

constexpr int kMaxStages = 16;

// pre/post operations fused into a pass (values of PassParams::preOp / postOp)
enum : uint32_t {
	OP_NONE = 0,
	OP_TWIDDLE_4STEP = 1,   // post: x[j] *= exp(-2*pi*i * j * col / fsN)      (vkFFT_4step.h:31)
	OP_R2C_EVEN_POST = 2,   // post: half-length complex FFT -> N/2+1 Hermitian outputs   (vkFFT_R2C_even_decomposition.h:181)
	OP_C2R_EVEN_PRE = 3,    // pre : Hermitian half -> packed half-length complex input
	OP_R2C_FULL = 4,        // pre : real -> (x,0); post: store first N/2+1               (vkFFT_R2C.h:27, "callback" form)
	OP_C2R_FULL = 5,        // pre : Hermitian half expanded to full length; post: store real part
	OP_DCT2_PRE = 6, OP_DCT2_POST = 7,   // vkFFT_R2R.h:193,784
	OP_DCT3_PRE = 8, OP_DCT3_POST = 9,
	OP_DCT1_PRE = 10, OP_DCT1_POST = 11, // vkFFT_R2R.h:28
	OP_DCT4_PRE = 12, OP_DCT4_POST = 13, // vkFFT_R2R.h:368,861
	OP_DST1_PRE = 14, OP_DST1_POST = 15,
	OP_DST2_PRE = 16, OP_DST2_POST = 17,
	OP_DST3_PRE = 18, OP_DST3_POST = 19,
	OP_DST4_PRE = 20, OP_DST4_POST = 21,
	OP_BLUESTEIN_PRE = 22,  // pre : x[n] *= conj(chirp[n]), zero-pad to L            (vkFFT_Bluestein.h:32)
	OP_BLUESTEIN_MID = 23,  // mid : pointwise * FFT(chirp), then the stage list runs again as inverse (vkFFT_Bluestein.h:201)
	OP_BLUESTEIN_POST = 24, // post: x[k] *= conj(chirp[k]) for k < N
	OP_MUL_LUT = 25,        // post: pointwise multiply by aux[j] (multi-pass Bluestein)
	// DCT/DST-II and -III of even length N through ONE complex FFT of length N/2 (Makhoul permutation + the even R2C /
	// C2R split fused with the quarter-wave twiddle): half the LDS footprint and arithmetic of the full-length form
	OP_DCT2H_PRE = 26, OP_DCT2H_POST = 27,
	OP_DCT3H_PRE = 28, OP_DCT3H_POST = 29,
	OP_DST2H_PRE = 30, OP_DST2H_POST = 31,
	OP_DST3H_PRE = 32, OP_DST3H_POST = 33,
	// DCT-I of length N through ONE complex FFT of length N-1: the even extension (period 2N-2) is a real sequence, so its transform is the even
	// R2C split of the packed pairs e[2n] + i e[2n+1] — half the points of the full-length form (vkFFT_R2R.h:28 runs the complex FFT of 2N-2)
	OP_DCT1H_PRE = 36, OP_DCT1H_POST = 37,
	OP_FOURSTEP_INV_COL_PRE = 35, // pre : column layout, swap, Four-Step twiddle (middle pass of a three-factor inverse run backwards)
	OP_FOURSTEP_INV_PRE = 34, // pre : rows of the transposed Four-Step scratch, swap, Four-Step twiddle (first pass of the inverse run backwards)
};

struct StageDesc {
	uint32_t radix;   // 2,3,4,5,7,8,11,13,16 or a Rader prime
	uint32_t S;       // product of the radices of earlier stages (Stockham stride)
	uint32_t lutOff;  // offset (complex elements) of this stage's twiddle run in PassParams::lut
	uint32_t kind;    // 0 radix butterfly, 1 Rader direct-multiplication, 2 Rader FFT-convolution
	uint32_t aux0, aux1; // Rader: offsets of generator tables / convolution kernel
};

// One dimension of the sub-FFT enumeration: count + element strides on the input and output side.
struct BatchDim {
	uint32_t count;
	int64_t inStride, outStride;
};

// FFT-convolution Rader stage (one prime per pass): sub-FFT of length P-1 run over the butterflies as columns
struct RaderDesc {
	uint32_t P;          // the prime
	uint32_t nSub;       // radix stages of the length P-1 sub-FFT
	StageDesc sub[8];    // lutOff relative to subLutOff
	uint32_t subLutOff;  // sub-FFT stage twiddles, complex elements from PassParams::lut
	uint32_t bhatOff;    // FFT(b)/(P-1), b_q = exp(-2 pi i g^-q / P): complex elements from PassParams::lut
	uint32_t gpowOff;    // uint32 tables (from PassParams::rader): g^q mod P, q < P-1
	uint32_t ginvOff;    // g^-m mod P, m < P-1
	uint32_t tailElems;  // extra LDS elements holding x_0 / X_0 of every butterfly
	FastDiv divU;        // butterflies per workgroup (columns of the sub-FFT)
	FastDiv divSubNb[8], divSubS[8];
};

constexpr uint32_t kTmOn = 1u, kTmTwo = 2u, kTmSplit = 4u, kTmRowB = 8u, kTmCplx = 16u; // PassParams::tmPreFlags / tmPostFlags (kernel_tmaps.h)
struct PassParams {
	const void* in;
	void* out;
	const void* lut;     // stage twiddles, cx<T>
	const void* aux;     // op-specific table (4-step two-level LUT, R2C/DCT twiddles, chirp, ...)
	const void* aux2;    // second op-specific table (Bluestein FFT(chirp), DCT-IV post twiddles)
	const void* aux3;    // pre-op table when aux is taken by the Four-Step LUT (multi-pass Bluestein chirp)
	const void* rader;   // uint32 tables of the FFT-Rader stage
	RaderDesc rd;
	uint32_t L;          // sub-FFT length computed by the stages
	uint32_t nStages;
	StageDesc st[kMaxStages];
	// element j of sub-FFT (g0,g1,g2) lives at  j*inStrideJ + g0*dim[0].inStride + g1*dim[1].inStride + g2*dim[2].inStride
	// (units: complex elements, or real scalars when the pre/post op reads/writes real data)
	int64_t inStrideJ, outStrideJ;
	BatchDim dim[3];     // dim[0] is the tiled one: a workgroup takes T consecutive g0
	uint32_t T;          // sub-FFTs per workgroup (power of two)
	uint32_t logT;
	uint32_t colMode;    // load side: 0 lanes run along j (unit inStrideJ), 1 lanes run along g0 (unit dim[0] stride)
	uint32_t colModeOut; // same for the store side
	uint32_t padShift;   // LDS index a -> a + (a >> padShift); 31 = no padding
	uint32_t Tp;         // LDS pitch between consecutive a (T rounded up to odd, 1 when T==1)
	uint32_t swapIn;     // swap re/im of the FFT input  } inverse transform = swap . forward . swap;
	uint32_t swapOut;    // swap re/im of the FFT output } multi-pass plans set swapIn on the first, swapOut on the last pass
	uint32_t preOp, midOp, postOp;
	uint32_t bluesteinSwapIn, bluesteinSwapOut;
	uint32_t inLen, outLen; // elements gathered / stored per sub-FFT (differ from L for real transforms, Bluestein)
	uint32_t opN;        // logical transform size of the pre/post op (e.g. real length N of R2C / DCT)
	uint32_t blueN;      // Bluestein-wrapped / multi-pass real transforms: length of the embedding sequence (opN stays the real N)
	// multi-pass real transforms: the pre-map of the first pass / post-map of the last pass address the ROW by the natural FFT index
	// n = g0*opStride0 + g1*opStride1 + j*opStrideJ; natDimMask bit i = grid dim i is part of that index (not of the row's address)
	uint32_t preNat, postNat, natDimMask, natOutLen;
	uint32_t opStrideJ, opStride0, opStride1; // position-indexed ops (Bluestein chirp, pointwise LUT): natural index = j*opStrideJ + g0*opStride0 + g1*opStride1
	uint32_t fsN;        // 4-step: twiddle exponent denominator (product of all pass lengths of this decomposition level)
	uint32_t fsLoBits;   // 4-step two-level LUT: aux = 2^fsLoBits low entries followed by the high entries
	FastDiv fsColDiv;    // 4-step: column index used in the twiddle = g0 / fsColDiv
	uint32_t fsColFromDim1; // 4-step along a strided axis: the twiddle's column index is g1 (dim[1]) instead
	double scale;        // multiplied into the output (1/N normalisation); 1.0 = off
	FastDiv divL, divOutLen;
	FastDiv divNb[kMaxStages]; // L / radix per stage
	FastDiv divS[kMaxStages];
	uint32_t ldsElems;   // elements per LDS buffer (two buffers are used)
	uint32_t tilesPerG0; // ceil(dim[0].count / T)
	uint32_t reverseTiles; // 1: workgroup i works on tile (grid - 1 - i): inverse plans sweep the buffer back to front
	uint32_t inElemBytes, outElemBytes; // bytes per global element on each side (real scalar or complex)
	// zero padding (VkFFTConfiguration::performZeropadding, vkFFT_Zeropad.h:28): elements [padInL, padInL + padInN) of every sub-FFT are taken as zero and NOT
	// read; elements [padOutL, padOutL + padOutN) of its output are NOT written.  Units: elements of the respective side; N = 0: off
	uint32_t padInL, padInN, padOutL, padOutN;
	uint32_t colMerge;   // mixconv_kernel column tiles: the tile index runs over dim[0] x dim[1] (column g of a tile = (g % dim[0].count, g / dim[0].count)): no partly
	                     // filled tiles when dim[0].count is not a multiple of the tile width (prime planes); tilesPerG0 then counts the tiles of both
	uint32_t raderM;     // mixrad_kernel (kernel_mixrad.h): cofactor M of a row of M * P points, P the Rader prime of the instance (0: not that kernel; 1: the prime's own rows)
	uint32_t raderA;     // ... its split M = raderA * B into the two column steps (1: one step; 0: M = P, the column transform is the prime's own convolution)
	uint32_t raderAligned; // ... 1: the thread groups of its convolution are laid out inside wavefronts (64 / TPF groups each, wave-level ordering); 0: densely (workgroup barriers)
	uint32_t pairRows;   // instance kernels between the generic maps (OPS = 1): two real rows per complex transform (kernel_generic.h ops_rows_in / ops_rows_out)
	// table-driven maps of the real transforms (kernel_tmaps.h): per FFT input position / per spectrum index { byte offsets o1, o2 } and { complex c1, c2 };
	// flags: kTmOn | kTmTwo (second term of the pre-map is used) | kTmSplit | kTmRowB | kTmCplx (post-map forms).  0: the maps of kernel_generic.h
	const void* tmPre; const void* tmPost;
	uint32_t tmPreFlags, tmPostFlags;
	uint32_t bigSpan;    // pow2_col_kernel: the tile spans 2 GiB or more on one side: 64-bit per-lane addresses instead of a buffer resource per tile
	// merged convolution along this axis (pow2_col_blue_kernel MODE 6; reference vkFFT_Convolution.h:125): convCf coordinate systems convSysStride elements
	// apart are transformed, multiplied per frequency by the convM x convM kernel matrix (convM <= 1: every coordinate by its own kernel component) and
	// transformed back.  aux2 = kernel spectra (same layout as one system, convSysStride apart); convKerStride1/2 = kernel strides of dim[1] / dim[2]
	// (0 for a batch dimension, which the kernel does not follow)
	uint32_t convM, convCf, convSymmetric, convConj;
	int64_t convSysStride, convKerStride1, convKerStride2;
	int64_t convKerStrideJ, convKerSysStride; // kernel strides along the axis and between kernel systems (they differ from the data's when the data sits in the Four-Step scratch)
};

// index of kernel component (j, l) among the systems of one convolution kernel.  symmetricKernel: the packed upper triangle in the DOCUMENTED order
// xx, xy, xz, yy, yz, zz (API guide, "symmetricKernel"): row a <= b starts after a*m - a*(a-1)/2 entries.  The reference's generated code uses
// a*m - a*a + b (vkFFT_Convolution.h:352-358), which is the same for m = 2 but collides for m = 3 ((1,2) and (2,2) both give 4, slot 5 is never read);
// this library follows the documented layout
__host__ __device__ inline uint32_t conv_kernel_index(uint32_t j, uint32_t l, uint32_t m, bool symmetric) {
	if (!symmetric) return j * m + l;
	const uint32_t a = l < j ? l : j, b = l < j ? j : l;
	return a * m - a * (a - 1u) / 2u - a + b;
}

// ---- fused Four-Step launch (kernel_pow2_fused.h): both passes of a two-factor transform in one persistent kernel ----
constexpr uint32_t kFusedCtrTicket = 0, kFusedCtrExit = 256, kFusedCtrDone = 320; // uint32 indices into FusedParams::ctr (ticket counter of queue q at 32*q)
constexpr uint32_t kFusedMaxQueues = 8;

struct FusedParams {
	const void* in; void* out; void* scratch; uint32_t* ctr;
	const void* lutA; const void* lutB; const void* tw4; // stage twiddles of the two factors, two-level Four-Step table
	const void* rowTab;   // packed-pair kernels (kernel_pow2_pk.h): n0 16-byte entries (1, Re w_N^k, 0, Im w_N^k), k < n0 (fp32 only, else nullptr)
	int64_t inBatchStride, outBatchStride; // complex elements between consecutive transforms
	uint32_t fsLoBits;
	uint32_t n0, n1, batch;
	uint32_t logG;        // transforms per chunk = 2^logG
	uint32_t logTiles;    // tiles per transform and phase = 2^logTiles
	uint32_t C, NS, D;    // chunks; ring slots per queue; lag (in slots) between a chunk's A and B tiles; NS > D >= 1
	uint32_t Q;           // work queues (1, or one per XCD): chunk c belongs to queue c % Q
	uint32_t swapIn, swapOut, reverse;
	double scale;
	uint32_t tiles, tpc;  // mix_fused_kernel (kernel_mix_fused.h): tickets per transform = max(tiles of A, tiles of B), tickets per chunk = tiles << logG (neither a power of two)
	unsigned long long* prof; // development only (VKFFT_MI355X_FUSED_PROFILE): per-workgroup cycle sums of the tile phases, else nullptr
};

// element-wise product of a convolution plan (kernels_aux.hip): spectra of `coordinates` systems per batch, `systemStride` complex
// elements apart; kernel k of `numKernels` holds `kernelSystems` component spectra with the same stride; output of kernel f goes to
// batch f * batches + b
struct ConvParams {
	void* data; const void* kernel;
	uint64_t systemStride;
	uint32_t matrix, coordinates, batches, numKernels, kernelSystems;
	uint32_t symmetric, conjugate, crossPower;
};
// zero padding (kernels_aux.hip): a slab [left, right) along `axis` of every system; offsets in elements of `words` 32-bit words
struct ZeroParams {
	void* base;
	uint32_t size[4]; uint64_t stride[4];
	uint64_t systemStride; uint32_t systems;
	uint32_t axis, left, right, words;
};

} // namespace vkfft_mi355x
