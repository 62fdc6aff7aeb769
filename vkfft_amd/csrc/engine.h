// Host-side engine interface: the planner turns a transform description into a list of passes; the
// executor enqueues them.  Replaces the reference's VkFFTScheduler / VkFFTPlanAxis / VkFFT_DispatchPlan
// (vkFFT_Scheduler.h:2223, vkFFT_Plan_FFT.h:33, vkFFT_DispatchPlan.h:26) with a design that selects among
// ahead-of-time compiled kernels instead of generating source.
#pragma once
#include "common.h"
#include <vector>
#include <string>
#include <cstddef>

namespace vkfft_mi355x {

// ROLE_TEMP2: a second scratch region behind ROLE_TEMP in the same allocation, for plans that wrap an inner plan which uses ROLE_TEMP itself
enum BufRole : int { ROLE_BUFFER = 0, ROLE_TEMP = 1, ROLE_INPUT = 2, ROLE_OUTPUT = 3, ROLE_TEMP2 = 4 };
enum KernelKind : int { KERNEL_GENERIC = 0, KERNEL_POW2_ROW = 1, KERNEL_POW2_COL = 2, KERNEL_R2C_PAIR = 3, KERNEL_MIXED_ROW = 5, KERNEL_OPFFT = 6, KERNEL_POW2_BLUE = 7, KERNEL_POW2_COL_BLUE = 8, KERNEL_POW2_BLUE_R2R = 9, KERNEL_POW2_FUSED = 10, KERNEL_TRANSPOSE = 11, KERNEL_REAL_MAP = 12, KERNEL_MIXCONV = 13, KERNEL_MIX_FUSED = 14 };

struct HostDim {
	uint64_t count;
	int64_t inStride, outStride;
};

// One kernel launch (possibly repeated over host-side outer dimensions).
struct PassPlan {
	PassParams prm;       // pointers are filled at launch
	int inRole = ROLE_BUFFER, outRole = ROLE_BUFFER;
	int64_t inOffset = 0, outOffset = 0; // element offsets (units of the pass's element type) added to the role base
	int inElemBytes = 8, outElemBytes = 8; // bytes per element on each side (real vs complex, fp32 vs fp64)
	bool dp = false;
	int kernel = KERNEL_GENERIC;
	int variant = 0;      // index into the fast-kernel table
	uint32_t threads = 256;
	size_t ldsBytes = 0;
	std::vector<HostDim> hostLoop; // outer dims iterated on the host (rare: >3 non-collapsible batch dims)
	// arena offsets (bytes) of the tables, SIZE_MAX = none
	size_t lutOff = (size_t)-1, auxOff = (size_t)-1, aux2Off = (size_t)-1, aux3Off = (size_t)-1, raderOff = (size_t)-1;
	size_t tmPreOff = (size_t)-1, tmPostOff = (size_t)-1; // table-driven maps (kernel_tmaps.h)
	// fused Four-Step launch (KERNEL_POW2_FUSED): parameter block (pointers bound at launch) and its extra arena offsets
	FusedParams fused = {};
	size_t fusedLutBOff = (size_t)-1, fusedCtrOff = (size_t)-1, fusedRowTabOff = (size_t)-1;
	int fusedWgPerCu = 0; // 0: what the occupancy query reports
	bool auxIsKernel = false; // merged convolution pass: aux2 is bound to the caller's kernel buffer at launch (LaunchBuffers::kernel)
	std::string label;
};

struct DirectionPlan {
	std::vector<PassPlan> passes;
	std::vector<unsigned char> arena;  // host image of every LUT of this direction
	void* dArena = nullptr;            // device copy
	uint64_t tempBytes = 0;            // scratch this direction needs in ROLE_TEMP (0: none)
	uint64_t temp2Bytes = 0;           // ... and in ROLE_TEMP2, which starts temp2Offset() bytes into the same allocation
	uint64_t temp2Offset() const { return (tempBytes + 255ull) & ~255ull; }
	uint64_t totalTemp() const { return temp2Bytes ? temp2Offset() + temp2Bytes : tempBytes; }
	uint32_t uploadsPerAxis[4] = {0, 0, 0, 0};
	uint32_t bigSequenceEvenR2C = 0;
	uint64_t axisSplit[4][4] = {};
	uint32_t padFallbackMask = 0;      // zero-padded axes whose passes cannot skip the padded range themselves: the range is written with zeros ahead of the transform
};

// ---- transform description handed to the planner (derived from VkFFTConfiguration) -------------------
struct TransformDesc {
	int fftDim = 1;
	uint64_t size[4] = {1, 1, 1, 1};
	uint64_t batch = 1;          // numberBatches * coordinateFeatures folded
	bool dp = false;
	int kind = 0;                // 0 C2C, 1 R2C/C2R, 2 DCT, 3 DST
	int r2rType = 0;             // 1..4
	bool inverse = false;
	bool normalize = false;
	bool omit[4] = {false, false, false, false};
	bool reorder = true;
	// element strides of the main buffer (complex elements for C2C, see planner for real kinds)
	uint64_t bufStride[5] = {0, 0, 0, 0, 0};  // [0] = pitch of axis 1 (elements between consecutive rows), ... [fftDim-1] = batch pitch
	bool inFormatted = false, outFormatted = false;
	uint64_t inStride[5] = {0, 0, 0, 0, 0}, outStride[5] = {0, 0, 0, 0, 0};
	bool inverseReturnToInput = false;
	uint64_t maxLds = 160 * 1024;
	uint64_t forceBluesteinSize = 0;
	int fixMaxRadixBluestein = 0;
	uint64_t raderMultMin = 17, raderMultMax = 128;
	uint64_t userTempBytes = 0;  // >0: temp supplied by the caller with this size
	bool disableFastKernels = false;
	// zero padding (performZeropadding / fft_zeropad_left / fft_zeropad_right / frequencyZeroPadding): range [padL, padR) of an axis, padR == 0: none
	uint64_t padL[4] = {0, 0, 0, 0}, padR[4] = {0, 0, 0, 0};
	bool padFrequency = false;
	// fused Four-Step (kernel_pow2_fused.h); the numeric fields are tuning knobs, 0 = planner default
	bool fused = true;
	int fusedMode = 2;
	uint64_t fusedChunkBytes = 0;
	uint32_t fusedLag = 0, fusedRing = 0, fusedWgPerCu = 0, fusedQueues = 0, fusedMarginPct = 0;
};

// returns 0 or a VkFFTResult code
int build_direction_plan(const TransformDesc& d, DirectionPlan& out);

// Merged convolution along the LAST axis of `d` (a strided, power-of-two axis of 64 .. 1024 points): one pass = forward transform of every coordinate
// system, kernel matrix product per frequency, inverse transform (pow2_col_blue_kernel MODE 6; reference vkFFT_Convolution.h:125, vkFFT_RunApp.h:235-345).
// d.batch = numberBatches (NOT folded with the coordinates); returns 3002 when no such pass exists for the shape (the caller keeps separate passes)
struct ConvAxisDesc { uint32_t matrix = 1, coordinates = 1, symmetric = 0, conjugate = 0; double scale = 1.0; uint64_t kernelSystems = 1; };
int build_conv_axis_plan(const TransformDesc& d, const ConvAxisDesc& c, DirectionPlan& out);

// launchers (kernels.hip)
struct LaunchBuffers {
	void* base[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // by BufRole
	const void* kernel = nullptr; // convolution kernel spectra (merged convolution pass)
};
int launch_pass(const PassPlan& pp, const PassParams& prm, hipStream_t stream);
// The caller's streams (VkFFTConfiguration::stream / num_streams).  Everything is ordered on s[0]; a pass that the host has to split
// into independent sub-launches deals them round-robin over all streams (reference: vkFFT_DispatchPlan.h:288-295 does the same with
// the blocks of an oversized dispatch) and joins them back into s[0] through the events (stream-side waits, no host synchronisation).
struct StreamSet {
	uint32_t n = 1;
	hipStream_t s[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};
// sweep: the application's zig-zag state: every launch walks the buffer opposite to the previous one; nullptr = always front to back
int execute_direction(const DirectionPlan& plan, const LaunchBuffers& bufs, const StreamSet& streams, uint32_t* sweep = nullptr);

// fast-kernel registry queries used by the planner
int launch_pow2(const PassPlan& pp, const PassParams& prm, hipStream_t stream);
int launch_pow2_blue(const PassPlan& pp, const PassParams& prm, hipStream_t stream);
int launch_pow2_col_blue(const PassPlan& pp, const PassParams& prm, hipStream_t stream);
bool pow2_row_lookup(uint32_t log2n, bool dp, int* variant, int bits[4], int* fpw, int* threads, bool padded = false); // padded: only kernels with zero-padding masks
bool pow2_col_lookup(uint32_t log2n, bool dp, int* variant, int bits[4], int* tc, int* threads);
bool pow2_col_blue_lookup(uint32_t log2l, bool dp, int mode, int* variant, int bits[4], int* tc, int* threads); // multi-pass Bluestein passes 1..3
bool pow2_blue_r2r_lookup(uint32_t log2m, bool dp, uint32_t pre, int* variant, int bits[4], int* fpw, int* threads); // Bluestein-wrapped DCT/DST/R2C
int launch_pow2_blue_r2r(const PassPlan& pp, const PassParams& prm, hipStream_t stream);
bool pow2_blue_lookup(uint32_t log2m, bool dp, int* variant, int bits[4], int* fpw, int* threads); // fused Bluestein on padded length 2^log2m
// fused Four-Step of 2^log2n = 2^la * 2^lb (kernels_fused.hip)
bool pow2_fused_lookup(uint32_t log2n, bool dp, int mode, int* variant, int* la, int* lb, int bitsA[4], int bitsB[4], int* tca, int* tcb, int* threads, int* wgPerCu);
int launch_pow2_fused(const PassPlan& pp, const FusedParams& prm, hipStream_t stream);
// the __global__ function behind a registry entry (vkfftMI355XDescribePlan: bench labels, rocprofv3 kernel names)
const char* pow2_fused_kernel_name(int variant);
// fused Four-Step of a non-power-of-two N = n0 * n1 (kernels_mixfused.hip, kernel_mix_fused.h)
bool mix_fused_lookup(uint64_t n, bool dp, int* variant, int* n0, int* n1, int radA[5], int radB[5], int* tca, int* tcb, int* threads, int* wgPerCu);
int launch_mix_fused(const PassPlan& pp, const FusedParams& prm, hipStream_t stream);
const char* pow2_row_kernel_name(int variant);
bool mixed_row_lookup(uint64_t n, bool dp, int* variant, int rad[5], int* fpw, int* threads);
int mixed_row_ops_fpw(int variant); // rows per workgroup of that variant's form between the maps of a real transform
int launch_mixed(const PassPlan& pp, const PassParams& prm, hipStream_t stream);
// one-kernel cyclic convolution (kernel_mixconv.h), unit-stride rows or (col) tiles of neighbouring columns of a strided axis.  rader: the instance of prime p (transform length p - 1); otherwise the Bluestein instance with
// the smallest padded length >= minLen.  *len = transform length
bool mixconv_lookup(bool rader, bool col, uint64_t pOrMinLen, bool dp, int* variant, uint64_t* len, int rad[5], int* fpw, int* threads);
int launch_mixconv(const PassPlan& pp, const PassParams& prm, hipStream_t stream);
bool mixrad_available(int variant); // the Rader row instance also exists as a stage of composite lengths (kernel_mixrad.h)
bool mixrad_geom(int variant, int* sp, int* lutn, int* groups, int* groupsDense); // ... its buffer pitch (elements), stage-twiddle count and thread groups of the wave-aligned layout (0: not offered)
// op-FFT family (kernel_opfft.h): pre/post are the DCT member of their family (DST variants share the instance)
bool opfft_lookup(uint64_t n, bool dp, bool col, bool trans, uint32_t pre, uint32_t post, int* variant, int rad[5], int* fpw, int* threads); // trans: column tile in, transposed (per-column contiguous) store out
int launch_opfft(const PassPlan& pp, const PassParams& prm, hipStream_t stream);

// convolution product and zero padding (kernels_aux.hip)
int launch_conv_pointwise(const ConvParams& p, bool dp, hipStream_t stream);
int launch_zero_slab(const ZeroParams& p, hipStream_t stream);
// tile transposition of a strided axis against its unit-stride companion (kernels_aux.hip): awkward strided axes run as rows of a dense scratch copy
int launch_transpose(const PassPlan& pp, const PassParams& prm, hipStream_t stream);
// pre / post map of a real transform as a pass of its own (kernels_aux.hip): coverage path around an arbitrary complex plan of the embedding length
int launch_real_map(const PassPlan& pp, const PassParams& prm, hipStream_t stream);

// misc
std::vector<uint32_t> factorize_radices(uint64_t n, bool* smooth);

} // namespace vkfft_mi355x
