// C-ABI entry points of libvkfft_mi355x.so — same names, argument meaning and error behaviour as the
// reference's header-only API:
//   initializeVkFFT  vkFFT_AppManagement/vkFFT_InitializeApp.h:1468 (+ setConfigurationVkFFT :428)
//   VkFFTAppend      vkFFT_AppManagement/vkFFT_RunApp.h:79
//   deleteVkFFT      vkFFT_AppManagement/vkFFT_DeleteApp.h:28
//   VkFFTGetVersion  vkFFT.h:109, getVkFFTErrorString vkFFT_Structs.h:479
// There is no CPU fallback anywhere in this library: without a usable HIP device plan creation fails.
#include "../../include/vkFFT.h"
#include "engine.h"
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <new>
#include <algorithm>

using namespace vkfft_mi355x;

namespace {

struct AppState {
	void* tempOwned = nullptr;     // temp buffer allocated by the library
	uint64_t tempOwnedBytes = 0;
	hipEvent_t* events = nullptr;
	uint32_t numEvents = 0;
	uint32_t sweep = 0;            // zig-zag state: direction of the next launch's tile sweep
	bool sweepEnabled = true;
	// convolution application (performConvolution): forward transform of the input, element-wise product with the kernel spectra,
	// inverse transform of numberKernels results — the reference merges the three into the last axis (vkFFT_RunApp.h:235-345); the
	// results are the same
	VkFFTApplication* convFwd = nullptr;
	VkFFTApplication* convInv = nullptr;
	VkFFTPlan* convMid = nullptr;  // merged form (strided power-of-two last axis): ONE pass does that axis forward, the kernel product and the axis backwards;
	                               // convFwd / convInv then omit the axis (planner.cpp build_conv_axis_plan)
	VkFFTApplication* convInvFull = nullptr; // merged form only: VkFFTAppend(app, 1) is a plain inverse over ALL axes; convInv lacks the merged one, so that
	VkFFTConfiguration convInvFullCfg = {};  // direction gets its own application, created at its first use from this configuration
	uint32_t zeroPadMask = 0;      // zero-padded axes whose plan cannot skip the range: it is written with zeros ahead of the transform that reads it
};

VkFFTResult hip_to_result(hipError_t e, VkFFTResult code) { return e == hipSuccess ? VKFFT_SUCCESS : code; }

void free_direction(VkFFTPlan* pl) {
	if (!pl) return;
	DirectionPlan* dp = (DirectionPlan*)pl->impl;
	if (dp) {
		if (dp->dArena) (void)hipFree(dp->dArena);
		delete dp;
	}
	free(pl);
}

VkFFTResult make_direction(VkFFTApplication* app, const TransformDesc& base, bool inverse, VkFFTPlan** outPlan) {
	VkFFTPlan* pl = (VkFFTPlan*)calloc(1, sizeof(VkFFTPlan));
	if (!pl) return VKFFT_ERROR_MALLOC_FAILED;
	DirectionPlan* dp = new (std::nothrow) DirectionPlan();
	if (!dp) { free(pl); return VKFFT_ERROR_MALLOC_FAILED; }
	pl->impl = dp;
	TransformDesc d = base;
	d.inverse = inverse;
	int r = build_direction_plan(d, *dp);
	if (r) { free_direction(pl); return (VkFFTResult)r; }
	if (!dp->arena.empty()) {
		if (hipMalloc(&dp->dArena, dp->arena.size()) != hipSuccess) { free_direction(pl); return VKFFT_ERROR_FAILED_TO_ALLOCATE; }
		if (hipMemcpy(dp->dArena, dp->arena.data(), dp->arena.size(), hipMemcpyHostToDevice) != hipSuccess) { free_direction(pl); return VKFFT_ERROR_FAILED_TO_COPY; }
	}
	for (uint64_t i = 0; i < app->configuration.FFTdim; i++) {
		pl->numAxisUploads[i] = dp->uploadsPerAxis[i];
		for (int k = 0; k < 4; k++) pl->axisSplit[i][k] = dp->axisSplit[i][k];
		for (uint64_t k = 0; k < VKFFT_MAX_FFT_DIMENSIONS; k++) pl->actualFFTSizePerAxis[i][k] = app->configuration.size[k];
		if (app->configuration.FFTdim == 1 && app->actualNumBatches > 1) pl->actualFFTSizePerAxis[i][1] = app->actualNumBatches;
		pl->actualPerformR2CPerAxis[i] = (i == 0) ? app->configuration.performR2C : 0;
		pl->bigSequenceEvenR2C = dp->bigSequenceEvenR2C;
	}
	*outPlan = pl;
	return VKFFT_SUCCESS;
}

VkFFTResult initialize_convolution(VkFFTApplication* app, const VkFFTConfiguration& in);
VkFFTResult append_convolution(VkFFTApplication* app, int inverse, VkFFTLaunchParams* lp);
VkFFTResult zero_padded_ranges(VkFFTApplication* app, bool inverse, void* base, hipStream_t stream);

} // namespace

namespace vkfft_mi355x { int launch_stream_copy(void* dst, const void* src, uint64_t bytes, hipStream_t stream); } // kernels_aux.hip

extern "C" {

VKFFT_API int VkFFTGetVersion(void) { return 10304; }

VKFFT_API void vkfftMI355XStructSizes(pfUINT out[4]) {
	out[0] = sizeof(VkFFTConfiguration);
	out[1] = sizeof(VkFFTLaunchParams);
	out[2] = sizeof(VkFFTPlan);
	out[3] = sizeof(VkFFTApplication);
}

static void describe_passes(const VkFFTPlan* pl, char* names, pfUINT cap, size_t& pos, int& launches) {
	static const char* kname[] = {"generic_pass_kernel", "pow2_row_kernel", "pow2_col_kernel", "r2c_even_pair_kernel", "?", "mixed_row_kernel", "opfft_kernel", "pow2_blue_kernel",
	                              "pow2_col_blue_kernel", "pow2_blue_r2r_kernel", "pow2_fused_kernel", "transpose_kernel", "real_map_kernel", "mixconv_kernel", "mix_fused_kernel"};
	if (!pl || !pl->impl) return;
	const DirectionPlan* dp = (const DirectionPlan*)pl->impl;
	for (const PassPlan& q : dp->passes) {
		uint64_t rep = 1;
		for (const HostDim& h : q.hostLoop) rep *= h.count;
		launches += (int)rep;
		if (!names || !cap) continue;
		const char* nm = kname[q.kernel >= 0 && q.kernel < 15 ? q.kernel : 4];
		if (q.kernel == KERNEL_POW2_FUSED) nm = pow2_fused_kernel_name(q.variant);       // (pow2_fused_kernel / _pipe_ / _pk_ / _pkh_)
		else if (q.kernel == KERNEL_POW2_ROW) nm = pow2_row_kernel_name(q.variant);      // (pow2_row_kernel / pow2_row_lean_kernel / pow2_row_lean_pk_kernel)
		else if (q.kernel == KERNEL_MIXCONV && q.prm.raderM) nm = "mixrad_kernel";        // (kernel_mixrad.h)
		int w = snprintf(names + pos, pos < cap ? (size_t)cap - pos : 0, "%s%s<%s>", pos ? "," : "", nm, q.dp ? "double" : "float");
		if (w > 0) pos = std::min<size_t>(pos + (size_t)w, (size_t)cap - 1);
	}
}

VKFFT_API int vkfftMI355XStreamCopy(void* dst, const void* src, pfUINT bytes, void* stream) {
	return vkfft_mi355x::launch_stream_copy(dst, src, (uint64_t)bytes, (hipStream_t)stream) == 0 ? VKFFT_SUCCESS : VKFFT_ERROR_FAILED_TO_LAUNCH_KERNEL;
}

VKFFT_API int vkfftMI355XDescribePlan(const VkFFTApplication* app, int inverse, char* names, pfUINT cap) {
	if (names && cap) names[0] = 0;
	if (!app) return 0;
	size_t pos = 0;
	int launches = 0;
	const AppState* st = (const AppState*)app->impl;
	if (st && st->convFwd) {
		// convolution application: what VkFFTAppend(app, -1) launches — forward transform (without the merged axis), the merged axis or the separate
		// element-wise product, inverse transform of the results
		if (inverse == 1) { // a plain inverse of the results (the merged form builds that application at its first use)
			const VkFFTApplication* inv = st->convMid ? st->convInvFull : st->convInv;
			if (inv) describe_passes(inv->localFFTPlan_inverse, names, cap, pos, launches);
			return launches;
		}
		describe_passes(st->convFwd->localFFTPlan, names, cap, pos, launches);
		if (st->convMid) describe_passes(st->convMid, names, cap, pos, launches);
		else {
			launches += 1;
			if (names && cap) { int w = snprintf(names + pos, pos < cap ? (size_t)cap - pos : 0, "%sconv_pointwise_kernel", pos ? "," : ""); if (w > 0) pos = std::min<size_t>(pos + (size_t)w, (size_t)cap - 1); }
		}
		if (st->convInv) describe_passes(st->convInv->localFFTPlan_inverse, names, cap, pos, launches);
		return launches;
	}
	describe_passes(inverse == 1 ? app->localFFTPlan_inverse : app->localFFTPlan, names, cap, pos, launches);
	return launches;
}

VKFFT_API void deleteVkFFT(VkFFTApplication* app) {
	if (!app) return;
	AppState* st = (AppState*)app->impl;
	if (st) {
		if (st->tempOwned) (void)hipFree(st->tempOwned);
		if (st->convFwd) { deleteVkFFT(st->convFwd); free(st->convFwd); }
		if (st->convInv) { deleteVkFFT(st->convInv); free(st->convInv); }
		if (st->convInvFull) { deleteVkFFT(st->convInvFull); free(st->convInvFull); }
		free_direction(st->convMid);
		if (app->saveApplicationString) free(app->saveApplicationString);
		if (st->events) {
			for (uint32_t i = 0; i < st->numEvents; i++) if (st->events[i]) (void)hipEventDestroy(st->events[i]);
			free(st->events);
		}
		delete st;
	}
	free_direction(app->localFFTPlan);
	free_direction(app->localFFTPlan_inverse);
	memset(app, 0, sizeof(VkFFTApplication));
}

VKFFT_API VkFFTResult initializeVkFFT(VkFFTApplication* app, VkFFTConfiguration in) {
	if (app == nullptr) return VKFFT_ERROR_EMPTY_app;
	{
		const unsigned char* t = (const unsigned char*)app;
		for (size_t i = 0; i < sizeof(VkFFTApplication); i++) if (t[i] != 0) return VKFFT_ERROR_NONZERO_APP_INITIALIZATION;
	}
	VkFFTConfiguration& c = app->configuration;
	// ---- validation in the reference's order (setConfigurationVkFFT) ---------------------------------
	if (in.FFTdim == 0) return VKFFT_ERROR_EMPTY_FFTdim;
	if (in.FFTdim > VKFFT_MAX_FFT_DIMENSIONS) return VKFFT_ERROR_FFTdim_GT_MAX_FFT_DIMENSIONS;
	if (in.device == nullptr) return VKFFT_ERROR_INVALID_DEVICE;
	if (in.size[0] == 0) return VKFFT_ERROR_EMPTY_size;
	if (in.saveApplicationToString && in.loadApplicationFromString) return VKFFT_ERROR_ENABLED_saveApplicationToString;
	if (in.loadApplicationFromString && in.loadApplicationString == nullptr) return VKFFT_ERROR_EMPTY_applicationString;
	if (in.useCustomBluesteinPaddingPattern && (!in.primeSizes || !in.paddedSizes)) return VKFFT_ERROR_EMPTY_useCustomBluesteinPaddingPattern_arrays;
	auto unsupported = [&](const char* what) {
		fprintf(stderr, "vkfft_mi355x: %s is outside the scope of this library (see DESIGN.md)\n", what);
		return VKFFT_ERROR_PLAN_NOT_INITIALIZED;
	};
	if (in.performConvolution) return initialize_convolution(app, in);
	if (in.bufferNum > 1 || in.inputBufferNum > 1 || in.outputBufferNum > 1 || in.tempBufferNum > 1 || in.kernelNum > 1) return unsupported("a buffer split over several allocations (bufferNum > 1)");
	bool zeroPad = false;
	for (pfUINT i = 0; i < in.FFTdim; i++) if (in.performZeropadding[i]) {
		zeroPad = true;
		if (in.fft_zeropad_left[i] > in.fft_zeropad_right[i] || in.fft_zeropad_right[i] > in.size[i]) return unsupported("a zero-padding range outside the axis");
	}
	if (in.halfPrecision || in.halfPrecisionMemoryOnly) return unsupported("half precision");
	if (in.quadDoubleDoublePrecision || in.quadDoubleDoublePrecisionDoubleMemory) return unsupported("double-double precision");
	if (in.doublePrecisionFloatMemory) return unsupported("doublePrecisionFloatMemory");
	if (in.performDCT > 4 || in.performDST > 4) return VKFFT_ERROR_UNSUPPORTED_FFT_LENGTH_R2R;
	if ((in.performDCT && in.performDST) || ((in.performDCT || in.performDST) && in.performR2C)) return VKFFT_ERROR_UNSUPPORTED_FFT_LENGTH_R2R;

	c = in;
	bool ldsCapped = false;
	// ---- device ----------------------------------------------------------------------------------------
	{
		int v = 0;
		hipDevice_t dev = *in.device;
		if (hipDeviceGetAttribute(&v, hipDeviceAttributeWarpSize, dev) != hipSuccess) { memset(app, 0, sizeof(*app)); return VKFFT_ERROR_FAILED_TO_GET_ATTRIBUTE; }
		c.warpSize = (pfUINT)v;
		if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxThreadsPerBlock, dev) != hipSuccess) { memset(app, 0, sizeof(*app)); return VKFFT_ERROR_FAILED_TO_GET_ATTRIBUTE; }
		c.maxThreadsNum = (pfUINT)v;
		c.maxComputeWorkGroupSize[0] = c.maxComputeWorkGroupSize[1] = c.maxComputeWorkGroupSize[2] = (pfUINT)v;
		int g[3] = {0, 0, 0};
		(void)hipDeviceGetAttribute(&g[0], hipDeviceAttributeMaxGridDimX, dev);
		(void)hipDeviceGetAttribute(&g[1], hipDeviceAttributeMaxGridDimY, dev);
		(void)hipDeviceGetAttribute(&g[2], hipDeviceAttributeMaxGridDimZ, dev);
		for (int i = 0; i < 3; i++) c.maxComputeWorkGroupCount[i] = (pfUINT)g[i];
		if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) { memset(app, 0, sizeof(*app)); return VKFFT_ERROR_FAILED_TO_GET_ATTRIBUTE; }
		// caller's cap on LDS per workgroup (a reference knob): clamped to what the device has and to a floor the planner's tiles need
		c.sharedMemorySize = in.sharedMemorySize ? std::min<pfUINT>(std::max<pfUINT>(in.sharedMemorySize, 16384), (pfUINT)v) : (pfUINT)v;
		ldsCapped = c.sharedMemorySize < (pfUINT)v;
		c.sharedMemorySizeStatic = c.sharedMemorySize;
		pfUINT p2 = 1; while (p2 * 2 <= c.sharedMemorySize) p2 *= 2;
		c.sharedMemorySizePow2 = p2;
		(void)hipDeviceGetAttribute(&v, hipDeviceAttributeComputeCapabilityMajor, dev); c.computeCapabilityMajor = (pfUINT)v;
		(void)hipDeviceGetAttribute(&v, hipDeviceAttributeComputeCapabilityMinor, dev); c.computeCapabilityMinor = (pfUINT)v;
		c.vendorID = 0x1002;
		if (!in.coalescedMemory) c.coalescedMemory = 256;
		if (!in.aimThreads) c.aimThreads = 256;
		if (!in.numSharedBanks) c.numSharedBanks = 64;
		c.useLUT = 1; c.useLUT_4step = 1;
		c.registerBoost = 1; c.registerBoost4Step = 1;
	}
	// ---- sizes, strides (reference: InitializeApp.h:984-1045) -------------------------------------------
	for (int i = 1; i < VKFFT_MAX_FFT_DIMENSIONS; i++) if (c.size[i] == 0) c.size[i] = 1;
	const bool r2c = in.performR2C != 0;
	if (in.bufferStride[0] == 0) c.bufferStride[0] = r2c ? c.size[0] / 2 + 1 : c.size[0];
	if (in.inputBufferStride[0] == 0) c.inputBufferStride[0] = (r2c && !in.isInputFormatted) ? c.size[0] + 2 : c.size[0];
	if (in.outputBufferStride[0] == 0) c.outputBufferStride[0] = (r2c && !in.isOutputFormatted) ? c.size[0] + 2 : c.size[0];
	for (int i = 1; i < VKFFT_MAX_FFT_DIMENSIONS; i++) {
		if (in.bufferStride[i] == 0) c.bufferStride[i] = c.bufferStride[i - 1] * c.size[i];
		if (in.inputBufferStride[i] == 0) c.inputBufferStride[i] = c.inputBufferStride[i - 1] * c.size[i];
		if (in.outputBufferStride[i] == 0) c.outputBufferStride[i] = c.outputBufferStride[i - 1] * c.size[i];
	}
	if (c.bufferNum == 0) c.bufferNum = 1;
	if (c.tempBufferNum == 0) c.tempBufferNum = 1;
	if (c.inputBufferNum == 0) c.inputBufferNum = 1;
	if (c.outputBufferNum == 0) c.outputBufferNum = 1;
	if (c.kernelNum == 0) c.kernelNum = 1;
	if (c.numberBatches == 0) c.numberBatches = 1;
	if (c.coordinateFeatures == 0) c.coordinateFeatures = 1;
	if (c.numberKernels == 0) c.numberKernels = 1;
	// disableReorderFourStep: the reference leaves the result of a multi-upload transform in an (unspecified) transposed order to save its last transposition
	// (vkFFT_InitializeApp.h:1312-1316, vkFFT_ReadWrite.h:1405-1424).  Here natural order costs nothing extra — the turn is part of the first pass — so the
	// request is not honoured, and the caller is TOLD so: the application's own copy of the configuration reports what is in effect (reorder on, flag cleared);
	// printMemoryLayout / VKFFT_MI355X_PRINT_PLAN print a line.  Natural order is one valid instance of the unspecified order for every forward -> pointwise ->
	// inverse pipeline whose operands all come from this library.
	if (in.disableReorderFourStep && (in.printMemoryLayout || getenv("VKFFT_MI355X_PRINT_PLAN")))
		fprintf(stderr, "[vkfft_mi355x] disableReorderFourStep requested: not applied, results stay in natural order (app->configuration.disableReorderFourStep reads back 0)\n");
	c.reorderFourStep = 1; c.disableReorderFourStep = 0;
	if (in.userTempBuffer) {
		if (in.tempBufferSize == nullptr) { memset(app, 0, sizeof(*app)); return VKFFT_ERROR_EMPTY_tempBufferSize; }
		if (in.tempBuffer == nullptr) { memset(app, 0, sizeof(*app)); return VKFFT_ERROR_EMPTY_tempBuffer; }
	}
	if (in.isInputFormatted && in.inverseReturnToInputBuffer == 0) { /* fine: inverse writes to buffer */ }
	for (pfUINT i = 0; i < c.FFTdim; i++) if (c.size[i] == 1) c.omitDimension[i] = 1;
	if (r2c && c.omitDimension[0] && c.size[0] > 1) { memset(app, 0, sizeof(*app)); return VKFFT_ERROR_UNSUPPORTED_FFT_OMIT; }
	// batch folding visible to callers (reference: vkFFT_Plan_FFT.h:55-61)
	app->actualNumBatches = c.numberBatches;
	const uint64_t totalBatch = c.numberBatches * c.coordinateFeatures;
	if (c.FFTdim == 1 && c.numberBatches > 1 && c.coordinateFeatures == 1) c.numberBatches = 1;
	app->firstAxis = 0; app->lastAxis = c.FFTdim - 1;
	for (pfUINT i = 0; i < c.FFTdim; i++) if (!c.omitDimension[i]) { app->firstAxis = i; break; }
	for (pfUINT i = c.FFTdim; i-- > 0;) if (!c.omitDimension[i]) { app->lastAxis = i; break; }

	// ---- transform description ---------------------------------------------------------------------------
	TransformDesc d;
	d.fftDim = (int)c.FFTdim;
	for (int i = 0; i < 4; i++) { d.size[i] = c.size[i]; d.omit[i] = c.omitDimension[i] != 0; }
	d.batch = totalBatch;
	d.dp = c.doublePrecision != 0;
	d.kind = r2c ? 1 : c.performDCT ? 2 : c.performDST ? 3 : 0;
	d.r2rType = (int)(c.performDCT ? c.performDCT : c.performDST);
	d.normalize = c.normalize != 0;
	d.reorder = c.reorderFourStep != 0;
	for (int i = 0; i < 4; i++) { d.bufStride[i] = c.bufferStride[i]; d.inStride[i] = c.inputBufferStride[i]; d.outStride[i] = c.outputBufferStride[i]; }
	d.inFormatted = c.isInputFormatted != 0; d.outFormatted = c.isOutputFormatted != 0;
	d.inverseReturnToInput = c.inverseReturnToInputBuffer != 0;
	d.maxLds = c.sharedMemorySize;
	d.forceBluesteinSize = c.forceBluesteinSequenceSize;
	d.fixMaxRadixBluestein = (int)c.fixMaxRadixBluestein;
	if (c.fixMaxRaderPrimeMult) d.raderMultMax = c.fixMaxRaderPrimeMult;
	if (c.userTempBuffer && c.tempBufferSize) d.userTempBytes = c.tempBufferSize[0];
	for (pfUINT i = 0; i < c.FFTdim && i < 4; i++) if (c.performZeropadding[i] && c.fft_zeropad_right[i] > c.fft_zeropad_left[i]) { d.padL[i] = c.fft_zeropad_left[i]; d.padR[i] = c.fft_zeropad_right[i]; }
	d.padFrequency = c.frequencyZeroPadding != 0;
	// fused Four-Step tuning knobs (experiments only; defaults are the planner's)
	if (const char* e = getenv("VKFFT_MI355X_FUSED")) d.fused = atoi(e) != 0;
	if (const char* e = getenv("VKFFT_MI355X_FUSED_MODE")) d.fusedMode = atoi(e);
	if (const char* e = getenv("VKFFT_MI355X_FUSED_CHUNK_KIB")) d.fusedChunkBytes = (uint64_t)atoll(e) << 10;
	if (const char* e = getenv("VKFFT_MI355X_FUSED_LAG")) d.fusedLag = (uint32_t)atoi(e);
	if (const char* e = getenv("VKFFT_MI355X_FUSED_RING")) d.fusedRing = (uint32_t)atoi(e);
	if (const char* e = getenv("VKFFT_MI355X_FUSED_WGS")) d.fusedWgPerCu = (uint32_t)atoi(e);
	if (const char* e = getenv("VKFFT_MI355X_FUSED_QUEUES")) d.fusedQueues = (uint32_t)atoi(e);
	if (const char* e = getenv("VKFFT_MI355X_FUSED_MARGIN")) d.fusedMarginPct = (uint32_t)atoi(e);
	if (const char* e = getenv("VKFFT_MI355X_GENERIC_ONLY")) d.disableFastKernels = atoi(e) != 0;
	if (ldsCapped) d.disableFastKernels = true; // the hand-specialised kernels have fixed LDS footprints (up to 155 KiB): under a cap the generic kernel, which sizes its tiles from maxLds, serves the plan

	AppState* st = new (std::nothrow) AppState();
	if (!st) { memset(app, 0, sizeof(*app)); return VKFFT_ERROR_MALLOC_FAILED; }
	app->impl = st;
	st->sweepEnabled = getenv("VKFFT_MI355X_NO_REVERSE") == nullptr;

	VkFFTResult res = VKFFT_SUCCESS;
	if (!c.makeForwardPlanOnly) {
		res = make_direction(app, d, true, &app->localFFTPlan_inverse);
		if (res != VKFFT_SUCCESS) { deleteVkFFT(app); return res; }
	}
	if (!c.makeInversePlanOnly) {
		res = make_direction(app, d, false, &app->localFFTPlan);
		if (res != VKFFT_SUCCESS) { deleteVkFFT(app); return res; }
	}
	if (zeroPad) {
		// Axes whose passes skip the padded range themselves need nothing here.  The others (plans of several passes, Bluestein, R2R maps) get the
		// range written with zeros ahead of the transform that reads it — only possible in a buffer this library may write to
		VkFFTPlan* reader = c.frequencyZeroPadding ? app->localFFTPlan_inverse : app->localFFTPlan;
		st->zeroPadMask = reader ? ((DirectionPlan*)reader->impl)->padFallbackMask : 0u;
		if (st->zeroPadMask && (c.frequencyZeroPadding ? c.isOutputFormatted : c.isInputFormatted)) {
			deleteVkFFT(app);
			return unsupported("zero-padding of a separate input buffer on an axis whose plan cannot skip the padded range (several passes, Bluestein, R2R)");
		}
	}
	if (c.printMemoryLayout || getenv("VKFFT_MI355X_PRINT_PLAN")) { // one line per launch: which kernel family serves it
		static const char* kname[] = {"generic", "pow2_row", "pow2_col", "r2c_pair", "?", "mixed_row", "opfft", "pow2_blue", "pow2_col_blue", "pow2_blue_r2r", "pow2_fused", "transpose", "real_map", "mixconv", "mix_fused"};
		for (int dir = 0; dir < 2; dir++) {
			VkFFTPlan* pl = dir ? app->localFFTPlan_inverse : app->localFFTPlan;
			if (!pl) continue;
			const DirectionPlan* dpn = (const DirectionPlan*)pl->impl;
			for (size_t i = 0; i < dpn->passes.size(); i++) {
				const PassPlan& q = dpn->passes[i];
				if (q.kernel == KERNEL_POW2_FUSED || q.kernel == KERNEL_MIX_FUSED) {
					fprintf(stderr, "[vkfft_mi355x] %s pass %zu: %-10s kernel=%s variant=%d N=%ux%u threads=%u chunk=%u transforms x %u chunks, %u queues, lag %u, ring %u (%.1f MiB)\n", dir ? "inverse" : "forward", i,
					        q.label.c_str(), q.kernel == KERNEL_MIX_FUSED ? "mix_fused" : "pow2_fused", q.variant, q.fused.n0, q.fused.n1, q.threads, 1u << q.fused.logG, q.fused.C, q.fused.Q, q.fused.D, q.fused.NS, (double)dpn->tempBytes / 1048576.0);
					continue;
				}
				fprintf(stderr, "[vkfft_mi355x] %s pass %zu: %-10s kernel=%s variant=%d L=%u threads=%u tile=%u%s grid=%llu\n", dir ? "inverse" : "forward", i, q.label.c_str(),
				        kname[q.kernel < 15 ? q.kernel : 4], q.variant, q.prm.L, q.threads, q.prm.T, q.prm.colMerge ? " (tiles over two dimensions)" : "",
				        (unsigned long long)q.prm.tilesPerG0 * (q.prm.colMerge ? 1u : q.prm.dim[1].count) * q.prm.dim[2].count);
			}
		}
	}
	// ---- scratch ------------------------------------------------------------------------------------------
	uint64_t need = 0;
	if (app->localFFTPlan) need = std::max<uint64_t>(need, ((DirectionPlan*)app->localFFTPlan->impl)->totalTemp());
	if (app->localFFTPlan_inverse) need = std::max<uint64_t>(need, ((DirectionPlan*)app->localFFTPlan_inverse->impl)->totalTemp());
	if (need && !c.userTempBuffer) {
		if (hipMalloc(&st->tempOwned, need) != hipSuccess) { deleteVkFFT(app); return VKFFT_ERROR_FAILED_TO_ALLOCATE; }
		st->tempOwnedBytes = need;
		c.allocateTempBuffer = 1;
	}
	// ---- saveApplicationToString: the reference serialises its run-time compiled kernels (vkFFT_InitializeApp.h:1468-1530) so that a later
	// initializeVkFFT can skip hiprtc.  Nothing is compiled at run time here; callers that write the blob to a file and hand it back
	// through loadApplicationString (sample_0_benchmark_VkFFT_single.cpp:169-199) get a small self-describing record.
	if (c.saveApplicationToString) {
		const size_t n = 64;
		char* blob = (char*)calloc(1, n);
		if (!blob) { deleteVkFFT(app); return VKFFT_ERROR_MALLOC_FAILED; }
		snprintf(blob, n, "vkfft_mi355x ahead-of-time kernels, API %d", VkFFTGetVersion());
		app->saveApplicationString = blob;
		app->applicationStringSize = n;
	}
	// ---- multi-stream events ------------------------------------------------------------------------------
	if (c.num_streams > 1 && c.stream) {
		st->events = (hipEvent_t*)calloc(c.num_streams, sizeof(hipEvent_t));
		if (!st->events) { deleteVkFFT(app); return VKFFT_ERROR_MALLOC_FAILED; }
		st->numEvents = (uint32_t)c.num_streams;
		for (uint32_t i = 0; i < st->numEvents; i++)
			if (hipEventCreateWithFlags(&st->events[i], hipEventDisableTiming) != hipSuccess) { deleteVkFFT(app); return VKFFT_ERROR_FAILED_TO_CREATE_EVENT; }
		c.stream_event = st->events;
	}
	return VKFFT_SUCCESS;
}

VKFFT_API VkFFTResult VkFFTAppend(VkFFTApplication* app, int inverse, VkFFTLaunchParams* lp) {
	if (app == nullptr) return VKFFT_ERROR_EMPTY_app;
	VkFFTConfiguration& c = app->configuration;
	AppState* st = (AppState*)app->impl;
	if (st == nullptr) return VKFFT_ERROR_PLAN_NOT_INITIALIZED;
	if (st->convFwd) return append_convolution(app, inverse, lp);
	VkFFTPlan* pl;
	if (inverse != 1) { // reference: anything but 1 is forward (vkFFT_RunApp.h:102-111)
		if (!app->localFFTPlan) return VKFFT_ERROR_ONLY_INVERSE_FFT_INITIALIZED;
		pl = app->localFFTPlan;
	} else {
		if (!app->localFFTPlan_inverse) return VKFFT_ERROR_ONLY_FORWARD_FFT_INITIALIZED;
		pl = app->localFFTPlan_inverse;
	}
	// launch-time buffer / offset override (reference: VkFFTCheckUpdateBufferSet, vkFFT_UpdateBuffers.h:628)
	if (lp) {
		if (lp->buffer) c.buffer = lp->buffer;
		if (lp->tempBuffer) c.tempBuffer = lp->tempBuffer;
		if (lp->inputBuffer) c.inputBuffer = lp->inputBuffer;
		if (lp->outputBuffer) c.outputBuffer = lp->outputBuffer;
		if (c.specifyOffsetsAtLaunch) {
			c.bufferOffset = lp->bufferOffset; c.tempBufferOffset = lp->tempBufferOffset;
			c.inputBufferOffset = lp->inputBufferOffset; c.outputBufferOffset = lp->outputBufferOffset;
		}
	}
	if (c.buffer == nullptr || c.buffer[0] == nullptr) return VKFFT_ERROR_EMPTY_buffer;
	if (c.isInputFormatted && (c.inputBuffer == nullptr || c.inputBuffer[0] == nullptr)) return VKFFT_ERROR_EMPTY_inputBuffer;
	if (c.isOutputFormatted && (c.outputBuffer == nullptr || c.outputBuffer[0] == nullptr)) return VKFFT_ERROR_EMPTY_outputBuffer;
	DirectionPlan* dp = (DirectionPlan*)pl->impl;
	LaunchBuffers lb;
	lb.base[ROLE_BUFFER] = (char*)c.buffer[0] + c.bufferOffset;
	if (c.isInputFormatted) lb.base[ROLE_INPUT] = (char*)c.inputBuffer[0] + c.inputBufferOffset;
	if (c.isOutputFormatted) lb.base[ROLE_OUTPUT] = (char*)c.outputBuffer[0] + c.outputBufferOffset;
	if (dp->totalTemp()) {
		if (c.userTempBuffer) {
			if (c.tempBuffer == nullptr || c.tempBuffer[0] == nullptr) return VKFFT_ERROR_EMPTY_tempBuffer;
			lb.base[ROLE_TEMP] = (char*)c.tempBuffer[0] + c.tempBufferOffset;
		} else lb.base[ROLE_TEMP] = st->tempOwned;
		lb.base[ROLE_TEMP2] = (char*)lb.base[ROLE_TEMP] + dp->temp2Offset();
	}
	StreamSet ss;
	if (c.stream && c.num_streams >= 1) {
		ss.n = (uint32_t)std::min<pfUINT>(c.num_streams, 8);
		for (uint32_t i = 0; i < ss.n; i++) ss.s[i] = c.stream[i];
		if (ss.n > 1) { for (uint32_t i = 0; i < ss.n; i++) ss.ev[i] = st->events[i]; }
	}
	if (st->zeroPadMask && (inverse == 1) == (c.frequencyZeroPadding != 0)) {
		VkFFTResult z = zero_padded_ranges(app, inverse == 1, lb.base[ROLE_BUFFER], ss.s[0]);
		if (z != VKFFT_SUCCESS) return z;
	}
	int r = execute_direction(*dp, lb, ss, st->sweepEnabled ? &st->sweep : nullptr);
	if (r) { fprintf(stderr, "vkfft_mi355x: kernel launch failed\n"); return (VkFFTResult)r; }
	return VKFFT_SUCCESS;
}

VKFFT_API const char* getVkFFTErrorString(VkFFTResult r) {
	switch (r) {
#define C(x) case x: return #x;
	C(VKFFT_SUCCESS) C(VKFFT_ERROR_MALLOC_FAILED) C(VKFFT_ERROR_INSUFFICIENT_CODE_BUFFER) C(VKFFT_ERROR_INSUFFICIENT_TEMP_BUFFER)
	C(VKFFT_ERROR_PLAN_NOT_INITIALIZED) C(VKFFT_ERROR_NULL_TEMP_PASSED) C(VKFFT_ERROR_MATH_FAILED) C(VKFFT_ERROR_FFTdim_GT_MAX_FFT_DIMENSIONS)
	C(VKFFT_ERROR_NONZERO_APP_INITIALIZATION) C(VKFFT_ERROR_INVALID_PHYSICAL_DEVICE) C(VKFFT_ERROR_INVALID_DEVICE) C(VKFFT_ERROR_INVALID_QUEUE)
	C(VKFFT_ERROR_INVALID_COMMAND_POOL) C(VKFFT_ERROR_INVALID_FENCE) C(VKFFT_ERROR_ONLY_FORWARD_FFT_INITIALIZED) C(VKFFT_ERROR_ONLY_INVERSE_FFT_INITIALIZED)
	C(VKFFT_ERROR_INVALID_CONTEXT) C(VKFFT_ERROR_INVALID_PLATFORM) C(VKFFT_ERROR_ENABLED_saveApplicationToString) C(VKFFT_ERROR_EMPTY_FILE)
	C(VKFFT_ERROR_EMPTY_FFTdim) C(VKFFT_ERROR_EMPTY_size) C(VKFFT_ERROR_EMPTY_bufferSize) C(VKFFT_ERROR_EMPTY_buffer) C(VKFFT_ERROR_EMPTY_tempBufferSize)
	C(VKFFT_ERROR_EMPTY_tempBuffer) C(VKFFT_ERROR_EMPTY_inputBufferSize) C(VKFFT_ERROR_EMPTY_inputBuffer) C(VKFFT_ERROR_EMPTY_outputBufferSize)
	C(VKFFT_ERROR_EMPTY_outputBuffer) C(VKFFT_ERROR_EMPTY_kernelSize) C(VKFFT_ERROR_EMPTY_kernel) C(VKFFT_ERROR_EMPTY_applicationString)
	C(VKFFT_ERROR_EMPTY_useCustomBluesteinPaddingPattern_arrays) C(VKFFT_ERROR_EMPTY_app) C(VKFFT_ERROR_INVALID_user_tempBuffer_too_small)
	C(VKFFT_ERROR_UNSUPPORTED_RADIX) C(VKFFT_ERROR_UNSUPPORTED_FFT_LENGTH) C(VKFFT_ERROR_UNSUPPORTED_FFT_LENGTH_R2C) C(VKFFT_ERROR_UNSUPPORTED_FFT_LENGTH_R2R)
	C(VKFFT_ERROR_UNSUPPORTED_FFT_OMIT) C(VKFFT_ERROR_FAILED_TO_ALLOCATE) C(VKFFT_ERROR_FAILED_TO_MAP_MEMORY) C(VKFFT_ERROR_FAILED_TO_ALLOCATE_COMMAND_BUFFERS)
	C(VKFFT_ERROR_FAILED_TO_BEGIN_COMMAND_BUFFER) C(VKFFT_ERROR_FAILED_TO_END_COMMAND_BUFFER) C(VKFFT_ERROR_FAILED_TO_SUBMIT_QUEUE)
	C(VKFFT_ERROR_FAILED_TO_WAIT_FOR_FENCES) C(VKFFT_ERROR_FAILED_TO_RESET_FENCES) C(VKFFT_ERROR_FAILED_TO_CREATE_DESCRIPTOR_POOL)
	C(VKFFT_ERROR_FAILED_TO_CREATE_DESCRIPTOR_SET_LAYOUT) C(VKFFT_ERROR_FAILED_TO_ALLOCATE_DESCRIPTOR_SETS) C(VKFFT_ERROR_FAILED_TO_CREATE_PIPELINE_LAYOUT)
	C(VKFFT_ERROR_FAILED_SHADER_PREPROCESS) C(VKFFT_ERROR_FAILED_SHADER_PARSE) C(VKFFT_ERROR_FAILED_SHADER_LINK) C(VKFFT_ERROR_FAILED_SPIRV_GENERATE)
	C(VKFFT_ERROR_FAILED_TO_CREATE_SHADER_MODULE) C(VKFFT_ERROR_FAILED_TO_CREATE_INSTANCE) C(VKFFT_ERROR_FAILED_TO_SETUP_DEBUG_MESSENGER)
	C(VKFFT_ERROR_FAILED_TO_FIND_PHYSICAL_DEVICE) C(VKFFT_ERROR_FAILED_TO_CREATE_DEVICE) C(VKFFT_ERROR_FAILED_TO_CREATE_FENCE)
	C(VKFFT_ERROR_FAILED_TO_CREATE_COMMAND_POOL) C(VKFFT_ERROR_FAILED_TO_CREATE_BUFFER) C(VKFFT_ERROR_FAILED_TO_ALLOCATE_MEMORY)
	C(VKFFT_ERROR_FAILED_TO_BIND_BUFFER_MEMORY) C(VKFFT_ERROR_FAILED_TO_FIND_MEMORY) C(VKFFT_ERROR_FAILED_TO_SYNCHRONIZE) C(VKFFT_ERROR_FAILED_TO_COPY)
	C(VKFFT_ERROR_FAILED_TO_CREATE_PROGRAM) C(VKFFT_ERROR_FAILED_TO_COMPILE_PROGRAM) C(VKFFT_ERROR_FAILED_TO_GET_CODE_SIZE) C(VKFFT_ERROR_FAILED_TO_GET_CODE)
	C(VKFFT_ERROR_FAILED_TO_DESTROY_PROGRAM) C(VKFFT_ERROR_FAILED_TO_LOAD_MODULE) C(VKFFT_ERROR_FAILED_TO_GET_FUNCTION)
	C(VKFFT_ERROR_FAILED_TO_SET_DYNAMIC_SHARED_MEMORY) C(VKFFT_ERROR_FAILED_TO_MODULE_GET_GLOBAL) C(VKFFT_ERROR_FAILED_TO_LAUNCH_KERNEL)
	C(VKFFT_ERROR_FAILED_TO_EVENT_RECORD) C(VKFFT_ERROR_FAILED_TO_ADD_NAME_EXPRESSION) C(VKFFT_ERROR_FAILED_TO_INITIALIZE)
	C(VKFFT_ERROR_FAILED_TO_SET_DEVICE_ID) C(VKFFT_ERROR_FAILED_TO_GET_DEVICE) C(VKFFT_ERROR_FAILED_TO_CREATE_CONTEXT) C(VKFFT_ERROR_FAILED_TO_CREATE_PIPELINE)
	C(VKFFT_ERROR_FAILED_TO_SET_KERNEL_ARG) C(VKFFT_ERROR_FAILED_TO_CREATE_COMMAND_QUEUE) C(VKFFT_ERROR_FAILED_TO_RELEASE_COMMAND_QUEUE)
	C(VKFFT_ERROR_FAILED_TO_ENUMERATE_DEVICES) C(VKFFT_ERROR_FAILED_TO_GET_ATTRIBUTE) C(VKFFT_ERROR_FAILED_TO_CREATE_EVENT)
	C(VKFFT_ERROR_FAILED_TO_CREATE_COMMAND_LIST) C(VKFFT_ERROR_FAILED_TO_DESTROY_COMMAND_LIST) C(VKFFT_ERROR_FAILED_TO_SUBMIT_BARRIER)
#undef C
	}
	return "Unknown VkFFT error";
}

} // extern "C"

namespace {

// Zero padding (VkFFTConfiguration::performZeropadding / fft_zeropad_left / fft_zeropad_right, vkFFT_Structs.h:150-155): the range
// [left, right) of an axis is taken as zero by the first transform that reads it — the forward one, or the inverse one with
// frequencyZeroPadding.  Like the reference (vkFFT_Zeropad.h:28) the single-pass kernels skip the range on the read side, leave it unwritten on the
// write side of the opposite direction and do not visit sequences inside the padded range of an axis still to come (planner.cpp, "zero padding").
// This is the fallback for the axes of AppState::zeroPadMask, whose plans cannot skip: the zeros are written, everything is transformed.
VkFFTResult zero_padded_ranges(VkFFTApplication* app, bool inverse, void* base, hipStream_t stream) {
	const VkFFTConfiguration& c = app->configuration;
	const bool r2c = c.performR2C != 0, real = c.performDCT || c.performDST || (r2c && !inverse);
	ZeroParams z;
	z.base = base;
	z.words = (c.doublePrecision ? 2u : 1u) * (real ? 1u : 2u);
	const uint64_t unit = (r2c && !inverse) ? 2 : 1; // the in-place real layout: strides in reals = twice the complex strides
	for (int d = 0; d < 4; d++) {
		z.size[d] = d < (int)c.FFTdim ? (uint32_t)c.size[d] : 1u;
		z.stride[d] = d == 0 ? 1 : (uint64_t)c.bufferStride[d - 1] * unit;
	}
	if (r2c && inverse) z.size[0] = (uint32_t)(c.size[0] / 2 + 1);
	z.systemStride = (uint64_t)c.bufferStride[c.FFTdim - 1] * unit;
	z.systems = (uint32_t)(app->actualNumBatches * c.coordinateFeatures);
	for (pfUINT i = 0; i < c.FFTdim; i++) {
		if (!c.performZeropadding[i] || !((((AppState*)app->impl)->zeroPadMask >> i) & 1u)) continue;
		z.axis = (uint32_t)i; z.left = (uint32_t)c.fft_zeropad_left[i]; z.right = (uint32_t)std::min<pfUINT>(c.fft_zeropad_right[i], z.size[i]);
		if (launch_zero_slab(z, stream)) return VKFFT_ERROR_FAILED_TO_LAUNCH_KERNEL;
	}
	return VKFFT_SUCCESS;
}

VkFFTResult initialize_convolution(VkFFTApplication* app, const VkFFTConfiguration& in) {
	auto unsupported = [&](const char* what) {
		fprintf(stderr, "vkfft_mi355x: convolution with %s is outside the scope of this library (see DESIGN.md)\n", what);
		return VKFFT_ERROR_PLAN_NOT_INITIALIZED;
	};
	const pfUINT m = in.matrixConvolution ? in.matrixConvolution : 1, nk = in.numberKernels ? in.numberKernels : 1, nb = in.numberBatches ? in.numberBatches : 1;
	if (m > 8) return unsupported("matrixConvolution > 8");
	if (nk > 1 && nb > 1) return unsupported("numberKernels > 1 and numberBatches > 1");
	if (in.isOutputFormatted) return unsupported("a separate output buffer");
	if (in.performDCT || in.performDST) return unsupported("R2R transforms");
	if (in.makeForwardPlanOnly || in.makeInversePlanOnly) return unsupported("makeForwardPlanOnly / makeInversePlanOnly");
	for (pfUINT i = 0; i < in.FFTdim; i++) if (in.omitDimension[i]) return VKFFT_ERROR_UNSUPPORTED_FFT_OMIT; // reference: vkFFT_InitializeApp.h:1385-1400
	AppState* st = new (std::nothrow) AppState();
	if (!st) return VKFFT_ERROR_MALLOC_FAILED;
	app->impl = st;
	VkFFTConfiguration& c = app->configuration;
	c = in;
	c.matrixConvolution = m; c.numberKernels = nk; c.numberBatches = nb;
	c.coordinateFeatures = m > 1 ? m : (in.coordinateFeatures ? in.coordinateFeatures : 1); // reference: vkFFT_InitializeApp.h:1374
	// (as for plain applications, setConfigurationVkFFT: the flag is reported back as not applied — and the notice is printed once, not by the sub-applications too)
	if (in.disableReorderFourStep && (in.printMemoryLayout || getenv("VKFFT_MI355X_PRINT_PLAN")))
		fprintf(stderr, "[vkfft_mi355x] disableReorderFourStep requested: not applied, results stay in natural order (app->configuration.disableReorderFourStep reads back 0)\n");
	c.reorderFourStep = 1; c.disableReorderFourStep = 0;
	app->actualNumBatches = nb;
	VkFFTConfiguration f = in, b = in;
	f.disableReorderFourStep = 0;
	f.performConvolution = 0; f.matrixConvolution = 0; f.numberKernels = 0; f.symmetricKernel = 0; f.conjugateConvolution = 0; f.crossPowerSpectrumNormalization = 0;
	f.kernel = nullptr; f.kernelSize = nullptr; f.kernelNum = 0;
	f.coordinateFeatures = c.coordinateFeatures;
	f.saveApplicationToString = 0; f.loadApplicationFromString = 0;
	b = f;
	f.makeForwardPlanOnly = 1;
	b.makeInversePlanOnly = 1;
	b.numberBatches = nb * nk;
	b.isInputFormatted = 0; b.inverseReturnToInputBuffer = 0; b.inputBuffer = nullptr; b.inputBufferSize = nullptr;
	if (nk > 1 && in.bufferSize) b.bufferSize = in.bufferSize;
	// merged last axis (the reference's convolution-merged kernel, vkFFT_Convolution.h:125): when a one-pass merged kernel exists for that axis
	// the two sub-applications omit it
	if (in.FFTdim >= 2 && nk == 1 && !in.crossPowerSpectrumNormalization && !in.frequencyZeroPadding && !in.isInputFormatted && !getenv("VKFFT_MI355X_CONV_SEPARATE")) {
		TransformDesc d;
		d.fftDim = (int)in.FFTdim;
		for (int i = 0; i < 4; i++) d.size[i] = in.size[i] ? in.size[i] : 1;
		d.batch = nb; d.dp = in.doublePrecision != 0; d.kind = in.performR2C ? 1 : 0;
		{ // default strides of the in-place layout (as the plain path derives them)
			uint64_t run = in.performR2C ? d.size[0] / 2 + 1 : d.size[0];
			for (int i = 0; i < 4; i++) { d.bufStride[i] = in.bufferStride[i] ? in.bufferStride[i] : run; if (i + 1 < (int)in.FFTdim) run = d.bufStride[i] * d.size[i + 1]; }
		}
		for (pfUINT i = 0; i < in.FFTdim && i < 4; i++) if (in.performZeropadding[i] && in.fft_zeropad_right[i] > in.fft_zeropad_left[i]) { d.padL[i] = in.fft_zeropad_left[i]; d.padR[i] = in.fft_zeropad_right[i]; }
		if (const char* e = getenv("VKFFT_MI355X_GENERIC_ONLY")) d.disableFastKernels = atoi(e) != 0;
		if (in.sharedMemorySize && in.sharedMemorySize < 160 * 1024) d.disableFastKernels = true;
		if (in.userTempBuffer && in.tempBufferSize) d.userTempBytes = in.tempBufferSize[0];
		ConvAxisDesc cd;
		cd.matrix = (uint32_t)m; cd.coordinates = (uint32_t)c.coordinateFeatures; cd.symmetric = in.symmetricKernel ? 1u : 0u; cd.conjugate = (uint32_t)in.conjugateConvolution;
		cd.kernelSystems = m > 1 ? (in.symmetricKernel ? m * (m + 1) / 2 : m * m) : c.coordinateFeatures;
		cd.scale = in.normalize ? 1.0 / (double)d.size[in.FFTdim - 1] : 1.0;
		VkFFTPlan* pl = (VkFFTPlan*)calloc(1, sizeof(VkFFTPlan));
		DirectionPlan* dpl = new (std::nothrow) DirectionPlan();
		if (pl && dpl) {
			pl->impl = dpl;
			// (a caller-supplied temp buffer that is too small for the split merged form: separate passes instead, whose own check reports 2016 if it is too small for them too)
			if (build_conv_axis_plan(d, cd, *dpl) == 0 && !dpl->arena.empty() && !(d.userTempBytes && dpl->totalTemp() > d.userTempBytes) && hipMalloc(&dpl->dArena, dpl->arena.size()) == hipSuccess &&
			    hipMemcpy(dpl->dArena, dpl->arena.data(), dpl->arena.size(), hipMemcpyHostToDevice) == hipSuccess &&
			    (dpl->totalTemp() == 0 || in.userTempBuffer || (hipMalloc(&st->tempOwned, dpl->totalTemp()) == hipSuccess && (st->tempOwnedBytes = dpl->totalTemp(), true)))) {
				st->convMid = pl;
				st->convInvFullCfg = b;
				f.omitDimension[in.FFTdim - 1] = 1; b.omitDimension[in.FFTdim - 1] = 1;
				if (in.printMemoryLayout || getenv("VKFFT_MI355X_PRINT_PLAN")) fprintf(stderr, "[vkfft_mi355x] convolution: axis %d merged (forward, %ux%u kernel product, inverse in one pass of pow2_col_blue_kernel)\n", (int)in.FFTdim - 1, cd.matrix, cd.matrix);
			} else free_direction(pl);
		} else { free(pl); delete dpl; }
	}
	st->convFwd = (VkFFTApplication*)calloc(1, sizeof(VkFFTApplication));
	st->convInv = (VkFFTApplication*)calloc(1, sizeof(VkFFTApplication));
	if (!st->convFwd || !st->convInv) { deleteVkFFT(app); return VKFFT_ERROR_MALLOC_FAILED; }
	VkFFTResult r = initializeVkFFT(st->convFwd, f);
	if (r == VKFFT_SUCCESS) r = initializeVkFFT(st->convInv, b);
	if (r == VKFFT_SUCCESS && st->convMid) {
		// the plain inverse over ALL axes of a merged application (VkFFTAppend(app, 1)): built here, not at its first use — planning, allocation and the table
		// upload do not belong inside a launch (stream capture, concurrent callers, and the caller's configuration storage may be gone by then)
		st->convInvFull = (VkFFTApplication*)calloc(1, sizeof(VkFFTApplication));
		if (!st->convInvFull) r = VKFFT_ERROR_MALLOC_FAILED;
		else {
			r = initializeVkFFT(st->convInvFull, st->convInvFullCfg);
			if (r != VKFFT_SUCCESS) { free(st->convInvFull); st->convInvFull = nullptr; }
		}
	}
	if (r != VKFFT_SUCCESS) { deleteVkFFT(app); return r; }
	for (int i = 0; i < VKFFT_MAX_FFT_DIMENSIONS; i++) c.bufferStride[i] = st->convFwd->configuration.bufferStride[i];
	app->firstAxis = 0; app->lastAxis = c.FFTdim - 1;
	return VKFFT_SUCCESS;
}

VkFFTResult append_convolution(VkFFTApplication* app, int inverse, VkFFTLaunchParams* lp) {
	VkFFTConfiguration& c = app->configuration;
	AppState* st = (AppState*)app->impl;
	if (lp) {
		if (lp->kernel) c.kernel = lp->kernel;
		if (lp->buffer) c.buffer = lp->buffer;
		if (c.specifyOffsetsAtLaunch) { c.kernelOffset = lp->kernelOffset; c.bufferOffset = lp->bufferOffset; }
	}
	VkFFTLaunchParams inv = VKFFT_ZERO_INIT;
	if (lp) { inv.buffer = lp->buffer; inv.tempBuffer = lp->tempBuffer; inv.bufferOffset = lp->bufferOffset; inv.tempBufferOffset = lp->tempBufferOffset; }
	if (inverse == 1) { // a plain inverse of the numberKernels results, over every axis
		if (!st->convMid) return VkFFTAppend(st->convInv, 1, lp ? &inv : nullptr);
		if (!st->convInvFull) return VKFFT_ERROR_PLAN_NOT_INITIALIZED; // (built by initializeVkFFT: convInv omits the merged axis)
		VkFFTLaunchParams full = inv;
		if (!lp || !lp->buffer) full.buffer = c.buffer;
		if (!full.tempBuffer && c.userTempBuffer) full.tempBuffer = c.tempBuffer;
		return VkFFTAppend(st->convInvFull, 1, &full);
	}
	if (c.kernel == nullptr || c.kernel[0] == nullptr) return VKFFT_ERROR_EMPTY_kernel;
	if (c.buffer == nullptr || c.buffer[0] == nullptr) return VKFFT_ERROR_EMPTY_buffer;
	VkFFTResult r = VkFFTAppend(st->convFwd, -1, lp);
	if (r != VKFFT_SUCCESS) return r;
	if (st->convMid) { // last axis forward, kernel product, last axis backwards: one pass
		LaunchBuffers lb;
		lb.base[ROLE_BUFFER] = (char*)c.buffer[0] + c.bufferOffset;
		lb.kernel = (const char*)c.kernel[0] + c.kernelOffset;
		if (((DirectionPlan*)st->convMid->impl)->totalTemp()) {
			if (c.userTempBuffer) { // (the launch parameters may replace the temp buffer, as in VkFFTAppend)
				void** tb = (lp && lp->tempBuffer) ? lp->tempBuffer : c.tempBuffer;
				if (tb == nullptr || tb[0] == nullptr) return VKFFT_ERROR_EMPTY_tempBuffer;
				lb.base[ROLE_TEMP] = (char*)tb[0] + ((lp && c.specifyOffsetsAtLaunch) ? lp->tempBufferOffset : c.tempBufferOffset);
			}
			else lb.base[ROLE_TEMP] = st->tempOwned;
		}
		StreamSet ss;
		if (c.stream && c.num_streams >= 1) ss.s[0] = c.stream[0];
		if (execute_direction(*(DirectionPlan*)st->convMid->impl, lb, ss, nullptr)) return VKFFT_ERROR_FAILED_TO_LAUNCH_KERNEL;
		return VkFFTAppend(st->convInv, 1, lp ? &inv : nullptr);
	}
	ConvParams p;
	p.data = (char*)c.buffer[0] + c.bufferOffset;
	p.kernel = (const char*)c.kernel[0] + c.kernelOffset;
	p.systemStride = c.bufferStride[c.FFTdim - 1];
	p.matrix = (uint32_t)c.matrixConvolution; p.coordinates = (uint32_t)c.coordinateFeatures;
	p.batches = (uint32_t)c.numberBatches; p.numKernels = (uint32_t)c.numberKernels;
	p.symmetric = c.symmetricKernel ? 1u : 0u; p.conjugate = (uint32_t)c.conjugateConvolution; p.crossPower = c.crossPowerSpectrumNormalization ? 1u : 0u;
	p.kernelSystems = p.matrix > 1 ? (p.symmetric ? p.matrix * (p.matrix + 1) / 2 : p.matrix * p.matrix) : p.coordinates;
	if (launch_conv_pointwise(p, c.doublePrecision != 0, c.stream ? c.stream[0] : nullptr)) return VKFFT_ERROR_FAILED_TO_LAUNCH_KERNEL;
	return VkFFTAppend(st->convInv, 1, lp ? &inv : nullptr);
}

} // namespace
