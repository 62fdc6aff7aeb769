/*
 * vkFFT.h — drop-in C API of the MI355X-native FFT library (libvkfft_mi355x.so).
 *
 * This header keeps the plan-create / append / delete interface of DTolm/VkFFT v1.3.4 so that
 * code written against the reference's HIP backend (`-DVKFFT_BACKEND=2`) recompiles unchanged and
 * links against the shared library instead of pulling in the header-only run-time code generator.
 *
 *   reference interface replaced                        reference file:line
 *   --------------------------------------------------  ------------------------------------------------------
 *   VkFFTConfiguration (HIP arm, same names/order)      vkFFT/vkFFT/vkFFT_Structs/vkFFT_Structs.h:93-324
 *   VkFFTLaunchParams  (HIP arm)                        vkFFT/vkFFT/vkFFT_Structs/vkFFT_Structs.h:326-379
 *   VkFFTResult                                         vkFFT/vkFFT/vkFFT_Structs/vkFFT_Structs.h:380-477
 *   getVkFFTErrorString                                 vkFFT/vkFFT/vkFFT_Structs/vkFFT_Structs.h:479
 *   VkFFTPlan / VkFFTApplication (public prefix)        vkFFT/vkFFT/vkFFT_Structs/vkFFT_Structs.h:1118-1191
 *   initializeVkFFT                                     vkFFT/vkFFT/vkFFT_AppManagement/vkFFT_InitializeApp.h:1468
 *   VkFFTAppend                                         vkFFT/vkFFT/vkFFT_AppManagement/vkFFT_RunApp.h:79
 *   deleteVkFFT                                         vkFFT/vkFFT/vkFFT_AppManagement/vkFFT_DeleteApp.h:28
 *   VkFFTGetVersion                                     vkFFT/vkFFT.h:109
 *
 * Differences from the reference, by design:
 *   - the five functions are exported C symbols (the reference has them `static inline`);
 *   - kernels are ahead-of-time compiled HIP for gfx950: the JIT/cache fields
 *     (saveApplicationToString, loadApplicationFromString, keepShaderCode, maxCodeLength, ...) are
 *     accepted and ignored;
 *   - only the HIP backend exists: defining VKFFT_BACKEND to anything but 2 is an error;
 *   - convolution and zero padding are supported with the restrictions listed in INTEGRATION.md
 *     (DESIGN.md section 4.11); half / double-double precision and buffers split over several
 *     allocations are accepted in the struct for layout compatibility and rejected at plan creation.
 */
#ifndef VKFFT_H
#define VKFFT_H

#include <stdint.h>
#include <stddef.h>

#ifndef VKFFT_BACKEND
#define VKFFT_BACKEND 2
#endif
#if (VKFFT_BACKEND != 2)
#error "vkfft_mi355x provides the HIP backend only: compile with -DVKFFT_BACKEND=2"
#endif

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
#define VKFFT_ZERO_INIT {}
#else
#define VKFFT_ZERO_INIT {0}
#endif

#ifndef VKFFT_MAX_FFT_DIMENSIONS
#define VKFFT_MAX_FFT_DIMENSIONS 4
#endif

#define pfLD long double
#define pfUINT uint64_t
#define pfINT int64_t

#if defined(_WIN32)
#define VKFFT_API
#else
#define VKFFT_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * Plan-creation parameters.  Zero means "default" for every field.
 * Field names, types and order follow the reference's HIP arm exactly.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
	pfUINT FFTdim;                         /* 1..VKFFT_MAX_FFT_DIMENSIONS */
	pfUINT size[VKFFT_MAX_FFT_DIMENSIONS]; /* WHD(+) sizes, size[0] is the unit-stride axis */

	hipDevice_t* device;  /* from hipDeviceGet; required */
	hipStream_t* stream;  /* optional array of streams kernels are enqueued on */
	pfUINT num_streams;   /* number of entries in stream[]; 0 -> default stream */

	pfUINT userTempBuffer; /* 1: caller supplies tempBuffer (+tempBufferSize) */

	pfUINT bufferNum;       /* accepted; always one buffer on HIP */
	pfUINT tempBufferNum;
	pfUINT inputBufferNum;
	pfUINT outputBufferNum;
	pfUINT kernelNum;

	pfUINT* bufferSize;       /* bytes; optional on HIP */
	pfUINT* tempBufferSize;
	pfUINT* inputBufferSize;
	pfUINT* outputBufferSize;
	pfUINT* kernelSize;

	void** buffer;       /* pointer to the device pointer holding the (in-place) data */
	void** tempBuffer;   /* scratch for multi-pass (Four-Step) and Bluestein plans */
	void** inputBuffer;  /* used when isInputFormatted */
	void** outputBuffer; /* used when isOutputFormatted */
	void** kernel;       /* convolution kernel spectra (performConvolution plans) */

	pfUINT bufferOffset; /* byte offsets of the first element */
	pfUINT tempBufferOffset;
	pfUINT inputBufferOffset;
	pfUINT outputBufferOffset;
	pfUINT kernelOffset;
	pfUINT specifyOffsetsAtLaunch; /* 1: take the offsets from VkFFTLaunchParams */

	pfUINT coalescedMemory; /* tuning knobs of the reference planner: accepted, the MI355X planner has its own */
	pfUINT aimThreads;
	pfUINT numSharedBanks;
	pfUINT inverseReturnToInputBuffer; /* inverse writes to inputBuffer (needs isInputFormatted) */
	pfUINT numberBatches;              /* N of WHDCN */
	pfUINT useUint64;
	pfUINT omitDimension[VKFFT_MAX_FFT_DIMENSIONS]; /* 1: skip the FFT along this axis */
	int performBandwidthBoost;
	pfUINT groupedBatch[VKFFT_MAX_FFT_DIMENSIONS];

	pfUINT doublePrecision; /* 1: fp64 data and arithmetic */
	pfUINT quadDoubleDoublePrecision;             /* not supported */
	pfUINT quadDoubleDoublePrecisionDoubleMemory; /* not supported */
	pfUINT halfPrecision;                         /* not supported */
	pfUINT halfPrecisionMemoryOnly;               /* not supported */
	pfUINT doublePrecisionFloatMemory;            /* not supported */

	pfUINT performR2C; /* 1: real-to-complex forward / complex-to-real inverse along axis 0 */
	pfUINT performDCT; /* 1..4: DCT type (FFTW REDFT00/10/01/11, unnormalised) */
	pfUINT performDST; /* 1..4: DST type (FFTW RODFT00/10/01/11, unnormalised) */
	pfUINT disableMergeSequencesR2C;
	pfUINT forceCallbackVersionRealTransforms;

	pfUINT normalize;              /* 1: scale the inverse by 1/N (1/(2N), 1/(2(N-1)) for R2R) */
	pfUINT disableReorderFourStep; /* reference: 1 = leave multi-upload output in its transposed order.  This library always returns natural order and says so: after initializeVkFFT app->configuration.disableReorderFourStep reads 0 */
	pfINT useLUT;                  /* accepted; twiddles always come from double-precision LUTs here */
	pfINT useLUT_4step;
	pfUINT makeForwardPlanOnly;
	pfUINT makeInversePlanOnly;

	pfUINT bufferStride[VKFFT_MAX_FFT_DIMENSIONS]; /* element strides; default W, W*H, W*H*D, ... */
	pfUINT isInputFormatted;
	pfUINT isOutputFormatted;
	pfUINT inputBufferStride[VKFFT_MAX_FFT_DIMENSIONS];
	pfUINT outputBufferStride[VKFFT_MAX_FFT_DIMENSIONS];
	pfUINT swapTo2Stage4Step;
	pfUINT swapTo3Stage4Step;

	pfUINT considerAllAxesStrided;
	pfUINT keepShaderCode;    /* no run-time code generation: ignored */
	pfUINT printMemoryLayout; /* 1: print which buffer every pass reads / writes */

	pfUINT saveApplicationToString;   /* kernel cache: ignored (kernels are AOT) */
	pfUINT loadApplicationFromString; /* ignored */
	void* loadApplicationString;      /* ignored */

	pfUINT disableSetLocale;

	pfUINT fixMaxRadixBluestein;      /* largest prime allowed in the Bluestein padded length (2..13) */
	pfUINT forceBluesteinSequenceSize;
	pfUINT useCustomBluesteinPaddingPattern;
	pfUINT* primeSizes;
	pfUINT* paddedSizes;

	pfUINT fixMinRaderPrimeMult; /* Rader policy, see DESIGN.md */
	pfUINT fixMaxRaderPrimeMult;
	pfUINT fixMinRaderPrimeFFT;
	pfUINT fixMaxRaderPrimeFFT;

	pfUINT performZeropadding[VKFFT_MAX_FFT_DIMENSIONS]; /* vkFFT_Structs.h; semantics in INTEGRATION.md */
	pfUINT fft_zeropad_left[VKFFT_MAX_FFT_DIMENSIONS];
	pfUINT fft_zeropad_right[VKFFT_MAX_FFT_DIMENSIONS];
	pfUINT frequencyZeroPadding;

	pfUINT performConvolution; /* forward -> product with `kernel` -> inverse in one VkFFTAppend(-1) */
	pfUINT conjugateConvolution;
	pfUINT crossPowerSpectrumNormalization;
	pfUINT coordinateFeatures; /* C of WHDCN */
	pfUINT matrixConvolution;
	pfUINT symmetricKernel;
	pfUINT numberKernels;
	pfUINT kernelConvolution;

	pfUINT registerBoost; /* reference planner knobs: ignored */
	pfUINT registerBoostNonPow2;
	pfUINT registerBoost4Step;

	pfUINT devicePageSize;
	pfUINT localPageSize;

	/* filled by initializeVkFFT from the device */
	pfUINT computeCapabilityMajor;
	pfUINT computeCapabilityMinor;
	pfUINT maxComputeWorkGroupCount[VKFFT_MAX_FFT_DIMENSIONS];
	pfUINT maxComputeWorkGroupSize[VKFFT_MAX_FFT_DIMENSIONS];
	pfUINT maxThreadsNum;
	pfUINT sharedMemorySizeStatic;
	pfUINT sharedMemorySize;
	pfUINT sharedMemorySizePow2;
	pfUINT warpSize;
	pfUINT halfThreads;
	pfUINT allocateTempBuffer; /* 1 when the library allocated tempBuffer itself */
	pfUINT reorderFourStep;
	pfINT maxCodeLength;
	pfINT maxTempLength;
	pfUINT autoCustomBluesteinPaddingPattern;
	pfUINT useRaderUintLUT;
	pfUINT vendorID;

	hipEvent_t* stream_event; /* per-stream events when num_streams > 1 */
	pfUINT streamCounter;
	pfUINT streamID;
	pfINT useStrict32BitAddress;
} VkFFTConfiguration;

/* Launch-time overrides: any non-NULL pointer replaces the one given at plan creation. */
typedef struct {
	void** buffer;
	void** tempBuffer;
	void** inputBuffer;
	void** outputBuffer;
	void** kernel;

	pfUINT bufferOffset;
	pfUINT tempBufferOffset;
	pfUINT inputBufferOffset;
	pfUINT outputBufferOffset;
	pfUINT kernelOffset;
} VkFFTLaunchParams;

typedef enum VkFFTResult {
	VKFFT_SUCCESS = 0,
	VKFFT_ERROR_MALLOC_FAILED = 1,
	VKFFT_ERROR_INSUFFICIENT_CODE_BUFFER = 2,
	VKFFT_ERROR_INSUFFICIENT_TEMP_BUFFER = 3,
	VKFFT_ERROR_PLAN_NOT_INITIALIZED = 4,
	VKFFT_ERROR_NULL_TEMP_PASSED = 5,
	VKFFT_ERROR_MATH_FAILED = 6,
	VKFFT_ERROR_FFTdim_GT_MAX_FFT_DIMENSIONS = 7,
	VKFFT_ERROR_NONZERO_APP_INITIALIZATION = 8,
	VKFFT_ERROR_INVALID_PHYSICAL_DEVICE = 1001,
	VKFFT_ERROR_INVALID_DEVICE = 1002,
	VKFFT_ERROR_INVALID_QUEUE = 1003,
	VKFFT_ERROR_INVALID_COMMAND_POOL = 1004,
	VKFFT_ERROR_INVALID_FENCE = 1005,
	VKFFT_ERROR_ONLY_FORWARD_FFT_INITIALIZED = 1006,
	VKFFT_ERROR_ONLY_INVERSE_FFT_INITIALIZED = 1007,
	VKFFT_ERROR_INVALID_CONTEXT = 1008,
	VKFFT_ERROR_INVALID_PLATFORM = 1009,
	VKFFT_ERROR_ENABLED_saveApplicationToString = 1010,
	VKFFT_ERROR_EMPTY_FILE = 1011,
	VKFFT_ERROR_EMPTY_FFTdim = 2001,
	VKFFT_ERROR_EMPTY_size = 2002,
	VKFFT_ERROR_EMPTY_bufferSize = 2003,
	VKFFT_ERROR_EMPTY_buffer = 2004,
	VKFFT_ERROR_EMPTY_tempBufferSize = 2005,
	VKFFT_ERROR_EMPTY_tempBuffer = 2006,
	VKFFT_ERROR_EMPTY_inputBufferSize = 2007,
	VKFFT_ERROR_EMPTY_inputBuffer = 2008,
	VKFFT_ERROR_EMPTY_outputBufferSize = 2009,
	VKFFT_ERROR_EMPTY_outputBuffer = 2010,
	VKFFT_ERROR_EMPTY_kernelSize = 2011,
	VKFFT_ERROR_EMPTY_kernel = 2012,
	VKFFT_ERROR_EMPTY_applicationString = 2013,
	VKFFT_ERROR_EMPTY_useCustomBluesteinPaddingPattern_arrays = 2014,
	VKFFT_ERROR_EMPTY_app = 2015,
	VKFFT_ERROR_INVALID_user_tempBuffer_too_small = 2016,
	VKFFT_ERROR_UNSUPPORTED_RADIX = 3001,
	VKFFT_ERROR_UNSUPPORTED_FFT_LENGTH = 3002,
	VKFFT_ERROR_UNSUPPORTED_FFT_LENGTH_R2C = 3003,
	VKFFT_ERROR_UNSUPPORTED_FFT_LENGTH_R2R = 3004,
	VKFFT_ERROR_UNSUPPORTED_FFT_OMIT = 3005,
	VKFFT_ERROR_FAILED_TO_ALLOCATE = 4001,
	VKFFT_ERROR_FAILED_TO_MAP_MEMORY = 4002,
	VKFFT_ERROR_FAILED_TO_ALLOCATE_COMMAND_BUFFERS = 4003,
	VKFFT_ERROR_FAILED_TO_BEGIN_COMMAND_BUFFER = 4004,
	VKFFT_ERROR_FAILED_TO_END_COMMAND_BUFFER = 4005,
	VKFFT_ERROR_FAILED_TO_SUBMIT_QUEUE = 4006,
	VKFFT_ERROR_FAILED_TO_WAIT_FOR_FENCES = 4007,
	VKFFT_ERROR_FAILED_TO_RESET_FENCES = 4008,
	VKFFT_ERROR_FAILED_TO_CREATE_DESCRIPTOR_POOL = 4009,
	VKFFT_ERROR_FAILED_TO_CREATE_DESCRIPTOR_SET_LAYOUT = 4010,
	VKFFT_ERROR_FAILED_TO_ALLOCATE_DESCRIPTOR_SETS = 4011,
	VKFFT_ERROR_FAILED_TO_CREATE_PIPELINE_LAYOUT = 4012,
	VKFFT_ERROR_FAILED_SHADER_PREPROCESS = 4013,
	VKFFT_ERROR_FAILED_SHADER_PARSE = 4014,
	VKFFT_ERROR_FAILED_SHADER_LINK = 4015,
	VKFFT_ERROR_FAILED_SPIRV_GENERATE = 4016,
	VKFFT_ERROR_FAILED_TO_CREATE_SHADER_MODULE = 4017,
	VKFFT_ERROR_FAILED_TO_CREATE_INSTANCE = 4018,
	VKFFT_ERROR_FAILED_TO_SETUP_DEBUG_MESSENGER = 4019,
	VKFFT_ERROR_FAILED_TO_FIND_PHYSICAL_DEVICE = 4020,
	VKFFT_ERROR_FAILED_TO_CREATE_DEVICE = 4021,
	VKFFT_ERROR_FAILED_TO_CREATE_FENCE = 4022,
	VKFFT_ERROR_FAILED_TO_CREATE_COMMAND_POOL = 4023,
	VKFFT_ERROR_FAILED_TO_CREATE_BUFFER = 4024,
	VKFFT_ERROR_FAILED_TO_ALLOCATE_MEMORY = 4025,
	VKFFT_ERROR_FAILED_TO_BIND_BUFFER_MEMORY = 4026,
	VKFFT_ERROR_FAILED_TO_FIND_MEMORY = 4027,
	VKFFT_ERROR_FAILED_TO_SYNCHRONIZE = 4028,
	VKFFT_ERROR_FAILED_TO_COPY = 4029,
	VKFFT_ERROR_FAILED_TO_CREATE_PROGRAM = 4030,
	VKFFT_ERROR_FAILED_TO_COMPILE_PROGRAM = 4031,
	VKFFT_ERROR_FAILED_TO_GET_CODE_SIZE = 4032,
	VKFFT_ERROR_FAILED_TO_GET_CODE = 4033,
	VKFFT_ERROR_FAILED_TO_DESTROY_PROGRAM = 4034,
	VKFFT_ERROR_FAILED_TO_LOAD_MODULE = 4035,
	VKFFT_ERROR_FAILED_TO_GET_FUNCTION = 4036,
	VKFFT_ERROR_FAILED_TO_SET_DYNAMIC_SHARED_MEMORY = 4037,
	VKFFT_ERROR_FAILED_TO_MODULE_GET_GLOBAL = 4038,
	VKFFT_ERROR_FAILED_TO_LAUNCH_KERNEL = 4039,
	VKFFT_ERROR_FAILED_TO_EVENT_RECORD = 4040,
	VKFFT_ERROR_FAILED_TO_ADD_NAME_EXPRESSION = 4041,
	VKFFT_ERROR_FAILED_TO_INITIALIZE = 4042,
	VKFFT_ERROR_FAILED_TO_SET_DEVICE_ID = 4043,
	VKFFT_ERROR_FAILED_TO_GET_DEVICE = 4044,
	VKFFT_ERROR_FAILED_TO_CREATE_CONTEXT = 4045,
	VKFFT_ERROR_FAILED_TO_CREATE_PIPELINE = 4046,
	VKFFT_ERROR_FAILED_TO_SET_KERNEL_ARG = 4047,
	VKFFT_ERROR_FAILED_TO_CREATE_COMMAND_QUEUE = 4048,
	VKFFT_ERROR_FAILED_TO_RELEASE_COMMAND_QUEUE = 4049,
	VKFFT_ERROR_FAILED_TO_ENUMERATE_DEVICES = 4050,
	VKFFT_ERROR_FAILED_TO_GET_ATTRIBUTE = 4051,
	VKFFT_ERROR_FAILED_TO_CREATE_EVENT = 4052,
	VKFFT_ERROR_FAILED_TO_CREATE_COMMAND_LIST = 4053,
	VKFFT_ERROR_FAILED_TO_DESTROY_COMMAND_LIST = 4054,
	VKFFT_ERROR_FAILED_TO_SUBMIT_BARRIER = 4055
} VkFFTResult;

/* Per-direction plan.  The three leading members are the ones reference callers read back
 * (e.g. sample_0 prints localFFTPlan->numAxisUploads[i]); everything else is library-private. */
typedef struct {
	pfUINT actualFFTSizePerAxis[VKFFT_MAX_FFT_DIMENSIONS][VKFFT_MAX_FFT_DIMENSIONS];
	pfUINT numAxisUploads[VKFFT_MAX_FFT_DIMENSIONS]; /* HBM passes used by each axis */
	pfUINT axisSplit[VKFFT_MAX_FFT_DIMENSIONS][4];   /* pass lengths, unit-stride pass first */
	pfUINT bigSequenceEvenR2C;
	pfUINT actualPerformR2CPerAxis[VKFFT_MAX_FFT_DIMENSIONS];
	void* impl; /* opaque: compiled pass list (vkfft_mi355x::DirectionPlan) */
} VkFFTPlan;

typedef struct {
	VkFFTConfiguration configuration;
	VkFFTPlan* localFFTPlan;
	VkFFTPlan* localFFTPlan_inverse;

	pfUINT actualNumBatches;
	pfUINT firstAxis;
	pfUINT lastAxis;
	pfUINT useBluesteinFFT[VKFFT_MAX_FFT_DIMENSIONS];
	void* bufferRaderUintLUT[VKFFT_MAX_FFT_DIMENSIONS][4];
	void* bufferBluestein[VKFFT_MAX_FFT_DIMENSIONS];
	void* bufferBluesteinFFT[VKFFT_MAX_FFT_DIMENSIONS];
	void* bufferBluesteinIFFT[VKFFT_MAX_FFT_DIMENSIONS];
	pfUINT bufferRaderUintLUTSize[VKFFT_MAX_FFT_DIMENSIONS][4];
	pfUINT bufferBluesteinSize[VKFFT_MAX_FFT_DIMENSIONS];
	void* applicationBluesteinString[VKFFT_MAX_FFT_DIMENSIONS];
	pfUINT applicationBluesteinStringSize[VKFFT_MAX_FFT_DIMENSIONS];

	pfUINT numRaderFFTPrimes;
	pfUINT rader_primes[30];
	pfUINT rader_buffer_size[30];
	void* raderFFTkernel[30];
	pfUINT applicationStringOffsetRader;
	pfUINT currentApplicationStringPos;

	pfUINT applicationStringSize; /* always 0: there is no kernel cache to save */
	void* saveApplicationString;  /* always NULL */

	void* impl; /* opaque: vkfft_mi355x::AppState (device LUTs, temp buffer, events) */
} VkFFTApplication;

/* Build forward and inverse plans.  `app` must be zero-initialised; the configuration is passed by
 * value as in the reference.  On failure everything is released and *app is zeroed again. */
VKFFT_API VkFFTResult initializeVkFFT(VkFFTApplication* app, VkFFTConfiguration inputLaunchConfiguration);

/* Enqueue one transform on the plan's stream(s).  inverse == 1 selects the inverse plan, any other
 * value the forward plan (callers pass -1).  Asynchronous: the caller synchronises. */
VKFFT_API VkFFTResult VkFFTAppend(VkFFTApplication* app, int inverse, VkFFTLaunchParams* launchParams);

/* Release everything the library owns and zero *app. */
VKFFT_API void deleteVkFFT(VkFFTApplication* app);

/* 10304 = 1.3.4, the reference version whose API this header mirrors. */
VKFFT_API int VkFFTGetVersion(void);

VKFFT_API const char* getVkFFTErrorString(VkFFTResult result);

/* Extension (not in the reference): sizes of the public structs as compiled into the library, for
 * FFI bindings to check their own layout: out[0]=sizeof(VkFFTConfiguration),
 * out[1]=sizeof(VkFFTLaunchParams), out[2]=sizeof(VkFFTPlan), out[3]=sizeof(VkFFTApplication). */
VKFFT_API void vkfftMI355XStructSizes(pfUINT out[4]);

/* Extension (not in the reference, which only prints such information under printMemoryLayout): what one VkFFTAppend of this
 * application enqueues.  Returns the number of kernel launches of the forward (inverse != 1) or inverse plan and writes their
 * kernel families, comma separated, into names[0..cap) (NUL terminated, truncated if cap is too small; names may be NULL). */
VKFFT_API int vkfftMI355XDescribePlan(const VkFFTApplication* app, int inverse, char* names, pfUINT cap);

/* Extension (not in the reference; measurement only): the library's own streaming device-to-device copy of `bytes` bytes (a multiple of 16, both
 * pointers 16-byte aligned), 16 bytes per lane, enqueued on `stream` (a hipStream_t, NULL = the default stream).  bench.py times it beside the
 * transforms: what a plain copy of the same buffer reaches on the same GPU is the practical ceiling of the HBM roofline.  Returns a VkFFTResult. */
VKFFT_API int vkfftMI355XStreamCopy(void* dst, const void* src, pfUINT bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VKFFT_H */
