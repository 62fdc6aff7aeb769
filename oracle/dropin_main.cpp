// TEST INFRASTRUCTURE ONLY — drop-in proof.  The reference's own benchmark caller
// (benchmark_scripts/vkFFT_scripts/src/sample_0_benchmark_VkFFT_single.cpp + utils_VkFFT.cpp) is compiled UNCHANGED, from where it
// lies under /root/reference, against THIS repository's include/vkFFT.h and linked with libvkfft_mi355x.so instead of the reference's
// header-only implementation (oracle/build_ref.sh).  This file only supplies what VkFFT_TestSuite.cpp does before calling a sample:
// pick the HIP device (VkFFT_TestSuite.cpp:283-297) and call the sample.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "vkFFT.h"
#include "utils_VkFFT.h"

VkFFTResult sample_0_benchmark_VkFFT_single(VkGPU* vkGPU, uint64_t file_output, FILE* output, uint64_t isCompilerInitialized);

int main(int argc, char** argv) {
	VkGPU gpu = {};
	gpu.device_id = argc > 1 ? (uint64_t)atoll(argv[1]) : 0;
	if (hipInit(0) != hipSuccess || hipSetDevice((int)gpu.device_id) != hipSuccess || hipDeviceGet(&gpu.device, (int)gpu.device_id) != hipSuccess) { printf("no HIP device\n"); return 2; }
	if (hipCtxCreate(&gpu.context, 0, gpu.device) != hipSuccess) { printf("no HIP context\n"); return 2; }
	VkFFTResult r = sample_0_benchmark_VkFFT_single(&gpu, 0, stdout, 0);
	printf("dropin sample_0 result: %d (%s)\n", (int)r, getVkFFTErrorString(r));
	return r == VKFFT_SUCCESS ? 0 : 1;
}
