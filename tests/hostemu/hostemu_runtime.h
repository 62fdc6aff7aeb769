// TEST INFRASTRUCTURE ONLY — a minimal CPU emulation of the HIP SIMT execution model, so that the *same*
// kernel, planner and C-ABI sources that build the product with hipcc can be compiled with g++ into
// tests/hostemu/_build/libvkfft_hostemu.so and exercised by the CPU-only part of the test-suite
// (`pytest -m "not gpu"`): index maps, planner decisions, API behaviour.  It is never part of the product:
// vkfft_amd/lib/libvkfft_mi355x.so is built without VKFFT_HOSTEMU and has no CPU path at all.
//
// Model: one workgroup at a time; every work-item is a fiber (own stack, cooperative switch);
// __syncthreads() and the wave-level sync yield to a scheduler that releases a barrier when every live
// fiber of the workgroup (wave) has arrived.  "Device memory" is host memory.
#pragma once
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>
#include <functional>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#undef __shared__
#define __shared__ static
#ifndef __launch_bounds__
#define __launch_bounds__(...)
#endif

struct hostemu_uint3 { unsigned x, y, z; };
extern hostemu_uint3 threadIdx, blockIdx, blockDim, gridDim;

namespace hostemu {
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void syncthreads();
void wave_sync();
extern char* dyn_smem;
hipError_t e_malloc(void** p, size_t n);
hipError_t e_free(void* p);
hipError_t e_memcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t e_attr(int* v, hipDeviceAttribute_t a, int dev);
hipError_t e_event_create(hipEvent_t* e, unsigned flags);
hipError_t e_event_destroy(hipEvent_t e);
} // namespace hostemu

#define __syncthreads() hostemu::syncthreads()
#define VKFFT_DYN_SMEM(var) char* var = hostemu::dyn_smem;
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) hostemu::launch((grid), (block), (shmem), [&]() { kern(__VA_ARGS__); })
#define hipMalloc(p, n) hostemu::e_malloc((void**)(p), (n))
#define hipFree(p) hostemu::e_free((p))
#define hipMemcpy(d, s, n, k) hostemu::e_memcpy((d), (s), (n), (k))
#define hipDeviceGetAttribute(v, a, d) hostemu::e_attr((v), (a), (d))
#define hipEventCreateWithFlags(e, f) hostemu::e_event_create((e), (f))
#define hipEventDestroy(e) hostemu::e_event_destroy((e))
#define hipGetLastError() hipSuccess
#define hipStreamCreateWithFlags(s, f) (*(s) = nullptr, hipSuccess)
#define hipStreamDestroy(s) hipSuccess
#define hipEventRecord(e, s) hipSuccess
#define hipStreamWaitEvent(s, e, f) hipSuccess

#include <cmath>
inline void sincospif(float x, float* s, float* c) { *s = (float)std::sin(3.14159265358979323846 * (double)x); *c = (float)std::cos(3.14159265358979323846 * (double)x); }
