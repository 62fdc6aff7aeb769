// GPU probe (tools/, not product): read-only and write-only ceilings by working-set size and cache policy.  JSON lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <functional>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ inline __amdgpu_buffer_rsrc_t rsrc(const void* p) {
	const uint64_t a = (uint64_t)p;
	const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
	return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), (short)0, 0x7FFFFFF0, 0x00020000);
}
constexpr uint32_t BLK = 32768;
template <int AUX> __global__ void __launch_bounds__(256) k_read(const char* src, uint64_t region, uint64_t nBlocks, uint32_t* sink) {
	u32x4 acc = {0, 0, 0, 0};
	for (uint64_t i = blockIdx.x; i < nBlocks; i += gridDim.x) {
		const __amdgpu_buffer_rsrc_t rs = rsrc(src + (i * BLK) % region);
		u32x4 v[8];
#pragma unroll
		for (int j = 0; j < 8; j++) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16u + j * 4096u, 0, AUX);
#pragma unroll
		for (int j = 0; j < 8; j++) acc ^= v[j];
	}
	if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = acc.x;
}
template <int AUX> __global__ void __launch_bounds__(256) k_write(char* dst, uint64_t region, uint64_t nBlocks) {
	u32x4 v = {threadIdx.x, blockIdx.x, 3, 4};
	for (uint64_t i = blockIdx.x; i < nBlocks; i += gridDim.x) {
		const __amdgpu_buffer_rsrc_t rd = rsrc(dst + (i * BLK) % region);
#pragma unroll
		for (int j = 0; j < 8; j++) __builtin_amdgcn_raw_buffer_store_b128(v, rd, threadIdx.x * 16u + j * 4096u, 0, AUX);
		v.x += 1;
	}
}
template <int LDAUX, int STAUX> __global__ void __launch_bounds__(256) k_copy(const char* src, uint64_t sregion, char* dst, uint64_t dregion, uint64_t nBlocks) {
	for (uint64_t i = blockIdx.x; i < nBlocks; i += gridDim.x) {
		const __amdgpu_buffer_rsrc_t rs = rsrc(src + (i * BLK) % sregion), rd = rsrc(dst + (i * BLK) % dregion);
		u32x4 v[8];
#pragma unroll
		for (int j = 0; j < 8; j++) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16u + j * 4096u, 0, LDAUX);
#pragma unroll
		for (int j = 0; j < 8; j++) __builtin_amdgcn_raw_buffer_store_b128(v[j], rd, threadIdx.x * 16u + j * 4096u, 0, STAUX);
	}
}
static float timeit(hipStream_t s, int iters, const std::function<void()>& f) {
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); f(); f(); (void)hipStreamSynchronize(s);
	(void)hipEventRecord(a, s); for (int i = 0; i < iters; i++) f(); (void)hipEventRecord(b, s); (void)hipEventSynchronize(b);
	float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / iters;
}
int main() {
	hipStream_t s; (void)hipStreamCreate(&s);
	const uint64_t GiB = 1ull << 30;
	char *A, *B; uint32_t* sink;
	CK(hipMalloc(&A, GiB)); CK(hipMalloc(&B, GiB)); CK(hipMalloc(&sink, 4096));
	(void)hipMemset(A, 1, GiB); (void)hipMemset(B, 2, GiB);
	const uint64_t nBlocks = GiB / BLK;
	auto rep = [&](const char* name, uint64_t reg, int wg, float ms, double bytes) {
		printf("{\"probe\":\"%s\",\"regionMiB\":%.1f,\"wgPerCu\":%d,\"ms\":%.4f,\"GBps\":%.1f}\n", name, reg / 1048576.0, wg, ms, bytes / ms / 1e6); fflush(stdout);
	};
	for (uint64_t reg : {1ull << 20, 16ull << 20, 64ull << 20, 128ull << 20, 1ull << 30}) {
		for (int wg : {8}) {
			float ms;
			ms = timeit(s, 5, [&] { hipLaunchKernelGGL((k_read<0>), dim3(256 * wg), dim3(256), 0, s, (const char*)A, reg, nBlocks, sink); }); rep("read_plain", reg, wg, ms, (double)GiB);
			ms = timeit(s, 5, [&] { hipLaunchKernelGGL((k_read<16>), dim3(256 * wg), dim3(256), 0, s, (const char*)A, reg, nBlocks, sink); }); rep("read_sc1", reg, wg, ms, (double)GiB);
			ms = timeit(s, 5, [&] { hipLaunchKernelGGL((k_read<2>), dim3(256 * wg), dim3(256), 0, s, (const char*)A, reg, nBlocks, sink); }); rep("read_nt", reg, wg, ms, (double)GiB);
			ms = timeit(s, 5, [&] { hipLaunchKernelGGL((k_write<0>), dim3(256 * wg), dim3(256), 0, s, B, reg, nBlocks); }); rep("write_plain", reg, wg, ms, (double)GiB);
			ms = timeit(s, 5, [&] { hipLaunchKernelGGL((k_write<16>), dim3(256 * wg), dim3(256), 0, s, B, reg, nBlocks); }); rep("write_sc1", reg, wg, ms, (double)GiB);
			ms = timeit(s, 5, [&] { hipLaunchKernelGGL((k_write<2>), dim3(256 * wg), dim3(256), 0, s, B, reg, nBlocks); }); rep("write_nt", reg, wg, ms, (double)GiB);
			ms = timeit(s, 5, [&] { hipLaunchKernelGGL((k_copy<0, 0>), dim3(256 * wg), dim3(256), 0, s, (const char*)A, reg, B, reg, nBlocks); }); rep("copy_plain(r+w)", reg, wg, ms, 2.0 * GiB);
			ms = timeit(s, 5, [&] { hipLaunchKernelGGL((k_copy<16, 16>), dim3(256 * wg), dim3(256), 0, s, (const char*)A, reg, B, reg, nBlocks); }); rep("copy_sc1(r+w)", reg, wg, ms, 2.0 * GiB);
		}
	}
	// HBM read + ring write, ring read + HBM write
	for (uint64_t reg : {16ull << 20, 64ull << 20}) {
		float ms;
		ms = timeit(s, 5, [&] { hipLaunchKernelGGL((k_copy<0, 16>), dim3(2048), dim3(256), 0, s, (const char*)A, GiB, B, reg, nBlocks); }); rep("copy_hbm_to_ring_sc1(r+w)", reg, 8, ms, 2.0 * GiB);
		ms = timeit(s, 5, [&] { hipLaunchKernelGGL((k_copy<16, 0>), dim3(2048), dim3(256), 0, s, (const char*)B, reg, A, GiB, nBlocks); }); rep("copy_ring_sc1_to_hbm(r+w)", reg, 8, ms, 2.0 * GiB);
	}
	for (int wg : {2, 4, 16}) {
		float ms = timeit(s, 5, [&] { hipLaunchKernelGGL((k_copy<0, 0>), dim3(256 * wg), dim3(256), 0, s, (const char*)A, GiB, B, GiB, nBlocks); }); rep("copy_plain(r+w)", GiB, wg, ms, 2.0 * GiB);
	}
	return 0;
}
