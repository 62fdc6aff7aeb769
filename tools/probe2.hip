// GPU probe (tools/, not product): what a persistent kernel can move per second when part of its traffic stays in the
// Infinity Cache (a small ring that is rewritten all the time) — the ceiling of the fused Four-Step design.  JSON lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <functional>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ inline __amdgpu_buffer_rsrc_t rsrc(const void* p) {
	const uint64_t a = (uint64_t)p;
	const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
	return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), (short)0, 0x7FFFFFF0, 0x00020000);
}

// blocks of BLK bytes; block i: src offset (i*BLK) mod srcRegion (+ srcBase), dst likewise.  MIX: odd blocks swap the roles
// (src <-> dst policies and regions), emulating A tiles (HBM -> ring) interleaved with B tiles (ring -> HBM).
template <int LDAUX, int STAUX, int W16> // W16: 1 = 16 bytes per lane, 0 = 8 bytes per lane
__global__ void __launch_bounds__(256) k_move(const char* src, uint64_t srcRegion, char* dst, uint64_t dstRegion, uint64_t nBlocks, uint32_t* ticket) {
	constexpr uint32_t BLK = 32768;
	__shared__ uint32_t sT;
	for (;;) {
		if (threadIdx.x == 0) sT = atomicAdd(ticket, 1u);
		__syncthreads();
		const uint64_t i = sT;
		__syncthreads();
		if (i >= nBlocks) break;
		const __amdgpu_buffer_rsrc_t rs = rsrc(src + (i * BLK) % srcRegion), rd = rsrc(dst + (i * BLK) % dstRegion);
		if constexpr (W16) {
			u32x4 v[8];
#pragma unroll
			for (int j = 0; j < 8; j++) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16u + j * 4096u, 0, LDAUX);
#pragma unroll
			for (int j = 0; j < 8; j++) __builtin_amdgcn_raw_buffer_store_b128(v[j], rd, threadIdx.x * 16u + j * 4096u, 0, STAUX);
		} else {
			u32x2 v[16];
#pragma unroll
			for (int j = 0; j < 16; j++) v[j] = __builtin_amdgcn_raw_buffer_load_b64(rs, threadIdx.x * 8u + j * 2048u, 0, LDAUX);
#pragma unroll
			for (int j = 0; j < 16; j++) __builtin_amdgcn_raw_buffer_store_b64(v[j], rd, threadIdx.x * 8u + j * 2048u, 0, STAUX);
		}
	}
}
// both directions in one launch: even blocks HBM -> ring (store policy STAUX), odd blocks ring -> HBM (load policy LDAUX)
template <int LDAUX, int STAUX>
__global__ void __launch_bounds__(256) k_mix(const char* big, uint64_t bigRegion, char* ring, uint64_t ringRegion, char* out, uint64_t nBlocks, uint32_t* ticket) {
	constexpr uint32_t BLK = 32768;
	__shared__ uint32_t sT;
	for (;;) {
		if (threadIdx.x == 0) sT = atomicAdd(ticket, 1u);
		__syncthreads();
		const uint64_t i = sT;
		__syncthreads();
		if (i >= nBlocks) break;
		const uint64_t h = i >> 1;
		u32x4 v[8];
		if (i & 1) {
			const __amdgpu_buffer_rsrc_t rs = rsrc(ring + (h * BLK) % ringRegion), rd = rsrc(out + (h * BLK) % bigRegion);
#pragma unroll
			for (int j = 0; j < 8; j++) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16u + j * 4096u, 0, LDAUX);
#pragma unroll
			for (int j = 0; j < 8; j++) __builtin_amdgcn_raw_buffer_store_b128(v[j], rd, threadIdx.x * 16u + j * 4096u, 0, 0);
		} else {
			const __amdgpu_buffer_rsrc_t rs = rsrc(big + (h * BLK) % bigRegion), rd = rsrc(ring + (h * BLK) % ringRegion);
#pragma unroll
			for (int j = 0; j < 8; j++) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16u + j * 4096u, 0, 0);
#pragma unroll
			for (int j = 0; j < 8; j++) __builtin_amdgcn_raw_buffer_store_b128(v[j], rd, threadIdx.x * 16u + j * 4096u, 0, STAUX);
		}
	}
}

static float timeit(hipStream_t s, int iters, const std::function<void()>& f) {
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); f(); f(); hipStreamSynchronize(s);
	hipEventRecord(a, s); for (int i = 0; i < iters; i++) f(); hipEventRecord(b, s); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b); return ms / iters;
}

int main() {
	hipStream_t s; hipStreamCreate(&s);
	const uint64_t GiB = 1ull << 30;
	char *A, *B, *R; uint32_t* tk;
	CK(hipMalloc(&A, GiB)); CK(hipMalloc(&B, GiB)); CK(hipMalloc(&R, GiB / 2)); CK(hipMalloc(&tk, 256));
	hipMemset(A, 1, GiB); hipMemset(B, 2, GiB); hipMemset(R, 3, GiB / 2);
	const uint64_t nBlocks = GiB / 32768;
	auto report = [&](const char* name, uint64_t srcReg, uint64_t dstReg, int wg, float ms, double bytes) {
		printf("{\"probe\":\"%s\",\"srcMiB\":%.1f,\"dstMiB\":%.1f,\"wgPerCu\":%d,\"ms\":%.4f,\"moved_GBps\":%.1f}\n", name, srcReg / 1048576.0, dstReg / 1048576.0, wg, ms, bytes / ms / 1e6);
		fflush(stdout);
	};
#define RUN(name, LD, ST, W, srcp, srcReg, dstp, dstReg, wg) { \
		float ms = timeit(s, 5, [&] { hipMemsetAsync(tk, 0, 4, s); hipLaunchKernelGGL((k_move<LD, ST, W>), dim3(256 * (wg)), dim3(256), 0, s, (const char*)(srcp), (uint64_t)(srcReg), (char*)(dstp), (uint64_t)(dstReg), nBlocks, tk); }); \
		report(name, srcReg, dstReg, wg, ms, 2.0 * GiB); }
	for (int wg : {4, 8}) {
		RUN("hbm_to_hbm_plain16", 0, 0, 1, A, GiB, B, GiB, wg);
		RUN("hbm_to_hbm_plain8", 0, 0, 0, A, GiB, B, GiB, wg);
		RUN("hbm_to_hbm_nt16", 2, 2, 1, A, GiB, B, GiB, wg);
	}
	for (uint64_t ring : {1ull << 20, 8ull << 20, 32ull << 20, 64ull << 20, 128ull << 20}) {
		RUN("ring_to_ring_plain16", 0, 0, 1, R, ring, R + (256ull << 20), ring, 8);
		RUN("ring_to_ring_sc1_16", 16, 16, 1, R, ring, R + (256ull << 20), ring, 8);
		RUN("ring_to_ring_sc1_8", 16, 16, 0, R, ring, R + (256ull << 20), ring, 8);
		RUN("ring_to_ring_ldsc1_stplain", 16, 0, 1, R, ring, R + (256ull << 20), ring, 8);
		RUN("hbm_to_ring_sc1st", 0, 16, 1, A, GiB, R, ring, 8);
		RUN("hbm_to_ring_plainst", 0, 0, 1, A, GiB, R, ring, 8);
		RUN("ring_to_hbm_sc1ld", 16, 0, 1, R, ring, B, GiB, 8);
		RUN("ring_to_hbm_sc1ld8", 16, 0, 0, R, ring, B, GiB, 8);
		RUN("ring_to_hbm_plainld", 0, 0, 1, R, ring, B, GiB, 8);
		{
			float ms = timeit(s, 5, [&] { hipMemsetAsync(tk, 0, 4, s); hipLaunchKernelGGL((k_mix<16, 16>), dim3(256 * 8), dim3(256), 0, s, (const char*)A, GiB, R, ring, B, 2 * nBlocks, tk); });
			report("mix_sc1 (1 GiB HBM rd + 1 GiB ring wr + 1 GiB ring rd + 1 GiB HBM wr)", GiB, ring, 8, ms, 4.0 * GiB);
			ms = timeit(s, 5, [&] { hipMemsetAsync(tk, 0, 4, s); hipLaunchKernelGGL((k_mix<0, 0>), dim3(256 * 8), dim3(256), 0, s, (const char*)A, GiB, R, ring, B, 2 * nBlocks, tk); });
			report("mix_plain", GiB, ring, 8, ms, 4.0 * GiB);
		}
	}
	return 0;
}
