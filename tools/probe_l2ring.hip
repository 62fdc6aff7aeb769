// GPU probe (tools/, not product): the question the round-5 review left open for 2^16 / 2^17 — can the intermediate of a two-factor transform make its round trip
// through the XCD's own L2 instead of the fabric?  Tile shapes of the real kernel: an A tile = 256 rows x 16 columns of fp32 complex = 256 segments of 128 bytes
// (row pitch n1 * 8 bytes) = 32 KiB; ring = SLOTS tiles per XCD (64 x 32 KiB = 2 MiB: every CU of the XCD with two tiles in flight).
// Each persistent workgroup: read an A tile from HBM (nt), write it to ITS XCD's ring slot, barrier + drain, read the slot back, write a B-shaped tile to HBM (nt).
// Modes: 0 = no ring (plain tile copy), 1 = XCD-private ring of SLOTS slots (plain stores / loads: the L2 is write-back), 2 = the same with sc1 (write-through /
// memory-side loads: what the shipping kernel does through the Infinity Cache), 3 = a slot of its own per tile (a 1 GiB scratch: through memory).
// Prints JSON lines: algorithmic GB/s = (bytes read from the input + bytes written to the output) / time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ inline __amdgpu_buffer_rsrc_t rsrc(const void* p) {
	const uint64_t a = (uint64_t)p;
	const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
	return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), (short)0, 0x7FFFFFF0, 0x00020000);
}
__device__ inline uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u; }
constexpr uint32_t ROWS = 256, SEG = 128, TILE = ROWS * SEG; // 32 KiB

// transforms of n0 x n1 points (n0 = ROWS); tile t of a transform = columns [16 t, 16 t + 16)
template <int MODE, int RAUX, int WAUX>
__global__ void __launch_bounds__(256) k_ring(const char* in, char* out, char* ring, uint32_t n1, uint64_t tiles, uint32_t slots, uint32_t* slotCtr) {
	const uint32_t tid = threadIdx.x, seg = tid >> 3, part = tid & 7u; // 8 lanes per 128-byte segment, 32 segments per instruction
	const uint32_t xcd = xcc_id();
	__shared__ uint32_t sSlot;
	const uint32_t tilesPerT = n1 / 16u;
	for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
		const uint64_t tr = t / tilesPerT; const uint32_t ti = (uint32_t)(t % tilesPerT);
		const char* src = in + tr * (uint64_t)ROWS * n1 * 8u + (uint64_t)ti * SEG;
		char* dst = out + tr * (uint64_t)ROWS * n1 * 8u + (uint64_t)ti * SEG;
		const __amdgpu_buffer_rsrc_t rs = rsrc(src), rd = rsrc(dst);
		u32x4 v[8];
#pragma unroll
		for (int j = 0; j < 8; j++) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (seg + 32u * j) * n1 * 8u + part * 16u, 0, 2); // nt
		if constexpr (MODE != 0) {
			if (tid == 0) sSlot = MODE == 3 ? 0u : atomicAdd(slotCtr + 32u * xcd, 1u) % slots;
			__syncthreads();
			char* slot = MODE == 3 ? ring + t * (uint64_t)TILE : ring + ((uint64_t)xcd * slots + sSlot) * TILE;
			const __amdgpu_buffer_rsrc_t rr = rsrc(slot);
#pragma unroll
			for (int j = 0; j < 8; j++) __builtin_amdgcn_raw_buffer_store_b128(v[j], rr, tid * 16u + j * 4096u, 0, WAUX);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__syncthreads();
			// read it back with the B tile's access pattern: 128-byte segments, the segments of one instruction 2 KiB apart
#pragma unroll
			for (int j = 0; j < 8; j++) { const uint32_t sg = seg + 32u * j; v[j] = __builtin_amdgcn_raw_buffer_load_b128(rr, ((sg % 16u) * 16u + sg / 16u) * SEG + part * 16u, 0, RAUX); } // 128-byte segments 2 KiB apart
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__syncthreads(); // (the slot may be handed to its next tenant)
		}
#pragma unroll
		for (int j = 0; j < 8; j++) __builtin_amdgcn_raw_buffer_store_b128(v[j], rd, (seg + 32u * j) * n1 * 8u + part * 16u, 0, 2); // nt
	}
}

template <int MODE, int RAUX, int WAUX> static void run(const char* label, const char* in, char* out, char* ring, uint32_t n1, uint64_t bytes, uint32_t slots, uint32_t* ctr, int wgPerCu) {
	const uint64_t tiles = bytes / TILE;
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const int grid = 256 * wgPerCu;
	for (int rep = 0; rep < 2; rep++) {
		CK(hipMemset(ctr, 0, 4096));
		CK(hipEventRecord(e0));
		for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k_ring<MODE, RAUX, WAUX>), dim3(grid), dim3(256), 0, 0, in, out, ring, n1, tiles, slots, ctr);
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
		if (rep == 1) printf("{\"probe\": \"%s\", \"n1\": %u, \"slots_per_xcd\": %u, \"ring_KiB_per_xcd\": %u, \"wg_per_cu\": %d, \"ms\": %.4f, \"alg_GBps\": %.1f}\n", label, n1, slots, slots * TILE / 1024, wgPerCu, ms, 2.0 * bytes / (ms * 1e-3) / 1e9);
	}
}

int main() {
	const uint64_t bytes = 1ull << 30;
	char *in, *out, *ring; uint32_t* ctr;
	CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMalloc(&ring, bytes)); CK(hipMalloc(&ctr, 4096));
	CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes)); CK(hipMemset(ring, 0, bytes));
	for (uint32_t n1 : {256u, 512u}) { // 2^16 = 256 x 256, 2^17 = 256 x 512
		for (int wg : {1, 2, 4}) {
			run<0, 0, 0>("tile copy, no ring", in, out, ring, n1, bytes, 64, ctr, wg);
			for (uint32_t slots : {32u, 64u, 128u}) {
				run<1, 0, 0>("XCD-private ring in L2, plain stores and loads", in, out, ring, n1, bytes, slots, ctr, wg);
				run<1, 0, 2>("XCD-private ring in L2, nt stores", in, out, ring, n1, bytes, slots, ctr, wg);
			}
			run<2, 16, 16>("XCD-private ring, sc1 stores and loads (memory side: the shipping kernel's policy)", in, out, ring, n1, bytes, 64, ctr, wg);
			run<3, 0, 0>("a scratch slot per tile (1 GiB: through memory)", in, out, ring, n1, bytes, 64, ctr, wg);
		}
	}
	return 0;
}
