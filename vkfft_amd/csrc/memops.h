// Global-memory access for the hand-specialised kernels: CDNA buffer addressing.
// A tile is addressed as  wave-uniform base (128-bit buffer resource in SGPRs) + wave-uniform scalar offset
// + ONE 32-bit per-lane byte offset, instead of a 64-bit address per element: ~2 VGPRs and 2-4 VALU ops
// less per element, and lanes outside a partial tile are simply given an out-of-range offset (loads return 0,
// stores are dropped by the hardware range check) — no divergent branches.
#pragma once
#include "common.h"

#if defined(VKFFT_HOSTEMU)
#define VKFFT_WAVE_SYNC() hostemu::wave_sync()
#define VKFFT_SYNC() __syncthreads()
#define VKFFT_SYNC_RAW() __syncthreads()
#define VKFFT_OPAQUE_ZERO(z) uint32_t z = 0
#define VKFFT_SCHED_FENCE() do { } while (0)
#define VKFFT_PIN(x) do { } while (0)
#else
// pins a value where it is computed: the optimiser may neither sink the instructions that produce it nor hoist later uses above this point
#define VKFFT_PIN(x) asm volatile("" : "+v"(x))
// hides a (wave-uniform) pointer's provenance from the optimiser: stops loop-invariant twiddle loads from being
// hoisted out of the persistent tile loop and pinned in dozens of VGPRs
#define VKFFT_OPAQUE_ZERO(z) uint32_t z = 0; asm volatile("" : "+s"(z))
// keeps the instruction scheduler from hoisting every load of an unrolled gather loop above the first use (register pressure)
#define VKFFT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// Workgroup barrier that first waits for this wave's own LDS operations.  __syncthreads() alone fences the "local" address space only,
// for which the compiler emits no wait (it takes LDS operations of all waves as totally ordered): at the back edge of the fused kernel's
// persistent loop the barrier then followed two ds_writes of thread 0 directly, and on MI355X other waves occasionally passed the barrier
// and read the previous contents (a stale ticket: 2-29 wrong transforms in 300 under unbalanced queues, none with the wait).
#define VKFFT_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __syncthreads(); } while (0)
// The same barrier without the compiler's fence: the wave's own LDS operations are waited for, vector-memory operations are NOT (a fence would
// drain them).  For kernels that keep LDS-DMA transfers (memops: gb_dma16) and stores in flight across barriers and count them themselves.
#define VKFFT_SYNC_RAW() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
// orders this wave's LDS writes before its later LDS reads without an s_barrier
#define VKFFT_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
#endif

namespace vkfft_mi355x {

constexpr uint32_t kGbRange = 0x7FFFFFF0u;  // bytes addressable through one resource
constexpr uint32_t kGbInvalid = 0x7FFFFFF8u; // per-lane offset that is always out of range

#if defined(VKFFT_HOSTEMU)
struct GBuf { char* base; };
inline GBuf make_gbuf(const void* p) { return {(char*)p}; }
template <typename T> inline cx<T> gb_load(GBuf b, uint32_t voff, uint32_t soff) {
	if (voff >= kGbRange) return cx<T>{(T)0, (T)0};
	return *(const cx<T>*)(b.base + (uint64_t)voff + soff);
}
template <typename T> inline void gb_store(GBuf b, uint32_t voff, uint32_t soff, cx<T> v) {
	if (voff >= kGbRange) return;
	*(cx<T>*)(b.base + (uint64_t)voff + soff) = v;
}
template <typename T> inline T gb_load_real(GBuf b, uint32_t voff, uint32_t soff) {
	if (voff >= kGbRange) return (T)0;
	return *(const T*)(b.base + (uint64_t)voff + soff);
}
template <typename T> inline void gb_store_real(GBuf b, uint32_t voff, uint32_t soff, T v) {
	if (voff >= kGbRange) return;
	*(T*)(b.base + (uint64_t)voff + soff) = v;
}
template <typename T> struct Real4 { T x, y, z, w; };
template <typename T> inline Real4<T> gb_load_real4(GBuf b, uint32_t voff, uint32_t soff) {
	if (voff >= kGbRange) return Real4<T>{(T)0, (T)0, (T)0, (T)0};
	const T* q = (const T*)(b.base + (uint64_t)voff + soff);
	return Real4<T>{q[0], q[1], q[2], q[3]};
}
template <typename T> inline void gb_store_real4(GBuf b, uint32_t voff, uint32_t soff, Real4<T> v) {
	if (voff >= kGbRange) return;
	T* q = (T*)(b.base + (uint64_t)voff + soff);
	q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
}
// cache-policy variants (AUX = sc0 | nt<<1 | sc1<<4 of the buffer instructions): plain accesses on the emulator
template <typename T, int AUX> inline cx<T> gb_load_x(GBuf b, uint32_t voff, uint32_t soff) { return gb_load<T>(b, voff, soff); }
template <typename T, int AUX> inline void gb_store_x(GBuf b, uint32_t voff, uint32_t soff, cx<T> v) { gb_store<T>(b, voff, soff, v); }
template <typename T, int AUX> inline void gb_store2_x(GBuf b, uint32_t voff, cx<T> v0, cx<T> v1) { gb_store<T>(b, voff, 0, v0); gb_store<T>(b, voff + (uint32_t)sizeof(cx<T>), 0, v1); }
template <typename T, int E> inline void gb_landed(cx<T>*) { }
template <typename T, int AUX> inline void gb_load2_x(GBuf b, uint32_t voff, uint32_t soff, cx<T>& v0, cx<T>& v1) { v0 = gb_load<T>(b, voff, soff); v1 = gb_load<T>(b, voff + (voff >= kGbRange ? 0u : (uint32_t)sizeof(cx<T>)), soff); }
// LDS-DMA (device arm below): on the emulator every work-item copies its own 16 bytes at once, the waits are empty
struct GDma { const char* base; };
inline GDma make_gdma(const void* p) { return {(const char*)p}; }
template <int AUX> inline void gb_dma16(void* ldsWaveBase, GDma b, uint32_t voff, uint32_t soff) {
	char* d = (char*)ldsWaveBase + (threadIdx.x & 63u) * 16u;
	if (voff >= kGbRange) memset(d, 0, 16); else memcpy(d, b.base + (uint64_t)voff + soff, 16);
}
template <int N> inline void gb_wait_vm() { }
#else
typedef unsigned int vk_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int vk_u32x4 __attribute__((ext_vector_type(4)));
struct GBuf { __amdgpu_buffer_rsrc_t r; };
__device__ inline GBuf make_gbuf(const void* p) {
	// the base is wave-uniform by construction (kernel arguments + blockIdx); readfirstlane makes that provable,
	// otherwise the compiler wraps every buffer op in a waterfall loop
	const uint64_t a = (uint64_t)p;
	const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
	GBuf b;
	b.r = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), (short)0, (int)kGbRange, 0x00020000);
	return b;
}
template <typename T> __device__ inline cx<T> gb_load(GBuf b, uint32_t voff, uint32_t soff);
template <> __device__ inline cx<float> gb_load<float>(GBuf b, uint32_t voff, uint32_t soff) {
	vk_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(b.r, voff, soff, 0);
	return cx<float>{__uint_as_float(t.x), __uint_as_float(t.y)};
}
template <> __device__ inline cx<double> gb_load<double>(GBuf b, uint32_t voff, uint32_t soff) {
	vk_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(b.r, voff, soff, 0);
	return cx<double>{__hiloint2double((int)t.y, (int)t.x), __hiloint2double((int)t.w, (int)t.z)};
}
template <typename T> __device__ inline void gb_store(GBuf b, uint32_t voff, uint32_t soff, cx<T> v);
template <> __device__ inline void gb_store<float>(GBuf b, uint32_t voff, uint32_t soff, cx<float> v) {
	vk_u32x2 t; t.x = __float_as_uint(v.x); t.y = __float_as_uint(v.y);
	__builtin_amdgcn_raw_buffer_store_b64(t, b.r, voff, soff, 0);
}
template <typename T> __device__ inline T gb_load_real(GBuf b, uint32_t voff, uint32_t soff);
template <> __device__ inline float gb_load_real<float>(GBuf b, uint32_t voff, uint32_t soff) {
	return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b.r, voff, soff, 0));
}
template <> __device__ inline double gb_load_real<double>(GBuf b, uint32_t voff, uint32_t soff) {
	vk_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(b.r, voff, soff, 0);
	return __hiloint2double((int)t.y, (int)t.x);
}
template <typename T> __device__ inline void gb_store_real(GBuf b, uint32_t voff, uint32_t soff, T v);
template <> __device__ inline void gb_store_real<float>(GBuf b, uint32_t voff, uint32_t soff, float v) {
	__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), b.r, voff, soff, 0);
}
template <> __device__ inline void gb_store_real<double>(GBuf b, uint32_t voff, uint32_t soff, double v) {
	vk_u32x2 t; t.x = (unsigned)__double2loint(v); t.y = (unsigned)__double2hiint(v);
	__builtin_amdgcn_raw_buffer_store_b64(t, b.r, voff, soff, 0);
}
template <> __device__ inline void gb_store<double>(GBuf b, uint32_t voff, uint32_t soff, cx<double> v) {
	vk_u32x4 t;
	t.x = (unsigned)__double2loint(v.x); t.y = (unsigned)__double2hiint(v.x);
	t.z = (unsigned)__double2loint(v.y); t.w = (unsigned)__double2hiint(v.y);
	// gfx950 erratum-like behaviour (measured): a 128-bit buffer store whose soffset is an SGPR still
	// reads its upper 64 data bits a cycle late, but the compiler's hazard recogniser only pads the soffset-less form —
	// the next VALU write to those VGPRs then corrupts the imaginary half under memory back-pressure.  Folding the
	// scalar offset into the per-lane offset keeps the store in the form the compiler protects with s_nop.
	__builtin_amdgcn_raw_buffer_store_b128(t, b.r, voff + soff, 0, 0);
}
// four consecutive reals (one 128-bit access for fp32, two for fp64; dword alignment is enough for buffer accesses)
template <typename T> struct Real4 { T x, y, z, w; };
template <typename T> __device__ inline Real4<T> gb_load_real4(GBuf b, uint32_t voff, uint32_t soff);
template <> __device__ inline Real4<float> gb_load_real4<float>(GBuf b, uint32_t voff, uint32_t soff) {
	vk_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(b.r, voff, soff, 0);
	return Real4<float>{__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
}
template <> __device__ inline Real4<double> gb_load_real4<double>(GBuf b, uint32_t voff, uint32_t soff) {
	const cx<double> a = gb_load<double>(b, voff, soff), c = gb_load<double>(b, voff + 16u, soff);
	return Real4<double>{a.x, a.y, c.x, c.y};
}
template <typename T> __device__ inline void gb_store_real4(GBuf b, uint32_t voff, uint32_t soff, Real4<T> v);
template <> __device__ inline void gb_store_real4<float>(GBuf b, uint32_t voff, uint32_t soff, Real4<float> v) {
	vk_u32x4 t; t.x = __float_as_uint(v.x); t.y = __float_as_uint(v.y); t.z = __float_as_uint(v.z); t.w = __float_as_uint(v.w);
	__builtin_amdgcn_raw_buffer_store_b128(t, b.r, voff + soff, 0, 0); // soffset folded: see gb_store<double>
}
template <> __device__ inline void gb_store_real4<double>(GBuf b, uint32_t voff, uint32_t soff, Real4<double> v) {
	gb_store<double>(b, voff, soff, cx<double>{v.x, v.y});
	gb_store<double>(b, voff + 16u, soff, cx<double>{v.z, v.w});
}
// cache-policy variants: AUX = sc0 | nt<<1 | sc1<<4 of the buffer instructions (sc1 = agent scope: write-through stores,
// loads served from the memory side; nt = streaming hint).  Used by the fused Four-Step kernel for its on-die scratch ring.
template <typename T, int AUX> __device__ inline cx<T> gb_load_x(GBuf b, uint32_t voff, uint32_t soff) {
	if constexpr (sizeof(T) == 4) {
		vk_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(b.r, voff, soff, AUX);
		return cx<T>{__uint_as_float(t.x), __uint_as_float(t.y)};
	} else {
		vk_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(b.r, voff, soff, AUX);
		return cx<T>{__hiloint2double((int)t.y, (int)t.x), __hiloint2double((int)t.w, (int)t.z)};
	}
}
template <typename T, int AUX> __device__ inline void gb_store_x(GBuf b, uint32_t voff, uint32_t soff, cx<T> v) {
	if constexpr (sizeof(T) == 4) {
		vk_u32x2 t; t.x = __float_as_uint(v.x); t.y = __float_as_uint(v.y);
		__builtin_amdgcn_raw_buffer_store_b64(t, b.r, voff, soff, AUX);
	} else {
		vk_u32x4 t;
		t.x = (unsigned)__double2loint(v.x); t.y = (unsigned)__double2hiint(v.x);
		t.z = (unsigned)__double2loint(v.y); t.w = (unsigned)__double2hiint(v.y);
		__builtin_amdgcn_raw_buffer_store_b128(t, b.r, voff + soff, 0, AUX); // soffset folded: see gb_store<double>
	}
}
// makes the compiler wait (counted s_waitcnt) until the loads that produce v[0..E) have returned, nothing more
template <typename T, int E> __device__ inline void gb_landed(cx<T>* v) {
#pragma unroll
	for (int m = 0; m < E; m++) asm volatile("" : "+v"(v[m].x), "+v"(v[m].y));
}
// two consecutive fp32 complex values in one 128-bit load
template <typename T, int AUX> __device__ inline void gb_load2_x(GBuf b, uint32_t voff, uint32_t soff, cx<T>& v0, cx<T>& v1) {
	static_assert(sizeof(T) == 4, "pairs of fp32 complex only");
	const vk_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(b.r, voff, soff, AUX);
	v0 = cx<T>{__uint_as_float(t.x), __uint_as_float(t.y)}; v1 = cx<T>{__uint_as_float(t.z), __uint_as_float(t.w)};
}
// two consecutive fp32 complex values in one 128-bit store
template <typename T, int AUX> __device__ inline void gb_store2_x(GBuf b, uint32_t voff, cx<T> v0, cx<T> v1) {
	static_assert(sizeof(T) == 4, "pairs of fp32 complex only");
	vk_u32x4 t; t.x = __float_as_uint(v0.x); t.y = __float_as_uint(v0.y); t.z = __float_as_uint(v1.x); t.w = __float_as_uint(v1.y);
	__builtin_amdgcn_raw_buffer_store_b128(t, b.r, voff, 0, AUX);
}
// ---- LDS-DMA: buffer_load_dwordx4 ... lds.  One wave-instruction moves 64 x 16 bytes from per-lane global addresses (resource base + soff + voff)
// to ONE contiguous KiB of LDS at ldsWaveBase + lane * 16 (the destination is wave-uniform base + lane-linear; measured on gfx950 with
// tools/probe_dma.hip: any LDS offset up to 160 KiB, out-of-range lanes write zeros).  No VGPR holds the data and the transfer stays in flight
// across barriers: what lets a persistent kernel fetch its next tile while the current one computes.  The statement is inline assembly on purpose:
// an LDS-DMA issued through the compiler's builtin is a pending LDS write to its wait-count pass, which then drains the vector-memory queue
// (vmcnt(0)) before the next LDS read or barrier — the opposite of what the transfer is for.  The caller counts: the queue is in order, so
// gb_wait_vm<N>() = "all but the N youngest vector-memory instructions of this wave have completed".
struct GDma { vk_u32x4 r; };
__device__ inline GDma make_gdma(const void* p) {
	const uint64_t a = (uint64_t)p;
	GDma b;
	b.r.x = __builtin_amdgcn_readfirstlane((uint32_t)a); b.r.y = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) & 0xffffu; b.r.z = kGbRange; b.r.w = 0x00020000u;
	return b;
}
template <int AUX> __device__ inline void gb_dma16(void* ldsWaveBase, GDma b, uint32_t voff, uint32_t soff) {
	const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ldsWaveBase);
	const uint32_t so = __builtin_amdgcn_readfirstlane(soff);
	uint32_t keep;
	static_assert(AUX == 0 || AUX == 2 || AUX == 16, "policy: default, nt, sc1");
	if constexpr (AUX == 0) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(la), "v"(voff), "s"(b.r), "s"(so) : "memory");
	else if constexpr (AUX == 2) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen nt lds\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(la), "v"(voff), "s"(b.r), "s"(so) : "memory");
	else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen sc1 lds\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(la), "v"(voff), "s"(b.r), "s"(so) : "memory");
}
template <int N> __device__ inline void gb_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
#endif

} // namespace vkfft_mi355x
