// Packed-pair power-of-two Stockham stages (round 5): the register-lean stages of kernel_pow2_lean.h with the TWO ADJACENT COLUMNS of a
// thread kept as structure-of-arrays register pairs, so that every butterfly runs on CDNA's packed fp32 pipe.
//
// Why: the ISA of the round-4 fused kernels (hipcc -O3, gfx950) spends 2 800 vector instructions per thread and ticket at 2^20, of which 765 are
// v_mov (pairing registers for the packed operations the vectoriser found on its own), and a complex multiply costs 6-8 instructions
// (two v_pk_mul, two scalar add/sub, two to four moves).  The tile phases of those kernels are VALU / LDS-issue bound (profiles/r04_fused_phase_profile_*).
// Here a thread's value at point m is   re[m] = (Re col c, Re col c+1),  im[m] = (Im col c, Im col c+1)   — two even-aligned VGPR pairs — and
//   * a complex add / subtract of both columns is 2 v_pk_add_f32,
//   * a multiply of both columns by a SHARED twiddle (stage twiddles, literal butterfly twiddles) is 2 v_pk_mul_f32 + 2 v_pk_fma_f32, the twiddle's
//     real / imaginary part broadcast by op_sel (no register is spent on the broadcast),
//   * a multiply by -i is a renaming folded into the neighbouring add (neg modifiers),
//   * the plane-split LDS exchange writes and reads the pairs as they are (8-byte accesses, no packing moves),
//   * the ring of the fused Four-Step kernels holds 16-byte units (Re p0, Re p1, Im p0, Im p1) of two consecutive points — the layout is private to the
//     kernel — so the turned tile is stored from, and the next phase's tile loaded into, register pairs directly.
// The reference emits scalar complex arithmetic for every backend (vkFFT_MathUtils.h: PfMul / PfFMA on .x/.y); packed issue is specific to CDNA.
#pragma once
#include "kernel_pow2_lean.h"
#include <utility>

namespace vkfft_mi355x {

#if defined(VKFFT_HOSTEMU)
template <typename T> struct pk2 { T x, y; };
template <typename T> inline pk2<T> operator+(pk2<T> a, pk2<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> inline pk2<T> operator-(pk2<T> a, pk2<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T> inline pk2<T> operator*(pk2<T> a, pk2<T> b) { return {a.x * b.x, a.y * b.y}; }
template <typename T> inline pk2<T> operator-(pk2<T> a) { return {-a.x, -a.y}; }
template <typename V> inline V pk_xx(V a) { return V{a.x, a.x}; }
template <typename V> inline V pk_yy(V a) { return V{a.y, a.y}; }
#define VKFFT_PIN2(x) do { } while (0)
#else
template <typename T> struct pk2_of;
template <> struct pk2_of<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct pk2_of<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <typename T> using pk2 = typename pk2_of<T>::type;
template <typename V> __device__ inline V pk_xx(V a) { return __builtin_shufflevector(a, a, 0, 0); }
template <typename V> __device__ inline V pk_yy(V a) { return __builtin_shufflevector(a, a, 1, 1); }
#define VKFFT_PIN2(x) asm volatile("" : "+v"(x))
#endif
template <typename T> __host__ __device__ inline pk2<T> pk_splat(T s) { return pk2<T>{s, s}; }

// development build only (MODE bit 2, -DVKFFT_MI355X_DEV): per-phase cycle sums of thread 0 in LDS (registers would change what is measured), flushed at exit
#if defined(VKFFT_HOSTEMU)
#define VKFFT_PKPROF_DECL do { } while (0)
#define VKFFT_PKPROF(i) do { } while (0)
#define VKFFT_PKPROF_FLUSH() do { } while (0)
#else
#define VKFFT_PKPROF_DECL __shared__ unsigned long long spc[12]; unsigned long long ptk = 0; if constexpr ((MODE & 4) != 0) { if (threadIdx.x < 12) spc[threadIdx.x] = 0; ptk = __builtin_readcyclecounter(); }
#define VKFFT_PKPROF(i) do { if constexpr ((MODE & 4) != 0) { if (threadIdx.x == 0) { const unsigned long long now = __builtin_readcyclecounter(); spc[i] += now - ptk; ptk = now; } } } while (0)
#define VKFFT_PKPROF_FLUSH() do { if constexpr ((MODE & 4) != 0) { if (threadIdx.x == 0 && p.prof) { for (int i = 0; i < 12; i++) p.prof[(size_t)blockIdx.x * 12 + i] = spc[i]; } } } while (0)
#endif

// the same point of two adjacent columns (or two consecutive points of one column)
template <typename T> struct cxp { pk2<T> re, im; };
template <typename T> __device__ inline cxp<T> pcadd(cxp<T> a, cxp<T> b) { return {a.re + b.re, a.im + b.im}; }
template <typename T> __device__ inline cxp<T> pcsub(cxp<T> a, cxp<T> b) { return {a.re - b.re, a.im - b.im}; }
// both values times ONE complex factor w = (w.x, w.y)
template <typename T> __device__ inline cxp<T> pcmul1(cxp<T> a, pk2<T> w) {
	const pk2<T> wx = pk_xx(w), wy = pk_yy(w);
	return {a.re * wx - a.im * wy, a.re * wy + a.im * wx};
}
// each value times its own factor, the factors as a pair of real parts and a pair of imaginary parts
template <typename T> __device__ inline cxp<T> pcmul2(cxp<T> a, pk2<T> wre, pk2<T> wim) { return {a.re * wre - a.im * wim, a.re * wim + a.im * wre}; }

// ONE complex value held as the pair (x, y), times another: (a.x b.x - a.y b.y, a.x b.y + a.y b.x) = a.xx * b + a.yy * (-b.y, b.x)
#if defined(VKFFT_HOSTEMU)
template <typename T> inline pk2<T> pk_cmul_aos(pk2<T> a, pk2<T> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
#else
template <typename T> __device__ inline pk2<T> pk_cmul_aos(pk2<T> a, pk2<T> b) {
	const pk2<T> br = __builtin_shufflevector(b, -b, 3, 0); // (-b.y, b.x)
	return pk_xx(a) * b + pk_yy(a) * br;
}
#endif

// radix-R decimation-in-frequency butterfly, in place, layer by layer; X[k] ends up at x[bitrev(k)]  (pow2_dif_inplace on pairs)
template <int R, typename T> __device__ inline void pk_dif_inplace(cxp<T>* x) {
#pragma unroll
	for (int h = R / 2; h >= 1; h >>= 1) {
#pragma unroll
		for (int blk = 0; blk < R; blk += 2 * h) {
#pragma unroll
			for (int j = 0; j < h; j++) {
				const cxp<T> a = x[blk + j], b = x[blk + j + h];
				x[blk + j] = pcadd(a, b);
				const int kk = j * (16 / h); // w_{2h}^j as a power of w_32
				if (kk == 0) x[blk + j + h] = pcsub(a, b);
				else if (kk == 8) x[blk + j + h] = cxp<T>{a.im - b.im, b.re - a.re}; // -i (a - b)
				else {
					const cxp<T> d = pcsub(a, b);
					const pk2<T> c = pk_splat<T>((T)pow2_cos32(kk)), s = pk_splat<T>((T)pow2_sin32(kk)); // w = c - i s
					x[blk + j + h] = cxp<T>{d.re * c + d.im * s, d.im * c - d.re * s};
				}
			}
		}
		VKFFT_SCHED_FENCE();
	}
}

// stage twiddle as a pair (x, y)
template <typename T, typename TW> __device__ inline pk2<T> pk_tw(const TW& lut, uint32_t s, uint32_t constOff) { const cx<T> w = lut.get(s, constOff); return pk2<T>{w.x, w.y}; }

// The butterflies of stage SI on the thread's E points of two adjacent columns (register m <-> point tau + m*TPF), results in natural order.
template <typename T, typename SCH, int SI, int TPF, typename TW, int TWG>
__device__ inline void pk_lean_butterflies(cxp<T>* v, const TW lut, const uint32_t tau) {
	constexpr int LOGE = SCH::LOGE, E = 1 << LOGE;
	constexpr int LOGR = SCH::bits[SI], R = 1 << LOGR, NB = E / R;
	constexpr int LOGS = SCH::logS(SI), S = 1 << LOGS;
#pragma unroll
	for (int b = 0; b < NB; b++) {
		if constexpr (SI > 0) {
			const uint32_t s = (tau + b * TPF) & (S - 1);
			constexpr int LO = SCH::lutOff(SI);
#pragma unroll
			for (int i0 = 1; i0 < R; i0 += TWG) {
				const int i1 = i0 + TWG < R ? i0 + TWG : R;
				pk2<T> w[TWG];
#pragma unroll
				for (int i = i0; i < i1; i++) w[i - i0] = pk_tw<T>(lut, s, (uint32_t)(LO + (i - 1) * S));
				VKFFT_SCHED_FENCE();
#pragma unroll
				for (int i = i0; i < i1; i++) { cxp<T>& q = v[b + i * NB]; q = pcmul1(q, w[i - i0]); VKFFT_PIN2(q.re); VKFFT_PIN2(q.im); }
				VKFFT_SCHED_FENCE();
			}
		}
		cxp<T> x[R];
#pragma unroll
		for (int i = 0; i < R; i++) x[i] = v[b + i * NB];
		pk_dif_inplace<R, T>(x);
#pragma unroll
		for (int k = 0; k < R; k++) v[b + k * NB] = x[pow2_bitrev(k, LOGR)];
	}
}

// one real-valued plane exchange of stage SI's results (PART 0: real parts, 1: imaginary parts); `plane` points at the thread's first column
template <typename T, typename SCH, int SI, int TPF, int TC, int PART>
__device__ inline void pk_lean_exchange_part(cxp<T>* v, T* plane, const uint32_t tau) {
	constexpr int LOGE = SCH::LOGE, E = 1 << LOGE, P = SCH::bits[0];
	constexpr int LOGR = SCH::bits[SI], R = 1 << LOGR, NB = E / R;
	constexpr int LOGS = SCH::logS(SI), S = 1 << LOGS;
	static_assert(TC > 0 && TC % 2 == 0, "column tiles of even width");
#pragma unroll
	for (int b = 0; b < NB; b++) {
		const uint32_t t = tau + b * TPF;
		const uint32_t s = t & (S - 1);
		const uint32_t ob = ((t - s) << LOGR) + s;
		T* const wp = plane + pow2_lean_slot<TC, P>(ob);
#pragma unroll
		for (int k = 0; k < R; k++) *(pk2<T>*)(wp + pow2_lean_step<TC, P>(k * S)) = PART ? v[b + k * NB].im : v[b + k * NB].re;
	}
	VKFFT_SYNC();
	const T* const rp = plane + pow2_lean_slot<TC, P>(tau);
#pragma unroll
	for (int m = 0; m < E; m++) {
		const pk2<T> r = *(const pk2<T>*)(rp + pow2_lean_step<TC, P>(m * TPF));
		if (PART) v[m].im = r; else v[m].re = r;
	}
}

// all stages of SCH on the thread's registers (contract of pow2_lean_stages: the plane is free on entry; on exit the last exchange's reads may
// still be in flight in other waves)
template <typename T, typename SCH, int SI, int TPF, int TC, typename TW, int TWG>
__device__ inline void pk_lean_stages(cxp<T>* v, T* plane, const TW lut, const uint32_t tau) {
	pk_lean_butterflies<T, SCH, SI, TPF, TW, TWG>(v, lut, tau);
	if constexpr (SI + 1 < SCH::NS) {
		pk_lean_exchange_part<T, SCH, SI, TPF, TC, 0>(v, plane, tau);
		VKFFT_SYNC();
		pk_lean_exchange_part<T, SCH, SI, TPF, TC, 1>(v, plane, tau);
		if constexpr (SI + 2 < SCH::NS) VKFFT_SYNC(); // another exchange will overwrite the plane
		pk_lean_stages<T, SCH, SI + 1, TPF, TC, TW, TWG>(v, plane, lut, tau);
	}
}

// Column tile in registers (v[m] = point tau + m*TPF of columns c, c + 1) -> per-column contiguous order through the plane laid out [column][point]
// (pow2_lean_transpose with a pitch chosen per tile width).  Item i of thread tid is the point pair (2kp, 2kp + 1) of column cc with idx = tid + i*NT,
// kp = idx % (L/2), cc = idx / (L/2):  r[i].re = (Re 2kp, Re 2kp+1), r[i].im = (Im 2kp, Im 2kp+1) — the ring unit.  The plane must be free on entry;
// it is free again after the caller's next barrier.
template <typename T, int L, int E, int TPF, int TC, int NT>
__device__ inline void pk_lean_transpose(const cxp<T>* v, cxp<T>* r, T* plane, const uint32_t tid, const uint32_t c, const uint32_t tau) {
	// pitch: a wave's 4-byte writes are TC/2 column pairs x 128/TC points; with PT = L + 64/TC the column pairs start 128/TC banks apart: all 32 banks, two lanes
	// each (L + 4 for 32-column tiles put 16 column pairs on 4 bank offsets: four lanes per bank, measured in the 2^19 phase profile, profiles/r05_*)
	constexpr int PT = L + 64 / TC;
	static_assert(E * NT * 2 == L * TC && (PT % 2) == 0, "two columns per thread");
#pragma unroll
	for (int part = 0; part < 2; part++) {
		if (part) VKFFT_SYNC();
		T* const w0 = plane + c * PT + tau;
#pragma unroll
		for (int m = 0; m < E; m++) { const pk2<T> q = part ? v[m].im : v[m].re; w0[m * TPF] = q.x; w0[PT + m * TPF] = q.y; }
		VKFFT_SYNC();
#pragma unroll
		for (int i = 0; i < E; i++) {
			const uint32_t idx = tid + i * NT;
			const uint32_t kp = idx % (L / 2), cc = idx / (L / 2);
			const pk2<T> pr = *(const pk2<T>*)(plane + cc * PT + 2u * kp);
			if (part) r[i].im = pr; else r[i].re = pr;
		}
	}
}

// ---- 16-byte global accesses of pairs -------------------------------------------------------------------------------------------------------
#if defined(VKFFT_HOSTEMU)
// two adjacent complex values (x0, y0, x1, y1) as they travel; pk_from_aos turns them into the pair form once they have landed
template <typename T> struct pk4 { T x, y, z, w; };
template <typename T, int AUX> inline pk4<T> gb_load_aos2(GBuf b, uint32_t voff, uint32_t soff) {
	if (voff >= kGbRange) return pk4<T>{(T)0, (T)0, (T)0, (T)0};
	const T* q = (const T*)(b.base + (uint64_t)voff + soff);
	return pk4<T>{q[0], q[1], q[2], q[3]};
}
template <typename T> inline cxp<T> pk_from_aos(pk4<T> t) { return cxp<T>{{t.x, t.z}, {t.y, t.w}}; }
template <typename T, int AUX> inline void gb_store_aos2(GBuf b, uint32_t voff, cxp<T> v) {
	if (voff >= kGbRange) return;
	T* q = (T*)(b.base + (uint64_t)voff);
	q[0] = v.re.x; q[1] = v.im.x; q[2] = v.re.y; q[3] = v.im.y;
}
template <typename T, int E> inline void gb_landed_raw(pk4<T>*) { }
// the ring unit (re0, re1, im0, im1)
template <typename T, int AUX> inline cxp<T> gb_load_soa2(GBuf b, uint32_t voff, uint32_t soff) {
	if (voff >= kGbRange) return cxp<T>{{(T)0, (T)0}, {(T)0, (T)0}};
	const T* q = (const T*)(b.base + (uint64_t)voff + soff);
	return cxp<T>{{q[0], q[1]}, {q[2], q[3]}};
}
template <typename T, int AUX> inline void gb_store_soa2(GBuf b, uint32_t voff, cxp<T> v) {
	if (voff >= kGbRange) return;
	T* q = (T*)(b.base + (uint64_t)voff);
	q[0] = v.re.x; q[1] = v.re.y; q[2] = v.im.x; q[3] = v.im.y;
}
template <typename T, int E> inline void gb_landed_pk(cxp<T>*) { }
#else
// two adjacent complex values (x0, y0, x1, y1) as they travel (one 128-bit register tuple); pk_from_aos turns them into the pair form once they have
// landed: ONE v_swap_b32 of the two middle registers, (x0, y0, x1, y1) -> (x0, x1 | y0, y1)
template <typename T> struct pk4_of;
template <> struct pk4_of<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <typename T> using pk4 = typename pk4_of<T>::type;
template <typename T, int AUX> __device__ inline pk4<T> gb_load_aos2(GBuf b, uint32_t voff, uint32_t soff) {
	static_assert(sizeof(T) == 4, "pairs of fp32 complex only");
	return __builtin_bit_cast(pk4<T>, __builtin_amdgcn_raw_buffer_load_b128(b.r, voff, soff, AUX));
}
template <typename T> __device__ inline cxp<T> pk_from_aos(pk4<T> t) {
	T a = t.y, b = t.z;
	asm("v_swap_b32 %0, %1" : "+v"(a), "+v"(b));
	return cxp<T>{pk2<T>{t.x, a}, pk2<T>{b, t.w}};
}
template <typename T, int AUX> __device__ inline void gb_store_aos2(GBuf b, uint32_t voff, cxp<T> v) {
	static_assert(sizeof(T) == 4, "pairs of fp32 complex only");
	T a = v.re.y, c = v.im.x;
	asm("v_swap_b32 %0, %1" : "+v"(a), "+v"(c));
	const pk4<T> t = {v.re.x, a, c, v.im.y};
	__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vk_u32x4, t), b.r, voff, 0, AUX);
}
template <typename T, int E> __device__ inline void gb_landed_raw(pk4<T>* v) {
#pragma unroll
	for (int m = 0; m < E; m++) asm volatile("" : "+v"(v[m]));
}
template <typename T, int AUX> __device__ inline cxp<T> gb_load_soa2(GBuf b, uint32_t voff, uint32_t soff) {
	static_assert(sizeof(T) == 4, "pairs of fp32 complex only");
	const vk_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(b.r, voff, soff, AUX);
	return cxp<T>{pk2<T>{__uint_as_float(t.x), __uint_as_float(t.y)}, pk2<T>{__uint_as_float(t.z), __uint_as_float(t.w)}};
}
template <typename T, int AUX> __device__ inline void gb_store_soa2(GBuf b, uint32_t voff, cxp<T> v) {
	static_assert(sizeof(T) == 4, "pairs of fp32 complex only");
	vk_u32x4 t; t.x = __float_as_uint(v.re.x); t.y = __float_as_uint(v.re.y); t.z = __float_as_uint(v.im.x); t.w = __float_as_uint(v.im.y);
	__builtin_amdgcn_raw_buffer_store_b128(t, b.r, voff, 0, AUX);
}
// makes the compiler wait (counted s_waitcnt) until the loads that produce v[0..E) have returned, nothing more
template <typename T, int E> __device__ inline void gb_landed_pk(cxp<T>* v) {
#pragma unroll
	for (int m = 0; m < E; m++) asm volatile("" : "+v"(v[m].re), "+v"(v[m].im));
}
#endif

// Four-Step twiddle of a column pair: point k = tau + m*TPF of column j gets w_N^(k j) (vkFFT_4step.h:31).  The even column's factor comes from the two-level
// table (pow2_fs_twiddle: 7 look-ups per 16 points, combined by products); the odd column's is that times w_N^k, and the pair of factors is formed directly
// in pair form from the row table = 16-byte entries (1, Re w_N^k, 0, Im w_N^k):
//   (Re w0, Re w0 w_N^k) = Re w0 * (1, Re) - Im w0 * (0, Im),   (Im w0, Im w0 w_N^k) = Im w0 * (1, Re) + Re w0 * (0, Im)
// In two steps: the look-ups are REQUESTED ahead of the stages that produce v (pk_fs_request: their L2 latency — two dependent levels, about 1 k cycles each in
// the round-5 phase profile, where this twiddle took 5.8 k cycles of a 44 k ticket — passes behind the stages) and combined when v is there (pk_fs_apply);
// the row table sits in LDS next to the stage twiddles (RowLds) where the budget allows, else it is read through L2 in chunks (RowGlobal).
template <typename T> struct RowGlobal {
	GBuf tab;
	__device__ inline cxp<T> get(uint32_t tau, uint32_t constK) const { return gb_load_soa2<T, 0>(tab, tau * 16u, constK * 16u); }
};
template <typename T> struct RowLds {
	const cx<T>* tab; // 2 complex slots per entry
	__device__ inline cxp<T> get(uint32_t tau, uint32_t constK) const {
#if defined(VKFFT_HOSTEMU)
		const T* q = (const T*)(tab + 2u * (tau + constK));
		return cxp<T>{{q[0], q[1]}, {q[2], q[3]}};
#else
		const pk4<T> t = *(const pk4<T>*)(tab + 2u * (tau + constK));
		return cxp<T>{pk2<T>{t.x, t.y}, pk2<T>{t.z, t.w}};
#endif
	}
};
template <typename T, int LOGE> struct PkFsTw {
	static constexpr int HIB = (LOGE + 1) / 2, LOB = LOGE - HIB, NA = 1 << HIB, NB = (1 << LOB) - 1;
	pk2<T> lo[NA + NB], hi[NA + NB]; // the two table levels of A[0..NA) and B[1..NB]
};
template <typename T, int LOGE, int TPF>
__device__ inline void pk_fs_request(PkFsTw<T, LOGE>& q, const GBuf gtab, const uint32_t fsLoBits, const uint32_t tau, const uint32_t colEven) {
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	constexpr int LOB = PkFsTw<T, LOGE>::LOB, NA = PkFsTw<T, LOGE>::NA, NB = PkFsTw<T, LOGE>::NB;
	const uint32_t loMask = (1u << fsLoBits) - 1u;
	const uint32_t hiBase = (loMask + 1u) * ES;
	auto ld = [&](uint32_t voff, uint32_t soff) { const cx<T> w = gb_load<T>(gtab, voff, soff); return pk2<T>{w.x, w.y}; };
#pragma unroll
	for (int j = 0; j < NA + NB; j++) {
		const uint32_t e = (j < NA ? tau + (uint32_t)((j << LOB) * TPF) : (uint32_t)((j - NA + 1) * TPF)) * colEven;
		q.lo[j] = ld((e & loMask) * ES, 0); q.hi[j] = ld((e >> fsLoBits) * ES, hiBase);
	}
}
// hiStep != nullptr: the points are k + 1024 ... of a 2048-point factor whose first half was requested with q: every factor times *hiStep = w_N^(1024 j)
template <typename T, int LOGE, int TPF, typename ROW>
__device__ inline void pk_fs_apply(cxp<T>* v, const PkFsTw<T, LOGE>& q, const ROW row, const uint32_t tau, const pk2<T>* hiStep = nullptr) {
	constexpr int E = 1 << LOGE;
	constexpr int LOB = PkFsTw<T, LOGE>::LOB, NA = PkFsTw<T, LOGE>::NA, NB = PkFsTw<T, LOGE>::NB;
	pk2<T> A[NA], B[NB + 1];
#pragma unroll
	for (int j = 0; j < NA; j++) { A[j] = pk_cmul_aos<T>(q.lo[j], q.hi[j]); if (hiStep) A[j] = pk_cmul_aos<T>(A[j], *hiStep); }
	B[0] = pk2<T>{(T)1, (T)0};
#pragma unroll
	for (int i = 1; i <= NB; i++) B[i] = pk_cmul_aos<T>(q.lo[NA + i - 1], q.hi[NA + i - 1]);
	constexpr int CH = 4; // row-table entries in flight at a time (16 registers: the scheduler would request all E of them at once)
#pragma unroll
	for (int m0 = 0; m0 < E; m0 += CH) {
		cxp<T> uv[CH];
#pragma unroll
		for (int i = 0; i < CH; i++) uv[i] = row.get(tau, (uint32_t)((m0 + i) * TPF)); // uv.re = (1, Re w_N^k), uv.im = (0, Im w_N^k)
		VKFFT_SCHED_FENCE();
#pragma unroll
		for (int i = 0; i < CH; i++) {
			const int m = m0 + i;
			const pk2<T> w0 = (m & ((1 << LOB) - 1)) ? pk_cmul_aos<T>(A[m >> LOB], B[m & ((1 << LOB) - 1)]) : A[m >> LOB];
			const pk2<T> w0x = pk_xx(w0), w0y = pk_yy(w0);
			v[m] = pcmul2(v[m], w0x * uv[i].re - w0y * uv[i].im, w0y * uv[i].re + w0x * uv[i].im);
			VKFFT_PIN2(v[m].re); VKFFT_PIN2(v[m].im); // (computed HERE: the optimiser otherwise sinks the products to the first use of v, with every table entry still in registers)
		}
		VKFFT_SCHED_FENCE();
	}
}

// ---- ONE complex value per register pair (x, y): the row kernels, whose threads have no second column to pair with --------------------------------------
// add / subtract are one v_pk_add_f32; a multiply is v_pk_mul_f32 + ONE v_pk_fma_f32 whose operand modifiers do the re/im rotation (op_sel swaps the halves of
// the data operand, neg_lo / neg_hi put the sign on one of them).  The compiler does not fold a swap-and-negate into those modifiers (it emits v_pk_add with 0
// and v_pk_mov_b32: 4 instructions), hence the inline assembly; plain VALU instructions, no hazard the assembler would have to pad.
#if defined(VKFFT_HOSTEMU)
template <typename T> inline pk2<T> pka_cmul(pk2<T> a, pk2<T> w) { return {a.x * w.x - a.y * w.y, a.y * w.x + a.x * w.y}; }
template <typename T> inline pk2<T> pka_cmul_const(pk2<T> d, T c, T s) { return {d.x * c + d.y * s, d.y * c - d.x * s}; } // d * (c - i s)
template <typename T> inline pk2<T> pka_sub_mi(pk2<T> a, pk2<T> b) { return {a.y - b.y, b.x - a.x}; }                    // -i (a - b)
#else
template <typename T> __device__ inline pk2<T> pka_cmul(pk2<T> a, pk2<T> w) {
	if constexpr (sizeof(T) == 4) {
		const pk2<T> t = a * pk_xx(w);
		pk2<T> r;
		asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
		return r;
	} else return pk2<T>{a.x * w.x - a.y * w.y, a.y * w.x + a.x * w.y};
}
template <typename T> __device__ inline pk2<T> pka_cmul_const(pk2<T> d, T c, T s) {
	if constexpr (sizeof(T) == 4) {
		const pk2<T> t = d * pk_splat<T>(c), ss = pk_splat<T>(s);
		pk2<T> r;
		asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(d), "s"(ss), "v"(t));
		return r;
	} else return pk2<T>{d.x * c + d.y * s, d.y * c - d.x * s};
}
template <typename T> __device__ inline pk2<T> pka_sub_mi(pk2<T> a, pk2<T> b) {
	if constexpr (sizeof(T) == 4) {
		pk2<T> r;
		asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
		return r;
	} else return pk2<T>{a.y - b.y, b.x - a.x};
}
#endif

// (compile-time loops: with inline assembly in the body the unroller's size estimate leaves rolled loops behind, and the register arrays in scratch)
template <typename F, int... I> __host__ __device__ inline void vk_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __host__ __device__ inline void vk_static_for(F&& f) { vk_static_for_impl(f, std::make_integer_sequence<int, N>{}); }
constexpr int vk_ilog2(int n) { int l = 0; while ((1 << l) < n) l++; return l; }

template <int R, typename T> __device__ inline void pka_dif_inplace(pk2<T>* x) {
	vk_static_for<vk_ilog2(R)>([&](auto li) {
		constexpr int h = (R / 2) >> decltype(li)::value;
		vk_static_for<R / 2>([&](auto bi) {
			constexpr int blk = (decltype(bi)::value / h) * 2 * h, j = decltype(bi)::value % h;
			const pk2<T> a = x[blk + j], b = x[blk + j + h];
			x[blk + j] = a + b;
			constexpr int kk = j * (16 / h);
			if constexpr (kk == 0) x[blk + j + h] = a - b;
			else if constexpr (kk == 8) x[blk + j + h] = pka_sub_mi<T>(a, b);
			else x[blk + j + h] = pka_cmul_const<T>(a - b, (T)pow2_cos32(kk), (T)pow2_sin32(kk));
		});
		VKFFT_SCHED_FENCE();
	});
}

// pow2_lean_butterflies / pow2_lean_exchange_part / pow2_lean_stages of kernel_pow2_lean.h for one column (rows: TC = 0) on (x, y) pairs
template <typename T, typename SCH, int SI, int TPF, typename TW, int TWG>
__device__ inline void pka_lean_butterflies(pk2<T>* v, const TW lut, const uint32_t tau) {
	constexpr int LOGE = SCH::LOGE, E = 1 << LOGE;
	constexpr int LOGR = SCH::bits[SI], R = 1 << LOGR, NB = E / R;
	constexpr int LOGS = SCH::logS(SI), S = 1 << LOGS;
#pragma unroll
	for (int b = 0; b < NB; b++) {
		if constexpr (SI > 0) {
			const uint32_t s = (tau + b * TPF) & (S - 1);
			constexpr int LO = SCH::lutOff(SI);
#pragma unroll
			for (int i0 = 1; i0 < R; i0 += TWG) {
				const int i1 = i0 + TWG < R ? i0 + TWG : R;
				pk2<T> w[TWG];
#pragma unroll
				for (int i = i0; i < i1; i++) w[i - i0] = pk_tw<T>(lut, s, (uint32_t)(LO + (i - 1) * S));
				VKFFT_SCHED_FENCE();
#pragma unroll
				for (int i = i0; i < i1; i++) { pk2<T>& q = v[b + i * NB]; q = pka_cmul<T>(q, w[i - i0]); VKFFT_PIN2(q); }
				VKFFT_SCHED_FENCE();
			}
		}
		pk2<T> x[R];
#pragma unroll
		for (int i = 0; i < R; i++) x[i] = v[b + i * NB];
		pka_dif_inplace<R, T>(x);
#pragma unroll
		for (int k = 0; k < R; k++) v[b + k * NB] = x[pow2_bitrev(k, LOGR)];
	}
}
// WAVE: a row's threads are lanes of ONE wavefront (TPF <= 64 divides 64): the exchange needs the wave's own LDS order only, no workgroup barrier — the waves of
// a workgroup then run their rows independently of each other
template <typename T, typename SCH, int SI, int TPF, int PART, bool WAVE>
__device__ inline void pka_lean_exchange_part(pk2<T>* v, T* plane, const uint32_t tau) {
	constexpr int LOGE = SCH::LOGE, E = 1 << LOGE, P = SCH::bits[0];
	constexpr int LOGR = SCH::bits[SI], R = 1 << LOGR, NB = E / R;
	constexpr int LOGS = SCH::logS(SI), S = 1 << LOGS;
#pragma unroll
	for (int b = 0; b < NB; b++) {
		const uint32_t t = tau + b * TPF;
		const uint32_t s = t & (S - 1);
		const uint32_t ob = ((t - s) << LOGR) + s;
		T* const wp = plane + pow2_lean_slot<0, P>(ob);
#pragma unroll
		for (int k = 0; k < R; k++) wp[pow2_lean_step<0, P>(k * S)] = PART ? v[b + k * NB].y : v[b + k * NB].x;
	}
	if constexpr (WAVE) VKFFT_WAVE_SYNC(); else VKFFT_SYNC();
	const T* const rp = plane + pow2_lean_slot<0, P>(tau);
#pragma unroll
	for (int m = 0; m < E; m++) {
		const T r = rp[pow2_lean_step<0, P>(m * TPF)];
		if (PART) v[m].y = r; else v[m].x = r;
	}
}
template <typename T, typename SCH, int SI, int TPF, typename TW, int TWG, bool WAVE = false>
__device__ inline void pka_lean_stages(pk2<T>* v, T* plane, const TW lut, const uint32_t tau) {
	pka_lean_butterflies<T, SCH, SI, TPF, TW, TWG>(v, lut, tau);
	if constexpr (SI + 1 < SCH::NS) {
		pka_lean_exchange_part<T, SCH, SI, TPF, 0, WAVE>(v, plane, tau);
		if constexpr (WAVE) VKFFT_WAVE_SYNC(); else VKFFT_SYNC();
		pka_lean_exchange_part<T, SCH, SI, TPF, 1, WAVE>(v, plane, tau);
		if constexpr (SI + 2 < SCH::NS) { if constexpr (WAVE) VKFFT_WAVE_SYNC(); else VKFFT_SYNC(); }
		pka_lean_stages<T, SCH, SI + 1, TPF, TW, TWG, WAVE>(v, plane, lut, tau);
	}
}

// ---- unit-stride rows of N = 2^9 ... 2^15 points: pow2_row_lean_kernel on (x, y) register pairs, FPW rows per workgroup (one for 2^13 ... 2^15) ----
template <typename T, typename SCH, int WPE, int TWG, int FPW = 1>
__global__ void __launch_bounds__(((1 << SCH::LOGN) >> SCH::LOGE) * FPW, WPE) pow2_row_lean_pk_kernel(const PassParams p) {
	constexpr int LOGN = SCH::LOGN, N = 1 << LOGN, LOGE = SCH::LOGE, E = 1 << LOGE, TPF = N / E;
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	constexpr int PL = (int)pow2_lean_plane_elems<SCH, 0>();
	__shared__ T planes[FPW * PL];
	const uint32_t tau = FPW > 1 ? threadIdx.x % (uint32_t)TPF : threadIdx.x, fl = FPW > 1 ? threadIdx.x / (uint32_t)TPF : 0u;
	T* const plane = planes + fl * PL;
	uint32_t wg = p.reverseTiles ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
	const uint32_t f0 = (wg % p.tilesPerG0) * (uint32_t)FPW + fl; // FPW consecutive rows per tile
	wg /= p.tilesPerG0;
	const uint32_t g1 = wg % p.dim[1].count, g2 = wg / p.dim[1].count;
	const bool valid = FPW == 1 || f0 < p.dim[0].count; // (lanes of a row beyond the last one get the out-of-range offset: they load zeros and store nothing, and keep the barriers)
	const GBuf gin = make_gbuf((const cx<T>*)p.in + ((int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride + (int64_t)(FPW == 1 ? f0 : f0 - fl) * p.dim[0].inStride));
	const GBuf gout = make_gbuf((cx<T>*)p.out + ((int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride + (int64_t)(FPW == 1 ? f0 : f0 - fl) * p.dim[0].outStride));
	const GBuf glut = make_gbuf(p.lut);
	const uint32_t laneIn = !valid ? kGbInvalid : (tau + (FPW == 1 ? 0u : fl * (uint32_t)p.dim[0].inStride)) * ES;
	const uint32_t laneOut = !valid ? kGbInvalid : (tau + (FPW == 1 ? 0u : fl * (uint32_t)p.dim[0].outStride)) * ES;
	pk2<T> v[E];
	auto ld = [&](uint32_t voff, uint32_t soff) { const cx<T> q = gb_load<T>(gin, voff, soff); return pk2<T>{q.x, q.y}; };
	if (p.padInN) { // zero padding: points of the padded range get an out-of-range offset (they read as zero and are not fetched)
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = ld((tau + (uint32_t)(m * TPF) - p.padInL < p.padInN) ? kGbInvalid : laneIn, (uint32_t)(m * TPF) * ES);
	} else {
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = ld(laneIn, (uint32_t)(m * TPF) * ES);
	}
	if (p.swapIn) { // inverse = conj . forward . conj
#pragma unroll
		for (int m = 0; m < E; m++) v[m].y = -v[m].y;
	}
	// (WAVE = false also where a row sits inside one wavefront: the wave-level ordering is a workgroup-scope fence, which waits for the wave's loads and stores
	// to HBM as well — measured 4 % slower than the barrier at 2^9 / 2^10, profiles/r05_ab_small_sizes_*)
	pka_lean_stages<T, SCH, 0, TPF, TwGlobal<T>, TWG, false>(v, plane, TwGlobal<T>{glut}, tau);
	const T sc = (T)p.scale;
	if (sc != (T)1 || p.swapOut) {
		const pk2<T> f = {sc, p.swapOut ? -sc : sc};
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = v[m] * f;
	}
	if (p.padOutN) { // (the padded range of the output is not written)
#pragma unroll
		for (int m = 0; m < E; m++) gb_store<T>(gout, (tau + (uint32_t)(m * TPF) - p.padOutL < p.padOutN) ? kGbInvalid : laneOut, (uint32_t)(m * TPF) * ES, cx<T>{v[m].x, v[m].y});
	} else {
#pragma unroll
		for (int m = 0; m < E; m++) gb_store<T>(gout, laneOut, (uint32_t)(m * TPF) * ES, cx<T>{v[m].x, v[m].y});
	}
}
template <typename T, typename SCH, int WPE, int TWG, int FPW = 1> void pow2_row_lean_pk_launch(const PassParams& prm, dim3 grid, hipStream_t s) {
	hipLaunchKernelGGL((pow2_row_lean_pk_kernel<T, SCH, WPE, TWG, FPW>), grid, dim3(((1 << SCH::LOGN) >> SCH::LOGE) * FPW), 0, s, prm);
}

// ---- unit-stride rows of 2^15 points as 2^14 PAIRS of neighbouring samples, software-pipelined over the rows (round 5) -------------------------------------------
// The one-pass 2^15 row kernel above holds a row in 1024 threads x 64 data registers: no room for a second row, so load, compute and store of a row run in series
// (4.0-4.3 TB/s; the stores alone keep the waves stalled at their issue for as long as the loads take).  Here a 16-byte load brings the samples (x[2i], x[2i+1]) —
// the i-th points of the two interleaved half-length sequences of a decimation-in-time split — which are exactly the "two adjacent columns" of the packed-pair
// stages: ONE 2^14-point transform on register pairs (SCH) gives E[k] and O[k] in the two lanes, and X[k] = E[k] + w^k O[k], X[k + 2^14] = E[k] - w^k O[k] is one layer in
// registers.  512 threads x 256 registers hold a row (128) plus half of the next one: the workgroup is persistent, the first half of the next row travels during the
// stages and the stores, the second half is requested behind the stores.
template <typename T, typename SCH, int TWG>
__global__ void __launch_bounds__((1 << SCH::LOGN) >> SCH::LOGE, 2) pow2_row_pairs_kernel(const PassParams p) {
	constexpr int LOGN2 = SCH::LOGN, N2 = 1 << LOGN2, LOGE = SCH::LOGE, E = 1 << LOGE, TPF = N2 / E, H = E / 2;
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	typedef Pow2Sched<SCH::bits[0], SCH::bits[1], SCH::bits[2], 1> SF; // (table layout of the planner: SCH's runs, then w_N^k, k < N / 2)
	constexpr int LUTC = SF::lutOff(3);
	static_assert(SCH::bits[3] == 0 && sizeof(T) == 4, "three stages on pairs of fp32 samples");
	__shared__ __attribute__((aligned(16))) T plane[pow2_lean_plane_elems<SCH, 2>()];
	const uint32_t tau = threadIdx.x;
	const uint32_t total = p.tilesPerG0 * p.dim[1].count * p.dim[2].count;
	const GBuf glut = make_gbuf(p.lut);
	pk4<T> ra[H], rb[H]; // the two halves of a row while they travel
	auto rowBase = [&](uint32_t w, bool out) -> int64_t {
		uint32_t wg = p.reverseTiles ? total - 1u - w : w;
		const uint32_t f0 = wg % p.tilesPerG0; wg /= p.tilesPerG0;
		const uint32_t g1 = wg % p.dim[1].count, g2 = wg / p.dim[1].count;
		return out ? (int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride + (int64_t)f0 * p.dim[0].outStride
		           : (int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride + (int64_t)f0 * p.dim[0].inStride;
	};
	// half h of row w: pairs i = tau + m TPF, m in [h H, h H + H).  (No zero-padding masks: Pow2Variant::noPadMasks sends a padded pass to the one-row kernel above)
	auto request = [&](uint32_t w, int h, pk4<T>* raw) {
		VKFFT_OPAQUE_ZERO(oq);
		const bool live = w < total;
		const GBuf gin = make_gbuf((const cx<T>*)p.in + (live ? rowBase(w, false) : 0));
		const uint32_t lane = live ? tau * 2u * ES + oq : kGbInvalid;
#pragma unroll
		for (int m = 0; m < H; m++) raw[m] = gb_load_aos2<T, 0>(gin, lane, (uint32_t)((h * H + m) * TPF) * 2u * ES);
	};
	uint32_t w = blockIdx.x;
	request(w, 0, ra);
	request(w, 1, rb);
	for (; w < total; w += gridDim.x) {
		cxp<T> v[E];
#pragma unroll
		for (int m = 0; m < H; m++) v[m] = pk_from_aos<T>(ra[m]);
#pragma unroll
		for (int m = 0; m < H; m++) v[H + m] = pk_from_aos<T>(rb[m]);
		if (p.swapIn) { // inverse = conj . forward . conj
#pragma unroll
			for (int m = 0; m < E; m++) v[m].im = -v[m].im;
		}
		request(w + gridDim.x, 0, ra); // the first half of the next row travels during the stages and the stores
		pk_lean_stages<T, SCH, 0, TPF, 2, TwGlobal<T>, TWG>(v, plane, TwGlobal<T>{glut}, tau);
		VKFFT_SYNC(); // the last exchange's reads are complete in every wave: the next row's first exchange may write the plane
		const T sc = (T)p.scale, sci = p.swapOut ? -sc : sc;
		const GBuf gout = make_gbuf((cx<T>*)p.out + rowBase(w, true));
		VKFFT_OPAQUE_ZERO(oz);
		const uint32_t lane = tau * ES + oz;
		constexpr int CH = 8; // twiddles of the last layer in flight at a time
#pragma unroll
		for (int m0 = 0; m0 < E; m0 += CH) {
			pk2<T> tw[CH];
#pragma unroll
			for (int i = 0; i < CH; i++) tw[i] = pk_tw<T>(TwGlobal<T>{glut}, tau, (uint32_t)(LUTC + (m0 + i) * TPF));
			VKFFT_SCHED_FENCE();
#pragma unroll
			for (int i = 0; i < CH; i++) {
				const int m = m0 + i;
				const T tr = v[m].re.y * tw[i].x - v[m].im.y * tw[i].y, ti = v[m].re.y * tw[i].y + v[m].im.y * tw[i].x; // w^k O[k]
				const cx<T> lo = {(v[m].re.x + tr) * sc, (v[m].im.x + ti) * sci}, hi = {(v[m].re.x - tr) * sc, (v[m].im.x - ti) * sci};
				gb_store<T>(gout, lane, (uint32_t)(m * TPF) * ES, lo);
				gb_store<T>(gout, lane, (uint32_t)(m * TPF + N2) * ES, hi);
			}
			VKFFT_SCHED_FENCE();
		}
		request(w + gridDim.x, 1, rb); // (behind the stores: it is there when they are)
	}
}
template <typename T, typename SCH, int TWG> void pow2_row_pairs_launch(const PassParams& prm, dim3 grid, hipStream_t s) {
	const unsigned resident = pow2_num_cus(); // persistent: one 135 KiB workgroup per CU, the rows dealt round-robin
	hipLaunchKernelGGL((pow2_row_pairs_kernel<T, SCH, TWG>), dim3(grid.x < resident ? grid.x : resident), dim3((1 << SCH::LOGN) >> SCH::LOGE), 0, s, prm);
}

} // namespace vkfft_mi355x
