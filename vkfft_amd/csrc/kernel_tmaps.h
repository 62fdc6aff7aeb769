// Table-driven pre / post maps of the real transforms (R2C / C2R in their full-length forms, DCT / DST I-IV) for the instance kernels that run a complex
// transform between two maps (kernel_mixed.h / kernel_mixconv.h / kernel_mixrad.h, OPS = 1).  The reference generates the map of each transform into its
// kernel (vkFFT_R2C.h:178,450; vkFFT_R2R.h:193-336, 784-1031, 1339-2318: index arithmetic, twiddle and store per family); the maps of kernel_generic.h do the
// same per element behind a run-time switch, in rolled loops with a division and a 64-bit address per element.  Here every family is ONE pair of tables
// built by the planner (planner.cpp: build_tmaps) and ONE unrolled, branch-free piece of code per side:
//   pre-map   FFT input pos of a row   V[pos] = c1[pos] * x[o1[pos]] + c2[pos] * x[o2[pos]]          x: the row's REAL scalars (a complex row = re, im, re, ...),
//             two rows a, b per transform (the reference's mergeSequencesR2C, vkFFT_SharedMemory.h:40):  z[pos] = V_a[pos] + i V_b[pos]
//   post-map  (a) split form, index k <= L/2:  X_a[k] = Z[k] + conj Z[L - k],  X_b[k] = -i (Z[k] - conj Z[L - k])   (twice the rows' spectra; the tables carry 1/2)
//                 y[o1[k]] = Re(c1[k] X[k]),  y[o2[k]] = Re(c2[k] X[k])    for both rows; kTmCplx: one complex store (Re(c1 X), Re(c2 X)) at o1
//             (b) direct form, FFT output m:  y[o1[m]] = Re(c1[m] Z[m]),  y[o2[m]] = Re(c2[m] Z[m]);  kTmRowB: the second value is row b's (real results of a pair)
// o1 / o2 are BYTE offsets inside a row; kGbInvalid = "no such term" (a load returns 0, a store is dropped by the range check of the buffer access).
// Scale, signs, quarter-wave twiddles and the 1/2 of the split are folded into c1 / c2.
#pragma once
#include "memops.h"

namespace vkfft_mi355x {

#if defined(VKFFT_HOSTEMU)
inline uint32_t tm_load_u1(GBuf b, uint32_t voff, uint32_t soff) { return voff >= kGbRange ? 0u : *(const uint32_t*)(b.base + (uint64_t)voff + soff); }
inline void tm_load_u2(GBuf b, uint32_t voff, uint32_t soff, uint32_t& a, uint32_t& c) {
	if (voff >= kGbRange) { a = c = 0; return; }
	const uint32_t* q = (const uint32_t*)(b.base + (uint64_t)voff + soff);
	a = q[0]; c = q[1];
}
#else
__device__ inline uint32_t tm_load_u1(GBuf b, uint32_t voff, uint32_t soff) { return __builtin_amdgcn_raw_buffer_load_b32(b.r, voff, soff, 0); }
__device__ inline void tm_load_u2(GBuf b, uint32_t voff, uint32_t soff, uint32_t& a, uint32_t& c) {
	const vk_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(b.r, voff, soff, 0);
	a = t.x; c = t.y;
}
#endif

// the tables of one side: offs = uint32 pairs, coef = pairs of cx<T>, one entry per position; data = the tile's rows
template <typename T> struct TmSide {
	GBuf data, offs, coef;
	uint32_t rowA, rowB; // byte offsets of the two rows of this transform (kGbInvalid: no such row)
};
template <typename T> __device__ inline TmSide<T> tm_side(const void* table, uint32_t entries, GBuf data, uint32_t rowA, uint32_t rowB) {
	TmSide<T> s;
	s.data = data; s.offs = make_gbuf(table); s.coef = make_gbuf((const char*)table + (((size_t)entries * 8u + 15u) & ~(size_t)15u));
	s.rowA = rowA; s.rowB = rowB;
	return s;
}
template <typename T> __device__ inline void tm_entry(const TmSide<T>& s, uint32_t t, uint32_t c, uint32_t& o1, uint32_t& o2, cx<T>& c1, cx<T>& c2) {
	tm_load_u2(s.offs, t * 8u, c * 8u, o1, o2);
	const Real4<T> q = gb_load_real4<T>(s.coef, t * (uint32_t)(4 * sizeof(T)), c * (uint32_t)(4 * sizeof(T)));
	c1 = cx<T>{q.x, q.y}; c2 = cx<T>{q.z, q.w};
}

// pre-map: FFT input t + c of the transform that carries rows a and b.  One term (TWO = false): a signed gather — every family of that form has a REAL c1
// (R2C, DCT / DST-I, -II, odd -IV), so only o1 and Re c1 are read: four registers per point in flight instead of ten
template <typename T, bool TWO> __device__ inline cx<T> tm_pre(const TmSide<T>& s, uint32_t t, uint32_t c) {
	if constexpr (!TWO) {
		const uint32_t o1 = tm_load_u1(s.offs, t * 8u, c * 8u);
		const T sg = gb_load_real<T>(s.coef, t * (uint32_t)(4 * sizeof(T)), c * (uint32_t)(4 * sizeof(T)));
		const T a1 = gb_load_real<T>(s.data, s.rowA + o1, 0), b1 = gb_load_real<T>(s.data, s.rowB + o1, 0);
		return cx<T>{sg * a1, sg * b1};
	} else {
		// two terms: c2 = +-i c1 in every family (C2R: the imaginary part of a bin; DCT / DST-III: -i x[N - k]; even DCT / DST-IV: +i x[N - 1 - 2n]); the sign
		// rides in bit 0 of o2 and only c1 is read — eight registers per point in flight instead of ten.  z = V_a + i V_b = c1 ((a1 - s b2) + i (s a2 + b1))
		uint32_t o1, o2;
		tm_load_u2(s.offs, t * 8u, c * 8u, o1, o2);
		const cx<T> c1 = gb_load<T>(s.coef, t * (uint32_t)(4 * sizeof(T)), c * (uint32_t)(4 * sizeof(T)));
		const bool minus = (o2 & 1u) != 0u;
		o2 &= ~1u;
		const T a1 = gb_load_real<T>(s.data, s.rowA + o1, 0), b1 = gb_load_real<T>(s.data, s.rowB + o1, 0);
		T a2 = gb_load_real<T>(s.data, s.rowA + o2, 0), b2 = gb_load_real<T>(s.data, s.rowB + o2, 0);
		if (minus) { a2 = -a2; b2 = -b2; }
		const cx<T> w = {a1 - b2, a2 + b1};
		return cx<T>{c1.x * w.x - c1.y * w.y, c1.x * w.y + c1.y * w.x};
	}
}

// ---- post-maps.  They run AFTER the transform's outputs are in LDS (rd(a) = FFT output a, natural order) and in two phases: every table entry and every
// output of the thread first, the stores after them.  A wave has ONE in-order counter for its loads and stores: a table load issued behind a store is
// waited for together with that store, so the interleaved form (entry, outputs, stores, next entry ...) exposes the latency of a store to memory once per
// point — measured on 169-point rows: 19 us per tile, every family alike.  Lanes beyond the last point read nothing and get an out-of-range store offset
// (no branch around the loads: at a join the compiler waits for everything that is in flight).
// split form: index k <= L/2 (kernel_tmaps.h header)
template <typename T, int L, int TPF, typename RD> __device__ inline void tm_post_split(const TmSide<T>& s, uint32_t flags, uint32_t tau, const RD& rd) {
	constexpr int H = L / 2 + 1, PB = (H + TPF - 1) / TPF;
	uint32_t o1[PB], o2[PB]; T ya1[PB], ya2[PB], yb1[PB], yb2[PB];
#pragma unroll
	for (int b = 0; b < PB; b++) {
		const uint32_t k = tau + (uint32_t)(b * TPF);
		const bool live = (b + 1) * TPF <= H || k < (uint32_t)H;
		const uint32_t kk = live ? k : 0u;
		const cx<T> zk = rd(kk), zm = rd(kk ? (uint32_t)L - kk : 0u);
		cx<T> c1, c2;
		uint32_t a1, a2;
		tm_load_u2(s.offs, live ? tau * 8u : kGbInvalid, (uint32_t)(b * TPF) * 8u, a1, a2);
		const Real4<T> q = gb_load_real4<T>(s.coef, live ? tau * (uint32_t)(4 * sizeof(T)) : kGbInvalid, (uint32_t)(b * TPF) * (uint32_t)(4 * sizeof(T)));
		c1 = cx<T>{q.x, q.y}; c2 = cx<T>{q.z, q.w};
		o1[b] = live ? a1 : kGbInvalid; o2[b] = live ? a2 : kGbInvalid;
		const cx<T> xa = {zk.x + zm.x, zk.y - zm.y}, xb = {zk.y + zm.y, zm.x - zk.x};
		ya1[b] = c1.x * xa.x - c1.y * xa.y; ya2[b] = c2.x * xa.x - c2.y * xa.y;
		yb1[b] = c1.x * xb.x - c1.y * xb.y; yb2[b] = c2.x * xb.x - c2.y * xb.y;
	}
	if (flags & kTmCplx) {
#pragma unroll
		for (int b = 0; b < PB; b++) { gb_store<T>(s.data, s.rowA + o1[b], 0, cx<T>{ya1[b], ya2[b]}); gb_store<T>(s.data, s.rowB + o1[b], 0, cx<T>{yb1[b], yb2[b]}); }
	} else {
#pragma unroll
		for (int b = 0; b < PB; b++) {
			gb_store_real<T>(s.data, s.rowA + o1[b], 0, ya1[b]); gb_store_real<T>(s.data, s.rowA + o2[b], 0, ya2[b]);
			gb_store_real<T>(s.data, s.rowB + o1[b], 0, yb1[b]); gb_store_real<T>(s.data, s.rowB + o2[b], 0, yb2[b]);
		}
	}
}
// direct forms: FFT output m -> y[o1] = Re(c1 Z), y[o2] = Re(c2 Z); kTmRowB: real results of a pair, y_a[o1] = s Re Z, y_b[o1] = s Im Z with s = Re c1
template <typename T, int L, int TPF, typename RD> __device__ inline void tm_post_rows(const TmSide<T>& s, uint32_t flags, uint32_t tau, const RD& rd) {
	constexpr int P = (L + TPF - 1) / TPF;
	if (flags & kTmRowB) {
		uint32_t o1[P]; T ya[P], yb[P];
#pragma unroll
		for (int b = 0; b < P; b++) {
			const uint32_t m = tau + (uint32_t)(b * TPF);
			const bool live = (b + 1) * TPF <= L || m < (uint32_t)L;
			const cx<T> z = rd(live ? m : 0u);
			const uint32_t a1 = tm_load_u1(s.offs, live ? tau * 8u : kGbInvalid, (uint32_t)(b * TPF) * 8u);
			const T sg = gb_load_real<T>(s.coef, live ? tau * (uint32_t)(4 * sizeof(T)) : kGbInvalid, (uint32_t)(b * TPF) * (uint32_t)(4 * sizeof(T)));
			o1[b] = live ? a1 : kGbInvalid; ya[b] = sg * z.x; yb[b] = sg * z.y;
		}
#pragma unroll
		for (int b = 0; b < P; b++) { gb_store_real<T>(s.data, s.rowA + o1[b], 0, ya[b]); gb_store_real<T>(s.data, s.rowB + o1[b], 0, yb[b]); }
	} else {
		constexpr int CH = 8; // (eight points per round: the entries of sixteen would not fit beside the rest)
#pragma unroll
		for (int b0 = 0; b0 < P; b0 += CH) {
			uint32_t o1[CH], o2[CH]; T y1[CH], y2[CH];
#pragma unroll
			for (int j = 0; j < CH; j++) {
				const int b = b0 + j;
				if (b < P) {
					const uint32_t m = tau + (uint32_t)(b * TPF);
					const bool live = (b + 1) * TPF <= L || m < (uint32_t)L;
					const cx<T> z = rd(live ? m : 0u);
					uint32_t a1, a2;
					tm_load_u2(s.offs, live ? tau * 8u : kGbInvalid, (uint32_t)(b * TPF) * 8u, a1, a2);
					const Real4<T> q = gb_load_real4<T>(s.coef, live ? tau * (uint32_t)(4 * sizeof(T)) : kGbInvalid, (uint32_t)(b * TPF) * (uint32_t)(4 * sizeof(T)));
					o1[j] = live ? a1 : kGbInvalid; o2[j] = live ? a2 : kGbInvalid;
					y1[j] = q.x * z.x - q.y * z.y; y2[j] = q.z * z.x - q.w * z.y;
				}
			}
#pragma unroll
			for (int j = 0; j < CH; j++) if (b0 + j < P) { gb_store_real<T>(s.data, s.rowA + o1[j], 0, y1[j]); gb_store_real<T>(s.data, s.rowA + o2[j], 0, y2[j]); }
		}
	}
}

} // namespace vkfft_mi355x
