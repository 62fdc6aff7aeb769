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

// Where the maps read the rows from / put their values: global memory (buffer accesses: an out-of-range offset reads 0 / drops the store) or a staging tile
// in LDS that holds the tile's rows as they lie in memory (kernel_mixed.h, fewer than eight threads per row: the tile moves as one contiguous run)
template <typename T> struct TmGlobal {
	GBuf g;
	__device__ inline T real(uint32_t off) const { return gb_load_real<T>(g, off, 0); }
	__device__ inline void put(uint32_t off, T v) const { gb_store_real<T>(g, off, 0, v); }
	__device__ inline void put2(uint32_t off, cx<T> v) const { gb_store<T>(g, off, 0, v); }
};
// slot of scalar i of the staging tile: one scalar of padding per 32.  The tile holds the rows as they lie in memory, and a row pitch that is a multiple of 32
// scalars (R2C of 31 reals in place: 32) put every lane of a one-thread-per-row instance on ONE bank (round 5: 0.34x the reference, 64-way conflicts)
__host__ __device__ constexpr uint32_t tm_slot(uint32_t i) { return i + (i >> 5); }
template <typename T> struct TmLds {
	T* base;
	__device__ inline T real(uint32_t off) const { const T v = base[tm_slot((off < kGbRange ? off : 0u) / (uint32_t)sizeof(T))]; return off < kGbRange ? v : (T)0; }
	__device__ inline void put(uint32_t off, T v) const { if (off < kGbRange) base[tm_slot(off / (uint32_t)sizeof(T))] = v; }
	__device__ inline void put2(uint32_t off, cx<T> v) const { if (off < kGbRange) { base[tm_slot(off / (uint32_t)sizeof(T))] = v.x; base[tm_slot(off / (uint32_t)sizeof(T) + 1u)] = v.y; } }
};
// the tables of one side: offs = uint32 pairs, coef = pairs of cx<T>, one entry per position; data = the tile's rows
template <typename T> struct TmSide {
	GBuf data, tab;      // the rows; the table: uint32 pairs, then (16-byte aligned) the coefficient pairs — ONE resource, the coefficients at a scalar offset (a second
	                     // resource per side cost the small instances their last scalar registers: 27 spilled, a private segment for them)
	uint32_t coefOff, rowA, rowB; // rowA / rowB: byte offsets of the two rows of this transform (kGbInvalid: no such row)
};
template <typename T> __device__ inline TmSide<T> tm_side(const void* table, uint32_t entries, GBuf data, uint32_t rowA, uint32_t rowB) {
	TmSide<T> s;
	s.data = data; s.tab = make_gbuf(table); s.coefOff = (entries * 8u + 15u) & ~15u;
	s.rowA = rowA; s.rowB = rowB;
	return s;
}
template <typename T> __device__ inline void tm_entry(const TmSide<T>& s, uint32_t t, uint32_t c, uint32_t& o1, uint32_t& o2, cx<T>& c1, cx<T>& c2) {
	tm_load_u2(s.tab, t * 8u + c * 8u, 0u, o1, o2);
	const Real4<T> q = gb_load_real4<T>(s.tab, t * (uint32_t)(4 * sizeof(T)) + s.coefOff + c * (uint32_t)(4 * sizeof(T)), 0u);
	c1 = cx<T>{q.x, q.y}; c2 = cx<T>{q.z, q.w};
}

// pre-map: FFT input t + c of the transform that carries rows a and b.  One term (TWO = false): a signed gather — every family of that form has a REAL c1
// (R2C, DCT / DST-I, -II, odd -IV), so only o1 and Re c1 are read: four registers per point in flight instead of ten
template <typename T, bool TWO, typename SRC> __device__ inline cx<T> tm_pre(const TmSide<T>& s, const SRC& src, uint32_t t, uint32_t c) {
	if constexpr (!TWO) {
		const uint32_t o1 = tm_load_u1(s.tab, t * 8u + c * 8u, 0u);
		const T sg = gb_load_real<T>(s.tab, t * (uint32_t)(4 * sizeof(T)) + s.coefOff + c * (uint32_t)(4 * sizeof(T)), 0u);
		const T a1 = src.real(s.rowA + o1), b1 = src.real(s.rowB + o1);
		return cx<T>{sg * a1, sg * b1};
	} else {
		// two terms: c2 = +-i c1 in every family (C2R: the imaginary part of a bin; DCT / DST-III: -i x[N - k]; even DCT / DST-IV: +i x[N - 1 - 2n]); the sign
		// rides in bit 0 of o2 and only c1 is read — eight registers per point in flight instead of ten.  z = V_a + i V_b = c1 ((a1 - s b2) + i (s a2 + b1))
		uint32_t o1, o2;
		tm_load_u2(s.tab, t * 8u + c * 8u, 0u, o1, o2);
		const cx<T> c1 = gb_load<T>(s.tab, t * (uint32_t)(4 * sizeof(T)) + s.coefOff + c * (uint32_t)(4 * sizeof(T)), 0u);
		const bool minus = (o2 & 1u) != 0u;
		o2 &= ~1u;
		const T a1 = src.real(s.rowA + o1), b1 = src.real(s.rowB + o1);
		T a2 = src.real(s.rowA + o2), b2 = src.real(s.rowB + o2);
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
template <typename T, int L, int TPF, typename RD, typename SINK> __device__ inline void tm_post_split(const TmSide<T>& s, const SINK& sink, uint32_t flags, uint32_t tau, const RD& rd) {
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
		tm_load_u2(s.tab, live ? tau * 8u + (uint32_t)(b * TPF) * 8u : kGbInvalid, 0u, a1, a2);
		const Real4<T> q = gb_load_real4<T>(s.tab, live ? tau * (uint32_t)(4 * sizeof(T)) + s.coefOff + (uint32_t)(b * TPF) * (uint32_t)(4 * sizeof(T)) : kGbInvalid, 0u);
		c1 = cx<T>{q.x, q.y}; c2 = cx<T>{q.z, q.w};
		o1[b] = live ? a1 : kGbInvalid; o2[b] = live ? a2 : kGbInvalid;
		const cx<T> xa = {zk.x + zm.x, zk.y - zm.y}, xb = {zk.y + zm.y, zm.x - zk.x};
		ya1[b] = c1.x * xa.x - c1.y * xa.y; ya2[b] = c2.x * xa.x - c2.y * xa.y;
		yb1[b] = c1.x * xb.x - c1.y * xb.y; yb2[b] = c2.x * xb.x - c2.y * xb.y;
	}
	if (flags & kTmCplx) {
#pragma unroll
		for (int b = 0; b < PB; b++) { sink.put2(s.rowA + o1[b], cx<T>{ya1[b], ya2[b]}); sink.put2(s.rowB + o1[b], cx<T>{yb1[b], yb2[b]}); }
	} else {
#pragma unroll
		for (int b = 0; b < PB; b++) {
			sink.put(s.rowA + o1[b], ya1[b]); sink.put(s.rowA + o2[b], ya2[b]);
			sink.put(s.rowB + o1[b], yb1[b]); sink.put(s.rowB + o2[b], yb2[b]);
		}
	}
}
// direct forms: FFT output m -> y[o1] = Re(c1 Z), y[o2] = Re(c2 Z); kTmRowB: real results of a pair, y_a[o1] = s Re Z, y_b[o1] = s Im Z with s = Re c1
template <typename T, int L, int TPF, typename RD, typename SINK> __device__ inline void tm_post_rows(const TmSide<T>& s, const SINK& sink, uint32_t flags, uint32_t tau, const RD& rd) {
	constexpr int P = (L + TPF - 1) / TPF;
	if (flags & kTmRowB) {
		uint32_t o1[P]; T ya[P], yb[P];
#pragma unroll
		for (int b = 0; b < P; b++) {
			const uint32_t m = tau + (uint32_t)(b * TPF);
			const bool live = (b + 1) * TPF <= L || m < (uint32_t)L;
			const cx<T> z = rd(live ? m : 0u);
			const uint32_t a1 = tm_load_u1(s.tab, live ? tau * 8u + (uint32_t)(b * TPF) * 8u : kGbInvalid, 0u);
			const T sg = gb_load_real<T>(s.tab, live ? tau * (uint32_t)(4 * sizeof(T)) + s.coefOff + (uint32_t)(b * TPF) * (uint32_t)(4 * sizeof(T)) : kGbInvalid, 0u);
			o1[b] = live ? a1 : kGbInvalid; ya[b] = sg * z.x; yb[b] = sg * z.y;
		}
#pragma unroll
		for (int b = 0; b < P; b++) { sink.put(s.rowA + o1[b], ya[b]); sink.put(s.rowB + o1[b], yb[b]); }
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
					tm_load_u2(s.tab, live ? tau * 8u + (uint32_t)(b * TPF) * 8u : kGbInvalid, 0u, a1, a2);
					const Real4<T> q = gb_load_real4<T>(s.tab, live ? tau * (uint32_t)(4 * sizeof(T)) + s.coefOff + (uint32_t)(b * TPF) * (uint32_t)(4 * sizeof(T)) : kGbInvalid, 0u);
					o1[j] = live ? a1 : kGbInvalid; o2[j] = live ? a2 : kGbInvalid;
					y1[j] = q.x * z.x - q.y * z.y; y2[j] = q.z * z.x - q.w * z.y;
				}
			}
#pragma unroll
			for (int j = 0; j < CH; j++) if (b0 + j < P) { sink.put(s.rowA + o1[j], y1[j]); sink.put(s.rowA + o2[j], y2[j]); }
		}
	}
}

// ---- the same maps over a tile of rows of RUN-TIME length held in LDS (kernel_mixrad.h: rows of M * P points, M a run-time cofactor): all NT threads of the
// workgroup sweep the tile's points, four per trip — the four points' table entries and values are requested together, the LDS writes / the stores follow
// (a rolled loop with one point per trip pays a memory round trip per point; and see the note on the post-maps above about loads behind stores).
// rows: transforms in the tile; realRows: real rows in the tile (mult per transform); pitchBytes: row pitch in memory
template <typename T, bool TWO, typename DST>
__device__ inline void tm_rows_in(const void* table, GBuf data, uint32_t N, const FastDiv divN, uint32_t rows, uint32_t realRows, uint32_t mult, uint32_t pitchBytes, bool swI,
                                  uint32_t tid, uint32_t NT, const DST& dst) {
	TmSide<T> s = tm_side<T>(table, N, data, 0u, 0u);
	const uint32_t total = rows * N;
	constexpr int U = 4;
	for (uint32_t e0 = tid; e0 < total; e0 += (uint32_t)U * NT) {
		cx<T> z[U]; uint32_t rr[U], pp[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			const uint32_t e = e0 + (uint32_t)u * NT;
			const bool live = e < total;
			divN.divmod(live ? e : 0u, rr[u], pp[u]);
			const uint32_t rA = rr[u] * mult;
			s.rowA = (live && rA < realRows) ? rA * pitchBytes : kGbInvalid;
			s.rowB = (live && mult == 2u && rA + 1u < realRows) ? (rA + 1u) * pitchBytes : kGbInvalid;
			z[u] = tm_pre<T, TWO>(s, TmGlobal<T>{data}, pp[u], 0u);
		}
#pragma unroll
		for (int u = 0; u < U; u++) if (e0 + (uint32_t)u * NT < total) dst(rr[u], pp[u], swI ? cswap(z[u]) : z[u]);
	}
}
// rd(r, a) = FFT output a of transform r (un-swapped by the caller)
template <typename T, typename RD>
__device__ inline void tm_rows_out(const void* table, GBuf data, uint32_t L, uint32_t rows, uint32_t realRows, uint32_t mult, uint32_t pitchBytes, uint32_t flags,
                                   uint32_t tid, uint32_t NT, const RD& rd) {
	const bool split = (flags & kTmSplit) != 0u, cplx = (flags & kTmCplx) != 0u, rowB = (flags & kTmRowB) != 0u;
	const uint32_t H = split ? L / 2u + 1u : L;
	FastDiv divH; divH.d = H; divH.rcp = 1.0f / (float)H;
	const TmSide<T> s = tm_side<T>(table, H, data, 0u, 0u);
	const uint32_t total = rows * H;
	constexpr int U = 4;
	for (uint32_t e0 = tid; e0 < total; e0 += (uint32_t)U * NT) {
		uint32_t a1[U], a2[U], b1[U], b2[U]; T ya1[U], ya2[U], yb1[U], yb2[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			const uint32_t e = e0 + (uint32_t)u * NT;
			const bool live = e < total;
			uint32_t r, k;
			divH.divmod(live ? e : 0u, r, k);
			const uint32_t rA = r * mult;
			const uint32_t offA = (live && rA < realRows) ? rA * pitchBytes : kGbInvalid, offB = (live && mult == 2u && rA + 1u < realRows) ? (rA + 1u) * pitchBytes : kGbInvalid;
			uint32_t o1, o2; cx<T> c1, c2;
			tm_entry<T>(s, k, 0u, o1, o2, c1, c2);
			const cx<T> zk = rd(r, k);
			if (split) {
				const cx<T> zm = rd(r, k ? L - k : 0u);
				const cx<T> xa = {zk.x + zm.x, zk.y - zm.y}, xb = {zk.y + zm.y, zm.x - zk.x};
				ya1[u] = c1.x * xa.x - c1.y * xa.y; ya2[u] = c2.x * xa.x - c2.y * xa.y;
				yb1[u] = c1.x * xb.x - c1.y * xb.y; yb2[u] = c2.x * xb.x - c2.y * xb.y;
				a1[u] = offA + o1; a2[u] = offA + o2; b1[u] = offB + o1; b2[u] = offB + o2;
			} else { // direct forms: y1 to row a at o1; y2 to row b at o1 (real results of a pair) or to row a at o2
				ya1[u] = c1.x * zk.x - c1.y * zk.y; ya2[u] = c2.x * zk.x - c2.y * zk.y;
				a1[u] = offA + o1; a2[u] = rowB ? offB + o1 : offA + o2;
				yb1[u] = yb2[u] = (T)0; b1[u] = b2[u] = kGbInvalid;
			}
		}
		if (cplx) {
#pragma unroll
			for (int u = 0; u < U; u++) { gb_store<T>(data, a1[u], 0, cx<T>{ya1[u], ya2[u]}); gb_store<T>(data, b1[u], 0, cx<T>{yb1[u], yb2[u]}); }
		} else {
#pragma unroll
			for (int u = 0; u < U; u++) {
				gb_store_real<T>(data, a1[u], 0, ya1[u]); gb_store_real<T>(data, a2[u], 0, ya2[u]);
				if (split) { gb_store_real<T>(data, b1[u], 0, yb1[u]); gb_store_real<T>(data, b2[u], 0, yb2[u]); }
			}
		}
	}
}

} // namespace vkfft_mi355x
