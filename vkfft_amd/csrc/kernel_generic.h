// Generic Stockham pass: any length L = product of radices {2,3,4,5,7,8,11,13,16} (+ Rader primes), any
// strided batch enumeration, optional fused pre/post operation (Four-Step twiddle, R2C/C2R packing,
// DCT/DST pre/post processing, Bluestein chirp).  One workgroup transforms T sub-FFTs held in LDS.
//
// Structure (cf. the reference's generated kernel, vkFFT_FFT.h:48-388, §3.3 of SURVEY.md):
//   gather-load  : global -> LDS, lanes mapped for coalescing (along the sub-FFT for unit-stride rows,
//                  across neighbouring sub-FFTs for strided columns), pre-op applied on the fly
//   stage loop   : Stockham autosort radix stages ping-ponging between two LDS buffers
//                  (read t + i*L/R, twiddle, butterfly, write (t-s)*R + s + k*S; vkFFT_RadixStage.h:104-137,
//                   vkFFT_RadixShuffle.h:127-190 compute the same index maps)
//   gather-store : LDS -> global with the post-op, again lane-mapped for coalescing (transposed when
//                  the output side is laid out differently: Four-Step reorder, vkFFT_ReadWrite.h:1405-1476)
// Everything length- or stride-dependent is a kernel argument (PassParams); nothing is generated at
// run time.  The hand-specialised power-of-two kernels in kernel_pow2.h replace this kernel on the
// headline path; this one provides coverage.
#pragma once
#include <type_traits>
#include "butterflies.h"
#include "memops.h"

namespace vkfft_mi355x {

// Access to the elements of ONE sub-FFT on the global side; element index j -> j * strideJ from the sub-FFT's base.
// Io64: plain pointers with 64-bit element offsets (any span).  Io32: CDNA buffer addressing (memops.h), wave-uniform
// tile base + 32-bit per-lane byte offsets (tile span < 2 GiB, checked by the planner).
// Zero padding (PassParams::padIn* / padOut*): elements [padInL, padInL + padInN) read as zero without touching memory, elements of the padded output
// range are not written (unsigned wrap-around makes one compare do for both bounds; N = 0: off).
template <typename T> struct Io64 {
	const void* in; void* out; int64_t inBase, outBase, inSj, outSj;
	uint32_t padInL = 0, padInN = 0, padOutL = 0, padOutN = 0;
	__device__ inline void set_pad(const PassParams& p) { padInL = p.padInL; padInN = p.padInN; padOutL = p.padOutL; padOutN = p.padOutN; }
	__device__ inline cx<T> ldc(uint32_t j) const { if (j - padInL < padInN) return cx<T>{(T)0, (T)0}; return ((const cx<T>*)in)[inBase + (int64_t)j * inSj]; }
	__device__ inline T ldr(uint32_t j) const { if (j - padInL < padInN) return (T)0; return ((const T*)in)[inBase + (int64_t)j * inSj]; }
	__device__ inline void stc(uint32_t j, cx<T> v) const { if (j - padOutL < padOutN) return; ((cx<T>*)out)[outBase + (int64_t)j * outSj] = v; }
	__device__ inline void str(uint32_t j, T v) const { if (j - padOutL < padOutN) return; ((T*)out)[outBase + (int64_t)j * outSj] = v; }
};
template <typename T> struct Io32 {
	GBuf gin, gout; uint32_t inOff, outOff, inSj, outSj; // byte offsets / byte strides; offsets = kGbInvalid for lanes of a partial tile
	uint32_t padInL = 0, padInN = 0, padOutL = 0, padOutN = 0; // zero padding: an element of the padded range gets an out-of-range offset (loads return 0, stores are dropped)
	__device__ inline void set_pad(const PassParams& p) { padInL = p.padInL; padInN = p.padInN; padOutL = p.padOutL; padOutN = p.padOutN; }
	// lanes of a partial tile carry inOff = kGbInvalid: adding j * stride (< 2 GiB, planner span guard) keeps them out of range
	__device__ inline uint32_t ia(uint32_t j) const { return (j - padInL < padInN) ? kGbInvalid : inOff + j * inSj; }
	__device__ inline uint32_t oa(uint32_t j) const { return (j - padOutL < padOutN) ? kGbInvalid : outOff + j * outSj; }
	__device__ inline cx<T> ldc(uint32_t j) const { return gb_load<T>(gin, ia(j), 0); }
	__device__ inline T ldr(uint32_t j) const { return gb_load_real<T>(gin, ia(j), 0); }
	__device__ inline void stc(uint32_t j, cx<T> v) const { gb_store<T>(gout, oa(j), 0, v); }
	__device__ inline void str(uint32_t j, T v) const { gb_store_real<T>(gout, oa(j), 0, v); }
};

// element k of a complex table of the pass: buffer addressing (one 32-bit VGPR offset instead of a 64-bit address per load)
template <typename T> __device__ inline cx<T> table_load(const void* tab, uint32_t k) {
	return gb_load<T>(make_gbuf(tab), k * (uint32_t)sizeof(cx<T>), 0);
}

template <typename T> __device__ inline cx<T> twiddle4(const PassParams& p, uint32_t e) {
	const cx<T>* tab = (const cx<T>*)p.aux;
	const uint32_t lo = e & ((1u << p.fsLoBits) - 1u), hi = e >> p.fsLoBits;
	return cmul(tab[lo], tab[(1u << p.fsLoBits) + hi]);
}

// Run f(std::integral_constant<uint32_t, OP>) with the pass's run-time pre / post operation as a COMPILE-TIME constant: inside f the switch of
// pre_gather / post_store folds to the one case, loops over a thread's elements unroll and their loads are requested back to back.  (Called per
// element with a run-time op, the switch is a branch tree in every iteration of a loop the compiler cannot unroll: a thread then waits for each
// of its dozen loads in turn — measured on R2C / DCT rows between the instance transforms of kernel_mixed.h: 2x the time of the complex transform.)
template <uint32_t OP> struct OpTag { static constexpr uint32_t value = OP; };
template <typename F> __device__ inline void dispatch_pre_op(uint32_t op, const F& f) {
	switch (op) {
	case OP_C2R_EVEN_PRE: f(OpTag<OP_C2R_EVEN_PRE>{}); break;
	case OP_R2C_FULL: f(OpTag<OP_R2C_FULL>{}); break;
	case OP_C2R_FULL: f(OpTag<OP_C2R_FULL>{}); break;
	case OP_DCT2_PRE: f(OpTag<OP_DCT2_PRE>{}); break;
	case OP_DCT3_PRE: f(OpTag<OP_DCT3_PRE>{}); break;
	case OP_DCT2H_PRE: f(OpTag<OP_DCT2H_PRE>{}); break;
	case OP_DCT3H_PRE: f(OpTag<OP_DCT3H_PRE>{}); break;
	case OP_DCT4_PRE: f(OpTag<OP_DCT4_PRE>{}); break;
	case OP_DCT1H_PRE: f(OpTag<OP_DCT1H_PRE>{}); break;
	case OP_NONE: f(OpTag<OP_NONE>{}); break;
	default: f(op); break; // (the DST members, DCT-I / DST-I in their full-length forms: the run-time switch)
	}
}
template <typename F> __device__ inline void dispatch_post_op(uint32_t op, const F& f) {
	switch (op) {
	case OP_R2C_EVEN_POST: f(OpTag<OP_R2C_EVEN_POST>{}); break;
	case OP_R2C_FULL: f(OpTag<OP_R2C_FULL>{}); break;
	case OP_C2R_FULL: f(OpTag<OP_C2R_FULL>{}); break;
	case OP_DCT2_POST: f(OpTag<OP_DCT2_POST>{}); break;
	case OP_DCT3_POST: f(OpTag<OP_DCT3_POST>{}); break;
	case OP_DCT2H_POST: f(OpTag<OP_DCT2H_POST>{}); break;
	case OP_DCT3H_POST: f(OpTag<OP_DCT3H_POST>{}); break;
	case OP_DCT4_POST: f(OpTag<OP_DCT4_POST>{}); break;
	case OP_DCT1H_POST: f(OpTag<OP_DCT1H_POST>{}); break;
	case OP_NONE: f(OpTag<OP_NONE>{}); break;
	default: f(op); break;
	}
}
__device__ inline constexpr uint32_t op_value(uint32_t op) { return op; }
template <uint32_t OP> __device__ inline constexpr uint32_t op_value(OpTag<OP>) { return OP; }

// value that goes to LDS position `pos` of sub-FFT f (before the optional inverse swap)
template <typename T, typename IO>
__device__ inline cx<T> pre_gather(const PassParams& p, const IO& io, uint32_t pos, uint32_t natBase, const uint32_t op) {
	const cx<T> zero = {(T)0, (T)0};
	switch (op) {
	default:
	case OP_NONE:
		return pos < p.inLen ? io.ldc(pos) : zero;
	case OP_C2R_EVEN_PRE: { // opN = real length N, L = N/2
		const uint32_t H = p.opN >> 1;
		cx<T> a = io.ldc(pos);
		cx<T> b = cconj(io.ldc(H - pos));
		cx<T> w = cconj(((const cx<T>*)p.aux)[pos]);
		cx<T> d = cmul(w, csub(a, b));
		cx<T> s = cadd(a, b);
		return {s.x - d.y, s.y + d.x}; // s + i*d
	}
	case OP_R2C_FULL:
		return {io.ldr(pos), (T)0};
	case OP_C2R_FULL: {
		const uint32_t N = p.opN;
		if (pos <= N / 2) return io.ldc(pos);
		return cconj(io.ldc(N - pos));
	}
	case OP_DCT2_PRE: case OP_DST2_PRE: {
		const uint32_t N = p.opN;
		const uint32_t src = pos < (N + 1) / 2 ? 2 * pos : 2 * (N - 1 - pos) + 1;
		T v = io.ldr(src);
		if (op == OP_DST2_PRE && (src & 1)) v = -v;
		return {v, (T)0};
	}
	case OP_DCT3_PRE: case OP_DST3_PRE: { // V_k = e^{+i pi k/2N} (x_k - i x_{N-k}), x_N = 0
		const uint32_t N = p.opN;
		T a, b;
		if (op == OP_DCT3_PRE) {
			a = io.ldr(pos);
			b = pos == 0 ? (T)0 : io.ldr(N - pos);
		} else { // DST-III = (-1)^n DCT-III(reversed input)
			a = io.ldr(N - 1 - pos);
			b = pos == 0 ? (T)0 : io.ldr(pos - 1);
		}
		cx<T> w = cconj(((const cx<T>*)(p.preNat ? p.aux3 : p.aux))[pos]); // (preNat: first pass of a multi-pass plan, whose aux is the Four-Step table: the map's own table is aux3)
		return cmul(w, cx<T>{a, -b});
	}
	case OP_DCT2H_PRE: case OP_DST2H_PRE: { // L = N/2: z[n] = v[2n] + i v[2n+1], v = Makhoul permutation of x
		const uint32_t N = p.opN, H = N >> 1;
		const uint32_t j0 = 2 * pos, j1 = 2 * pos + 1;
		const uint32_t s0 = j0 < H ? 2 * j0 : 2 * (N - 1 - j0) + 1, s1 = j1 < H ? 2 * j1 : 2 * (N - 1 - j1) + 1;
		T a = io.ldr(s0), b = io.ldr(s1);
		if (op == OP_DST2H_PRE) { if (s0 & 1) a = -a; if (s1 & 1) b = -b; }
		return {a, b};
	}
	case OP_DCT3H_PRE: case OP_DST3H_PRE: { // L = N/2: Hermitian V_k = e^{+i pi k/2N}(x_k - i x_{N-k}) folded by the even C2R split
		const uint32_t N = p.opN, H = N >> 1, m = H - pos;
		const bool dst = op == OP_DST3H_PRE;
		auto X = [&](uint32_t k) -> T { return k >= N ? (T)0 : io.ldr(dst ? N - 1 - k : k); }; // x_N = 0; DST-III reads the reversed input
		const auto c = [&](uint32_t k) { return table_load<T>(p.aux, k); };
		const cx<T> a = cmul(cconj(c(pos)), cx<T>{X(pos), -X(N - pos)});
		const cx<T> b = cconj(cmul(cconj(c(m)), cx<T>{X(m), -X(N - m)}));
		const cx<T> w = cconj(table_load<T>(p.aux2, pos));
		const cx<T> d = cmul(w, csub(a, b)), s2 = cadd(a, b);
		return {s2.x - d.y, s2.y + d.x};
	}
	case OP_DCT1H_PRE: { // L = N-1: z[n] = e[2n] + i e[2n+1], e = even extension of x (period 2N-2)
		const uint32_t N = p.opN, M = 2 * N - 2;
		const uint32_t j0 = 2 * pos, j1 = 2 * pos + 1;
		return {io.ldr(j0 < N ? j0 : M - j0), io.ldr(j1 < N ? j1 : M - j1)};
	}
	case OP_DCT1_PRE: { // even extension, L = 2N-2
		const uint32_t N = p.opN, M = 2 * N - 2;
		const uint32_t src = pos < N ? pos : M - pos;
		return {io.ldr(src), (T)0};
	}
	case OP_DST1_PRE: { // odd extension, L = 2N+2
		const uint32_t N = p.opN;
		if (pos == 0 || pos == N + 1) return zero;
		if (pos <= N) return {io.ldr(pos - 1), (T)0};
		return {-io.ldr(2 * N + 1 - pos), (T)0};
	}
	case OP_DCT4_PRE: case OP_DST4_PRE: {
		const uint32_t N = p.opN;
		const bool dst = op == OP_DST4_PRE;
		if ((p.blueN ? p.blueN : p.L) * 2 == N) { // even N: half-length complex FFT
			uint32_t i0 = 2 * pos, i1 = N - 1 - 2 * pos;
			if (dst) { i0 = N - 1 - i0; i1 = N - 1 - i1; }
			T a = io.ldr(i0), b = io.ldr(i1);
			return cmul(cx<T>{a, b}, ((const cx<T>*)(p.preNat ? p.aux3 : p.aux))[pos]);
		}
		if ((p.blueN ? p.blueN : p.L) == N) {
			// odd N, same-length form (vkFFT_R2R.h:414-481, 922-972): with r = 2n + 1 the kernel cos(pi r u / 4N), u = 2k + 1, is even in r and changes sign
			// under r -> r + 4N, so the row extends to every n; sampled at n = 4 i + (N - 1) / 2, i.e. r = 8 i + N, it is the real sequence whose
			// N-point DFT Z gives y[k] = 2 Re(e^{-i pi u / 4} Z[u mod N]) — no zero padding, no twiddle table
			if (pos >= N) return zero;
			const uint32_t m = 4 * pos + (N >> 1);
			uint32_t src; bool neg = false;
			if (m < N) src = m;
			else if (m < 2 * N) { src = 2 * N - 1 - m; neg = true; }
			else if (m < 3 * N) { src = m - 2 * N; neg = true; }
			else if (m < 4 * N) src = 4 * N - 1 - m;
			else src = m - 4 * N;
			const T a = io.ldr(dst ? N - 1 - src : src);
			return {neg ? -a : a, (T)0};
		}
		if (pos >= N) return zero; // L = 2N, zero padded
		T a = io.ldr(dst ? N - 1 - pos : pos);
		return cscale(((const cx<T>*)(p.preNat ? p.aux3 : p.aux))[pos], a);
	}
	case OP_BLUESTEIN_PRE: {
		const uint32_t n = natBase + pos * p.opStrideJ; // natural position inside the transform
		if (n >= p.opN) return zero;
		cx<T> v = io.ldc(pos);
		if (p.bluesteinSwapIn) v = cswap(v);
		return cmulc(v, ((const cx<T>*)(p.aux3 ? p.aux3 : p.aux))[n]);
	}
	}
}

// output element k of sub-FFT f, gathered from LDS buffer `buf` (values already un-swapped by `rd`)
template <typename T, typename IO, typename RD>
__device__ inline void post_store(const PassParams& p, const IO& io, uint32_t k, uint32_t colIdx, uint32_t natBase, RD rd, const uint32_t op) {
	const T sc = (T)p.scale;
	switch (op) {
	default:
	case OP_NONE: {
		cx<T> v = rd(k);
		io.stc(k, cscale(v, sc));
		return;
	}
	case OP_TWIDDLE_4STEP: {
		cx<T> v = cmul(rd(k), twiddle4<T>(p, k * colIdx));
		io.stc(k, cscale(v, sc));
		return;
	}
	case OP_MUL_LUT: { // pointwise multiply by a table indexed with the natural position (Bluestein: FFT(chirp)/M)
		cx<T> v = cmul(rd(k), ((const cx<T>*)p.aux2)[natBase + k * p.opStrideJ]);
		io.stc(k, cscale(v, sc));
		return;
	}
	case OP_R2C_EVEN_POST: { // k in [0, N/2]
		const uint32_t H = p.opN >> 1;
		cx<T> zk = rd(k == H ? 0 : k), zm = cconj(rd(k == 0 ? 0 : H - k));
		cx<T> w = ((const cx<T>*)p.aux)[k];
		cx<T> s = cadd(zk, zm), d = cmul(w, csub(zk, zm));
		// 0.5 * (s - i d)
		io.stc(k, cx<T>{(T)0.5 * sc * (s.x + d.y), (T)0.5 * sc * (s.y - d.x)});
		return;
	}
	case OP_R2C_FULL: {
		io.stc(k, cscale(rd(k), sc));
		return;
	}
	case OP_C2R_FULL: {
		io.str(k, rd(k).x * sc);
		return;
	}
	case OP_DCT2_POST: case OP_DST2_POST: {
		const uint32_t N = p.opN;
		const uint32_t kk = op == OP_DST2_POST ? N - 1 - k : k;
		cx<T> v = cmul(((const cx<T>*)p.aux)[kk], rd(kk));
		io.str(k, (T)2 * sc * v.x);
		return;
	}
	case OP_DCT3_POST: case OP_DST3_POST: { // y[src(m)] = Re v_m : gather form, output index k
		const uint32_t N = p.opN;
		const uint32_t m = (k & 1) ? N - 1 - (k >> 1) : (k >> 1);
		T v = rd(m).x * sc;
		if (op == OP_DST3_POST && (k & 1)) v = -v;
		io.str(k, v);
		return;
	}
	case OP_DCT2H_POST: case OP_DST2H_POST: { // k in [0, N/2]: y[k] = 2 Re(c^k V_k), y[N-k] = -2 Im(c^k V_k), V = even R2C split of Z
		const uint32_t N = p.opN, H = N >> 1;
		const bool dst = op == OP_DST2H_POST;
		const cx<T> zk = rd(k == H ? 0 : k), zm = cconj(rd(k == 0 ? 0 : H - k));
		const cx<T> w = ((const cx<T>*)p.aux2)[k];
		const cx<T> s2 = cadd(zk, zm), d = cmul(w, csub(zk, zm));
		const cx<T> t = cmul(((const cx<T>*)p.aux)[k], cx<T>{s2.x + d.y, s2.y - d.x}); // c^k * 2 V_k
		io.str(dst ? N - 1 - k : k, sc * t.x);
		if (k >= 1 && k < H) io.str(dst ? k - 1 : N - k, -sc * t.y);
		return;
	}
	case OP_DCT3H_POST: case OP_DST3H_POST: { // gather form: output k <- v[m], v[2n] + i v[2n+1] = FFT output n
		const uint32_t N = p.opN;
		const uint32_t m = (k & 1) ? N - 1 - (k >> 1) : (k >> 1);
		const cx<T> z = rd(m >> 1);
		T v = ((m & 1) ? z.y : z.x) * sc;
		if (op == OP_DST3H_POST && (k & 1)) v = -v;
		io.str(k, v);
		return;
	}
	case OP_DCT1_POST:
		io.str(k, rd(k).x * sc);
		return;
	case OP_DCT1H_POST: { // k in [0, N-1]: y[k] = Re X_k, X = even R2C split of Z (H = N-1 complex points)
		const uint32_t H = p.opN - 1;
		const cx<T> zk = rd(k == H ? 0 : k), zm = cconj(rd(k == 0 ? 0 : H - k));
		const cx<T> w = ((const cx<T>*)p.aux)[k];
		const cx<T> s2 = cadd(zk, zm), d = cmul(w, csub(zk, zm));
		io.str(k, (T)0.5 * sc * (s2.x + d.y));
		return;
	}
	case OP_DST1_POST:
		io.str(k, -rd(k + 1).y * sc);
		return;
	case OP_DCT4_POST: case OP_DST4_POST: {
		const uint32_t N = p.opN;
		T v;
		if ((p.blueN ? p.blueN : p.L) * 2 == N) {
			const uint32_t m = (k & 1) ? (N - 1 - k) >> 1 : k >> 1;
			cx<T> c = cmul(rd(m), ((const cx<T>*)p.aux2)[m]);
			v = (k & 1) ? (T)-2 * c.y : (T)2 * c.x;
		} else if ((p.blueN ? p.blueN : p.L) == N) { // odd N, same-length form: y[k] = 2 Re(e^{-i pi u / 4} Z[u mod N]), u = 2k + 1
			const uint32_t u = 2 * k + 1;
			const cx<T> z = rd(u < N ? u : u - N);
			const uint32_t r8 = u & 7u;
			const T c = (r8 == 1 || r8 == 7) ? z.x : -z.x, s = (r8 == 1 || r8 == 3) ? z.y : -z.y;
			v = (T)1.41421356237309504880168872420969807856967 * (c + s);
		} else {
			cx<T> c = cmul(rd(k), ((const cx<T>*)p.aux2)[k]);
			v = (T)2 * c.x;
		}
		if (op == OP_DST4_POST && (k & 1)) v = -v;
		io.str(k, v * sc);
		return;
	}
	case OP_BLUESTEIN_POST: {
		const uint32_t n = natBase + k * p.opStrideJ;
		if (n >= p.opN) return;
		cx<T> v = cmulc(rd(k), ((const cx<T>*)p.aux)[n]);
		if (p.bluesteinSwapOut) v = cswap(v);
		io.stc(k, cscale(v, sc));
		return;
	}
	}
}

// scatter form of the post-maps: FFT output `a` of this sub-FFT with value v -> its output element(s)
template <typename T, typename IO>
__device__ inline void post_scatter(const PassParams& p, const IO& io, const uint32_t a, const cx<T> v, const uint32_t colIdx, const uint32_t nat, const uint32_t op, const uint32_t outLimit) {
	auto rd = [&](uint32_t) { return v; };
	const uint32_t N = p.opN;
	switch (op) {
	default: // OP_NONE, OP_TWIDDLE_4STEP, OP_MUL_LUT, OP_BLUESTEIN_POST, OP_R2C_FULL, OP_C2R_FULL, OP_DCT1_POST: output a <- FFT output a
		if (a < outLimit) post_store<T>(p, io, a, colIdx, nat, rd, op);
		return;
	case OP_DST1_POST:
		if (a >= 1 && a <= N) post_store<T>(p, io, a - 1, colIdx, nat, rd, op);
		return;
	case OP_DCT2_POST:
		post_store<T>(p, io, a, colIdx, nat, rd, op);
		return;
	case OP_DST2_POST:
		post_store<T>(p, io, N - 1 - a, colIdx, nat, rd, op);
		return;
	case OP_DCT3_POST: case OP_DST3_POST:
		post_store<T>(p, io, a < (N + 1) / 2 ? 2 * a : 2 * (N - 1 - a) + 1, colIdx, nat, rd, op);
		return;
	case OP_DCT3H_POST: case OP_DST3H_POST: { // FFT output a = v[2a] + i v[2a+1]; v[m] is output m < N/2 ? 2m : 2(N-1-m)+1
		const uint32_t m0 = 2 * a, m1 = 2 * a + 1, H = N >> 1;
		post_store<T>(p, io, m0 < H ? 2 * m0 : 2 * (N - 1 - m0) + 1, colIdx, nat, rd, op);
		post_store<T>(p, io, m1 < H ? 2 * m1 : 2 * (N - 1 - m1) + 1, colIdx, nat, rd, op);
		return;
	}
	case OP_DCT4_POST: case OP_DST4_POST:
		if ((p.blueN ? p.blueN : p.L) * 2 == N) { // half-length form: FFT output m feeds outputs 2m and N-1-2m
			post_store<T>(p, io, 2 * a, colIdx, nat, rd, op);
			post_store<T>(p, io, N - 1 - 2 * a, colIdx, nat, rd, op);
		} else if ((p.blueN ? p.blueN : p.L) == N) { // odd N, same-length form: FFT output a is Z[u mod N] of the one output with 2k + 1 = a (mod N)
			if (a < N) post_store<T>(p, io, (a & 1) ? (a - 1) >> 1 : (a + N - 1) >> 1, colIdx, nat, rd, op);
		} else if (a < N) post_store<T>(p, io, a, colIdx, nat, rd, op);
		return;
	}
}

template <int R, typename T>
__device__ inline void run_stage(const PassParams& p, const StageDesc& sd, int si, const cx<T>* __restrict__ src,
                                 cx<T>* __restrict__ dst, uint32_t tid, uint32_t nthr) {
	const uint32_t nb = p.L / R;            // butterflies per sub-FFT
	const uint32_t total = nb << p.logT;    // over the T sub-FFTs of the workgroup
	const uint32_t S = sd.S;
	const uint32_t Tp = p.Tp, ps = p.padShift;
	const cx<T>* lut = (const cx<T>*)p.lut + sd.lutOff;
	const bool alongF = p.T >= 16;
	for (uint32_t u = tid; u < total; u += nthr) {
		uint32_t f, t;
		if (alongF) { f = u & (p.T - 1); t = u >> p.logT; }
		else p.divNb[si].divmod(u, f, t);
		uint32_t q, s;
		if (S == 1) { q = t; s = 0; }
		else p.divS[si].divmod(t, q, s);
		cx<T> v[R];
#pragma unroll
		for (int i = 0; i < R; i++) {
			const uint32_t a = t + i * nb;
			v[i] = src[(a + (a >> ps)) * Tp + f];
		}
		if (S > 1) {
#pragma unroll
			for (int i = 1; i < R; i++) v[i] = cmul(v[i], lut[(i - 1) * S + s]);
		}
		dft<R, T>(v);
		const uint32_t ob = q * S * R + s;
#pragma unroll
		for (int k = 0; k < R; k++) {
			const uint32_t a = ob + k * S;
			dst[(a + (a >> ps)) * Tp + f] = v[k];
		}
	}
}

// Rader stage, direct-multiplication form (reference: appendMultRaderStage, vkFFT_RaderKernels.h:1278):
// a prime-radix butterfly evaluated as a dense (P x P) DFT from a table of the P-th roots of unity.
template <typename T>
__device__ inline void run_stage_direct(const PassParams& p, const StageDesc& sd, int si, const cx<T>* __restrict__ src,
                                        cx<T>* __restrict__ dst, uint32_t tid, uint32_t nthr) {
	const uint32_t R = sd.radix;
	const uint32_t nb = p.L / R;
	const uint32_t S = sd.S;
	const uint32_t Tp = p.Tp, ps = p.padShift;
	const cx<T>* lut = (const cx<T>*)p.lut + sd.lutOff;   // stage twiddles (R-1 runs of S)
	const cx<T>* root = (const cx<T>*)p.lut + sd.aux0;    // exp(-2 pi i m / R), m = 0..R-1
	// one thread per output element: (f, t, k)
	const uint32_t total = (nb * R) << p.logT;
	for (uint32_t u = tid; u < total; u += nthr) {
		uint32_t f = u & (p.T - 1), rest = u >> p.logT;
		uint32_t k, t;
		p.divNb[si].divmod(rest, k, t); // k slow so that neighbouring lanes share k (broadcast roots)
		uint32_t q, s;
		if (S == 1) { q = t; s = 0; }
		else p.divS[si].divmod(t, q, s);
		cx<T> acc = {(T)0, (T)0};
		uint32_t m = 0; // (i*k) mod R
		for (uint32_t i = 0; i < R; i++) {
			const uint32_t a = t + i * nb;
			cx<T> x = src[(a + (a >> ps)) * Tp + f];
			if (S > 1 && i > 0) x = cmul(x, lut[(i - 1) * S + s]);
			cx<T> w = root[m];
			acc.x += x.x * w.x - x.y * w.y;
			acc.y += x.x * w.y + x.y * w.x;
			m += k; if (m >= R) m -= R;
		}
		const uint32_t a = q * S * R + s + k * S;
		dst[(a + (a >> ps)) * Tp + f] = acc;
	}
}

// one radix stage of the Rader sub-FFT: length P1 = P-1 over `U` columns (column = one radix-P butterfly), data laid
// out [q][u] (u fastest) inside the ping-pong buffers
template <int R, typename T>
__device__ inline void run_substage(const PassParams& p, int si, const cx<T>* __restrict__ src, cx<T>* __restrict__ dst, uint32_t U, uint32_t tid, uint32_t nthr) {
	const StageDesc& sd = p.rd.sub[si];
	const uint32_t P1 = p.rd.P - 1, nbq = P1 / R, S = sd.S;
	const cx<T>* lut = (const cx<T>*)p.lut + p.rd.subLutOff + sd.lutOff;
	const uint32_t total = nbq * U;
	for (uint32_t v = tid; v < total; v += nthr) {
		uint32_t tt, u;
		p.rd.divU.divmod(v, tt, u);
		uint32_t q, s;
		if (S == 1) { q = tt; s = 0; }
		else p.rd.divSubS[si].divmod(tt, q, s);
		cx<T> x[R];
#pragma unroll
		for (int i = 0; i < R; i++) x[i] = src[(tt + i * nbq) * U + u];
		if (S > 1) {
#pragma unroll
			for (int i = 1; i < R; i++) x[i] = cmul(x[i], lut[(i - 1) * S + s]);
		}
		dft<R, T>(x);
		const uint32_t ob = q * S * R + s;
#pragma unroll
		for (int k = 0; k < R; k++) dst[(ob + k * S) * U + u] = x[k];
	}
}
template <typename T>
__device__ inline void run_subfft(const PassParams& p, cx<T>*& cur, cx<T>*& oth, uint32_t U, uint32_t tid, uint32_t nthr) {
	for (uint32_t si = 0; si < p.rd.nSub; si++) {
		switch (p.rd.sub[si].radix) {
			case 2: run_substage<2, T>(p, si, cur, oth, U, tid, nthr); break;
			case 3: run_substage<3, T>(p, si, cur, oth, U, tid, nthr); break;
			case 4: run_substage<4, T>(p, si, cur, oth, U, tid, nthr); break;
			case 5: run_substage<5, T>(p, si, cur, oth, U, tid, nthr); break;
			case 7: run_substage<7, T>(p, si, cur, oth, U, tid, nthr); break;
			case 8: run_substage<8, T>(p, si, cur, oth, U, tid, nthr); break;
			case 11: run_substage<11, T>(p, si, cur, oth, U, tid, nthr); break;
			case 13: run_substage<13, T>(p, si, cur, oth, U, tid, nthr); break;
			case 16: run_substage<16, T>(p, si, cur, oth, U, tid, nthr); break;
			default: break;
		}
		VKFFT_SYNC();
		cx<T>* t = cur; cur = oth; oth = t;
	}
}

// Rader stage, FFT-convolution form (reference: appendFFTRaderStage, vkFFT_RaderKernels.h:30; tree construction
// vkFFT_Scheduler.h:1733-1873; kernel precompute vkFFT_RecursiveFFTGenerators.h:996-1048).  For prime radix P with
// generator g:  X_0 = sum x_i;  X_{g^-m} = x_0 + (a (*) b)_m,  a_q = x_{g^q},  b_q = exp(-2 pi i g^-q / P),
// the cyclic convolution of length P-1 evaluated as IFFT(FFT(a) . FFT(b)).  All butterflies of the workgroup are
// convolved together: they are the columns of one batched sub-FFT held in the ping-pong buffers.
// Returns the buffer that holds the stage's output.
template <typename T>
__device__ inline cx<T>* run_stage_rader_fft(const PassParams& p, const StageDesc& sd, int si, cx<T>* A, cx<T>* B, cx<T>* tail, uint32_t tid, uint32_t nthr) {
	const uint32_t P = sd.radix, P1 = P - 1;
	const uint32_t nb = p.L / P, S = sd.S;
	const uint32_t U = nb << p.logT; // butterflies (columns)
	const uint32_t Tp = p.Tp, ps = p.padShift;
	const cx<T>* lut = (const cx<T>*)p.lut + sd.lutOff;
	const cx<T>* bhat = (const cx<T>*)p.lut + p.rd.bhatOff;
	const uint32_t* gpow = (const uint32_t*)p.rader + p.rd.gpowOff;
	const uint32_t* ginv = (const uint32_t*)p.rader + p.rd.ginvOff;
	// (1) gather a_q = x_{g^q} * twiddle into B[q][u], x_0 into the tail
	for (uint32_t v = tid; v < P1 * U; v += nthr) {
		uint32_t q, u;
		p.rd.divU.divmod(v, q, u);
		const uint32_t f = u & (p.T - 1), t = u >> p.logT;
		const uint32_t i = gpow[q];
		const uint32_t a = t + i * nb;
		cx<T> x = A[(a + (a >> ps)) * Tp + f];
		if (S > 1) { uint32_t qq, s; p.divS[si].divmod(t, qq, s); x = cmul(x, lut[(i - 1) * S + s]); }
		B[q * U + u] = x;
		if (q == 0) tail[u] = A[(t + (t >> ps)) * Tp + f];
	}
	VKFFT_SYNC();
	cx<T>* cur = B; cx<T>* oth = A;
	run_subfft<T>(p, cur, oth, U, tid, nthr);
	// (2) pointwise product with FFT(b)/(P-1); x_0 enters the zero bin so that every convolution output carries it;
	//     X_0 = x_0 + A_0 replaces x_0 in the tail.  Result is swapped for the inverse transform (swap identity).
	for (uint32_t v = tid; v < P1 * U; v += nthr) {
		uint32_t m, u;
		p.rd.divU.divmod(v, m, u);
		cx<T> a = cur[m * U + u];
		cx<T> c = cmul(a, bhat[m]);
		if (m == 0) { const cx<T> x0 = tail[u]; tail[u] = cadd(x0, a); c = cadd(c, x0); }
		cur[m * U + u] = cswap(c);
	}
	VKFFT_SYNC();
	run_subfft<T>(p, cur, oth, U, tid, nthr);
	// (3) scatter X_{g^-m} = conv_m (+x_0 already inside) and X_0 to the Stockham output positions in the other buffer
	for (uint32_t v = tid; v < P * U; v += nthr) {
		uint32_t kk, u;
		p.rd.divU.divmod(v, kk, u); // kk = 0: X_0, else m = kk-1
		const uint32_t f = u & (p.T - 1), t = u >> p.logT;
		uint32_t qq, s;
		if (S == 1) { qq = t; s = 0; } else p.divS[si].divmod(t, qq, s);
		cx<T> val; uint32_t k;
		if (kk == 0) { val = tail[u]; k = 0; }
		else { val = cswap(cur[(kk - 1) * U + u]); k = ginv[kk - 1]; }
		const uint32_t a = qq * S * P + s + k * S;
		oth[(a + (a >> ps)) * Tp + f] = val;
	}
	VKFFT_SYNC();
	return oth;
}

// ---- rows between the pre- and the post-map of a real transform, shared by the instance kernels (kernel_mixed.h / kernel_mixconv.h, OPS = 1) -------
// The operation is a compile-time constant INSIDE the loop (dispatch_pre_op / dispatch_post_op hoist the switch out of it): with a run-time operation
// every element pays a branch tree of five or six taken branches, which on this machine cost more than the arithmetic of the transform (measured:
// R2C / DCT rows between the instance transforms took 2x the time of the complex transform of the same length).  The loops stay rolled: unrolled,
// they are 4-9x the code of the transform they surround, once per kernel instance.
// divN divides by the row length, rows of the tile at lds + fi * SP, natural order.
// Two real rows per transform (PassParams::pairRows; the reference's mergeSequencesR2C: vkFFT_SharedMemory.h:40, vkFFT_R2C.h:178,450, vkFFT_R2R.h:2083,3663): where
// the pre-map of a row is a REAL sequence (R2C of odd length, DCT / DST-I, -II, odd -IV in their full-length forms) rows 2f and 2f + 1 travel as real and
// imaginary part of ONE complex row, z = a + i b, and the post-map reads the two spectra back through the even / odd split X_a[q] = (Z[q] + conj Z[L-q]) / 2,
// X_b[q] = (Z[q] - conj Z[L-q]) / 2i.  Where the transform's RESULT is real (C2R of odd length, DCT / DST-III: Hermitian pre-maps) the same packing of the
// pre-maps gives the two real rows as real and imaginary part of the result.  Half the transforms per row.  The second row is one more trip through the
// SAME rolled loop body (one inlined copy of the map, as before).
__host__ __device__ inline bool op_pair_result_is_real(uint32_t postOp) { return postOp == OP_C2R_FULL || postOp == OP_DCT3_POST || postOp == OP_DST3_POST; }
template <typename T, typename OPC>
// subM > 1: the row is stored sub-sequence-major (element M a + b at b * subP + a: kernel_mixrad.h); nvalid counts ROWS (two per slot when paired)
__device__ __attribute__((always_inline)) inline void ops_rows_in(const PassParams& p, OPC opc, const FastDiv divN, cx<T>* lds, uint32_t SP, uint32_t TOT, uint32_t nvalid, int64_t inBase, uint32_t nat0, uint32_t subM = 1, uint32_t subP = 0) {
	const uint32_t tid = threadIdx.x, NT = blockDim.x;
	const bool swI = p.swapIn != 0;
	// (pairs only where the operation is a compile-time constant: with the run-time operation — the DST members, DCT-I / DST-I — the body with the second trip
	// and the split outgrows the inliner, and ONE out-of-line copy costs every kernel of the family a call frame: 1.4 KB of scratch, 131 VGPRs, 5-10x the time)
	constexpr bool kCanPair = !std::is_same<OPC, uint32_t>::value;
	const uint32_t nh = (kCanPair && p.pairRows) ? 2u : 1u;
	// (the second row of a pair is a second sweep of the SAME loop, added into what the first one laid down — every thread meets its own elements again, no
	// barrier; the two trips nested inside one sweep cost the instances 20 VGPRs)
#pragma unroll 1
	for (uint32_t h = 0; h < nh; h++) {
#pragma unroll 1
		for (uint32_t idx = tid; idx < TOT; idx += NT) {
			uint32_t fi, pos;
			divN.divmod(idx, fi, pos);
			const uint32_t row = fi * nh + h;
			cx<T> a = {(T)0, (T)0};
			if (row < nvalid) {
				Io64<T> io{p.in, p.out, inBase + (int64_t)row * p.dim[0].inStride, 0, p.inStrideJ, p.outStrideJ};
				io.set_pad(p);
				a = pre_gather<T>(p, io, pos, nat0 + row * p.opStride0, op_value(opc));
			}
			const uint32_t dst = subM > 1 ? (pos % subM) * subP + pos / subM : pos;
			cx<T>* const q = lds + fi * SP + dst;
			if (h == 0u) *q = swI ? cswap(a) : a;
			else { const cx<T> s = *q; *q = swI ? cx<T>{s.x + a.x, s.y - a.y} : cx<T>{s.x - a.y, s.y + a.x}; } // z + i a (in the swapped frame: swap(z + i a))
		}
	}
}
// rows of the tile -> post-map -> global memory; fetch(fi, a) delivers element a of row fi (un-swapped); rows at lds + fi * SP unless `dc` (Rader: element 0 of a row lives in dc[fi])
// Lc = length of the complex row (the mirror index of the paired split); nvalid counts ROWS
template <typename T, typename OPC>
__device__ __attribute__((always_inline)) inline void ops_rows_out(const PassParams& p, OPC opc, const cx<T>* lds, const cx<T>* dc, uint32_t SP, uint32_t FPW, uint32_t nvalid, int64_t outBase, uint32_t nat0, uint32_t Lc = 0) {
	const uint32_t tid = threadIdx.x, NT = blockDim.x;
	const uint32_t total = p.outLen * FPW;
	const bool swO = p.swapOut != 0;
	constexpr bool kCanPair = !std::is_same<OPC, uint32_t>::value;
	const uint32_t nh = (kCanPair && p.pairRows) ? 2u : 1u;
	const bool realResult = op_pair_result_is_real(op_value(opc));
#pragma unroll 1
	for (uint32_t h = 0; h < nh; h++) {
#pragma unroll 1
		for (uint32_t idx = tid; idx < total; idx += NT) {
			uint32_t fi, k;
			p.divOutLen.divmod(idx, fi, k);
			const uint32_t row = fi * nh + h;
			if (row >= nvalid) continue;
			auto rd = [&](uint32_t a) -> cx<T> {
				cx<T> v = (dc && a == 0u) ? dc[fi] : lds[fi * SP + a];
				if (swO) v = cswap(v);
				if constexpr (kCanPair) {
					if (nh == 1u) return v;
					if (realResult) return cx<T>{h ? v.y : v.x, (T)0};
					const uint32_t ma = a ? Lc - a : 0u; // (both operands of the split come from LDS: a select between a register pair and memory went through scratch)
					cx<T> m = (dc && ma == 0u) ? dc[fi] : lds[fi * SP + ma];
					if (swO) m = cswap(m);
					return h ? cx<T>{(T)0.5 * (v.y + m.y), (T)0.5 * (m.x - v.x)} : cx<T>{(T)0.5 * (v.x + m.x), (T)0.5 * (v.y - m.y)};
				} else return v;
			};
			Io64<T> io{p.in, p.out, 0, outBase + (int64_t)row * p.dim[0].outStride, p.inStrideJ, p.outStrideJ};
			io.set_pad(p);
			post_store<T>(p, io, k, 0u, nat0 + row * p.opStride0, rd, op_value(opc));
		}
	}
}

template <typename T> __global__ void __launch_bounds__(1024) generic_pass_kernel(const PassParams p) {
	VKFFT_DYN_SMEM(smem_raw)
	cx<T>* bufA = (cx<T>*)smem_raw;
	cx<T>* bufB = bufA + p.ldsElems;
	const uint32_t tid = threadIdx.x, nthr = blockDim.x;

	// which tile of which (g1,g2)
	uint32_t wg = p.reverseTiles ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
	const uint32_t tile = wg % p.tilesPerG0;
	wg /= p.tilesPerG0;
	const uint32_t g1 = wg % p.dim[1].count, g2 = wg / p.dim[1].count;
	const uint32_t g0base = tile << p.logT;
	const uint32_t remain = p.dim[0].count - g0base;
	const uint32_t nvalid = remain < p.T ? remain : p.T;
	const int64_t inBase = (int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride + (int64_t)g0base * p.dim[0].inStride;
	const int64_t outBase = (int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride + (int64_t)g0base * p.dim[0].outStride;
	const uint32_t Tp = p.Tp, ps = p.padShift;

	// ---- gather-load -------------------------------------------------------------------------------
	{
		const uint32_t total = p.L << p.logT;
		for (uint32_t idx = tid; idx < total; idx += nthr) {
			uint32_t f, pos;
			if (p.colMode) { f = idx & (p.T - 1); pos = idx >> p.logT; }
			else p.divL.divmod(idx, f, pos);
			cx<T> v = {(T)0, (T)0};
			if (f < nvalid) {
				if (p.preNat) { // multi-pass real transform: FFT input n of the row = pre-map of the row's elements
					const uint32_t n = (g0base + f) * p.opStride0 + g1 * p.opStride1 + pos * p.opStrideJ;
					const int64_t rowBase = ((p.natDimMask & 2u) ? 0 : (int64_t)g1 * p.dim[1].inStride) + ((p.natDimMask & 4u) ? 0 : (int64_t)g2 * p.dim[2].inStride);
					const Io64<T> io{p.in, p.out, rowBase, 0, 1, 1};
					v = pre_gather<T>(p, io, n, 0, p.preOp);
				} else {
					Io64<T> io{p.in, p.out, inBase + (int64_t)f * p.dim[0].inStride, 0, p.inStrideJ, p.outStrideJ};
					io.set_pad(p);
					v = pre_gather<T>(p, io, pos, (g0base + f) * p.opStride0 + g1 * p.opStride1, p.preOp);
				}
			}
			if (p.swapIn) v = cswap(v);
			bufA[(pos + (pos >> ps)) * Tp + f] = v;
		}
	}
	VKFFT_SYNC();

	cx<T>* src = bufA;
	cx<T>* dst = bufB;
	const int reps = p.midOp == OP_BLUESTEIN_MID ? 2 : 1;
	for (int rep = 0; rep < reps; rep++) {
		for (uint32_t si = 0; si < p.nStages; si++) {
			const StageDesc& sd = p.st[si];
			if (sd.kind == 2) {
				cx<T>* res = run_stage_rader_fft<T>(p, sd, si, src, dst, bufA + 2 * (size_t)p.ldsElems, tid, nthr);
				cx<T>* other = (res == dst) ? src : dst;
				src = res; dst = other; // the next stage reads the buffer the Rader stage finished in
				continue;
			}
			if (sd.kind == 1) run_stage_direct<T>(p, sd, si, src, dst, tid, nthr);
			else switch (sd.radix) {
				case 2: run_stage<2, T>(p, sd, si, src, dst, tid, nthr); break;
				case 3: run_stage<3, T>(p, sd, si, src, dst, tid, nthr); break;
				case 4: run_stage<4, T>(p, sd, si, src, dst, tid, nthr); break;
				case 5: run_stage<5, T>(p, sd, si, src, dst, tid, nthr); break;
				case 7: run_stage<7, T>(p, sd, si, src, dst, tid, nthr); break;
				case 8: run_stage<8, T>(p, sd, si, src, dst, tid, nthr); break;
				case 11: run_stage<11, T>(p, sd, si, src, dst, tid, nthr); break;
				case 13: run_stage<13, T>(p, sd, si, src, dst, tid, nthr); break;
				case 16: run_stage<16, T>(p, sd, si, src, dst, tid, nthr); break;
				default: break;
			}
			VKFFT_SYNC();
			cx<T>* tmp = src; src = dst; dst = tmp;
		}
		if (rep == 0 && reps == 2) {
			// Bluestein: multiply the spectrum by FFT(chirp) (aux2 already carries 1/L) and run the stage
			// list again as an inverse transform through the swap identity.
			const uint32_t total = p.L << p.logT;
			const cx<T>* bh = (const cx<T>*)p.aux2;
			for (uint32_t idx = tid; idx < total; idx += nthr) {
				uint32_t f, pos;
				if (p.T >= 16) { f = idx & (p.T - 1); pos = idx >> p.logT; }
				else p.divL.divmod(idx, f, pos);
				const uint32_t li = (pos + (pos >> ps)) * Tp + f;
				src[li] = cswap(cmul(src[li], bh[pos]));
			}
			VKFFT_SYNC();
		}
	}

	// ---- gather-store ------------------------------------------------------------------------------
	{
		const uint32_t total = p.outLen << p.logT;
		const bool swapO = p.swapOut || reps == 2;
		for (uint32_t idx = tid; idx < total; idx += nthr) {
			uint32_t f, k;
			if (p.colModeOut) { f = idx & (p.T - 1); k = idx >> p.logT; }
			else p.divOutLen.divmod(idx, f, k);
			if (f >= nvalid) continue;
			auto rd = [&](uint32_t a) -> cx<T> {
				cx<T> v = src[(a + (a >> ps)) * Tp + f];
				return swapO ? cswap(v) : v;
			};
			uint32_t colIdx = 0;
			if (p.postOp == OP_TWIDDLE_4STEP) { if (p.fsColFromDim1) colIdx = g1; else { uint32_t qq, rr; p.fsColDiv.divmod(g0base + f, qq, rr); colIdx = qq; } }
			if (p.postNat) { // multi-pass real transform: FFT output n of the row -> post-map -> the row's elements
				const uint32_t n = (g0base + f) * p.opStride0 + g1 * p.opStride1 + k * p.opStrideJ;
				const int64_t rowBase = ((p.natDimMask & 2u) ? 0 : (int64_t)g1 * p.dim[1].outStride) + ((p.natDimMask & 4u) ? 0 : (int64_t)g2 * p.dim[2].outStride);
				const Io64<T> io{p.in, p.out, 0, rowBase, 1, 1};
				post_scatter<T>(p, io, n, rd(k), 0, 0, p.postOp, p.natOutLen);
				continue;
			}
			Io64<T> io{p.in, p.out, 0, outBase + (int64_t)f * p.dim[0].outStride, p.inStrideJ, p.outStrideJ};
			io.set_pad(p);
			post_store<T>(p, io, k, colIdx, (g0base + f) * p.opStride0 + g1 * p.opStride1, rd, p.postOp);
		}
	}
}

// Post / pre pass of the even-length R2C / C2R decomposition for rows that need a multi-pass half-length FFT
// (reference: shaderGen_R2C_even_decomposition, vkFFT_R2C_even_decomposition.h:40, math :132-230; plan
// vkFFT_Plan_R2C.h:30).  In place on rows of H+1 complex (H = N/2): thread k handles the conjugate pair (k, H-k).
//   forward: X_k = 1/2[(Z_k + conj Z_{H-k}) - i w^k (Z_k - conj Z_{H-k})],  w = exp(-2 pi i / N)
//   inverse: Z_k = (X_k + conj X_{H-k}) + i conj(w^k) (X_k - conj X_{H-k})      (unnormalised, = 2x the packed spectrum)
template <typename T> __global__ void __launch_bounds__(256) r2c_even_pair_kernel(const PassParams p) {
	const uint32_t H = p.opN >> 1;
	const uint32_t npair = H / 2 + 1;
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t row = p.reverseTiles ? gridDim.y - 1u - blockIdx.y : blockIdx.y; // zig-zag sweep
	if (k >= npair) return;
	const uint32_t g0 = row % p.dim[0].count; row /= p.dim[0].count;
	const uint32_t g1 = row % p.dim[1].count, g2 = row / p.dim[1].count;
	cx<T>* z = (cx<T>*)p.out + ((int64_t)g0 * p.dim[0].outStride + (int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride);
	const cx<T> w = twiddle4<T>(p, k); // exp(-2 pi i k / N)
	const uint32_t km = H - k;
	const T sc = (T)p.scale;
	if (!p.swapIn) { // forward
		const cx<T> a = z[k], b = k == 0 ? z[0] : z[km];
		if (k == 0) {
			z[0] = cx<T>{(a.x + a.y) * sc, (T)0};
			z[H] = cx<T>{(a.x - a.y) * sc, (T)0};
			return;
		}
		const cx<T> s = cadd(a, cconj(b)), d = cmul(w, csub(a, cconj(b)));
		z[k] = cx<T>{(T)0.5 * sc * (s.x + d.y), (T)0.5 * sc * (s.y - d.x)};
		if (km != k) {
			const cx<T> w2 = cx<T>{-w.x, w.y}; // w^(H-k) = -conj(w^k)
			const cx<T> s2 = cadd(b, cconj(a)), d2 = cmul(w2, csub(b, cconj(a)));
			z[km] = cx<T>{(T)0.5 * sc * (s2.x + d2.y), (T)0.5 * sc * (s2.y - d2.x)};
		}
	} else { // inverse pre-pass
		const cx<T> a = z[k], b = z[km];
		const cx<T> wc = cconj(w);
		const cx<T> s = cadd(a, cconj(b)), d = cmul(wc, csub(a, cconj(b)));
		z[k] = cx<T>{(s.x - d.y) * sc, (s.y + d.x) * sc};
		if (km != k && k != 0) {
			const cx<T> wc2 = cx<T>{-w.x, -w.y}; // conj(w^(H-k)) = -w^k
			const cx<T> s2 = cadd(b, cconj(a)), d2 = cmul(wc2, csub(b, cconj(a)));
			z[km] = cx<T>{(s2.x - d2.y) * sc, (s2.y + d2.x) * sc};
		}
	}
}

} // namespace vkfft_mi355x
