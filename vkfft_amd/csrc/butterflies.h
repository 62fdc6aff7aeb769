// In-register forward DFT butterflies (sign -), natural-order in, natural-order out.
// Radix set of the reference's Stockham stages (vkFFT_RadixKernels.h:43-2747): 2,3,4,5,7,8,11,13 (+16, 32,
// which the reference also builds from radix-2 layers).  The inverse direction never needs its own
// butterflies: the kernels use IFFT(x) = swap(FFT(swap(x))).
//
// Power-of-two radices are built by compile-time decimation-in-time recursion with literal twiddles;
// odd primes use the symmetric (cos/sin half-table) form, O((p-1)^2/2) real multiplies, all constants
// folded at compile time.
#pragma once
#include "common.h"

namespace vkfft_mi355x {

// ---- constants -------------------------------------------------------------------------------------
// cos/sin(2*pi*(i+1)/P) for i = 0..(P-3)/2, as constexpr functions so device code never ODR-uses a host array
__host__ __device__ constexpr double prime_cos(int P, int i) {
	if (P == 3) return -5.00000000000000000e-01;
	if (P == 5) return i == 0 ? 3.09016994374947451e-01 : -8.09016994374947340e-01;
	if (P == 7) return i == 0 ? 6.23489801858733594e-01 : i == 1 ? -2.22520933956314337e-01 : -9.00968867902419035e-01;
	if (P == 11) return i == 0 ? 8.41253532831181206e-01 : i == 1 ? 4.15415013001886435e-01 : i == 2 ? -1.42314838273285005e-01 : i == 3 ? -6.54860733945284990e-01 : -9.59492973614497369e-01;
	if (P == 13) return i == 0 ? 8.85456025653209911e-01 : i == 1 ? 5.68064746731155923e-01 : i == 2 ? 1.20536680255323006e-01 : i == 3 ? -3.54604887042535455e-01 : i == 4 ? -7.48510748171101192e-01 : -9.70941817426052012e-01;
	return 0.0;
}
__host__ __device__ constexpr double prime_sin(int P, int i) {
	if (P == 3) return 8.66025403784438708e-01;
	if (P == 5) return i == 0 ? 9.51056516295153531e-01 : 5.87785252292473248e-01;
	if (P == 7) return i == 0 ? 7.81831482468029804e-01 : i == 1 ? 9.74927912181823619e-01 : 4.33883739117558231e-01;
	if (P == 11) return i == 0 ? 5.40640817455597555e-01 : i == 1 ? 9.09631995354518330e-01 : i == 2 ? 9.89821441880932795e-01 : i == 3 ? 7.55749574354258269e-01 : 2.81732556841429671e-01;
	if (P == 13) return i == 0 ? 4.64723172043768507e-01 : i == 1 ? 8.22983865893656352e-01 : i == 2 ? 9.92708874098053973e-01 : i == 3 ? 9.35016242685414833e-01 : i == 4 ? 6.63122658240795193e-01 : 2.39315664287557683e-01;
	return 0.0;
}

// cos/sin(2*pi*k/32), k = 0..15  (w32^k = c - i s)
__host__ __device__ constexpr double pow2_cos32(int k) {
	return k == 0 ? 1.00000000000000000000e+00 : k == 1 ? 9.80785280403230430579e-01 : k == 2 ? 9.23879532511286738483e-01 : k == 3 ? 8.31469612302545235671e-01 : k == 4 ? 7.07106781186547572737e-01 : k == 5 ? 5.55570233019602288671e-01 : k == 6 ? 3.82683432365089837290e-01 : k == 7 ? 1.95090322016128331351e-01 : k == 8 ? 6.12323399573676603587e-17 : k == 9 ? -1.95090322016128192573e-01 : k == 10 ? -3.82683432365089726268e-01 : k == 11 ? -5.55570233019601955604e-01 : k == 12 ? -7.07106781186547461715e-01 : k == 13 ? -8.31469612302545346694e-01 : k == 14 ? -9.23879532511286738483e-01 : -9.80785280403230430579e-01;
}
__host__ __device__ constexpr double pow2_sin32(int k) {
	return k == 0 ? 0.00000000000000000000e+00 : k == 1 ? 1.95090322016128248084e-01 : k == 2 ? 3.82683432365089781779e-01 : k == 3 ? 5.55570233019602177649e-01 : k == 4 ? 7.07106781186547461715e-01 : k == 5 ? 8.31469612302545235671e-01 : k == 6 ? 9.23879532511286738483e-01 : k == 7 ? 9.80785280403230430579e-01 : k == 8 ? 1.00000000000000000000e+00 : k == 9 ? 9.80785280403230430579e-01 : k == 10 ? 9.23879532511286738483e-01 : k == 11 ? 8.31469612302545457716e-01 : k == 12 ? 7.07106781186547572737e-01 : k == 13 ? 5.55570233019602177649e-01 : k == 14 ? 3.82683432365089892802e-01 : 1.95090322016128608906e-01;
}

// ---- power-of-two radices -----------------------------------------------------------------------------
template <int R, typename T> struct DftPow2 {
	// v[0..R) with element stride `str` in a register array; result in natural order, same slots.
	__host__ __device__ static inline void run(cx<T>* v) {
		constexpr int H = R / 2;
		cx<T> e[H], o[H];
#pragma unroll
		for (int i = 0; i < H; i++) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
		DftPow2<H, T>::run(e);
		DftPow2<H, T>::run(o);
#pragma unroll
		for (int k = 0; k < H; k++) {
			constexpr int step = 32 / R; // index into the w32 table
			cx<T> w;
			const int kk = k * step;
			cx<T> t;
			if (kk == 0) t = o[k];
			else if (kk == 8) t = cmul_mi(o[k]);
			else { w.x = (T)pow2_cos32(kk); w.y = (T)(-pow2_sin32(kk)); t = cmul(o[k], w); }
			v[k] = cadd(e[k], t);
			v[k + H] = csub(e[k], t);
		}
	}
};
template <typename T> struct DftPow2<1, T> {
	__host__ __device__ static inline void run(cx<T>*) {}
};
template <typename T> struct DftPow2<2, T> {
	__host__ __device__ static inline void run(cx<T>* v) {
		cx<T> a = v[0], b = v[1];
		v[0] = cadd(a, b);
		v[1] = csub(a, b);
	}
};
template <typename T> struct DftPow2<4, T> {
	__host__ __device__ static inline void run(cx<T>* v) {
		cx<T> a = cadd(v[0], v[2]), b = csub(v[0], v[2]);
		cx<T> c = cadd(v[1], v[3]), d = cmul_mi(csub(v[1], v[3]));
		v[0] = cadd(a, c);
		v[1] = cadd(b, d);
		v[2] = csub(a, c);
		v[3] = csub(b, d);
	}
};

// ---- odd primes ----------------------------------------------------------------------------------------
template <int P, typename T> struct DftPrime {
	__host__ __device__ static inline void run(cx<T>* v) {
		constexpr int H = (P - 1) / 2;
		cx<T> t[H], u[H];
#pragma unroll
		for (int j = 0; j < H; j++) { t[j] = cadd(v[j + 1], v[P - 1 - j]); u[j] = csub(v[j + 1], v[P - 1 - j]); }
		cx<T> x0 = v[0];
		cx<T> sum = x0;
#pragma unroll
		for (int j = 0; j < H; j++) sum = cadd(sum, t[j]);
		v[0] = sum;
#pragma unroll
		for (int k = 1; k <= H; k++) {
			cx<T> a = x0, b = {(T)0, (T)0};
#pragma unroll
			for (int j = 1; j <= H; j++) {
				const int m = (j * k) % P;
				const int mi = m <= H ? m : P - m;
				const T c = (T)prime_cos(P, mi - 1);
				const T s = (T)(m <= H ? prime_sin(P, mi - 1) : -prime_sin(P, mi - 1));
				a.x += c * t[j - 1].x; a.y += c * t[j - 1].y;
				b.x += s * u[j - 1].x; b.y += s * u[j - 1].y;
			}
			// X_k = a - i b ; X_{P-k} = a + i b
			v[k] = {a.x + b.y, a.y - b.x};
			v[P - k] = {a.x - b.y, a.y + b.x};
		}
	}
};

template <int R, typename T> __host__ __device__ inline void dft(cx<T>* v) {
	if constexpr (R == 1) { }
	else if constexpr (R == 2 || R == 4 || R == 8 || R == 16 || R == 32) DftPow2<R, T>::run(v);
	else DftPrime<R, T>::run(v);
}

} // namespace vkfft_mi355x
