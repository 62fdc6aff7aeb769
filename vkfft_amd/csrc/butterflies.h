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

// cos/sin(2*pi*m/R) for the composite radices (generated; exact to double rounding)
__host__ __device__ constexpr double root_cos(int R, int m) {
	if (R == 6) return m == 0 ? 1.00000000000000000000e+00 : m == 1 ? 5.00000000000000111022e-01 : m == 2 ? -4.99999999999999777955e-01 : m == 3 ? -1.00000000000000000000e+00 : m == 4 ? -5.00000000000000444089e-01 : 5.00000000000000111022e-01;
	if (R == 9) return m == 0 ? 1.00000000000000000000e+00 : m == 1 ? 7.66044443118978013452e-01 : m == 2 ? 1.73648177666930414453e-01 : m == 3 ? -4.99999999999999777955e-01 : m == 4 ? -9.39692620785908316883e-01 : m == 5 ? -9.39692620785908427905e-01 : m == 6 ? -5.00000000000000444089e-01 : m == 7 ? 1.73648177666929970364e-01 : 7.66044443118977791407e-01;
	if (R == 10) return m == 0 ? 1.00000000000000000000e+00 : m == 1 ? 8.09016994374947451263e-01 : m == 2 ? 3.09016994374947451263e-01 : m == 3 ? -3.09016994374947340241e-01 : m == 4 ? -8.09016994374947340241e-01 : m == 5 ? -1.00000000000000000000e+00 : m == 6 ? -8.09016994374947562285e-01 : m == 7 ? -3.09016994374947562285e-01 : m == 8 ? 3.09016994374947229218e-01 : 8.09016994374947340241e-01;
	if (R == 12) return m == 0 ? 1.00000000000000000000e+00 : m == 1 ? 8.66025403784438707611e-01 : m == 2 ? 5.00000000000000111022e-01 : m == 3 ? 6.12323399573676603587e-17 : m == 4 ? -4.99999999999999777955e-01 : m == 5 ? -8.66025403784438707611e-01 : m == 6 ? -1.00000000000000000000e+00 : m == 7 ? -8.66025403784438818633e-01 : m == 8 ? -5.00000000000000444089e-01 : m == 9 ? -1.83697019872102968750e-16 : m == 10 ? 5.00000000000000111022e-01 : 8.66025403784438374544e-01;
	if (R == 14) return m == 0 ? 1.00000000000000000000e+00 : m == 1 ? 9.00968867902419145999e-01 : m == 2 ? 6.23489801858733594386e-01 : m == 3 ? 2.22520933956314448388e-01 : m == 4 ? -2.22520933956314337365e-01 : m == 5 ? -6.23489801858733483364e-01 : m == 6 ? -9.00968867902419034976e-01 : m == 7 ? -1.00000000000000000000e+00 : m == 8 ? -9.00968867902419145999e-01 : m == 9 ? -6.23489801858733705409e-01 : m == 10 ? -2.22520933956314587165e-01 : m == 11 ? 2.22520933956313338165e-01 : m == 12 ? 6.23489801858733372342e-01 : 9.00968867902419368043e-01;
	if (R == 15) return m == 0 ? 1.00000000000000000000e+00 : m == 1 ? 9.13545457642600866599e-01 : m == 2 ? 6.69130606358858237570e-01 : m == 3 ? 3.09016994374947451263e-01 : m == 4 ? -1.04528463267653332069e-01 : m == 5 ? -4.99999999999999777955e-01 : m == 6 ? -8.09016994374947340241e-01 : m == 7 ? -9.78147600733805688833e-01 : m == 8 ? -9.78147600733805688833e-01 : m == 9 ? -8.09016994374947562285e-01 : m == 10 ? -5.00000000000000444089e-01 : m == 11 ? -1.04528463267654234126e-01 : m == 12 ? 3.09016994374947229218e-01 : m == 13 ? 6.69130606358858459615e-01 : 9.13545457642600977621e-01;
	return 1.0;
}
__host__ __device__ constexpr double root_sin(int R, int m) {
	if (R == 6) return m == 0 ? 0.00000000000000000000e+00 : m == 1 ? 8.66025403784438596588e-01 : m == 2 ? 8.66025403784438707611e-01 : m == 3 ? 1.22464679914735320717e-16 : m == 4 ? -8.66025403784438374544e-01 : -8.66025403784438596588e-01;
	if (R == 9) return m == 0 ? 0.00000000000000000000e+00 : m == 1 ? 6.42787609686539251896e-01 : m == 2 ? 9.84807753012208020316e-01 : m == 3 ? 8.66025403784438707611e-01 : m == 4 ? 3.42020143325668879442e-01 : m == 5 ? -3.42020143325668657397e-01 : m == 6 ? -8.66025403784438374544e-01 : m == 7 ? -9.84807753012208131338e-01 : -6.42787609686539584963e-01;
	if (R == 10) return m == 0 ? 0.00000000000000000000e+00 : m == 1 ? 5.87785252292473137103e-01 : m == 2 ? 9.51056516295153531182e-01 : m == 3 ? 9.51056516295153642204e-01 : m == 4 ? 5.87785252292473248126e-01 : m == 5 ? 1.22464679914735320717e-16 : m == 6 ? -5.87785252292473026081e-01 : m == 7 ? -9.51056516295153531182e-01 : m == 8 ? -9.51056516295153642204e-01 : -5.87785252292473359148e-01;
	if (R == 12) return m == 0 ? 0.00000000000000000000e+00 : m == 1 ? 4.99999999999999944489e-01 : m == 2 ? 8.66025403784438596588e-01 : m == 3 ? 1.00000000000000000000e+00 : m == 4 ? 8.66025403784438707611e-01 : m == 5 ? 4.99999999999999944489e-01 : m == 6 ? 1.22464679914735320717e-16 : m == 7 ? -4.99999999999999722444e-01 : m == 8 ? -8.66025403784438374544e-01 : m == 9 ? -1.00000000000000000000e+00 : m == 10 ? -8.66025403784438596588e-01 : -5.00000000000000444089e-01;
	if (R == 14) return m == 0 ? 0.00000000000000000000e+00 : m == 1 ? 4.33883739117558120402e-01 : m == 2 ? 7.81831482468029803634e-01 : m == 3 ? 9.74927912181823619342e-01 : m == 4 ? 9.74927912181823619342e-01 : m == 5 ? 7.81831482468029914656e-01 : m == 6 ? 4.33883739117558231424e-01 : m == 7 ? 1.22464679914735320717e-16 : m == 8 ? -4.33883739117558009379e-01 : m == 9 ? -7.81831482468029692612e-01 : m == 10 ? -9.74927912181823619342e-01 : m == 11 ? -9.74927912181823841387e-01 : m == 12 ? -7.81831482468029914656e-01 : -4.33883739117557509779e-01;
	if (R == 15) return m == 0 ? 0.00000000000000000000e+00 : m == 1 ? 4.06736643075800152758e-01 : m == 2 ? 7.43144825477394133095e-01 : m == 3 ? 9.51056516295153531182e-01 : m == 4 ? 9.94521895368273400884e-01 : m == 5 ? 8.66025403784438707611e-01 : m == 6 ? 5.87785252292473248126e-01 : m == 7 ? 2.07911690817759314820e-01 : m == 8 ? -2.07911690817759065020e-01 : m == 9 ? -5.87785252292473026081e-01 : m == 10 ? -8.66025403784438374544e-01 : m == 11 ? -9.94521895368273289861e-01 : m == 12 ? -9.51056516295153642204e-01 : m == 13 ? -7.43144825477394022073e-01 : -4.06736643075800152758e-01;
	return 0.0;
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

template <int R, typename T> __host__ __device__ inline void dft(cx<T>* v);

// ---- composite radices 6, 9, 10, 12, 14, 15: one Cooley-Tukey step in registers (n = B*n1 + n2, k = k1 + A*k2) ----------
// (the reference's radix-6/9/10/12/14/15 butterflies, vkFFT_RadixKernels.h:499-2126, are built the same way)
template <int A, int B, typename T> struct DftComposite {
	__host__ __device__ static inline void run(cx<T>* v) {
		constexpr int R = A * B;
		cx<T> y[R];
#pragma unroll
		for (int n2 = 0; n2 < B; n2++) {
			cx<T> tmp[A];
#pragma unroll
			for (int n1 = 0; n1 < A; n1++) tmp[n1] = v[B * n1 + n2];
			dft<A, T>(tmp);
#pragma unroll
			for (int k1 = 0; k1 < A; k1++) {
				const int m = (n2 * k1) % R;
				if (m == 0) y[n2 * A + k1] = tmp[k1];
				else y[n2 * A + k1] = cmul(tmp[k1], cx<T>{(T)root_cos(R, m), (T)(-root_sin(R, m))});
			}
		}
#pragma unroll
		for (int k1 = 0; k1 < A; k1++) {
			cx<T> tmp[B];
#pragma unroll
			for (int n2 = 0; n2 < B; n2++) tmp[n2] = y[n2 * A + k1];
			dft<B, T>(tmp);
#pragma unroll
			for (int k2 = 0; k2 < B; k2++) v[k1 + A * k2] = tmp[k2];
		}
	}
};

template <int R, typename T> __host__ __device__ inline void dft(cx<T>* v) {
	if constexpr (R == 1) { }
	else if constexpr (R == 2 || R == 4 || R == 8 || R == 16 || R == 32) DftPow2<R, T>::run(v);
	else if constexpr (R == 6) DftComposite<2, 3, T>::run(v);
	else if constexpr (R == 9) DftComposite<3, 3, T>::run(v);
	else if constexpr (R == 10) DftComposite<2, 5, T>::run(v);
	else if constexpr (R == 12) DftComposite<4, 3, T>::run(v);
	else if constexpr (R == 14) DftComposite<2, 7, T>::run(v);
	else if constexpr (R == 15) DftComposite<3, 5, T>::run(v);
	else DftPrime<R, T>::run(v);
}

} // namespace vkfft_mi355x
