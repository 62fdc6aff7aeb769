/* TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's algorithms (see vkfft_oracle.c for the
 * header).  This file is included twice, once with REAL=float and once with REAL=double.
 * Arithmetic is carried out in type REAL exactly where the reference's generated kernels compute in the
 * plan precision; twiddles are produced in double (long double for REAL=double) and rounded once, as the
 * reference's LUT path does (vkFFT_ManageLUT.h:973-1120). */

typedef struct { REAL x, y; } CPX;

static inline CPX FN(cmul)(CPX a, CPX b) { CPX r = {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; return r; }
static inline CPX FN(tw)(long double num, long double den, int dir) { /* exp(dir*2*pi*i*num/den) */
	long double a = 2.0L * 3.14159265358979323846264338327950288L * (num / den);
	CPX r = {(REAL)cosl(a), (REAL)(dir * sinl(a))};
	return r;
}

/* One Stockham autosort pass of length n over a contiguous vector (reference device loop):
 *   inputs  x[t + i*n/R]                         vkFFT_RadixStage.h:136-137
 *   s = t mod S, twiddle w_i = exp(dir*2*pi*i*i_*s/(R*S))   vkFFT_RadixStage.h:113-126, vkFFT_RadixKernels.h:360,384
 *   R-point DFT (the reference inlines Winograd-style butterflies, vkFFT_RadixKernels.h:43-2747; a plain
 *   O(R^2) DFT in the same precision is used here - it is the same linear map)
 *   outputs y[(t - s)*R + s + k*S]               vkFFT_RadixShuffle.h:136-141,156-157
 * Radix order: the reference picks composite radices from large to small (vkFFT_Scheduler.h:3230-3237);
 * here 13,11,7,5,3, then 8/4/2. */
static void FN(stockham)(CPX* x, CPX* scratch, size_t n, int dir) {
	size_t radices[64]; int ns = 0; size_t m = n;
	const size_t odd[5] = {13, 11, 7, 5, 3};
	for (int i = 0; i < 5; i++) while (m % odd[i] == 0) { radices[ns++] = odd[i]; m /= odd[i]; }
	while (m % 8 == 0) { radices[ns++] = 8; m /= 8; }
	while (m % 4 == 0) { radices[ns++] = 4; m /= 4; }
	while (m % 2 == 0) { radices[ns++] = 2; m /= 2; }
	/* m == 1 guaranteed by the caller */
	CPX* src = x; CPX* dst = scratch;
	size_t S = 1;
	for (int st = 0; st < ns; st++) {
		const size_t R = radices[st], nb = n / R;
		CPX root[16];
		for (size_t k = 0; k < R; k++) root[k] = FN(tw)((long double)k, (long double)R, dir);
		for (size_t t = 0; t < nb; t++) {
			const size_t s = t % S;
			CPX v[16];
			for (size_t i = 0; i < R; i++) {
				CPX a = src[t + i * nb];
				if (S > 1 && i > 0) a = FN(cmul)(a, FN(tw)((long double)(i * s), (long double)(R * S), dir));
				v[i] = a;
			}
			for (size_t k = 0; k < R; k++) {
				CPX acc = v[0];
				for (size_t i = 1; i < R; i++) { CPX p = FN(cmul)(v[i], root[(i * k) % R]); acc.x += p.x; acc.y += p.y; }
				dst[(t - s) * R + s + k * S] = acc;
			}
		}
		S *= R;
		CPX* tmp = src; src = dst; dst = tmp;
	}
	if (src != x) memcpy(x, src, n * sizeof(CPX));
}

static int FN(is_smooth13)(size_t n) {
	const size_t p[6] = {2, 3, 5, 7, 11, 13};
	for (int i = 0; i < 6; i++) while (n % p[i] == 0) n /= p[i];
	return n == 1;
}

/* Bluestein (chirp-z) for lengths with prime factors > 13.
 * chirp b_n = exp(i*pi*n^2/N) with n^2 reduced mod 2N     vkFFT_RecursiveFFTGenerators.h:139-148
 * a_n = x_n * conj(b_n), zero-pad to M >= 2N-1, FFT_M, multiply by FFT_M(b), IFFT_M, multiply by conj(b_k)
 *                                                           vkFFT_Bluestein.h:32,201; SURVEY A.4
 * (The reference also has Rader paths for such primes - vkFFT_RaderKernels.h:30,1278 - which compute the same
 *  DFT; the oracle restates only the universal fallback.) */
static void FN(bluestein)(CPX* x, size_t n, int dir) {
	size_t M = 1; while (M < 2 * n - 1) M *= 2;
	CPX* a = (CPX*)calloc(M, sizeof(CPX)); CPX* b = (CPX*)calloc(M, sizeof(CPX)); CPX* s = (CPX*)malloc(M * sizeof(CPX));
	CPX* chirp = (CPX*)malloc(n * sizeof(CPX));
	for (size_t k = 0; k < n; k++) {
		unsigned long long e = (unsigned long long)(((unsigned __int128)k * k) % (2 * n));
		chirp[k] = FN(tw)((long double)e, (long double)(2 * n), -dir); /* forward (dir=-1): exp(+i*pi*k^2/n) */
		CPX cj = {chirp[k].x, -chirp[k].y};
		a[k] = FN(cmul)(x[k], cj);
		b[k] = chirp[k]; if (k) b[M - k] = chirp[k];
	}
	FN(stockham)(a, s, M, -1); FN(stockham)(b, s, M, -1);
	for (size_t k = 0; k < M; k++) a[k] = FN(cmul)(a[k], b[k]);
	FN(stockham)(a, s, M, +1);
	for (size_t k = 0; k < n; k++) {
		CPX cj = {chirp[k].x, -chirp[k].y};
		CPX v = FN(cmul)(a[k], cj);
		x[k].x = v.x / (REAL)M; x[k].y = v.y / (REAL)M;
	}
	free(a); free(b); free(s); free(chirp);
}

/* Four-Step decomposition N = N0*N1 as the reference executes it:
 *  upload 1: for every a in [0,N0) a length-N1 FFT over elements a + N0*b   vkFFT_ReadWrite.h:1469-1476
 *            then multiply element b' by exp(dir*2*pi*i*a*b'/(N0*N1))       vkFFT_4step.h:54-104
 *  upload 0: length-N0 FFTs over contiguous runs, written transposed so that the result is in
 *            natural order (reorderFourStep)                                  vkFFT_ReadWrite.h:1405-1424,1457-1461 */
static void FN(fourstep)(CPX* x, size_t n, size_t n0, int dir) {
	const size_t n1 = n / n0;
	CPX* col = (CPX*)malloc((n1 > n0 ? n1 : n0) * sizeof(CPX)); CPX* s = (CPX*)malloc((n1 > n0 ? n1 : n0) * sizeof(CPX));
	CPX* tmp = (CPX*)malloc(n * sizeof(CPX));
	for (size_t a = 0; a < n0; a++) {
		for (size_t b = 0; b < n1; b++) col[b] = x[a + n0 * b];
		FN(stockham)(col, s, n1, dir);
		for (size_t b = 0; b < n1; b++) tmp[a + n0 * b] = FN(cmul)(col[b], FN(tw)((long double)(a * b), (long double)n, dir));
	}
	for (size_t b = 0; b < n1; b++) {
		memcpy(col, tmp + n0 * b, n0 * sizeof(CPX));
		FN(stockham)(col, s, n0, dir);
		for (size_t k = 0; k < n0; k++) x[b + n1 * k] = col[k];
	}
	free(col); free(s); free(tmp);
}

/* 1D C2C on a strided vector; dir = -1 forward, +1 inverse; unnormalised (vkFFT_FFT.h:154, InitializeApp.h:1307).
 * n0 != 0 forces the Four-Step route with that split (used to restate multi-upload plans). */
static void FN(c2c_strided)(CPX* data, size_t n, ptrdiff_t stride, int dir, size_t n0) {
	CPX* v = (CPX*)malloc(n * sizeof(CPX)); CPX* s = (CPX*)malloc(n * sizeof(CPX));
	for (size_t i = 0; i < n; i++) v[i] = data[(ptrdiff_t)i * stride];
	if (!FN(is_smooth13)(n)) FN(bluestein)(v, n, dir);
	else if (n0 > 1 && n % n0 == 0 && n0 < n) FN(fourstep)(v, n, n0, dir);
	else FN(stockham)(v, s, n, dir);
	for (size_t i = 0; i < n; i++) data[(ptrdiff_t)i * stride] = v[i];
	free(v); free(s);
}

/* N-dimensional C2C, WHD layout (size[0] fastest), batch outermost.  Axis order as the reference:
 * forward 0..d-1, inverse d-1..0 (vkFFT_RunApp.h:114-321, :469-648). */
void FN(oracle_c2c)(REAL* data, int ndim, const uint64_t* size, uint64_t batch, int inverse, uint64_t split0) {
	CPX* d = (CPX*)data;
	size_t tot = 1; for (int i = 0; i < ndim; i++) tot *= size[i];
	const int dir = inverse ? +1 : -1;
	for (uint64_t b = 0; b < batch; b++) {
		CPX* base = d + b * tot;
		for (int ai = 0; ai < ndim; ai++) {
			const int ax = inverse ? ndim - 1 - ai : ai;
			size_t stride = 1; for (int i = 0; i < ax; i++) stride *= size[i];
			const size_t n = size[ax];
			if (n == 1) continue;
			for (size_t i = 0; i < tot; i++) {
				if ((i / stride) % n != 0) continue; /* i is the first element of a line along ax */
				FN(c2c_strided)(base + i, n, (ptrdiff_t)stride, dir, ax == 0 ? (size_t)split0 : 0);
			}
		}
	}
}

/* R2C forward along axis 0 of rows of N reals -> N/2+1 complex, "even decomposition":
 *   Z = FFT_{N/2}(x_{2j} + i x_{2j+1});  X_k = 1/2[(Z_k + conj Z_{N/2-k}) - i w^k (Z_k - conj Z_{N/2-k})], w = exp(-2 pi i/N)
 *   vkFFT_R2C_even_decomposition.h:181-230 (DC/Nyquist special case :132-179).  Odd N: full-length C2C of the real
 *   data, first N/2+1 outputs (the reference's "callback" R2C, vkFFT_R2C.h:27-177).
 * in: rows x N reals (row pitch inPitch reals); out: rows x (N/2+1) complex (row pitch outPitch complex). */
void FN(oracle_r2c_rows)(const REAL* in, REAL* out, uint64_t N, uint64_t rows, uint64_t inPitch, uint64_t outPitch) {
	const size_t H = N / 2;
	CPX* z = (CPX*)malloc((N + 1) * sizeof(CPX)); CPX* X = (CPX*)malloc((N + 1) * sizeof(CPX));
	for (uint64_t r = 0; r < rows; r++) {
		const REAL* x = in + r * inPitch; CPX* o = (CPX*)out + r * outPitch;
		if (N % 2 == 0 && N >= 2) {
			for (size_t j = 0; j < H; j++) { z[j].x = x[2 * j]; z[j].y = x[2 * j + 1]; }
			FN(c2c_strided)(z, H, 1, -1, 0);
			for (size_t k = 0; k <= H; k++) {
				CPX zk = z[k % H], zm = z[(H - k) % H]; zm.y = -zm.y;
				CPX w = FN(tw)((long double)k, (long double)N, -1);
				CPX s = {zk.x + zm.x, zk.y + zm.y}, dlt = {zk.x - zm.x, zk.y - zm.y};
				CPX d2 = FN(cmul)(w, dlt);
				X[k].x = (REAL)0.5 * (s.x + d2.y); X[k].y = (REAL)0.5 * (s.y - d2.x);
			}
		} else {
			for (size_t j = 0; j < N; j++) { z[j].x = x[j]; z[j].y = 0; }
			FN(c2c_strided)(z, N, 1, -1, 0);
			for (size_t k = 0; k <= H; k++) X[k] = z[k];
		}
		for (size_t k = 0; k <= H; k++) o[k] = X[k];
	}
	free(z); free(X);
}

/* C2R inverse (unnormalised: C2R(R2C(x)) = N x, as the reference's round-trip sample expects,
 * sample_15_precision_VkFFT_single_r2c.cpp:143-209).  Mirror of the above (vkFFT_R2C.h:178, even
 * decomposition pre-pass vkFFT_R2C_even_decomposition.h:181-230 run backwards). */
void FN(oracle_c2r_rows)(const REAL* in, REAL* out, uint64_t N, uint64_t rows, uint64_t inPitch, uint64_t outPitch) {
	const size_t H = N / 2;
	CPX* z = (CPX*)malloc((N + 1) * sizeof(CPX));
	for (uint64_t r = 0; r < rows; r++) {
		const CPX* X = (const CPX*)in + r * inPitch; REAL* x = out + r * outPitch;
		if (N % 2 == 0 && N >= 2) {
			for (size_t k = 0; k < H; k++) {
				CPX a = X[k], b = X[H - k]; b.y = -b.y;
				CPX w = FN(tw)((long double)k, (long double)N, +1);
				CPX s = {a.x + b.x, a.y + b.y}, dlt = {a.x - b.x, a.y - b.y};
				CPX d2 = FN(cmul)(w, dlt);
				z[k].x = s.x - d2.y; z[k].y = s.y + d2.x;
			}
			FN(c2c_strided)(z, H, 1, +1, 0);
			for (size_t j = 0; j < H; j++) { x[2 * j] = z[j].x; x[2 * j + 1] = z[j].y; }
		} else {
			for (size_t k = 0; k < N; k++) { if (k <= H) z[k] = X[k]; else { z[k] = X[N - k]; z[k].y = -z[k].y; } }
			FN(c2c_strided)(z, N, 1, +1, 0);
			for (size_t j = 0; j < N; j++) x[j] = z[j].x;
		}
	}
	free(z);
}

/* DCT-I..IV of a strided real vector, FFTW REDFT00/10/01/11 conventions (unnormalised), computed through
 * the complex-FFT mappings the reference uses:
 *   type 1: even extension of length 2N-2, read back Re             vkFFT_R2R.h:28 (index map), size rule Scheduler.h:2271-2280
 *   type 2: even/odd reorder, same-length C2C, 2 Re(e^{-i pi k/2N} V_k)   vkFFT_R2R.h:193-229, :784-859
 *   type 3: the transpose of type 2 (pre-twiddle, inverse C2C, un-reorder) vkFFT_R2R.h:193-229, :784-859
 *   type 4: even N through an N/2-point C2C with pre/post twiddles   vkFFT_R2R.h:368,414, :861-1031
 *           odd N in the same-length form: signed permutation, N-point C2C of the real sequence, +-Re +-Im   vkFFT_R2R.h:414-481, :922-972, :1032-1272 */
void FN(oracle_dct_strided)(REAL* data, uint64_t N, int64_t stride, int type) {
	CPX* v = (CPX*)calloc(2 * N + 2, sizeof(CPX));
	REAL* x = (REAL*)malloc(N * sizeof(REAL));
	for (size_t i = 0; i < N; i++) x[i] = data[(ptrdiff_t)i * stride];
	if (type == 1) {
		const size_t M = 2 * N - 2;
		if (N == 1) { free(v); free(x); return; }
		for (size_t m = 0; m < M; m++) { v[m].x = x[m < N ? m : M - m]; v[m].y = 0; }
		FN(c2c_strided)(v, M, 1, -1, 0);
		for (size_t k = 0; k < N; k++) data[(ptrdiff_t)k * stride] = v[k].x;
	} else if (type == 2) {
		for (size_t m = 0; m < N; m++) { size_t src = m < (N + 1) / 2 ? 2 * m : 2 * (N - 1 - m) + 1; v[m].x = x[src]; v[m].y = 0; }
		FN(c2c_strided)(v, N, 1, -1, 0);
		for (size_t k = 0; k < N; k++) { CPX w = FN(tw)((long double)k, (long double)(4 * N), -1); data[(ptrdiff_t)k * stride] = 2 * FN(cmul)(w, v[k]).x; }
	} else if (type == 3) {
		for (size_t k = 0; k < N; k++) {
			CPX w = FN(tw)((long double)k, (long double)(4 * N), +1);
			CPX a = {x[k], k == 0 ? (REAL)0 : -x[N - k]};
			v[k] = FN(cmul)(w, a);
		}
		FN(c2c_strided)(v, N, 1, +1, 0);
		for (size_t m = 0; m < N; m++) { size_t dst = m < (N + 1) / 2 ? 2 * m : 2 * (N - 1 - m) + 1; data[(ptrdiff_t)dst * stride] = v[m].x; }
	} else {
		if (N % 2 == 0) {
			const size_t H = N / 2;
			for (size_t n = 0; n < H; n++) { CPX a = {x[2 * n], x[N - 1 - 2 * n]}; v[n] = FN(cmul)(a, FN(tw)((long double)(4 * n + 1), (long double)(8 * N), -1)); }
			FN(c2c_strided)(v, H, 1, -1, 0);
			for (size_t k = 0; k < H; k++) {
				CPX c = FN(cmul)(v[k], FN(tw)((long double)k, (long double)(2 * N), -1));
				data[(ptrdiff_t)(2 * k) * stride] = 2 * c.x; data[(ptrdiff_t)(N - 1 - 2 * k) * stride] = -2 * c.y;
			}
		} else if (N < 3) {
			for (size_t n = 0; n < N; n++) { CPX w = FN(tw)((long double)n, (long double)(4 * N), -1); v[n].x = w.x * x[n]; v[n].y = w.y * x[n]; }
			FN(c2c_strided)(v, 2 * N, 1, -1, 0);
			for (size_t k = 0; k < N; k++) { CPX w = FN(tw)((long double)(2 * k + 1), (long double)(8 * N), -1); data[(ptrdiff_t)k * stride] = 2 * FN(cmul)(w, v[k]).x; }
		} else {
			/* odd N: the reference's same-length form — read map vkFFT_R2R.h:414-481 (position i takes element 4 i + N/2 of the row continued
			 * evenly about -1/2 and with a sign change every 2N), signs :922-972, an N-point transform of that REAL sequence, write map and the
			 * +-Re +-Im combination times sqrt 2 :549-699, :1032-1272.  In closed form: with r = 2n + 1 and u = 2k + 1 the kernel
			 * cos(pi r u / 4N) sampled at r = 8 i + N is cos(2 pi i u / N + pi u / 4), so y[k] = 2 Re(e^{-i pi u / 4} Z[u mod N]). */
			for (size_t i = 0; i < N; i++) {
				const size_t m = 4 * i + N / 2;
				REAL val;
				if (m < N) val = x[m];
				else if (m < 2 * N) val = -x[2 * N - 1 - m];
				else if (m < 3 * N) val = -x[m - 2 * N];
				else if (m < 4 * N) val = x[4 * N - 1 - m];
				else val = x[m - 4 * N];
				v[i].x = val; v[i].y = 0;
			}
			FN(c2c_strided)(v, N, 1, -1, 0);
			for (size_t k = 0; k < N; k++) {
				const size_t u = 2 * k + 1;
				CPX w = FN(tw)((long double)(u % 8), (long double)8, -1);
				data[(ptrdiff_t)k * stride] = 2 * FN(cmul)(w, v[u % N]).x;
			}
		}
	}
	free(v); free(x);
}

/* DST-I..IV (FFTW RODFT00/10/01/11) through the DCT identities (the reference's DST kernels apply the same
 * sign/reversal maps around its DCT code, vkFFT_R2R.h:1541-3552). */
void FN(oracle_dst_strided)(REAL* data, uint64_t N, int64_t stride, int type) {
	REAL* x = (REAL*)malloc(N * sizeof(REAL));
	for (size_t i = 0; i < N; i++) x[i] = data[(ptrdiff_t)i * stride];
	if (type == 1) {
		const size_t M = 2 * N + 2;
		CPX* v = (CPX*)calloc(M, sizeof(CPX));
		for (size_t n = 0; n < N; n++) { v[n + 1].x = x[n]; v[M - 1 - n].x = -x[n]; }
		FN(c2c_strided)(v, M, 1, -1, 0);
		for (size_t k = 0; k < N; k++) data[(ptrdiff_t)k * stride] = -v[k + 1].y;
		free(v);
	} else if (type == 2) {
		for (size_t n = 0; n < N; n++) data[(ptrdiff_t)n * stride] = (n & 1) ? -x[n] : x[n];
		FN(oracle_dct_strided)(data, N, stride, 2);
		for (size_t k = 0; k < N; k++) x[k] = data[(ptrdiff_t)k * stride];
		for (size_t k = 0; k < N; k++) data[(ptrdiff_t)k * stride] = x[N - 1 - k];
	} else {
		for (size_t n = 0; n < N; n++) data[(ptrdiff_t)n * stride] = x[N - 1 - n];
		FN(oracle_dct_strided)(data, N, stride, type);
		for (size_t k = 0; k < N; k++) if (k & 1) data[(ptrdiff_t)k * stride] = -data[(ptrdiff_t)k * stride];
	}
	free(x);
}

/* N-dimensional R2R: the 1D transform along every axis (separable), WHD layout, batch outermost. */
void FN(oracle_r2r)(REAL* data, int ndim, const uint64_t* size, uint64_t batch, int type, int dst) {
	size_t tot = 1; for (int i = 0; i < ndim; i++) tot *= size[i];
	for (uint64_t b = 0; b < batch; b++) {
		REAL* base = data + b * tot;
		for (int ax = 0; ax < ndim; ax++) {
			size_t stride = 1; for (int i = 0; i < ax; i++) stride *= size[i];
			const size_t n = size[ax];
			if (n == 1) continue;
			for (size_t i = 0; i < tot; i++) {
				if ((i / stride) % n != 0) continue;
				if (dst) FN(oracle_dst_strided)(base + i, n, (int64_t)stride, type);
				else FN(oracle_dct_strided)(base + i, n, (int64_t)stride, type);
			}
		}
	}
}
