/* TEST / BASELINE INFRASTRUCTURE ONLY.  The reference's CPU path is FFTW (its precision samples call
 * fftw_plan_dft_1d etc., sample_11_precision_VkFFT_single.cpp:116-132).  FFTW itself is not installed in this
 * image; Intel MKL's libmkl_rt exports the FFTW3 API.  This shim dlopen()s it at run time (no link-time
 * dependency) and exposes batched 1D C2C in fp32/fp64 for (a) cross-checking the oracle and (b) the timed
 * "CPU baseline" leg of bench.py. */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

typedef void* plan_t;
typedef plan_t (*plan_many_f)(int, const int*, int, void*, const int*, int, int, void*, const int*, int, int, int, unsigned);
typedef void (*exec_f)(plan_t);
typedef void (*destroy_f)(plan_t);
static void* g_lib = NULL;
static plan_many_f p_many_f32, p_many_f64;
static exec_f ex_f32, ex_f64;
static destroy_f de_f32, de_f64;
typedef int (*init_threads_f)(void);
typedef void (*with_nthreads_f)(int);
static init_threads_f it_f32, it_f64;
static with_nthreads_f nt_f32, nt_f64;
static int g_threads = 0;

int fftw_mkl_available(void) {
	if (g_lib) return 1;
	const char* names[] = {"libmkl_rt.so.1", "/opt/conda/lib/libmkl_rt.so.1", "libmkl_rt.so", "/opt/conda/lib/libmkl_rt.so", "libfftw3f.so.3"};
	for (unsigned i = 0; i < sizeof(names) / sizeof(names[0]) && !g_lib; i++) g_lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
	if (!g_lib) return 0;
	p_many_f32 = (plan_many_f)dlsym(g_lib, "fftwf_plan_many_dft"); ex_f32 = (exec_f)dlsym(g_lib, "fftwf_execute"); de_f32 = (destroy_f)dlsym(g_lib, "fftwf_destroy_plan");
	p_many_f64 = (plan_many_f)dlsym(g_lib, "fftw_plan_many_dft"); ex_f64 = (exec_f)dlsym(g_lib, "fftw_execute"); de_f64 = (destroy_f)dlsym(g_lib, "fftw_destroy_plan");
	it_f32 = (init_threads_f)dlsym(g_lib, "fftwf_init_threads"); nt_f32 = (with_nthreads_f)dlsym(g_lib, "fftwf_plan_with_nthreads");
	it_f64 = (init_threads_f)dlsym(g_lib, "fftw_init_threads"); nt_f64 = (with_nthreads_f)dlsym(g_lib, "fftw_plan_with_nthreads");
	return p_many_f32 && ex_f32 && de_f32 && p_many_f64 && ex_f64 && de_f64;
}

/* FFTW threading API (fftw_init_threads / fftw_plan_with_nthreads): applies to plans created afterwards; returns 1 when the
 * library exports it (MKL's wrappers and threaded FFTW builds do) */
int fftw_mkl_set_threads(int n) {
	if (!fftw_mkl_available() || !it_f32 || !nt_f32 || !it_f64 || !nt_f64) return 0;
	if (!g_threads) { it_f32(); it_f64(); }
	g_threads = n > 0 ? n : 1;
	nt_f32(g_threads); nt_f64(g_threads);
	return 1;
}

/* in-place batched 1D C2C; sign -1 forward / +1 backward; returns seconds per execute averaged over reps
 * (plan creation excluded), or a negative number on failure. */
double fftw_mkl_c2c(void* data, int n, int batch, int sign, int dp, int reps) {
	if (!fftw_mkl_available()) return -1.0;
	const unsigned FFTW_ESTIMATE_ = 1U << 6;
	int nn[1] = {n};
	plan_t p = (dp ? p_many_f64 : p_many_f32)(1, nn, batch, data, NULL, 1, n, data, NULL, 1, n, sign, FFTW_ESTIMATE_);
	if (!p) return -2.0;
	struct timespec t0, t1;
	if (reps > 1) (dp ? ex_f64 : ex_f32)(p); /* timing runs only (the data is transformed in place): untimed warm-up, thread pool start, page touch */
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (int r = 0; r < reps; r++) (dp ? ex_f64 : ex_f32)(p);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	(dp ? de_f64 : de_f32)(p);
	return ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec)) / reps;
}
