/* TEST INFRASTRUCTURE ONLY.  CPU oracle for the MI355X FFT library.
 *
 * A plain-C restatement of the algorithms on the reference's hot path (DTolm/VkFFT v1.3.4): Stockham
 * autosort radix stages, Four-Step multi-upload decomposition, R2C/C2R even decomposition, DCT/DST pre/post
 * maps, Bluestein — each function cites the reference file:line it follows (vkfft_oracle_impl.h).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call this file.
 * The product library (vkfft_amd/lib/libvkfft_mi355x.so) never links or calls it.
 *
 * Pinning: the reference stores no golden vectors.  The oracle is pinned against
 *   (1) outputs of the reference itself (HIP backend, oracle/_ref) captured on an MI355X and committed under
 *       tests/golden/ with the generating script (tests/golden/make_golden.py), and
 *   (2) the FFTW conventions the reference's precision samples use as ground truth
 *       (sample_11/14/15/16: fftw_plan_dft_*, r2c/c2r, REDFT00/10/01/11), taken from scipy.fft/MKL in double.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define REAL float
#define FN(n) f32_##n
#define CPX cpx_f32
#include "vkfft_oracle_impl.h"
#undef REAL
#undef FN
#undef CPX

#define REAL double
#define FN(n) f64_##n
#define CPX cpx_f64
#include "vkfft_oracle_impl.h"
#undef REAL
#undef FN
#undef CPX

/* glibc rand() stream with the default seed 1, as the reference's samples use it unseeded:
 *   x = (float)(2*((float)rand())/RAND_MAX - 1.0)    sample_0_benchmark_VkFFT_single.cpp:73-75 */
void oracle_fill_rand_sample(float* dst, uint64_t n) {
	srand(1);
	for (uint64_t i = 0; i < n; i++) dst[i] = (float)(2 * ((float)rand()) / RAND_MAX - 1.0);
}
