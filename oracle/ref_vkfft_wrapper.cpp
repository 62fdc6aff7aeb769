// TEST INFRASTRUCTURE ONLY (oracle/_ref): thin driver around the *reference* VkFFT (HIP backend,
// hiprtc run-time kernels), compiled from the sources where they lie under /root/reference by
// oracle/build_ref.sh.  Never linked into the product library.  It gives (a) a GPU comparison oracle
// (ref_transform: run the reference on host data and return its output) and (b) the measured
// "VkFFT-HIP on MI355X" baseline with the sample-0 timing protocol
// (/root/reference/benchmark_scripts/vkFFT_scripts/src/sample_0_benchmark_VkFFT_single.cpp:80-263,
//  utils_VkFFT.cpp:920-933).
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <hip/hip_complex.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include "vkFFT.h"

extern "C" {

// kind: 0 C2C, 1 R2C/C2R (in-place padded layout), 11..14 DCT-I..IV, 21..24 DST-I..IV
// dims: FFTdim, size[3]; batch; doublePrecision; inverse (0 fwd, 1 inv); normalize
// data: host pointer, nbytes bytes, transformed in place (H2D, run, D2H).
int ref_transform(int kind, int fftdim, const uint64_t* size, uint64_t batch, int dp, int inverse,
                  int normalize, void* data, uint64_t nbytes, uint64_t* uploads_out) {
    hipError_t e = hipInit(0); if (e != hipSuccess) return -1;
    hipDevice_t dev; if (hipDeviceGet(&dev, 0) != hipSuccess) return -2;
    hipSetDevice(0);
    void* buf = nullptr; if (hipMalloc(&buf, nbytes) != hipSuccess) return -3;
    hipMemcpy(buf, data, nbytes, hipMemcpyHostToDevice);
    VkFFTConfiguration cfg = {}; VkFFTApplication app = {};
    cfg.FFTdim = fftdim; for (int i = 0; i < fftdim; i++) cfg.size[i] = size[i];
    cfg.numberBatches = batch; cfg.device = &dev; cfg.buffer = &buf; uint64_t bs = nbytes; cfg.bufferSize = &bs;
    cfg.doublePrecision = dp; cfg.normalize = normalize;
    if (kind == 1) cfg.performR2C = 1;
    if (kind >= 11 && kind <= 14) cfg.performDCT = kind - 10;
    if (kind >= 21 && kind <= 24) cfg.performDST = kind - 20;
    VkFFTResult r = initializeVkFFT(&app, cfg);
    if (r != VKFFT_SUCCESS) { hipFree(buf); return (int)r; }
    if (uploads_out) for (int i = 0; i < fftdim; i++) uploads_out[i] = app.localFFTPlan->numAxisUploads[i];
    VkFFTLaunchParams lp = {};
    r = VkFFTAppend(&app, inverse ? 1 : -1, &lp);
    hipDeviceSynchronize();
    hipMemcpy(data, buf, nbytes, hipMemcpyDeviceToHost);
    deleteVkFFT(&app); hipFree(buf);
    return (int)r;
}

// zero padding with the reference (performZeropadding / fft_zeropad_left / fft_zeropad_right / frequencyZeroPadding): as ref_transform
int ref_transform_zeropad(int kind, int fftdim, const uint64_t* size, uint64_t batch, int dp, int inverse, const uint64_t* flags,
                          const uint64_t* left, const uint64_t* right, int frequency, void* data, uint64_t nbytes) {
    hipError_t e = hipInit(0); if (e != hipSuccess) return -1;
    hipDevice_t dev; if (hipDeviceGet(&dev, 0) != hipSuccess) return -2;
    hipSetDevice(0);
    void* buf = nullptr; if (hipMalloc(&buf, nbytes) != hipSuccess) return -3;
    hipMemcpy(buf, data, nbytes, hipMemcpyHostToDevice);
    VkFFTConfiguration cfg = {}; VkFFTApplication app = {};
    cfg.FFTdim = fftdim; for (int i = 0; i < fftdim; i++) cfg.size[i] = size[i];
    cfg.numberBatches = batch; cfg.device = &dev; cfg.buffer = &buf; uint64_t bs = nbytes; cfg.bufferSize = &bs;
    cfg.doublePrecision = dp;
    if (kind == 1) cfg.performR2C = 1;
    if (kind >= 11 && kind <= 14) cfg.performDCT = kind - 10;
    if (kind >= 21 && kind <= 24) cfg.performDST = kind - 20;
    for (int i = 0; i < fftdim; i++) { cfg.performZeropadding[i] = flags[i]; cfg.fft_zeropad_left[i] = left[i]; cfg.fft_zeropad_right[i] = right[i]; }
    cfg.frequencyZeroPadding = frequency;
    VkFFTResult r = initializeVkFFT(&app, cfg);
    if (r != VKFFT_SUCCESS) { hipFree(buf); return (int)r; }
    VkFFTLaunchParams lp = {};
    r = VkFFTAppend(&app, inverse ? 1 : -1, &lp);
    hipDeviceSynchronize();
    hipMemcpy(data, buf, nbytes, hipMemcpyDeviceToHost);
    deleteVkFFT(&app); hipFree(buf);
    return (int)r;
}

// convolution with the reference (sample_50/51/52 call sequence): a kernelConvolution plan transforms `kernel` (kernelSystems systems
// per kernel, numKernels kernels), then a performConvolution plan convolves `data` in place.  Host buffers in the library's layouts.
// ms_out (optional): average time of one convolution append over `iters` runs (the data is convolved repeatedly: timing only).
int ref_convolution(int fftdim, const uint64_t* size, int r2c, int dp, uint64_t coordinates, uint64_t matrix, uint64_t numKernels,
                    int symmetric, int conjugate, int crossPower, uint64_t kernelSystems, void* kernel, uint64_t kbytes,
                    void* data, uint64_t dbytes, double* ms_out, int iters) {
    hipError_t e = hipInit(0); if (e != hipSuccess) return -1;
    hipDevice_t dev; if (hipDeviceGet(&dev, 0) != hipSuccess) return -2;
    hipSetDevice(0);
    void *kbuf = nullptr, *dbuf = nullptr;
    if (hipMalloc(&kbuf, kbytes) != hipSuccess || hipMalloc(&dbuf, dbytes) != hipSuccess) return -3;
    hipMemcpy(kbuf, kernel, kbytes, hipMemcpyHostToDevice);
    hipMemcpy(dbuf, data, dbytes, hipMemcpyHostToDevice);
    VkFFTConfiguration cfg = {}; VkFFTApplication appK = {}, appC = {};
    cfg.FFTdim = fftdim; for (int i = 0; i < fftdim; i++) cfg.size[i] = size[i];
    cfg.device = &dev; cfg.doublePrecision = dp; cfg.normalize = 1; cfg.performR2C = r2c;
    cfg.kernelConvolution = 1; cfg.coordinateFeatures = kernelSystems; cfg.numberBatches = numKernels;
    cfg.buffer = &kbuf; uint64_t kb = kbytes; cfg.bufferSize = &kb;
    VkFFTResult r = initializeVkFFT(&appK, cfg);
    if (r != VKFFT_SUCCESS) { hipFree(kbuf); hipFree(dbuf); return (int)r; }
    VkFFTLaunchParams lp = {};
    r = VkFFTAppend(&appK, -1, &lp);
    hipDeviceSynchronize();
    VkFFTConfiguration cc = cfg;
    cc.kernelConvolution = 0; cc.performConvolution = 1; cc.matrixConvolution = matrix; cc.coordinateFeatures = coordinates;
    cc.numberBatches = 1; cc.numberKernels = numKernels; cc.symmetricKernel = symmetric; cc.conjugateConvolution = conjugate;
    cc.crossPowerSpectrumNormalization = crossPower;
    cc.kernel = &kbuf; cc.kernelSize = &kb; cc.buffer = &dbuf; uint64_t db = dbytes; cc.bufferSize = &db;
    if (r == VKFFT_SUCCESS) r = initializeVkFFT(&appC, cc);
    if (r == VKFFT_SUCCESS) r = VkFFTAppend(&appC, -1, &lp);
    hipDeviceSynchronize();
    if (r == VKFFT_SUCCESS) hipMemcpy(data, dbuf, dbytes, hipMemcpyDeviceToHost);
    if (r == VKFFT_SUCCESS && ms_out && iters > 0) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters; i++) VkFFTAppend(&appC, -1, &lp);
        hipDeviceSynchronize();
        *ms_out = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / iters;
    }
    deleteVkFFT(&appC); deleteVkFFT(&appK); hipFree(kbuf); hipFree(dbuf);
    return (int)r;
}

// sample-0 protocol on a device buffer of `nbytes` (random data): returns ms per FFT+iFFT pair.
double ref_bench_pair_zeropad_ms(int fftdim, const uint64_t* size, uint64_t batch, int dp, int kind, uint64_t nbytes,
                                 int num_iter, uint64_t* uploads_out, const uint64_t* flags, const uint64_t* left, const uint64_t* right);
double ref_bench_pair_ms(int fftdim, const uint64_t* size, uint64_t batch, int dp, int kind, uint64_t nbytes,
                         int num_iter, uint64_t* uploads_out) {
    return ref_bench_pair_zeropad_ms(fftdim, size, batch, dp, kind, nbytes, num_iter, uploads_out, nullptr, nullptr, nullptr);
}
// the same timing loop with the reference's native zero padding (flags / left / right per axis; nullptr: none)
double ref_bench_pair_zeropad_ms(int fftdim, const uint64_t* size, uint64_t batch, int dp, int kind, uint64_t nbytes,
                                 int num_iter, uint64_t* uploads_out, const uint64_t* flags, const uint64_t* left, const uint64_t* right) {
    hipInit(0); hipDevice_t dev; hipDeviceGet(&dev, 0); hipSetDevice(0);
    void* buf = nullptr; if (hipMalloc(&buf, nbytes) != hipSuccess) return -3.0;
    { // deterministic fill in [-1,1]
        size_t n = nbytes / 4; float* h = (float*)malloc(nbytes); uint32_t s = 12345u;
        for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = (float)((s >> 8) * (2.0 / 16777216.0) - 1.0); }
        hipMemcpy(buf, h, nbytes, hipMemcpyHostToDevice); free(h);
    }
    VkFFTConfiguration cfg = {}; VkFFTApplication app = {};
    cfg.FFTdim = fftdim; for (int i = 0; i < fftdim; i++) cfg.size[i] = size[i];
    cfg.numberBatches = batch; cfg.device = &dev; cfg.buffer = &buf; uint64_t bs = nbytes; cfg.bufferSize = &bs;
    cfg.doublePrecision = dp;
    if (kind == 1) cfg.performR2C = 1;
    if (kind >= 11 && kind <= 14) cfg.performDCT = kind - 10;
    if (kind >= 21 && kind <= 24) cfg.performDST = kind - 20;
    if (flags) for (int i = 0; i < fftdim; i++) { cfg.performZeropadding[i] = flags[i]; cfg.fft_zeropad_left[i] = left[i]; cfg.fft_zeropad_right[i] = right[i]; }
    VkFFTResult r = initializeVkFFT(&app, cfg);
    if (r != VKFFT_SUCCESS) { hipFree(buf); return -(double)r; }
    if (uploads_out) for (int i = 0; i < fftdim; i++) uploads_out[i] = app.localFFTPlan->numAxisUploads[i];
    VkFFTLaunchParams lp = {};
    for (int w = 0; w < 2; w++) { VkFFTAppend(&app, -1, &lp); VkFFTAppend(&app, 1, &lp); }
    hipDeviceSynchronize();
    double best = 1e30, sum = 0; const int runs = 3;
    for (int rr = 0; rr < runs; rr++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < num_iter; i++) { VkFFTAppend(&app, -1, &lp); VkFFTAppend(&app, 1, &lp); }
        hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        double ms = std::chrono::duration<double, std::milli>(t1 - t0).count() / num_iter;
        sum += ms; if (ms < best) best = ms;
    }
    deleteVkFFT(&app); hipFree(buf);
    return sum / runs;
}
} // extern "C"

#ifdef REF_MAIN
// usage: vkfft_ref_bench <kmin> <kmax> [dp]   -> one JSON line per N=2^k, 1 GiB (fp32) buffer
int main(int argc, char** argv) {
    int kmin = argc > 1 ? atoi(argv[1]) : 8, kmax = argc > 2 ? atoi(argv[2]) : 22, dp = argc > 3 ? atoi(argv[3]) : 0;
    uint64_t total_log2 = argc > 4 ? atoi(argv[4]) : 27;
    for (int k = kmin; k <= kmax; k++) {
        uint64_t N = 1ull << k, B = (1ull << total_log2) / N; if (B < 1) B = 1;
        uint64_t nbytes = N * B * (dp ? 16 : 8), up[4] = {0, 0, 0, 0};
        int iters = (int)((3ull * 4096ull * 1024 * 1024) / nbytes); if (iters > 1000) iters = 1000; if (iters < 1) iters = 1;
        double ms = ref_bench_pair_ms(1, &N, B, dp, 0, nbytes, iters, up);
        double gbps = 2.0 * 2.0 * nbytes / (ms * 1e-3) / 1e9;  // algorithmic bytes: (read+write) x (fwd+inv)
        double gflops = 2.0 * 5.0 * N * k * B / (ms * 1e-3) / 1e9;
        printf("{\"impl\":\"vkfft_ref_hip\",\"log2N\":%d,\"N\":%llu,\"batch\":%llu,\"dp\":%d,\"uploads\":%llu,\"pair_ms\":%.5f,\"alg_GBps\":%.1f,\"GFLOPs\":%.1f}\n",
               k, (unsigned long long)N, (unsigned long long)B, dp, (unsigned long long)up[0], ms, gbps, gflops);
        fflush(stdout);
    }
    return 0;
}
#endif
