// Command-line benchmark / self-check driver of libvkfft_mi355x through its public C API (include/vkFFT.h) — the role the
// reference's VkFFT_TestSuite binary plays (VkFFT_TestSuite.cpp:557-1058, benchmark_scripts/.../user_benchmark_VkFFT.cpp:100-135):
// the same flag names (-vkfft <id>, -benchmark_vkfft -X -Y -Z -P -B -N -R2C -DCT, -d, -o, -devices) and the same timing
// protocol (plan once, num_iter FFT+iFFT pairs, one synchronisation, repeated 3 times), so that side-by-side runs with the
// reference binary are trivial.  A caller-side program: it links the C-ABI library like any user of the reference would.
#include "../include/vkFFT.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static FILE* g_out = nullptr;
static void say(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt); vprintf(fmt, ap); va_end(ap);
	if (g_out) { va_start(ap, fmt); vfprintf(g_out, fmt, ap); va_end(ap); }
	fflush(stdout);
}
static bool has_flag(int argc, char** argv, const char* f) { for (int i = 1; i < argc; i++) if (!strcmp(argv[i], f)) return true; return false; }
static const char* flag_value(int argc, char** argv, const char* f) { for (int i = 1; i + 1 < argc; i++) if (!strcmp(argv[i], f)) return argv[i + 1]; return nullptr; }
static uint64_t flag_u64(int argc, char** argv, const char* f, uint64_t dflt) { const char* v = flag_value(argc, argv, f); return v ? strtoull(v, nullptr, 10) : dflt; }

struct Problem {
	uint64_t X = 1, Y = 1, Z = 1, B = 1;
	int P = 0;      // 0 fp32, 1 fp64
	int R2C = 0, DCT = 0, DST = 0;
	uint64_t N = 0; // pairs per timed run (0: the reference's rule min(1000, 3*4096 MiB / buffer))
};
struct Timing { double ms = 0, stderr_ms = 0; uint64_t iters = 0; uint64_t bytes = 0; uint64_t uploads[3] = {0, 0, 0}; VkFFTResult res = VKFFT_SUCCESS; };

static uint64_t buffer_bytes(const Problem& q) {
	const uint64_t real = q.P ? 8 : 4;
	if (q.R2C) return (q.X / 2 + 1) * 2 * real * q.Y * q.Z * q.B; // in-place padded rows
	if (q.DCT || q.DST) return q.X * q.Y * q.Z * q.B * real;
	return q.X * q.Y * q.Z * q.B * 2 * real;
}

// plan once, run `iters` FFT+iFFT pairs, one synchronisation per run, 3 runs (sample_0_benchmark_VkFFT_single.cpp:202-233)
static Timing time_problem(const Problem& q, hipDevice_t* dev, bool fillRandom) {
	Timing t;
	t.bytes = buffer_bytes(q);
	void* buf = nullptr;
	if (hipMalloc(&buf, t.bytes) != hipSuccess) { t.res = VKFFT_ERROR_FAILED_TO_ALLOCATE; return t; }
	if (fillRandom) { // uniform [-1, 1] like the reference's host fill, generated once on the host in modest blocks
		const size_t block = 1 << 22;
		std::vector<float> h(block);
		std::vector<double> hd(q.P ? block : 0);
		uint64_t state = 0x9E3779B97F4A7C15ull;
		for (uint64_t off = 0; off < t.bytes;) {
			const size_t elems = (size_t)std::min<uint64_t>(block, (t.bytes - off) / (q.P ? 8 : 4));
			for (size_t i = 0; i < elems; i++) {
				state = state * 6364136223846793005ull + 1442695040888963407ull;
				const double u = (double)(state >> 11) / 9007199254740992.0 * 2.0 - 1.0;
				if (q.P) hd[i] = u; else h[i] = (float)u;
			}
			const size_t nb = elems * (q.P ? 8 : 4);
			hipMemcpy((char*)buf + off, q.P ? (void*)hd.data() : (void*)h.data(), nb, hipMemcpyHostToDevice);
			off += nb;
			if (!elems) break;
		}
	} else hipMemset(buf, 0, t.bytes);
	VkFFTConfiguration cfg = {};
	VkFFTApplication app = {};
	cfg.FFTdim = 1 + (q.Y > 1 || q.Z > 1) + (q.Z > 1);
	cfg.size[0] = q.X; cfg.size[1] = q.Y; cfg.size[2] = q.Z;
	cfg.numberBatches = q.B;
	cfg.doublePrecision = q.P == 1;
	cfg.performR2C = q.R2C; cfg.performDCT = q.DCT; cfg.performDST = q.DST;
	cfg.normalize = 1; // keeps the data bounded over many pairs (the reference's unnormalised loop overflows to inf)
	cfg.device = dev;
	uint64_t bs = t.bytes;
	cfg.buffer = &buf; cfg.bufferSize = &bs;
	t.res = initializeVkFFT(&app, cfg);
	if (t.res != VKFFT_SUCCESS) { hipFree(buf); return t; }
	for (int i = 0; i < 3; i++) t.uploads[i] = app.localFFTPlan ? app.localFFTPlan->numAxisUploads[i] : 0;
	t.iters = q.N ? q.N : std::min<uint64_t>(1000, std::max<uint64_t>(1, 3ull * 4096 * 1024 * 1024 / t.bytes));
	VkFFTLaunchParams lp = {};
	for (int i = 0; i < 2 && t.res == VKFFT_SUCCESS; i++) { t.res = VkFFTAppend(&app, -1, &lp); if (t.res == VKFFT_SUCCESS) t.res = VkFFTAppend(&app, 1, &lp); }
	hipDeviceSynchronize();
	double runs[3] = {0, 0, 0};
	for (int r = 0; r < 3 && t.res == VKFFT_SUCCESS; r++) {
		const auto t0 = std::chrono::steady_clock::now();
		for (uint64_t i = 0; i < t.iters && t.res == VKFFT_SUCCESS; i++) { t.res = VkFFTAppend(&app, -1, &lp); if (t.res == VKFFT_SUCCESS) t.res = VkFFTAppend(&app, 1, &lp); }
		if (hipDeviceSynchronize() != hipSuccess) t.res = VKFFT_ERROR_FAILED_TO_SYNCHRONIZE;
		runs[r] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (double)t.iters;
	}
	t.ms = (runs[0] + runs[1] + runs[2]) / 3.0;
	double var = 0; for (double x : runs) var += (x - t.ms) * (x - t.ms);
	t.stderr_ms = std::sqrt(var / 3.0);
	deleteVkFFT(&app);
	hipFree(buf);
	return t;
}

static void report(const char* tag, const Problem& q, const Timing& t) {
	if (t.res != VKFFT_SUCCESS) { say("%s System: %llux%llux%llu batch %llu: error %d (%s)\n", tag, (unsigned long long)q.X, (unsigned long long)q.Y, (unsigned long long)q.Z, (unsigned long long)q.B, (int)t.res, getVkFFTErrorString(t.res)); return; }
	const double mib = (double)t.bytes / (1024.0 * 1024.0);
	// "bandwidth": bytes actually moved = one read + one write per upload of every axis, both directions (the reference's
	// sample_0 column); "scaled bandwidth": one read + one write per direction (sample_1000's column = the algorithmic figure)
	double passes = 0; for (int i = 0; i < 3; i++) passes += (double)t.uploads[i];
	const double gb = (double)t.bytes / 1e9;
	say("%s System: %llux%llux%llu Batch: %llu Precision: %s%s Buffer: %.0f MB avg_time_per_step: %.4f ms std_error: %.4f num_iter: %llu uploads: %llu %llu %llu bandwidth: %.1f GB/s scaled_bandwidth: %.1f GB/s\n",
	    tag, (unsigned long long)q.X, (unsigned long long)q.Y, (unsigned long long)q.Z, (unsigned long long)q.B, q.P ? "double" : "single",
	    q.R2C ? " R2C" : q.DCT ? " DCT" : q.DST ? " DST" : "", mib, t.ms, t.stderr_ms, (unsigned long long)t.iters,
	    (unsigned long long)t.uploads[0], (unsigned long long)t.uploads[1], (unsigned long long)t.uploads[2],
	    4.0 * passes * gb / (t.ms * 1e-3), 4.0 * gb / (t.ms * 1e-3));
}

// round trip on the device with host-side invariants only (no host FFT: the library has no CPU path and neither has this tool)
static int self_check(hipDevice_t* dev) {
	const uint64_t N = 4096, B = 8;
	std::vector<float> h(2 * N * B), back(2 * N * B);
	uint64_t state = 12345;
	for (auto& x : h) { state = state * 6364136223846793005ull + 1442695040888963407ull; x = (float)((double)(state >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
	void* buf = nullptr; uint64_t bs = h.size() * sizeof(float);
	if (hipMalloc(&buf, bs) != hipSuccess) return 1;
	hipMemcpy(buf, h.data(), bs, hipMemcpyHostToDevice);
	VkFFTConfiguration cfg = {}; VkFFTApplication app = {};
	cfg.FFTdim = 1; cfg.size[0] = N; cfg.numberBatches = B; cfg.device = dev; cfg.buffer = &buf; cfg.bufferSize = &bs;
	VkFFTResult r = initializeVkFFT(&app, cfg);
	VkFFTLaunchParams lp = {};
	if (r == VKFFT_SUCCESS) r = VkFFTAppend(&app, -1, &lp);
	std::vector<float> spec(h.size());
	hipDeviceSynchronize(); hipMemcpy(spec.data(), buf, bs, hipMemcpyDeviceToHost);
	if (r == VKFFT_SUCCESS) r = VkFFTAppend(&app, 1, &lp);
	hipDeviceSynchronize(); hipMemcpy(back.data(), buf, bs, hipMemcpyDeviceToHost);
	deleteVkFFT(&app); hipFree(buf);
	if (r != VKFFT_SUCCESS) { say("self-check: error %d (%s)\n", (int)r, getVkFFTErrorString(r)); return 1; }
	double ein = 0, eout = 0, err = 0, dc = 0;
	for (size_t i = 0; i < h.size(); i++) { ein += (double)h[i] * h[i]; eout += (double)spec[i] * spec[i]; const double d = back[i] / (double)N - h[i]; err += d * d; }
	for (uint64_t n = 0; n < N; n++) dc += h[2 * n];
	const double parseval = std::fabs(eout / (N * ein) - 1.0), rt = std::sqrt(err / ein), dcerr = std::fabs(spec[0] - dc) / std::sqrt((double)N);
	say("self-check N=4096 batch 8: round-trip rel-L2 %.3e, Parseval %.3e, DC bin %.3e -> %s\n", rt, parseval, dcerr, (rt < 2e-6 && parseval < 1e-5 && dcerr < 1e-4) ? "PASS" : "FAIL");
	return (rt < 2e-6 && parseval < 1e-5 && dcerr < 1e-4) ? 0 : 1;
}

static void usage() {
	printf("vkfft_mi355x_cli — benchmark driver of libvkfft_mi355x (VkFFT API version %d)\n"
	       "  -h                     this text\n"
	       "  -devices               list HIP devices\n"
	       "  -d <id>                device index (default 0)\n"
	       "  -o <file>              also write the report lines to <file>\n"
	       "  -vkfft <id>            benchmark sweep: 0 = batched 1D C2C fp32 N=2^8..2^22 (1 GiB), 1 = same in fp64 (2^8..2^21),\n"
	       "                         3 = 3D C2C cubes 32^3..512^3, 6 = 2D R2C/C2R squares 256..4096, 14 = non-power-of-two 1D C2C,\n"
	       "                         100 = 2D DCT-II squares 256..2048, 1000 = sweep 0 reporting the scaled (algorithmic) bandwidth only\n"
	       "  -benchmark_vkfft       one user-defined system: -X <n> [-Y <n>] [-Z <n>] [-B batch] [-P 0|1] [-N pairs] [-R2C 1] [-DCT 1..4] [-DST 1..4]\n"
	       "  -test                  device round trip with host-side invariants (Parseval, DC bin, forward+inverse = N x)\n",
	       VkFFTGetVersion());
}

int main(int argc, char** argv) {
	if (argc < 2 || has_flag(argc, argv, "-h")) { usage(); return 0; }
	if (const char* o = flag_value(argc, argv, "-o")) g_out = fopen(o, "w");
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { say("no HIP device: this library has no CPU path\n"); return 2; }
	if (has_flag(argc, argv, "-devices")) {
		for (int i = 0; i < ndev; i++) { hipDeviceProp_t pr; hipGetDeviceProperties(&pr, i); say("Device id: %d name: %s CUs: %d memory: %.1f GiB\n", i, pr.name, pr.multiProcessorCount, (double)pr.totalGlobalMem / (1 << 30)); }
		return 0;
	}
	const int id = (int)flag_u64(argc, argv, "-d", 0);
	hipDevice_t dev;
	if (hipSetDevice(id) != hipSuccess || hipDeviceGet(&dev, id) != hipSuccess) { say("invalid device %d\n", id); return 2; }
	int rc = 0;
	if (has_flag(argc, argv, "-test")) rc |= self_check(&dev);
	if (has_flag(argc, argv, "-benchmark_vkfft")) {
		Problem q;
		q.X = flag_u64(argc, argv, "-X", 0); q.Y = flag_u64(argc, argv, "-Y", 1); q.Z = flag_u64(argc, argv, "-Z", 1); q.B = flag_u64(argc, argv, "-B", 1);
		q.P = (int)flag_u64(argc, argv, "-P", 0); q.N = flag_u64(argc, argv, "-N", 0);
		q.R2C = (int)flag_u64(argc, argv, "-R2C", 0); q.DCT = (int)flag_u64(argc, argv, "-DCT", 0); q.DST = (int)flag_u64(argc, argv, "-DST", 0);
		if (!q.X) { say("-benchmark_vkfft needs -X\n"); return 2; }
		const Timing t = time_problem(q, &dev, true);
		report("VkFFT", q, t);
		rc |= t.res != VKFFT_SUCCESS;
	}
	if (const char* v = flag_value(argc, argv, "-vkfft")) {
		const int sweep = atoi(v);
		std::vector<Problem> list;
		const uint64_t gib = 1ull << 30;
		if (sweep == 0 || sweep == 1 || sweep == 1000) for (int k = 8; k <= (sweep == 1 ? 21 : 22); k++) { Problem q; q.X = 1ull << k; q.P = sweep == 1; q.B = gib / ((sweep == 1 ? 16 : 8) * q.X); list.push_back(q); }
		else if (sweep == 3) for (uint64_t n = 32; n <= 512; n *= 2) { Problem q; q.X = q.Y = q.Z = n; q.B = std::max<uint64_t>(1, gib / (8 * n * n * n)); list.push_back(q); }
		else if (sweep == 6) for (uint64_t n = 256; n <= 4096; n *= 2) { Problem q; q.X = q.Y = n; q.R2C = 1; q.B = std::max<uint64_t>(1, (gib / 4) / (4 * (n + 2) * n)); list.push_back(q); }
		else if (sweep == 14) for (uint64_t n : {1080ull, 2160ull, 3840ull, 4000ull, 7680ull, 2187ull, 3125ull, 2401ull, 1331ull, 2197ull, 127ull, 257ull, 1009ull, 4093ull, 59049ull, 390625ull, 15319ull}) { Problem q; q.X = n; q.B = std::max<uint64_t>(1, (gib / 2) / (8 * n)); list.push_back(q); }
		else if (sweep == 100) for (uint64_t n = 256; n <= 2048; n *= 2) { Problem q; q.X = q.Y = n; q.DCT = 2; q.B = std::max<uint64_t>(1, (gib / 4) / (4 * n * n)); list.push_back(q); }
		else { say("unknown sweep %d\n", sweep); return 2; }
		double score = 0;
		for (const Problem& q : list) {
			const Timing t = time_problem(q, &dev, true);
			report(sweep == 1000 ? "VkFFT(scaled)" : "VkFFT", q, t);
			rc |= t.res != VKFFT_SUCCESS;
			if (t.res == VKFFT_SUCCESS) score += (double)t.bytes / (1024.0 * 1024.0) / t.ms;
		}
		say("Benchmark score (sum of buffer MiB per ms of an FFT+iFFT pair): %.0f\n", score);
	}
	if (g_out) fclose(g_out);
	return rc;
}
