// Multi-GPU host drivers in C++ above the C-ABI: new functionality — the reference is single-device (README.md:27-29).
// One host thread per GPU inside ONE process; every thread makes its device current, plans with `VkFFTConfiguration::device` = that device and
// enqueues through `VkFFTAppend` like any caller of the reference would.
//
//   -batch   batched 1D C2C sharded on the batch axis: every rank owns a contiguous block of the transforms, no collective on the data path
//            (weak scaling: -B is the batch PER GPU unless -total is given); the timed region is bracketed by a barrier over the ranks and the
//            slowest rank's time counts.
//   -slab3d  one 3D C2C of n^3 points distributed as z-slabs: local (x, y) transforms -> ONE exchange that re-partitions z <-> y (every pair of
//            ranks trades about 8 n^3 / g^2 bytes; n need not be a multiple of g: slabs of ceil / floor(n / g) planes and rows) -> local z transforms; the result is left in y-slab layout [nz][ny/g][nx] (no second exchange), the
//            inverse takes that layout back.  Exchange transports:
//              rccl  ncclSend / ncclRecv inside one group call per rank (RCCL over xGMI; needs g distinct devices)
//              copy  device-to-device copies into the peers' receive buffers (hipMemcpyAsync; also what lets g "virtual ranks" share ONE device,
//                    the way the algorithm is checked on a single-GPU box)
//            -verify compares against the same volume transformed on one device by the library itself.
// Output: one JSON line per run.
#include "../include/vkFFT.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static bool has_flag(int argc, char** argv, const char* f) { for (int i = 1; i < argc; i++) if (!strcmp(argv[i], f)) return true; return false; }
static const char* flag_value(int argc, char** argv, const char* f) { for (int i = 1; i + 1 < argc; i++) if (!strcmp(argv[i], f)) return argv[i + 1]; return nullptr; }
static uint64_t flag_u64(int argc, char** argv, const char* f, uint64_t d) { const char* v = flag_value(argc, argv, f); return v ? strtoull(v, nullptr, 10) : d; }

struct Barrier { // reusable barrier over the rank threads
	std::mutex m; std::condition_variable cv; int n, waiting = 0; uint64_t gen = 0;
	explicit Barrier(int n_) : n(n_) {}
	void wait() {
		std::unique_lock<std::mutex> lk(m);
		const uint64_t g = gen;
		if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); }
		else cv.wait(lk, [&] { return gen != g; });
	}
};
#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); fail.store(1); } } while (0)

static void fill(std::vector<float>& h, uint64_t seed) { // uniform [-1, 1], deterministic
	uint64_t s = seed * 0x9E3779B97F4A7C15ull + 12345;
	for (auto& x : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
}

// ---------------------------------------------------------------------------------------------------------------- batch sharding
static int run_batch(int g, const std::vector<int>& devs, uint64_t X, uint64_t Bper, uint64_t pairs, uint64_t warm) {
	Barrier bar(g);
	std::atomic<int> fail{0};
	std::vector<double> ms(g, 0.0);
	auto rank_main = [&](int r) {
		HIPOK(hipSetDevice(devs[r]));
		hipDevice_t dev; HIPOK(hipDeviceGet(&dev, devs[r]));
		hipStream_t st; HIPOK(hipStreamCreate(&st));
		const uint64_t bytes = X * Bper * 8;
		void* buf = nullptr; HIPOK(hipMalloc(&buf, bytes));
		{ std::vector<float> h(1 << 20); fill(h, r + 1); for (uint64_t off = 0; off < bytes; off += h.size() * 4) HIPOK(hipMemcpy((char*)buf + off, h.data(), std::min<uint64_t>(h.size() * 4, bytes - off), hipMemcpyHostToDevice)); }
		VkFFTConfiguration cfg = {}; VkFFTApplication app = {};
		cfg.FFTdim = 1; cfg.size[0] = X; cfg.numberBatches = Bper; cfg.normalize = 1; cfg.device = &dev; cfg.stream = &st; cfg.num_streams = 1;
		uint64_t bs = bytes; cfg.buffer = &buf; cfg.bufferSize = &bs;
		VkFFTResult res = initializeVkFFT(&app, cfg);
		if (res != VKFFT_SUCCESS) { fprintf(stderr, "rank %d: initializeVkFFT %d (%s)\n", r, (int)res, getVkFFTErrorString(res)); fail.store(1); }
		VkFFTLaunchParams lp = {};
		for (uint64_t i = 0; i < warm && !fail.load(); i++) { VkFFTAppend(&app, -1, &lp); VkFFTAppend(&app, 1, &lp); }
		HIPOK(hipStreamSynchronize(st));
		bar.wait();
		const auto t0 = std::chrono::steady_clock::now();
		for (uint64_t i = 0; i < pairs && !fail.load(); i++) { if (VkFFTAppend(&app, -1, &lp) != VKFFT_SUCCESS || VkFFTAppend(&app, 1, &lp) != VKFFT_SUCCESS) fail.store(1); }
		HIPOK(hipStreamSynchronize(st));
		bar.wait(); // the slowest rank closes the timed region
		ms[r] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		if (app.impl) deleteVkFFT(&app);
		hipFree(buf); hipStreamDestroy(st);
	};
	std::vector<std::thread> th;
	for (int r = 0; r < g; r++) th.emplace_back(rank_main, r);
	for (auto& t : th) t.join();
	if (fail.load()) return 1;
	double worst = 0; for (double m : ms) worst = std::max(worst, m);
	const double perPair = worst / (double)pairs, k = std::log2((double)X);
	const double gflops = 2.0 * 5.0 * (double)X * k * (double)Bper * g / (perPair * 1e-3) / 1e9;
	const double gbps = 4.0 * (double)(X * Bper * 8) * g / (perPair * 1e-3) / 1e9;
	printf("{\"driver\": \"batch_sharded_c2c\", \"gpus\": %d, \"N\": %llu, \"batch_per_gpu\": %llu, \"pairs\": %llu, \"pair_ms\": %.4f, \"GFLOPs_all_gpus\": %.1f, \"alg_GBps_all_gpus\": %.1f, \"collectives\": \"none\"}\n",
	       g, (unsigned long long)X, (unsigned long long)Bper, (unsigned long long)pairs, perPair, gflops, gbps);
	return 0;
}

// ---------------------------------------------------------------------------------------------------------------- slab 3D
struct SlabRank {
	int r = 0, dev = 0;
	hipStream_t st = nullptr;
	void *slab = nullptr, *send = nullptr, *recv = nullptr; // z-slab [nzl][ny][nx]; send / receive layouts [peer][nzl][nyl][nx]
	VkFFTApplication xy = {}, z = {};
	hipDevice_t hdev;
	ncclComm_t comm = nullptr;
};

// contiguous block of `total` items owned by rank r of g (the first total % g ranks one more): the split of vkfft_amd/distributed.py shard_range
static void shard(uint64_t total, int r, int g, uint64_t& lo, uint64_t& cnt) {
	const uint64_t base = total / g, extra = total % g;
	lo = (uint64_t)r * base + std::min<uint64_t>((uint64_t)r, extra);
	cnt = base + ((uint64_t)r < extra ? 1 : 0);
}

static int run_slab(int g, const std::vector<int>& devs, uint64_t n, const std::string& transport, bool verify, uint64_t reps) {
	if (n < (uint64_t)g) { fprintf(stderr, "every rank needs at least one plane\n"); return 2; }
	// rank r owns the planes zlo[r] .. zlo[r] + nzr[r] before the exchange and the rows ylo[r] .. + nyr[r] after it; n need not be a multiple of g
	const uint64_t nx = n, ny = n, nz = n;
	std::vector<uint64_t> zlo(g), nzr(g), ylo(g), nyr(g);
	for (int r = 0; r < g; r++) { shard(nz, r, g, zlo[r], nzr[r]); shard(ny, r, g, ylo[r], nyr[r]); }
	const uint64_t maxBlock = nzr[0] * nyr[0] * nx; // largest pair message (complex elements)
	Barrier bar(g);
	std::atomic<int> fail{0};
	std::vector<SlabRank> R(g);
	std::vector<ncclComm_t> comms(g, nullptr);
	const bool rccl = transport == "rccl";
	if (rccl) {
		if (ncclCommInitAll(comms.data(), g, devs.data()) != ncclSuccess) { fprintf(stderr, "ncclCommInitAll failed (rccl needs %d distinct devices; use -transport copy for virtual ranks)\n", g); return 2; }
	}
	std::vector<double> ms(g, 0.0);
	std::vector<float> hostIn, hostOut;
	if (verify) { hostIn.resize(2 * nx * ny * nz); fill(hostIn, 7); hostOut.resize(hostIn.size()); }
	auto rank_main = [&](int r) {
		SlabRank& k = R[r];
		k.r = r; k.dev = devs[r]; k.comm = comms[r];
		const uint64_t nzl = nzr[r], nyl = nyr[r];
		const uint64_t slabElems = nzl * ny * nx;   // my planes, every row:   z-slab [nzl][ny][nx] and the send layout [peer s][nzl][nyr[s]][nx]
		const uint64_t yslabElems = nz * nyl * nx;  // every plane, my rows:   y-slab [nz][nyl][nx] = [peer s][nzr[s]][nyl][nx]
		const uint64_t bufElems = std::max(slabElems, yslabElems);
		HIPOK(hipSetDevice(k.dev));
		HIPOK(hipDeviceGet(&k.hdev, k.dev));
		HIPOK(hipStreamCreate(&k.st));
		HIPOK(hipMalloc(&k.slab, bufElems * 8)); HIPOK(hipMalloc(&k.send, bufElems * 8)); HIPOK(hipMalloc(&k.recv, bufElems * 8));
		if (verify) HIPOK(hipMemcpy(k.slab, hostIn.data() + 2 * zlo[r] * ny * nx, slabElems * 8, hipMemcpyHostToDevice));
		else { std::vector<float> h(1 << 20); fill(h, r + 1); for (uint64_t off = 0; off < slabElems * 8; off += h.size() * 4) HIPOK(hipMemcpy((char*)k.slab + off, h.data(), std::min<uint64_t>(h.size() * 4, slabElems * 8 - off), hipMemcpyHostToDevice)); }
		// local plans: (x, y) of the nzl owned planes in place; z lines of the [nz][nyl][nx] volume (axes 0 and 1 omitted)
		{
			VkFFTConfiguration c = {};
			c.FFTdim = 2; c.size[0] = nx; c.size[1] = ny; c.numberBatches = nzl; c.device = &k.hdev; c.stream = &k.st; c.num_streams = 1;
			uint64_t bs = slabElems * 8; c.buffer = &k.slab; c.bufferSize = &bs;
			if (initializeVkFFT(&k.xy, c) != VKFFT_SUCCESS) fail.store(1);
			VkFFTConfiguration d = {};
			d.FFTdim = 3; d.size[0] = nx; d.size[1] = nyl; d.size[2] = nz; d.omitDimension[0] = 1; d.omitDimension[1] = 1; d.device = &k.hdev; d.stream = &k.st; d.num_streams = 1;
			uint64_t bz = yslabElems * 8; d.buffer = &k.recv; d.bufferSize = &bz;
			if (initializeVkFFT(&k.z, d) != VKFFT_SUCCESS) fail.store(1);
		}
		bar.wait(); // every rank's buffers exist (the copy transport writes into peers' receive buffers)
		// element offsets of the block exchanged with peer s: in the layout of MY planes ([s][nzl][nyr[s]][nx]) and in the layout of MY rows ([s][nzr[s]][nyl][nx])
		auto offPlanes = [&](int s) { return nzl * ylo[s] * nx; };
		auto lenPlanes = [&](int s) { return nzl * nyr[s] * nx; };
		auto offRows = [&](int s) { return zlo[s] * nyl * nx; };
		auto lenRows = [&](int s) { return nzr[s] * nyl * nx; };
		// forward: the blocks of my planes go out, the blocks of my rows come in; backward: the other way round
		auto exchange = [&](void* from, void* SlabRank::*to, bool fwd) {
			if (rccl) {
				ncclGroupStart();
				for (int s = 0; s < g; s++) {
					ncclSend((const char*)from + (fwd ? offPlanes(s) : offRows(s)) * 8, (fwd ? lenPlanes(s) : lenRows(s)) * 2, ncclFloat, s, k.comm, k.st);
					ncclRecv((char*)(k.*to) + (fwd ? offRows(s) : offPlanes(s)) * 8, (fwd ? lenRows(s) : lenPlanes(s)) * 2, ncclFloat, s, k.comm, k.st);
				}
				if (ncclGroupEnd() != ncclSuccess) fail.store(1);
			} else {
				HIPOK(hipStreamSynchronize(k.st)); bar.wait(); // everybody's send layout is complete
				for (int s = 0; s < g; s++) {
					// my block for peer s lands in peer s's buffer where ITS layout expects the block of peer r (= me)
					const uint64_t dstOff = fwd ? zlo[r] * nyr[s] * nx : nzr[s] * ylo[r] * nx;
					HIPOK(hipMemcpyAsync((char*)(R[s].*to) + dstOff * 8, (const char*)from + (fwd ? offPlanes(s) : offRows(s)) * 8, (fwd ? lenPlanes(s) : lenRows(s)) * 8, hipMemcpyDeviceToDevice, k.st));
				}
				HIPOK(hipStreamSynchronize(k.st)); bar.wait(); // everything has arrived
			}
		};
		VkFFTLaunchParams lp = {};
		auto forward = [&]() {
			if (VkFFTAppend(&k.xy, -1, &lp) != VKFFT_SUCCESS) fail.store(1);
			// pack: [z][y][x] -> [peer][z][y in block][x]: per peer one strided copy (rows of nyr[s]*nx complex, pitch ny*nx)
			for (int s = 0; s < g; s++)
				HIPOK(hipMemcpy2DAsync((char*)k.send + offPlanes(s) * 8, nyr[s] * nx * 8, (const char*)k.slab + ylo[s] * nx * 8, ny * nx * 8, nyr[s] * nx * 8, nzl, hipMemcpyDeviceToDevice, k.st));
			exchange(k.send, &SlabRank::recv, true); // recv = [s][nzr[s]][nyl][nx] = [nz][nyl][nx]
			if (VkFFTAppend(&k.z, -1, &lp) != VKFFT_SUCCESS) fail.store(1);
		};
		auto inverse = [&]() {
			if (VkFFTAppend(&k.z, 1, &lp) != VKFFT_SUCCESS) fail.store(1);
			exchange(k.recv, &SlabRank::send, false); // z-block s of the y-slab goes back to rank s: send = [peer][nzl][nyr[peer]][nx]
			for (int s = 0; s < g; s++)
				HIPOK(hipMemcpy2DAsync((char*)k.slab + ylo[s] * nx * 8, ny * nx * 8, (const char*)k.send + offPlanes(s) * 8, nyr[s] * nx * 8, nyr[s] * nx * 8, nzl, hipMemcpyDeviceToDevice, k.st));
			if (VkFFTAppend(&k.xy, 1, &lp) != VKFFT_SUCCESS) fail.store(1);
		};
		if (!fail.load()) forward();
		HIPOK(hipStreamSynchronize(k.st));
		if (verify) HIPOK(hipMemcpy(hostOut.data() + 2 * nz * ylo[r] * nx, k.recv, yslabElems * 8, hipMemcpyDeviceToHost)); // y-slab r: [nz][nyl][nx], slabs one after the other
		if (!fail.load()) inverse();
		HIPOK(hipStreamSynchronize(k.st));
		bar.wait();
		const auto t0 = std::chrono::steady_clock::now();
		for (uint64_t i = 0; i < reps && !fail.load(); i++) { forward(); inverse(); }
		HIPOK(hipStreamSynchronize(k.st));
		bar.wait();
		ms[r] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (double)std::max<uint64_t>(reps, 1);
	};
	std::vector<std::thread> th;
	for (int r = 0; r < g; r++) th.emplace_back(rank_main, r);
	for (auto& t : th) t.join();
	double err = -1;
	if (verify && !fail.load()) {
		// the same volume on one device through the library itself (3D plan), compared slab by slab in the y-slab layout
		hipSetDevice(devs[0]);
		hipDevice_t d0; hipDeviceGet(&d0, devs[0]);
		void* vol = nullptr; const uint64_t bytes = nx * ny * nz * 8;
		if (hipMalloc(&vol, bytes) == hipSuccess) {
			hipMemcpy(vol, hostIn.data(), bytes, hipMemcpyHostToDevice);
			VkFFTConfiguration c = {}; VkFFTApplication a = {};
			c.FFTdim = 3; c.size[0] = nx; c.size[1] = ny; c.size[2] = nz; c.device = &d0; uint64_t bs = bytes; c.buffer = &vol; c.bufferSize = &bs;
			VkFFTLaunchParams lp = {};
			if (initializeVkFFT(&a, c) == VKFFT_SUCCESS && VkFFTAppend(&a, -1, &lp) == VKFFT_SUCCESS) {
				hipDeviceSynchronize();
				std::vector<float> ref(hostIn.size());
				hipMemcpy(ref.data(), vol, bytes, hipMemcpyDeviceToHost);
				double num = 0, den = 0;
				for (int r = 0; r < g; r++) for (uint64_t zz = 0; zz < nz; zz++) for (uint64_t yy = 0; yy < nyr[r]; yy++) for (uint64_t xx = 0; xx < 2 * nx; xx++) {
					const double a1 = hostOut[2 * nz * ylo[r] * nx + (zz * nyr[r] + yy) * 2 * nx + xx];
					const double b1 = ref[(zz * ny + ylo[r] + yy) * 2 * nx + xx];
					num += (a1 - b1) * (a1 - b1); den += b1 * b1;
				}
				err = std::sqrt(num / den);
				deleteVkFFT(&a);
			}
			hipFree(vol);
		}
	}
	for (auto& k : R) { if (k.xy.impl) { hipSetDevice(k.dev); deleteVkFFT(&k.xy); } if (k.z.impl) deleteVkFFT(&k.z); hipFree(k.slab); hipFree(k.send); hipFree(k.recv); if (k.st) hipStreamDestroy(k.st); }
	if (rccl) for (auto c : comms) if (c) ncclCommDestroy(c);
	if (fail.load()) return 1;
	double worst = 0; for (double m : ms) worst = std::max(worst, m);
	const double pts = (double)nx * ny * nz, gflops = 2.0 * 5.0 * pts * std::log2(pts) / (worst * 1e-3) / 1e9;
	char errs[32]; if (err < 0) snprintf(errs, sizeof(errs), "null"); else snprintf(errs, sizeof(errs), "%.3e", err);
	printf("{\"driver\": \"slab_3d_c2c\", \"ranks\": %d, \"n\": %llu, \"transport\": \"%s\", \"pair_message_MiB\": %.2f, \"fwd_inv_ms\": %.4f, \"GFLOPs\": %.1f, \"rel_l2_vs_single_device_plan\": %s}\n",
	       g, (unsigned long long)n, transport.c_str(), maxBlock * 8 / 1048576.0, worst, gflops, errs);
	return (verify && !(err >= 0 && err < 2e-6)) ? 1 : 0;
}

int main(int argc, char** argv) {
	if (argc < 2 || has_flag(argc, argv, "-h")) {
		printf("vkfft_mi355x_multi — multi-GPU drivers above the C-ABI (one host thread per GPU)\n"
		       "  -batch  -X <N> -B <batch per GPU> [-g <gpus>] [-pairs <n>]      batched 1D C2C fp32 sharded on the batch axis (no collective)\n"
		       "  -slab3d -n <edge> [-g <ranks>] [-transport rccl|copy] [-verify] [-reps <n>]   slab-decomposed 3D C2C with one exchange\n"
		       "  -virtual                                                       put all ranks on device 0 (-transport copy only)\n");
		return 0;
	}
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fprintf(stderr, "no HIP device\n"); return 2; }
	int g = (int)flag_u64(argc, argv, "-g", (uint64_t)ndev);
	const bool virt = has_flag(argc, argv, "-virtual");
	if (!virt && g > ndev) { fprintf(stderr, "%d GPUs requested, %d visible: using %d\n", g, ndev, ndev); g = ndev; }
	std::vector<int> devs(g);
	for (int r = 0; r < g; r++) devs[r] = virt ? 0 : r;
	if (has_flag(argc, argv, "-batch")) {
		const uint64_t X = flag_u64(argc, argv, "-X", 1u << 20), B = flag_u64(argc, argv, "-B", (1ull << 27) / X);
		return run_batch(g, devs, X, B, flag_u64(argc, argv, "-pairs", 20), 3);
	}
	if (has_flag(argc, argv, "-slab3d")) {
		const char* tr = flag_value(argc, argv, "-transport");
		std::string transport = tr ? tr : (virt || g == 1 ? "copy" : "rccl");
		if (virt && transport == "rccl") { fprintf(stderr, "virtual ranks share one device: -transport copy\n"); return 2; }
		return run_slab(g, devs, flag_u64(argc, argv, "-n", 256), transport, has_flag(argc, argv, "-verify"), flag_u64(argc, argv, "-reps", 5));
	}
	fprintf(stderr, "nothing to do (-batch or -slab3d)\n");
	return 2;
}
