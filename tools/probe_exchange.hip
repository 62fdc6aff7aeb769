// GPU probe (tools/, not product): is the inter-stage exchange of the small power-of-two row kernels on the critical path?
// Times the product kernel pow2_row_kernel (N = 2^8, 2^9, 2^10, 1 GiB, forward only) as built, and the same kernel with the exchange
// compiled out (-DVKFFT_PROBE_NO_EXCHANGE: registers pass straight through, results wrong, timing only).  If the two agree, no other
// exchange mechanism (DPP / ds_bpermute / __shfl instead of the wave-synchronous LDS exchange) can make the kernel faster.
#include "kernel_pow2.h"
#include <cstdio>
#include <functional>
#include <cstring>
using namespace vkfft_mi355x;
static float timeit(int iters, const std::function<void()>& f) {
	hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); f(); f(); (void)hipDeviceSynchronize();
	(void)hipEventRecord(a, 0); for (int i = 0; i < iters; i++) f(); (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
	float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / iters;
}
template <typename SCH, int FPW> void run(void* buf, void* lut, const char* tag) {
	constexpr int N = 1 << SCH::LOGN;
	PassParams p; std::memset((void*)&p, 0, sizeof(p));
	const uint32_t B = (1u << 27) / N;
	p.in = buf; p.out = buf; p.lut = lut; p.scale = 1.0;
	p.dim[0] = {B, (int64_t)N, (int64_t)N}; p.dim[1] = {1, 0, 0}; p.dim[2] = {1, 0, 0};
	p.tilesPerG0 = (B + FPW - 1) / FPW;
	const float ms = timeit(20, [&] { hipLaunchKernelGGL((pow2_row_kernel<float, SCH, FPW>), dim3(p.tilesPerG0), dim3((N >> SCH::LOGE) * FPW), 0, 0, p); });
	printf("{\"probe\":\"row_kernel_%s\",\"log2N\":%d,\"ms\":%.4f,\"alg_GBps\":%.1f}\n", tag, SCH::LOGN, ms, 2.0 * (8ull << 27) / ms / 1e6);
}
int main() {
	void *buf, *lut; (void)hipMalloc(&buf, 8ull << 27); (void)hipMalloc(&lut, 1 << 20);
	(void)hipMemset(buf, 0, 8ull << 27); (void)hipMemset(lut, 0, 1 << 20);
#if defined(VKFFT_PROBE_NO_EXCHANGE)
	const char* tag = "no_exchange";
#else
	const char* tag = "as_built";
#endif
	run<Pow2Sched<4, 4, 0, 0>, 8>(buf, lut, tag);
	run<Pow2Sched<5, 4, 0, 0>, 8>(buf, lut, tag);
	run<Pow2Sched<5, 5, 0, 0>, 8>(buf, lut, tag);
	return 0;
}
