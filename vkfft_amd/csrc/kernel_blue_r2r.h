// DCT/DST whose embedding FFT length has a prime factor the radix / Rader stages do not cover (e.g. DST-I of N = 100:
// 2N + 2 = 202 = 2 * 101, DCT-I of N = 240: 478 = 2 * 239): the real transform's pre/post maps (vkFFT_R2R.h) around a fused
// Bluestein transform of the embedding length (vkFFT_Bluestein.h:32,201), in one kernel.  Same register-resident persistent
// structure as pow2_blue_kernel (kernel_pow2.h): the embedding sequence of length blueN <= M/2 is gathered with pre_gather<PRE>,
// chirp-multiplied, zero-padded to M = 2^k, transformed, multiplied by FFT(chirp)/M, transformed back, chirp-multiplied, and its
// outputs leave through post_scatter<POST>.  Families: DCT-I, DST-I, DCT/DST-II and -III in their full-length forms, DCT/DST-IV,
// and R2C / C2R in their full-length ("callback") forms for real rows whose half length has such a prime factor.
#pragma once
#include "kernel_generic.h"
#include "kernel_pow2_core.h"
#include "kernel_opfft.h"

namespace vkfft_mi355x {

template <typename T, typename SCH, int FPW, int PRE, int POST>
__global__ void __launch_bounds__(((1 << SCH::LOGN) >> SCH::LOGE) * FPW) pow2_blue_r2r_kernel(const PassParams p) {
	constexpr int LOGN = SCH::LOGN, M = 1 << LOGN, LOGE = SCH::LOGE, E = 1 << LOGE, TPF = M / E, EH = E / 2;
	constexpr int LDSPF = SCH::NS > 1 ? M + (M >> LOGE) : 1;
	constexpr bool waveOnly = TPF <= 64;
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	static_assert(SCH::NS > 1, "the paired form turns the spectrum through the exchange buffer");
	__shared__ cx<T> lds[FPW * LDSPF];
	const uint32_t tid = threadIdx.x;
	const uint32_t f = tid / TPF, tau = tid % TPF;
	const uint32_t n = p.blueN; // embedding length (2N-2, 2N+2 or N); p.opN stays the real transform's N for the maps
	// aux3 = chirp[0..n) followed by FFT(chirp)/M [0..M): aux and aux2 belong to the real transform's own maps
	const GBuf glut = make_gbuf(p.lut), gch = make_gbuf(p.aux3), gbh = make_gbuf((const cx<T>*)p.aux3 + n);
	// the chirp and the kernel spectrum of a thread's points are the same for every tile: kept in registers by the persistent workgroups — except in the two
	// largest shapes (8192 / 16384 points: 512 / 1024 threads at 256 / 128 registers), where they were what spilled (up to 1 248 bytes of scratch per lane, round 5):
	// there they are read again for every tile (L2 hits)
	constexpr bool TABREG = TPF * FPW < 512;
	cx<T> ch[EH], bh[E];
	if constexpr (TABREG) {
#pragma unroll
		for (int m = 0; m < EH; m++) { const uint32_t pos = tau + m * TPF; ch[m] = gb_load<T>(gch, pos < n ? pos * ES : kGbInvalid, 0); }
#pragma unroll
		for (int m = 0; m < E; m++) bh[m] = gb_load<T>(gbh, (tau + m * TPF) * ES, 0);
	}
	auto chv = [&](int m) -> cx<T> { if constexpr (TABREG) return ch[m]; else { const uint32_t pos = tau + m * TPF; return gb_load<T>(gch, pos < n ? pos * ES : kGbInvalid, 0); } };
	auto bhv = [&](int m) -> cx<T> { if constexpr (TABREG) return bh[m]; else return gb_load<T>(gbh, (tau + m * TPF) * ES, 0); };
	const uint32_t tiles = p.tilesPerG0 * p.dim[1].count * p.dim[2].count;
	for (uint32_t wgi = blockIdx.x; wgi < tiles; wgi += gridDim.x) {
		uint32_t wg = p.reverseTiles ? tiles - 1u - wgi : wgi;
		const uint32_t tile = wg % p.tilesPerG0;
		wg /= p.tilesPerG0;
		const uint32_t g1 = wg % p.dim[1].count, g2 = wg / p.dim[1].count;
		// two real rows per transform (PassParams::pairRows; the reference's mergeSequencesR2C, vkFFT_SharedMemory.h:40, vkFFT_R2C.h:178,450): thread group f owns rows
		// 2f and 2f + 1 of the tile.  Families whose pre-map is a real sequence: z = a + i b, the two spectra come back through the even / odd split, which needs
		// Y[k] and Y[n - k] together — one more trip through the exchange buffer; families whose result is real (C2R, DCT / DST-III): real and imaginary part.
		const bool pair = p.pairRows != 0;
		const uint32_t mult = pair ? 2u : 1u;
		const uint32_t f0 = tile * FPW * mult, g0 = f0 + f * mult;
		const bool valid = g0 < p.dim[0].count, validB = pair && g0 + 1u < p.dim[0].count;
		const int64_t inBase = (int64_t)g1 * p.dim[1].inStride + (int64_t)g2 * p.dim[2].inStride + (int64_t)f0 * p.dim[0].inStride;
		const int64_t outBase = (int64_t)g1 * p.dim[1].outStride + (int64_t)g2 * p.dim[2].outStride + (int64_t)f0 * p.dim[0].outStride;
		Io32<T> io;
		io.gin = make_gbuf((const char*)p.in + inBase * (int64_t)p.inElemBytes);
		io.gout = make_gbuf((char*)p.out + outBase * (int64_t)p.outElemBytes);
		io.inOff = valid ? f * mult * (uint32_t)p.dim[0].inStride * p.inElemBytes : kGbInvalid;
		io.outOff = valid ? f * mult * (uint32_t)p.dim[0].outStride * p.outElemBytes : kGbInvalid;
		io.inSj = (uint32_t)p.inStrideJ * p.inElemBytes;
		io.outSj = (uint32_t)p.outStrideJ * p.outElemBytes;
		io.set_pad(p);
		Io32<T> ioB = io;
		ioB.inOff = validB ? (f * mult + 1u) * (uint32_t)p.dim[0].inStride * p.inElemBytes : kGbInvalid;
		ioB.outOff = validB ? (f * mult + 1u) * (uint32_t)p.dim[0].outStride * p.outElemBytes : kGbInvalid;
		const uint32_t nat = g0 * p.opStride0 + g1 * p.opStride1;
		const bool realResult = op_pair_result_is_real(p.postOp);
		cx<T> v[E];
#pragma unroll
		for (int m = 0; m < EH; m++) { // embedding points >= n are the zero padding (n <= M/2)
			const uint32_t pos = tau + m * TPF;
			cx<T> x = {(T)0, (T)0};
			if (pos < n) {
				x = pre_gather<T>(p, io, pos, nat, op_resolve<PRE>(p.preOp));
				if (pair) { const cx<T> xb = pre_gather<T>(p, ioB, pos, nat, op_resolve<PRE>(p.preOp)); x = cx<T>{x.x - xb.y, x.y + xb.x}; }
			}
			if (p.swapIn) x = cswap(x);
			v[m] = cmulc(x, chv(m));
		}
#pragma unroll
		for (int m = EH; m < E; m++) v[m] = cx<T>{(T)0, (T)0};
		pow2_stages<T, SCH, 0, TPF, 0, TwGlobal<T>>(v, lds + f * LDSPF, TwGlobal<T>{glut}, tau, waveOnly);
#pragma unroll
		for (int m = 0; m < E; m++) v[m] = cswap(cmul(v[m], bhv(m)));
		if constexpr (SCH::NS > 1) { if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC(); } // the exchange buffer is reused
		pow2_stages<T, SCH, 0, TPF, 0, TwGlobal<T>>(v, lds + f * LDSPF, TwGlobal<T>{glut}, tau, waveOnly);
		const bool split = pair && !realResult;
		if (split) { if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC(); } // (the spectrum of the pair goes through the exchange buffer once more)
#pragma unroll
		for (int m = 0; m < EH; m++) {
			const uint32_t pos = tau + m * TPF;
			cx<T> y = cmulc(cswap(v[m]), chv(m));
			if (p.swapOut) y = cswap(y);
			v[m] = y;
			if (pos < n) {
				if (split) lds[f * LDSPF + pos] = y;
				else if (pair) { post_scatter<T>(p, io, pos, cx<T>{y.x, (T)0}, 0, nat, op_resolve<POST>(p.postOp), p.outLen); post_scatter<T>(p, ioB, pos, cx<T>{y.y, (T)0}, 0, nat, op_resolve<POST>(p.postOp), p.outLen); }
				else post_scatter<T>(p, io, pos, y, 0, nat, op_resolve<POST>(p.postOp), p.outLen); // applies the scale
			}
		}
		if (split) {
			if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC();
#pragma unroll
			for (int m = 0; m < EH; m++) {
				const uint32_t pos = tau + m * TPF;
				if (pos < n) {
					const cx<T> y = v[m], ym = lds[f * LDSPF + (pos ? n - pos : 0u)];
					post_scatter<T>(p, io, pos, cx<T>{(T)0.5 * (y.x + ym.x), (T)0.5 * (y.y - ym.y)}, 0, nat, op_resolve<POST>(p.postOp), p.outLen);
					post_scatter<T>(p, ioB, pos, cx<T>{(T)0.5 * (y.y + ym.y), (T)0.5 * (ym.x - y.x)}, 0, nat, op_resolve<POST>(p.postOp), p.outLen);
				}
			}
		}
		if constexpr (SCH::NS > 1) { if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC(); }
	}
}

// ---- registry: one entry per (log2 M, dp, family) ------------------------------------------------------------------
struct Pow2BlueR2rVariant { Pow2Variant v; int pre; };
template <typename T, typename SCH, int FPW, int PRE, int POST> void pow2_blue_r2r_launch(const PassParams& prm, dim3 grid, hipStream_t s) {
	constexpr int threads = ((1 << SCH::LOGN) >> SCH::LOGE) * FPW;
	const unsigned resident = pow2_num_cus() * 8u;
	hipLaunchKernelGGL((pow2_blue_r2r_kernel<T, SCH, FPW, PRE, POST>), dim3(grid.x < resident ? grid.x : resident), dim3(threads), 0, s, prm);
}
#define VKFFT_P2BR1(T, dp, b0, b1, b2, b3, fpw, pre, post) \
	{ { (b0) + (b1) + (b2) + (b3), dp, {b0, b1, b2, b3}, fpw, (((1 << ((b0) + (b1) + (b2) + (b3))) >> Pow2Sched<b0, b1, b2, b3>::LOGE) * (fpw)), &pow2_blue_r2r_launch<T, Pow2Sched<b0, b1, b2, b3>, fpw, pre, post> }, pre }
#define VKFFT_P2BR(T, dp, b0, b1, b2, b3, fpw) \
	VKFFT_P2BR1(T, dp, b0, b1, b2, b3, fpw, OP_DCT1_PRE, OP_DCT1_POST), VKFFT_P2BR1(T, dp, b0, b1, b2, b3, fpw, OP_DST1_PRE, OP_DST1_POST), \
	VKFFT_P2BR1(T, dp, b0, b1, b2, b3, fpw, OP_DCT2_PRE, OP_DCT2_POST), VKFFT_P2BR1(T, dp, b0, b1, b2, b3, fpw, OP_DCT3_PRE, OP_DCT3_POST), \
	VKFFT_P2BR1(T, dp, b0, b1, b2, b3, fpw, OP_DCT4_PRE, OP_DCT4_POST), VKFFT_P2BR1(T, dp, b0, b1, b2, b3, fpw, OP_R2C_FULL, OP_R2C_FULL), \
	VKFFT_P2BR1(T, dp, b0, b1, b2, b3, fpw, OP_C2R_FULL, OP_C2R_FULL)
static const Pow2BlueR2rVariant kPow2BlueR2rVariants[] = {
	VKFFT_P2BR(float, false, 3, 3, 0, 0, 32),
	VKFFT_P2BR(float, false, 4, 3, 0, 0, 16),
	VKFFT_P2BR(float, false, 4, 4, 0, 0, 16),
	VKFFT_P2BR(float, false, 4, 3, 2, 0, 8),
	VKFFT_P2BR(float, false, 4, 3, 3, 0, 4),
	VKFFT_P2BR(float, false, 4, 4, 3, 0, 2),
	VKFFT_P2BR(float, false, 4, 4, 4, 0, 1),
	VKFFT_P2BR(float, false, 4, 3, 3, 3, 1),
	VKFFT_P2BR(float, false, 4, 4, 3, 3, 1),
	VKFFT_P2BR(double, true, 3, 3, 0, 0, 32),
	VKFFT_P2BR(double, true, 3, 2, 2, 0, 16),
	VKFFT_P2BR(double, true, 3, 3, 2, 0, 8),
	VKFFT_P2BR(double, true, 3, 3, 3, 0, 4),
	VKFFT_P2BR(double, true, 3, 3, 2, 2, 2),
	VKFFT_P2BR(double, true, 3, 3, 3, 2, 1),
	VKFFT_P2BR(double, true, 3, 3, 3, 3, 1),
	VKFFT_P2BR(double, true, 4, 3, 3, 3, 1),
};
constexpr int kNumPow2BlueR2rVariants = (int)(sizeof(kPow2BlueR2rVariants) / sizeof(kPow2BlueR2rVariants[0]));

int launch_pow2_blue_r2r(const PassPlan& pp, const PassParams& prm, hipStream_t stream) {
	const uint64_t grid64 = (uint64_t)prm.tilesPerG0 * prm.dim[1].count * prm.dim[2].count;
	if (grid64 == 0) return 0;
	if (grid64 > 0x7fffffffull || pp.variant < 0 || pp.variant >= kNumPow2BlueR2rVariants) return 4039;
	kPow2BlueR2rVariants[pp.variant].v.launch(prm, dim3((uint32_t)grid64), stream);
	return hipGetLastError() == hipSuccess ? 0 : 4039;
}

bool pow2_blue_r2r_lookup(uint32_t log2m, bool dp, uint32_t pre, int* variant, int bits[4], int* fpw, int* threads) {
	switch (pre) { // DST members run on the DCT instance of their family
	case OP_DST2_PRE: pre = OP_DCT2_PRE; break;
	case OP_DST3_PRE: pre = OP_DCT3_PRE; break;
	case OP_DST4_PRE: pre = OP_DCT4_PRE; break;
	default: break;
	}
	for (int i = 0; i < kNumPow2BlueR2rVariants; i++) {
		const Pow2BlueR2rVariant& e = kPow2BlueR2rVariants[i];
		if (e.v.log2n != (int)log2m || e.v.dp != dp || (uint32_t)e.pre != pre) continue;
		*variant = i;
		for (int k = 0; k < 4; k++) bits[k] = e.v.bits[k];
		*fpw = e.v.fpw; *threads = e.v.threads;
		return true;
	}
	return false;
}

} // namespace vkfft_mi355x
