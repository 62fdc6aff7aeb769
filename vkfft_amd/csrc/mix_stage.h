// One mixed-radix transform between two functors (shared by kernel_mixconv.h and the staged short rows of kernel_mixed.h): the compile-time radix
// schedule of mix_sched.h, ONE LDS buffer per transform that is the exchange buffer of the stages and may also be what the functors read and write.
#pragma once
#include "engine.h"
#include "butterflies.h"
#include "memops.h"
#include "mix_sched.h"

namespace vkfft_mi355x {

// LS / PADDED: the exchange buffer of a row transform is dense with a per-exchange padding (MixPad); a column tile interleaves its FPW columns
// (element pitch LS = FPW + 1, lanes along the columns: conflict-free without padding).
// SF / SL: in() reads / out() writes the exchange buffer itself (the carrier row of the convolution lives there between the phases).
// one transform of SCH::N points: in(t, c) delivers input t + c, out(t, c, v) receives output t + c (natural order on both sides); t is the lane's
// butterfly index, c a compile-time multiple of the butterfly count / stride (so that c can ride in the scalar offset of a buffer access)
// IN = McRegs: the inputs of the first stage are already in registers (x[b][i] = input tau + b * TPF + i * N / R0), loaded by the caller
template <typename T, int R> struct McRegs { const cx<T> (*x)[R]; };
template <typename X> struct McIsRegs { static constexpr bool value = false; };
template <typename T, int R> struct McIsRegs<McRegs<T, R>> { static constexpr bool value = true; };
template <typename T, typename SCH, int SI, int TPF, int LS, bool PADDED, bool SF, bool SL, typename IN, typename OUT>
__device__ inline void mc_stage(cx<T>* ldsf, const GBuf glut, const uint32_t tau, const bool waveOnly, const IN& in, const OUT& out) {
	constexpr int N = SCH::N, R = SCH::rad[SI], NB = N / R, P = (NB + TPF - 1) / TPF, S = SCH::S(SI);
	constexpr bool first = SI == 0, last = SI == SCH::NS - 1;
	constexpr uint32_t ES = (uint32_t)sizeof(cx<T>);
	using PAD = MixPad<SCH, TPF, (int)sizeof(cx<T>)>;
	cx<T> x[P][R];
#pragma unroll
	for (int b = 0; b < P; b++) {
		const uint32_t t = tau + b * TPF;
		if ((b + 1) * TPF <= NB || t < (uint32_t)NB) {
#pragma unroll
			for (int i = 0; i < R; i++) {
				if constexpr (first) {
					if constexpr (McIsRegs<IN>::value) x[b][i] = in.x[b][i];
					else x[b][i] = in(t, (uint32_t)(i * NB));
				}
				else x[b][i] = ldsf[mix_slot<PADDED ? PAD::shift(SI - 1) : 0>(t + i * NB) * LS];
			}
		}
	}
	// every input is in registers before the buffer is overwritten: middle stages always; the first stage when in() reads the buffer (SF), the last
	// stage when out() writes it (SL)
	if constexpr ((!first && !last) || (first && SF) || (last && SL)) { if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC(); }
#pragma unroll
	for (int b = 0; b < P; b++) {
		const uint32_t t = tau + b * TPF;
		if ((b + 1) * TPF <= NB || t < (uint32_t)NB) {
			const uint32_t s = t % (uint32_t)S;
			if constexpr (!first) {
				constexpr int LO = SCH::lutOff(SI);
#pragma unroll
				for (int i = 1; i < R; i++) x[b][i] = cmul(x[b][i], gb_load<T>(glut, s * ES, (uint32_t)(LO + (i - 1) * S) * ES));
			}
			dft<R, T>(x[b]);
			if constexpr (last) {
#pragma unroll
				for (int k = 0; k < R; k++) out(t, (uint32_t)(k * S), x[b][k]); // last stage: s = t
			} else {
				const uint32_t ob = (t - s) * (uint32_t)R + s;
#pragma unroll
				for (int k = 0; k < R; k++) ldsf[mix_slot<PADDED ? PAD::shift(SI) : 0>(ob + k * S) * LS] = x[b][k];
			}
		}
	}
	if constexpr (!last) {
		if (waveOnly) VKFFT_WAVE_SYNC(); else VKFFT_SYNC();
		mc_stage<T, SCH, SI + 1 < SCH::NS ? SI + 1 : SI, TPF, LS, PADDED, SF, SL>(ldsf, glut, tau, waveOnly, in, out);
	}
}

} // namespace vkfft_mi355x
