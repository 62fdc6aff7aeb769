// Kernel instantiations and the launch layer (the counterpart of the reference's hipModuleLaunchKernel
// glue, vkFFT_DispatchPlan.h:226-295 — but on ahead-of-time compiled gfx950 kernels).
#include "engine.h"
#include "kernel_generic.h"
#include "kernel_opfft.h"
#include "kernel_mixed.h"
#include "kernel_mixconv.h"
#include <cstdio>
#include <cstdlib>
#include <algorithm>

namespace vkfft_mi355x {

int launch_pass(const PassPlan& pp, const PassParams& prm, hipStream_t stream) {
	const uint64_t grid64 = (uint64_t)prm.tilesPerG0 * (prm.colMerge ? 1u : prm.dim[1].count) * prm.dim[2].count;
	if (grid64 == 0) return 0;
	if (grid64 > 0x7fffffffull) return 4039;
	const dim3 grid((uint32_t)grid64), block(pp.threads);
	switch (pp.kernel) {
	case KERNEL_GENERIC:
		if (pp.dp) hipLaunchKernelGGL(generic_pass_kernel<double>, grid, block, pp.ldsBytes, stream, prm);
		else hipLaunchKernelGGL(generic_pass_kernel<float>, grid, block, pp.ldsBytes, stream, prm);
		break;
	case KERNEL_POW2_ROW:
	case KERNEL_POW2_COL:
		return launch_pow2(pp, prm, stream);
	case KERNEL_POW2_BLUE:
		return launch_pow2_blue(pp, prm, stream);
	case KERNEL_POW2_COL_BLUE:
		return launch_pow2_col_blue(pp, prm, stream);
	case KERNEL_TRANSPOSE:
		return launch_transpose(pp, prm, stream);
	case KERNEL_REAL_MAP:
		return launch_real_map(pp, prm, stream);
	case KERNEL_POW2_BLUE_R2R:
		return launch_pow2_blue_r2r(pp, prm, stream);
	case KERNEL_MIXED_ROW:
		return launch_mixed(pp, prm, stream);
	case KERNEL_MIXCONV:
		return launch_mixconv(pp, prm, stream);
	case KERNEL_OPFFT:
		return launch_opfft(pp, prm, stream);
	case KERNEL_R2C_PAIR: {
		const uint32_t npair = (prm.opN >> 2) + 1;
		const uint64_t rows = (uint64_t)prm.dim[0].count * prm.dim[1].count * prm.dim[2].count;
		if (rows > 65535) { // grid.y limit: split over dim[2]/dim[1] on the host
			PassParams q = prm;
			if (prm.dim[2].count > 1) {
				for (uint32_t i = 0; i < prm.dim[2].count; i++) { q.dim[2].count = 1; q.out = (char*)prm.out + (int64_t)i * prm.dim[2].outStride * pp.outElemBytes; int r = launch_pass(pp, q, stream); if (r) return r; }
			} else if (prm.dim[1].count > 1) {
				for (uint32_t i = 0; i < prm.dim[1].count; i++) { q.dim[1].count = 1; q.out = (char*)prm.out + (int64_t)i * prm.dim[1].outStride * pp.outElemBytes; int r = launch_pass(pp, q, stream); if (r) return r; }
			} else {
				for (uint32_t i = 0; i < prm.dim[0].count; i += 32768) { q.dim[0].count = std::min<uint32_t>(32768, prm.dim[0].count - i); q.out = (char*)prm.out + (int64_t)i * prm.dim[0].outStride * pp.outElemBytes; int r = launch_pass(pp, q, stream); if (r) return r; }
			}
			return 0;
		}
		const dim3 g2((npair + 255) / 256, (uint32_t)rows);
		if (pp.dp) hipLaunchKernelGGL(r2c_even_pair_kernel<double>, g2, dim3(256), 0, stream, prm);
		else hipLaunchKernelGGL(r2c_even_pair_kernel<float>, g2, dim3(256), 0, stream, prm);
		break;
	}
	default:
		return 4039;
	}
	return hipGetLastError() == hipSuccess ? 0 : 4039;
}

// ---- mixed-radix registry: six table parts, one translation unit each (kernels_mixed_*.hip) --------------------
constexpr int kMixedParts = 20; // (parts 6-8: tools/gen_long_rows_table.py — the long rows; the 7-smooth lengths of 4097 ... 8192 points outside the first six tables)
const MixedVariant* mixed_table_0(int*);
const MixedVariant* mixed_table_1(int*);
const MixedVariant* mixed_table_2(int*);
const MixedVariant* mixed_table_3(int*);
const MixedVariant* mixed_table_4(int*);
const MixedVariant* mixed_table_5(int*);
const MixedVariant* mixed_table_6(int*);
const MixedVariant* mixed_table_7(int*);
const MixedVariant* mixed_table_8(int*);
const MixedVariant* mixed_table_9(int*);
const MixedVariant* mixed_table_10(int*);
const MixedVariant* mixed_table_11(int*);
const MixedVariant* mixed_table_12(int*);
const MixedVariant* mixed_table_13(int*);
const MixedVariant* mixed_table_14(int*);
const MixedVariant* mixed_table_15(int*);
const MixedVariant* mixed_table_16(int*);
const MixedVariant* mixed_table_17(int*);
const MixedVariant* mixed_table_18(int*);
const MixedVariant* mixed_table_19(int*);
static const MixedVariant* mixed_part(int part, int* count) {
	typedef const MixedVariant* (*Fn)(int*);
	static const Fn fns[kMixedParts] = {&mixed_table_0, &mixed_table_1, &mixed_table_2, &mixed_table_3, &mixed_table_4, &mixed_table_5, &mixed_table_6, &mixed_table_7, &mixed_table_8, &mixed_table_9, &mixed_table_10, &mixed_table_11, &mixed_table_12, &mixed_table_13, &mixed_table_14, &mixed_table_15, &mixed_table_16, &mixed_table_17, &mixed_table_18, &mixed_table_19};
	return fns[part % kMixedParts](count);
}
bool mixed_row_lookup(uint64_t n, bool dp, int* variant, int rad[5], int* fpw, int* threads) {
	for (int part = 0; part < kMixedParts; part++) {
		int cnt = 0;
		const MixedVariant* tab = mixed_part(part, &cnt);
		for (int i = 0; i < cnt; i++) {
			if ((uint64_t)tab[i].n != n || tab[i].dp != dp) continue;
			*variant = (part << 16) | i;
			for (int k = 0; k < 5; k++) rad[k] = tab[i].rad[k];
			*fpw = tab[i].fpw; *threads = tab[i].tpf * tab[i].fpw;
			return true;
		}
	}
	return false;
}
// rows per workgroup of the variant's form between the maps (kernel_mixed.h mixed_ops_fpw)
int mixed_row_ops_fpw(int variant) {
	int cnt = 0;
	const MixedVariant* tab = mixed_part((variant >> 16) % kMixedParts, &cnt);
	const int idx = variant & 0xffff;
	return (variant < 0 || idx >= cnt) ? 0 : tab[idx].fpwOps;
}
int launch_mixed(const PassPlan& pp, const PassParams& prm, hipStream_t stream) {
	const uint64_t grid64 = (uint64_t)prm.tilesPerG0 * prm.dim[1].count * prm.dim[2].count;
	if (grid64 == 0) return 0;
	int cnt = 0;
	const MixedVariant* tab = mixed_part((pp.variant >> 16) % kMixedParts, &cnt);
	const int idx = pp.variant & 0xffff;
	if (grid64 > 0x7fffffffull || pp.variant < 0 || idx >= cnt) return 4039;
	if (prm.preOp != OP_NONE || prm.postOp != OP_NONE) tab[idx].launchOps(prm, dim3((uint32_t)grid64), stream);
	else tab[idx].launch(prm, dim3((uint32_t)grid64), stream);
	return hipGetLastError() == hipSuccess ? 0 : 4039;
}

// ---- one-kernel cyclic convolution registry: six table parts (kernels_mixconv_*.hip) ---------------------------------
constexpr int kMixConvParts = 6;
const MixConvVariant* mixconv_table_0(int*);
const MixConvVariant* mixconv_table_1(int*);
const MixConvVariant* mixconv_table_2(int*);
const MixConvVariant* mixconv_table_3(int*);
const MixConvVariant* mixconv_table_4(int*);
const MixConvVariant* mixconv_table_5(int*);
static const MixConvVariant* mixconv_part(int part, int* count) {
	typedef const MixConvVariant* (*Fn)(int*);
	static const Fn fns[kMixConvParts] = {&mixconv_table_0, &mixconv_table_1, &mixconv_table_2, &mixconv_table_3, &mixconv_table_4, &mixconv_table_5};
	return fns[part % kMixConvParts](count);
}
bool mixconv_lookup(bool rader, bool col, uint64_t pOrMinLen, bool dp, int* variant, uint64_t* len, int rad[5], int* fpw, int* threads) {
	const MixConvVariant* best = nullptr;
	int bestId = -1;
	for (int part = 0; part < kMixConvParts; part++) {
		int cnt = 0;
		const MixConvVariant* tab = mixconv_part(part, &cnt);
		for (int i = 0; i < cnt; i++) {
			const MixConvVariant& v = tab[i];
			if (v.dp != dp || (v.rader != 0) != rader || (v.col != 0) != col) continue;
			if (rader ? (uint64_t)v.l + 1 != pOrMinLen : (uint64_t)v.l < pOrMinLen) continue;
			if (best && best->l <= v.l) continue;
			best = &v; bestId = (part << 16) | i;
		}
	}
	if (!best) return false;
	*variant = bestId; *len = (uint64_t)best->l;
	for (int k = 0; k < 5; k++) rad[k] = best->rad[k];
	*fpw = best->fpw; *threads = best->tpf * best->fpw;
	return true;
}
bool mixrad_available(int variant) {
	int cnt = 0;
	const MixConvVariant* tab = mixconv_part((variant >> 16) % kMixConvParts, &cnt);
	const int idx = variant & 0xffff;
	return variant >= 0 && idx < cnt && tab[idx].launchRad != nullptr;
}
bool mixrad_geom(int variant, int* sp, int* lutn, int* groups, int* groupsDense) {
	if (!mixrad_available(variant)) return false;
	int cnt = 0;
	const MixConvVariant* tab = mixconv_part((variant >> 16) % kMixConvParts, &cnt);
	*sp = tab[variant & 0xffff].radSP; *lutn = tab[variant & 0xffff].radLutN; *groups = tab[variant & 0xffff].radGroups; *groupsDense = tab[variant & 0xffff].radGroupsDense;
	return true;
}
int launch_mixconv(const PassPlan& pp, const PassParams& prm, hipStream_t stream) {
	const uint64_t grid64 = (uint64_t)prm.tilesPerG0 * (prm.colMerge ? 1u : prm.dim[1].count) * prm.dim[2].count;
	if (grid64 == 0) return 0;
	int cnt = 0;
	const MixConvVariant* tab = mixconv_part((pp.variant >> 16) % kMixConvParts, &cnt);
	const int idx = pp.variant & 0xffff;
	if (grid64 > 0x7fffffffull || pp.variant < 0 || idx >= cnt) return 4039;
	if (prm.raderM >= 1) { // the prime as a stage of the composite length raderM * P (kernel_mixrad.h; 1: the prime's own rows)
		if (!tab[idx].launchRad) return 4039;
		tab[idx].launchRad(prm, dim3((uint32_t)grid64), stream);
		return hipGetLastError() == hipSuccess ? 0 : 4039;
	}
	if ((prm.preOp != OP_NONE || prm.postOp != OP_NONE) && prm.preOp != OP_BLUESTEIN_PRE) { // (the Bluestein form handles its chirp itself: not the interpreter's maps)
		if (!tab[idx].launchOps) return 4039;
		tab[idx].launchOps(prm, dim3((uint32_t)grid64), stream);
		return hipGetLastError() == hipSuccess ? 0 : 4039;
	}
	tab[idx].launch(prm, dim3((uint32_t)grid64), stream);
	return hipGetLastError() == hipSuccess ? 0 : 4039;
}

// ---- op-FFT registry: eight table parts, one translation unit each (kernels_opfft_*.hip) ---------------------------
#define VKFFT_OPFFT_PARTS(X) X(f32_row_0) X(f32_row_1) X(f32_col_0) X(f32_col_1) X(f64_row_0) X(f64_row_1) X(f64_col_0) X(f64_col_1) X(f32_col_2) /* part 8: tools/gen_opfft_col_extra.py */
#define VKFFT_DECL(t) const OpfftVariant* opfft_table_##t(int*);
VKFFT_OPFFT_PARTS(VKFFT_DECL)
#undef VKFFT_DECL
static const OpfftVariant* opfft_part(int part, int* count) { // part = 2 * ((dp ? 2 : 0) + (col ? 1 : 0)) + half
	typedef const OpfftVariant* (*Fn)(int*);
#define VKFFT_REF(t) &opfft_table_##t,
	static const Fn fns[9] = { VKFFT_OPFFT_PARTS(VKFFT_REF) };
#undef VKFFT_REF
	return fns[part >= 0 && part < 9 ? part : 0](count);
}
static uint32_t opfft_family(uint32_t op) { // DST members run on the DCT instance of their family
	switch (op) {
	case OP_DST2_PRE: return OP_DCT2_PRE; case OP_DST2_POST: return OP_DCT2_POST;
	case OP_DST3_PRE: return OP_DCT3_PRE; case OP_DST3_POST: return OP_DCT3_POST;
	case OP_DST4_PRE: return OP_DCT4_PRE; case OP_DST4_POST: return OP_DCT4_POST;
	case OP_DST2H_PRE: return OP_DCT2H_PRE; case OP_DST2H_POST: return OP_DCT2H_POST;
	case OP_DST3H_PRE: return OP_DCT3H_PRE; case OP_DST3H_POST: return OP_DCT3H_POST;
	default: return op;
	}
}
bool opfft_lookup(uint64_t n, bool dp, bool col, bool trans, uint32_t pre, uint32_t post, int* variant, int rad[5], int* fpw, int* threads) {
	pre = opfft_family(pre); post = opfft_family(post);
	for (int half = 0; half < ((!dp && col) ? 3 : 2); half++) {
		const int part = half == 2 ? 8 : 2 * ((dp ? 2 : 0) + (col ? 1 : 0)) + half; // (fp32 column tiles have a third part)
		int cnt = 0;
		const OpfftVariant* tab = opfft_part(part, &cnt);
		for (int i = 0; i < cnt; i++) {
			if ((uint64_t)tab[i].n != n || (uint32_t)tab[i].pre != pre || (uint32_t)tab[i].post != post || tab[i].trans != trans) continue;
			*variant = (part << 16) | i;
			for (int k = 0; k < 5; k++) rad[k] = tab[i].rad[k];
			*fpw = tab[i].fpw; *threads = tab[i].tpf * tab[i].fpw;
			return true;
		}
	}
	return false;
}
int launch_opfft(const PassPlan& pp, const PassParams& prm, hipStream_t stream) {
	const uint64_t grid64 = (uint64_t)prm.tilesPerG0 * (prm.colMerge ? 1u : prm.dim[1].count) * prm.dim[2].count;
	if (grid64 == 0) return 0;
	int cnt = 0;
	const OpfftVariant* tab = opfft_part(pp.variant >> 16, &cnt);
	const int idx = pp.variant & 0xffff;
	if (grid64 > 0x7fffffffull || pp.variant < 0 || idx >= cnt) return 4039;
	tab[idx].launch(prm, dim3((uint32_t)grid64), stream);
	return hipGetLastError() == hipSuccess ? 0 : 4039;
}

static int launch_with_hostloop(const PassPlan& pp, PassParams prm, const StreamSet& ss, uint32_t& rr, size_t level) {
	if (level == pp.hostLoop.size()) return launch_pass(pp, prm, ss.s[pp.hostLoop.empty() ? 0 : (rr++ % ss.n)]);
	const HostDim& h = pp.hostLoop[level];
	for (uint64_t i = 0; i < h.count; i++) {
		PassParams q = prm;
		q.in = (const char*)prm.in + (int64_t)i * h.inStride * pp.inElemBytes;
		q.out = (char*)prm.out + (int64_t)i * h.outStride * pp.outElemBytes;
		int r = launch_with_hostloop(pp, q, ss, rr, level + 1);
		if (r) return r;
	}
	return 0;
}

static void bind(const DirectionPlan& plan, const PassPlan& pp, const LaunchBuffers& bufs, PassParams& prm) {
	prm.in = (const char*)bufs.base[pp.inRole] + pp.inOffset * pp.inElemBytes;
	prm.out = (char*)bufs.base[pp.outRole] + pp.outOffset * pp.outElemBytes;
	const char* ar = (const char*)plan.dArena;
	prm.lut = pp.lutOff != (size_t)-1 ? ar + pp.lutOff : nullptr;
	prm.aux = pp.auxOff != (size_t)-1 ? ar + pp.auxOff : nullptr;
	prm.aux2 = pp.auxIsKernel ? bufs.kernel : pp.aux2Off != (size_t)-1 ? ar + pp.aux2Off : nullptr;
	prm.aux3 = pp.aux3Off != (size_t)-1 ? ar + pp.aux3Off : nullptr;
	prm.rader = pp.raderOff != (size_t)-1 ? ar + pp.raderOff : nullptr;
	prm.tmPre = pp.tmPreOff != (size_t)-1 ? ar + pp.tmPreOff : nullptr;
	prm.tmPost = pp.tmPostOff != (size_t)-1 ? ar + pp.tmPostOff : nullptr;
}

int execute_direction(const DirectionPlan& plan, const LaunchBuffers& bufs, const StreamSet& ss, uint32_t* sweep) {
	hipStream_t stream = ss.s[0];
	const int np = (int)plan.passes.size();
	for (int i = 0; i < np; i++) {
		const PassPlan& pp = plan.passes[i];
		if (pp.kernel == KERNEL_POW2_FUSED || pp.kernel == KERNEL_MIX_FUSED) {
			FusedParams f = pp.fused;
			const char* ar = (const char*)plan.dArena;
			f.in = (const char*)bufs.base[pp.inRole] + pp.inOffset * pp.inElemBytes;
			f.out = (char*)bufs.base[pp.outRole] + pp.outOffset * pp.outElemBytes;
			f.scratch = bufs.base[ROLE_TEMP];
			f.lutA = ar + pp.lutOff; f.lutB = ar + pp.fusedLutBOff; f.tw4 = ar + pp.auxOff;
			f.rowTab = pp.fusedRowTabOff != (size_t)-1 ? ar + pp.fusedRowTabOff : nullptr;
			f.ctr = (uint32_t*)((char*)plan.dArena + pp.fusedCtrOff);
			if (sweep) { f.reverse = *sweep & 1u; *sweep ^= 1u; }
			int r = pp.kernel == KERNEL_MIX_FUSED ? launch_mix_fused(pp, f, stream) : launch_pow2_fused(pp, f, stream);
			if (r) return r;
			continue;
		}
		PassParams prm = pp.prm;
		bind(plan, pp, bufs, prm);
		if (sweep) { prm.reverseTiles = *sweep & 1u; *sweep ^= 1u; } // zig-zag: opposite to the previous launch of this application
		const bool fan = ss.n > 1 && !pp.hostLoop.empty(); // independent sub-launches: fan out over the caller's streams, join into s[0]
		if (fan) {
			if (hipEventRecord(ss.ev[0], ss.s[0]) != hipSuccess) return 4040;
			for (uint32_t k = 1; k < ss.n; k++) if (hipStreamWaitEvent(ss.s[k], ss.ev[0], 0) != hipSuccess) return 4040;
		}
		uint32_t rr = 0;
		int r = launch_with_hostloop(pp, prm, ss, rr, 0);
		if (r) return r;
		if (fan) {
			for (uint32_t k = 1; k < ss.n; k++) {
				if (hipEventRecord(ss.ev[k], ss.s[k]) != hipSuccess) return 4040;
				if (hipStreamWaitEvent(ss.s[0], ss.ev[k], 0) != hipSuccess) return 4040;
			}
		}
	}
	return 0;
}

} // namespace vkfft_mi355x
