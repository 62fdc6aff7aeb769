// Planner: transform description -> list of passes (kernel launches) with all index maps, twiddle tables
// and pre/post operations resolved.  Behavioural counterpart of the reference's VkFFTScheduler
// (vkFFT_Scheduler.h:2223: factorisation, pass count from on-chip capacity, radix list),
// VkFFTSplitAxisBlock (vkFFT_AxisBlockSplitter.h:26: workgroup shape), VkFFTPlanAxis
// (vkFFT_Plan_FFT.h:33: strides, batch folding) and VkFFT_AllocateLUT (vkFFT_ManageLUT.h:28: tables
// computed in extended precision on the CPU).  The decisions themselves are re-derived for MI355X:
// 160 KiB LDS per workgroup, 256-byte coalescing segments for strided tiles, fused Four-Step through the Infinity Cache.
#include "engine.h"
#include "mixrad_plan.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <complex>
#include <map>
#include <mutex>

namespace vkfft_mi355x {

typedef long double ld;
typedef std::complex<ld> cld;
static const ld PI_LD = 3.14159265358979323846264338327950288419716939937510L;

// exp(-2*pi*i*num/den), argument reduced exactly in integers first
static cld unit_root(uint64_t num, uint64_t den) {
	num %= den;
	// octant reduction keeps |angle| <= pi/4 for best accuracy
	ld x = (ld)num / (ld)den; // in [0,1)
	uint64_t oct8 = (uint64_t)(8 * x + 0.5L);
	(void)oct8;
	ld ang = 2 * PI_LD * x;
	return cld(cosl(ang), -sinl(ang));
}

struct Arena {
	std::vector<unsigned char>& b;
	explicit Arena(std::vector<unsigned char>& v) : b(v) {}
	size_t alloc(size_t bytes) {
		size_t off = (b.size() + 255) & ~(size_t)255;
		b.resize(off + bytes);
		return off;
	}
	template <typename T> void put(size_t off, size_t idx, cld v) {
		T* p = (T*)(b.data() + off);
		p[2 * idx] = (T)v.real();
		p[2 * idx + 1] = (T)v.imag();
	}
	void putc(size_t off, size_t idx, cld v, bool dp) {
		if (dp) put<double>(off, idx, v); else put<float>(off, idx, v);
	}
};

std::vector<uint32_t> factorize_radices(uint64_t n, bool* smooth) {
	std::vector<uint32_t> f;
	const uint32_t primes[6] = {2, 3, 5, 7, 11, 13};
	for (uint32_t p : primes) while (n % p == 0) { f.push_back(p); n /= p; }
	// remaining part: trial division (larger primes are handled by Rader/Bluestein)
	for (uint64_t p = 17; p * p <= n; p += 2) while (n % p == 0) { f.push_back((uint32_t)p); n /= p; }
	if (n > 1) f.push_back((uint32_t)n);
	if (smooth) { *smooth = true; for (uint32_t p : f) if (p > 13) *smooth = false; }
	return f;
}

// radix list for one pass of length L (L's prime factors <= 13, or small Rader primes)
static std::vector<uint32_t> radix_schedule(uint64_t L) {
	bool smooth;
	std::vector<uint32_t> pf = factorize_radices(L, &smooth);
	int twos = 0;
	std::vector<uint32_t> rad;
	for (uint32_t p : pf) { if (p == 2) twos++; else rad.push_back(p); }
	// split 2^twos into ceil(twos/4) radices of (almost) equal size, largest first
	if (twos > 0) {
		int parts = (twos + 3) / 4;
		int base = twos / parts, extra = twos % parts;
		for (int i = 0; i < parts; i++) rad.push_back(1u << (base + (i < extra ? 1 : 0)));
	}
	std::sort(rad.begin(), rad.end(), [](uint32_t a, uint32_t b) { return a > b; });
	return rad;
}

static bool smooth13(uint64_t n) {
	for (uint64_t q : {2, 3, 5, 7, 11, 13}) while (n % q == 0) n /= q;
	return n == 1;
}
static bool is_prime_u(uint64_t n) { if (n < 2) return false; for (uint64_t d = 2; d * d <= n; d++) if (n % d == 0) return false; return true; }
// FFT-convolution Rader is available for primes whose P-1 is {2..13}-smooth (reference: VkFFTConstructRaderTree,
// vkFFT_Scheduler.h:1764-1777) and small enough to sit in LDS beside the data
static bool rader_fft_ok(uint64_t p) { return p > 16 && p <= 8191 && is_prime_u(p) && smooth13(p - 1); }
static uint64_t powmod(uint64_t b, uint64_t e, uint64_t m) { uint64_t r = 1; b %= m; while (e) { if (e & 1) r = r * b % m; b = b * b % m; e >>= 1; } return r; }
static uint64_t primitive_root(uint64_t p) { // brute force as the reference does (vkFFT_Scheduler.h:1818-1831)
	std::vector<uint64_t> pf; uint64_t n = p - 1;
	for (uint64_t q = 2; q * q <= n; q++) if (n % q == 0) { pf.push_back(q); while (n % q == 0) n /= q; }
	if (n > 1) pf.push_back(n);
	for (uint64_t g = 2; g < p; g++) { bool ok = true; for (uint64_t q : pf) if (powmod(g, (p - 1) / q, p) == 1) { ok = false; break; } if (ok) return g; }
	return 0;
}

static uint32_t ilog2(uint64_t v) { uint32_t l = 0; while ((1ull << (l + 1)) <= v) l++; return l; }
static uint32_t ceil_log2(uint64_t v) { uint32_t l = 0; while ((1ull << l) < v) l++; return l; }

static void host_fft(std::vector<cld>& a);

struct PassBuild {
	uint64_t L = 0;
	int64_t inStrideJ = 1, outStrideJ = 1;
	std::vector<HostDim> dims; // dims[0] tiled
	bool colIn = false, colOut = false;
	bool dp = false;
	uint32_t preOp = OP_NONE, midOp = OP_NONE, postOp = OP_NONE;
	uint32_t inLen = 0, outLen = 0, opN = 0, blueN = 0;
	bool preNat = false, postNat = false; uint32_t natDimMask = 1, natOutLen = 0; // multi-pass real transforms (PassParams::preNat ...)
	bool swapIn = false, swapOut = false, bsSwapIn = false, bsSwapOut = false;
	uint64_t fsN = 0; uint32_t fsColDiv = 1; bool fsColFromDim1 = false;
	uint32_t opStrideJ = 1, opStride0 = 0, opStride1 = 0; // natural-position index of element j of sub-FFT (g0,g1) for position-indexed ops
	size_t auxOff2ForPre = (size_t)-1;
	double scale = 1.0;
	int inRole = ROLE_BUFFER, outRole = ROLE_BUFFER;
	int64_t inOffset = 0, outOffset = 0;
	bool realIn = false, realOut = false;
	uint32_t raderDirectMax = 0;
	std::string label;
	uint32_t forceT = 0;
	uint32_t raderM = 0, raderA = 0, raderAligned = 0; // mixrad_kernel: cofactor of the composite length, its split, layout of the thread groups (kernel_mixrad.h)
	std::vector<uint32_t> radices; // explicit stage radices (fast kernels fix their own schedule)
	int fastKernel = KERNEL_GENERIC, fastVariant = -1, fastThreads = 0;
	bool allowFast = true;
	bool allowOp = false;   // op-FFT family (kernel_opfft.h): fused pre/post map kernels
	bool noCollapse = false;
	uint32_t padInL = 0, padInN = 0, padOutL = 0, padOutN = 0; // zero padding along J (PassParams::padIn* / padOut*)
	bool bigSpan = false;
	uint64_t maxLds = 160 * 1024;
	// tables prepared by the caller (arena offsets)
	size_t auxOff = (size_t)-1, aux2Off = (size_t)-1;
};

// collapse dims that form one arithmetic progression on both sides; keep dims[0] (tiled) separate unless
// it merges with dims[1] exactly (then tiles simply continue across the boundary)
static void collapse_dims(std::vector<HostDim>& d) {
	// drop count-1 dims (except keep at least one)
	std::vector<HostDim> r;
	for (auto& x : d) if (x.count > 1) r.push_back(x);
	if (r.empty()) r.push_back({1, 0, 0});
	for (size_t i = 0; i + 1 < r.size();) {
		if ((int64_t)r[i].count * r[i].inStride == r[i + 1].inStride && (int64_t)r[i].count * r[i].outStride == r[i + 1].outStride
		    && r[i].count * r[i + 1].count < (1ull << 31)) {
			r[i].count *= r[i + 1].count;
			r.erase(r.begin() + i + 1);
		} else i++;
	}
	d = r;
}

// Tables of the one-kernel Rader transform of prime length P (kernel_mixconv.h): uint32 g^a, g^-k mod P (a, k < P-1) and the FFT of
// b_q = exp(-2 pi i g^-q / P) scaled by 1/(P-1)  (vkFFT_RecursiveFFTGenerators.h:1021-1048)
static void make_mixconv_rader_tables(uint64_t P, bool dp, Arena& ar, size_t& tabOff, size_t& bhatOff) {
	const uint64_t L = P - 1, g = primitive_root(P), gi = powmod(g, P - 2, P);
	tabOff = ar.alloc(2 * (size_t)L * sizeof(uint32_t));
	std::vector<cld> bk(L);
	{
		uint32_t* tab = (uint32_t*)(ar.b.data() + tabOff);
		uint64_t gp = 1, gm = 1;
		for (uint64_t q = 0; q < L; q++) { tab[q] = (uint32_t)gp; tab[L + q] = (uint32_t)gm; bk[q] = unit_root(gm, P); gp = gp * g % P; gm = gm * gi % P; }
	}
	host_fft(bk);
	bhatOff = ar.alloc(L * (dp ? 16 : 8));
	for (uint64_t m = 0; m < L; m++) ar.putc(bhatOff, m, bk[m] / (ld)L, dp);
}

// ... followed by the tables of the column steps of the composite form (kernel_mixrad.h), N = M * P: W_N^lo (lo < 64), W_N^(64 hi) (hi < ceil(N / 64)) — the
// twiddle W_N^(b k2) is the product of two entries — and W_M^e (e < M)
static void make_mixrad_tables(uint64_t P, uint64_t M, bool dp, Arena& ar, size_t& tabOff, size_t& bhatOff) {
	const uint64_t L = P - 1, g = primitive_root(P), gi = powmod(g, P - 2, P), N = M * P, NH = (N + 63) / 64;
	tabOff = ar.alloc(2 * (size_t)L * sizeof(uint32_t));
	std::vector<cld> bk(L);
	{
		uint32_t* tab = (uint32_t*)(ar.b.data() + tabOff);
		uint64_t gp = 1, gm = 1;
		for (uint64_t q = 0; q < L; q++) { tab[q] = (uint32_t)gp; tab[L + q] = (uint32_t)gm; bk[q] = unit_root(gm, P); gp = gp * g % P; gm = gm * gi % P; }
	}
	host_fft(bk);
	bhatOff = ar.alloc((L + 64 + NH + M) * (dp ? 16 : 8));
	for (uint64_t m = 0; m < L; m++) ar.putc(bhatOff, m, bk[m] / (ld)L, dp);
	for (uint64_t e = 0; e < 64; e++) ar.putc(bhatOff, L + e, unit_root(e % N, N), dp);
	for (uint64_t h = 0; h < NH; h++) ar.putc(bhatOff, L + 64 + h, unit_root((64 * h) % N, N), dp);
	for (uint64_t e = 0; e < M; e++) ar.putc(bhatOff, L + 64 + NH + e, unit_root(e, M), dp);
}
// A row of L = M * P complex points on the Rader-stage kernel (kernel_mixrad.h): P the largest prime factor (37 or more, with a Rader row instance), M a cofactor
// that splits into the kernel's column radices (mixrad_plan.h).  VKFFT_MI355X_MIXRAD=0: off; VKFFT_MI355X_MIXRAD_LDS_KIB: the LDS budget of a tile (tuning)
struct MixradChoice { uint64_t P = 0, M = 0, len = 0; uint32_t A = 0, rows = 0, aligned = 0; int variant = -1, rad[5] = {1, 1, 1, 1, 1}, fpw = 0, threads = 0; double cost = 2.0; };
static bool mixrad_choose(uint64_t L, bool dp, bool ops, MixradChoice& c) { // ops: a real transform between the table-driven maps
	if (dp || L < 37 || L > kMixradLongest) return false;
	if (getenv("VKFFT_MI355X_MIXRAD") && atoi(getenv("VKFFT_MI355X_MIXRAD")) == 0) return false;
	uint64_t P = 0, rest = L;
	for (uint64_t q = 2; q * q <= rest; q++) while (rest % q == 0) { P = q; rest /= q; }
	if (rest > 1) P = rest; // largest prime factor
	if (P < 37) return false;
	const uint64_t M = L / P;
	uint32_t A = 0, B = 1;
	if (M == 1) { // a prime's own rows: no column step (VKFFT_MI355X_MIXRAD_PRIMES=1; default: kernel_mixconv.h)
		if (!(getenv("VKFFT_MI355X_MIXRAD_PRIMES") && atoi(getenv("VKFFT_MI355X_MIXRAD_PRIMES")) == 1)) return false; // (off by default: measured slower than kernel_mixconv.h)
		A = 1;
	} else if (M == P) A = 0; // P * P: the column transform is the prime's own convolution (kernel_mixrad.h, 2b)
	else if (!mixrad_split((uint32_t)M, A, B)) return false;
	uint64_t len; int sp = 0, lutn = 0, groups = 0, gd = 0;
	if (!mixconv_lookup(true, false, P, dp, &c.variant, &len, c.rad, &c.fpw, &c.threads) || !mixrad_geom(c.variant, &sp, &lutn, &groups, &gd)) return false;
	if (gd <= 0 || sp <= 0) return false; // (an instance without the stage form — the registry leaves its geometry at zero: DST-I of 1782 reals, 2 * 1783 complex points, divided by it)
	const uint64_t budget = (getenv("VKFFT_MI355X_MIXRAD_LDS_KIB") ? (uint64_t)atoll(getenv("VKFFT_MI355X_MIXRAD_LDS_KIB")) : 40ull) << 10;
	const bool twoSets = ops && mixrad_two_sets((uint32_t)M, A);
	c.P = P; c.M = M; c.A = A; c.len = len;
	// layout of the convolution's thread groups (kernel_mixrad.h MixradGeom): the wave-aligned one runs its rounds without workgroup barriers (a round costs about 0.8
	// of a dense one: 3144, 3130, 3611, 314 1.2-1.5x faster) but may have fewer groups than the dense FPW — a round more where the cofactor was matched to FPW
	// (3232 = 32 * 101, 2032 = 16 * 127: 0.9x).  Cost of a row = rounds per tile / rows per tile, per layout with its own tile
	c.rows = mixrad_rows((uint32_t)P, (uint32_t)sp, (uint32_t)lutn, (uint32_t)gd, (uint32_t)M, dp ? 16u : 8u, budget, twoSets);
	if (groups > 0 && !getenv("VKFFT_MI355X_MIXRAD_DENSE")) {
		const uint32_t ra = mixrad_rows((uint32_t)P, (uint32_t)sp, (uint32_t)lutn, (uint32_t)groups, (uint32_t)M, dp ? 16u : 8u, budget, twoSets);
		auto cost = [&](uint32_t rows, uint32_t g, double w) { const uint64_t jobs = (uint64_t)rows * M; return w * (double)((jobs + g - 1) / g) / (double)rows; };
		if (cost(ra, (uint32_t)groups, 0.8) <= cost(c.rows, (uint32_t)gd, 1.0)) { c.rows = ra; c.aligned = 1; }
	}
	if (mixrad_lds_bytes((uint32_t)P, (uint32_t)sp, (uint32_t)lutn, (uint32_t)M, c.rows, dp ? 16u : 8u, twoSets) > 160ull * 1024) return false;
	// points of the fused power-of-two Bluestein transform that one point of the row costs (profiles/r06_rader_stage_forced_vs_bluestein.jsonl: every served class
	// forced either way): 1.2-2.2 with register column steps; a direct last step of radix B adds (B / 33)^2.2 (23: 2.4, 29: 2.5, 37: 3.4, 49: 4.3); primes whose
	// own convolution has a radix-13 stage (131, 157, 313: 2.2-3.2) one more, 521 and up (three stages, 138 registers: 3.9) two
	c.cost = 1.9;
	if (M > 1 && A != 0 && !mixrad_radix_reg(B)) c.cost += std::pow((double)B / 33.0, 2.2);
	if ((P - 1) % 13 == 0) c.cost += P >= 500 ? 2.0 : 1.0;
	return true;
}

// Real-transform families that can carry TWO rows per complex transform (kernel_generic.h ops_rows_in / ops_rows_out): the pre-map of a row is a real
// sequence (R2C of odd length, DCT / DST-I, -II and odd -IV in their full-length forms) or the result is real (C2R of odd length, DCT / DST-III).
// 0: never over a fused-map instance; 1: every pairable family; 2 (default since round 6): DCT-II / -III and the odd DCT-IV — measured with the planner forced either way on every
// length 4 ... 400 (profiles/r06_real_rows_pairs_preferred_over_fused_map_instances.jsonl): DCT-IV of 5, 25, 35 ... 245 reals 1.10-1.83x faster between the tables than on
// their fused-map instance (none slower), DCT-II / -III of 7, 25, 49, 175, 343 1.03-1.45x; R2C / C2R mixed (7: 1.29x, 25 and 49: 0.75x) and left where they were
constexpr int kPairPreferDefault = 2;
static bool pairable_family(uint32_t pre, uint32_t post, uint64_t cplxLen, uint32_t opN) {
	auto fam = [&](uint32_t a, uint32_t c) { return pre == a && post == c; };
	const bool odd4 = (fam(OP_DCT4_PRE, OP_DCT4_POST) || fam(OP_DST4_PRE, OP_DST4_POST)) && cplxLen == opN;
	// (the members whose operation the kernels hold as a compile-time constant — dispatch_pre_op / dispatch_post_op; the DST members and DCT-I / DST-I run through the
	// run-time switch, where the paired loop body would not inline: kernel_generic.h)
	const bool odd4c = fam(OP_DCT4_PRE, OP_DCT4_POST) && cplxLen == opN;
	(void)odd4;
	return fam(OP_R2C_FULL, OP_R2C_FULL) || fam(OP_C2R_FULL, OP_C2R_FULL) || fam(OP_DCT2_PRE, OP_DCT2_POST) || fam(OP_DCT3_PRE, OP_DCT3_POST) || odd4c;
}
// ---- table-driven maps of the real transforms (kernel_tmaps.h) -------------------------------------------------------------------------------------
// One pair of tables per (family, length): what pre_gather / post_store / post_scatter of kernel_generic.h compute per element, written out per position
// (reference: the per-family index arithmetic and twiddles of vkFFT_R2C.h:178,450 and vkFFT_R2R.h:193-336, 414-481, 784-1031, 1339-2318).
// Families: R2C (post) / C2R (pre) in their full-length forms; DCT / DST-I, -II, -III in their full-length forms; DCT / DST-IV of odd length (same-length form)
// and of even length (half-length complex form).  All but the last carry two rows per transform.
constexpr uint32_t kTmNoTerm = 0x7FFFFFF8u; // = kGbInvalid (memops.h): a byte offset no buffer access reaches
static_assert((kTmNoTerm & 1u) == 0u && kTmNoTerm >= 0x7FFFFFF0u, "bit 0 of an offset carries the sign of the second term (set_pre2); the value must lie beyond the range of a buffer resource (memops.h kGbRange)");
enum TmFamily { TM_NONE = 0, TM_R2C, TM_C2R, TM_R2R2, TM_R2R3, TM_DCT1, TM_DST1, TM_R2R4_ODD, TM_R2R4_EVEN };
static TmFamily tm_family(uint32_t pre, uint32_t post, uint64_t Lc, uint32_t N) {
	auto fam = [&](uint32_t a, uint32_t c) { return pre == a && post == c; };
	if (N < 2) return TM_NONE;
	if (fam(OP_R2C_FULL, OP_R2C_FULL) && Lc == N) return TM_R2C;
	if (fam(OP_C2R_FULL, OP_C2R_FULL) && Lc == N) return TM_C2R;
	if ((fam(OP_DCT2_PRE, OP_DCT2_POST) || fam(OP_DST2_PRE, OP_DST2_POST)) && Lc == N) return TM_R2R2;
	if ((fam(OP_DCT3_PRE, OP_DCT3_POST) || fam(OP_DST3_PRE, OP_DST3_POST)) && Lc == N) return TM_R2R3;
	if (fam(OP_DCT1_PRE, OP_DCT1_POST) && Lc == 2ull * N - 2) return TM_DCT1;
	if (fam(OP_DST1_PRE, OP_DST1_POST) && Lc == 2ull * N + 2) return TM_DST1;
	if (fam(OP_DCT4_PRE, OP_DCT4_POST) || fam(OP_DST4_PRE, OP_DST4_POST)) {
		if (Lc == N && (N & 1u) && N >= 3) return TM_R2R4_ODD;
		if (2 * Lc == N) return TM_R2R4_EVEN;
	}
	return TM_NONE;
}
// threads per row from which the table-driven maps replace the generic ones: 1 — since the staging tile of round 5 (rows with fewer than eight threads move as one
// contiguous run through LDS, kernel_mixed.h) the tables win at every shape measured (31 reals, one thread per row: 0.45 -> 0.19 ms); the switch stays for A/B runs
static int tmaps_min_tpf() { return getenv("VKFFT_MI355X_TMAPS_MIN_TPF") ? atoi(getenv("VKFFT_MI355X_TMAPS_MIN_TPF")) : 1; }
static bool tm_family_pairs(TmFamily f) { return f != TM_NONE && f != TM_R2R4_EVEN; }
struct TmTable {
	Arena& ar; size_t off; uint32_t n; bool dp;
	TmTable(Arena& a, uint32_t entries, bool dp_) : ar(a), n(entries), dp(dp_) {
		const size_t offsBytes = ((size_t)entries * 8 + 15) & ~(size_t)15;
		off = ar.alloc(offsBytes + (size_t)entries * (dp ? 32 : 16));
		for (uint32_t i = 0; i < entries; i++) set(i, kTmNoTerm, kTmNoTerm, cld(0, 0), cld(0, 0));
	}
	// two-term pre-map entries: c2 = +-i c1 in every family, and the kernel reads only c1: bit 0 of o2 set = the minus sign (kernel_tmaps.h tm_pre)
	void set_pre2(uint32_t i, uint32_t o1, uint32_t o2, cld c1, bool minus) { set(i, o1, o2 | (minus ? 1u : 0u), c1, (minus ? cld(0, -1) : cld(0, 1)) * c1); }
	void set(uint32_t i, uint32_t o1, uint32_t o2, cld c1, cld c2) {
		uint32_t* o = (uint32_t*)(ar.b.data() + off);
		o[2 * i] = o1; o[2 * i + 1] = o2;
		const size_t coef = off + (((size_t)n * 8 + 15) & ~(size_t)15);
		ar.putc(coef, 2 * (size_t)i, c1, dp); ar.putc(coef, 2 * (size_t)i + 1, c2, dp);
	}
};
// preFlags / postFlags = 0: that side keeps the kernel's own form (R2C reads its reals, C2R writes its reals directly)
// direct: the kernel moves the plain sides itself (eight or more threads per row, kernel_mixed.h DIRECT)
static void build_tmaps(TmFamily fam, bool dst, uint64_t Lc, uint32_t N, bool dp, double scaleD, bool direct, Arena& ar, PassPlan& pp) {
	PassParams& p = pp.prm;
	const uint32_t L = (uint32_t)Lc, RS = dp ? 8u : 4u, H = L / 2 + 1;
	const ld sc = (ld)scaleD;
	const cld I(0, 1);
	auto ro = [&](uint64_t idx) { return (uint32_t)(idx * RS); };
	// ---- pre-map: FFT input pos
	if (fam != TM_R2C || !direct) {
		TmTable t(ar, L, dp);
		bool two = false;
		for (uint32_t pos = 0; pos < L; pos++) {
			switch (fam) {
			case TM_R2C: t.set(pos, ro(pos), kTmNoTerm, cld(1, 0), cld(0, 0)); break;
			case TM_C2R: { // X[pos] for pos <= N/2, conj X[N - pos] beyond; the row = (re, im) pairs
				const bool lo = pos <= N / 2; const uint64_t e = lo ? pos : N - pos;
				t.set_pre2(pos, ro(2 * e), ro(2 * e + 1), cld(1, 0), !lo); two = true; break;
			}
			case TM_R2R2: { // Makhoul permutation (kernel_generic.h OP_DCT2_PRE / OP_DST2_PRE)
				const uint64_t src = pos < (N + 1) / 2 ? 2ull * pos : 2ull * (N - 1 - pos) + 1;
				t.set(pos, ro(src), kTmNoTerm, cld((dst && (src & 1)) ? -1 : 1, 0), cld(0, 0)); break;
			}
			case TM_R2R3: { // V_k = e^{+i pi k / 2N} (x_k - i x_{N-k}), x_N = 0; DST-III reads the reversed row
				const cld w = std::conj(unit_root(pos, 4ull * N));
				const uint64_t ia = dst ? N - 1 - pos : pos;
				const uint32_t ob = pos == 0 ? kTmNoTerm : ro(dst ? pos - 1 : N - pos);
				t.set_pre2(pos, ro(ia), ob, w, true); two = true; break;
			}
			case TM_DCT1: { const uint64_t M = 2ull * N - 2; t.set(pos, ro(pos < N ? pos : M - pos), kTmNoTerm, cld(1, 0), cld(0, 0)); break; }
			case TM_DST1:
				if (pos == 0 || pos == N + 1) break;
				if (pos <= N) t.set(pos, ro(pos - 1), kTmNoTerm, cld(1, 0), cld(0, 0));
				else t.set(pos, ro(2ull * N + 1 - pos), kTmNoTerm, cld(-1, 0), cld(0, 0));
				break;
			case TM_R2R4_ODD: { // the row sampled at r = 8 i + N (kernel_generic.h OP_DCT4_PRE, same-length form)
				const uint64_t m = 4ull * pos + (N >> 1);
				uint64_t src; bool neg = false;
				if (m < N) src = m;
				else if (m < 2ull * N) { src = 2ull * N - 1 - m; neg = true; }
				else if (m < 3ull * N) { src = m - 2ull * N; neg = true; }
				else if (m < 4ull * N) src = 4ull * N - 1 - m;
				else src = m - 4ull * N;
				t.set(pos, ro(dst ? N - 1 - src : src), kTmNoTerm, cld(neg ? -1 : 1, 0), cld(0, 0)); break;
			}
			case TM_R2R4_EVEN: { // z = (x[2 pos] + i x[N - 1 - 2 pos]) e^{-i pi (4 pos + 1) / 4N}
				uint64_t i0 = 2ull * pos, i1 = N - 1 - 2ull * pos;
				if (dst) { i0 = N - 1 - i0; i1 = N - 1 - i1; }
				const cld w = unit_root(4ull * pos + 1, 8ull * N);
				t.set_pre2(pos, ro(i0), ro(i1), w, false); two = true; break;
			}
			default: break;
			}
		}
		pp.tmPreOff = t.off; p.tmPreFlags = kTmOn | (two ? kTmTwo : 0u);
	}
	// ---- post-map
	if (fam == TM_C2R && direct) return; // (the real parts go out directly, scaled: kernel_mixed.h)
	if (fam == TM_R2R3 || fam == TM_C2R) { // real results: output o of both rows from FFT output m
		TmTable t(ar, L, dp);
		for (uint32_t m = 0; m < L; m++) {
			const uint64_t o = fam == TM_C2R ? m : m < (N + 1) / 2 ? 2ull * m : 2ull * (N - 1 - m) + 1;
			const ld s = (dst && (o & 1)) ? -sc : sc;
			t.set(m, ro(o), kTmNoTerm, cld(s, 0), cld(0, -s));
		}
		pp.tmPostOff = t.off; p.tmPostFlags = kTmOn | kTmRowB;
		return;
	}
	if (fam == TM_R2R4_EVEN) { // FFT output m feeds outputs 2m and N - 1 - 2m
		TmTable t(ar, L, dp);
		for (uint32_t m = 0; m < L; m++) {
			const cld c = unit_root(m, 2ull * N) * (2 * sc);
			t.set(m, ro(2ull * m), ro(N - 1 - 2ull * m), c, (dst ? -I : I) * c);
		}
		pp.tmPostOff = t.off; p.tmPostFlags = kTmOn;
		return;
	}
	// split forms: index k <= L/2 carries X[k] and (conjugated) X[L - k]; X_a, X_b arrive doubled
	TmTable t(ar, H, dp);
	const ld h = sc / 2;
	for (uint32_t k = 0; k < H; k++) {
		const uint32_t km = k ? L - k : 0; // the mirror index; outputs that belong to it are written from here unless it is k itself
		const bool mirror = km != k;
		switch (fam) {
		case TM_R2C: t.set(k, ro(2ull * k), kTmNoTerm, cld(h, 0), cld(0, -h)); break; // one complex store (Re X, Im X)
		case TM_R2R2: { // y[q] = 2 Re(e^{-i pi q / 2N} X[q]); DST-II: output N - 1 - q
			auto outOf = [&](uint64_t q) { return ro(dst ? N - 1 - q : q); };
			const cld c1 = unit_root(k, 4ull * N) * (2 * h);
			if (mirror) t.set(k, outOf(k), outOf(km), c1, std::conj(unit_root(km, 4ull * N)) * (2 * h));
			else t.set(k, outOf(k), kTmNoTerm, c1, cld(0, 0));
			break;
		}
		case TM_DCT1: t.set(k, ro(k), kTmNoTerm, cld(h, 0), cld(0, 0)); break; // y[k] = Re X[k], k <= N - 1 = L/2
		case TM_DST1: if (k >= 1 && k <= N) t.set(k, ro(k - 1), kTmNoTerm, I * h, cld(0, 0)); break; // y[k - 1] = -Im X[k]
		case TM_R2R4_ODD: { // y[j] = 2 Re(e^{-i pi u / 4} X[u mod N]), u = 2 j + 1 (kernel_generic.h OP_DCT4_POST, same-length form)
			auto outJ = [&](uint64_t q) { return (q & 1) ? (q - 1) >> 1 : (q + N - 1) >> 1; }; // the output whose u = q (mod N)
			auto coefJ = [&](uint64_t j) {
				const uint64_t r8 = (2 * j + 1) & 7u;
				const ld c = (r8 == 1 || r8 == 7) ? 1 : -1, s2 = (r8 == 1 || r8 == 3) ? 1 : -1;
				ld g = 1.41421356237309504880168872420969807856967L * h;
				if (dst && (j & 1)) g = -g;
				return cld(c * g, -s2 * g); // Re((c - i s)(x + i y)) = c x + s y
			};
			const uint64_t j1 = outJ(k);
			if (mirror) { const uint64_t j2 = outJ(km); t.set(k, ro(j1), ro(j2), coefJ(j1), std::conj(coefJ(j2))); }
			else t.set(k, ro(j1), kTmNoTerm, coefJ(j1), cld(0, 0));
			break;
		}
		default: break;
		}
	}
	pp.tmPostOff = t.off;
	p.tmPostFlags = kTmOn | kTmSplit | (fam == TM_R2C ? kTmCplx : 0u);
}

static int finish_pass(const PassBuild& bIn, Arena& ar, PassPlan& pp) {
	PassBuild b = bIn;
	size_t mixconvTabOff = (size_t)-1;
	// strided-tile passes of power-of-two length run on the hand-specialised column kernel
	if (b.allowFast && b.fastKernel == KERNEL_GENERIC && b.colIn && b.L >= 2 && b.L <= 1024 && (b.L & (b.L - 1)) == 0 && b.preOp == OP_NONE
	    && b.midOp == OP_NONE && (b.postOp == OP_NONE || b.postOp == OP_TWIDDLE_4STEP) && !b.realIn && !b.realOut && !b.forceT) {
		int variant, bits[4], tc, thr;
		// buffer addressing of the fast kernels: a tile must span less than 2 GiB on both sides
		const uint64_t esz = b.dp ? 16 : 8;
		const HostDim d0 = b.dims.empty() ? HostDim{1, 0, 0} : b.dims[0];
		const uint64_t spanIn = (b.L * (uint64_t)std::llabs(b.inStrideJ) + 64 * (uint64_t)std::llabs(d0.inStride)) * esz;
		const uint64_t spanOut = (b.L * (uint64_t)std::llabs(b.outStrideJ) + 64 * (uint64_t)std::llabs(d0.outStride)) * esz;
		if (pow2_col_lookup(ilog2(b.L), b.dp, &variant, bits, &tc, &thr)) {
			static const bool forceBig = getenv("VKFFT_MI355X_FORCE_BIGSPAN") != nullptr; // (tests: the 64-bit form on small problems)
			b.bigSpan = forceBig || !(spanIn < 0x7FFFFF00ull && spanOut < 0x7FFFFF00ull); // (then the kernel's 64-bit form: DESIGN 7, the z axis of 1024^3 on one GPU)
			b.fastKernel = KERNEL_POW2_COL; b.fastVariant = variant; b.fastThreads = thr; b.forceT = (uint32_t)tc;
			b.radices.clear();
			for (int k = 0; k < 4; k++) if (bits[k]) b.radices.push_back(1u << bits[k]);
		}
	}
	// real transforms (fused pre/post map) and strided C2C of curated lengths: op-FFT family
	const bool fusedBluestein = b.preOp == OP_BLUESTEIN_PRE && b.midOp == OP_BLUESTEIN_MID && b.postOp == OP_BLUESTEIN_POST && !b.colIn && b.auxOff2ForPre == (size_t)-1;
	const bool padMask = b.padInN || b.padOutN; // (the interpreter, pow2_row / pow2_col, the fused Bluestein kernels (rdMask / wrMask, round 4) and the op-FFT kernels honour the masks: Io64 / Io32 / explicit)
	if (fusedBluestein && b.allowOp && b.fastKernel == KERNEL_GENERIC && (b.L & (b.L - 1)) == 0 && !b.forceT && b.radices.empty()) { // power-of-two padded length: register-resident persistent kernel
		int variant, bits[4], fpw, thr;
		const HostDim d0 = b.dims.empty() ? HostDim{1, 0, 0} : b.dims[0];
		const uint64_t span = (b.L + 64 * (uint64_t)std::max<int64_t>(std::llabs(d0.inStride), std::llabs(d0.outStride))) * (b.dp ? 16 : 8);
		if (span < 0x7FFFFF00ull && b.opN * 2 <= b.L && pow2_blue_lookup(ilog2(b.L), b.dp, &variant, bits, &fpw, &thr)) {
			b.fastKernel = KERNEL_POW2_BLUE; b.fastVariant = variant; b.fastThreads = thr; b.forceT = (uint32_t)fpw;
			b.radices.clear();
			for (int k = 0; k < 4; k++) if (bits[k]) b.radices.push_back(1u << bits[k]);
		}
	}
	// ... and its column form for strided axes (tiles of neighbouring columns, one pass)
	const bool fusedBluesteinCol = b.preOp == OP_BLUESTEIN_PRE && b.midOp == OP_BLUESTEIN_MID && b.postOp == OP_BLUESTEIN_POST && b.colIn && b.colOut && b.auxOff2ForPre == (size_t)-1;
	if (fusedBluesteinCol && b.allowOp && b.fastKernel == KERNEL_GENERIC && (b.L & (b.L - 1)) == 0 && !b.forceT && b.radices.empty()) {
		int variant, bits[4], tc, thr;
		const uint64_t esz = b.dp ? 16 : 8;
		const HostDim d0 = b.dims.empty() ? HostDim{1, 0, 0} : b.dims[0];
		const uint64_t spanIn = (b.L * (uint64_t)std::llabs(b.inStrideJ) + 64 * (uint64_t)std::llabs(d0.inStride)) * esz;
		const uint64_t spanOut = (b.L * (uint64_t)std::llabs(b.outStrideJ) + 64 * (uint64_t)std::llabs(d0.outStride)) * esz;
		if (spanIn < 0x7FFFFF00ull && spanOut < 0x7FFFFF00ull && b.opN * 2 <= b.L && pow2_col_blue_lookup(ilog2(b.L), b.dp, 5, &variant, bits, &tc, &thr)) {
			b.fastKernel = KERNEL_POW2_COL_BLUE; b.fastVariant = variant; b.fastThreads = thr; b.forceT = (uint32_t)tc;
			b.radices.clear();
			for (int k = 0; k < 4; k++) if (bits[k]) b.radices.push_back(1u << bits[k]);
		}
	}
	// (column tile in, per-column contiguous run out = the first Four-Step pass: the transposed-store variant)
	const bool transOut = b.colIn && !b.colOut && b.preOp == OP_NONE && (b.postOp == OP_NONE || b.postOp == OP_TWIDDLE_4STEP) && b.outStrideJ == 1 && !b.realIn && !b.realOut;
	// (op-FFT and zero padding: only the maps that go through Io32::ia / oa element by element — C2C columns, the R2C / C2R forms)
	const bool opMaskOK = !padMask || ((b.preOp == OP_NONE || b.preOp == OP_C2R_EVEN_PRE || b.preOp == OP_R2C_FULL || b.preOp == OP_C2R_FULL) &&
	                                 (b.postOp == OP_NONE || b.postOp == OP_R2C_EVEN_POST || b.postOp == OP_R2C_FULL || b.postOp == OP_C2R_FULL));
	// (short real rows: the instance transform between the interpreter's maps moves the tile as one contiguous run; measured against the fused-map
	// kernels, whose threads read a short row 4 or 8 bytes at a time — VKFFT_MI355X_MIXED_OPS_MAX = longest complex length that prefers it)
	uint64_t opsCplxLen = 0; // complex length of a pass that runs an instance transform between the generic maps (0: not that form)
	bool preferMixedOps = false;
	{
		// measured (tools/tune_mixed_ops.py, profiles/r03_short_real_rows_fused_maps_vs_instance_between_maps.jsonl): complex lengths 8 and 16 (R2C / DCT of 16 and
		// 32 reals) run 1.1-5x faster between the maps; from 20 on the fused-map kernels win (only the powers of two were measured: the others keep their fused-map kernel)
		const uint64_t lim = getenv("VKFFT_MI355X_MIXED_OPS_MAX") ? (uint64_t)atoll(getenv("VKFFT_MI355X_MIXED_OPS_MAX")) : 16;
		int v, r5[5], f, t;
		// (round 4: the other lengths of that range too — their fused-map instances are one thread per row, radix-10 / 14 / 15 butterflies fed by loads a row pitch
		// apart per lane: DCT-IV of 20 and 30 reals ran at 0.19 / 0.14x the reference, DCT-II of 28 at 0.22x, profiles/r04_dct4_rows_5_400_*)
		const bool rowOp = !b.colIn && !b.colOut && !padMask && b.inStrideJ == 1 && b.outStrideJ == 1 && (b.preOp != OP_NONE || b.postOp != OP_NONE);
		preferMixedOps = lim && b.L >= 8 && b.L <= lim && rowOp && mixed_row_lookup(b.L, b.dp, &v, r5, &f, &t);
		// two real rows per transform exist only between the generic maps (PassParams::pairRows): VKFFT_MI355X_PAIR_PREFER=1 sends the pairable families there
		// even where a fused-map instance exists (measurement switch)
		const int pairPrefer = getenv("VKFFT_MI355X_PAIR_PREFER") ? atoi(getenv("VKFFT_MI355X_PAIR_PREFER")) : kPairPreferDefault;
		const bool r2cFam = b.preOp == OP_R2C_FULL || b.preOp == OP_C2R_FULL || b.postOp == OP_R2C_FULL || b.postOp == OP_C2R_FULL;
		if (!preferMixedOps && (pairPrefer == 1 || (pairPrefer == 2 && !r2cFam)) && rowOp && pairable_family(b.preOp, b.postOp, b.L, b.opN) && mixed_row_lookup(b.L, b.dp, &v, r5, &f, &t)) preferMixedOps = true;
	}
	if (b.allowOp && opMaskOK && !preferMixedOps && b.fastKernel == KERNEL_GENERIC && !b.forceT && b.midOp == OP_NONE && (b.colIn == b.colOut || transOut) && b.radices.empty()
	    && !(b.preOp == OP_NONE && b.postOp == OP_NONE && !b.colIn)) {
		const uint64_t ib = (b.realIn ? 1 : 2) * (b.dp ? 8 : 4), ob = (b.realOut ? 1 : 2) * (b.dp ? 8 : 4);
		const HostDim d0 = b.dims.empty() ? HostDim{1, 0, 0} : b.dims[0];
		const uint64_t maxPos = std::max<uint64_t>(std::max<uint64_t>(b.L, b.inLen), std::max<uint64_t>(b.outLen, b.opN)) * 2 + 4;
		const uint64_t spanIn = (maxPos * (uint64_t)std::llabs(b.inStrideJ) + 64 * (uint64_t)std::llabs(d0.inStride)) * ib;
		const uint64_t spanOut = (maxPos * (uint64_t)std::llabs(b.outStrideJ) + 64 * (uint64_t)std::llabs(d0.outStride)) * ob;
		int variant, rad5[5], fpw, thr;
		if (spanIn < 0x7FFFFF00ull && spanOut < 0x7FFFFF00ull && opfft_lookup(b.L, b.dp, b.colIn, transOut, b.preOp, b.postOp, &variant, rad5, &fpw, &thr)) {
			b.fastKernel = KERNEL_OPFFT; b.fastVariant = variant; b.fastThreads = thr; b.forceT = (uint32_t)fpw;
			for (int k = 0; k < 5; k++) if (rad5[k] > 1) b.radices.push_back((uint32_t)rad5[k]);
		}
	}
	// real transforms on unit-stride rows whose complex length has a mixed-radix instance but no fused-map kernel above: the ahead-of-time transform
	// between the interpreter's gather-load and gather-store (mixed_row_kernel OPS = 1) instead of the interpreter's own stages
	{
		auto realOp = [](uint32_t op) {
			return op == OP_NONE || (op >= OP_R2C_EVEN_POST && op <= OP_DST4_POST) || (op >= OP_DCT2H_PRE && op <= OP_DST3H_POST) || op == OP_DCT1H_PRE || op == OP_DCT1H_POST;
		};
		if (b.allowOp && b.fastKernel == KERNEL_GENERIC && !b.forceT && b.midOp == OP_NONE && !b.colIn && !b.colOut && b.radices.empty() && !b.preNat && !b.postNat &&
		    b.auxOff2ForPre == (size_t)-1 && (b.preOp != OP_NONE || b.postOp != OP_NONE) && realOp(b.preOp) && realOp(b.postOp) && b.inStrideJ == 1 && b.outStrideJ == 1 &&
		    !getenv("VKFFT_MI355X_NO_MIXED_OPS")) {
			int variant, rad5[5], fpw, thr;
			uint64_t len = 0;
			opsCplxLen = b.L;
			// (the maps inside the stages address the rows of a tile — up to 2 * FPW <= 128 of them — with 32-bit byte offsets from the tile's base: a row pitch that
			// takes the tile past 2 GiB leaves the pass to the interpreter, as on the complex paths)
			const HostDim d0o = b.dims.empty() ? HostDim{1, 0, 0} : b.dims[0];
			const bool spanOK = (2 * b.L + 128 * (uint64_t)std::max<int64_t>(std::llabs(d0o.inStride), std::llabs(d0o.outStride))) * (b.dp ? 16 : 8) < 0x7FFFFF00ull;
			if (!spanOK) { /* interpreter */ }
			else if (mixed_row_lookup(b.L, b.dp, &variant, rad5, &fpw, &thr)) {
				const int fo = mixed_row_ops_fpw(variant); // (the form between the maps may take fewer rows per workgroup: kernel_mixed.h mixed_ops_fpw)
				if (fo > 0 && fo != fpw) { thr = thr / fpw * fo; fpw = fo; }
				b.fastKernel = KERNEL_MIXED_ROW; b.fastVariant = variant; b.fastThreads = thr; b.forceT = (uint32_t)fpw;
				for (int k = 0; k < 5; k++) if (rad5[k] > 1) b.radices.push_back((uint32_t)rad5[k]);
			} else if (!padMask && tm_family(b.preOp, b.postOp, b.L, b.opN) != TM_NONE && !getenv("VKFFT_MI355X_NO_TMAPS") && [&]() {
				// the complex length is M * P with a Rader prime and a served cofactor: mixrad_kernel between the table-driven maps (kernel_mixrad.h, kernel_tmaps.h)
				MixradChoice mr;
				if (!mixrad_choose(b.L, b.dp, true, mr)) return false;
				if ((2 * (uint64_t)mr.rows + 2) * (uint64_t)std::max<int64_t>(std::llabs(d0o.inStride), std::llabs(d0o.outStride)) * (b.dp ? 16 : 8) >= 0x7FFFFF00ull) return false;
				const uint64_t N = b.L;
				if (!b.inLen) b.inLen = (uint32_t)N;
				if (!b.outLen) b.outLen = (uint32_t)N;
				if (!b.blueN) b.blueN = (uint32_t)N;
				size_t bhatOff;
				make_mixrad_tables(mr.P, mr.M, b.dp, ar, mixconvTabOff, bhatOff);
				b.auxOff2ForPre = bhatOff;
				b.L = mr.len; b.raderM = (uint32_t)mr.M; b.raderA = mr.A; b.raderAligned = mr.aligned;
				b.fastKernel = KERNEL_MIXCONV; b.fastVariant = mr.variant; b.fastThreads = mr.threads;
				b.forceT = mr.rows;
				for (int k = 0; k < 5; k++) if (mr.rad[k] > 1) b.radices.push_back((uint32_t)mr.rad[k]);
				return true;
			}()) {
			} else if (b.L >= 37 && is_prime_u(b.L) && mixconv_lookup(true, false, b.L, b.dp, &variant, &len, rad5, &fpw, &thr)) {
				// the complex length is a Rader prime: mixconv_kernel OPS = 1 (transform length P - 1; kernel spectrum through aux3)
				const uint64_t P = b.L;
				if (!b.inLen) b.inLen = (uint32_t)P;
				if (!b.outLen) b.outLen = (uint32_t)P;
				if (!b.blueN) b.blueN = (uint32_t)P; // (the maps tell the half-length forms from the full-length ones by the complex length)
				size_t bhatOff;
				make_mixconv_rader_tables(P, b.dp, ar, mixconvTabOff, bhatOff);
				b.auxOff2ForPre = bhatOff;
				b.L = len;
				b.fastKernel = KERNEL_MIXCONV; b.fastVariant = variant; b.fastThreads = thr; b.forceT = (uint32_t)fpw;
				for (int k = 0; k < 5; k++) if (rad5[k] > 1) b.radices.push_back((uint32_t)rad5[k]);
			}
		}
	}
	PassParams& p = pp.prm;
	memset(&p, 0, sizeof(p));
	const bool dp = b.dp;
	const size_t es = dp ? 16 : 8;
	p.L = (uint32_t)b.L;
	std::vector<uint32_t> rad = b.radices.empty() ? radix_schedule(b.L) : b.radices;
	if (b.L == 1) rad.clear();
	if (rad.size() > (size_t)kMaxStages) return 3002;
	p.nStages = (uint32_t)rad.size();
	// stage twiddles (+ the tables of an FFT-Rader stage)
	uint64_t lutElems = 0, S = 1;
	uint32_t raderP = 0;
	// radices above 16 other than 32: Rader stages of the interpreter — except inside the hand-specialised kernels, whose
	// schedules may hold the composite butterfly 25 = 5*5
	auto plainRadix = [&](uint32_t R) { return R <= 16 || R == 32 || b.fastKernel != KERNEL_GENERIC; };
	for (uint32_t R : rad) {
		if (S > 1) lutElems += (uint64_t)(R - 1) * S;
		if (!plainRadix(R) && R <= b.raderDirectMax) lutElems += R;
		if (!plainRadix(R) && R > b.raderDirectMax) raderP = R;
		S *= R;
	}
	std::vector<uint32_t> subRad;
	uint64_t subLutElems = 0;
	if (raderP) {
		if (!rader_fft_ok(raderP)) return 3002;
		subRad = radix_schedule(raderP - 1);
		if (subRad.size() > 8) return 3002;
		uint64_t S2 = 1;
		for (uint32_t R : subRad) { if (S2 > 1) subLutElems += (uint64_t)(R - 1) * S2; S2 *= R; }
		lutElems += subLutElems + (raderP - 1);
	}
	size_t lutOff = ar.alloc((lutElems + 1) * es);
	uint64_t cur = 0; S = 1;
	for (size_t si = 0; si < rad.size(); si++) {
		uint32_t R = rad[si];
		StageDesc& sd = p.st[si];
		sd.radix = R; sd.S = (uint32_t)S; sd.lutOff = (uint32_t)cur;
		sd.kind = plainRadix(R) ? 0 : (R <= b.raderDirectMax ? 1 : 2);
		if (S > 1) {
			for (uint32_t i = 1; i < R; i++)
				for (uint64_t s = 0; s < S; s++) ar.putc(lutOff, cur + (uint64_t)(i - 1) * S + s, unit_root((uint64_t)i * s, (uint64_t)R * S), dp);
			cur += (uint64_t)(R - 1) * S;
		}
		if (sd.kind == 1) {
			sd.aux0 = (uint32_t)cur;
			for (uint32_t m = 0; m < R; m++) ar.putc(lutOff, cur + m, unit_root(m, R), dp);
			cur += R;
		}
		p.divNb[si] = make_fastdiv((uint32_t)(b.L / R));
		p.divS[si] = make_fastdiv((uint32_t)S);
		S *= R;
	}
	if (raderP) {
		RaderDesc& rd = p.rd;
		const uint32_t P = raderP, P1 = P - 1;
		rd.P = P; rd.nSub = (uint32_t)subRad.size();
		rd.subLutOff = (uint32_t)cur;
		uint64_t c2 = 0, S2 = 1;
		for (size_t si = 0; si < subRad.size(); si++) {
			uint32_t R = subRad[si];
			rd.sub[si].radix = R; rd.sub[si].S = (uint32_t)S2; rd.sub[si].lutOff = (uint32_t)c2; rd.sub[si].kind = 0;
			if (S2 > 1) {
				for (uint32_t i = 1; i < R; i++)
					for (uint64_t s = 0; s < S2; s++) ar.putc(lutOff, cur + c2 + (uint64_t)(i - 1) * S2 + s, unit_root((uint64_t)i * s, (uint64_t)R * S2), dp);
				c2 += (uint64_t)(R - 1) * S2;
			}
			rd.divSubNb[si] = make_fastdiv(P1 / R);
			rd.divSubS[si] = make_fastdiv((uint32_t)S2);
			S2 *= R;
		}
		cur += subLutElems;
		// generator tables and FFT of the convolution kernel b_q = exp(-2 pi i g^-q / P)  (RecursiveFFTGenerators.h:1021-1048)
		const uint64_t g = primitive_root(P), gi = powmod(g, P - 2, P);
		size_t tabOff = ar.alloc(2 * (size_t)P1 * sizeof(uint32_t));
		uint32_t* tab = (uint32_t*)(ar.b.data() + tabOff);
		std::vector<cld> bk(P1);
		uint64_t gp = 1, gm = 1;
		for (uint32_t q = 0; q < P1; q++) { tab[q] = (uint32_t)gp; tab[P1 + q] = (uint32_t)gm; bk[q] = unit_root(gm, P); gp = gp * g % P; gm = gm * gi % P; }
		host_fft(bk);
		rd.bhatOff = (uint32_t)cur;
		for (uint32_t m = 0; m < P1; m++) ar.putc(lutOff, cur + m, bk[m] / (ld)P1, dp);
		cur += P1;
		rd.gpowOff = 0; rd.ginvOff = P1;
		pp.raderOff = tabOff;
	}
	pp.lutOff = lutOff;
	pp.auxOff = b.auxOff; pp.aux2Off = b.aux2Off; pp.aux3Off = b.auxOff2ForPre;

	std::vector<HostDim> dims = b.dims;
	if (dims.empty()) dims.push_back({1, 0, 0});
	// dims[0] stays the tiled dim; collapse the rest among themselves
	if (!b.noCollapse) {
		std::vector<HostDim> rest(dims.begin() + 1, dims.end());
		collapse_dims(rest);
		HostDim d0 = dims[0];
		// merge rest[0] into d0 when contiguous (typical: rows of a batch)
		if (!rest.empty() && rest[0].count > 1 && (int64_t)d0.count * d0.inStride == rest[0].inStride
		    && (int64_t)d0.count * d0.outStride == rest[0].outStride && d0.count * rest[0].count < (1ull << 31)) {
			d0.count *= rest[0].count;
			rest.erase(rest.begin());
		}
		dims.clear(); dims.push_back(d0);
		for (auto& r : rest) if (r.count > 1) dims.push_back(r);
	}
	while (dims.size() < 3) dims.push_back({1, 0, 0});
	pp.hostLoop.clear();
	while (dims.size() > 3) { pp.hostLoop.push_back(dims.back()); dims.pop_back(); }
	for (int i = 0; i < 3; i++) { p.dim[i].count = (uint32_t)dims[i].count; p.dim[i].inStride = dims[i].inStride; p.dim[i].outStride = dims[i].outStride; }
	p.inStrideJ = b.inStrideJ; p.outStrideJ = b.outStrideJ;

	// workgroup tile
	const bool anyCol = b.colIn || b.colOut;
	uint32_t T;
	// (a register-direct single-buffer interpreter was tried and dropped: the monolithic kernel spills)
	const uint64_t ldsPerSub = 2 * (b.L + b.L / 16 + 2) * es; // both ping-pong buffers
	if (b.forceT) T = b.forceT;
	else if (anyCol) {
		T = dp ? 16 : 32; // 256-byte segments
		while (T > 1 && (uint64_t)(T + 1) * ldsPerSub > (b.maxLds * 7) / 10) T >>= 1; // leave room for two workgroups per CU when possible
		while (T > 1 && (uint64_t)(T + 1) * ldsPerSub > b.maxLds) T >>= 1;
	} else {
		// unit-stride rows: enough sub-FFTs for >= ~2048 points per workgroup
		T = 1;
		while (T < 64 && (uint64_t)T * b.L < 2048 && (uint64_t)(2 * T + 1) * ldsPerSub <= 64 * 1024) T <<= 1;
	}
	if (b.fastKernel == KERNEL_GENERIC) {
		while (T > 1 && T / 2 >= dims[0].count) T >>= 1;
		if ((uint64_t)(T == 1 ? 1 : T + 1) * ldsPerSub > b.maxLds) return 3002;
	}
	p.T = T; p.logT = ilog2(T);
	p.Tp = T == 1 ? 1 : T + 1;
	p.padShift = T >= 16 ? 31 : 4;
	p.colMode = b.colIn ? 1 : 0; p.colModeOut = b.colOut ? 1 : 0;
	p.swapIn = b.swapIn; p.swapOut = b.swapOut;
	p.preOp = b.preOp; p.midOp = b.midOp; p.postOp = b.postOp;
	p.bluesteinSwapIn = b.bsSwapIn; p.bluesteinSwapOut = b.bsSwapOut;
	p.inLen = b.inLen ? b.inLen : (uint32_t)b.L;
	p.outLen = b.outLen ? b.outLen : (uint32_t)b.L;
	p.opN = b.opN; p.blueN = b.blueN;
	p.preNat = b.preNat ? 1 : 0; p.postNat = b.postNat ? 1 : 0; p.natDimMask = b.natDimMask; p.natOutLen = b.natOutLen;
	p.opStrideJ = b.opStrideJ; p.opStride0 = b.opStride0; p.opStride1 = b.opStride1;
	p.fsN = (uint32_t)b.fsN;
	p.fsColDiv = make_fastdiv(b.fsColDiv);
	p.fsColFromDim1 = b.fsColFromDim1 ? 1 : 0;
	p.scale = b.scale;
	p.padInL = b.padInL; p.padInN = b.padInN; p.padOutL = b.padOutL; p.padOutN = b.padOutN;
	p.bigSpan = b.bigSpan ? 1u : 0u;
	p.divL = make_fastdiv((uint32_t)b.L);
	p.divOutLen = make_fastdiv(p.outLen);
	p.raderM = b.raderM; p.raderA = b.raderA; p.raderAligned = b.raderAligned; // mixrad_kernel: rows of raderM * (L + 1) points
	const uint64_t padded = p.padShift >= 31 ? b.L : b.L + (b.L >> p.padShift);
	p.ldsElems = (uint32_t)((padded + 1) * p.Tp);
	p.tilesPerG0 = (uint32_t)((dims[0].count + T - 1) / T);
	const bool noPairs = getenv("VKFFT_MI355X_NO_ROW_PAIRS") != nullptr;
	if (opsCplxLen && b.fastKernel != KERNEL_GENERIC && (!noPairs || b.raderM)) { // (the Rader-stage kernel has no other maps than the tables: kernel_mixrad.h)
		// two real rows per complex transform (the reference's mergeSequencesR2C, vkFFT_SharedMemory.h:40): the families whose pre-map is a real sequence
		// (post-map through the even / odd split) or whose result is real (kernel_generic.h ops_rows_in / ops_rows_out); the tile holds 2 T rows
		// table-driven maps (kernel_tmaps.h): the instance transform of kernel_mixed.h; with them the families whose operation the generic maps only know at run time pair too
		const TmFamily tmf = ((((b.fastKernel == KERNEL_MIXED_ROW || (b.fastKernel == KERNEL_MIXCONV && !b.raderM)) && b.fastThreads / (int)T >= tmaps_min_tpf()) || (b.fastKernel == KERNEL_MIXCONV && b.raderM)) && !b.padInN && !b.padOutN && !getenv("VKFFT_MI355X_NO_TMAPS")) ? tm_family(b.preOp, b.postOp, opsCplxLen, b.opN) : TM_NONE;
		// (the one family that does not pair — even DCT / DST-IV on its half-length complex form — with fewer than four threads per row: the staged tile of ONE row per
		// thread measured 1.8x slower than the generic loops: DCT-IV of 20 and 30 reals, profiles/r05_dct4_rows_reference_every_length_step3_*; and the maps address
		// the 2 T real rows of a tile with 32-bit byte offsets from the tile's base)
		const uint64_t tmReach = (2 * (uint64_t)T + 2) * (uint64_t)std::max<int64_t>(std::llabs(dims[0].inStride), std::llabs(dims[0].outStride)) * (dp ? 16 : 8);
		const bool tmOn = tmf != TM_NONE && !(tmf == TM_R2R4_EVEN && !b.raderM && b.fastThreads / (int)T < 4) && tmReach < 0x7FFFFF00ull;
		if (b.raderM && !tmOn) return 3002;
		if (!noPairs && (pairable_family(b.preOp, b.postOp, opsCplxLen, b.opN) || (tmOn && tm_family_pairs(tmf)))) {
			p.pairRows = 1;
			p.tilesPerG0 = (uint32_t)((dims[0].count + 2 * (uint64_t)T - 1) / (2 * (uint64_t)T));
		}
		if (tmOn) {
			const bool dstFam = b.preOp == OP_DST2_PRE || b.preOp == OP_DST3_PRE || b.preOp == OP_DST4_PRE || b.preOp == OP_DST1_PRE;
			build_tmaps(tmf, dstFam, opsCplxLen, b.opN, dp, b.scale, b.fastKernel == KERNEL_MIXED_ROW && b.fastThreads / (int)T >= 8, ar, pp);
		}
	}
	if (b.fastKernel == KERNEL_POW2_BLUE_R2R && !b.padInN && !b.padOutN && !getenv("VKFFT_MI355X_NO_ROW_PAIRS") && !getenv("VKFFT_MI355X_NO_BLUE_PAIRS")) {
		// two real rows per fused Bluestein transform (kernel_blue_r2r.h): the families whose embedding sequence is real, or whose result is
		auto fam = [&](uint32_t a, uint32_t c) { return b.preOp == a && b.postOp == c; };
		const bool same4 = (fam(OP_DCT4_PRE, OP_DCT4_POST) || fam(OP_DST4_PRE, OP_DST4_POST)) && b.blueN == b.opN;
		if (fam(OP_R2C_FULL, OP_R2C_FULL) || fam(OP_C2R_FULL, OP_C2R_FULL) || fam(OP_DCT2_PRE, OP_DCT2_POST) || fam(OP_DST2_PRE, OP_DST2_POST) || fam(OP_DCT3_PRE, OP_DCT3_POST) ||
		    fam(OP_DST3_PRE, OP_DST3_POST) || fam(OP_DCT1_PRE, OP_DCT1_POST) || fam(OP_DST1_PRE, OP_DST1_POST) || same4) {
			p.pairRows = 1;
			p.tilesPerG0 = (uint32_t)((dims[0].count + 2 * (uint64_t)T - 1) / (2 * (uint64_t)T));
		}
	}
	const bool mergeable = (b.fastKernel == KERNEL_MIXCONV && b.colIn) ||
	                       ((b.fastKernel == KERNEL_OPFFT || (b.fastKernel == KERNEL_POW2_COL && !b.bigSpan)) && b.colIn && b.colOut && b.preOp == OP_NONE && b.midOp == OP_NONE &&
	                        b.postOp == OP_NONE && !b.realIn && !b.realOut);
	if (mergeable && dims[1].count > 1 && dims[0].count % T != 0 && dims[0].count < 4 * T && (uint64_t)dims[0].count * dims[1].count < (1ull << 31)) {
		// column tiles of kernel_mixconv.h and of the plain strided C2C kernels (kernel_opfft.h, pow2_col_kernel) over dim[0] x dim[1]: a companion axis that is not a multiple of the tile width (37 columns, tiles of 32) would
		// leave the last tile of every plane mostly empty
		const uint64_t esz = dp ? 16 : 8;
		const uint64_t reach = ((uint64_t)T / dims[0].count + 2) * (uint64_t)std::max<int64_t>(std::llabs(dims[1].inStride), std::llabs(dims[1].outStride))
		                     + (b.L + 1) * (uint64_t)std::max<int64_t>(std::llabs(b.inStrideJ), std::llabs(b.outStrideJ)) + dims[0].count;
		if (reach * esz < 0x7FFFFF00ull && dims[1].inStride > 0 && dims[1].outStride > 0) {
			p.colMerge = 1;
			p.tilesPerG0 = (uint32_t)(((uint64_t)dims[0].count * dims[1].count + T - 1) / T);
		}
	}
	if (p.rd.P) { p.rd.tailElems = (uint32_t)((b.L / p.rd.P) * T + 1); p.rd.divU = make_fastdiv((uint32_t)((b.L / p.rd.P) * T)); }
	pp.ldsBytes = (2 * (size_t)p.ldsElems + p.rd.tailElems) * es;
	if (pp.ldsBytes > b.maxLds && b.fastKernel == KERNEL_GENERIC) return 3002;
	p.inElemBytes = (uint32_t)((b.realIn ? 1 : 2) * (dp ? 8 : 4));
	p.outElemBytes = (uint32_t)((b.realOut ? 1 : 2) * (dp ? 8 : 4));
	// threads: about one radix-8 butterfly per thread per stage
	uint64_t work = (uint64_t)T * b.L / 8;
	uint32_t thr = 64;
	while (thr < work && thr < 1024) thr <<= 1;
	if (pp.ldsBytes > 48 * 1024 && thr < 256) thr = 256;
	pp.threads = thr;
	pp.dp = dp;
	pp.kernel = KERNEL_GENERIC;
	pp.inRole = b.inRole; pp.outRole = b.outRole;
	pp.inOffset = b.inOffset; pp.outOffset = b.outOffset;
	pp.inElemBytes = (int)((b.realIn ? 1 : 2) * (dp ? 8 : 4));
	pp.outElemBytes = (int)((b.realOut ? 1 : 2) * (dp ? 8 : 4));
	pp.label = b.label;
	if (b.fastKernel != KERNEL_GENERIC) { // hand-specialised kernel: fixed tile, static LDS
		pp.kernel = b.fastKernel; pp.variant = b.fastVariant; pp.threads = (uint32_t)b.fastThreads; pp.ldsBytes = 0;
	}

	// Four-Step two-level table
	if (b.postOp == OP_TWIDDLE_4STEP || b.preOp == OP_FOURSTEP_INV_PRE || b.preOp == OP_FOURSTEP_INV_COL_PRE) {
		uint32_t lo = (ceil_log2(b.fsN) + 1) / 2;
		uint64_t nlo = 1ull << lo, nhi = (b.fsN + nlo - 1) / nlo;
		size_t off = ar.alloc((nlo + nhi) * es);
		for (uint64_t i = 0; i < nlo; i++) ar.putc(off, i, unit_root(i, b.fsN), dp);
		for (uint64_t i = 0; i < nhi; i++) ar.putc(off, nlo + i, unit_root(i * nlo, b.fsN), dp);
		p.fsLoBits = lo;
		pp.auxOff = off;
	}
	if (mixconvTabOff != (size_t)-1) pp.raderOff = mixconvTabOff;
	return 0;
}

// ---- single-pass capacity ------------------------------------------------------------------------------
static uint64_t max_row_len(bool dp, uint64_t maxLds) { // unit-stride, T = 1
	uint64_t es = dp ? 16 : 8;
	uint64_t L = 1;
	while (2 * (2 * L + 2 * L / 16 + 2) * es <= maxLds) L *= 2;
	return L; // fp32: 8192, fp64: 4096 at 160 KiB
}
static uint64_t max_col_len(bool dp, uint64_t maxLds, uint32_t T) {
	uint64_t es = dp ? 16 : 8;
	const uint64_t c = maxLds / ((uint64_t)(T + 1) * 2 * es);
	return c > 2 ? c - 2 : 0;
}

static bool is_supported_len(uint64_t L, uint32_t directMax) {
	std::vector<uint32_t> pf = factorize_radices(L, nullptr);
	uint32_t fftPrime = 0;
	for (uint32_t p : pf) if (p > 13 && p > directMax) {
		if (!rader_fft_ok(p)) return false;
		if (fftPrime && fftPrime != p) return false; // one FFT-Rader prime per pass
		fftPrime = p;
	}
	return true;
}

// choose N = n[0]*n[1](*n[2]); n[0] is the pass that runs over the largest stride (executed first).
// Preference order: two passes with wide tiles and two workgroups per CU, ..., three passes last.
// a factor length the hand-specialised column kernels serve in all three Four-Step roles (first pass with transposed store,
// middle pass with twiddle, last pass)
static bool fast_col_len(uint64_t L, bool dp) {
	int v, r5[5], bits[4], f, t;
	if ((L & (L - 1)) == 0 && L >= 16 && L <= 1024) return pow2_col_lookup(ilog2(L), dp, &v, bits, &f, &t);
	return opfft_lookup(L, dp, true, true, OP_NONE, OP_TWIDDLE_4STEP, &v, r5, &f, &t) && opfft_lookup(L, dp, true, false, OP_NONE, OP_TWIDDLE_4STEP, &v, r5, &f, &t)
	       && opfft_lookup(L, dp, true, false, OP_NONE, OP_NONE, &v, r5, &f, &t);
}

static bool choose_split(uint64_t N, bool dp, uint64_t maxLds, uint32_t directMax, bool fast, std::vector<uint64_t>& out) {
	const bool fastP2 = fast && (N & (N - 1)) == 0; // fast column kernels: single LDS buffer, L <= 1024
	if (fast && !fastP2) { // non-power-of-two: prefer a split whose factors all run on hand-specialised column kernels
		std::vector<uint64_t> fd;
		for (uint64_t d = 2; d * d <= N; d++) if (N % d == 0) { fd.push_back(d); if (d != N / d) fd.push_back(N / d); }
		std::sort(fd.begin(), fd.end());
		uint64_t best = 0;
		for (uint64_t d : fd) if (d <= N / d && fast_col_len(d, dp) && fast_col_len(N / d, dp)) best = d; // most balanced pair
		if (best) { out = {N / best, best}; return true; }
		uint64_t ba = 0, bb = 0; double bestCost = 1e300;
		for (uint64_t a : fd) if (fast_col_len(a, dp)) for (uint64_t b2 : fd) if ((N / a) % b2 == 0 && N / a / b2 > 1 && fast_col_len(b2, dp) && fast_col_len(N / a / b2, dp)) {
			const double m = (double)std::max(a, std::max(b2, N / a / b2));
			if (m < bestCost) { bestCost = m; ba = a; bb = b2; }
		}
		if (ba) { out = {ba, bb, N / ba / bb}; return true; }
	}
	auto capOf = [&](uint32_t T, uint64_t budget) { uint64_t c = max_col_len(dp, fastP2 ? 2 * budget : budget, T); return fastP2 ? std::min<uint64_t>(c, 1024) : c; };
	struct Opt { uint32_t T; uint64_t budget; };
	const uint32_t Tw = dp ? 16 : 32;
	const Opt opts[5] = {{Tw, maxLds / 2}, {Tw / 2, maxLds / 2}, {Tw, maxLds}, {Tw / 2, maxLds}, {Tw / 4, maxLds}};
	std::vector<uint64_t> divs;
	for (uint64_t d = 1; d * d <= N; d++) if (N % d == 0) { divs.push_back(d); if (d != N / d) divs.push_back(N / d); }
	std::sort(divs.begin(), divs.end());
	auto ok = [&](uint64_t L, uint64_t cap) { return L >= 2 && L <= cap && is_supported_len(L, directMax); };
	for (const Opt& o : opts) {
		const uint64_t cap = capOf(o.T, o.budget);
		uint64_t best = 0;
		for (uint64_t d : divs) if (d <= N / d && ok(d, cap) && ok(N / d, cap)) best = d;
		if (best) { out = {N / best, best}; return true; }
	}
	for (const Opt& o : opts) {
		const uint64_t cap = capOf(o.T, o.budget);
		double bestCost = 1e300; uint64_t ba = 0, bb = 0;
		for (uint64_t a : divs) if (ok(a, cap)) for (uint64_t b2 : divs) if ((N / a) % b2 == 0 && ok(b2, cap) && ok(N / a / b2, cap)) {
			uint64_t c = N / a / b2;
			double m = (double)std::max(a, std::max(b2, c));
			if (m < bestCost) { bestCost = m; ba = a; bb = b2; }
		}
		if (ba) { out = {ba, bb, N / ba / bb}; return true; }
	}
	return false;
}

struct AxisJob {
	uint64_t N = 0;                 // logical length of this axis
	int64_t inStrideJ = 1, outStrideJ = 1;
	std::vector<HostDim> others;    // every other dimension (count, inStride, outStride); others[0] should be the unit-stride one for strided axes
	bool dp = false;
	bool inverse = false;
	double scale = 1.0;
	int inRole = ROLE_BUFFER, outRole = ROLE_BUFFER;
	int axisIndex = 0;
	uint32_t padInL = 0, padInN = 0, padOutL = 0, padOutN = 0; // zero padding along this axis: elements not read / not written
};
constexpr int kPadUnsupported = -77; // (internal) no kernel of the plan this axis needs can skip the padded range: the caller falls back

static uint32_t direct_max(const TransformDesc& d) { return (uint32_t)std::min<uint64_t>(d.raderMultMax, 61); }

// host-side mixed-radix FFT in long double (for Bluestein's FFT(chirp)); roots = exp(-2 pi i k / M) for the top-level M
static void host_fft_rec(std::vector<cld>& a, const std::vector<cld>& roots) {
	const size_t n = a.size(), M = roots.size();
	if (n <= 1) return;
	size_t p = 0;
	for (size_t q : {2, 3, 5, 7, 11, 13}) if (n % q == 0) { p = q; break; }
	const size_t rs = M / n; // roots[(k * rs) % M] = exp(-2 pi i k / n)
	if (!p) {
		std::vector<cld> r(n);
		for (size_t k = 0; k < n; k++) { cld s = 0; for (size_t j = 0; j < n; j++) s += a[j] * roots[((j * k) % n) * rs]; r[k] = s; }
		a = r; return;
	}
	const size_t m = n / p;
	std::vector<std::vector<cld>> sub(p, std::vector<cld>(m));
	for (size_t j = 0; j < n; j++) sub[j % p][j / p] = a[j];
	for (auto& s2 : sub) host_fft_rec(s2, roots);
	for (size_t k = 0; k < n; k++) {
		cld acc = sub[0][k % m];
		for (size_t r = 1; r < p; r++) acc += sub[r][k % m] * roots[((r * k) % n) * rs];
		a[k] = acc;
	}
}
// exp(-2 pi i k / M) for k = 0..count-1 from a two-level table: 2*sqrt(M) sincosl calls and one complex multiply per entry
// (|error| ~ 1e-19: below the long double rounding of the direct evaluation's argument reduction for large M)
static void fill_roots(std::vector<cld>& roots, uint64_t M) {
	uint32_t sh = 0; while ((1ull << (2 * sh)) < M) sh++;
	const uint64_t nlo = 1ull << sh, nhi = (M + nlo - 1) >> sh;
	std::vector<cld> lo(nlo), hi(nhi);
	for (uint64_t i = 0; i < nlo; i++) lo[i] = unit_root(i, M);
	for (uint64_t i = 0; i < nhi; i++) hi[i] = unit_root(i << sh, M);
	roots.resize(M);
	for (uint64_t k = 0; k < M; k++) roots[k] = (k & (nlo - 1)) ? hi[k >> sh] * lo[k & (nlo - 1)] : hi[k >> sh];
}
// power-of-two lengths (every padded Bluestein length of the fast paths): iterative radix-2, no allocation per level
static void host_fft_pow2(std::vector<cld>& a, const std::vector<cld>& roots) {
	const size_t n = a.size();
	for (size_t i = 1, j = 0; i < n; i++) {
		size_t bit = n >> 1;
		for (; j & bit; bit >>= 1) j ^= bit;
		j ^= bit;
		if (i < j) std::swap(a[i], a[j]);
	}
	for (size_t len = 2; len <= n; len <<= 1) {
		const size_t half = len >> 1, step = n / len;
		for (size_t i = 0; i < n; i += len)
			for (size_t j = 0; j < half; j++) {
				const cld u = a[i + j], v = a[i + j + half] * roots[j * step];
				a[i + j] = u + v; a[i + j + half] = u - v;
			}
	}
}
static void host_fft(std::vector<cld>& a) {
	const size_t M = a.size();
	std::vector<cld> roots;
	fill_roots(roots, M);
	if ((M & (M - 1)) == 0) host_fft_pow2(a, roots);
	else host_fft_rec(a, roots);
}

static uint64_t next_smooth(uint64_t n, int maxPrime) {
	for (uint64_t m = n;; m++) {
		uint64_t r = m;
		for (int p : {2, 3, 5, 7, 11, 13}) { if (p > maxPrime) break; while (r % p == 0) r /= p; }
		if (r == 1) return m;
	}
}

// ---- multi-pass (Four-Step) emitter ---------------------------------------------------------------------
// N = n0*M (M = n1*n2 for three passes).  Pass A reads the input as an n0 x M matrix column-wise and stores every
// column as a contiguous run into scratch region T1; the last pass reads T1 column-wise again and writes natural
// order to the output (reference: vkFFT_4step.h:31, vkFFT_ReadWrite.h:1405-1476; execution order RunApp.h:114).
struct MultiPassIO {
	std::vector<HostDim> othersIn, othersOut;  // the other dims (batch, ...) with the strides of the input / output side
	int inRole = ROLE_BUFFER, outRole = ROLE_BUFFER;
	int64_t inOffset = 0, outOffset = 0, t1Offset = 0; // element offsets (T1 lives in ROLE_TEMP)
	bool swapIn = false, swapOut = false;
	double scale = 1.0;
	// hooks for Bluestein: operation on the very first load / the very last store, indexed by the natural position
	uint32_t firstPre = OP_NONE, lastPost = OP_NONE;
	size_t firstAux = (size_t)-1, lastAux = (size_t)-1, lastAux2 = (size_t)-1;
	bool bsSwapIn = false, bsSwapOut = false;
	uint32_t opN = 0;
	// real transforms through a multi-pass complex FFT of the embedding length: the first load / last store apply the real
	// transform's pre / post map to the ROW, addressed by the natural FFT index (PassParams::preNat / postNat)
	bool natural = false, firstRealIn = false, lastRealOut = false;
	uint32_t natOutLen = 0, blueN = 0;
	// zero padding by the natural index, ranges aligned with the split (plan_c2c_axis): read mask of the first pass, write mask of the last
	uint64_t padInL = 0, padInN = 0, padOutL = 0, padOutN = 0;
};

static int emit_multipass(const PassBuild& proto, uint64_t N, const std::vector<uint64_t>& sp, const MultiPassIO& io, Arena& ar, std::vector<PassPlan>& passes) {
	const size_t nOthers = io.othersIn.size();
	std::vector<HostDim> othersTmp = io.othersIn;
	{ int64_t run = (int64_t)N; for (auto& o : othersTmp) { o.inStride = o.outStride = run; run *= (int64_t)o.count; } }
	auto dimsFor = [&](const HostDim& tiled, std::vector<HostDim> extra, int inKind, int outKind) {
		// inKind/outKind: 0 = caller's layout (input / output side), 1 = scratch layout
		std::vector<HostDim> r; r.push_back(tiled);
		for (auto& e : extra) r.push_back(e);
		for (size_t i = 0; i < nOthers; i++) {
			HostDim h; h.count = io.othersIn[i].count;
			h.inStride = inKind == 0 ? io.othersIn[i].inStride : othersTmp[i].inStride;
			h.outStride = outKind == 0 ? io.othersOut[i].outStride : othersTmp[i].outStride;
			r.push_back(h);
		}
		return r;
	};
	const uint64_t n0 = sp[0];
	const uint64_t M = N / n0;
	// pass A: x[n0][M] columns, FFT over n0, twiddle, store transposed Y^T[m][k0] into T1
	PassBuild a = proto;
	a.padInL = (uint32_t)(io.padInL / M); a.padInN = (uint32_t)(io.padInN / M); a.padOutL = a.padOutN = 0; // (aligned ranges only: plan_c2c_axis)
	a.L = n0; a.inStrideJ = (int64_t)M; a.outStrideJ = 1;
	a.colIn = true; a.colOut = false;
	a.dims = dimsFor({M, 1, (int64_t)n0}, {}, 0, 1);
	a.swapIn = io.swapIn; a.postOp = OP_TWIDDLE_4STEP; a.fsN = N; a.fsColDiv = 1;
	a.inRole = io.inRole; a.inOffset = io.inOffset; a.outRole = ROLE_TEMP; a.outOffset = io.t1Offset;
	a.label = "4step-A"; a.noCollapse = true;
	if (io.firstPre != OP_NONE) {
		a.preOp = io.firstPre; a.auxOff2ForPre = io.firstAux; a.bsSwapIn = io.bsSwapIn; a.opN = io.opN;
		a.opStrideJ = (uint32_t)M; a.opStride0 = 1; a.opStride1 = 0;
		if (io.natural) { a.preNat = true; a.natDimMask = 1; a.realIn = io.firstRealIn; a.blueN = io.blueN; }
	}
	PassPlan pa; int r = finish_pass(a, ar, pa); if (r) return r;
	passes.push_back(pa);
	if (sp.size() == 2) {
		// pass B: T1[m][k0]: FFT over m (stride n0) for T adjacent k0; X[k0 + n0*k1]
		PassBuild c = proto;
		c.padInL = c.padInN = 0; c.padOutL = (uint32_t)(io.padOutL / n0); c.padOutN = (uint32_t)(io.padOutN / n0);
		c.L = M; c.inStrideJ = (int64_t)n0; c.outStrideJ = (int64_t)n0;
		c.colIn = c.colOut = true;
		c.dims = dimsFor({n0, 1, 1}, {}, 1, 0);
		c.swapOut = io.swapOut; c.scale = io.scale;
		c.inRole = ROLE_TEMP; c.inOffset = io.t1Offset; c.outRole = io.outRole; c.outOffset = io.outOffset;
		c.label = "4step-B"; c.noCollapse = true;
		if (io.lastPost != OP_NONE) {
			c.postOp = io.lastPost; c.auxOff = io.lastAux; c.aux2Off = io.lastAux2; c.bsSwapOut = io.bsSwapOut; c.opN = io.opN;
			c.opStrideJ = (uint32_t)n0; c.opStride0 = 1; c.opStride1 = 0;
			if (io.natural) { c.postNat = true; c.natDimMask = 1; c.natOutLen = io.natOutLen; c.realOut = io.lastRealOut; c.blueN = io.blueN; }
		}
		PassPlan pb; r = finish_pass(c, ar, pb); if (r) return r;
		passes.push_back(pb);
	} else {
		const uint64_t n1 = sp[1], n2 = sp[2];
		// pass B (in place on T1): layout [m = i1*n2 + i2][k0]; FFT over i1 (stride n2*n0); tiled dim c = i2*n0 + k0
		PassBuild bb = proto;
		bb.padInL = bb.padInN = bb.padOutL = bb.padOutN = 0;
		bb.L = n1; bb.inStrideJ = bb.outStrideJ = (int64_t)(n2 * n0);
		bb.colIn = bb.colOut = true;
		bb.dims = dimsFor({n2 * n0, 1, 1}, {}, 1, 1);
		bb.postOp = OP_TWIDDLE_4STEP; bb.fsN = M; bb.fsColDiv = (uint32_t)n0;
		bb.inRole = bb.outRole = ROLE_TEMP; bb.inOffset = bb.outOffset = io.t1Offset;
		bb.label = "4step3-B"; bb.noCollapse = true;
		PassPlan pb; r = finish_pass(bb, ar, pb); if (r) return r;
		// pass C: T1 [k1][i2][k0]: FFT over i2 (stride n0); out X[k0 + n0*(k1 + n1*k2)]
		PassBuild c = proto;
		c.padInL = c.padInN = 0; c.padOutL = (uint32_t)(io.padOutL / (n0 * n1)); c.padOutN = (uint32_t)(io.padOutN / (n0 * n1));
		c.L = n2; c.inStrideJ = (int64_t)n0; c.outStrideJ = (int64_t)(n1 * n0);
		c.colIn = c.colOut = true;
		c.dims = dimsFor({n0, 1, 1}, {{n1, (int64_t)(n2 * n0), (int64_t)n0}}, 1, 0);
		c.swapOut = io.swapOut; c.scale = io.scale;
		c.inRole = ROLE_TEMP; c.inOffset = io.t1Offset; c.outRole = io.outRole; c.outOffset = io.outOffset;
		c.label = "4step3-C"; c.noCollapse = true;
		if (io.lastPost != OP_NONE) {
			c.postOp = io.lastPost; c.auxOff = io.lastAux; c.aux2Off = io.lastAux2; c.bsSwapOut = io.bsSwapOut; c.opN = io.opN;
			c.opStrideJ = (uint32_t)(n0 * n1); c.opStride0 = 1; c.opStride1 = (uint32_t)n0;
			if (io.natural) { c.postNat = true; c.natDimMask = 3; c.natOutLen = io.natOutLen; c.realOut = io.lastRealOut; c.blueN = io.blueN; }
		}
		PassPlan pc; r = finish_pass(c, ar, pc); if (r) return r;
		passes.push_back(pb); passes.push_back(pc);
	}
	return 0;
}

// stage twiddles of a register-resident power-of-two schedule, laid out [(i-1)*S + s] per stage (kernel_pow2_core.h, Pow2Sched::lutOff)
static size_t build_pow2_stage_lut(Arena& ar, const int bits[4], bool dp) {
	uint64_t elems = 0, S = 1;
	for (int k = 0; k < 4; k++) if (bits[k]) { const uint64_t R = 1ull << bits[k]; if (S > 1) elems += (R - 1) * S; S *= R; }
	const size_t off = ar.alloc((elems + 1) * (dp ? 16 : 8));
	uint64_t cur = 0; S = 1;
	for (int k = 0; k < 4; k++) if (bits[k]) {
		const uint64_t R = 1ull << bits[k];
		if (S > 1) {
			for (uint64_t i = 1; i < R; i++) for (uint64_t sidx = 0; sidx < S; sidx++) ar.putc(off, cur + (i - 1) * S + sidx, unit_root(i * sidx, R * S), dp);
			cur += (R - 1) * S;
		}
		S *= R;
	}
	return off;
}

// Fused Four-Step (kernel_pow2_fused.h): a two-factor power-of-two transform on a unit-stride axis as ONE persistent launch whose
// intermediate lives in a small scratch ring (Infinity-Cache resident) instead of a full-size temp buffer.  The other dimensions must
// collapse into one batch progression on both sides.  Returns false when the plan does not qualify (the caller emits separate passes).
static bool emit_fused(const TransformDesc& d, const AxisJob& j, Arena& ar, DirectionPlan& out, std::vector<PassPlan>& passes) {
	if (!d.fused || d.disableFastKernels || (j.N & (j.N - 1)) != 0 || j.inStrideJ != 1 || j.outStrideJ != 1) return false;
	const bool dp = j.dp;
	const uint64_t es = dp ? 16 : 8;
	int variant, la, lb, bitsA[4], bitsB[4], tca, tcb, thr, wgPerCu;
	if (!pow2_fused_lookup(ilog2(j.N), dp, d.fusedMode, &variant, &la, &lb, bitsA, bitsB, &tca, &tcb, &thr, &wgPerCu)) return false;
	// one batch progression
	uint64_t batch = 1; int64_t inS = (int64_t)j.N, outS = (int64_t)j.N; bool first = true;
	for (const HostDim& o : j.others) {
		if (o.count <= 1) continue;
		if (first) { inS = o.inStride; outS = o.outStride; batch = o.count; first = false; }
		else { if (o.inStride != inS * (int64_t)batch || o.outStride != outS * (int64_t)batch) return false; batch *= o.count; }
	}
	if (inS < (int64_t)j.N || outS < (int64_t)j.N || batch >= (1ull << 31)) return false;
	const uint64_t n0 = 1ull << la, n1 = 1ull << lb;
	const uint64_t tileBytes = n0 * (uint64_t)tca * es, fftBytes = j.N * es;
	const uint32_t logTiles = ilog2(fftBytes / tileBytes);
	// chunk: about a MiB of transforms, dealt round-robin to Q queues (one per XCD when there are enough chunks).  Tickets of a queue
	// are handed out in order, so at any moment its Wq workgroups hold a window of about Wq consecutive tickets.  A wait is avoided
	// when the producer tiles of a dependency left that window before the consumer enters it: lag D = 1 + X slots between a chunk's A
	// and B tiles, ring NS = D + 1 + X slots before a slot is rewritten, X slots ~ margin * Wq tickets.
	const uint64_t chunkTarget = d.fusedChunkBytes ? d.fusedChunkBytes : (1ull << 20);
	uint32_t logG = 0;
	while ((fftBytes << (logG + 1)) <= chunkTarget && (1ull << (logG + 1)) <= batch) logG++;
	const uint64_t G = 1ull << logG;
	const uint64_t C = (batch + G - 1) / G;
	const uint64_t tpc = G << logTiles; // tickets per slot (= tiles per chunk and phase)
	// queues: one per XCD when the ring that goes with it still fits the Infinity Cache (256 MiB, shared with what streams through),
	// else one queue; as a last resort a shorter lag (some tiles will poll)
	// (2^22 fp32: a transform is 32 MiB, the budget decides between lag 3 / ring 6 and lag 4 / ring 8 — measured 2.73 against 3.04 TB/s with the tiles of two
	// halves, profiles/r05_fused_lag_ring_pairs.jsonl: with lag 3 a tile waits 5-10 k cycles per ticket for the previous tenant of its ring slot)
	const uint64_t ringBudget = fftBytes >= (32ull << 20) ? (256ull << 20) : (224ull << 20);
	const uint64_t wgs = 256ull * (uint64_t)(d.fusedWgPerCu ? d.fusedWgPerCu : wgPerCu);
	// measured (tools/tune_fused.py): completions are published up to a ticket late and the ticket rate rises with the speed of the
	// kernel, so the window is taken generously: 3 windows where two or more workgroups share a CU, 2 with one workgroup per CU
	// (round 4, pipelined form of 2^19 / 2^20: one workgroup per CU, but the A tile of the NEXT ticket is requested early: 3 windows measured +2.7 % over 2)
	uint64_t marginPct = d.fusedMarginPct ? d.fusedMarginPct : ((wgPerCu >= 2 || (!dp && j.N <= (1ull << 20))) ? 300 : 200);
	uint64_t Q = 1, X = 1, D = 1, NS = 1, Cq = C;
	auto shape = [&](uint64_t q, uint64_t pct) {
		Q = q; Cq = (C + Q - 1) / Q;
		X = ((wgs / Q) * pct / 100 + tpc - 1) / tpc;
		if (X < 1) X = 1;
		D = d.fusedLag ? d.fusedLag : 1 + X;
		NS = d.fusedRing ? d.fusedRing : D + 1 + X;
		if (NS <= D) NS = D + 1;
		if (NS > Cq) NS = Cq; // fewer chunks than ring slots: no slot is ever reused
		if (D > Cq) D = Cq;   // (then every A tile of the queue precedes its first B tile)
		if (NS < 1) NS = 1;
		return Q * NS * G * fftBytes;
	};
	if (d.fusedQueues) (void)shape(std::min<uint64_t>(std::min<uint64_t>(d.fusedQueues, kFusedMaxQueues), C), marginPct);
	else {
		const uint64_t q8 = C >= 4 * kFusedMaxQueues ? kFusedMaxQueues : 1;
		if (shape(q8, marginPct) > ringBudget && q8 > 1) (void)shape(1, marginPct);
		while (!d.fusedMarginPct && shape(Q, marginPct) > ringBudget && marginPct > 50) marginPct -= 25;
	}
	const uint64_t scratch = Q * NS * G * fftBytes;
	if (d.userTempBytes && scratch > d.userTempBytes) return false;
	if (((Cq + D) << (logG + logTiles)) >= (1ull << 31)) return false; // 32-bit tickets
	PassPlan pp;
	memset(&pp.prm, 0, sizeof(pp.prm));
	pp.prm.L = (uint32_t)std::min<uint64_t>(j.N, 0xffffffffu);
	pp.kernel = KERNEL_POW2_FUSED; pp.variant = variant; pp.threads = (uint32_t)thr; pp.dp = dp;
	pp.inRole = j.inRole; pp.outRole = j.outRole; pp.inElemBytes = pp.outElemBytes = (int)es;
	pp.label = "4step-fused";
	pp.lutOff = build_pow2_stage_lut(ar, bitsA, dp);
	pp.fusedLutBOff = build_pow2_stage_lut(ar, bitsB, dp);
	{ // two-level Four-Step table w_N^e = lo[e & mask] * hi[e >> bits]  (vkFFT_4step.h:31 computes the same factor per element)
		const uint32_t lo = (ceil_log2(j.N) + 1) / 2;
		const uint64_t nlo = 1ull << lo, nhi = (j.N + nlo - 1) / nlo;
		const size_t off = ar.alloc((nlo + nhi) * es);
		for (uint64_t i = 0; i < nlo; i++) ar.putc(off, i, unit_root(i, j.N), dp);
		for (uint64_t i = 0; i < nhi; i++) ar.putc(off, nlo + i, unit_root(i * nlo, j.N), dp);
		pp.auxOff = off; pp.fused.fsLoBits = lo;
	}
	if (!dp) { // row table of the packed-pair kernels (kernel_pow2_pk.h pk_fs_twiddle): (1, Re w_N^k, 0, Im w_N^k) for k < n0
		const size_t off = ar.alloc(n0 * 16);
		for (uint64_t k = 0; k < n0; k++) {
			const cld w = unit_root(k, j.N);
			ar.put<float>(off, 2 * k, cld(1.0L, std::real(w)));
			ar.put<float>(off, 2 * k + 1, cld(0.0L, std::imag(w)));
		}
		pp.fusedRowTabOff = off;
	}
	pp.fusedCtrOff = ar.alloc((kFusedCtrDone + 2 * C) * sizeof(uint32_t)); // zero in the host image; the kernel leaves it zeroed
	memset(ar.b.data() + pp.fusedCtrOff, 0, (kFusedCtrDone + 2 * C) * sizeof(uint32_t));
	FusedParams& f = pp.fused;
	f.inBatchStride = inS; f.outBatchStride = outS;
	f.n0 = (uint32_t)n0; f.n1 = (uint32_t)n1; f.batch = (uint32_t)batch;
	f.logG = logG; f.logTiles = logTiles; f.C = (uint32_t)C; f.NS = (uint32_t)NS; f.D = (uint32_t)D; f.Q = (uint32_t)Q;
	f.swapIn = f.swapOut = j.inverse ? 1 : 0; f.reverse = 0; f.scale = j.scale;
	pp.fusedWgPerCu = (int)d.fusedWgPerCu;
	passes.push_back(pp);
	out.uploadsPerAxis[j.axisIndex] = 2;
	out.axisSplit[j.axisIndex][0] = n1; out.axisSplit[j.axisIndex][1] = n0;
	out.tempBytes = std::max<uint64_t>(out.tempBytes, scratch);
	return true;
}

// stage twiddles of a compile-time mixed-radix schedule, laid out [(i-1)*S + s] per stage from the second on (mix_sched.h, MixSched::lutOff)
static size_t build_mix_stage_lut(Arena& ar, const int rad[5], bool dp) {
	uint64_t elems = 0, S = 1;
	for (int k = 0; k < 5; k++) if (rad[k] > 1) { if (S > 1) elems += (uint64_t)(rad[k] - 1) * S; S *= (uint64_t)rad[k]; }
	const size_t off = ar.alloc((elems + 1) * (dp ? 16 : 8));
	uint64_t cur = 0; S = 1;
	for (int k = 0; k < 5; k++) if (rad[k] > 1) {
		const uint64_t R = (uint64_t)rad[k];
		if (S > 1) {
			for (uint64_t i = 1; i < R; i++) for (uint64_t sidx = 0; sidx < S; sidx++) ar.putc(off, cur + (i - 1) * S + sidx, unit_root(i * sidx, R * S), dp);
			cur += (R - 1) * S;
		}
		S *= R;
	}
	return off;
}

// Fused Four-Step of a NON-power-of-two two-factor length (kernel_mix_fused.h): the same launch shape as emit_fused (chunks, queues, lag, ring), tiles of either
// phase that need not divide their factor.  One launch of `batch` transforms of M = n0 * n1 points, inS / outS elements apart on either side.
struct MixFusedShape { int variant, n0, n1, radA[5], radB[5], tca, tcb, thr, wgPerCu; };
static bool build_mix_fused_pass(const TransformDesc& d, bool dp, uint64_t M, const MixFusedShape& v, uint64_t batch, int64_t inS, int64_t outS, bool inverse, double scale,
                                 Arena& ar, PassPlan& pp, uint64_t& scratch) {
	const uint64_t es = dp ? 16 : 8;
	const uint64_t n0 = (uint64_t)v.n0, n1 = (uint64_t)v.n1;
	const uint64_t tilesA = (n1 + (uint64_t)v.tca - 1) / (uint64_t)v.tca, tilesB = (n0 + (uint64_t)v.tcb - 1) / (uint64_t)v.tcb, tiles = std::max(tilesA, tilesB);
	const uint64_t fftBytes = ((n0 + 15) & ~15ull) * n1 * es; // a transform's share of a ring slot: n1 columns at a pitch of n0 rounded up to 16 elements (kernel_mix_fused.h NAP)
	const uint64_t chunkTarget = d.fusedChunkBytes ? d.fusedChunkBytes : (1ull << 20);
	uint32_t logG = 0;
	while ((fftBytes << (logG + 1)) <= chunkTarget && (1ull << (logG + 1)) <= batch) logG++;
	const uint64_t G = 1ull << logG;
	const uint64_t C = (batch + G - 1) / G;
	const uint64_t tpc = tiles << logG;
	const uint64_t ringBudget = fftBytes >= (32ull << 20) ? (256ull << 20) : (224ull << 20);
	const uint64_t wgs = 256ull * (uint64_t)(d.fusedWgPerCu ? d.fusedWgPerCu : v.wgPerCu);
	uint64_t marginPct = d.fusedMarginPct ? d.fusedMarginPct : (v.wgPerCu >= 2 ? 300 : 200);
	uint64_t Q = 1, X = 1, D = 1, NS = 1, Cq = C;
	auto shape = [&](uint64_t q, uint64_t pct) {
		Q = q; Cq = (C + Q - 1) / Q;
		X = ((wgs / Q) * pct / 100 + tpc - 1) / tpc;
		if (X < 1) X = 1;
		D = d.fusedLag ? d.fusedLag : 1 + X;
		NS = d.fusedRing ? d.fusedRing : D + 1 + X;
		if (NS <= D) NS = D + 1;
		if (NS > Cq) NS = Cq;
		if (D > Cq) D = Cq;
		if (NS < 1) NS = 1;
		return Q * NS * G * fftBytes;
	};
	if (d.fusedQueues) (void)shape(std::min<uint64_t>(std::min<uint64_t>(d.fusedQueues, kFusedMaxQueues), C), marginPct);
	else {
		const uint64_t q8 = C >= 4 * kFusedMaxQueues ? kFusedMaxQueues : 1;
		if (shape(q8, marginPct) > ringBudget && q8 > 1) (void)shape(1, marginPct);
		while (!d.fusedMarginPct && shape(Q, marginPct) > ringBudget && marginPct > 50) marginPct -= 25;
	}
	scratch = Q * NS * G * fftBytes;
	if ((Cq + D) * tpc >= (1ull << 31)) return false; // 32-bit tickets
	memset(&pp.prm, 0, sizeof(pp.prm));
	pp.prm.L = (uint32_t)std::min<uint64_t>(M, 0xffffffffu);
	pp.kernel = KERNEL_MIX_FUSED; pp.variant = v.variant; pp.threads = (uint32_t)v.thr; pp.dp = dp;
	pp.inElemBytes = pp.outElemBytes = (int)es;
	pp.label = "4step-fused";
	pp.lutOff = build_mix_stage_lut(ar, v.radA, dp);
	pp.fusedLutBOff = build_mix_stage_lut(ar, v.radB, dp);
	{ // two-level Four-Step table w_M^e = lo[e & mask] * hi[e >> bits], e < M
		const uint32_t lo = (ceil_log2(M) + 1) / 2;
		const uint64_t nlo = 1ull << lo, nhi = (M + nlo - 1) / nlo;
		const size_t off = ar.alloc((nlo + nhi) * es);
		for (uint64_t i = 0; i < nlo; i++) ar.putc(off, i, unit_root(i, M), dp);
		for (uint64_t i = 0; i < nhi; i++) ar.putc(off, nlo + i, unit_root(i * nlo, M), dp);
		pp.auxOff = off; pp.fused.fsLoBits = lo;
	}
	pp.fusedCtrOff = ar.alloc((kFusedCtrDone + 2 * C) * sizeof(uint32_t)); // zero in the host image; the kernel leaves it zeroed
	memset(ar.b.data() + pp.fusedCtrOff, 0, (kFusedCtrDone + 2 * C) * sizeof(uint32_t));
	FusedParams& f = pp.fused;
	f.inBatchStride = inS; f.outBatchStride = outS;
	f.n0 = (uint32_t)n0; f.n1 = (uint32_t)n1; f.batch = (uint32_t)batch;
	f.logG = logG; f.logTiles = 0; f.tiles = (uint32_t)tiles; f.tpc = (uint32_t)tpc;
	f.C = (uint32_t)C; f.NS = (uint32_t)NS; f.D = (uint32_t)D; f.Q = (uint32_t)Q;
	f.swapIn = f.swapOut = inverse ? 1 : 0; f.reverse = 0; f.scale = scale;
	pp.fusedWgPerCu = (int)d.fusedWgPerCu;
	return true;
}
// the other dimensions of an axis job as ONE batch progression (what a fused launch can follow)
static bool one_batch_progression(const AxisJob& j, uint64_t& batch, int64_t& inS, int64_t& outS) {
	batch = 1; inS = (int64_t)j.N; outS = (int64_t)j.N; bool first = true;
	for (const HostDim& o : j.others) {
		if (o.count <= 1) continue;
		if (first) { inS = o.inStride; outS = o.outStride; batch = o.count; first = false; }
		else { if (o.inStride != inS * (int64_t)batch || o.outStride != outS * (int64_t)batch) return false; batch *= o.count; }
	}
	return inS >= (int64_t)j.N && outS >= (int64_t)j.N && batch < (1ull << 31);
}
// Returns false when no instance serves the length or the plan does not qualify (the caller emits separate passes).
static bool emit_mix_fused(const TransformDesc& d, const AxisJob& j, Arena& ar, DirectionPlan& out, std::vector<PassPlan>& passes) {
	if (!d.fused || d.disableFastKernels || (j.N & (j.N - 1)) == 0 || j.inStrideJ != 1 || j.outStrideJ != 1) return false;
	if (const char* e = getenv("VKFFT_MI355X_MIXFUSED")) { if (atoi(e) == 0) return false; }
	MixFusedShape v;
	if (!mix_fused_lookup(j.N, j.dp, &v.variant, &v.n0, &v.n1, v.radA, v.radB, &v.tca, &v.tcb, &v.thr, &v.wgPerCu)) return false;
	uint64_t batch; int64_t inS, outS;
	if (!one_batch_progression(j, batch, inS, outS)) return false;
	PassPlan pp; uint64_t scratch = 0;
	if (!build_mix_fused_pass(d, j.dp, j.N, v, batch, inS, outS, j.inverse, j.scale, ar, pp, scratch)) return false;
	if (d.userTempBytes && scratch > d.userTempBytes) return false;
	pp.inRole = j.inRole; pp.outRole = j.outRole;
	passes.push_back(pp);
	out.uploadsPerAxis[j.axisIndex] = 2;
	out.axisSplit[j.axisIndex][0] = (uint64_t)v.n1; out.axisSplit[j.axisIndex][1] = (uint64_t)v.n0;
	out.tempBytes = std::max<uint64_t>(out.tempBytes, scratch);
	return true;
}

static void make_bluestein_tables(uint64_t N, uint64_t M, bool dp, Arena& ar, size_t& chirpOff, size_t& bhatOff, bool oneBlock = false);
// Chirp-z transform of a length without a Rader or single-pass Bluestein form (unit stride, fp32) as TWO launches of the fused Four-Step kernel on a registered padded
// length M >= 2N - 1 (kernel_mix_fused.h, MixFusedOps): chirp and zero padding on the first launch's loads, FFT(chirp) / M on its stores into ROLE_TEMP2, the inverse
// transform with the second chirp on its stores and nothing stored beyond N.  Reference: vkFFT_Scheduler.h:2406-2578 (multi-upload Bluestein), vkFFT_Bluestein.h:32,201.
constexpr uint64_t kMixFusedBlueQuery = 1ull << 63; // (kernels_mixfused.hip: mix_fused_lookup(n | this) = the smallest chirp-z instance of n points or more)
static bool emit_mix_fused_blue(const TransformDesc& d, const AxisJob& j, Arena& ar, DirectionPlan& out, std::vector<PassPlan>& passes) {
	if (!d.fused || d.disableFastKernels || j.dp || j.inStrideJ != 1 || j.outStrideJ != 1 || d.forceBluesteinSize || d.fixMaxRadixBluestein) return false;
	if (j.inRole == ROLE_TEMP2 || j.outRole == ROLE_TEMP2) return false; // (an inner plan of a wrapped transform: its rows live where this plan keeps its spectrum)
	if (const char* e = getenv("VKFFT_MI355X_MIXFUSED")) { if (atoi(e) == 0) return false; }
	// Measured on the device (profiles/r06_chirp_z_two_fused_launches_vs_separate_passes.jsonl): correct, and SLOWER than the three / five separate passes on a power of
	// two it was built to replace (524309: 157 against 256 GB/s, 15319: 530 against 764) — two launches still move the padded sequence through memory four times,
	// and the instances with the hooks sit at the 128-register cap.  Off unless asked for; what would win is ONE launch with both intermediates in the ring.
	{ const char* e = getenv("VKFFT_MI355X_MIXFUSED_BLUE"); if (!e || atoi(e) == 0) return false; }
	const uint64_t N = j.N;
	MixFusedShape v;
	if (!mix_fused_lookup((2 * N - 1) | kMixFusedBlueQuery, false, &v.variant, &v.n0, &v.n1, v.radA, v.radB, &v.tca, &v.tcb, &v.thr, &v.wgPerCu)) return false;
	const uint64_t M = (uint64_t)v.n0 * (uint64_t)v.n1;
	uint64_t batch; int64_t inS, outS;
	if (!one_batch_progression(j, batch, inS, outS)) return false;
	if (batch * M >= (1ull << 40) || N >= (1ull << 31)) return false;
	size_t chirpOff, bhatOff;
	make_bluestein_tables(N, M, false, ar, chirpOff, bhatOff, false);
	PassPlan p1, p2; uint64_t s1 = 0, s2 = 0;
	if (!build_mix_fused_pass(d, false, M, v, batch, inS, (int64_t)M, false, 1.0, ar, p1, s1)) return false;
	if (!build_mix_fused_pass(d, false, M, v, batch, (int64_t)M, outS, true, j.scale, ar, p2, s2)) return false;
	const uint64_t scratch = std::max(s1, s2), mid = batch * M * 8;
	if (d.userTempBytes && ((scratch + 255ull) & ~255ull) + mid > d.userTempBytes) return false;
	// (the hooks travel in the pass descriptor's otherwise unused fields: launch_mix_fused)
	p1.inRole = j.inRole; p1.outRole = ROLE_TEMP2; p1.label = "bluestein2-1";
	p1.prm.preOp = OP_BLUESTEIN_PRE; p1.prm.postOp = OP_MUL_LUT; p1.prm.opN = (uint32_t)N; p1.prm.bluesteinSwapIn = j.inverse ? 1 : 0;
	p1.aux3Off = chirpOff; p1.aux2Off = bhatOff;
	p2.inRole = ROLE_TEMP2; p2.outRole = j.outRole; p2.label = "bluestein2-2";
	p2.prm.postOp = OP_BLUESTEIN_POST; p2.prm.opN = (uint32_t)N; p2.prm.bluesteinSwapOut = j.inverse ? 1 : 0;
	p2.aux3Off = chirpOff; p2.aux2Off = bhatOff;
	passes.push_back(p1); passes.push_back(p2);
	out.uploadsPerAxis[j.axisIndex] = 2;
	out.axisSplit[j.axisIndex][0] = (uint64_t)v.n1; out.axisSplit[j.axisIndex][1] = (uint64_t)v.n0;
	out.tempBytes = std::max<uint64_t>(out.tempBytes, scratch);
	out.temp2Bytes = std::max<uint64_t>(out.temp2Bytes, mid);
	return true;
}

// Four-Step along a NON-unit-stride axis (element stride W, a unit-stride dimension x of extent nx beside it):
// coalescing comes from x, so no transposition is needed; the decomposition index becomes an extra batch dim.
//   2 passes: A  FFT over i0 for every (m, x), twiddle w_N^(k0*m)   [in -> T1, same shape]
//             B  FFT over m  for every (k0, x), X[(k0 + n0*k1)]      [T1 -> out]
static int emit_multipass_strided(const PassBuild& proto, uint64_t N, const std::vector<uint64_t>& sp, int64_t strideIn, int64_t strideOut,
                                  const HostDim& xdim, const std::vector<HostDim>& rest, int inRole, int outRole, bool inverse, double scale,
                                  Arena& ar, std::vector<PassPlan>& passes, uint64_t& tempElems) {
	// scratch layout: dense [n][x] per sub-problem, sub-problems enumerated densely
	const int64_t W = (int64_t)xdim.count;
	std::vector<HostDim> restTmp = rest;
	{ int64_t run = (int64_t)N * W; for (auto& o : restTmp) { o.inStride = o.outStride = run; run *= (int64_t)o.count; } tempElems = (uint64_t)run; }
	auto withRest = [&](std::vector<HostDim> lead, int inKind, int outKind) {
		for (size_t i = 0; i < rest.size(); i++) {
			HostDim h; h.count = rest[i].count;
			h.inStride = inKind == 0 ? rest[i].inStride : restTmp[i].inStride;
			h.outStride = outKind == 0 ? rest[i].outStride : restTmp[i].outStride;
			lead.push_back(h);
		}
		return lead;
	};
	const uint64_t n0 = sp[0], M = N / n0;
	PassBuild a = proto;
	a.L = n0; a.inStrideJ = (int64_t)M * strideIn; a.outStrideJ = (int64_t)M * W;
	a.colIn = a.colOut = true;
	a.dims = withRest({{(uint64_t)W, xdim.inStride, 1}, {M, strideIn, W}}, 0, 1);
	a.swapIn = inverse; a.postOp = OP_TWIDDLE_4STEP; a.fsN = N; a.fsColFromDim1 = true;
	a.inRole = inRole; a.outRole = ROLE_TEMP; a.label = "4step-strided-A"; a.noCollapse = true;
	PassPlan pa; int r = finish_pass(a, ar, pa); if (r) return r;
	passes.push_back(pa);
	if (sp.size() == 2) {
		PassBuild c = proto;
		c.L = M; c.inStrideJ = W; c.outStrideJ = (int64_t)n0 * strideOut;
		c.colIn = c.colOut = true;
		c.dims = withRest({{(uint64_t)W, 1, xdim.outStride}, {n0, (int64_t)M * W, strideOut}}, 1, 0);
		c.swapOut = inverse; c.scale = scale;
		c.inRole = ROLE_TEMP; c.outRole = outRole; c.label = "4step-strided-B"; c.noCollapse = true;
		PassPlan pb; r = finish_pass(c, ar, pb); if (r) return r;
		passes.push_back(pb);
	} else {
		const uint64_t n1 = sp[1], n2 = sp[2];
		PassBuild bb = proto;
		bb.L = n1; bb.inStrideJ = bb.outStrideJ = (int64_t)n2 * W;
		bb.colIn = bb.colOut = true;
		bb.dims = withRest({{(uint64_t)W, 1, 1}, {n2, W, W}, {n0, (int64_t)M * W, (int64_t)M * W}}, 1, 1);
		bb.postOp = OP_TWIDDLE_4STEP; bb.fsN = M; bb.fsColFromDim1 = true;
		bb.inRole = bb.outRole = ROLE_TEMP; bb.label = "4step3-strided-B"; bb.noCollapse = true;
		PassPlan pb; r = finish_pass(bb, ar, pb); if (r) return r;
		PassBuild c = proto;
		c.L = n2; c.inStrideJ = W; c.outStrideJ = (int64_t)(n0 * n1) * strideOut;
		c.colIn = c.colOut = true;
		c.dims = withRest({{(uint64_t)W, 1, xdim.outStride}, {n1, (int64_t)n2 * W, (int64_t)n0 * strideOut}, {n0, (int64_t)M * W, strideOut}}, 1, 0);
		c.swapOut = inverse; c.scale = scale;
		c.inRole = ROLE_TEMP; c.outRole = outRole; c.label = "4step3-strided-C"; c.noCollapse = true;
		PassPlan pc; r = finish_pass(c, ar, pc); if (r) return r;
		passes.push_back(pb); passes.push_back(pc);
	}
	return 0;
}

// Bluestein tables of a length-N transform through padded length M: chirp[n] = exp(+i pi n^2 / N) (the kernels multiply by its
// conjugate; vkFFT_RecursiveFFTGenerators.h:139-148) and FFT_M of the wrapped chirp, scaled by 1/M.
static void make_bluestein_tables(uint64_t N, uint64_t M, bool dp, Arena& ar, size_t& chirpOff, size_t& bhatOff, bool oneBlock) {
	const size_t es = dp ? 16 : 8;
	if (oneBlock) { chirpOff = ar.alloc((N + M) * es); bhatOff = chirpOff + N * es; } // FFT(chirp) right behind the chirp: one pointer serves both
	else { chirpOff = ar.alloc(N * es); bhatOff = ar.alloc(M * es); }
	// FFT(chirp) is identical for the forward and the inverse plan of an application: keep the last one
	static std::mutex mtx; static uint64_t cN = 0, cM = 0; static std::vector<cld> cChirp, cBhat;
	std::lock_guard<std::mutex> lock(mtx);
	if (cN != N || cM != M) {
		cChirp.assign(N, cld(0, 0));
		std::vector<cld> bext(M, cld(0, 0));
		std::vector<cld> r2n; // exp(-2 pi i e / 2N), e < 2N
		fill_roots(r2n, 2 * N);
		for (uint64_t n = 0; n < N; n++) {
			unsigned __int128 sq = (unsigned __int128)n * n;
			uint64_t e = (uint64_t)(sq % (2 * N));
			cld c = std::conj(r2n[e]);
			cChirp[n] = c;
			bext[n] = c;
			if (n) bext[M - n] = c;
		}
		host_fft(bext);
		cBhat.swap(bext); cN = N; cM = M;
	}
	for (uint64_t n = 0; n < N; n++) ar.putc(chirpOff, n, cChirp[n], dp);
	for (uint64_t k = 0; k < M; k++) ar.putc(bhatOff, k, cBhat[k] / (ld)M, dp);
	// the cache exists for the second direction of the same application; long rows would pin (N + M) * 32 bytes of host memory
	// for the life of the process: keep it only while it is small
	if ((N + M) * sizeof(cld) > (64ull << 20)) { std::vector<cld>().swap(cChirp); std::vector<cld>().swap(cBhat); cN = cM = 0; }
}

// ---- C2C along one axis -------------------------------------------------------------------------------
static int plan_c2c_axis(const TransformDesc& d, const AxisJob& j, Arena& ar, DirectionPlan& out, std::vector<PassPlan>& passes) {
	const bool dp = j.dp;
	const uint32_t dmax = direct_max(d);
	const bool unit = j.inStrideJ == 1 && j.outStrideJ == 1;
	bool smoothOK = is_supported_len(j.N, dmax);
	const uint64_t rowCap = max_row_len(dp, d.maxLds);
	PassBuild b;
	b.dp = dp; b.maxLds = d.maxLds; b.raderDirectMax = dmax; b.allowFast = !d.disableFastKernels; b.allowOp = !d.disableFastKernels;
	b.inRole = j.inRole; b.outRole = j.outRole;
	out.axisSplit[j.axisIndex][0] = j.N;
	// zero padding along this axis: served by the single-pass kernels that can skip elements (finish_pass); everything else reports kPadUnsupported
	const bool padded = j.padInN || j.padOutN;
	b.padInL = j.padInL; b.padInN = j.padInN; b.padOutL = j.padOutL; b.padOutN = j.padOutN;
	// (round 4: the one-pass Bluestein / Rader kernels skip the padded range too, and so do plans of several passes when the range is aligned with their split)

	{ // a supported length that neither fits one pass nor splits into supported factors (large Rader primes) also goes to Bluestein
		const uint64_t cap1 = unit ? rowCap : max_col_len(dp, d.maxLds, 1);
		const bool fastRow = unit && !d.disableFastKernels && (j.N & (j.N - 1)) == 0 && j.N <= (dp ? 8192u : 16384u);
		std::vector<uint64_t> probe;
		if (smoothOK && j.N > cap1 && !fastRow && !choose_split(j.N, dp, d.maxLds, dmax, !d.disableFastKernels, probe)) smoothOK = false;
	}
	// lengths with a prime factor above 13 on unit-stride rows: the fused Bluestein kernel on a compile-time schedule of the
	// padded length beats the interpreter's Rader stages (measured), so it also takes the Rader-capable lengths
	// ... and the 13-smooth lengths between 1024 and 4096 that have no ahead-of-time mixed-radix instance (the interpreter runs them at
	// 1.1-1.3 TB/s, the fused chirp-z kernel at 1.5-2 TB/s)
	uint64_t fusedM = 0;
	bool smoothNoInstance = false;
	if (unit && !d.disableFastKernels && smooth13(j.N) && (j.N & (j.N - 1)) != 0 && j.N > 1024 && j.N <= 4096) {
		int v, r5[5], f, t;
		smoothNoInstance = !mixed_row_lookup(j.N, dp, &v, r5, &f, &t);
	}
	// lengths with a direct prime butterfly (17 .. 31: tools/gen_mixed_table.py DIRECT_PRIMES) among their radices stay on the radix kernels where an
	// instance exists (rows: every such length up to 4096; strided axes: the pure primes)
	bool nativeInstance = false;
	if (!d.disableFastKernels && !smooth13(j.N) && j.N <= 4096) {
		int v, r5[5], f, t;
		nativeInstance = unit ? mixed_row_lookup(j.N, dp, &v, r5, &f, &t) : opfft_lookup(j.N, dp, true, false, OP_NONE, OP_NONE, &v, r5, &f, &t);
	}
	if (unit && !nativeInstance && !d.disableFastKernels && !d.forceBluesteinSize && !d.fixMaxRadixBluestein && (!smooth13(j.N) || smoothNoInstance)) {
		const uint64_t rowPitch = j.others.empty() ? j.N : (uint64_t)std::max<int64_t>(std::llabs(j.others[0].inStride), std::llabs(j.others[0].outStride));
		uint64_t Mp = 64; while (Mp < 2 * j.N - 1) Mp *= 2; // measured: the power-of-two padded length wins even at 1.6x the {1,3,5}*2^k one
		int v, bits[4], fpw, thr;
		if ((rowPitch * 64 + j.N) * (dp ? 16 : 8) < 0x7FFFFF00ull && pow2_blue_lookup(ilog2(Mp), dp, &v, bits, &fpw, &thr)) fusedM = Mp;
	}
	if (!unit && !nativeInstance && !d.disableFastKernels && !d.forceBluesteinSize && !d.fixMaxRadixBluestein && !smooth13(j.N) && !j.others.empty()
	    && j.others[0].inStride == 1 && j.others[0].outStride == 1) {
		// strided axes of non-smooth length (prime x prime planes): the one-pass column Bluestein kernel on the power-of-two padded length beats
		// the interpreter's Rader / Bluestein stages by 3-6x (measured on the reference's sample-7 systems)
		uint64_t Mp = 64; while (Mp < 2 * j.N - 1) Mp *= 2;
		int v, bits[4], tc, thr;
		if (pow2_col_blue_lookup(ilog2(Mp), dp, 5, &v, bits, &tc, &thr)) fusedM = Mp;
	}
	// ... or the one-kernel cyclic convolution on a smooth transform length (kernel_mixconv.h), unit-stride rows and tiles of neighbouring columns of a
	// strided axis: Rader for a prime with 13-smooth p-1 (p-1 points, no padding), Bluestein on the smallest ladder length >= 2N-1.  Against the
	// register-resident power-of-two kernels a point of these costs kCostRader / kCostBlue times as much (an LDS round trip more per transform, table
	// look-ups through L2); VKFFT_MI355X_MIXCONV=0 turns the family off, =2 always prefers it (tests, tuning)
	// ... or, for a composite length M * P with ONE prime factor above 31 whose P - 1 is 13-smooth and a cofactor of at most 32: the Rader convolution as a
	// stage of the row (kernel_mixrad.h; the reference's Rader stage inside its radix kernels, vkFFT_Scheduler.h:1733-1873).  VKFFT_MI355X_MIXRAD=0: off
	if (unit && !padded && !nativeInstance && !d.disableFastKernels && !d.forceBluesteinSize && !d.fixMaxRadixBluestein && !smooth13(j.N)) {
		MixradChoice mr;
		const uint64_t rowPitch = j.others.empty() ? j.N : (uint64_t)std::max<int64_t>(std::llabs(j.others[0].inStride), std::llabs(j.others[0].outStride));
		// taken where it is faster than the fused Bluestein kernel on the next power of two M2 >= 2N - 1 (every served length forced either way on the device).  A point of
		// the row costs c points of the padded power-of-two transform; Bluestein's 8192-point rows leave one workgroup per CU: 1.5 per point
		uint64_t M2 = 64; while (M2 < 2 * j.N - 1) M2 *= 2;
		const bool radForced = getenv("VKFFT_MI355X_MIXRAD") && atoi(getenv("VKFFT_MI355X_MIXRAD")) == 2; // (tests: always)
		bool radTake = mixrad_choose(j.N, dp, false, mr) && (rowPitch * 64 + j.N) * (dp ? 16 : 8) < 0x7FFFFF00ull;
		if (radTake) {
			const double c = getenv("VKFFT_MI355X_MIXRAD_COST") ? atof(getenv("VKFFT_MI355X_MIXRAD_COST")) : mr.cost;
			radTake = radForced || c * (double)j.N < (double)M2 * (M2 >= 8192 ? 1.5 : 1.0);
		}
		if (radTake) {
			b.L = mr.len; b.inLen = b.outLen = (uint32_t)j.N; b.opN = (uint32_t)j.N;
			for (int k = 0; k < 5; k++) if (mr.rad[k] > 1) b.radices.push_back((uint32_t)mr.rad[k]);
			b.fastKernel = KERNEL_MIXCONV; b.fastVariant = mr.variant; b.fastThreads = mr.threads;
			b.forceT = mr.rows; // rows per workgroup
			b.raderM = (uint32_t)mr.M; b.raderA = mr.A; b.raderAligned = mr.aligned;
			b.bsSwapIn = b.bsSwapOut = j.inverse; b.scale = j.scale;
			b.inStrideJ = j.inStrideJ; b.outStrideJ = j.outStrideJ; b.dims = j.others;
			b.colIn = b.colOut = false;
			size_t tabOff, bhatOff;
			make_mixrad_tables(mr.P, mr.M, dp, ar, tabOff, bhatOff);
			b.aux2Off = bhatOff;
			b.label = "rader-stage";
			PassPlan pp; int r = finish_pass(b, ar, pp); if (r) return r;
			pp.raderOff = tabOff;
			passes.push_back(pp);
			out.uploadsPerAxis[j.axisIndex] = 1;
			return 0;
		}
	}
	struct { bool use = false, rader = false, col = false; int variant = -1; uint64_t len = 0; int rad[5] = {1, 1, 1, 1, 1}; int fpw = 0, thr = 0; } mc;
	const bool colTile = !unit && !j.others.empty() && j.others[0].inStride == 1 && j.others[0].outStride == 1;
	if ((unit || colTile) && !nativeInstance && !d.disableFastKernels && !d.forceBluesteinSize && !d.fixMaxRadixBluestein && (!smooth13(j.N) || (unit && smoothNoInstance))) {
		const int mode = getenv("VKFFT_MI355X_MIXCONV") ? atoi(getenv("VKFFT_MI355X_MIXCONV")) : 1; // (read per plan: tests switch it)
		// measured (tools/tune_mixconv.py, profiles/r03_mixconv_*): time per point relative to the power-of-two kernels
		const double kCostRader = getenv("VKFFT_MI355X_MIXCONV_COST_RADER") ? atof(getenv("VKFFT_MI355X_MIXCONV_COST_RADER")) : 1.9;
		const double kCostBlue = getenv("VKFFT_MI355X_MIXCONV_COST_BLUE") ? atof(getenv("VKFFT_MI355X_MIXCONV_COST_BLUE")) : 1.6;
		const double kPow2Big = 1.9; // the power-of-two kernel at its longest padded length (128 KiB per row, one workgroup per CU) costs that much more per point
		const uint64_t esz = dp ? 16 : 8;
		bool spanOK;
		if (unit) {
			const uint64_t rowPitch = j.others.empty() ? j.N : (uint64_t)std::max<int64_t>(std::llabs(j.others[0].inStride), std::llabs(j.others[0].outStride));
			spanOK = (rowPitch * 64 + j.N) * esz < 0x7FFFFF00ull;
		} else spanOK = (2 * j.N * (uint64_t)std::max<int64_t>(std::llabs(j.inStrideJ), std::llabs(j.outStrideJ)) + 64) * esz < 0x7FFFFF00ull;
		if (mode && spanOK) {
			double best = fusedM && mode < 2 ? (double)fusedM * (fusedM * esz >= (128ull << 10) ? kPow2Big : 1.0) : 1e300;
			int v, r5[5], f, t; uint64_t len;
			if (is_prime_u(j.N) && mixconv_lookup(true, !unit, j.N, dp, &v, &len, r5, &f, &t) && kCostRader * (double)len < best) {
				best = kCostRader * (double)len;
				mc.use = true; mc.rader = true; mc.variant = v; mc.len = len; mc.fpw = f; mc.thr = t; for (int k = 0; k < 5; k++) mc.rad[k] = r5[k];
			}
			if (mixconv_lookup(false, !unit, 2 * j.N - 1, dp, &v, &len, r5, &f, &t) && kCostBlue * (double)len < best) {
				best = kCostBlue * (double)len;
				mc.use = true; mc.rader = false; mc.variant = v; mc.len = len; mc.fpw = f; mc.thr = t; for (int k = 0; k < 5; k++) mc.rad[k] = r5[k];
			}
			mc.col = !unit;
		}
	}
	if (mc.use) {
		const uint64_t N = j.N;
		b.L = mc.len; b.inLen = b.outLen = (uint32_t)N; b.opN = (uint32_t)N;
		for (int k = 0; k < 5; k++) if (mc.rad[k] > 1) b.radices.push_back((uint32_t)mc.rad[k]);
		b.fastKernel = KERNEL_MIXCONV; b.fastVariant = mc.variant; b.fastThreads = mc.thr; b.forceT = (uint32_t)mc.fpw;
		b.bsSwapIn = b.bsSwapOut = j.inverse; b.scale = j.scale;
		b.inStrideJ = j.inStrideJ; b.outStrideJ = j.outStrideJ; b.dims = j.others;
		b.colIn = b.colOut = mc.col;
		size_t tabOff = (size_t)-1;
		if (mc.rader) {
			size_t bhatOff;
			make_mixconv_rader_tables(N, dp, ar, tabOff, bhatOff);
			b.aux2Off = bhatOff;
			b.label = "rader";
		} else {
			size_t chirpOff, bhatOff;
			make_bluestein_tables(N, mc.len, dp, ar, chirpOff, bhatOff);
			b.preOp = OP_BLUESTEIN_PRE; b.midOp = OP_BLUESTEIN_MID; b.postOp = OP_BLUESTEIN_POST;
			b.auxOff = chirpOff; b.aux2Off = bhatOff;
			b.label = "bluestein";
		}
		PassPlan pp; int r = finish_pass(b, ar, pp); if (r) return r;
		pp.raderOff = tabOff;
		passes.push_back(pp);
		out.uploadsPerAxis[j.axisIndex] = 1;
		return 0;
	}
	if (!unit && !padded && !fusedM && !nativeInstance && !d.disableFastKernels && !d.forceBluesteinSize && !d.fixMaxRadixBluestein && !smooth13(j.N) && !j.others.empty() && j.others.size() <= 3
	    && j.others[0].inStride == 1 && j.others[0].outStride == 1 && j.inStrideJ > 0 && j.outStrideJ > 0) {
		// a strided axis of non-smooth length beyond the reach of the column Bluestein kernel (padded length above 2048): transpose it against its
		// unit-stride companion into a dense scratch copy, run it there as unit-stride rows (the fused Bluestein row kernel) and transpose back.
		// Two extra passes at copy speed instead of an interpreter pass with two columns per workgroup (1087 x 1087: 0.23 -> 0.9 TB/s).
		uint64_t Mp = 64; while (Mp < 2 * j.N - 1) Mp *= 2;
		int v, bits[4], fpw, thr;
		const uint64_t C = j.others[0].count;
		uint64_t outer = 1; for (size_t i = 1; i < j.others.size(); i++) outer *= j.others[i].count;
		if (pow2_blue_lookup(ilog2(Mp), dp, &v, bits, &fpw, &thr) && C * j.N * outer < (1ull << 40)) {
			const size_t mark = passes.size();
			auto transpose = [&](bool back) {
				PassPlan t; memset(&t.prm, 0, sizeof(t.prm));
				PassParams& q = t.prm;
				// scratch layout: [outer dims][c][n] dense.  Forward copy: lanes along c on the read side (dim 0 = companion), along n on the write
				// side (J = the axis); the copy back swaps the names so that the read side again runs along the scratch rows.
				int64_t run = (int64_t)(C * j.N);
				HostDim o1 = {1, 0, 0}, o2 = {1, 0, 0};
				int64_t d1 = 0, d2 = 0;
				if (j.others.size() > 1) { o1 = j.others[1]; d1 = run; run *= (int64_t)o1.count; }
				if (j.others.size() > 2) { o2 = j.others[2]; d2 = run; }
				if (!back) {
					q.L = (uint32_t)j.N; q.inStrideJ = j.inStrideJ; q.outStrideJ = 1;
					q.dim[0] = {(uint32_t)C, 1, (int64_t)j.N};
					q.dim[1] = {(uint32_t)o1.count, o1.inStride, d1}; q.dim[2] = {(uint32_t)o2.count, o2.inStride, d2};
					t.inRole = j.inRole; t.outRole = ROLE_TEMP2;
				} else {
					q.L = (uint32_t)C; q.inStrideJ = (int64_t)j.N; q.outStrideJ = 1;
					q.dim[0] = {(uint32_t)j.N, 1, j.outStrideJ};
					q.dim[1] = {(uint32_t)o1.count, d1, o1.outStride}; q.dim[2] = {(uint32_t)o2.count, d2, o2.outStride};
					t.inRole = ROLE_TEMP2; t.outRole = j.outRole;
				}
				q.tilesPerG0 = 1; // (launch_pass skips empty passes by this count; the kernel derives its own grid)
				t.kernel = KERNEL_TRANSPOSE; t.dp = dp; t.threads = 256; t.inElemBytes = t.outElemBytes = (int)(dp ? 16 : 8);
				t.label = back ? "transpose-back" : "transpose";
				passes.push_back(t);
			};
			transpose(false);
			AxisJob rj;
			rj.N = j.N; rj.inStrideJ = rj.outStrideJ = 1; rj.dp = dp; rj.inverse = j.inverse; rj.scale = j.scale;
			// the dense copy lives in ROLE_TEMP2: the row plan is free to use ROLE_TEMP for a scratch of its own
			rj.inRole = rj.outRole = ROLE_TEMP2; rj.axisIndex = j.axisIndex;
			rj.others.push_back({C * outer, (int64_t)j.N, (int64_t)j.N}); // the scratch rows are dense: one collapsed batch dimension
			const int r = plan_c2c_axis(d, rj, ar, out, passes);
			if (r == 0) {
				transpose(true);
				out.temp2Bytes = std::max<uint64_t>(out.temp2Bytes, C * j.N * outer * (dp ? 16 : 8));
				out.uploadsPerAxis[j.axisIndex] = 1;
				return 0;
			}
			passes.resize(mark);
		}
	}
	if (!smoothOK || fusedM) {
		// Bluestein (chirp-z) through a padded smooth length M >= 2N-1
		if (padded && !fusedM) return kPadUnsupported; // (the interpreter's and the multi-pass Bluestein plans do not skip elements)
		const uint64_t N = j.N;
		uint64_t M = fusedM ? fusedM : d.forceBluesteinSize ? d.forceBluesteinSize : next_smooth(2 * N - 1, d.fixMaxRadixBluestein ? d.fixMaxRadixBluestein : 7);
		uint64_t cap = unit ? rowCap : max_col_len(dp, d.maxLds, 1);
		if (unit && !fusedM && !padded && 2 * N - 1 > cap && emit_mix_fused_blue(d, j, ar, out, passes)) return 0; // longer than one pass holds: two fused launches
		if (unit && !fusedM && !d.disableFastKernels && !d.forceBluesteinSize && !d.fixMaxRadixBluestein) {
			// multi-pass rows: a power-of-two padded length runs as three passes on the column kernels (below)
			uint64_t Mp = 1; while (Mp < 2 * N - 1) Mp *= 2;
			if (Mp > cap && Mp * (dp ? 16 : 8) <= (1ull << 30)) M = Mp; // 32-bit byte offsets inside one padded row (buffer addressing)
		}
		std::vector<uint64_t> spM;
		if (M > cap && !fusedM) { // (the fused kernel holds its padded row in up to 139 KiB of LDS: one pass)
			if (!unit) return 3002;
			// M must split into column-kernel lengths; prefer a power of two when the smooth size does not split well
			if (!choose_split(M, dp, d.maxLds, dmax, !d.disableFastKernels, spM)) {
				if (d.forceBluesteinSize) return 3002;
				M = 1; while (M < 2 * N - 1) M *= 2;
				if (!choose_split(M, dp, d.maxLds, dmax, !d.disableFastKernels, spM)) return 3002;
			}
		}
		const size_t es = dp ? 16 : 8;
		size_t chirpOff, bhatOff;
		make_bluestein_tables(N, M, dp, ar, chirpOff, bhatOff);
		if (spM.size() == 2 && (M & (M - 1)) == 0 && !d.disableFastKernels) {
			// power-of-two padded length whose two factors are column-kernel lengths: three passes (kernel_pow2.h,
			// pow2_col_blue_kernel) — the middle one is FFT over m, * FFT(chirp), inverse FFT over m in registers
			const uint64_t n0 = spM[0], n1 = M / n0;
			int v1, v2, v3, bits1[4], bits2[4], bits3[4], tc1, tc2, tc3, th1, th2, th3;
			if (pow2_col_blue_lookup(ilog2(n0), dp, 1, &v1, bits1, &tc1, &th1) && pow2_col_blue_lookup(ilog2(n1), dp, 2, &v2, bits2, &tc2, &th2)
			    && pow2_col_blue_lookup(ilog2(n0), dp, 3, &v3, bits3, &tc3, &th3)) {
				uint64_t nsub = 1;
				for (auto& o : j.others) nsub *= o.count;
				std::vector<HostDim> dense = j.others;
				{ int64_t run = (int64_t)M; for (auto& o : dense) { o.inStride = o.outStride = run; run *= (int64_t)o.count; } }
				auto withOthers = [&](HostDim tiled, int inKind, int outKind) { // 0: caller's layout, 1: dense scratch rows of M
					std::vector<HostDim> r; r.push_back(tiled);
					for (size_t i = 0; i < j.others.size(); i++) r.push_back({j.others[i].count, inKind ? dense[i].inStride : j.others[i].inStride, outKind ? dense[i].outStride : j.others[i].outStride});
					return r;
				};
				auto setFast = [&](PassBuild& q, int variant, const int bits[4], int tc, int thr) {
					q.fastKernel = KERNEL_POW2_COL_BLUE; q.fastVariant = variant; q.fastThreads = thr; q.forceT = (uint32_t)tc;
					q.radices.clear();
					for (int k = 0; k < 4; k++) if (bits[k]) q.radices.push_back(1u << bits[k]);
					q.noCollapse = true;
				};
				PassBuild a = b;
				a.L = n0; a.inStrideJ = (int64_t)n1; a.outStrideJ = 1; a.colIn = true; a.colOut = false;
				a.dims = withOthers({n1, 1, (int64_t)n0}, 0, 1);
				a.preOp = OP_BLUESTEIN_PRE; a.auxOff2ForPre = chirpOff; a.bsSwapIn = j.inverse; a.opN = (uint32_t)N; a.opStrideJ = (uint32_t)n1; a.opStride0 = 1;
				a.postOp = OP_TWIDDLE_4STEP; a.fsN = M; a.fsColDiv = 1;
				a.inRole = j.inRole; a.outRole = ROLE_TEMP; a.label = "bluestein-1";
				setFast(a, v1, bits1, tc1, th1);
				PassBuild m2 = b;
				m2.L = n1; m2.inStrideJ = m2.outStrideJ = (int64_t)n0; m2.colIn = m2.colOut = true;
				m2.dims = withOthers({n0, 1, 1}, 1, 1);
				m2.midOp = OP_BLUESTEIN_MID; m2.aux2Off = bhatOff; m2.opStrideJ = (uint32_t)n0; m2.opStride0 = 1; m2.opStride1 = 0;
				m2.inRole = m2.outRole = ROLE_TEMP; m2.label = "bluestein-2";
				setFast(m2, v2, bits2, tc2, th2);
				PassBuild c3 = b;
				c3.L = n0; c3.inStrideJ = 1; c3.outStrideJ = (int64_t)n1; c3.colIn = c3.colOut = true;
				c3.dims = withOthers({n1, (int64_t)n0, 1}, 1, 0);
				c3.preOp = OP_FOURSTEP_INV_PRE; c3.fsN = M; c3.fsColDiv = 1;
				c3.postOp = OP_BLUESTEIN_POST; c3.auxOff2ForPre = chirpOff; c3.bsSwapOut = j.inverse; c3.opN = (uint32_t)N; c3.opStrideJ = (uint32_t)n1; c3.opStride0 = 1;
				c3.scale = j.scale;
				c3.inRole = ROLE_TEMP; c3.outRole = j.outRole; c3.label = "bluestein-3";
				setFast(c3, v3, bits3, tc3, th3);
				for (PassBuild* q : {&a, &m2, &c3}) { PassPlan pp; int r = finish_pass(*q, ar, pp); if (r) return r; passes.push_back(pp); }
				out.uploadsPerAxis[j.axisIndex] = 3;
				out.tempBytes = std::max<uint64_t>(out.tempBytes, nsub * M * es);
				return 0;
			}
		}
		if (spM.size() == 3 && (M & (M - 1)) == 0 && !d.disableFastKernels) {
			// three factors: five passes (see pow2_col_blue_kernel) instead of the seven of two Four-Step transforms + multiply
			const uint64_t n0 = spM[0], n1 = spM[1], n2 = spM[2], M1 = n1 * n2;
			int v1, v3, v4, v5, vb, bits1[4], bits3[4], bits4[4], bits5[4], bitsb[4], tc1, tc3, tc4, tc5, tcb, th1, th3, th4, th5, thb;
			if (pow2_col_blue_lookup(ilog2(n0), dp, 1, &v1, bits1, &tc1, &th1) && pow2_col_lookup(ilog2(n1), dp, &vb, bitsb, &tcb, &thb)
			    && pow2_col_blue_lookup(ilog2(n2), dp, 2, &v3, bits3, &tc3, &th3) && pow2_col_blue_lookup(ilog2(n1), dp, 4, &v4, bits4, &tc4, &th4)
			    && pow2_col_blue_lookup(ilog2(n0), dp, 3, &v5, bits5, &tc5, &th5)) {
				uint64_t nsub = 1;
				for (auto& o : j.others) nsub *= o.count;
				std::vector<HostDim> dense = j.others;
				{ int64_t run = (int64_t)M; for (auto& o : dense) { o.inStride = o.outStride = run; run *= (int64_t)o.count; } }
				auto withOthers = [&](std::vector<HostDim> lead, int inKind, int outKind) {
					for (size_t i = 0; i < j.others.size(); i++) lead.push_back({j.others[i].count, inKind ? dense[i].inStride : j.others[i].inStride, outKind ? dense[i].outStride : j.others[i].outStride});
					return lead;
				};
				auto setFast = [&](PassBuild& q, int kernel, int variant, const int bits[4], int tc, int thr) {
					q.fastKernel = kernel; q.fastVariant = variant; q.fastThreads = thr; q.forceT = (uint32_t)tc;
					q.radices.clear();
					for (int k = 0; k < 4; k++) if (bits[k]) q.radices.push_back(1u << bits[k]);
					q.noCollapse = true;
				};
				const int64_t sB = (int64_t)(n2 * n0); // stride of the middle factor's index inside T[m = i1*n2 + i2][k0]
				PassBuild p1 = b;
				p1.L = n0; p1.inStrideJ = (int64_t)M1; p1.outStrideJ = 1; p1.colIn = true; p1.colOut = false;
				p1.dims = withOthers({{M1, 1, (int64_t)n0}}, 0, 1);
				p1.preOp = OP_BLUESTEIN_PRE; p1.auxOff2ForPre = chirpOff; p1.bsSwapIn = j.inverse; p1.opN = (uint32_t)N; p1.opStrideJ = (uint32_t)M1; p1.opStride0 = 1;
				p1.postOp = OP_TWIDDLE_4STEP; p1.fsN = M; p1.fsColDiv = 1;
				p1.inRole = j.inRole; p1.outRole = ROLE_TEMP; p1.label = "bluestein5-1";
				setFast(p1, KERNEL_POW2_COL_BLUE, v1, bits1, tc1, th1);
				PassBuild p2 = b; // forward middle pass: FFT over i1, twiddle w_M1^(k1*i2), in place
				p2.L = n1; p2.inStrideJ = p2.outStrideJ = sB; p2.colIn = p2.colOut = true;
				p2.dims = withOthers({{n2 * n0, 1, 1}}, 1, 1);
				p2.postOp = OP_TWIDDLE_4STEP; p2.fsN = M1; p2.fsColDiv = (uint32_t)n0;
				p2.inRole = p2.outRole = ROLE_TEMP; p2.label = "bluestein5-2";
				setFast(p2, KERNEL_POW2_COL, vb, bitsb, tcb, thb);
				PassBuild p3 = b; // FFT over i2, * FFT(chirp)[k0 + n0*(k1 + n1*k2)], inverse FFT over k2, in place
				p3.L = n2; p3.inStrideJ = p3.outStrideJ = (int64_t)n0; p3.colIn = p3.colOut = true;
				p3.dims = withOthers({{n0, 1, 1}, {n1, sB, sB}}, 1, 1);
				p3.midOp = OP_BLUESTEIN_MID; p3.aux2Off = bhatOff; p3.opStrideJ = (uint32_t)(n0 * n1); p3.opStride0 = 1; p3.opStride1 = (uint32_t)n0;
				p3.inRole = p3.outRole = ROLE_TEMP; p3.label = "bluestein5-3";
				setFast(p3, KERNEL_POW2_COL_BLUE, v3, bits3, tc3, th3);
				PassBuild p4 = b; // the middle pass backwards
				p4.L = n1; p4.inStrideJ = p4.outStrideJ = sB; p4.colIn = p4.colOut = true;
				p4.dims = withOthers({{n2 * n0, 1, 1}}, 1, 1);
				p4.preOp = OP_FOURSTEP_INV_COL_PRE; p4.fsN = M1; p4.fsColDiv = (uint32_t)n0;
				p4.inRole = p4.outRole = ROLE_TEMP; p4.label = "bluestein5-4";
				setFast(p4, KERNEL_POW2_COL_BLUE, v4, bits4, tc4, th4);
				PassBuild p5 = b;
				p5.L = n0; p5.inStrideJ = 1; p5.outStrideJ = (int64_t)M1; p5.colIn = p5.colOut = true;
				p5.dims = withOthers({{M1, (int64_t)n0, 1}}, 1, 0);
				p5.preOp = OP_FOURSTEP_INV_PRE; p5.fsN = M; p5.fsColDiv = 1;
				p5.postOp = OP_BLUESTEIN_POST; p5.auxOff2ForPre = chirpOff; p5.bsSwapOut = j.inverse; p5.opN = (uint32_t)N; p5.opStrideJ = (uint32_t)M1; p5.opStride0 = 1;
				p5.scale = j.scale;
				p5.inRole = ROLE_TEMP; p5.outRole = j.outRole; p5.label = "bluestein5-5";
				setFast(p5, KERNEL_POW2_COL_BLUE, v5, bits5, tc5, th5);
				for (PassBuild* q : {&p1, &p2, &p3, &p4, &p5}) { PassPlan pp; int r = finish_pass(*q, ar, pp); if (r) return r; passes.push_back(pp); }
				out.uploadsPerAxis[j.axisIndex] = 5;
				out.tempBytes = std::max<uint64_t>(out.tempBytes, nsub * M * es);
				return 0;
			}
		}
		if (!spM.empty()) {
			// multi-pass Bluestein: FFT_M (Four-Step) with the chirp fused into its first load and FFT(chirp) into its last
			// store, then the inverse FFT_M with the second chirp fused into its last store.  Scratch: T1 | T2.
			uint64_t nsub = 1;
			for (auto& o : j.others) nsub *= o.count;
			auto dense = [&](std::vector<HostDim> v) { int64_t run = (int64_t)M; for (auto& o : v) { o.inStride = o.outStride = run; run *= (int64_t)o.count; } return v; };
			MultiPassIO f;
			f.othersIn = j.others; for (auto& o : f.othersIn) o.outStride = o.inStride;
			f.othersOut = dense(j.others);
			f.inRole = j.inRole; f.outRole = ROLE_TEMP; f.outOffset = (int64_t)(nsub * M); f.t1Offset = 0;
			f.firstPre = OP_BLUESTEIN_PRE; f.firstAux = chirpOff; f.bsSwapIn = j.inverse; f.opN = (uint32_t)N;
			f.lastPost = OP_MUL_LUT; f.lastAux2 = bhatOff;
			int r = emit_multipass(b, M, spM, f, ar, passes); if (r) return r;
			MultiPassIO g;
			g.othersIn = dense(j.others);
			g.othersOut = j.others; for (auto& o : g.othersOut) o.inStride = o.outStride;
			g.inRole = ROLE_TEMP; g.inOffset = (int64_t)(nsub * M); g.outRole = j.outRole; g.t1Offset = 0;
			g.swapIn = g.swapOut = true; g.scale = j.scale;
			g.lastPost = OP_BLUESTEIN_POST; g.lastAux = chirpOff; g.bsSwapOut = j.inverse; g.opN = (uint32_t)N;
			r = emit_multipass(b, M, spM, g, ar, passes); if (r) return r;
			out.uploadsPerAxis[j.axisIndex] = (uint32_t)(2 * spM.size());
			out.tempBytes = std::max<uint64_t>(out.tempBytes, 2 * nsub * M * es);
			return 0;
		}
		b.L = M; b.inLen = (uint32_t)N; b.outLen = (uint32_t)N; b.opN = (uint32_t)N;
		b.preOp = OP_BLUESTEIN_PRE; b.midOp = OP_BLUESTEIN_MID; b.postOp = OP_BLUESTEIN_POST;
		b.bsSwapIn = b.bsSwapOut = j.inverse;
		b.auxOff = chirpOff; b.aux2Off = bhatOff;
		b.scale = j.scale;
		b.inStrideJ = j.inStrideJ; b.outStrideJ = j.outStrideJ;
		b.colIn = b.colOut = !unit;
		b.dims = j.others;
		b.label = "bluestein";
		PassPlan pp; int r = finish_pass(b, ar, pp);
		if (r) return padded ? kPadUnsupported : r; // (a padded row whose fused kernel was refused — span, table — must not die on the interpreter's LDS limit: zero-fill fallback)
		if (padded && pp.kernel == KERNEL_GENERIC && M * (dp ? 16 : 8) * 2 > d.maxLds) return kPadUnsupported;
		passes.push_back(pp);
		out.uploadsPerAxis[j.axisIndex] = 1;
		return 0;
	}

	uint64_t singleCap = unit ? rowCap : max_col_len(dp, d.maxLds, 1);
	if (!unit && !d.disableFastKernels && j.N > 2048 && j.N <= singleCap && !j.others.empty() && j.others[0].inStride == 1 && j.others[0].outStride == 1) {
		// a strided axis longer than the hand-specialised column kernels reach (2048): one pass would put ONE column in LDS per workgroup
		// (uncoalesced 8-byte accesses, measured 0.4 TB/s on 4096 x 4096); two passes over tiles of neighbouring columns run at copy speed
		int variant, bits[4], tc, thr, rad5[5], fpw;
		const bool p2 = (j.N & (j.N - 1)) == 0;
		std::vector<uint64_t> probe;
		if (!(p2 && pow2_col_lookup(ilog2(j.N), dp, &variant, bits, &tc, &thr)) && !opfft_lookup(j.N, dp, true, false, 0, 0, &variant, rad5, &fpw, &thr) &&
		    choose_split(j.N, dp, d.maxLds, dmax, true, probe)) {
			if (padded) return kPadUnsupported; // (two passes: no kernel of that plan skips elements)
			singleCap = 2048;
		}
	}
	// 2^15 fp32 as ONE pass of the register-lean row kernel (measured 4.2 TB/s against 3.2 for the fused two-pass kernel); VKFFT_MI355X_ROW15=0: two passes
	const bool row15 = !(getenv("VKFFT_MI355X_ROW15") && atoi(getenv("VKFFT_MI355X_ROW15")) == 0);
	// (a power-of-two row beyond the interpreter's reach takes this branch only when its register-resident kernel is really there — table entry and 32-bit
	// span —: otherwise the interpreter would be handed a row that does not fit LDS (error 3002) instead of the Four-Step plan below)
	bool p2rowOK = false;
	if (unit && !d.disableFastKernels && (j.N & (j.N - 1)) == 0 && j.N >= 4 && j.N <= (dp ? 8192u : (row15 ? 32768u : 16384u))) {
		int variant, bits[4], fpw, thr;
		const uint64_t rowPitch = j.others.empty() ? j.N : (uint64_t)std::max<int64_t>(std::llabs(j.others[0].inStride), std::llabs(j.others[0].outStride));
		p2rowOK = (rowPitch * 64 + j.N) * (dp ? 16 : 8) < 0x7FFFFF00ull && pow2_row_lookup(ilog2(j.N), dp, &variant, bits, &fpw, &thr, padded);
	}
	// ... and the hand-written long rows of the mixed-radix family (mixed_table_6.inc: 11^4, 5^6, 7^5 in ONE LDS buffer of 117-151 KB, one workgroup per CU): one pass where
	// the Four-Step plan would take two (VKFFT_MI355X_LONGROWS=0: the fused Four-Step launch of kernel_mix_fused.h instead)
	bool mixLongOK = false;
	if (unit && !padded && !d.disableFastKernels && (j.N & (j.N - 1)) != 0 && j.N > singleCap && j.N <= (dp ? 8192u : 16807u) && !(getenv("VKFFT_MI355X_LONGROWS") && atoi(getenv("VKFFT_MI355X_LONGROWS")) == 0)) {
		int variant, rad5[5], fpw, thr;
		const uint64_t rowPitch = j.others.empty() ? j.N : (uint64_t)std::max<int64_t>(std::llabs(j.others[0].inStride), std::llabs(j.others[0].outStride));
		mixLongOK = (rowPitch * 64 + j.N) * (dp ? 16 : 8) < 0x7FFFFF00ull && mixed_row_lookup(j.N, dp, &variant, rad5, &fpw, &thr);
	}
	if (j.N <= singleCap || p2rowOK || mixLongOK) {
		b.L = j.N;
		if (unit && !padded && !d.disableFastKernels && ((j.N & (j.N - 1)) != 0 || j.N == 2)) { // curated non-power-of-two lengths (and N = 2): hand-specialised mixed-radix kernel
			int variant, rad5[5], fpw, thr;
			uint64_t rowPitch = j.others.empty() ? j.N : (uint64_t)std::max<int64_t>(std::llabs(j.others[0].inStride), std::llabs(j.others[0].outStride));
			if ((rowPitch * 64 + j.N) * (dp ? 16 : 8) < 0x7FFFFF00ull && mixed_row_lookup(j.N, dp, &variant, rad5, &fpw, &thr)) {
				b.fastKernel = KERNEL_MIXED_ROW; b.fastVariant = variant; b.fastThreads = thr; b.forceT = (uint32_t)fpw;
				for (int k = 0; k < 5; k++) if (rad5[k] > 1) b.radices.push_back((uint32_t)rad5[k]);
			}
		}
		if (unit && !d.disableFastKernels && (j.N & (j.N - 1)) == 0 && j.N >= 4) {
			int variant, bits[4], fpw, thr;
			uint64_t rowPitch = j.others.empty() ? j.N : (uint64_t)std::max<int64_t>(std::llabs(j.others[0].inStride), std::llabs(j.others[0].outStride));
			if ((rowPitch * 64 + j.N) * (dp ? 16 : 8) < 0x7FFFFF00ull && pow2_row_lookup(ilog2(j.N), dp, &variant, bits, &fpw, &thr, padded)) {
				b.fastKernel = KERNEL_POW2_ROW; b.fastVariant = variant; b.fastThreads = thr; b.forceT = (uint32_t)fpw;
				for (int k = 0; k < 4; k++) if (bits[k]) b.radices.push_back(1u << bits[k]);
			}
		}
		b.inStrideJ = j.inStrideJ; b.outStrideJ = j.outStrideJ;
		b.colIn = b.colOut = !unit;
		b.dims = j.others;
		b.swapIn = b.swapOut = j.inverse;
		b.scale = j.scale;
		b.label = unit ? "c2c-row" : "c2c-col";
		PassPlan pp; int r = finish_pass(b, ar, pp); if (r) return r;
		passes.push_back(pp);
		out.uploadsPerAxis[j.axisIndex] = 1;
		return 0;
	}
	if (padded && j.N > (unit ? rowCap : max_col_len(dp, d.maxLds, 1)) && !unit) return kPadUnsupported; // (strided multi-pass plans do not skip elements)
	if (!unit) {
		// multi-pass along a strided axis: needs a unit-stride companion dimension (others[0]) to tile over
		if (j.others.empty() || j.others[0].inStride != 1 || j.others[0].outStride != 1) return 3002;
		std::vector<uint64_t> sps;
		if (!choose_split(j.N, dp, d.maxLds, dmax, !d.disableFastKernels, sps)) return 3002;
		std::vector<HostDim> rest(j.others.begin() + 1, j.others.end());
		uint64_t tempElems = 0;
		int r = emit_multipass_strided(b, j.N, sps, j.inStrideJ, j.outStrideJ, j.others[0], rest, j.inRole, j.outRole, j.inverse, j.scale, ar, passes, tempElems);
		if (r) return r;
		out.uploadsPerAxis[j.axisIndex] = (uint32_t)sps.size();
		for (size_t i = 0; i < sps.size(); i++) out.axisSplit[j.axisIndex][i] = sps[sps.size() - 1 - i];
		out.tempBytes = std::max<uint64_t>(out.tempBytes, tempElems * (dp ? 16 : 8));
		return 0;
	}

	// ---- Four-Step on a unit-stride axis: N = n0 * M, recursively M = n1 * n2 -------------------------
	if (!padded && emit_fused(d, j, ar, out, passes)) return 0;
	if (!padded && emit_mix_fused(d, j, ar, out, passes)) return 0;
	std::vector<uint64_t> sp;
	if (!choose_split(j.N, dp, d.maxLds, dmax, !d.disableFastKernels, sp)) return 3002;
	if (padded) {
		// zero padding on an axis of several passes (vkFFT_Zeropad.h:28-182 masks every upload by the natural index): element j of a first-pass column is the
		// point j * M + m and output k of a last-pass column the frequency k0 + (N / len) k, so a range whose ends are multiples of those strides is a range of
		// j (of k) alone and the per-pass masks of the column kernels apply; other ranges fall back
		const uint64_t sIn = j.N / sp[0], sOut = j.N / sp.back();
		if ((j.padInN && (j.padInL % sIn || j.padInN % sIn)) || (j.padOutN && (j.padOutL % sOut || j.padOutN % sOut))) return kPadUnsupported;
	}
	out.uploadsPerAxis[j.axisIndex] = (uint32_t)sp.size();
	for (size_t i = 0; i < sp.size(); i++) out.axisSplit[j.axisIndex][i] = sp[sp.size() - 1 - i];
	MultiPassIO io;
	io.othersIn = j.others; io.othersOut = j.others;
	for (auto& o : io.othersIn) o.outStride = o.inStride;
	for (auto& o : io.othersOut) o.inStride = o.outStride;
	io.inRole = j.inRole; io.outRole = j.outRole;
	io.swapIn = io.swapOut = j.inverse; io.scale = j.scale;
	io.padInL = j.padInL; io.padInN = j.padInN; io.padOutL = j.padOutL; io.padOutN = j.padOutN;
	int r = emit_multipass(b, j.N, sp, io, ar, passes);
	if (r) return r;
	uint64_t nsub = 1;
	for (auto& o : j.others) nsub *= o.count;
	out.tempBytes = std::max<uint64_t>(out.tempBytes, nsub * j.N * (dp ? 16 : 8));
	return 0;
}

// pair pass of the even R2C / C2R decomposition (r2c_even_pair_kernel), in place on the complex rows
static int make_r2c_pair_pass(uint64_t N, bool dp, bool inverse, const std::vector<HostDim>& othersCplx, int cplxRole, Arena& ar, PassPlan& pair) {
	const size_t es = dp ? 16 : 8;
	memset(&pair.prm, 0, sizeof(pair.prm));
	PassParams& q = pair.prm;
	std::vector<HostDim> cd; for (auto& o : othersCplx) cd.push_back(o);
	collapse_dims(cd);
	while (cd.size() < 3) cd.push_back({1, 0, 0});
	if (cd.size() > 3) return 3003;
	for (int i = 0; i < 3; i++) { q.dim[i].count = (uint32_t)cd[i].count; q.dim[i].inStride = q.dim[i].outStride = cd[i].inStride; }
	q.opN = (uint32_t)N; q.fsN = (uint32_t)N; q.scale = 1.0; q.swapIn = inverse ? 1 : 0;
	uint32_t lo = (ceil_log2(N) + 1) / 2; uint64_t nlo = 1ull << lo, nhi = (N + nlo - 1) / nlo;
	size_t off = ar.alloc((nlo + nhi) * es);
	for (uint64_t i = 0; i < nlo; i++) ar.putc(off, i, unit_root(i, N), dp);
	for (uint64_t i = 0; i < nhi; i++) ar.putc(off, nlo + i, unit_root(i * nlo, N), dp);
	q.fsLoBits = lo; pair.auxOff = off;
	q.tilesPerG0 = 1;
	pair.kernel = KERNEL_R2C_PAIR; pair.dp = dp; pair.inRole = pair.outRole = cplxRole;
	pair.inElemBytes = pair.outElemBytes = (int)es; pair.threads = 256; pair.label = inverse ? "c2r-pair" : "r2c-pair";
	return 0;
}

// Real rows whose complex length has a prime factor above 31 that the interpreter would take as a direct O(p^2) Rader stage (47, 59: p - 1 is not smooth) or
// that no instance kernel serves (a Rader prime with an unserved cofactor): measured at 0.12-0.26x the reference (R2C / DCT rows of 94, 118, 235, 295, 376
// reals, profiles/r04_*_rows_*) — the fused Bluestein kernel of the real transforms (kernel_blue_r2r.h) takes them instead.
static bool real_row_prefers_bluestein(uint64_t L, bool dp) {
	if (L < 2 || L > 4096) return false;
	uint64_t P = 0, rest = L;
	for (uint64_t q = 2; q * q <= rest; q++) while (rest % q == 0) { P = q; rest /= q; }
	if (rest > 1) P = rest;
	if (P <= 31) return false;
	int v, r5[5], f, t; uint64_t len;
	if (mixed_row_lookup(L, dp, &v, r5, &f, &t)) return false;
	if (P == L) return !mixconv_lookup(true, false, P, dp, &v, &len, r5, &f, &t);
	MixradChoice mr;
	return !mixrad_choose(L, dp, true, mr);
}

// ---- real transforms: coverage path -------------------------------------------------------------------------
// A real transform along one axis as  pre-map pass -> complex transform of the embedding sequence on dense rows in ROLE_TEMP2 -> post-map pass
// (kernels_aux.hip real_map_kernel).  The complex transform is whatever plan its length needs: Bluestein of any size, several passes.  Used where no
// fused form exists: the embedding length needs Bluestein on a strided axis or beyond the fused Bluestein kernels' reach (the reference covers
// these inside its Bluestein kernels: vkFFT_Scheduler.h:2271-2280, 2894-2944; vkFFT_R2R.h; vkFFT_R2C.h:27).  Two extra trips through memory.
struct RealMapJob {
	uint32_t preOp = 0, postOp = 0;
	uint64_t N = 0, Lm = 0;            // real length, length of the embedding sequence
	uint32_t outLen = 0;               // outputs per row (limit of the post-map's scatter)
	bool cinverse = false;             // the embedding sequence is transformed backwards
	int64_t strideIn = 1, strideOut = 1; // element strides along the axis (elements of the respective side)
	std::vector<HostDim> others;       // rows: count, inStride (input elements), outStride (output elements)
	int inRole = ROLE_BUFFER, outRole = ROLE_BUFFER, inElemBytes = 4, outElemBytes = 4;
	size_t preAux3 = (size_t)-1, postAux = (size_t)-1, postAux2 = (size_t)-1;
	double scale = 1.0;
	int axisIndex = 0;
};
static int plan_c2c_axis(const TransformDesc& d, const AxisJob& j, Arena& ar, DirectionPlan& out, std::vector<PassPlan>& passes);
static int plan_real_by_maps(const TransformDesc& d, const RealMapJob& m, Arena& ar, DirectionPlan& out, std::vector<PassPlan>& passes) {
	const bool dp = d.dp;
	const size_t es = dp ? 16 : 8;
	std::vector<HostDim> rows = m.others;
	collapse_dims(rows);
	while (rows.size() < 3) rows.push_back({1, 0, 0});
	if (rows.size() > 3 || m.Lm < 2 || m.Lm >= (1ull << 31)) return 3002;
	uint64_t nsub = 1; for (auto& r : rows) { if (r.count >= (1ull << 31)) return 3002; nsub *= r.count; }
	if (nsub * ((m.Lm + 255) / 256) >= (1ull << 31)) return 3002;
	const size_t mark = passes.size();
	auto mapPass = [&](bool pre) {
		PassPlan t; memset(&t.prm, 0, sizeof(t.prm));
		PassParams& q = t.prm;
		q.L = (uint32_t)m.Lm; q.opN = (uint32_t)m.N; q.tilesPerG0 = 1; q.scale = pre ? 1.0 : m.scale;
		for (int i = 0; i < 3; i++) q.dim[i] = {(uint32_t)rows[i].count, rows[i].inStride, rows[i].outStride};
		q.inStrideJ = m.strideIn; q.outStrideJ = m.strideOut;
		if (pre) { q.preNat = 1; q.preOp = m.preOp; t.aux3Off = m.preAux3; t.inRole = m.inRole; t.outRole = ROLE_TEMP2; t.inElemBytes = m.inElemBytes; t.outElemBytes = (int)es; }
		else { q.postNat = 1; q.postOp = m.postOp; q.natOutLen = m.outLen; t.auxOff = m.postAux; t.aux2Off = m.postAux2; t.inRole = ROLE_TEMP2; t.outRole = m.outRole; t.inElemBytes = (int)es; t.outElemBytes = m.outElemBytes; }
		t.kernel = KERNEL_REAL_MAP; t.dp = dp; t.threads = 256; t.label = pre ? "real-pre-map" : "real-post-map";
		passes.push_back(t);
	};
	mapPass(true);
	AxisJob hj;
	hj.N = m.Lm; hj.dp = dp; hj.inverse = m.cinverse; hj.scale = 1.0; hj.inRole = hj.outRole = ROLE_TEMP2; hj.axisIndex = m.axisIndex;
	hj.others.push_back({nsub, (int64_t)m.Lm, (int64_t)m.Lm});
	const int r = plan_c2c_axis(d, hj, ar, out, passes);
	if (r) { passes.resize(mark); return r; }
	mapPass(false);
	out.temp2Bytes = std::max<uint64_t>(out.temp2Bytes, nsub * m.Lm * es);
	out.uploadsPerAxis[m.axisIndex] += 2;
	out.axisSplit[m.axisIndex][0] = m.N;
	return 0;
}

// ---- real transforms ---------------------------------------------------------------------------------------
// R2C/C2R along axis 0 (reference: two-sequences packing vkFFT_R2C.h:450/:178 for single-upload, even
// decomposition vkFFT_R2C_even_decomposition.h:40 for long even N, callback form vkFFT_R2C.h:27 otherwise).
// Here: even N -> one half-length complex FFT per row with the split fused as a post/pre operation of the
// same kernel; odd N -> full-length complex FFT of the real row.
// Even real lengths as TWO rows per full-length complex transform instead of one row per half-length transform (the same points per row; kernel_tmaps.h): the
// maps of the paired form are a signed gather and the even / odd split, those of the half-length forms carry a twiddle per point and go through the generic
// maps.  Where an instance transform of the full length exists with eight or more threads per row, and (mode 1) the half-length form has no fused-map kernel.
// VKFFT_MI355X_EVEN_FULL = 0 off, 1 (default), 2 also over a fused-map kernel.
// rows that can travel in pairs: the extent of the tiled dimension after finish_pass has merged the contiguous batch dimensions into it (the pairs are formed inside
// dims[0] only: a row count of 1 there — non-collapsible outer dimensions — would carry ONE real row per full-length transform, twice the work of the half-length form)
static uint64_t pairable_rows(const std::vector<HostDim>& others) {
	if (others.empty()) return 1;
	uint64_t c = others[0].count;
	for (size_t i = 1; i < others.size(); i++) {
		if (others[i].count > 1 && ((int64_t)c * others[0].inStride != others[i].inStride || (int64_t)c * others[0].outStride != others[i].outStride)) break;
		c *= others[i].count;
	}
	return c;
}
// a unit-stride fp32 row longer than the two-buffer single-pass limit that has one of the long instances of tools/gen_long_rows_table.py (one LDS buffer, one workgroup per CU)
static bool long_row_instance(const TransformDesc& d, uint64_t L, bool dp) {
	if (d.disableFastKernels || L <= max_row_len(dp, d.maxLds) || L > (dp ? 8192u : 16807u) || (getenv("VKFFT_MI355X_LONGROWS") && atoi(getenv("VKFFT_MI355X_LONGROWS")) == 0)) return false;
	int v, r5[5], f, t;
	return mixed_row_lookup(L, dp, &v, r5, &f, &t);
}
static bool prefer_full_length_pairs(const TransformDesc& d, uint64_t N, bool unit, uint64_t rows, uint32_t preHalf, uint32_t postHalf, uint64_t halfLen) {
	const int mode = getenv("VKFFT_MI355X_EVEN_FULL") ? atoi(getenv("VKFFT_MI355X_EVEN_FULL")) : 1;
	if (N > max_row_len(d.dp, d.maxLds)) return false; // (the long rows keep their half-length forms: one workgroup per CU is no place for twice the points)
	if (!mode || d.disableFastKernels || !unit || rows < 2 || getenv("VKFFT_MI355X_NO_TMAPS") || getenv("VKFFT_MI355X_NO_ROW_PAIRS") || getenv("VKFFT_MI355X_NO_MIXED_OPS")) return false;
	int v, r5[5], f, t;
	// (eight: the plain sides of R2C / C2R then move directly, kernel_mixed.h DIRECT; round 6: or the full length runs on the Rader-stage kernel, whose maps are the
	// tables of the full-length forms only — the half-length complex form of 328 = 2 * 4 * 41 reals fell back to the interpreter: 0.17x the reference)
	MixradChoice mr;
	if (!(mixed_row_lookup(N, d.dp, &v, r5, &f, &t) && t / f >= 8) && !mixrad_choose(N, d.dp, true, mr)) return false;
	if (mode == 1 && opfft_lookup(halfLen, d.dp, false, false, preHalf, postHalf, &v, r5, &f, &t)) return false;
	return true;
}
static int plan_r2c_axis0_fused(const TransformDesc& d, bool inverse, const std::vector<HostDim>& othersReal, const std::vector<HostDim>& othersCplx,
                                int realRole, int cplxRole, double scale, Arena& ar, DirectionPlan& out, std::vector<PassPlan>& passes, bool usePad);
static int plan_r2c_axis0(const TransformDesc& d, bool inverse, const std::vector<HostDim>& othersReal, const std::vector<HostDim>& othersCplx,
                          int realRole, int cplxRole, double scale, Arena& ar, DirectionPlan& out, std::vector<PassPlan>& passes) {
	const size_t mark = passes.size();
	const bool padded = d.padR[0] > d.padL[0];
	int r = plan_r2c_axis0_fused(d, inverse, othersReal, othersCplx, realRole, cplxRole, scale, ar, out, passes, padded);
	if (r == kPadUnsupported) { // the range is zeroed ahead of the transform instead (api.cpp)
		passes.resize(mark);
		out.padFallbackMask |= 1u;
		r = plan_r2c_axis0_fused(d, inverse, othersReal, othersCplx, realRole, cplxRole, scale, ar, out, passes, false);
	}
	if (r != 3003) return r;
	if (padded) out.padFallbackMask |= 1u;
	// no fused form (a row length that needs Bluestein beyond the fused kernels' reach): full-length "callback" form (vkFFT_R2C.h:27) as separate map passes
	passes.resize(mark);
	const uint64_t N = d.size[0];
	RealMapJob m;
	m.N = N; m.Lm = N; m.cinverse = inverse; m.scale = scale; m.axisIndex = 0;
	m.preOp = m.postOp = inverse ? OP_C2R_FULL : OP_R2C_FULL;
	m.outLen = (uint32_t)(inverse ? N : N / 2 + 1);
	const int rb = d.dp ? 8 : 4;
	m.inRole = inverse ? cplxRole : realRole; m.outRole = inverse ? realRole : cplxRole;
	m.inElemBytes = inverse ? 2 * rb : rb; m.outElemBytes = inverse ? rb : 2 * rb;
	for (size_t i = 0; i < othersReal.size(); i++) {
		const int64_t rs = othersReal[i].inStride, cs = othersCplx[i].inStride;
		m.others.push_back(inverse ? HostDim{othersReal[i].count, cs, rs} : HostDim{othersReal[i].count, rs, cs});
	}
	const int r2 = plan_real_by_maps(d, m, ar, out, passes);
	return r2 == 3002 ? 3003 : r2;
}
static int plan_r2c_axis0_fused(const TransformDesc& d, bool inverse, const std::vector<HostDim>& othersReal, const std::vector<HostDim>& othersCplx,
                                int realRole, int cplxRole, double scale, Arena& ar, DirectionPlan& out, std::vector<PassPlan>& passes, bool usePad) {
	const uint64_t N = d.size[0];
	const bool dp = d.dp;
	const size_t es = dp ? 16 : 8;
	const uint32_t dmax = direct_max(d);
	// zero padding of the real rows (spatial padding: read side of R2C, write side of C2R), in real elements; only the single-pass forms skip elements
	const bool padReal = usePad && !d.padFrequency;
	if (usePad && !padReal) return kPadUnsupported; // (padding of the half spectrum along axis 0)
	const uint32_t padL = padReal ? (uint32_t)d.padL[0] : 0u, padN = padReal ? (uint32_t)(d.padR[0] - d.padL[0]) : 0u;
	PassBuild b;
	b.dp = dp; b.maxLds = d.maxLds; b.raderDirectMax = dmax; b.allowFast = false; b.allowOp = !d.disableFastKernels;
	b.opN = (uint32_t)N; b.scale = scale;
	bool even = (N % 2 == 0);
	{
		const uint64_t rows = std::min(pairable_rows(othersReal), pairable_rows(othersCplx));
		if (even && N >= 4 && !padReal && prefer_full_length_pairs(d, N, true, rows, inverse ? OP_C2R_EVEN_PRE : OP_NONE, inverse ? OP_NONE : OP_R2C_EVEN_POST, N / 2)) even = false;
	}
	b.L = even ? N / 2 : N;
	uint64_t blueM = 0; // padded length of the Bluestein-wrapped full-length form (kernel_blue_r2r.h), 0: not used
	int blueVariant = 0, blueBits[4] = {0, 0, 0, 0}, blueFpw = 0, blueThr = 0;
	if (padReal && (!is_supported_len(b.L, dmax) || b.L > max_row_len(dp, d.maxLds))) return kPadUnsupported;
	bool blueByChoice = false; // the length is within the interpreter's reach, the fused Bluestein kernel is the faster form (real_row_prefers_bluestein)
	if (is_supported_len(b.L, dmax) && !d.disableFastKernels && !padReal && real_row_prefers_bluestein(b.L, dp) && !getenv("VKFFT_MI355X_NO_REAL_BLUE_CHOICE")) {
		uint64_t Mp = 64; while (Mp < 2 * N - 1) Mp *= 2;
		uint64_t pitch = N + 2;
		if (!othersReal.empty()) pitch = (uint64_t)std::max<int64_t>(std::llabs(othersReal[0].inStride), 2 * std::llabs(othersCplx[0].inStride));
		int v, bits[4], fpw, thr;
		blueByChoice = Mp <= (dp ? 4096u : 8192u) && (pitch * 64 + 2 * N) * (dp ? 8 : 4) < 0x7FFFFF00ull && pow2_blue_r2r_lookup(ilog2(Mp), dp, inverse ? OP_C2R_FULL : OP_R2C_FULL, &v, bits, &fpw, &thr);
	}
	if (!is_supported_len(b.L, dmax) || blueByChoice) {
		// the (half) length has a prime factor outside the radix / Rader stages: full-length "callback" form (real -> (x, 0),
		// keep the first N/2+1 outputs; vkFFT_R2C.h:27) around a fused Bluestein transform of length N
		if (d.disableFastKernels) return 3003;
		uint64_t Mp = 64; while (Mp < 2 * N - 1) Mp *= 2;
		if (even && Mp > (dp ? 4096u : 8192u)) {
			// long even rows: the full-length form would need a padded length of 16384 or more (one 128 KiB workgroup per CU, measured
			// 0.24 TB/s at N = 5606).  Half-length complex transform of the packed pairs — whatever plan that length needs, here the fused
			// Bluestein kernel on a quarter of the padded length — plus the pair pass of the even decomposition (vkFFT_R2C_even_decomposition.h:40)
			const size_t mark = passes.size();
			PassPlan pair;
			int pr = make_r2c_pair_pass(N, dp, inverse, othersCplx, cplxRole, ar, pair);
			if (pr == 0) {
				AxisJob hj;
				hj.N = N / 2; hj.dp = dp; hj.inverse = inverse; hj.scale = scale; hj.axisIndex = 0;
				hj.inRole = inverse ? cplxRole : realRole; hj.outRole = inverse ? realRole : cplxRole;
				bool ok = true;
				for (size_t i = 0; i < othersReal.size(); i++) {
					int64_t rs = othersReal[i].inStride; const int64_t cs = othersCplx[i].inStride;
					if (rs % 2) { ok = false; break; }
					rs /= 2;
					hj.others.push_back(inverse ? HostDim{othersReal[i].count, cs, rs} : HostDim{othersReal[i].count, rs, cs});
				}
				if (ok) {
					if (inverse) passes.push_back(pair);
					const int r = plan_c2c_axis(d, hj, ar, out, passes);
					if (r == 0) {
						if (!inverse) passes.push_back(pair);
						out.uploadsPerAxis[0] += 1;
						out.axisSplit[0][0] = N; out.bigSequenceEvenR2C = 1;
						return 0;
					}
					passes.resize(mark);
				}
			}
		}
		uint64_t pitch = N + 2;
		if (!othersReal.empty()) pitch = (uint64_t)std::max<int64_t>(std::llabs(othersReal[0].inStride), 2 * std::llabs(othersCplx[0].inStride));
		if ((pitch * 64 + 2 * N) * (dp ? 8 : 4) >= 0x7FFFFF00ull) return 3003; // 32-bit buffer offsets inside a tile of rows
		if (!pow2_blue_r2r_lookup(ilog2(Mp), dp, inverse ? OP_C2R_FULL : OP_R2C_FULL, &blueVariant, blueBits, &blueFpw, &blueThr)) return 3003;
		blueM = Mp; even = false; b.L = N;
	}
	// rows: combine the real-side and complex-side strides per dim
	std::vector<HostDim> dims;
	for (size_t i = 0; i < othersReal.size(); i++) {
		HostDim h; h.count = othersReal[i].count;
		int64_t rs = othersReal[i].inStride, cs = othersCplx[i].inStride;
		if (even) { if (rs % 2) return 3003; rs /= 2; } // real rows viewed as packed complex pairs
		if (!inverse) { h.inStride = rs; h.outStride = cs; } else { h.inStride = cs; h.outStride = rs; }
		dims.push_back(h);
	}
	if (b.L > max_row_len(dp, d.maxLds) && !(!blueM && !padReal && long_row_instance(d, b.L, dp))) {
		// long even rows: multi-pass half-length complex FFT + the pair pass of the even decomposition
		// (reference: VkFFTPlanR2CMultiUploadDecomposition, vkFFT_Plan_R2C.h:30; kernel vkFFT_R2C_even_decomposition.h:40)
		if (!even) {
			// odd rows longer than one pass: the full-length "callback" form (vkFFT_R2C.h:27) through a multi-pass complex FFT of
			// length N; first load real -> (x, 0) / Hermitian expansion, last store the first N/2+1 outputs / the real part
			if (blueM) return 3003;
			std::vector<uint64_t> sp;
			if (!choose_split(N, dp, d.maxLds, dmax, !d.disableFastKernels, sp)) return 3003;
			MultiPassIO io;
			for (auto& h : dims) { io.othersIn.push_back({h.count, h.inStride, h.inStride}); io.othersOut.push_back({h.count, h.outStride, h.outStride}); }
			io.inRole = inverse ? cplxRole : realRole; io.outRole = inverse ? realRole : cplxRole;
			io.swapIn = io.swapOut = inverse; io.scale = scale;
			io.firstPre = inverse ? OP_C2R_FULL : OP_R2C_FULL; io.lastPost = io.firstPre;
			io.natural = true; io.firstRealIn = !inverse; io.lastRealOut = inverse;
			io.natOutLen = (uint32_t)(inverse ? N : N / 2 + 1); io.opN = (uint32_t)N; io.blueN = (uint32_t)N;
			PassBuild proto; proto.dp = dp; proto.maxLds = d.maxLds; proto.raderDirectMax = dmax; proto.allowFast = !d.disableFastKernels; proto.allowOp = !d.disableFastKernels;
			int r = emit_multipass(proto, N, sp, io, ar, passes); if (r) return r == 3002 ? 3003 : r;
			uint64_t nsub = 1; for (auto& h : dims) nsub *= h.count;
			out.tempBytes = std::max<uint64_t>(out.tempBytes, nsub * N * es);
			out.uploadsPerAxis[0] = (uint32_t)sp.size();
			out.axisSplit[0][0] = N;
			return 0;
		}
		const uint64_t H = N / 2;
		std::vector<uint64_t> sp;
		if (!choose_split(H, dp, d.maxLds, dmax, !d.disableFastKernels, sp)) return 3003;
		PassBuild proto; proto.dp = dp; proto.maxLds = d.maxLds; proto.raderDirectMax = dmax; proto.allowFast = !d.disableFastKernels;
		// pair pass descriptor (in place on the complex rows)
		PassPlan pair;
		if (int pr = make_r2c_pair_pass(N, dp, inverse, othersCplx, cplxRole, ar, pair)) return pr;
		MultiPassIO io;
		for (auto& h : dims) { io.othersIn.push_back({h.count, h.inStride, h.inStride}); io.othersOut.push_back({h.count, h.outStride, h.outStride}); }
		io.inRole = inverse ? cplxRole : realRole; io.outRole = inverse ? realRole : cplxRole;
		io.swapIn = io.swapOut = inverse; io.scale = scale;
		if (inverse) passes.push_back(pair);
		int r = emit_multipass(proto, H, sp, io, ar, passes); if (r) return r == 3002 ? 3003 : r;
		if (!inverse) passes.push_back(pair);
		uint64_t nsub = 1; for (auto& h : dims) nsub *= h.count;
		out.tempBytes = std::max<uint64_t>(out.tempBytes, nsub * H * es);
		out.uploadsPerAxis[0] = (uint32_t)sp.size() + 1;
		out.axisSplit[0][0] = N; out.bigSequenceEvenR2C = 1;
		return 0;
	}
	b.dims = dims;
	b.inStrideJ = b.outStrideJ = 1;
	if (padN) { // the even forms move the reals as packed pairs: an odd boundary would cut a pair
		if (even && ((padL | padN) & 1u)) return kPadUnsupported;
		const uint32_t pl = even ? padL / 2 : padL, pn = even ? padN / 2 : padN;
		if (!inverse) { b.padInL = pl; b.padInN = pn; } else { b.padOutL = pl; b.padOutN = pn; }
	}
	if (even) {
		size_t aux = ar.alloc((N / 2 + 1) * es);
		for (uint64_t k = 0; k <= N / 2; k++) ar.putc(aux, k, unit_root(k, N), dp);
		b.auxOff = aux;
		if (!inverse) { b.postOp = OP_R2C_EVEN_POST; b.outLen = (uint32_t)(N / 2 + 1); b.inLen = (uint32_t)(N / 2); }
		else { b.preOp = OP_C2R_EVEN_PRE; b.swapIn = b.swapOut = true; b.inLen = (uint32_t)(N / 2 + 1); b.outLen = (uint32_t)(N / 2); }
	} else {
		if (!inverse) { b.preOp = OP_R2C_FULL; b.postOp = OP_R2C_FULL; b.realIn = true; b.outLen = (uint32_t)(N / 2 + 1); }
		else { b.preOp = OP_C2R_FULL; b.postOp = OP_C2R_FULL; b.realOut = true; b.swapIn = b.swapOut = true; b.inLen = (uint32_t)(N / 2 + 1); b.outLen = (uint32_t)N; }
	}
	b.inRole = inverse ? cplxRole : realRole;
	b.outRole = inverse ? realRole : cplxRole;
	b.label = inverse ? "c2r" : "r2c";
	if (blueM) {
		size_t chirpOff, bhatOff;
		make_bluestein_tables(N, blueM, dp, ar, chirpOff, bhatOff, true);
		b.L = blueM; b.blueN = (uint32_t)N;
		b.midOp = OP_BLUESTEIN_MID; b.auxOff2ForPre = chirpOff;
		b.fastKernel = KERNEL_POW2_BLUE_R2R; b.fastVariant = blueVariant; b.fastThreads = blueThr; b.forceT = (uint32_t)blueFpw;
		for (int k = 0; k < 4; k++) if (blueBits[k]) b.radices.push_back(1u << blueBits[k]);
	}
	PassPlan pp; int r = finish_pass(b, ar, pp); if (r) return r == 3002 ? 3003 : r;
	passes.push_back(pp);
	out.uploadsPerAxis[0] = 1;
	out.axisSplit[0][0] = N;
	return 0;
}

// DCT / DST of type 1..4 along one axis through a complex FFT with fused pre/post maps
// (reference: vkFFT_R2R.h, size rules vkFFT_Scheduler.h:2271-2280).
static int plan_r2r_axis_fused(const TransformDesc& d, int type, bool dst, uint64_t N, int64_t strideIn, int64_t strideOut, const std::vector<HostDim>& others,
                               int inRole, int outRole, double scale, int axisIndex, Arena& ar, DirectionPlan& out, std::vector<PassPlan>& passes);
static int plan_r2r_axis(const TransformDesc& d, int type, bool dst, uint64_t N, int64_t strideIn, int64_t strideOut, const std::vector<HostDim>& others,
                         int inRole, int outRole, double scale, int axisIndex, Arena& ar, DirectionPlan& out, std::vector<PassPlan>& passes) {
	const size_t mark = passes.size();
	const int r = plan_r2r_axis_fused(d, type, dst, N, strideIn, strideOut, others, inRole, outRole, scale, axisIndex, ar, out, passes);
	if (r != 3004 || type < 1 || type > 4 || N < 2) return r;
	// no fused form (the embedding length needs Bluestein on a strided axis or beyond the fused kernels' reach): the element-wise full-length
	// forms of the maps as separate passes around the complex plan of the embedding length
	passes.resize(mark);
	const bool dp = d.dp;
	const size_t es = dp ? 16 : 8;
	const int rb = dp ? 8 : 4;
	RealMapJob m;
	m.N = N; m.scale = scale; m.axisIndex = axisIndex; m.strideIn = strideIn; m.strideOut = strideOut; m.others = others;
	m.inRole = inRole; m.outRole = outRole; m.inElemBytes = m.outElemBytes = rb; m.outLen = (uint32_t)N;
	auto quarter = [&]() { const size_t a = ar.alloc(N * es); for (uint64_t k = 0; k < N; k++) ar.putc(a, k, unit_root(k, 4 * N), dp); return a; };
	switch (type) {
	case 1:
		if (!dst) { m.Lm = 2 * N - 2; m.preOp = OP_DCT1_PRE; m.postOp = OP_DCT1_POST; }
		else { m.Lm = 2 * N + 2; m.preOp = OP_DST1_PRE; m.postOp = OP_DST1_POST; }
		break;
	case 2: m.Lm = N; m.preOp = dst ? OP_DST2_PRE : OP_DCT2_PRE; m.postOp = dst ? OP_DST2_POST : OP_DCT2_POST; m.postAux = quarter(); break;
	case 3: m.Lm = N; m.preOp = dst ? OP_DST3_PRE : OP_DCT3_PRE; m.postOp = dst ? OP_DST3_POST : OP_DCT3_POST; m.preAux3 = quarter(); m.cinverse = true; break;
	default: { // type 4: odd lengths in the same-length form (no tables), even ones in the zero-padded 2N form; both maps are element-wise
		m.preOp = dst ? OP_DST4_PRE : OP_DCT4_PRE; m.postOp = dst ? OP_DST4_POST : OP_DCT4_POST;
		if ((N & 1) && N >= 3) { m.Lm = N; break; }
		m.Lm = 2 * N; m.preOp = dst ? OP_DST4_PRE : OP_DCT4_PRE; m.postOp = dst ? OP_DST4_POST : OP_DCT4_POST;
		m.preAux3 = quarter();
		const size_t a2 = ar.alloc(N * es);
		for (uint64_t n = 0; n < N; n++) ar.putc(a2, n, unit_root(2 * n + 1, 8 * N), dp);
		m.postAux2 = a2;
		break;
	}
	}
	const int r2 = plan_real_by_maps(d, m, ar, out, passes);
	return r2 == 3002 ? 3004 : r2;
}
static int plan_r2r_axis_fused(const TransformDesc& d, int type, bool dst, uint64_t N, int64_t strideIn, int64_t strideOut, const std::vector<HostDim>& others,
                               int inRole, int outRole, double scale, int axisIndex, Arena& ar, DirectionPlan& out, std::vector<PassPlan>& passes) {
	const bool dp = d.dp;
	const size_t es = dp ? 16 : 8;
	const uint32_t dmax = direct_max(d);
	PassBuild b;
	b.dp = dp; b.maxLds = d.maxLds; b.raderDirectMax = dmax; b.allowFast = false; b.allowOp = !d.disableFastKernels;
	b.opN = (uint32_t)N; b.scale = scale;
	b.realIn = b.realOut = true;
	b.inLen = b.outLen = (uint32_t)N;
	const bool unit = strideIn == 1 && strideOut == 1;
	b.inStrideJ = strideIn; b.outStrideJ = strideOut;
	b.colIn = b.colOut = !unit;
	b.dims = others;
	b.inRole = inRole; b.outRole = outRole;
	switch (type) {
	case 1:
		if (!dst) {
			if (N < 2) return 3004;
			int v, r5[5], f, t;
			if (!d.disableFastKernels && N >= 5 && opfft_lookup(N - 1, dp, !unit, false, OP_DCT1H_PRE, OP_DCT1H_POST, &v, r5, &f, &t)) {
				// half-length form on an ahead-of-time instance: complex FFT of N-1 points + the even R2C split (real parts only)
				const uint64_t H = N - 1;
				b.L = H; b.preOp = OP_DCT1H_PRE; b.postOp = OP_DCT1H_POST;
				size_t aux = ar.alloc((H + 1) * es);
				for (uint64_t k = 0; k <= H; k++) ar.putc(aux, k, unit_root(k, 2 * H), dp);
				b.auxOff = aux;
			} else { b.L = 2 * N - 2; b.preOp = OP_DCT1_PRE; b.postOp = OP_DCT1_POST; }
		}
		else { b.L = 2 * N + 2; b.preOp = OP_DST1_PRE; b.postOp = OP_DST1_POST; }
		break;
	case 2: case 3: if (N % 2 == 0 && N >= 4 && ![&]() {
			const uint64_t rows = pairable_rows(others);
			const uint32_t ph = type == 2 ? (dst ? OP_DST2H_PRE : OP_DCT2H_PRE) : (dst ? OP_DST3H_PRE : OP_DCT3H_PRE), qh = type == 2 ? (dst ? OP_DST2H_POST : OP_DCT2H_POST) : (dst ? OP_DST3H_POST : OP_DCT3H_POST);
			return prefer_full_length_pairs(d, N, unit, rows, ph, qh, N / 2);
		}()) {
		// even length: one complex FFT of length N/2 (Makhoul permutation + even R2C/C2R split + quarter-wave twiddle)
		const uint64_t H = N / 2;
		b.L = H;
		if (type == 2) { b.preOp = dst ? OP_DST2H_PRE : OP_DCT2H_PRE; b.postOp = dst ? OP_DST2H_POST : OP_DCT2H_POST; b.outLen = (uint32_t)(H + 1); }
		else { b.preOp = dst ? OP_DST3H_PRE : OP_DCT3H_PRE; b.postOp = dst ? OP_DST3H_POST : OP_DCT3H_POST; b.swapIn = b.swapOut = true; }
		size_t aux = ar.alloc(N * es), aux2 = ar.alloc((H + 1) * es);
		for (uint64_t k = 0; k < N; k++) ar.putc(aux, k, unit_root(k, 4 * N), dp);
		for (uint64_t k = 0; k <= H; k++) ar.putc(aux2, k, unit_root(k, N), dp);
		b.auxOff = aux; b.aux2Off = aux2;
		break;
	}
	if (type == 3) goto full3;
	{
		b.L = N; b.preOp = dst ? OP_DST2_PRE : OP_DCT2_PRE; b.postOp = dst ? OP_DST2_POST : OP_DCT2_POST;
		size_t aux = ar.alloc(N * es);
		for (uint64_t k = 0; k < N; k++) ar.putc(aux, k, unit_root(k, 4 * N), dp);
		b.auxOff = aux;
		break;
	}
	full3: {
		b.L = N; b.preOp = dst ? OP_DST3_PRE : OP_DCT3_PRE; b.postOp = dst ? OP_DST3_POST : OP_DCT3_POST;
		b.swapIn = b.swapOut = true;
		size_t aux = ar.alloc(N * es);
		for (uint64_t k = 0; k < N; k++) ar.putc(aux, k, unit_root(k, 4 * N), dp);
		b.auxOff = aux;
		break;
	}
	case 4: {
		b.preOp = dst ? OP_DST4_PRE : OP_DCT4_PRE; b.postOp = dst ? OP_DST4_POST : OP_DCT4_POST;
		if (N % 2 == 0) {
			b.L = N / 2;
			size_t aux = ar.alloc((N / 2) * es), aux2 = ar.alloc((N / 2) * es);
			for (uint64_t n = 0; n < N / 2; n++) { ar.putc(aux, n, unit_root(4 * n + 1, 8 * N), dp); ar.putc(aux2, n, unit_root(n, 2 * N), dp); }
			b.auxOff = aux; b.aux2Off = aux2;
		} else {
			// odd length: the same-length form (vkFFT_R2R.h:414-481, 922-972, 1032) — a signed permutation of the row, an N-point transform, one
			// rotation by a multiple of pi/4 per output (kernel_generic.h); no tables.  (Rounds 1-3 zero-padded to 2N complex points.)
			b.L = N;
			if (N < 3) { // a single point stays on the zero-padded two-point form
				b.L = 2 * N;
				size_t aux = ar.alloc(N * es), aux2 = ar.alloc(N * es);
				for (uint64_t n = 0; n < N; n++) { ar.putc(aux, n, unit_root(n, 4 * N), dp); ar.putc(aux2, n, unit_root(2 * n + 1, 8 * N), dp); }
				b.auxOff = aux; b.aux2Off = aux2;
			}
		}
		break;
	}
	default: return 3004;
	}
	b.label = dst ? "dst" : "dct";
	bool blueByChoice = false; // within the interpreter's reach, but the fused Bluestein kernel is the faster form (real_row_prefers_bluestein)
	if (is_supported_len(b.L, dmax) && unit && !d.disableFastKernels && real_row_prefers_bluestein(b.L, dp) && !getenv("VKFFT_MI355X_NO_REAL_BLUE_CHOICE")) {
		const uint64_t Lb = (type == 2 || type == 3) ? N : b.L;
		uint64_t Mp = 64; while (Mp < 2 * Lb - 1) Mp *= 2;
		const uint64_t rowPitch = others.empty() ? N : (uint64_t)std::max<int64_t>(std::llabs(others[0].inStride), std::llabs(others[0].outStride));
		const uint32_t pre = type == 2 ? (uint32_t)OP_DCT2_PRE : type == 3 ? (uint32_t)OP_DCT3_PRE : b.preOp;
		int v, bits[4], fpw, thr;
		blueByChoice = Mp <= (dp ? 4096u : 8192u) && (rowPitch * 64 + 2 * N) * (dp ? 8 : 4) < 0x7FFFFF00ull && pow2_blue_r2r_lookup(ilog2(Mp), dp, pre, &v, bits, &fpw, &thr);
	}
	if (!is_supported_len(b.L, dmax) || blueByChoice) {
		// the embedding length has a prime factor outside the radix / Rader stages (e.g. DST-I of 100: 202 = 2 * 101): the
		// real transform's maps around a fused Bluestein transform of the embedding length (kernel_blue_r2r.h), unit-stride rows
		if (!unit || d.disableFastKernels) return 3004;
		uint64_t Lb = b.L;
		if (type == 2 || type == 3) { // full-length forms (the half-length post-map is not element-wise)
			Lb = N;
			b.preOp = type == 2 ? (dst ? OP_DST2_PRE : OP_DCT2_PRE) : (dst ? OP_DST3_PRE : OP_DCT3_PRE);
			b.postOp = type == 2 ? (dst ? OP_DST2_POST : OP_DCT2_POST) : (dst ? OP_DST3_POST : OP_DCT3_POST);
			b.swapIn = b.swapOut = type == 3;
			b.outLen = (uint32_t)N;
			if (b.auxOff == (size_t)-1) { // quarter-wave table e^{-i pi k / 2N} (already there for even N)
				size_t aux = ar.alloc(N * es);
				for (uint64_t k = 0; k < N; k++) ar.putc(aux, k, unit_root(k, 4 * N), dp);
				b.auxOff = aux;
			}
		}
		uint64_t Mp = 64; while (Mp < 2 * Lb - 1) Mp *= 2;
		// The 16384-point instance with the DCT / DST-IV maps returns WRONG results on the device (relative error 0.5-0.9 at every odd length 4097 ... 8192 that reaches it;
		// the emulator is right, and the same 16384 points with the DCT-II / III / I and R2C maps are right on both — found by tools/scan_device_parity.py in round 6, not
		// root-caused: the instance is the one with 900 bytes of scratch).  Those lengths take the maps as passes around the complex plan instead (plan_real_by_maps).
		if (type == 4 && !dp && Mp > 8192) return 3004;
		int variant, bits[4], fpw, thr;
		const uint64_t rowPitch = others.empty() ? N : (uint64_t)std::max<int64_t>(std::llabs(others[0].inStride), std::llabs(others[0].outStride));
		if ((rowPitch * 64 + 2 * N) * (dp ? 8 : 4) >= 0x7FFFFF00ull || !pow2_blue_r2r_lookup(ilog2(Mp), dp, b.preOp, &variant, bits, &fpw, &thr)) return 3004;
		size_t chirpOff, bhatOff;
		make_bluestein_tables(Lb, Mp, dp, ar, chirpOff, bhatOff, true); // aux / aux2 stay with the real transform's own tables
		b.L = Mp; b.blueN = (uint32_t)Lb;
		b.midOp = OP_BLUESTEIN_MID; b.auxOff2ForPre = chirpOff;
		b.fastKernel = KERNEL_POW2_BLUE_R2R; b.fastVariant = variant; b.fastThreads = thr; b.forceT = (uint32_t)fpw;
		b.radices.clear();
		for (int k = 0; k < 4; k++) if (bits[k]) b.radices.push_back(1u << bits[k]);
	} else if (b.L > (unit ? max_row_len(dp, d.maxLds) : max_col_len(dp, d.maxLds, 1)) && !(unit && long_row_instance(d, b.L, dp))) {
		// longer than one pass: the full-length form of the real transform through a multi-pass (Four-Step) complex FFT of the
		// embedding length; the first load / last store apply the pre / post map to the row by natural index (the reference
		// lifts the same limit: vkFFT_Scheduler.h:2894-2897)
		if (!unit) return 3004;
		uint64_t Lm = b.L;
		MultiPassIO io;
		if (type == 2 || type == 3) {
			Lm = N;
			io.firstPre = type == 2 ? (dst ? OP_DST2_PRE : OP_DCT2_PRE) : (dst ? OP_DST3_PRE : OP_DCT3_PRE);
			io.lastPost = type == 2 ? (dst ? OP_DST2_POST : OP_DCT2_POST) : (dst ? OP_DST3_POST : OP_DCT3_POST);
			io.swapIn = io.swapOut = type == 3;
			if (b.auxOff == (size_t)-1) { size_t aux = ar.alloc(N * es); for (uint64_t k = 0; k < N; k++) ar.putc(aux, k, unit_root(k, 4 * N), dp); b.auxOff = aux; }
			io.firstAux = type == 3 ? b.auxOff : (size_t)-1; io.lastAux = b.auxOff;
		} else if (type == 4 && (N & 1) && N >= 3) { // odd length: same-length form, element-wise maps without tables
			Lm = N;
			io.firstPre = b.preOp; io.lastPost = b.postOp;
		} else if (type == 4) { // zero-padded 2N form: its maps are element-wise
			Lm = 2 * N;
			size_t aux = ar.alloc(N * es), aux2 = ar.alloc(N * es);
			for (uint64_t n = 0; n < N; n++) { ar.putc(aux, n, unit_root(n, 4 * N), dp); ar.putc(aux2, n, unit_root(2 * n + 1, 8 * N), dp); }
			io.firstPre = b.preOp; io.lastPost = b.postOp; io.firstAux = aux; io.lastAux = aux; io.lastAux2 = aux2;
		} else { io.firstPre = b.preOp; io.lastPost = b.postOp; }
		if (Lm >= (1ull << 31) || !is_supported_len(Lm, dmax)) return 3004;
		std::vector<uint64_t> sp;
		if (!choose_split(Lm, dp, d.maxLds, dmax, !d.disableFastKernels, sp)) return 3004;
		io.othersIn = others; io.othersOut = others;
		for (auto& o : io.othersIn) o.outStride = o.inStride;
		for (auto& o : io.othersOut) o.inStride = o.outStride;
		io.inRole = inRole; io.outRole = outRole; io.scale = scale;
		io.natural = true; io.firstRealIn = io.lastRealOut = true; io.natOutLen = (uint32_t)N; io.opN = (uint32_t)N; io.blueN = (uint32_t)Lm;
		PassBuild proto; proto.dp = dp; proto.maxLds = d.maxLds; proto.raderDirectMax = dmax; proto.allowFast = !d.disableFastKernels; proto.allowOp = !d.disableFastKernels;
		int r = emit_multipass(proto, Lm, sp, io, ar, passes); if (r) return r == 3002 ? 3004 : r;
		uint64_t nsub = 1; for (auto& o : others) nsub *= o.count;
		out.tempBytes = std::max<uint64_t>(out.tempBytes, nsub * Lm * es);
		out.uploadsPerAxis[axisIndex] = (uint32_t)sp.size();
		out.axisSplit[axisIndex][0] = N;
		return 0;
	}
	PassPlan pp; int r = finish_pass(b, ar, pp); if (r) return r == 3002 ? 3004 : r;
	passes.push_back(pp);
	out.uploadsPerAxis[axisIndex] = 1;
	out.axisSplit[axisIndex][0] = N;
	return 0;
}

// ---- merged convolution along the last axis ------------------------------------------------------------------
// One pass for an axis of 64 .. 1024 points (512 with a kernel matrix: three coordinate systems stay in registers).  Longer axes (two power-of-two
// factors n0 x M) take three passes instead of the five of "two forward passes, product, two inverse passes": the forward first pass A (FFT over n0,
// Four-Step twiddle, data -> scratch), the merged pass on the inner factor M (its outputs k1 of column k0 are the frequencies k0 + n0 k1 of the
// kernel) in place in the scratch, and pass A run backwards (scratch -> data) — the shape of the multi-pass Bluestein plan (MODE 1 / 2 / 3).
int build_conv_axis_plan(const TransformDesc& d, const ConvAxisDesc& c, DirectionPlan& out) {
	out = DirectionPlan();
	Arena ar(out.arena);
	const int nd = d.fftDim, a = nd - 1;
	if (nd < 2 || d.kind > 1 || d.omit[a] || d.padFrequency || d.disableFastKernels || d.inFormatted || d.outFormatted) return 3002;
	const uint64_t L = d.size[a];
	if ((L & (L - 1)) != 0 || c.coordinates > 3 || c.coordinates < 1 || (c.matrix > 1 && c.matrix != c.coordinates)) return 3002;
	const bool dp = d.dp;
	const uint64_t es = dp ? 16 : 8;
	const bool padded = d.padR[a] > d.padL[a];
	const uint64_t matrixCap = dp ? 256 : 512; // (beyond: 1024 threads per tile, 128 registers each — three systems do not fit)
	uint64_t n0 = 1, M = L;
	int mode = 6;
	{ int v, b4[4], t, th; if (c.matrix > 1 && L == 2 * matrixCap && pow2_col_blue_lookup(ilog2(L), dp, 7, &v, b4, &t, &th)) mode = 7; } // (the narrow-tile instance)
	if (L > 1024 || (c.matrix > 1 && L > matrixCap && mode == 6)) { // two factors, the inner one as large as a merged kernel allows
		if (padded) return 3002; // (the passes of a split axis address by factor, not by the natural index the padded range is given in)
		M = c.matrix > 1 ? matrixCap : 1024;
		while (M > 64 && L / M < 64) M >>= 1;
		n0 = L / M;
		if (n0 < 64 || n0 > 1024 || M < 64) return 3002;
	}
	int variant, bits[4], tc, thr;
	if (!pow2_col_blue_lookup(ilog2(M), dp, mode, &variant, bits, &tc, &thr)) return 3002;
	const int64_t strideJ = (int64_t)d.bufStride[a - 1], sys = (int64_t)d.bufStride[nd - 1];
	const uint64_t W = d.kind == 1 ? d.size[0] / 2 + 1 : d.size[0];
	// buffer addressing: a column tile and every kernel system behind it lie within one 2 GiB resource
	if (((uint64_t)L * (uint64_t)strideJ + 64) * es + c.kernelSystems * (uint64_t)sys * es >= 0x7FFFFF00ull) return 3002;
	// the other spatial dimensions (between axis 0 and the last axis), then the systems: coordinates (inside the merged kernel) and batches
	std::vector<HostDim> spatial; // strides in the data buffer
	for (int o = 1; o < a; o++) spatial.push_back({d.size[o], (int64_t)d.bufStride[o - 1], (int64_t)d.bufStride[o - 1]});
	uint64_t nother = 1; for (auto& h : spatial) nother *= h.count;
	const bool split = n0 > 1;
	// scratch layout of the split form: T[(batch * cf + coordinate)][other spatial, dense][k0 * M + m][x]
	const int64_t tRow = (int64_t)W, tSub = (int64_t)(L * W), tSys = (int64_t)(nother * L * W);
	PassPlan pp; memset(&pp.prm, 0, sizeof(pp.prm));
	{
		PassParams& q = pp.prm;
		q.L = (uint32_t)M;
		std::vector<HostDim> dims; std::vector<int64_t> kstr;
		dims.push_back({W, 1, 1}); kstr.push_back(1);
		if (!split) {
			q.inStrideJ = q.outStrideJ = strideJ; q.convKerStrideJ = strideJ; q.convSysStride = sys;
			for (auto& h : spatial) { dims.push_back(h); kstr.push_back(h.inStride); }
			dims.push_back({d.batch, (int64_t)c.coordinates * sys, (int64_t)c.coordinates * sys}); kstr.push_back(0);
		} else {
			q.inStrideJ = q.outStrideJ = tRow; q.convKerStrideJ = (int64_t)n0 * strideJ; q.convSysStride = tSys;
			dims.push_back({n0, (int64_t)M * tRow, (int64_t)M * tRow}); kstr.push_back(strideJ); // column k0: frequencies k0 + n0 k1
			int64_t run = tSub;
			for (auto& h : spatial) { dims.push_back({h.count, run, run}); kstr.push_back(h.inStride); run *= (int64_t)h.count; }
			dims.push_back({d.batch, (int64_t)c.coordinates * tSys, (int64_t)c.coordinates * tSys}); kstr.push_back(0);
		}
		while (dims.size() < 3) { dims.push_back({1, 0, 0}); kstr.push_back(0); }
		if (dims.size() > 3) { // only the batch may go to the host loop (the kernel pointer does not follow it)
			if (dims.size() > 4) return 3002;
			pp.hostLoop.push_back(dims.back()); dims.pop_back(); kstr.pop_back();
		}
		for (int i = 0; i < 3; i++) q.dim[i] = {(uint32_t)dims[i].count, dims[i].inStride, dims[i].outStride};
		q.convKerStride1 = kstr[1]; q.convKerStride2 = kstr[2]; q.convKerSysStride = sys;
		q.convM = c.matrix; q.convCf = c.coordinates; q.convSymmetric = c.symmetric; q.convConj = c.conjugate;
		q.tilesPerG0 = (uint32_t)((W + (uint64_t)tc - 1) / (uint64_t)tc);
		q.scale = split ? 1.0 : c.scale;
		if (padded) { q.padInL = q.padOutL = (uint32_t)d.padL[a]; q.padInN = q.padOutN = (uint32_t)(d.padR[a] - d.padL[a]); } // (spatial padding: read side of the forward half, write side of the inverse half)
		pp.lutOff = build_pow2_stage_lut(ar, bits, dp);
		pp.kernel = KERNEL_POW2_COL_BLUE; pp.variant = variant; pp.threads = (uint32_t)thr; pp.dp = dp; pp.auxIsKernel = true;
		pp.inRole = pp.outRole = split ? ROLE_TEMP : ROLE_BUFFER; pp.inElemBytes = pp.outElemBytes = (int)es;
		pp.label = "convolution";
	}
	if (!split) {
		out.passes.push_back(pp);
		out.uploadsPerAxis[a] = 1;
		return 0;
	}
	// ---- pass A: FFT over i0 for every (x, m), twiddle w_L^(k0 m), data -> scratch (the first pass of emit_multipass_strided)
	std::vector<HostDim> restData, restT; // (other spatial dims, systems): strides in the data buffer / in the scratch
	{ int64_t run = tSub; for (auto& h : spatial) { restData.push_back(h); restT.push_back({h.count, run, run}); run *= (int64_t)h.count; } }
	restData.push_back({d.batch * c.coordinates, sys, sys}); restT.push_back({d.batch * c.coordinates, tSys, tSys});
	{
		PassBuild b;
		b.dp = dp; b.maxLds = d.maxLds; b.allowFast = true; b.allowOp = true;
		b.L = n0; b.inStrideJ = (int64_t)M * strideJ; b.outStrideJ = (int64_t)M * tRow;
		b.colIn = b.colOut = true;
		b.dims = {{W, 1, 1}, {M, strideJ, tRow}};
		for (size_t i = 0; i < restData.size(); i++) b.dims.push_back({restData[i].count, restData[i].inStride, restT[i].inStride});
		b.postOp = OP_TWIDDLE_4STEP; b.fsN = L; b.fsColFromDim1 = true;
		b.inRole = ROLE_BUFFER; b.outRole = ROLE_TEMP; b.label = "convolution-A"; b.noCollapse = true;
		PassPlan pa; int r = finish_pass(b, ar, pa); if (r) return 3002;
		if (pa.kernel != KERNEL_POW2_COL) return 3002;
		out.passes.push_back(pa);
	}
	out.passes.push_back(pp);
	{ // ---- pass A backwards (pow2_col_blue_kernel MODE 8): scratch -> data, conj twiddle, inverse FFT over k0, normalisation
		int v8, b8[4], tc8, thr8;
		if (!pow2_col_blue_lookup(ilog2(n0), dp, 8, &v8, b8, &tc8, &thr8)) return 3002;
		PassPlan pc; memset(&pc.prm, 0, sizeof(pc.prm));
		PassParams& q = pc.prm;
		q.L = (uint32_t)n0; q.inStrideJ = (int64_t)M * tRow; q.outStrideJ = (int64_t)M * strideJ;
		std::vector<HostDim> dims = {{W, 1, 1}, {M, tRow, strideJ}};
		for (size_t i = 0; i < restData.size(); i++) dims.push_back({restData[i].count, restT[i].inStride, restData[i].inStride});
		{ // collapse what is beyond dim[1] where both sides allow, the rest goes to the host loop
			std::vector<HostDim> tail(dims.begin() + 2, dims.end());
			collapse_dims(tail);
			dims.resize(2); for (auto& h : tail) dims.push_back(h);
		}
		while (dims.size() < 3) dims.push_back({1, 0, 0});
		while (dims.size() > 3) { pc.hostLoop.push_back(dims.back()); dims.pop_back(); }
		for (int i = 0; i < 3; i++) q.dim[i] = {(uint32_t)dims[i].count, dims[i].inStride, dims[i].outStride};
		q.fsN = (uint32_t)L; q.fsColFromDim1 = 1; q.scale = c.scale;
		q.tilesPerG0 = (uint32_t)((W + (uint64_t)tc8 - 1) / (uint64_t)tc8);
		{ // two-level Four-Step table of w_L (as finish_pass builds it)
			const uint32_t lo = (ceil_log2(L) + 1) / 2;
			const uint64_t nlo = 1ull << lo, nhi = (L + nlo - 1) / nlo;
			const size_t off = ar.alloc((nlo + nhi) * es);
			for (uint64_t i = 0; i < nlo; i++) ar.putc(off, i, unit_root(i, L), dp);
			for (uint64_t i = 0; i < nhi; i++) ar.putc(off, nlo + i, unit_root(i * nlo, L), dp);
			q.fsLoBits = lo; pc.auxOff = off;
		}
		pc.lutOff = build_pow2_stage_lut(ar, b8, dp);
		pc.kernel = KERNEL_POW2_COL_BLUE; pc.variant = v8; pc.threads = (uint32_t)thr8; pc.dp = dp;
		pc.inRole = ROLE_TEMP; pc.outRole = ROLE_BUFFER; pc.inElemBytes = pc.outElemBytes = (int)es;
		pc.label = "convolution-A-back";
		out.passes.push_back(pc);
	}
	out.tempBytes = (uint64_t)d.batch * c.coordinates * (uint64_t)tSys * es;
	out.uploadsPerAxis[a] = 3;
	return 0;
}

// ---- top level ----------------------------------------------------------------------------------------
int build_direction_plan(const TransformDesc& d, DirectionPlan& out) {
	out = DirectionPlan();
	Arena ar(out.arena);
	const bool dp = d.dp;
	const int nd = d.fftDim;

	// source / destination roles and their strides
	int srcRole = ROLE_BUFFER, dstRole = ROLE_BUFFER;
	uint64_t sStr[5], dStr[5];
	for (int i = 0; i < 5; i++) { sStr[i] = d.bufStride[i]; dStr[i] = d.bufStride[i]; }
	if (!d.inverse) {
		if (d.inFormatted) { srcRole = ROLE_INPUT; for (int i = 0; i < 5; i++) sStr[i] = d.inStride[i]; }
		if (d.outFormatted) { dstRole = ROLE_OUTPUT; for (int i = 0; i < 5; i++) dStr[i] = d.outStride[i]; }
	} else {
		if (d.outFormatted) { srcRole = ROLE_OUTPUT; for (int i = 0; i < 5; i++) sStr[i] = d.outStride[i]; }
		if (d.inFormatted && d.inverseReturnToInput) { dstRole = ROLE_INPUT; for (int i = 0; i < 5; i++) dStr[i] = d.inStride[i]; }
	}

	// axis execution order (reference: forward 0..n-1, inverse n-1..0; vkFFT_RunApp.h:114-321, :469-648)
	std::vector<int> order;
	for (int a = 0; a < nd; a++) if (!d.omit[a] && d.size[a] > 1) order.push_back(a);
	if (d.inverse) std::reverse(order.begin(), order.end());
	if (order.empty()) return 0;

	// normalisation factor
	double scale = 1.0;
	if (d.inverse && d.normalize) {
		for (int a : order) {
			double n = (double)d.size[a];
			if (d.kind == 2 || d.kind == 3) {
				if (d.r2rType == 1) n = d.kind == 2 ? 2.0 * (n - 1) : 2.0 * (n + 1);
				else n = 2.0 * n;
			}
			scale /= n;
		}
	}

	// ---- zero padding (reference: vkFFT_Plan_FFT.h:522-560, vkFFT_Zeropad.h).  Along the padded axis itself the pass skips the range on its read side
	// (forward transform of a spatially padded system, inverse of a frequency-padded one) or on its write side (the other direction).  Sequences that
	// lie entirely inside the padded range of ANOTHER axis that is still to come (spatial: axes above this one; frequency: below) are not visited
	// at all — a tail range [left, size) simply shortens the enumeration of that axis.
	auto axisPadded = [&](int a) { return d.padR[a] > d.padL[a]; };
	auto applyPad = [&](int a, AxisJob& j) {
		if (!axisPadded(a)) return;
		const uint32_t L = (uint32_t)d.padL[a], N = (uint32_t)(d.padR[a] - d.padL[a]);
		if (d.inverse == d.padFrequency) { j.padInL = L; j.padInN = N; } else { j.padOutL = L; j.padOutN = N; }
	};
	auto rowsOf = [&](int a, int o, uint64_t count) -> uint64_t {
		if (!axisPadded(o) || d.padR[o] != d.size[o]) return count;
		return (d.padFrequency ? o < a : o > a) ? d.padL[o] : count;
	};
	auto plan_c2c_padded = [&](AxisJob& j) -> int {
		int r = plan_c2c_axis(d, j, ar, out, out.passes);
		if (r == kPadUnsupported) { // (a plan of several passes, Bluestein, ...): the range is zeroed ahead of the transform instead (api.cpp)
			out.padFallbackMask |= 1u << j.axisIndex;
			j.padInL = j.padInN = j.padOutL = j.padOutN = 0;
			r = plan_c2c_axis(d, j, ar, out, out.passes);
		}
		return r;
	};

	if (d.kind == 0) {
		for (size_t oi = 0; oi < order.size(); oi++) {
			const int a = order[oi];
			// the "mover" pass: forward -> first executed axis reads SRC writes DST; inverse -> last executed axis
			const bool mover = d.inverse ? (oi + 1 == order.size()) : (oi == 0);
			const bool before = d.inverse; // non-mover passes run in place on SRC (inverse) or DST (forward)
			int inRole, outRole; const uint64_t *is, *os;
			if (mover) { inRole = srcRole; outRole = dstRole; is = sStr; os = dStr; }
			else if (before) { inRole = outRole = srcRole; is = os = sStr; }
			else { inRole = outRole = dstRole; is = os = dStr; }
			AxisJob j;
			j.N = d.size[a]; j.dp = dp; j.inverse = d.inverse; j.axisIndex = a;
			j.inRole = inRole; j.outRole = outRole;
			j.inStrideJ = a == 0 ? 1 : (int64_t)is[a - 1];
			j.outStrideJ = a == 0 ? 1 : (int64_t)os[a - 1];
			j.scale = (oi + 1 == order.size()) ? scale : 1.0;
			// other dims, unit-stride axis 0 first (tiled for strided axes); for axis 0 the next axis comes first
			for (int o = 0; o < nd; o++) if (o != a) {
				HostDim h; h.count = rowsOf(a, o, d.size[o]);
				h.inStride = o == 0 ? 1 : (int64_t)is[o - 1];
				h.outStride = o == 0 ? 1 : (int64_t)os[o - 1];
				j.others.push_back(h);
			}
			j.others.push_back({d.batch, (int64_t)is[nd - 1], (int64_t)os[nd - 1]});
			applyPad(a, j);
			int r = plan_c2c_padded(j);
			if (r) return r;
		}
	} else if (d.kind == 1) {
		// R2C (forward) / C2R (inverse).  Real side: `input` when isInputFormatted, else the padded rows of `buffer`.
		if (d.outFormatted) return 3003;
		const bool realSeparate = d.inFormatted && (!d.inverse || d.inverseReturnToInput);
		const int realRole = realSeparate ? ROLE_INPUT : ROLE_BUFFER;
		uint64_t rStr[5]; // real-element strides of the real side
		for (int i = 0; i < 5; i++) rStr[i] = realSeparate ? d.inStride[i] : 2 * d.bufStride[i];
		const uint64_t W = d.size[0] / 2 + 1; // complex row width
		std::vector<HostDim> oReal, oCplx;
		for (int o = 1; o < nd; o++) { const uint64_t cnt = rowsOf(0, o, d.size[o]); oReal.push_back({cnt, (int64_t)rStr[o - 1], (int64_t)rStr[o - 1]}); oCplx.push_back({cnt, (int64_t)d.bufStride[o - 1], (int64_t)d.bufStride[o - 1]}); }
		oReal.push_back({d.batch, (int64_t)rStr[nd - 1], (int64_t)rStr[nd - 1]});
		oCplx.push_back({d.batch, (int64_t)d.bufStride[nd - 1], (int64_t)d.bufStride[nd - 1]});
		auto complexAxes = [&](bool inv) -> int {
			for (size_t oi = 0; oi < order.size(); oi++) {
				const int a = order[oi];
				if (a == 0) continue;
				AxisJob j; j.N = d.size[a]; j.dp = dp; j.inverse = inv; j.axisIndex = a;
				j.inRole = j.outRole = ROLE_BUFFER;
				j.inStrideJ = j.outStrideJ = (int64_t)d.bufStride[a - 1];
				j.scale = 1.0;
				for (int o = 0; o < nd; o++) if (o != a) {
					HostDim h; h.count = o == 0 ? W : rowsOf(a, o, d.size[o]);
					h.inStride = h.outStride = o == 0 ? 1 : (int64_t)d.bufStride[o - 1];
					j.others.push_back(h);
				}
				j.others.push_back({d.batch, (int64_t)d.bufStride[nd - 1], (int64_t)d.bufStride[nd - 1]});
				applyPad(a, j);
				int r = plan_c2c_padded(j);
				if (r) return r;
			}
			return 0;
		};
		if (d.omit[0] || d.size[0] < 2) return 3005;
		if (!d.inverse) {
			int r = plan_r2c_axis0(d, false, oReal, oCplx, realRole, ROLE_BUFFER, 1.0, ar, out, out.passes); if (r) return r;
			r = complexAxes(false); if (r) return r;
		} else {
			int r = complexAxes(true); if (r) return r;
			r = plan_r2c_axis0(d, true, oReal, oCplx, realRole, ROLE_BUFFER, scale, ar, out, out.passes); if (r) return r;
		}
	} else {
		// DCT / DST: the same R2R type along every axis; the inverse of type 2 is type 3 and vice versa
		int type = d.r2rType;
		if (d.inverse) type = type == 2 ? 3 : type == 3 ? 2 : type;
		const bool dst = d.kind == 3;
		for (size_t oi = 0; oi < order.size(); oi++) {
			const int a = order[oi];
			const bool mover = d.inverse ? (oi + 1 == order.size()) : (oi == 0);
			const bool before = d.inverse;
			int inRole, outRole; const uint64_t *is, *os;
			if (mover) { inRole = srcRole; outRole = dstRole; is = sStr; os = dStr; }
			else if (before) { inRole = outRole = srcRole; is = os = sStr; }
			else { inRole = outRole = dstRole; is = os = dStr; }
			std::vector<HostDim> others;
			for (int o = 0; o < nd; o++) if (o != a) others.push_back({rowsOf(a, o, d.size[o]), o == 0 ? 1 : (int64_t)is[o - 1], o == 0 ? 1 : (int64_t)os[o - 1]});
			others.push_back({d.batch, (int64_t)is[nd - 1], (int64_t)os[nd - 1]});
			if (axisPadded(a)) out.padFallbackMask |= 1u << a; // (the real transforms' maps do not skip elements: the range is zeroed ahead of the transform)
			int r = plan_r2r_axis(d, type, dst, d.size[a], a == 0 ? 1 : (int64_t)is[a - 1], a == 0 ? 1 : (int64_t)os[a - 1], others, inRole, outRole,
			                      (oi + 1 == order.size()) ? scale : 1.0, a, ar, out, out.passes);
			if (r) return r;
		}
	}

	if (d.userTempBytes && out.totalTemp() > d.userTempBytes) return 2016;
	return 0;
}

} // namespace vkfft_mi355x
