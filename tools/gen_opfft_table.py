#!/usr/bin/env python3
"""Generates vkfft_amd/csrc/opfft_table_{f32,f64}_{row,col}_{0,1}.inc: the curated instances of kernel_opfft.h (single-pass FFT
with a fused pre/post map: R2C/C2R even split, DCT/DST I-IV, strided C2C of non-power-of-two length).  One line per
(complex FFT length L, precision, row|col, op family): radix list, threads per FFT, FFTs (rows or columns) per workgroup.
Re-run after changing the heuristics; the generated files are committed."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_mixed_table import best_radices, DIRECT_PRIMES

POW2 = [1 << k for k in range(2, 14)]  # complex FFT lengths up to 8192 (fp32) / 4096 (fp64): R2C rows up to 16384 / 8192 reals
# complex FFT lengths L of popular non-power-of-two real sizes (N = L for DCT-II/III, N = 2L for R2C / DCT-IV)
NONPOW2 = [6, 10, 12, 20, 24, 30, 36, 40, 48, 50, 60, 72, 80, 90, 96, 100, 108, 120, 125, 144, 150, 160, 180, 192, 200, 216, 240, 243, 250, 270, 300, 320, 324, 343, 350, 360, 384, 400,
           432, 450, 480, 500, 540, 600, 625,
           640, 720, 729, 768, 800, 960, 1000, 1080, 1200, 1280, 1440, 1536, 1600, 1920, 2000, 2160, 2187, 2400, 2560, 3000, 3072, 3125, 3840, 4000]
NONPOW2_DP = [12, 24, 48, 60, 96, 100, 120, 150, 192, 200, 240, 300, 350, 360, 384, 480, 500, 540, 600, 720, 768, 960, 1000, 1080, 1200, 1536, 1920, 2000]

# family -> (pre, post, real data?, row?, col?, pow2 only?)
FAMILIES = {
    "r2c": ("OP_NONE", "OP_R2C_EVEN_POST", False, True, False, False),
    "c2r": ("OP_C2R_EVEN_PRE", "OP_NONE", False, True, False, False),
    "dct2": ("OP_DCT2_PRE", "OP_DCT2_POST", True, True, True, False),      # full-length form: odd N only
    "dct3": ("OP_DCT3_PRE", "OP_DCT3_POST", True, True, True, False),
    "dct2h": ("OP_DCT2H_PRE", "OP_DCT2H_POST", True, True, True, False),   # even N through N/2 complex points
    "dct3h": ("OP_DCT3H_PRE", "OP_DCT3H_POST", True, True, True, False),
    "dct4": ("OP_DCT4_PRE", "OP_DCT4_POST", True, True, True, False),
    "dct1h": ("OP_DCT1H_PRE", "OP_DCT1H_POST", True, True, True, True),    # DCT-I of N = L + 1 through L complex points
    "dst1": ("OP_DST1_PRE", "OP_DST1_POST", True, True, True, True),
    "r2cf": ("OP_R2C_FULL", "OP_R2C_FULL", False, True, False, False),    # odd real rows: full-length "callback" form
    "c2rf": ("OP_C2R_FULL", "OP_C2R_FULL", False, True, False, False),
    "c2c": ("OP_NONE", "OP_NONE", False, False, True, False),
    "c2c4": ("OP_NONE", "OP_TWIDDLE_4STEP", False, False, True, False),   # middle Four-Step pass: column FFT + twiddle, in place
    "c2cT": ("OP_NONE", "OP_TWIDDLE_4STEP", False, False, True, False),   # first Four-Step pass: column FFT + twiddle + transposed store
}
# factor lengths of multi-pass plans of prime-power sizes (3^k, 5^k, 7^k, 11^k, 13^k: BASELINE config 3)
FOURSTEP_EXTRA = [9, 27, 81, 25, 49, 11, 121, 1331, 13, 169, 2197]
# odd real-row lengths (R2C / C2R of odd N run a complex FFT of the full length N)
ODD_EXTRA = [15, 25, 27, 45, 75, 81, 105, 135, 225, 315, 375, 405, 675, 945, 1125, 1215, 2025, 3375]


def smooth13(n):
    for q in (2, 3, 5, 7, 11, 13):
        while n % q == 0: n //= q
    return n == 1


# strided C2C (plain column passes: the second / third axis of planes and volumes, incl. the complex side of multi-dimensional real transforms): every
# 13-smooth length up to 1024 (fp64: 512) — round 3: such axes of a length outside the curated list ran on the interpreter
COL_C2C_EXTRA = [n for n in range(3, 1025) if smooth13(n) and n & (n - 1)]
COL_C2C_EXTRA_DP = [n for n in COL_C2C_EXTRA if n <= 512]


def smooth7(n):
    for q in (2, 3, 5, 7):
        while n % q == 0: n //= q
    return n == 1


# real rows of SHORT arbitrary 7-smooth lengths (fp32): every complex length up to 256 for the even forms (R2C / C2R / DCT-II/III half-length / DCT-IV of
# N = 2L reals) and every odd one for the full-length forms — the lengths outside the curated list run between the interpreter's generic maps at about half the speed
ROW_REAL_EXTRA = [n for n in range(3, 257) if smooth7(n) and n & (n - 1)]


def pitch(n, fpw, col):
    p = n + (n >> 4) + 1
    if col:
        q = 1 if fpw >= 32 else 32 // fpw
        while p % (2 * q) != q:
            p += 1
    return p


def tpf_for(n, rad, emax):
    return max(math.ceil((n // r) / max(1, emax // r)) for r in rad)


def plan_row(n, dp, pair=None):
    """pair = "first" / "last": the half-length DCT-II (DCT-III) moves its rows with 16-byte accesses when the first (last)
    stage has an even radix and an even number of butterflies per thread (kernel_opfft.h, opfft_can_pair): put the smallest
    even radix there and pick the threads per FFT accordingly."""
    rad = best_radices(n)
    es = 16 if dp else 8
    tpf = tpf_for(n, rad, 16)
    evens = [r for r in rad if r % 2 == 0]
    if pair and evens:
        rp = min(evens)
        rest = list(rad); rest.remove(rp)
        rad = [rp] + rest if pair == "first" else rest + [rp]
        nbp = n // rp
        if nbp % (2 * tpf) != 0 and nbp % 2 == 0:
            cand = nbp // 2
            if max(math.ceil((n // r) / cand) * r for r in rad) <= (32 if not dp else 24):
                tpf = cand
    if pair == "both" and len(rad) >= 1:
        # half-length DCT-IV: mirrored butterfly pairs in the first AND the last stage (8-byte accesses, kernel_opfft.h pairIn4/pairOut4):
        # even radices at both ends and a thread count that halves both butterfly counts
        evens = sorted(r for r in rad if r % 2 == 0)
        if len(evens) >= (2 if len(rad) > 1 else 1):
            rest = list(rad)
            r0 = evens[0]; rest.remove(r0)
            if len(rad) > 1:
                rl = min(r for r in rest if r % 2 == 0); rest.remove(rl)
                order = [r0] + sorted(rest, reverse=True) + [rl]
            else:
                order = [r0]
            nb0, nbl = n // order[0], n // order[-1]
            import math as _m
            g = _m.gcd(nb0 // 2 if nb0 % 2 == 0 else 0, nbl // 2 if nbl % 2 == 0 else 0)
            cands = [t for t in range(1, g + 1) if g % t == 0 and max(_m.ceil((n // r) / t) * r for r in order) <= (32 if not dp else 24)] if g else []
            if cands:
                rad = order
                tpf = min(cands)  # fewest threads that keep every stage within the register budget... smallest admissible divisor
                tpf = max(c for c in cands if c <= max(tpf_for(n, rad, 16), min(cands)))
    if n <= 64:
        # short rows: with one or two threads per FFT the lanes of a wave are a whole row apart (dozens of cache lines per access, measured
        # 1.0-1.6 TB/s on 64-point real rows); eight threads per FFT read 64-byte runs
        tpf = max(tpf, min(8, max(n // r for r in rad)))
    if tpf > 1024:
        return None
    lds_per = pitch(n, 1, False) * es
    best = None
    for f in range(1, 65):
        thr = tpf * f
        if thr > 512 and f > 1: break
        if f * lds_per > 64 * 1024 and f > 1: break
        score = (thr % 64 != 0, abs(thr - 256))
        if best is None or score < best[0]: best = (score, f)
    return rad, tpf, best[1]


def plan_col(n, dp, real):
    rad = best_radices(n)
    es = 16 if dp else 8
    cands = ([64, 32, 16, 8] if real else [32, 16, 8]) if not dp else ([32, 16, 8] if real else [16, 8])
    def fits(fpw, budget, emax):
        tpf = tpf_for(n, rad, emax)
        return fpw * pitch(n, fpw, True) * es <= budget and tpf * fpw <= 1024, tpf
    # 4-byte reals: 32 columns are the first width with 128-byte segments (64-byte segments measured ~40 % slower), worth
    # one workgroup per CU; complex data and fp64 reals reach 128 bytes at 16 columns
    tiers = ((80 * 1024, 32), (156 * 1024, 32), (80 * 1024, 16), (156 * 1024, 8)) if (real and not dp) else ((80 * 1024, 16), (156 * 1024, 8))
    for budget, minf in tiers:
        for fpw in cands:
            if fpw < minf: continue
            for emax in ((16, 32) if not dp else (16,)):
                ok, tpf = fits(fpw, budget, emax)
                if ok:
                    # do not leave most of the threads idle on tiny FFTs: at least 64 threads
                    while tpf * fpw < 64 and fpw < 64: fpw *= 2
                    return rad, tpf, fpw
    return None


def main():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vkfft_amd", "csrc")
    total = 0
    for dp, tname, tag in ((False, "float", "f32"), (True, "double", "f64")):
        lens = POW2 + (NONPOW2_DP if dp else NONPOW2)
        for col in (False, True):
            lines = ["// GENERATED by tools/gen_opfft_table.py — do not edit.  VKFFT_OPX(type, dp, col, pre, post, R0..R4, threads per FFT, FFTs per workgroup)"]
            for fam, (pre, post, real, row_ok, col_ok, pow2only) in FAMILIES.items():
                if (col and not col_ok) or (not col and not row_ok): continue
                fourstep = fam in ("c2c", "c2c4", "c2cT")
                oddreal = fam in ("r2cf", "c2rf")
                for n in sorted(set(lens) | (set(FOURSTEP_EXTRA) if fourstep and not dp else set()) | (set(ODD_EXTRA) if oddreal else set()) | (set(DIRECT_PRIMES) if fam == "c2c" else set())
                                | (set(COL_C2C_EXTRA_DP if dp else COL_C2C_EXTRA) if (fam == "c2c" and col) else set())
                                | (set(ROW_REAL_EXTRA) if (not col and not dp and fam in ("r2c", "c2r", "dct2h", "dct3h", "dct4", "r2cf", "c2rf", "dct2", "dct3")) else set())
                                # ... and the DCT / DST families on strided axes (planes and volumes of such lengths), complex lengths up to 128
                                | (set(n for n in ROW_REAL_EXTRA if n <= 128) if (col and not dp and fam in ("dct2h", "dct3h", "dct4", "dct2", "dct3")) else set())):
                    ispow2 = n & (n - 1) == 0
                    if oddreal and n % 2 == 0: continue
                    if pow2only and not ispow2: continue
                    if dp and n > 4096: continue
                    if fourstep and ispow2 and n <= 1024: continue  # pow2_col_kernel covers these
                    if fam in ("dct2", "dct3") and n % 2 == 0: continue  # even lengths take the half-length form
                    r = plan_col(n, dp, real) if col else plan_row(n, dp, {"dct2h": "first", "dct3h": "last", "dct4": "both"}.get(fam))
                    if r is None: continue
                    rad, tpf, fpw = r
                    rr = rad + [1] * (5 - len(rad))
                    lines.append("VKFFT_OPX(%s, %s, %s, %s, %s, %d, %d, %d, %d, %d, %d, %d, %s) // %s L=%d" %
                                 (tname, "true" if dp else "false", "true" if col else "false", pre, post, *rr, tpf, fpw, "true" if fam == "c2cT" else "false", fam, n))
                    total += 1
            # two translation units per table (kernels_opfft_*_{0,1}.hip): halves the longest compile of a parallel build
            head, body = lines[0], lines[1:]
            for half in (0, 1):
                open(os.path.join(root, "opfft_table_%s_%s_%d.inc" % (tag, "col" if col else "row", half)), "w").write("\n".join([head] + body[half::2]) + "\n")
    print("wrote", total, "entries")


if __name__ == "__main__":
    main()
