#!/usr/bin/env python3
"""Generates vkfft_amd/csrc/mixed_table_{0..5}.inc: the curated list of non-power-of-two lengths that get a hand-specialised
ahead-of-time kernel (kernel_mixed.h).  For every length: radix list (fewest stages, radices <= 16), threads per FFT,
FFTs per workgroup.  Re-run after changing the heuristics; the generated file is committed."""
import itertools, math, os, sys

ALLOWED = [16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2]
# primes whose butterfly is the direct symmetric form in registers ((p-1)^2 FMAs per transform: cheaper than Bluestein on 64 padded points, and
# the reference's sample 7 is made of such axes); a thread owns a whole transform.  From 37 on the butterfly does not fit the register file
# (measured: 37^3 and 47^2 at 0.03 TB/s, all scratch traffic) — those stay on the Bluestein kernels
DIRECT_PRIMES = [17, 19, 23, 29, 31]

# measured: the composite radix 25 = 5*5 pays off where it removes two stages (5^5: 5 -> 3 stages, +13 %), not in
# mixed lengths (4000 = 25*16*10: -2 %); radix 27 = 9*3 was slower everywhere (3^7: -11 %) and is not used
def allowed_for(n):
    m = n
    while m % 5 == 0: m //= 5
    return ([25] if m == 1 and n >= 125 else []) + [p for p in reversed(DIRECT_PRIMES) if n % p == 0] + ALLOWED

def best_radices(n, max_stages=5):
    best = None
    def rec(rem, seq):
        nonlocal best
        if rem == 1:
            key = (len(seq), -min(seq), tuple(sorted(seq, reverse=True)))
            if best is None or key < best[0]:
                best = (key, sorted(seq, reverse=True))
            return
        if len(seq) >= max_stages: return
        if best is not None and len(seq) + 1 > best[0][0]: return
        for r in allowed_for(n):
            if (not seq or r <= seq[-1]) and rem % r == 0:
                rec(rem // r, seq + [r])
    rec(n, [])
    return best[1] if best else None

def plan(n, dp):
    rad = best_radices(n)
    if rad is None: return None
    emax = 16
    # threads per FFT: every stage's butterflies fit floor(emax/R) per thread
    tpf = max(math.ceil((n // r) / max(1, emax // r)) for r in rad)
    # prefer a thread count that divides the butterfly counts well (less predication): try a few larger candidates
    cands = [t for t in range(tpf, min(int(1.3 * tpf) + 1, 1024) + 1)]
    def waste(t):
        w = 0
        for r in rad:
            nb = n // r; p = math.ceil(nb / t)
            w += (p * t - nb) * r
        regs = max(math.ceil((n // r) / t) * r for r in rad)
        return (regs > 20, w / (t * len(rad)), t)
    tpf = min(cands, key=waste)
    if tpf > 1024: return None
    es = 16 if dp else 8
    lds_per = (n + n // 16 + 1) * es
    if lds_per > 150 * 1024: return None
    # FFTs per workgroup: ~256 threads, LDS <= 64 KiB, threads multiple of 64 where possible
    fpw = 1
    best = None
    for f in range(1, 65):
        thr = tpf * f
        if thr > 512 and f > 1: break
        if f * lds_per > 64 * 1024 and f > 1: break
        score = (thr % 64 != 0, abs(thr - 256))
        if best is None or score < best[0]: best = (score, f)
    fpw = best[1]
    return rad, tpf, fpw

def smooth(n, primes):
    for p in primes:
        while n % p == 0: n //= p
    return n == 1

def sizes():
    s = set()
    for n in range(3, 8193):
        if n & (n - 1) == 0: continue
        if n <= 4096 and smooth(n, (2, 3, 5, 7)):
            # keep lengths whose odd part is modest (common FFT sizes), all pure prime powers, and every length <= 512
            e7 = 0; m = n
            while m % 7 == 0: m //= 7; e7 += 1
            if n <= 512 or e7 <= 1 or smooth(n, (7,)) or smooth(n, (2, 7)): s.add(n)
        elif n > 4096 and smooth(n, (2, 3, 5)) and (n % 16 == 0 or smooth(n, (3,)) or smooth(n, (5,)) or n % 1000 == 0): s.add(n)
    # every 13-smooth length up to 1024 (small kernels, cheap to build) and the lengths of the reference's own non-power-of-two
    # sample lists (sample_14_precision_VkFFT_single_nonPow2.cpp:78-107) that the rules above miss
    for n in range(3, 1025):
        if n & (n - 1) and smooth(n, (2, 3, 5, 7, 11, 13)): s.add(n)
    s.update([1001, 1287, 7000])
    s.update(DIRECT_PRIMES)
    s.add(2)  # (the power-of-two row kernels start at 4 points)
    s.update([4, 8, 16, 32, 64, 128])  # short power-of-two lengths: only for real transforms between the interpreter's maps (mixed_row_kernel OPS = 1)
    # lengths whose largest prime factor is 17 .. 31 (the direct butterflies as radices of a mixed schedule): up to 4096 (fp64: 1024, see main)
    for n in range(34, 4097):
        m = n
        for q in (2, 3, 5, 7, 11, 13) + tuple(DIRECT_PRIMES):
            while m % q == 0: m //= q
        if m == 1 and not smooth(n, (2, 3, 5, 7, 11, 13)): s.add(n)
    # every 13-smooth length up to 4096 (fp64: up to 2048, see main): the alternative for such a length is the fused Bluestein kernel on twice
    # to four times the points (1092 = 4*3*7*13 ran at 1.2 TB/s against the reference's 5.1)
    for n in range(1025, 4097):
        if n & (n - 1) and smooth(n, (2, 3, 5, 7, 11, 13)): s.add(n)
    for p in (11, 13):
        k = p
        while k <= 4096:
            m = k
            while m <= 4096: s.add(m); m *= 2
            k *= p
    return sorted(s)

if __name__ == "__main__":
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vkfft_amd", "csrc")
    head = "// GENERATED by tools/gen_mixed_table.py — do not edit.  VKFFT_MX(type, dp, R0..R4, threads per FFT, FFTs per workgroup)"
    lines = []
    limit = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    for dp, tname in ((False, "float"), (True, "double")):
        cnt = 0
        for n in sizes():
            if cnt >= limit: break
            if dp and n > 4096: continue
            if dp and n > 2048 and not (smooth(n, (2, 3, 5, 7)) or smooth(n, (2, 11)) or smooth(n, (2, 13))): continue
            if dp and n > 1024 and not smooth(n, (2, 3, 5, 7, 11, 13)): continue
            r = plan(n, dp)
            if r is None: continue
            rad, tpf, fpw = r
            rr = rad + [1] * (5 - len(rad))
            lines.append("VKFFT_MX(%s, %s, %d, %d, %d, %d, %d, %d, %d) // N=%d" % (tname, "true" if dp else "false", *rr, tpf, fpw, n))
            cnt += 1
    for part in range(6):  # six translation units (kernels_mixed_{0..5}.hip), entries dealt round-robin
        open(os.path.join(root, "mixed_table_%d.inc" % part), "w").write("\n".join([head] + lines[part::6]) + "\n")
    print("wrote", len(lines), "entries")
