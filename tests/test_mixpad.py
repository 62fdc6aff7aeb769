"""Host-side check of the compile-time LDS padding chooser (vkfft_amd/csrc/mix_sched.h, MixPad): built with g++ as a plain
C++17 program — the chosen shift per exchange must have the minimum modelled conflict count, stride-16 exchanges must be
padded and odd-stride exchanges must not."""
import os, subprocess, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#define __host__
#define __device__
#include <cstdint>
#include <cstdio>
namespace vkfft_mi355x {}
#include "mix_sched.h"
using namespace vkfft_mi355x;
template <typename S, int TPF> int check() {
	using P = MixPad<S, TPF, 8>;
	int bad = 0;
	for (int e = 0; e + 1 < S::NS; e++) {
		const int chosen = P::shift(e);
		for (int sh : {0, 3, 4, 5}) if (P::cost(e, sh) < P::cost(e, chosen)) bad++;
	}
	if (P::elems() < S::N + 1) bad++;
	return bad;
}
int main() {
	int bad = 0;
	bad += check<MixSched<13, 13, 13, 1, 1>, 169>();
	bad += check<MixSched<10, 10, 8, 5, 1>, 400>();
	bad += check<MixSched<16, 16, 8, 1, 1>, 128>();
	bad += check<MixSched<25, 25, 5, 1, 1>, 209>();
	// stride 16 (radix-16 first stage) needs padding; stride 13 is conflict-free as it is
	if (MixPad<MixSched<16, 16, 8, 1, 1>, 128, 8>::shift(0) == 0) bad++;
	if (MixPad<MixSched<13, 13, 13, 1, 1>, 169, 8>::shift(0) != 0) bad++;
	printf("%d\n", bad);
	return bad;
}
'''


def test_mixpad_chooser_minimises_modelled_conflicts():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "pad.cpp"); exe = os.path.join(d, "pad")
        open(src, "w").write(SRC)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-DVKFFT_HOSTEMU", "-I" + os.path.join(ROOT, "vkfft_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "tests", "hostemu"), "-I/opt/rocm/include", "-o", exe, src])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.strip() == "0", out.stdout + out.stderr
