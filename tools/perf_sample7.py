"""The reference's sample 7 (Bluestein / Rader benchmark: prime x prime and prime^3 systems, sample_7_benchmark_VkFFT_single_Bluestein.cpp:71-76),
this library and the reference VkFFT-HIP in the same process.  usage: python tools/perf_sample7.py [every k-th case]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_configs import run
step = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = [(p, p) for p in (17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97)]
cases += [(p, p, p) for p in (17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97)]
cases += [(p, p) for p in (179, 283, 419, 547, 661, 811, 947, 1087, 1229, 1381, 1523, 2909, 4241, 6841, 7727)]
for shape in cases[::step]:
    print(json.dumps(run(0, shape, False, total_log2=25)), flush=True)
