"""Which kernel family serves a unit-stride C2C row of length N (fp32)?  Uses plan creation only (CPU test double or the real library).
Lists the lengths of the reference's non-power-of-two sample lists (sample_14 / sample_18: :78-107, sample_7: :71-82) and the share of ALL
lengths <= 8192 that run on a hand-specialised kernel."""
import sys, os, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vkfft_amd import api
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = api.load_test_double(os.path.join(ROOT, "tests", "hostemu", "_build", "libvkfft_hostemu.so"))
import numpy as np
buf = np.zeros(1 << 16, np.complex64)

def family(N, dp=False):
    a = api.App([N], 1, dp=dp, buffer_ptr=buf.ctypes.data, lib=lib)
    n, dom = a.launch_info()
    a.delete()
    return n, dom

S14 = [3,5,6,7,9,10,11,12,13,14,15,17,19,21,22,23,24,25,26,27,28,29,30,31,33,35,37,39,41,43,42,44,45,47,49,52,53,55,56,59,60,61,65,66,67,71,73,79,81,83,89,97,
       121,125,137,143,169,191,243,286,343,383,429,509,572,625,720,1080,1001,1213,1287,1400,1440,1920,2160,2731,3024,3500,3840,4000,4050,4320,4391,7000,7680,7879,
       480,1280,2560,123,127,129,20000]
S7 = [17,19,23,29,31,37,41,43,47,53,59,61,67,71,73,79,83,89,97,179,283,419,547,661,811,947,1087,1229,1381,1523,2909,4241,6841,7727]
rep = collections.OrderedDict()
for name, lst in (("sample_14", S14), ("sample_7", S7)):
    gen = []
    for N in sorted(set(lst)):
        if N > 8192: continue
        n, dom = family(N)
        if "generic" in dom: gen.append(N)
    rep[name + "_lengths_on_generic_kernel"] = gen
cnt = collections.Counter()
for N in range(2, 8193):
    n, dom = family(N)
    cnt[dom.split("<")[0] if n == 1 else "multi-pass"] += 1
rep["all_lengths_2_to_8192_by_family"] = dict(cnt)
rep["share_on_hand_specialised_kernels"] = round(1 - (cnt.get("generic_pass_kernel", 0) + cnt.get("multi-pass", 0)) / 8191.0, 4)
rep["library_bytes"] = os.path.getsize(os.path.join(ROOT, "vkfft_amd", "lib", "libvkfft_mi355x.so"))
print(json.dumps(rep, indent=1))
