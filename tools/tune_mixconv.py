"""Development tool: the one-kernel cyclic convolution family (kernel_mixconv.h) against the power-of-two Bluestein kernels on the same lengths —
VKFFT_MI355X_MIXCONV=0 (family off) vs =2 (always preferred) — to calibrate the planner's cost factor (planner.cpp kMixConvCost).
usage: python tools/tune_mixconv.py [rows|cols]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ["NO_REF"] = "1"
from perf_configs import run
what = sys.argv[1] if len(sys.argv) > 1 else "rows"
rows = [37, 41, 47, 59, 67, 73, 83, 97, 127, 131, 179, 257, 263, 283, 419, 521, 547, 641, 661, 811, 947, 1009, 1031, 1046, 1087, 1229, 1381, 1523, 2053, 2311, 2909, 3001,
        3343, 4093, 4099, 4241, 5003, 6841, 7727, 8191, 10141, 12289, 13313]
cols = [(64, 37), (64, 47), (64, 67), (64, 97), (128, 179), (128, 257), (128, 283), (128, 419), (128, 547), (256, 661), (256, 811), (256, 947), (256, 1009)]
for item in (rows if what == "rows" else cols):
    shape = (item,) if what == "rows" else item
    out = {"shape": list(shape)}
    for mode in ("0", "2"):
        os.environ["VKFFT_MI355X_MIXCONV"] = mode
        r = run(0, shape, False, total_log2=25)
        out["off_GBps" if mode == "0" else "on_GBps"] = r["alg_GBps"]
    out["gain"] = round(out["on_GBps"] / out["off_GBps"], 2)
    N = shape[-1]
    Mp = 64
    while Mp < 2 * N - 1: Mp *= 2
    out["pow2_M"] = Mp
    print(json.dumps(out), flush=True)
