"""Development tool: short real rows on the fused-map kernels (kernel_opfft.h) vs the instance transform between the interpreter's maps
(VKFFT_MI355X_MIXED_OPS_MAX = longest complex length that prefers the latter)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ["NO_REF"] = "1"
from perf_configs import run
for kind in (1, 12, 13, 14):
    for n in (8, 16, 32, 40, 64, 100, 128, 200, 256, 400, 512, 1000, 1024):
        out = {"kind": kind, "n": n}
        for lim in ("0", "4096"):
            os.environ["VKFFT_MI355X_MIXED_OPS_MAX"] = lim
            try:
                out["opfft" if lim == "0" else "mixed_ops"] = run(kind, (n,), False, total_log2=25)["alg_GBps"]
            except Exception as e:
                out["err"] = str(e)
        print(json.dumps(out), flush=True)
