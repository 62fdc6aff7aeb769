"""1-D C2C fp32 rows of EVERY length lo..hi beside the reference in the same process (development tool: finds the instances whose heuristics are off)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_configs import run
lo, hi = int(sys.argv[1]), int(sys.argv[2])
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # 0 C2C, 1 R2C, 12 DCT-II ...
step = int(sys.argv[4]) if len(sys.argv) > 4 else 1
for n in range(lo, hi + 1, step):
    try:
        print(json.dumps(run(kind, (n,), False, total_log2=25)), flush=True)
    except Exception as e:
        print(json.dumps({"shape": [n], "error": str(e)}), flush=True)
