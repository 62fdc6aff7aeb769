"""2-D / 3-D systems whose axis lengths are smooth but outside the curated lists (development tool; reference in the same process)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_configs import run
for kind, shape in ((0, (84, 84)), (0, (168, 168)), (0, (252, 252)), (0, (84, 84, 84)), (0, (260, 140)), (1, (84, 84)), (1, (168, 168)), (1, (126, 126, 126)), (12, (84, 84)), (0, (1001, 91))):
    try:
        print(json.dumps(run(kind, shape, False, total_log2=25)), flush=True)
    except Exception as e:
        print(json.dumps({"shape": list(shape), "error": str(e)}), flush=True)
