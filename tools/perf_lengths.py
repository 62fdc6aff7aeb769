"""1-D rows of the given lengths with the reference in the same process: python tools/perf_lengths.py [kind=0] N [N ...]   (kind: 0 C2C, 1 R2C, 12 DCT-II, 14 DCT-IV ...; env DP=1: fp64)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_configs import run
args = sys.argv[1:]
kind = 0
if args and args[0].startswith("kind="):
    kind = int(args[0][5:]); args = args[1:]
dp = bool(int(os.environ.get("DP", "0")))
for n in [int(a) for a in args]:
    try:
        print(json.dumps(run(kind, (n,), dp, total_log2=25)), flush=True)
    except Exception as ex:
        print(json.dumps(dict(kind=kind, shape=[n], error=str(ex))), flush=True)
