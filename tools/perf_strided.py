"""2-D C2C (256 x N): the strided axis of length N with the reference in the same process: python tools/perf_strided.py N [N ...]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_configs import run
for n in [int(a) for a in sys.argv[1:]]:
    try: print(json.dumps(run(0, (256, n), False, total_log2=25)), flush=True)
    except Exception as ex: print(json.dumps(dict(shape=[256, n], error=str(ex))), flush=True)
