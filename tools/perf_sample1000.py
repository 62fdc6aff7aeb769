"""A sampling of the reference's sample 1000 (1-D C2C of EVERY length 2 … 4096), reference in the same process: python tools/perf_sample1000.py [count] [seed]"""
import sys, os, json, random
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_configs import run
cnt = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
sizes = sorted(set([rnd.randint(2, 4096) for _ in range(cnt)] + [17, 97, 127, 251, 509, 1021, 2039, 4093, 289, 1369, 3721, 221, 2021, 3599]))
for n in sizes:
    print(json.dumps(run(0, (n,), False, total_log2=25)), flush=True)
