"""Short unit-stride rows (N = 4 … 128) and small cubes, this library and the reference in the same process: python tools/perf_small_rows.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_configs import run
for kind, shape, dp in [(0, (4,), False), (0, (8,), False), (0, (16,), False), (0, (32,), False), (0, (64,), False), (0, (128,), False), (0, (16,), True), (0, (64,), True), (0, (6,), False), (0, (12,), False), (0, (15,), False), (0, (30,), False), (0, (48,), False), (0, (63,), False), (0, (100,), False), (0, (120,), False), (0, (15,), True), (0, (12, 12, 12), False),
                        (0, (16, 16, 16), False), (0, (32, 32), False), (1, (64, 64), False), (1, (32, 32, 32), False), (12, (64, 64, 64), False)]:
    print(json.dumps(run(kind, shape, dp, total_log2=26)), flush=True)
