"""Development tool: real rows of chosen lengths next to the complex rows of the same transform length, and the reference in the same process
(python tools/perf_real_rows.py [kind:N ...], kind 0 c2c, 1 r2c, 12 dct2, 13 dct3, 14 dct4)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_configs import run
cases = [a.split(":") for a in sys.argv[1:]] or [("1", "169"), ("0", "169"), ("12", "169"), ("1", "31"), ("0", "31"), ("12", "16"), ("1", "145"), ("0", "145"), ("1", "385"), ("0", "385"),
                                                 ("1", "100"), ("12", "100"), ("14", "145"), ("1", "265"), ("1", "328"), ("12", "265")]
for k, n in cases:
    print(json.dumps(run(int(k), (int(n),), False, total_log2=25)), flush=True)
