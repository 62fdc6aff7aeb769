"""Run perf_sweep for one size under several environment settings (development tool).
usage: tune.py k "ENV1=a ENV2=b" "ENV1=c" ...   -> one line per setting"""
import sys, os, subprocess, json
k = sys.argv[1]
here = os.path.dirname(os.path.abspath(__file__))
for setting in sys.argv[2:] or [""]:
    env = dict(os.environ)
    for kv in setting.split():
        a, b = kv.split("=")
        env[a] = b
    r = subprocess.run([sys.executable, os.path.join(here, "perf_sweep.py"), k, k], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if line:
        d = json.loads(line[-1])
        print(f"k={k} [{setting}] pair_ms={d['pair_ms']} alg_GBps={d['alg_GBps']} uploads={d['uploads']}", flush=True)
    else:
        print(f"k={k} [{setting}] FAILED {r.stderr[-300:]}", flush=True)
