"""Development tool: hammer the fused Four-Step kernel under dependency pressure and count wrong results.
usage: python tools/stress_fused.py <log2N> <launch pairs> KEY=VALUE ...   (keys without the VKFFT_MI355X_ prefix)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
k = int(sys.argv[1]); pairs = int(sys.argv[2])
for kv in sys.argv[3:]:
    a, b = kv.split("="); os.environ["VKFFT_MI355X_" + a] = b
N = 1 << k; B = (1 << 27) // N
g = torch.Generator(device="cuda"); g.manual_seed(7)
x = torch.empty(2 * N * B, dtype=torch.float32, device="cuda").uniform_(-1, 1, generator=g)
buf = x.clone()
app = api.App([N], B, buffer_ptr=buf.data_ptr(), normalize=True)
bad = 0; worst = 0.0
for i in range(pairs):
    buf.copy_(x)
    app.forward(); app.inverse()
    torch.cuda.synchronize()
    e = (torch.linalg.norm(buf - x) / torch.linalg.norm(x)).item()
    worst = max(worst, e)
    if e > 2e-6:
        bad += 1
app.delete()
print(json.dumps(dict(log2N=k, pairs=pairs, env=sys.argv[3:], wrong=bad, worst_rel_err=worst)))
