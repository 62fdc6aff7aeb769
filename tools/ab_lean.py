"""Development tool: A/B of the register-lean kernels (kernel_pow2_lean.h) against the shipping shapes through the C-ABI — 1 GiB batched 1-D C2C
fp32, FFT + normalised iFFT pairs, with a result check per configuration (spot transforms against torch's double FFT, round trip of the whole buffer).
usage: python tools/ab_lean.py [sizes ...]   prints one JSON line per (size, configuration)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api

TOTAL = 27
CONFIGS = {
    21: [{}] + [{"FUSED_LAG": l, "FUSED_RING": r, "FUSED_QUEUES": q} for q in (1, 2, 4, 8) for (l, r) in ((1, 2), (1, 3), (2, 3), (2, 4), (3, 5))] + [{"FUSED_MARGIN": m} for m in (100, 300, 400)],
    22: [{}] + [{"FUSED_LAG": l, "FUSED_RING": r, "FUSED_QUEUES": q} for q in (1, 2, 4, 8) for (l, r) in ((1, 2), (1, 3), (2, 3), (2, 4), (3, 5))] + [{"FUSED_MARGIN": m} for m in (100, 300, 400)],
    19: [{}] + [{"FUSED_CHUNK_KIB": c} for c in (2048, 8192, 16384)] + [{"FUSED_QUEUES": q} for q in (1, 4)],
    20: [{}] + [{"FUSED_CHUNK_KIB": c} for c in (4096, 16384, 32768)] + [{"FUSED_QUEUES": q} for q in (1, 4)],
}


def run(k, env, iters=6):
    for key in list(os.environ):
        if key.startswith("VKFFT_MI355X_"):
            del os.environ[key]
    os.environ.update({"VKFFT_MI355X_" + a: str(b) for a, b in env.items()})
    N = 1 << k; B = (1 << TOTAL) // N
    g = torch.Generator(device="cuda"); g.manual_seed(k)
    t = torch.empty(2 << TOTAL, dtype=torch.float32, device="cuda").uniform_(-1, 1, generator=g)
    ref = t.clone()
    app = api.App([N], B, buffer_ptr=t.data_ptr(), normalize=True)
    launches, kern = app.launch_info()
    app.forward(); torch.cuda.synchronize()
    X = torch.view_as_complex(t.view(-1, 2)).view(B, N); xc = torch.view_as_complex(ref.view(-1, 2)).view(B, N)
    spot = 0.0
    for b in (0, B // 3, B - 1):
        r = torch.fft.fft(xc[b].to(torch.complex128))
        spot = max(spot, float(torch.abs(X[b].to(torch.complex128) - r).max() / (torch.sqrt(torch.mean(torch.abs(r) ** 2)) * 2.0 ** -23)))
    app.inverse(); torch.cuda.synchronize()
    rt = float((t - ref).abs().max() / (0.577 * 2.0 ** -23))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for rep in range(3):
        e0.record()
        for _ in range(iters):
            app.forward(); app.inverse()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    app.delete()
    return dict(k=k, cfg=env, kernel=kern, launches=launches, pair_ms=round(best, 4), alg_GBps=round(4 * (8 << TOTAL) / (best * 1e-3) / 1e9, 1),
                fwd_max_ulp=round(spot, 1), roundtrip_max_ulp=round(rt, 1))


def stress(k, env, pairs=100):
    """many launch pairs under unbalanced queues: every pair must return the buffer (the race class of DESIGN 4.10)"""
    for key in list(os.environ):
        if key.startswith("VKFFT_MI355X_"):
            del os.environ[key]
    os.environ.update({"VKFFT_MI355X_" + a: str(b) for a, b in env.items()})
    N = 1 << k; B = (1 << TOTAL) // N
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    x = torch.empty(2 << TOTAL, dtype=torch.float32, device="cuda").uniform_(-1, 1, generator=g)
    buf = x.clone()
    app = api.App([N], B, buffer_ptr=buf.data_ptr(), normalize=True)
    nx = torch.linalg.norm(x); worst = 0.0; bad = 0
    for _ in range(pairs):
        buf.copy_(x)
        app.forward(); app.inverse()
        e = float(torch.linalg.norm(buf - x) / nx)
        worst = max(worst, e); bad += e > 2e-6
    app.delete()
    return dict(stress=k, cfg=env, pairs=pairs, worst_rel_l2=float(f"{worst:.3e}"), bad_pairs=int(bad))


if __name__ == "__main__":
    if sys.argv[1:2] == ["stress"]:
        for k, q in ((16, 3), (18, 5), (20, 6), (19, 3), (17, 7), (15, 3)):
            print(json.dumps(stress(k, {f"FUV{k}": 3, "ROW15": 0, "FUSED_LAG": 1, "FUSED_RING": 4, "FUSED_QUEUES": q})), flush=True)
        for k in (16, 20):
            print(json.dumps(stress(k, {f"FUV{k}": 3, "ROW15": 0})), flush=True)
        sys.exit(0)
    ks = [int(a) for a in sys.argv[1:]] or sorted(CONFIGS)
    for k in ks:
        for env in CONFIGS[k]:
            try:
                print(json.dumps(run(k, env)), flush=True)
            except Exception as e:  # a configuration that does not plan must not end the sweep
                print(json.dumps(dict(k=k, cfg=env, error=str(e))), flush=True)
