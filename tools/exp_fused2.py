"""Development tool: A/B of the fused Four-Step kernel variants (VKFFT_MI355X_FUV<k> = index into the table of kernels_fused.hip) on the
1 GiB headline buffers: correctness against torch.fft on the device, run-to-run bit equality (hand-off races), pair time.  JSON lines."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api


def run(k, variant, iters=20, check=True, extra=None):
    os.environ[f"VKFFT_MI355X_FUV{k}"] = str(variant)
    for a, b in (extra or {}).items():
        os.environ["VKFFT_MI355X_" + a] = str(b)
    N = 1 << k; B = (1 << 27) // N
    x = torch.empty(B, N, 2, dtype=torch.float32, device="cuda").uniform_(-1, 1)
    t = x.clone()
    app = api.App([N], B, buffer_ptr=t.data_ptr(), normalize=True)
    desc = app.describe() if hasattr(app, "describe") else ""
    out = dict(log2N=k, variant=variant, extra=extra or {})
    if check:
        app.forward(); torch.cuda.synchronize()
        ref = torch.view_as_real(torch.fft.fft(torch.view_as_complex(x[: min(B, 64)])))
        got = t[: min(B, 64)]
        out["rel_l2_fwd"] = float((got - ref).norm() / ref.norm())
        y1 = t.clone()
        app.inverse(); torch.cuda.synchronize()
        out["rel_l2_roundtrip"] = float((t - x).norm() / x.norm())
        # run-to-run equality of the forward transform
        bad = 0
        for _ in range(6):
            t.copy_(x); app.forward(); torch.cuda.synchronize()
            bad += int((t != y1).any().item())
        out["runs_differing"] = bad
        del y1
    t.copy_(x)
    for _ in range(3):
        app.forward(); app.inverse()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for rep in range(3):
        e0.record()
        for _ in range(iters):
            app.forward(); app.inverse()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    out["pair_ms"] = round(best, 4)
    out["alg_GBps"] = round(4 * N * B * 8 / (best * 1e-3) / 1e9, 1)
    app.delete()
    for a in (extra or {}):
        os.environ.pop("VKFFT_MI355X_" + a, None)
    os.environ.pop(f"VKFFT_MI355X_FUV{k}", None)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    plan = json.loads(sys.argv[1]) if len(sys.argv) > 1 else [[15, 0], [15, 1], [16, 0], [16, 1], [16, 2], [16, 3], [17, 0], [17, 1], [18, 0], [18, 1]]
    for item in plan:
        k, v = item[0], item[1]
        extra = item[2] if len(item) > 2 else None
        try:
            run(k, v, extra=extra, check=(item[3] if len(item) > 3 else True))
        except Exception as e:  # keep going: one bad variant must not hide the others
            print(json.dumps(dict(log2N=k, variant=v, error=repr(e))), flush=True)
