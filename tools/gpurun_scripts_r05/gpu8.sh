# round 5, call 9: several-rows-per-workgroup packed rows at 2^10 ... 2^12 against the shipping row kernels; streaming copy
export TMPDIR=/tmp; O=gpurun_out/r05m; mkdir -p $O
timeout 300 python tools/ab_r05.py 9 10 11 12 > $O/ab.jsonl 2> $O/ab.err
timeout 300 python tools/ab_r05.py 9 10 11 12 >> $O/ab.jsonl 2>> $O/ab.err
cut -c1-200 $O/ab.jsonl
python - <<'PY'
import torch, time, sys
sys.path.insert(0, ".")
from vkfft_amd import api
lib = api.load()
a = torch.empty(2 << 27, dtype=torch.float32, device="cuda").uniform_(-1, 1); b = torch.empty_like(a)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn in (("own", lambda: lib.vkfftMI355XStreamCopy(b.data_ptr(), a.data_ptr(), 8 << 27, None)), ("torch", lambda: b.copy_(a))):
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(20): fn()
    e1.record(); e1.synchronize()
    print(name, "copy GB/s", 2 * (8 << 27) / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e9, "equal", bool(torch.equal(a, b)))
PY
