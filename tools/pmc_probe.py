"""Workload for the rocprofv3 PMC passes (one counter set per pass): a calibration copy of known size, then two forward + inverse pairs of EVERY plan
bench.py launches (2^8 ... 2^22 on the 1 GiB buffer, default plans), so that tools/summarize_profiles.py can key the traffic by instance; then the same for the
fused Four-Step of the non-power-of-two lengths of BASELINE config 3 (kernel_mix_fused.h; 2^25 points each: summary key by_length)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
if os.environ.get("VKFFT_PMC_HASH_FILE"):
    open(os.environ["VKFFT_PMC_HASH_FILE"], "w").write(api.source_hash() + "\n")
buf = torch.empty(2 << 27, dtype=torch.float32, device="cuda").uniform_(-1, 1)
dst = torch.empty_like(buf)
for _ in range(2):
    dst.copy_(buf)          # calibration: reads 1 GiB, writes 1 GiB
lib = api.load()
for _ in range(2):
    lib.vkfftMI355XStreamCopy(dst.data_ptr(), buf.data_ptr(), 8 << 27, None)
torch.cuda.synchronize()
del dst
ks = [int(a) for a in sys.argv[1:]] or list(range(8, 23))
for k in ks:
    N = 1 << k
    app = api.App([N], (1 << 27) // N, buffer_ptr=buf.data_ptr(), normalize=True)
    for _ in range(2):
        app.forward(); app.inverse()
    torch.cuda.synchronize()
    app.delete()
MIX = [59049, 177147, 531441, 78125, 390625, 117649, 161051, 1771561, 28561]
if not sys.argv[1:]:
    for N in MIX:
        app = api.App([N], (1 << 25) // N, buffer_ptr=buf.data_ptr(), normalize=True)
        for _ in range(2):
            app.forward(); app.inverse()
        torch.cuda.synchronize()
        app.delete()
