import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vkfft_amd import api
k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
buf = torch.empty(2 << 27, dtype=torch.float32, device="cuda").uniform_(-1, 1)
N = 1 << k
app = api.App([N], (1 << 27) // N, buffer_ptr=buf.data_ptr(), normalize=True)
for _ in range(3):
    app.forward(); app.inverse()
torch.cuda.synchronize()
app.delete()
