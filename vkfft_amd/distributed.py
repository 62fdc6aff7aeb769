"""Multi-GPU drivers (new functionality: the reference is single-device, README.md:27-29).

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
  * batch sharding  — embarrassingly parallel: every rank plans and runs its own contiguous block of the batch,
                      no collective on the data path;
  * slab 3D C2C     — z-slabs: local 2D transforms over (x,y), ONE all-to-all that re-partitions z<->y, local 1D
                      transforms along z.  The result is left in y-slab layout
                      [nz][ny/P][nx] (no second exchange); `inverse()` takes that layout back to z-slabs.
The local transforms always go through the C-ABI (`lib` = the HIP library, or the CPU test double in tests)."""
import numpy as np

from . import api


def shard_range(total, rank, world):
    """contiguous block of `total` items owned by `rank` (first `total % world` ranks get one extra)"""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class BatchShardedFFT:
    """Batched nD transform whose batch axis is split across the ranks of `group`; each rank holds only its shard."""

    def __init__(self, shape, total_batch, rank, world, *, dp=False, device_index=0, lib=None, **kw):
        self.lo, self.hi = shard_range(total_batch, rank, world)
        self.local_batch = self.hi - self.lo
        self.app = api.App(list(shape), max(self.local_batch, 1), dp=dp, device_index=device_index, lib=lib, **kw) if self.local_batch else None

    def forward(self, ptr):
        if self.app:
            self.app.forward(buffer_ptr=ptr)

    def inverse(self, ptr):
        if self.app:
            self.app.inverse(buffer_ptr=ptr)

    def delete(self):
        if self.app:
            self.app.delete()


def _ptr(t):
    return t.data_ptr()


class SlabFFT3D:
    """3D C2C of an (nx, ny, nz) volume distributed as z-slabs over `world` ranks (nz % world == 0, ny % world == 0).

    forward(x): x = local z-slab, torch complex tensor [nz/P, ny, nx] (contiguous)  ->  y-slab [nz, ny/P, nx]
    inverse(y): y-slab -> z-slab (unnormalised unless normalize=True)."""

    def __init__(self, nx, ny, nz, group=None, *, dp=False, device_index=0, lib=None, normalize=False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        assert nz % self.P == 0 and ny % self.P == 0, "slab decomposition needs nz and ny divisible by the number of ranks"
        self.nx, self.ny, self.nz = nx, ny, nz
        self.nzl, self.nyl = nz // self.P, ny // self.P
        self.dp = dp
        # (x,y) sweep: nzl independent 2D transforms
        self.xy = api.App([nx, ny], self.nzl, dp=dp, device_index=device_index, lib=lib)
        # z sweep: 1D transforms along the slowest axis of [nz][nyl][nx], unit-stride axes omitted
        self.z = api.App([nx, self.nyl, nz], 1, dp=dp, device_index=device_index, lib=lib, omitDimension=[1, 1, 0, 0], normalize=int(normalize))
        self.normalize = normalize

    def _all_to_all(self, send):
        """send: [P, ...] chunks (chunk r goes to rank r); returns [P, ...] (chunk s came from rank s)."""
        import torch
        if self.P == 1:
            return send
        recv = torch.empty_like(send)
        real_s, real_r = torch.view_as_real(send), torch.view_as_real(recv)  # collectives on the underlying real storage
        try:
            self.dist.all_to_all_single(real_r, real_s, group=self.group)
        except Exception:
            ops = []
            for r in range(self.P):
                ops.append(self.dist.P2POp(self.dist.isend, real_s[r], r, self.group))
                ops.append(self.dist.P2POp(self.dist.irecv, real_r[r], r, self.group))
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        return recv

    def forward(self, x):
        import torch
        P, nzl, nyl, nx = self.P, self.nzl, self.nyl, self.nx
        assert tuple(x.shape) == (nzl, self.ny, nx) and x.is_contiguous()
        self.xy.forward(buffer_ptr=_ptr(x))
        _sync(x)
        # chunk r = y-range of rank r: [nzl, P, nyl, nx] -> [P, nzl, nyl, nx]
        send = x.view(nzl, P, nyl, nx).permute(1, 0, 2, 3).contiguous()
        recv = self._all_to_all(send)  # [P(src = z-block), nzl, nyl, nx] == [nz, nyl, nx]
        y = recv.view(self.nz, nyl, nx)
        self.z.forward(buffer_ptr=_ptr(y))
        _sync(y)
        return y

    def inverse(self, y):
        import torch
        P, nzl, nyl, nx = self.P, self.nzl, self.nyl, self.nx
        assert tuple(y.shape) == (self.nz, nyl, nx) and y.is_contiguous()
        self.z.inverse(buffer_ptr=_ptr(y))
        _sync(y)
        send = y.view(P, nzl, nyl, nx)  # chunk r = z-block of rank r
        recv = self._all_to_all(send.contiguous())  # [P(src = y-block), nzl, nyl, nx]
        x = recv.permute(1, 0, 2, 3).contiguous().view(nzl, self.ny, nx)
        self.xy.inverse(buffer_ptr=_ptr(x))
        _sync(x)
        if self.normalize:
            x /= (self.nx * self.ny)
        return x

    def delete(self):
        self.xy.delete(); self.z.delete()


def _sync(t):
    if t.is_cuda:
        import torch
        torch.cuda.current_stream().synchronize()
