"""Multi-GPU drivers (new functionality: the reference is single-device, README.md:27-29).

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
  * batch sharding  — embarrassingly parallel: every rank plans and runs its own contiguous block of the batch,
                      no collective on the data path;
  * slab 3D C2C     — z-slabs: local transforms along y and x, ONE all-to-all that re-partitions z<->y, local 1D
                      transforms along z.  The result is left in y-slab layout
                      [nz][ny/P][nx] (no second exchange); `inverse()` takes that layout back to z-slabs.

Slab exchange without pack/unpack passes.  Rank r must receive, from every rank s, the block
[z in slab s][y in block r][x].  The library's own strided output does the packing: the y transform runs in place on the
natural [z][y][x] slab, then the x transform (unit-stride rows) is planned as a 4-D problem (nx, ny/P, P, planes) of which
only axis 0 is transformed, reading the natural layout and WRITING rows straight into the send layout
[y-block][z][y in block][x] (VkFFTConfiguration::isInputFormatted + bufferStride, vkFFT_Structs.h:93-379), so every
message is one contiguous run and what arrives is already the [nz][ny/P][nx] volume of the z transform.  The planes are
processed in groups: while group g is on the wire (point-to-point sends of the group's P-1 runs, one RCCL group call), the
transforms of group g+1 run on the compute stream.  The inverse mirrors it (contiguous sends of the z-blocks, the x
transform gathers rows from the receive layout back into the natural slab).
The local transforms always go through the C-ABI (`lib` = the HIP library, or the CPU test double in tests)."""
from . import api


def shard_range(total, rank, world):
    """contiguous block of `total` items owned by `rank` (first `total % world` ranks get one extra)"""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _check_device(device_index):
    """the library allocates its tables and scratch on the CURRENT HIP device: it must be the one the plan is made for"""
    import torch
    if torch.cuda.is_available() and torch.cuda.current_device() != device_index:
        raise RuntimeError(f"current device {torch.cuda.current_device()} != plan device {device_index}: call torch.cuda.set_device first")


class BatchShardedFFT:
    """Batched nD transform whose batch axis is split across the ranks of `group`; each rank holds only its shard."""

    def __init__(self, shape, total_batch, rank, world, *, dp=False, device_index=0, lib=None, stream=None, **kw):
        _check_device(device_index)
        self.lo, self.hi = shard_range(total_batch, rank, world)
        self.local_batch = self.hi - self.lo
        self.app = api.App(list(shape), max(self.local_batch, 1), dp=dp, device_index=device_index, lib=lib, stream=stream, **kw) if self.local_batch else None

    def forward(self, ptr):
        if self.app:
            self.app.forward(buffer_ptr=ptr)

    def inverse(self, ptr):
        if self.app:
            self.app.inverse(buffer_ptr=ptr)

    def delete(self):
        if self.app:
            self.app.delete()


class SlabFFT3D:
    """3D C2C of an (nx, ny, nz) volume distributed as z-slabs over the ranks of `group`.  nz and ny divisible by their number: the pipelined form
    below; otherwise construction hands over to UnevenSlabFFT3D (slabs of shard_range sizes, one exchange, no plane groups) — same interface.

    forward(x): x = local z-slab, torch complex tensor [nz/P, ny, nx] (contiguous; overwritten)  ->  y-slab [nz, ny/P, nx]
    inverse(y): y-slab (overwritten) -> z-slab; unnormalised unless normalize=True.
    The exchange buffers (2 x the slab) and the inverse's output slab belong to the plan and are allocated once: a result is a view of them and
    stays valid until the next call of the same direction.
    `groups`: number of plane groups the exchange is pipelined over (must divide nz/P)."""

    def __new__(cls, nx, ny, nz, group=None, **kw):
        import torch.distributed as dist
        P = dist.get_world_size(group) if dist.is_initialized() else 1
        if cls is SlabFFT3D and (nz % P or ny % P):
            kw.pop("groups", None)
            return UnevenSlabFFT3D(nx, ny, nz, group, **kw)
        return super().__new__(cls)

    def __init__(self, nx, ny, nz, group=None, *, dp=False, device_index=0, lib=None, normalize=False, groups=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        P = self.P
        assert nz % P == 0 and ny % P == 0
        self.nx, self.ny, self.nz = nx, ny, nz
        self.nzl, self.nyl = nz // P, ny // P
        nzl, nyl = self.nzl, self.nyl
        if groups is None:
            groups = 1
            while groups < 8 and nzl % (2 * groups) == 0 and (nzl // (2 * groups)) * nyl * nx >= (1 << 16):
                groups *= 2  # messages of at least 2^16 points
        assert nzl % groups == 0
        self.G, self.zg = groups, nzl // groups
        self.dp, self.normalize = dp, normalize
        self.cuda = torch.cuda.is_available() and not getattr(lib, "_vkfft_test_double", False)  # the CPU test double works on host tensors
        self.compute = None
        stream = None
        if self.cuda:
            _check_device(device_index)
            self.compute = torch.cuda.Stream()
            stream = self.compute.cuda_stream
        zg = self.zg
        # y transform of a group of planes, in place on the natural slab: 2-D plan (nx, ny) with axis 0 omitted
        self.fy = api.App([nx, ny], zg, dp=dp, device_index=device_index, lib=lib, stream=stream, omitDimension=[1, 0, 0, 0])
        # x transform of a group of planes: rows of the natural slab (input side) <-> rows of the exchange layout (buffer side)
        #   point (x, yl, yb, z):  natural  x + nx*(yl + nyl*yb) + nx*ny*z      exchange  x + nx*yl + nx*nyl*z + nx*nyl*nzl*yb
        self.fx = api.App([nx, nyl, P, zg], 1, dp=dp, device_index=device_index, lib=lib, stream=stream, omitDimension=[0, 1, 1, 1],
                          isInputFormatted=1, inverseReturnToInputBuffer=1,
                          inputBufferStride=[nx, nx * nyl, nx * ny, nx * ny * zg],
                          bufferStride=[nx, nx * nyl * nzl, nx * nyl, nx * nyl * nzl * P])
        # z transform: 1-D along the slowest axis of [nz][nyl][nx]
        self.fz = api.App([nx, nyl, nz], 1, dp=dp, device_index=device_index, lib=lib, stream=stream, omitDimension=[1, 1, 0, 0], normalize=int(normalize))
        self.es = 16 if dp else 8
        self._bufs = None

    def _buffers(self, like):
        """send / receive layouts [P, nzl, nyl, nx] and the inverse's z-slab [nzl, ny, nx]: allocated on first use, then reused"""
        if self._bufs is None or self._bufs[0].dtype != like.dtype or self._bufs[0].device != like.device:
            t = self.torch
            self._bufs = (t.empty((self.P, self.nzl, self.nyl, self.nx), dtype=like.dtype, device=like.device),
                          t.empty((self.P, self.nzl, self.nyl, self.nx), dtype=like.dtype, device=like.device),
                          t.empty((self.nzl, self.ny, self.nx), dtype=like.dtype, device=like.device))
        return self._bufs

    # ---- exchange of one plane group: P-1 contiguous runs each way, one grouped point-to-point call ----
    def _exchange(self, send, recv, g):
        """send / recv: [P, nzl, nyl, nx]; moves planes [g*zg, (g+1)*zg) of every chunk: send[r] -> rank r's recv[me]"""
        dist, torch = self.dist, self.torch
        lo, hi = g * self.zg, (g + 1) * self.zg
        recv[self.rank, lo:hi].copy_(send[self.rank, lo:hi])
        if self.P == 1:
            return []
        ops = []
        for d in range(1, self.P):
            to, frm = (self.rank + d) % self.P, (self.rank - d) % self.P
            ops.append(dist.P2POp(dist.isend, torch.view_as_real(send[to, lo:hi]), to, self.group))
            ops.append(dist.P2POp(dist.irecv, torch.view_as_real(recv[frm, lo:hi]), frm, self.group))
        return dist.batch_isend_irecv(ops)

    def forward(self, x):
        torch = self.torch
        P, nzl, nyl, nx, ny, zg = self.P, self.nzl, self.nyl, self.nx, self.ny, self.zg
        assert tuple(x.shape) == (nzl, ny, nx) and x.is_contiguous()
        send, recv, _ = self._buffers(x)
        caller = torch.cuda.current_stream() if self.cuda else None
        if self.cuda:
            self.compute.wait_stream(caller)  # x, send and recv are ready for the compute stream
        pending = []
        for g in range(self.G):
            off_nat = g * zg * ny * nx * self.es
            off_ex = g * zg * nyl * nx * self.es
            self.fy.forward(buffer_ptr=x.data_ptr() + off_nat)
            self.fx.forward(buffer_ptr=send.data_ptr() + off_ex, input_ptr=x.data_ptr() + off_nat)
            if self.cuda:
                ev = torch.cuda.Event(); ev.record(self.compute)
                caller.wait_event(ev)  # the group's sends are enqueued behind its transforms; the next group's transforms overlap them
            pending += self._exchange(send, recv, g)
        for w in pending:
            w.wait()
        y = recv.view(self.nz, nyl, nx)
        if self.cuda:
            self.compute.wait_stream(caller)
        self.fz.forward(buffer_ptr=y.data_ptr())
        if self.cuda:
            caller.wait_stream(self.compute)
            y.record_stream(self.compute); send.record_stream(self.compute); x.record_stream(self.compute)
        return y

    def inverse(self, y):
        torch = self.torch
        P, nzl, nyl, nx, ny, zg = self.P, self.nzl, self.nyl, self.nx, self.ny, self.zg
        assert tuple(y.shape) == (self.nz, nyl, nx) and y.is_contiguous()
        caller = torch.cuda.current_stream() if self.cuda else None
        a, b, x = self._buffers(y)
        recv = a if y.data_ptr() != a.data_ptr() else b  # (forward returns a view of b: the inverse then receives into a)
        if self.cuda:
            self.compute.wait_stream(caller)
        self.fz.inverse(buffer_ptr=y.data_ptr())
        if self.cuda:
            caller.wait_stream(self.compute)
        send = y.view(P, nzl, nyl, nx)  # chunk r = the z-block of rank r: already contiguous
        works = [self._exchange(send, recv, g) for g in range(self.G)]
        for g in range(self.G):
            for w in works[g]:
                w.wait()
            if self.cuda:
                self.compute.wait_stream(caller)  # group g has arrived (its local copy and receives are on the caller's stream)
            off_nat = g * zg * ny * nx * self.es
            off_ex = g * zg * nyl * nx * self.es
            self.fx.inverse(buffer_ptr=recv.data_ptr() + off_ex, input_ptr=x.data_ptr() + off_nat)
            self.fy.inverse(buffer_ptr=x.data_ptr() + off_nat)
        if self.cuda:
            caller.wait_stream(self.compute)
            x.record_stream(self.compute); recv.record_stream(self.compute); y.record_stream(self.compute)
        if self.normalize:
            x /= (self.nx * self.ny)
        return x

    def exchange_bytes_per_rank(self):
        """bytes this rank puts on the wire per exchange (everything but its own chunk)"""
        return (self.P - 1) * self.nzl * self.nyl * self.nx * self.es

    def delete(self):
        self.fy.delete(); self.fx.delete(); self.fz.delete()



class UnevenSlabFFT3D:
    """The slab transform for nz or ny NOT divisible by the number of ranks: rank r owns the planes shard_range(nz, r, P) before the exchange and the
    y-rows shard_range(ny, r, P) after it (the first `total % P` ranks one more than the others).  Same decomposition and the same packing-free exchange
    as SlabFFT3D — the x transform writes its rows straight into the send layout, now ONE strided plan per block size (blocks of base + 1 rows, then
    blocks of base rows) — with messages of rank-dependent size and no pipelining over plane groups.

    forward(x): x = local z-slab [nz_r, ny, nx] (contiguous, overwritten) -> y-slab [nz, ny_r, nx];  inverse(y) takes that back to the z-slab."""

    def __init__(self, nx, ny, nz, group=None, *, dp=False, device_index=0, lib=None, normalize=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.P = P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        assert nz >= P and ny >= P, "every rank needs at least one plane and one row"
        self.nx, self.ny, self.nz = nx, ny, nz
        self.zr = [shard_range(nz, r, P) for r in range(P)]   # planes of rank r before the exchange
        self.yr = [shard_range(ny, r, P) for r in range(P)]   # rows of rank r after it
        self.nzl = self.zr[self.rank][1] - self.zr[self.rank][0]
        self.nyl = self.yr[self.rank][1] - self.yr[self.rank][0]
        self.G = 1
        self.dp, self.normalize = dp, normalize
        self.es = 16 if dp else 8
        self.cuda = torch.cuda.is_available() and not getattr(lib, "_vkfft_test_double", False)
        self.compute = None
        stream = None
        if self.cuda:
            _check_device(device_index)
            self.compute = torch.cuda.Stream()
            stream = self.compute.cuda_stream
        nzl, nyl = self.nzl, self.nyl
        kw = dict(dp=dp, device_index=device_index, lib=lib, stream=stream)
        self.fy = api.App([nx, ny], nzl, omitDimension=[1, 0, 0, 0], **kw)
        # x transforms, one plan per block size: blocks [b0, b0 + cnt) of `rows` rows each start at row y0 of the natural slab and at element e0 of
        # the send layout  [block][z][row in block][x]  (block b holds nzl * rows_b * nx elements)
        self.fx = []
        base, extra = divmod(ny, P)
        y0 = e0 = 0
        for cnt, rows in ((extra, base + 1), (P - extra, base)):
            if cnt and rows:
                app = api.App([nx, rows, cnt, nzl], 1, omitDimension=[0, 1, 1, 1], isInputFormatted=1, inverseReturnToInputBuffer=1,
                              inputBufferStride=[nx, nx * rows, nx * ny, nx * ny * nzl],
                              bufferStride=[nx, nx * rows * nzl, nx * rows, nx * rows * nzl * cnt], **kw)
                self.fx.append((app, y0 * nx, e0))
            y0 += cnt * rows; e0 += cnt * rows * nzl * nx
        self.fz = api.App([nx, nyl, nz], 1, omitDimension=[1, 1, 0, 0], normalize=int(normalize), **kw)
        self._bufs = None

    def _buffers(self, like):
        if self._bufs is None or self._bufs[0].dtype != like.dtype or self._bufs[0].device != like.device:
            t = self.torch
            self._bufs = (t.empty(self.nzl * self.ny * self.nx, dtype=like.dtype, device=like.device),   # [block][z][row][x]: this rank's planes, every rank's rows
                          t.empty(self.nz * self.nyl * self.nx, dtype=like.dtype, device=like.device),   # [z of every rank][row][x]: this rank's rows
                          t.empty((self.nzl, self.ny, self.nx), dtype=like.dtype, device=like.device))
        return self._bufs

    def _chunks(self):
        """(offset, length) in elements of rank r's chunk: in the [block][z][row][x] layout of MY planes, and in the [z][row][x] layout of MY rows"""
        nx = self.nx
        byrows, e = [], 0
        for (lo, hi) in self.yr:
            byrows.append((e, self.nzl * (hi - lo) * nx)); e += self.nzl * (hi - lo) * nx
        byplanes = [(lo * self.nyl * nx, (hi - lo) * self.nyl * nx) for (lo, hi) in self.zr]
        return byrows, byplanes

    def _exchange(self, src, src_chunks, dst, dst_chunks):
        """chunk r of src -> rank r's dst chunk of this rank; returns the outstanding requests"""
        dist, torch = self.dist, self.torch
        me = self.rank
        so, sl = src_chunks[me]; do, dl = dst_chunks[me]
        dst[do:do + dl].copy_(src[so:so + sl])
        if self.P == 1:
            return []
        ops = []
        for d in range(1, self.P):
            to, frm = (me + d) % self.P, (me - d) % self.P
            so, sl = src_chunks[to]; do, dl = dst_chunks[frm]
            ops.append(dist.P2POp(dist.isend, torch.view_as_real(src[so:so + sl]), to, self.group))
            ops.append(dist.P2POp(dist.irecv, torch.view_as_real(dst[do:do + dl]), frm, self.group))
        return dist.batch_isend_irecv(ops)

    def forward(self, x):
        torch = self.torch
        assert tuple(x.shape) == (self.nzl, self.ny, self.nx) and x.is_contiguous()
        send, recv, _ = self._buffers(x)
        caller = torch.cuda.current_stream() if self.cuda else None
        if self.cuda:
            self.compute.wait_stream(caller)
        self.fy.forward(buffer_ptr=x.data_ptr())
        for app, yoff, eoff in self.fx:
            app.forward(buffer_ptr=send.data_ptr() + eoff * self.es, input_ptr=x.data_ptr() + yoff * self.es)
        if self.cuda:
            caller.wait_stream(self.compute)
        byrows, byplanes = self._chunks()
        for w in self._exchange(send, byrows, recv, byplanes):
            w.wait()
        y = recv.view(self.nz, self.nyl, self.nx)
        if self.cuda:
            self.compute.wait_stream(caller)
        self.fz.forward(buffer_ptr=y.data_ptr())
        if self.cuda:
            caller.wait_stream(self.compute)
            y.record_stream(self.compute); send.record_stream(self.compute); x.record_stream(self.compute)
        return y

    def inverse(self, y):
        torch = self.torch
        assert tuple(y.shape) == (self.nz, self.nyl, self.nx) and y.is_contiguous()
        caller = torch.cuda.current_stream() if self.cuda else None
        a, b, x = self._buffers(y)
        if self.cuda:
            self.compute.wait_stream(caller)
        self.fz.inverse(buffer_ptr=y.data_ptr())
        if self.cuda:
            caller.wait_stream(self.compute)
        byrows, byplanes = self._chunks()
        for w in self._exchange(y.view(-1), byplanes, a, byrows):
            w.wait()
        if self.cuda:
            self.compute.wait_stream(caller)
        for app, yoff, eoff in self.fx:
            app.inverse(buffer_ptr=a.data_ptr() + eoff * self.es, input_ptr=x.data_ptr() + yoff * self.es)
        self.fy.inverse(buffer_ptr=x.data_ptr())
        if self.cuda:
            caller.wait_stream(self.compute)
            x.record_stream(self.compute); a.record_stream(self.compute); y.record_stream(self.compute)
        if self.normalize:
            x /= (self.nx * self.ny)
        return x

    def exchange_bytes_per_rank(self):
        return (self.nzl * self.ny - self.nzl * self.nyl) * self.nx * self.es

    def delete(self):
        self.fy.delete(); self.fz.delete()
        for app, _, _ in self.fx:
            app.delete()
