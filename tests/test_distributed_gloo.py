"""world_size-2 tests of the multi-GPU drivers on CPU (gloo): the same code path as on 2..8 MI355X (nccl/RCCL),
with the CPU test double standing in for the local transforms.  Index permutations of the slab exchange must be
bit-exact; numerics are checked against the oracle's double-precision truth."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, emu_path, case, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vkfft_amd import api
    from vkfft_amd.distributed import BatchShardedFFT, SlabFFT3D, shard_range
    lib = api.load_test_double(emu_path)
    try:
        if case == "batch":
            N, B = 256, 11  # uneven split on purpose
            rng = np.random.default_rng(0)
            full = (rng.uniform(-1, 1, N * B) + 1j * rng.uniform(-1, 1, N * B)).astype(np.complex64)
            lo, hi = shard_range(B, rank, world)
            local = torch.from_numpy(full.reshape(B, N)[lo:hi].copy())
            plan = BatchShardedFFT([N], B, rank, world, lib=lib)
            plan.forward(local.data_ptr())
            out = [None] * world
            dist.all_gather_object(out, (lo, hi, local.numpy()))
            if rank == 0:
                got = np.concatenate([o[2] for o in sorted(out, key=lambda t: t[0])], axis=0)
                ref = np.fft.fft(full.astype(np.complex128).reshape(B, N), axis=1)
                ret["err"] = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
            plan.delete()
        elif case == "uneven":
            nx, ny, nz = 16, 9, 11  # neither ny nor nz divisible by 2 or 3: UnevenSlabFFT3D
            rng = np.random.default_rng(3)
            vol = (rng.uniform(-1, 1, nx * ny * nz) + 1j * rng.uniform(-1, 1, nx * ny * nz)).astype(np.complex64).reshape(nz, ny, nx)
            zlo, zhi = shard_range(nz, rank, world); ylo, yhi = shard_range(ny, rank, world)
            plan = SlabFFT3D(nx, ny, nz, lib=lib)
            assert type(plan).__name__ == "UnevenSlabFFT3D"
            y = plan.forward(torch.from_numpy(vol[zlo:zhi].copy()))
            ref = np.fft.fftn(vol.astype(np.complex128))[:, ylo:yhi, :]
            e_f = float(np.linalg.norm(y.numpy() - ref) / np.linalg.norm(ref))
            back = plan.inverse(y.clone())
            want = vol[zlo:zhi].astype(np.complex128) * (nx * ny * nz)
            e_b = float(np.linalg.norm(back.numpy() - want) / np.linalg.norm(want))
            out = [None] * world
            dist.all_gather_object(out, (True, e_f, e_b))
            if rank == 0:
                ret["exact"] = True; ret["e_f"] = max(o[1] for o in out); ret["e_b"] = max(o[2] for o in out)
            plan.delete()
        else:
            case_groups = 3 if case == "slab3" else None
            nx, ny, nz = 16, 8, 12
            rng = np.random.default_rng(1)
            vol = (rng.uniform(-1, 1, nx * ny * nz) + 1j * rng.uniform(-1, 1, nx * ny * nz)).astype(np.complex64).reshape(nz, ny, nx)
            nzl = nz // world
            x = torch.from_numpy(vol[rank * nzl:(rank + 1) * nzl].copy())
            plan = SlabFFT3D(nx, ny, nz, lib=lib, groups=case_groups)
            # the exchange alone must be a bit-exact permutation: tag every element with its global index, lay the slab out the way
            # the x transform writes it ([y-block][z][y in block][x]) and move it group by group
            nyl = ny // world
            tag = torch.arange(nz * ny * nx, dtype=torch.float32).view(nz, ny, nx)[rank * nzl:(rank + 1) * nzl]
            tagc = torch.complex(tag, -tag).contiguous()
            send = tagc.view(nzl, world, nyl, nx).permute(1, 0, 2, 3).contiguous()
            recv = torch.zeros_like(send)
            for g in range(plan.G):
                for w in plan._exchange(send, recv, g):
                    w.wait()
            recv = recv.view(nz, nyl, nx)
            want = torch.arange(nz * ny * nx, dtype=torch.float32).view(nz, ny, nx)[:, rank * nyl:(rank + 1) * nyl, :]
            exact = bool(torch.equal(recv.real, want)) and bool(torch.equal(recv.imag, -want))
            y = plan.forward(x.clone())
            ref = np.fft.fftn(vol.astype(np.complex128))[:, rank * (ny // world):(rank + 1) * (ny // world), :]
            e_f = float(np.linalg.norm(y.numpy() - ref) / np.linalg.norm(ref))
            back = plan.inverse(y.clone())
            e_b = float(np.linalg.norm(back.numpy() - vol[rank * nzl:(rank + 1) * nzl] * (nx * ny * nz)) / np.linalg.norm(vol[rank * nzl:(rank + 1) * nzl] * (nx * ny * nz)))
            out = [None] * world
            dist.all_gather_object(out, (exact, e_f, e_b))
            if rank == 0:
                ret["exact"] = all(o[0] for o in out); ret["e_f"] = max(o[1] for o in out); ret["e_b"] = max(o[2] for o in out)
            plan.delete()
    finally:
        dist.destroy_process_group()


def _run(case, emu_lib_path, world=2):
    mgr = mp.Manager(); ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + 7 * world
    mp.spawn(_worker, args=(world, port, emu_lib_path, case, ret), nprocs=world, join=True)
    return dict(ret)


@pytest.fixture(scope="module")
def emu_path(emu_lib):
    return os.path.join(ROOT, "tests", "hostemu", "_build", "libvkfft_hostemu.so")


def test_batch_sharding_two_ranks(emu_path):
    r = _run("batch", emu_path)
    assert r["err"] < 1e-6


@pytest.mark.parametrize("case", ["slab", "slab3"])
def test_slab_3d_all_to_all_two_ranks(emu_path, case):
    """slab 3-D transform on two ranks: exchange bit-exact, forward and inverse against numpy; "slab3": three pipelined plane groups"""
    r = _run(case, emu_path)
    assert r["exact"], "slab exchange is not a bit-exact permutation"
    assert r["e_f"] < 1e-6 and r["e_b"] < 2e-6


@pytest.mark.parametrize("world", [2, 3])
def test_slab_3d_uneven_slabs(emu_path, world):
    """nz = 11 planes and ny = 9 rows over 2 and 3 ranks: slabs of shard_range sizes, messages of rank-dependent size, forward and inverse against numpy"""
    r = _run("uneven", emu_path, world)
    assert r["e_f"] < 1e-6 and r["e_b"] < 2e-6, r


def test_shard_range_partitions_exactly():
    from vkfft_amd.distributed import shard_range
    for total in (1, 7, 8, 128, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
