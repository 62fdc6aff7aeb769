#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native FFT library.

Metric (BASELINE.json): GFLOP/s (5 N log2 N) + achieved HBM GB/s, batched 1D C2C fp32, 1/2/4/8 GPU.
Workload (BASELINE.json configs[1]): N = 2^8 .. 2^22, batch = 2^27/N  (one 1 GiB in-place buffer per GPU),
the reference's sample-0 protocol (sample_0_benchmark_VkFFT_single.cpp:85-88, utils_VkFFT.cpp:920-933).

One *step* = one forward + one inverse transform of the 1 GiB buffer for every one of the 15 sizes
(plans are created once, outside the timed region; inputs are resident in HBM).  Inverse plans are
normalised (1/N) so that the data stays uniform-random in [-1,1] for the whole run instead of overflowing
to inf/NaN as it does in the reference's unnormalised loop (data-dependent clocks: MI355X_MICROARCH.md DVFS).
Multi-GPU: the batch axis is sharded, every rank owns an independent 1 GiB buffer ("weak" scaling), no
collective on the data path; ranks only meet in the barrier around the timed region.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     : dominant kernel of the sweep, algorithmic bytes per launch / HIP-event launch duration
  cpu_baseline : the reference's CPU path (FFTW3 API, served by MKL) timed on this box's host cores on a
                 bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KMIN, KMAX, TOTAL_LOG2 = 8, 22, 27
HBM_PEAK_GBPS = 8000.0        # MI355X spec (MI355X_MICROARCH.md); measured float4-copy ceiling 6290


def flops_pair(k):
    N = 1 << k
    return 2 * 5.0 * N * k * ((1 << TOTAL_LOG2) // N)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` run directly (no torchrun): start one process per GPU on this node, let rank 0's JSON line through."""
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: this node exposes {have} GPU(s)")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="c2c-sweep", choices=["c2c-sweep", "slab3d"])
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:])
    if args.workload == "slab3d":
        return slab3d_main(args)

    import numpy as np
    import torch
    from vkfft_amd import api

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.cuda.current_device()
    api.load()  # fails loudly if the HIP extension is missing

    nfloat = 2 << TOTAL_LOG2
    gen = torch.Generator(device="cuda"); gen.manual_seed(1 + rank)
    buf = torch.empty(nfloat, dtype=torch.float32, device="cuda").uniform_(-1, 1, generator=gen)
    stream = torch.cuda.current_stream().cuda_stream
    apps = {}
    for k in range(KMIN, KMAX + 1):
        N = 1 << k
        apps[k] = api.App([N], (1 << TOTAL_LOG2) // N, device_index=dev, buffer_ptr=buf.data_ptr(), normalize=True,
                          stream=stream if stream else None)

    def step():
        for k in range(KMIN, KMAX + 1):
            apps[k].forward()
            apps[k].inverse()

    initial = buf.clone()  # every pair is FFT followed by the normalised inverse: the buffer must come back (checked after the timed region)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # ---- result check of the timed work itself: (warmup + steps) x 15 forward/inverse pairs later the buffer is its initial contents up to rounding
    diff = buf - initial
    roundtrip_rel_l2 = float(torch.linalg.vector_norm(diff.double()) / torch.linalg.vector_norm(initial.double()))
    max_abs_err = float(diff.abs().max())
    finite = bool(torch.isfinite(buf).all())
    del diff, initial
    # The rounding of a transform is a FIXED perturbation of the exact operator (the same twiddle tables and butterflies every time), so the errors of
    # repeated pairs on the same buffer add coherently: the limit is linear in the number of pairs, 3e-7 per pair (a single pair measures 2.2-3.0e-7
    # and is asserted below 2e-6 in tests/; measured over 90 pairs: 1.06e-5 = 1.2e-7 per pair); the largest element error is bounded at 8x the
    # same figure in absolute units (inputs uniform in [-1, 1]: a single wrong element anywhere in the 2^27 points is off by ~0.5)
    pairs = (args.steps + args.warmup) * (KMAX - KMIN + 1)
    rt_limit = 3e-7 * pairs
    if dist:
        t = torch.tensor([roundtrip_rel_l2, max_abs_err, 0.0 if finite else 1.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        roundtrip_rel_l2, max_abs_err, finite = float(t[0]), float(t[1]), float(t[2]) == 0.0
    if not finite or not (roundtrip_rel_l2 <= rt_limit) or not (max_abs_err <= 8 * rt_limit):
        raise SystemExit(f"bench.py: result check failed after the timed loop: round-trip rel. L2 {roundtrip_rel_l2:.3e} (limit {rt_limit:.3e}), "
                         f"max |err| {max_abs_err:.3e}, finite={finite}")
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    total_flops = sum(flops_pair(k) for k in range(KMIN, KMAX + 1)) * world
    bytes_step = (KMAX - KMIN + 1) * 2 * 2 * (8 << TOTAL_LOG2) * world  # 15 sizes x (fwd+inv) x (read+write) x 1 GiB
    value = total_flops / (ms_per_step * 1e-3) / 1e9

    # ---- per-size table + dominant kernel, HIP events on the launch stream (outside the contract-timed region) ----
    GIB2 = 2.0 * (8 << TOTAL_LOG2)  # algorithmic bytes of one transform of the buffer: one read + one write
    per_size = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 6

    def timed(fn, n):
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / n

    # what a plain device-to-device copy of the same 1 GiB reaches on this box, same clock (the practical HBM ceiling)
    other = torch.empty_like(buf)
    other.copy_(buf)
    copy_ms = min(timed(lambda: other.copy_(buf), 10) for _ in range(2))
    copy_GBps = GIB2 / (copy_ms * 1e-3) / 1e9
    # ... and the library's own streaming copy (16 bytes per lane, non-temporal on both sides: extension vkfftMI355XStreamCopy)
    lib = api.load()
    own_rc = []
    own = lambda: own_rc.append(lib.vkfftMI355XStreamCopy(other.data_ptr(), buf.data_ptr(), 8 << TOTAL_LOG2, stream if stream else None))
    other.zero_()
    own()
    own_ms = min(timed(own, 10) for _ in range(2))
    own_GBps = GIB2 / (own_ms * 1e-3) / 1e9
    # (the copy counts as a ceiling only when every launch returned success and the bytes arrived: a refused launch would time as "infinitely fast")
    copy_ok = all(rc == 0 for rc in own_rc) and bool(torch.equal(other, buf))
    del other
    for k in range(KMIN, KMAX + 1):
        ms = timed(lambda: (apps[k].forward(), apps[k].inverse()), reps)
        launches, kern = apps[k].launch_info()
        # forward transforms only, every launch sweeping front to back: no pairing with the inverse, no zig-zag reuse of what the
        # previous launch left in the Infinity Cache
        os.environ["VKFFT_MI355X_NO_REVERSE"] = "1"
        fwd = api.App([1 << k], (1 << TOTAL_LOG2) >> k, device_index=dev, buffer_ptr=buf.data_ptr(), stream=stream if stream else None)
        del os.environ["VKFFT_MI355X_NO_REVERSE"]
        fwd.forward()
        fms = timed(fwd.forward, 2 * reps)
        fwd.delete()
        buf.uniform_(-1, 1, generator=gen)  # (unnormalised forwards grow the data)
        per_size[k] = dict(pair_ms=round(ms, 4), passes=apps[k].uploads()[0], launches=launches, kernel=kern,
                           alg_GBps=round(2 * GIB2 / (ms * 1e-3) / 1e9, 1), fwd_only_alg_GBps=round(GIB2 / (fms * 1e-3) / 1e9, 1),
                           GFLOPs=round(flops_pair(k) / (ms * 1e-3) / 1e9, 1))
    # dominant kernel = the kernel family the sweep spends most time in, at its slowest size
    fam_time = {}
    for k, v in per_size.items():
        fam_time[v["kernel"]] = fam_time.get(v["kernel"], 0.0) + v["pair_ms"]
    dom = max(fam_time, key=fam_time.get)
    kd = max((k for k in per_size if per_size[k]["kernel"] == dom), key=lambda k: per_size[k]["pair_ms"])
    launches_per_pair = 2 * per_size[kd]["launches"]
    launch_ms = timed(lambda: (apps[kd].forward(), apps[kd].inverse()), reps) / launches_per_pair
    # algorithmic bytes of one launch (16 B per point per transform): a plan of P launches spreads them over P launches
    alg_bytes_launch = GIB2 / per_size[kd]["launches"]
    achieved = alg_bytes_launch / (launch_ms * 1e-3) / 1e9
    roofline = dict(bound="hbm", kernel=dom, time_share=round(fam_time[dom] / sum(fam_time.values()), 3), size_log2N=kd,
                    launches_per_transform=per_size[kd]["launches"], launch_ms=round(launch_ms, 5),
                    achieved=round(achieved, 1), peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBPS, 4),
                    copy_GBps_same_box=round(copy_GBps, 1), copy_GBps_own_float4=round(own_GBps, 1) if copy_ok else None,
                    frac_of_copy=round(achieved / (max(copy_GBps, own_GBps) if copy_ok else copy_GBps), 4), traffic=None)
    # HBM-side bytes per launch from the PMC passes (tools/pmc_probe.py -> tools/summarize_profiles.py); valid only for the build they were
    # collected on, so the newest summary is used only when its source hash is the one of the sources this library was built from
    import glob
    for prof in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            pj = json.load(open(prof))
        except Exception:
            continue
        if pj.get("source_hash") != api.source_hash():
            roofline["traffic_source"] = f"null: {os.path.basename(prof)} was collected on other sources ({pj.get('source_hash')} vs {api.source_hash()})"
            break
        inst = pj.get("by_log2N", {}).get(str(kd))  # the instance bench.py launches at the size the roofline names (not a family's largest figure)
        if inst and dom.split("<")[0] in inst.get("kernel", ""):
            roofline["traffic"] = inst.get("bytes_per_launch")
            roofline["traffic_fetch"] = inst.get("fetch_bytes_corrected"); roofline["traffic_write"] = inst.get("write_bytes")
            roofline["traffic_source"] = (f"profiles/{os.path.basename(prof)} by_log2N[{kd}] = {inst.get('kernel', '')[:90]} (rocprofv3 --pmc, separate passes, sources "
                                          f"{pj.get('source_hash')}; L2<->fabric requests: Infinity-Cache hits included)")
        else:
            roofline["traffic_source"] = f"null: {os.path.basename(prof)} holds no counters for the instance launched at 2^{kd}"
        break

    # ---- CPU baseline (rank 0, N=1 only) ------------------------------------------------------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    for a in apps.values():
        a.delete()
    if rank == 0:
        out = dict(metric="GFLOP/s (5N log2 N) + achieved HBM GB/s, batched 1D C2C fp32, 1/2/4/8 GPU", value=round(value, 1),
                   unit="GFLOP/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 4),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload="batched 1D C2C fp32 in-place, N=2^8..2^22, batch=2^27/N (1 GiB per GPU), FFT+iFFT pair per size per step (sample-0 protocol)",
                               sizes_log2=[KMIN, KMAX], buffer_bytes_per_gpu=8 << TOTAL_LOG2, parallelism=f"batch-sharded x{world}, no collectives",
                               source_hash=api.source_hash(), library_newer_than_sources=bool(api.library_is_current())),
                   alg_GBps=round(bytes_step / (ms_per_step * 1e-3) / 1e9, 1),
                   roundtrip_rel_l2=float(f"{roundtrip_rel_l2:.3e}"), max_abs_err=float(f"{max_abs_err:.3e}"), roundtrip_limit_rel_l2=float(f"{rt_limit:.3e}"),
                   roundtrip_pairs=pairs, per_size=per_size, roofline=roofline, cpu_baseline=cpu)
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


def slab3d_main(args):
    """BASELINE config 5, second half: ONE slab-decomposed 3-D C2C fp32 transform (default 1024^3) over all ranks, one all-to-all
    between the (y,x) sweep and the z sweep (vkfft_amd/distributed.py).  A step = forward + inverse of the volume."""
    import torch
    from vkfft_amd import api
    from vkfft_amd.distributed import SlabFFT3D
    n = int(os.environ.get("VKFFT_BENCH_SLAB_N", "1024"))
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    api.load()
    plan = SlabFFT3D(n, n, n, device_index=local_rank, normalize=True)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1 + rank)
    x = torch.view_as_complex(torch.empty((n // world, n, n, 2), dtype=torch.float32, device="cuda").uniform_(-1, 1, generator=gen))

    def step(v):
        return plan.inverse(plan.forward(v))

    for _ in range(args.warmup):
        x = step(x)
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = step(x)
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms = 1e3 * float(t.item()) / args.steps
    import math
    flops = 2 * 5.0 * n ** 3 * math.log2(n ** 3)
    xb = plan.exchange_bytes_per_rank()
    if rank == 0:
        print(json.dumps(dict(metric="GFLOP/s (5N log2 N), slab-decomposed 3D C2C fp32, one all-to-all per transform", value=round(flops / (ms * 1e-3) / 1e9, 1), unit="GFLOP/s",
                              n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms, 4), higher_is_better=True, scaling="strong", vs_baseline=None,
                              dtype="f32", data="synthetic",
                              config=dict(workload=f"3D C2C {n}^3 fp32, z-slabs over {world} rank(s), forward + inverse per step", plane_groups=plan.G,
                                          exchange_bytes_per_rank_per_transform=xb,
                                          exchange_GBps_per_rank_if_exchange_were_the_whole_step=round(2 * xb / (ms * 1e-3) / 1e9, 1) if world > 1 else None))))
    plan.delete()
    dist.destroy_process_group()


def cpu_baseline():
    """FFTW3 API (the reference's CPU ground-truth path, sample_11_precision_VkFFT_single.cpp:116-132) served by MKL,
    all host cores, on a bounded sample: sizes 2^8, 2^12, 2^16, 2^20 with 2^24 points each, forward+inverse."""
    import numpy as np
    from oracle import oracle as O
    try:
        O.build()
    except Exception:
        return None
    cores = os.cpu_count() or 1
    os.environ.setdefault("MKL_NUM_THREADS", str(cores))
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    pts_log2 = 24
    rng = np.random.default_rng(0)
    x = (rng.uniform(-1, 1, 1 << pts_log2) + 1j * rng.uniform(-1, 1, 1 << pts_log2)).astype(np.complex64)
    flops = 0.0; secs = 0.0
    ks = [8, 12, 16, 20]
    if O.fftw_available():
        threaded = O.fftw_set_threads(cores)  # fftwf_init_threads + fftwf_plan_with_nthreads(all cores); each timing has an untimed warm-up
        for k in ks:
            N = 1 << k; B = (1 << pts_log2) // N
            _, tf = O.fftw_c2c(x, N, B, inverse=False, reps=3)
            _, ti = O.fftw_c2c(x, N, B, inverse=True, reps=3)
            secs += tf + ti; flops += 2 * 5.0 * N * k * B
        return dict(value=round(flops / secs / 1e9, 1), unit="GFLOP/s", cores=cores, kind="reference",
                    sample=f"FFTW3 API via MKL libmkl_rt (FFTW proper is not installed), fftwf_plan_many_dft in-place, N=2^{ks}, 2^{pts_log2} points each, fwd+inv, warm, MKL_NUM_THREADS={os.environ['MKL_NUM_THREADS']}, fftwf_plan_with_nthreads={'yes' if threaded else 'not exported'}",
                    alg_GBps=round(len(ks) * 2 * 16.0 * (1 << pts_log2) / secs / 1e9, 1))
    # fallback: the C restatement of the reference's algorithm (oracle/vkfft_oracle.c), one core
    pts_log2 = 18
    x = x[: 1 << pts_log2]
    for k in ks[:3]:
        N = 1 << k; B = (1 << pts_log2) // N
        t0 = time.perf_counter(); O.c2c(x, (N,), B); O.c2c(x, (N,), B, inverse=True); secs += time.perf_counter() - t0
        flops += 2 * 5.0 * N * k * B
    return dict(value=round(flops / secs / 1e9, 2), unit="GFLOP/s", cores=1, kind="port",
                sample=f"oracle/vkfft_oracle.c (scalar C restatement), N=2^{ks[:3]}, 2^{pts_log2} points each, fwd+inv")


if __name__ == "__main__":
    main()
