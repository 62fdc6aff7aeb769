"""Reproduction harness for the worker death once seen in test_paired_rows_and_odd_dct4_against_the_reference_live on DCT-IV of 1451 reals x 2
(VERDICT r05, weak #1): the test's exact flow — reference library and this library in ONE process, fresh plan per iteration — repeated in a child
process whose exit status and stderr are reported.  python tools/repro_dct4_1451.py [iterations] [mode]   mode: both | ours | ref | big (ours, 4099 rows)"""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(iters, mode):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import numpy as np, torch
    import make_golden
    from helpers import Runner, rel_l2
    from vkfft_amd import api
    run = Runner(api.load(), "gpu")
    ref = None
    if mode in ("both", "ref"):
        ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libvkfft_ref.so")); ref.ref_transform.restype = C.c_int
    N, B, kind = 1451, (4099 if mode == "big" else 2), 14
    worst = 0.0
    for it in range(iters):
        case = dict(kind=kind, shape=(N,), batch=B, dp=0)
        x = np.ascontiguousarray(make_golden.golden_input(case, seed=4 + it)).copy()
        r = x.copy()
        if ref is not None:
            size = (C.c_uint64 * 4)(N)
            rc = ref.ref_transform(C.c_int(kind), C.c_int(1), size, C.c_uint64(B), C.c_int(0), C.c_int(0), C.c_int(0), r.ctypes.data_as(C.c_void_p), C.c_uint64(r.nbytes), None)
            assert rc == 0, rc
        if mode != "ref":
            y, _ = run.transform(x, (N,), B, dct=4)
            if ref is not None:
                worst = max(worst, rel_l2(y.astype(np.float64), r.astype(np.float64)))
            else:
                import scipy.fft
                t = scipy.fft.dct(x.astype(np.float64).reshape(B, N), type=4, axis=1).reshape(-1)
                worst = max(worst, rel_l2(y.astype(np.float64), t))
        if it % 100 == 99:
            print(json.dumps(dict(it=it + 1, worst_rel_l2=worst)), flush=True)
    print(json.dumps(dict(done=iters, mode=mode, worst_rel_l2=worst)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), sys.argv[3]); sys.exit(0)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    mode = sys.argv[2] if len(sys.argv) > 2 else "both"
    env = dict(os.environ); env.setdefault("AMD_LOG_LEVEL", "1"); env.setdefault("HSA_ENABLE_DEBUG", "0")
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(iters), mode], env=env, capture_output=True, text=True)
    print(json.dumps(dict(mode=mode, iterations=iters, returncode=p.returncode, stdout_tail=p.stdout.strip().splitlines()[-3:], stderr_tail=p.stderr.strip().splitlines()[-12:])), flush=True)
