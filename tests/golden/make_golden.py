"""Generates tests/golden/ref_vkfft.npz: outputs of the *reference* VkFFT (HIP backend, oracle/_ref/libvkfft_ref.so,
built by oracle/build_ref.sh from /root/reference) on seeded inputs, captured on an MI355X.

Run on the GPU box:   python tests/golden/make_golden.py gpurun_out/golden/ref_vkfft.npz
then copy the file to tests/golden/.  The reference stores no golden vectors of its own;
these fixtures pin our oracle and our library to the reference's actual results.
Inputs are not stored: they are regenerated from (seed, shape) by golden_input()."""
import ctypes as C
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

# kind: 0 C2C, 1 R2C (in-place padded rows), 11..14 DCT-I..IV, 21..24 DST-I..IV
CASES = [
    dict(name="c2c_4096_f32_fwd", kind=0, shape=(4096,), batch=1, dp=0, inverse=0),   # BASELINE config 1
    dict(name="c2c_4096_f32_inv", kind=0, shape=(4096,), batch=1, dp=0, inverse=1),
    dict(name="c2c_8_f32", kind=0, shape=(8,), batch=4, dp=0, inverse=0),
    dict(name="c2c_64_f32", kind=0, shape=(64,), batch=3, dp=0, inverse=0),
    dict(name="c2c_243_f32", kind=0, shape=(243,), batch=2, dp=0, inverse=0),
    dict(name="c2c_1000_f32", kind=0, shape=(1000,), batch=2, dp=0, inverse=0),
    dict(name="c2c_1001_f32", kind=0, shape=(1001,), batch=2, dp=0, inverse=0),
    dict(name="c2c_127_f32", kind=0, shape=(127,), batch=2, dp=0, inverse=0),         # Rader/Bluestein territory
    dict(name="c2c_1009_f32", kind=0, shape=(1009,), batch=1, dp=0, inverse=0),
    dict(name="c2c_32768_f32", kind=0, shape=(32768,), batch=1, dp=0, inverse=0),     # multi-upload (Four-Step) in the reference
    dict(name="c2c_1024_f64", kind=0, shape=(1024,), batch=2, dp=1, inverse=0),
    dict(name="c2c_1080_f64_inv", kind=0, shape=(1080,), batch=1, dp=1, inverse=1),
    dict(name="c2c_32x24_f32", kind=0, shape=(32, 24), batch=2, dp=0, inverse=0),
    dict(name="c2c_16x12x10_f32", kind=0, shape=(16, 12, 10), batch=1, dp=0, inverse=0),
    dict(name="r2c_256_f32", kind=1, shape=(256,), batch=3, dp=0, inverse=0),
    dict(name="r2c_1000_f32", kind=1, shape=(1000,), batch=2, dp=0, inverse=0),
    dict(name="r2c_64x32_f32", kind=1, shape=(64, 32), batch=1, dp=0, inverse=0),
    dict(name="dct1_65_f32", kind=11, shape=(65,), batch=2, dp=0, inverse=0),
    dict(name="dct2_64_f32", kind=12, shape=(64,), batch=2, dp=0, inverse=0),
    dict(name="dct2_100_f32", kind=12, shape=(100,), batch=2, dp=0, inverse=0),
    dict(name="dct3_64_f32", kind=13, shape=(64,), batch=2, dp=0, inverse=0),
    dict(name="dct4_64_f32", kind=14, shape=(64,), batch=2, dp=0, inverse=0),
    dict(name="dct2_32x16_f32", kind=12, shape=(32, 16), batch=1, dp=0, inverse=0),
    dict(name="dct2_64_f64", kind=12, shape=(64,), batch=1, dp=1, inverse=0),
    # round 5: the holes the round-4 review named — DST I..IV, DCT-IV of odd length (same-length form), fp64 R2C, 3-D R2C, multi-upload lengths of the
    # reference (2^20, 2^22), a composite with a Rader stage (2670 = 30 * 89), a Bluestein prime (15319), a 3-D fp64 volume, an inverse R2C (C2R)
    dict(name="dst1_63_f32", kind=21, shape=(63,), batch=2, dp=0, inverse=0),
    dict(name="dst2_64_f32", kind=22, shape=(64,), batch=2, dp=0, inverse=0),
    dict(name="dst3_100_f32", kind=23, shape=(100,), batch=2, dp=0, inverse=0),
    dict(name="dst4_64_f32", kind=24, shape=(64,), batch=2, dp=0, inverse=0),
    dict(name="dst2_48_f64", kind=22, shape=(48,), batch=1, dp=1, inverse=0),
    dict(name="dct4_45_f32", kind=14, shape=(45,), batch=3, dp=0, inverse=0),
    dict(name="dct4_1125_f32", kind=14, shape=(1125,), batch=2, dp=0, inverse=0),
    dict(name="dct3_100_f32", kind=13, shape=(100,), batch=2, dp=0, inverse=0),
    dict(name="r2c_1024_f64", kind=1, shape=(1024,), batch=2, dp=1, inverse=0),
    dict(name="r2c_243_f32", kind=1, shape=(243,), batch=2, dp=0, inverse=0),
    dict(name="r2c_32x24x10_f32", kind=1, shape=(32, 24, 10), batch=1, dp=0, inverse=0),
    dict(name="c2c_2p20_f32", kind=0, shape=(1 << 20,), batch=1, dp=0, inverse=0, sample=257),   # (stored: every 257th bin + the L2 norm of the whole result)
    dict(name="c2c_2p22_f32", kind=0, shape=(1 << 22,), batch=1, dp=0, inverse=0, sample=1031),
    dict(name="c2c_2p16_f32_inv", kind=0, shape=(1 << 16,), batch=2, dp=0, inverse=1),
    dict(name="c2c_2670_f32", kind=0, shape=(2670,), batch=2, dp=0, inverse=0),
    dict(name="c2c_15319_f32", kind=0, shape=(15319,), batch=1, dp=0, inverse=0),
    dict(name="c2c_12x10x8_f64", kind=0, shape=(12, 10, 8), batch=1, dp=1, inverse=0),
    dict(name="c2c_2p14_f64", kind=0, shape=(1 << 14,), batch=1, dp=1, inverse=0),
    # round 6: the Rader-stage composites of the rewritten kernel_mixrad.h — a direct column step of radix 29 (2813 = 29 * 97), P * P (1369 = 37 * 37), two register
    # steps (3232 = 4 * 8 * 101), a real row on the kernel (R2C of 355 = 5 * 71 reals, DCT-II of 328 = 8 * 41), and the DCT-IV of 1451 reals of BASELINE config 4
    dict(name="c2c_2813_f32", kind=0, shape=(2813,), batch=2, dp=0, inverse=0),
    dict(name="c2c_1369_f32_inv", kind=0, shape=(1369,), batch=2, dp=0, inverse=1),
    dict(name="c2c_3232_f32", kind=0, shape=(3232,), batch=1, dp=0, inverse=0),
    dict(name="r2c_355_f32", kind=1, shape=(355,), batch=3, dp=0, inverse=0),
    dict(name="dct2_328_f32", kind=12, shape=(328,), batch=2, dp=0, inverse=0),
    dict(name="dct4_1451_f32", kind=14, shape=(1451,), batch=2, dp=0, inverse=0),
]


def golden_input(case, seed=20260923):
    """Deterministic input of a case as the flat array that is handed to the library (padded layout for R2C)."""
    rng = np.random.default_rng(seed + (sum(case["shape"]) + 7 * case["kind"] + case["dp"]) % (1 << 30))
    rt = np.float64 if case["dp"] else np.float32
    n = int(np.prod(case["shape"])) * case["batch"]
    if case["kind"] == 0:
        v = rng.uniform(-1, 1, 2 * n).astype(rt)
        return v.view(np.complex128 if case["dp"] else np.complex64)
    if case["kind"] == 1:
        W = case["shape"][0]
        rows = n // W
        buf = np.zeros((rows, 2 * (W // 2 + 1)), dtype=rt)
        buf[:, :W] = rng.uniform(-1, 1, (rows, W)).astype(rt)
        return buf.reshape(-1)
    return rng.uniform(-1, 1, n).astype(rt)


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "golden", "ref_vkfft.npz")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libvkfft_ref.so"))
    ref.ref_transform.restype = C.c_int
    res = {}
    for c in CASES:
        x = np.ascontiguousarray(golden_input(c)).copy()
        size = (C.c_uint64 * 4)(*c["shape"])
        up = (C.c_uint64 * 4)()
        r = ref.ref_transform(C.c_int(c["kind"]), C.c_int(len(c["shape"])), size, C.c_uint64(c["batch"]), C.c_int(c["dp"]),
                              C.c_int(c["inverse"]), C.c_int(0), x.ctypes.data_as(C.c_void_p), C.c_uint64(x.nbytes), up)
        print(c["name"], "rc", r, "uploads", list(up)[:len(c["shape"])], flush=True)
        if r == 0:
            res[c["name"]] = x[:: c["sample"]] if c.get("sample") else x
            if c.get("sample"):
                res[c["name"] + "__l2"] = np.array([np.linalg.norm(x.astype(np.complex128))])
            res[c["name"] + "__uploads"] = np.array(list(up), dtype=np.uint64)
    np.savez_compressed(out, **res)
    print("wrote", out, os.path.getsize(out), "bytes")
