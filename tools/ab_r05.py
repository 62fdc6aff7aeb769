"""Development tool (round 5): A/B of the packed-pair kernels (kernel_pow2_pk.h, kernel_pow2_fused_pk.h, kernel_pow2_fused_pkh.h) against the round-4
shapes through the C-ABI — 1 GiB batched 1-D C2C fp32, FFT + normalised iFFT pairs, result check per configuration (ab_lean.run).
usage: python tools/ab_r05.py [tune] [sizes ...]   one JSON line per (size, configuration)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ab_lean import run, stress

AB = {
    9: [{}, {"P2V9": 1}], 10: [{}, {"P2V10": 1}], 11: [{}, {"P2V11": 1}], 12: [{}, {"P2V12": 1}],
    13: [{}, {"P2V13": 2}], 14: [{}, {"P2V14": 2}], 15: [{}, {"P2V15": 1}, {"P2V15": 2}],
    16: [{}, {"FUV16": 1}], 17: [{}, {"FUV17": 1}], 18: [{}, {"FUV18": 1}], 19: [{}, {"FUV19": 1}], 20: [{}, {"FUV20": 1}, {"FUV20": 2}],
    21: [{}, {"FUV21": 1}], 22: [{}, {"FUV22": 1}],
}
TUNE = {
    16: [{"FUSED_MARGIN": m} for m in (200, 400)] + [{"FUSED_WGS": 1}, {"FUSED_CHUNK_KIB": 2048}],
    18: [{"FUSED_MARGIN": m} for m in (200, 400)] + [{"FUSED_WGS": 1}],
    20: [{"FUSED_MARGIN": m} for m in (100, 200, 400)] + [{"FUSED_CHUNK_KIB": 16384}, {"FUSED_QUEUES": 4}],
    21: [{"FUSED_MARGIN": m} for m in (100, 300, 400)] + [{"FUSED_LAG": 1, "FUSED_RING": 3}, {"FUSED_LAG": 2, "FUSED_RING": 4}, {"FUSED_QUEUES": 4}, {"FUSED_QUEUES": 1}],
    22: [{"FUSED_MARGIN": m} for m in (100, 300, 400)] + [{"FUSED_LAG": 1, "FUSED_RING": 3}, {"FUSED_LAG": 2, "FUSED_RING": 4}, {"FUSED_QUEUES": 4}, {"FUSED_QUEUES": 1}],
}

LAGS = {
    22: [{"FUSED_LAG": d, "FUSED_RING": n} for d, n in ((3, 6), (4, 6), (4, 8), (5, 8), (5, 10), (6, 9), (6, 12), (7, 10))],
    21: [{"FUSED_LAG": d, "FUSED_RING": n} for d, n in ((7, 14), (9, 14), (9, 18), (11, 16), (12, 18), (13, 20))],
    20: [{"FUSED_LAG": d, "FUSED_RING": n} for d, n in ((13, 26), (17, 26), (17, 34), (20, 30), (25, 38))],
    19: [{"FUSED_LAG": d, "FUSED_RING": n} for d, n in ((25, 50), (33, 50), (33, 66), (40, 60), (50, 76))],
}

if __name__ == "__main__":
    args = sys.argv[1:]
    if args[:1] == ["stress"]:
        for k, q in ((16, 3), (18, 5), (20, 6), (19, 3), (17, 7), (21, 3), (22, 5), (21, 7)):
            print(json.dumps(stress(k, {"FUSED_LAG": 1, "FUSED_RING": 4, "FUSED_QUEUES": q}, pairs=60)), flush=True)
        sys.exit(0)
    table = AB
    if args[:1] == ["tune"]:
        table, args = TUNE, args[1:]
    if args[:1] == ["lags"]:
        table, args = LAGS, args[1:]
    for k in ([int(a) for a in args] or sorted(table)):
        for env in table.get(k, []):
            try:
                print(json.dumps(run(k, env)), flush=True)
            except Exception as e:  # a configuration the planner refuses
                print(json.dumps(dict(k=k, cfg=env, error=str(e))), flush=True)
