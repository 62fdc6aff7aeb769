"""A sampling of the size lists of the reference's multi-dimensional benchmarks (sample 3: C2C, sample 6: R2C, sample 100: DCT), this library and the
reference VkFFT-HIP in the same process.  usage: python tools/perf_samples.py [c2c|r2c|dct ...]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_configs import run
SETS = {
    "c2c": (0, [(720, 480), (1920, 1080), (3840, 2160), (7680, 4320), (64, 64), (512, 512), (2048, 1024), (4096, 4096), (8192, 8192), (16384, 8192),
                (16, 16, 16), (64, 64, 64), (256, 256, 128), (512, 512, 512)]),
    "r2c": (1, [(64, 64), (1024, 256), (4096, 256), (4096, 4096), (1280, 720), (3840, 2160), (32, 32, 32), (256, 256, 256), (2048, 1024, 8), (4096, 512, 8)]),
    "dct": (12, [(512, 256), (720, 480), (300, 300), (300, 300, 300), (500, 500, 500), (700, 700, 100), (4096, 1024), (64, 64, 64), (512, 512, 512)]),
}
for name in (sys.argv[1:] or list(SETS)):
    kind, shapes = SETS[name]
    for shape in shapes:
        try:
            print(json.dumps(run(kind, shape, False, total_log2=26)), flush=True)
        except Exception as e:
            print(json.dumps(dict(kind=kind, shape=list(shape), error=str(e))), flush=True)
