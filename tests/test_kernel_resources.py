"""Compiler-hazard guard (CPU, needs hipcc): a few representative kernel instances are compiled for gfx950 and their code-object metadata is checked.
Neither the emulator nor any parity test can see these hazards, and both have happened:
  * a device function that outgrows the inliner is emitted ONCE out of line, and every kernel that calls it gets a call frame — 1.4 KB of scratch, 130 VGPRs —
    (round 4: the map loops between the instance transforms; 5-10x the time on 1 668 kernels);
  * operands of per-element tests kept live across a persistent tile loop spill scalar registers in the instance that is already at the limit
    (round 4: the 8192-point Bluestein row kernel, 60 lane moves, 11 %)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

SRC = r'''
#include "kernel_mixed.h"
#include "kernel_pow2.h"
namespace vkfft_mi355x {
static const MixedVariant kTable[] = {
VKFFT_MX(float, false, 13, 13, 1, 1, 1, 13, 20)
VKFFT_MX(float, false, 11, 1, 1, 1, 1, 1, 64)
VKFFT_MX(float, false, 19, 8, 1, 1, 1, 10, 32)
VKFFT_MX(double, true, 9, 5, 1, 1, 1, 9, 16)
};
const MixedVariant* resource_probe_table(int* count) { *count = 4; return kTable; }
template __global__ void pow2_blue_kernel<float, Pow2Sched<4, 3, 3, 3>, 1>(const PassParams);
}
'''


def _kernels(asm):
    """{mangled name: (body text, private segment bytes, vgprs)} of every kernel in a device assembly listing"""
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", asm):
        meta[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    out = {}
    for name, (priv, vgpr) in meta.items():
        i = asm.find("\n" + name + ":")
        j = asm.find(".amdhsa_kernel", i)
        out[name] = (asm[i:j] if i >= 0 else "", priv, vgpr)
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_representative_kernels_have_no_call_frames_scratch_or_spill_storms(tmp_path):
    src = tmp_path / "probe.hip"
    src.write_text(SRC)
    asm = tmp_path / "probe.s"
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{ROOT}/vkfft_amd/csrc", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                           str(src), "-o", str(asm)], stderr=subprocess.DEVNULL)
    text = asm.read_text()
    ks = _kernels(text)
    assert len(ks) >= 9, sorted(ks)
    assert "s_swappc_b64" not in text, "a device function is called out of line: every kernel that calls it pays a call frame"
    for name, (body, priv, vgpr) in ks.items():
        if "mixed_row_kernel" in name:
            assert priv == 0, (name, priv)
            assert "scratch_" not in body, name
            assert vgpr <= 128, (name, vgpr)
        if "pow2_blue_kernel" in name:
            lanes = len(re.findall(r"v_readlane_b32|v_writelane_b32", body))
            assert priv == 0 and lanes <= 24, (name, priv, lanes)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_headline_power_of_two_kernels_have_no_scratch(tmp_path):
    """Every instance bench.py launches for 2^13 ... 2^22 (packed-pair rows, kernel_pow2_pk.h; packed-pair fused Four-Step kernels, kernel_pow2_fused_pk.h / _pkh.h):
    0 bytes of scratch.  A reload from scratch inside their software-pipelined loops waits for every vector-memory operation in flight (one wait counter for loads and
    stores on gfx950): round 5 measured 2^22 at 2.03 TB/s with 250-640 bytes of scratch per lane and at 2.73 TB/s without."""
    asm_rows = tmp_path / "rows.s"; asm_fused = tmp_path / "fused.s"
    for unit, out in (("kernels_pow2.hip", asm_rows), ("kernels_fused.hip", asm_fused)):
        subprocess.check_call([HIPCC, "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{ROOT}/vkfft_amd/csrc", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                               f"{ROOT}/vkfft_amd/csrc/{unit}", "-o", str(out)], stderr=subprocess.DEVNULL)
    seen = 0
    for asm in (asm_rows, asm_fused):
        for name, (body, priv, vgpr) in _kernels(asm.read_text()).items():
            if "pow2_row_lean_pk_kernel" in name or "pow2_row_pairs_kernel" in name or "pow2_fused_pk_kernel" in name or "pow2_fused_pkh_kernel" in name:
                seen += 1
                assert priv == 0 and "scratch_" not in body, (name, priv)
    assert seen >= 11, seen  # three row lengths, six two-factor shapes, two shapes of two halves


# Instances that still carry a private segment (scratch), by family: (substring of the demangled name, most bytes per lane allowed).  An EXPLICIT, shrinking list:
# a new instance with scratch, or more bytes than written here, fails the test; an entry that no longer matches anything must be deleted (the test says so).
# Round 5 had 102 such instances (up to 1 248 bytes); round 6 took the tables of the two largest fused Bluestein shapes out of the registers (kernel_blue_r2r.h,
# kernel_pow2.h TABREG) and replaced the Rader-stage kernels (kernel_mixrad.h: none).  None of the instances bench.py launches is on the list.
SCRATCH_ALLOWED = [
    ("mixed_row_kernel", 52),                                        # 29 / 31-point butterflies, mostly fp64: scalar-register spills (0 vector registers spilled); the long rows (tables 6, 12-14): 15625 = 25^3 (8 / 40 bytes) and eight forms between the maps of 10080 ... 16128-point rows (24-52)
    ("mixconv_kernel", 264),                                         # four-stage schedules of the longest Bluestein ladder lengths; 625 = 25 x 25 x 25
    ("opfft_kernel", 96),                                            # the half-length DCT-III maps (op pair 28 / 29) on 500 ... 2000 points
    ("pow2_blue_r2r_kernel<float, vkfft_mi355x::Pow2Sched<4, 4, 3, 3>", 900),   # 16384 points, 1024 threads at 128 registers: the maps of 16 points per thread
    ("pow2_blue_r2r_kernel<float, vkfft_mi355x::Pow2Sched<4, 3, 3, 3>", 244),   # 8192 points: the odd DCT-IV pair only
    ("pow2_blue_r2r_kernel<double, vkfft_mi355x::Pow2Sched<4, 3, 3, 3>", 412),
    ("pow2_blue_kernel<float, vkfft_mi355x::Pow2Sched<4, 4, 3, 3>", 132),
    ("pow2_col_blue_kernel", 316),
    ("pow2_fused_kernel", 140),                                      # round-2 shapes kept as FUV<k> alternatives, not launched by default
    ("conv_pointwise_kernel<double>", 144),
    ("mix_fused_kernel", 640),                                       # the nine instances with the chirp-z hooks (off by default: VKFFT_MI355X_MIXFUSED_BLUE), and the two one-tile-at-a-time shapes (3^11, 5^7: 12 / 32 bytes at the register budget of two workgroups per CU, measured + 6 / + 13 % all the same); the other ten that ship: 0 bytes
]


def test_instances_with_scratch_are_on_the_shrinking_allow_list():
    """profiles/r06_kernel_resources.json (clang's resource remarks over the whole library, tools/kernel_resources.py) against SCRATCH_ALLOWED.  The summary must be of the
    sources in the tree (its hash is checked): rebuild with the remarks and regenerate it after touching a kernel."""
    import json, subprocess, sys
    sys.path.insert(0, ROOT)
    from vkfft_amd import api
    path = os.path.join(ROOT, "profiles", "r06_kernel_resources.json")
    d = json.load(open(path))
    assert d["source_hash"] == api.source_hash(), "profiles/r06_kernel_resources.json is of other sources: rebuild with -Rpass-analysis=kernel-resource-usage and run tools/kernel_resources.py"
    names = list(d["instances_with_scratch"])
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().splitlines() if names else []
    used = set()
    for mangled, pretty in zip(names, dem):
        sc = d["instances_with_scratch"][mangled]["scratch"]
        hit = [i for i, (pat, cap) in enumerate(SCRATCH_ALLOWED) if pat in pretty]
        assert hit, f"{pretty}: {sc} bytes of scratch and not on the allow-list"
        assert any(sc <= SCRATCH_ALLOWED[i][1] for i in hit), f"{pretty}: {sc} bytes of scratch, allowed {[SCRATCH_ALLOWED[i][1] for i in hit]}"
        used.update(hit)
    stale = [SCRATCH_ALLOWED[i][0] for i in range(len(SCRATCH_ALLOWED)) if i not in used]
    assert not stale, f"allow-list entries without a spilling instance (delete them): {stale}"
    for fam in ("mixrad_kernel",):
        assert not any(fam in p for p in dem), fam
    # the fused Four-Step instances of non-power-of-two lengths that ship (template argument BLUE = 0: "..., 10, 0, 4>") carry none
    assert not any("mix_fused_kernel" in p and ", 0, 4>(" in p for p in dem)
