"""Build-time guard for a round-4 finding: a spilled register in gemm_phased_kernel is not a small cost -- every reload is an
`s_waitcnt vmcnt(0)` that drains the epilogue's stores / the K loop's LDS-DMA (profiles/r04_experiments.txt 10: 6-10 spilled registers
in the residual instantiations cost ~3 % per step).  The four tower instantiations of the 256 x 256 kernel and the 16-bit attention
cores must compile without scratch; the attention cores up to 64 tokens must stay within 128 registers (4 waves per SIMD)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _metadata(src, tmp_path):
    out = tmp_path / (os.path.basename(src) + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out), src],
                   check=True, capture_output=True, timeout=900)
    text = out.read_text()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n){0,14}?\s+\.private_segment_fixed_size:\s+(\d+)(?:.*\n){0,14}?\s+\.vgpr_count:\s+(\d+)\n"
                         r"\s+\.vgpr_spill_count:\s+(\d+)", text):
        meta[m.group(1)] = dict(scratch=int(m.group(2)), vgpr=int(m.group(3)), spill=int(m.group(4)))
    return meta


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_tower_gemm_instantiations_compile_without_spills(tmp_path):
    meta = _metadata(os.path.join(ROOT, "vtc_amd", "csrc", "gemm.hip"), tmp_path)
    # MODE 8 / 9 (folded QKV, c_fc + QuickGELU: 16-bit store) and 10 / 11 (residual on the (hi, lo) stream), bf16 (`t`) and f16, DEEP 1
    wanted = [f"gemm_phased_kernelILi{mode}E{types}Li1EEE" for mode, types in
              ((8, "tt"), (9, "tt"), (10, "ft"), (11, "ft"), (8, "5f16_tS1_"), (9, "5f16_tS1_"), (10, "f5f16_t"), (11, "f5f16_t"))]
    # the sweep's distance GEMM: block-minima epilogues with four planes (mode 6) and two (mode 12, round 5), round-3 K loop (`Li0E`)
    wanted += ["gemm_phased_kernelILi6EftLi0EEE", "gemm_phased_kernelILi12EftLi0EEE", "gemm_phased_kernelILi13EftLi0EEE"]
    for w in wanted:
        hits = [k for k in meta if w in k]
        assert len(hits) == 1, (w, hits)
        assert meta[hits[0]]["spill"] == 0 and meta[hits[0]]["scratch"] == 0, (hits[0], meta[hits[0]])


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_16_bit_attention_cores_keep_four_waves_per_simd(tmp_path):
    meta = _metadata(os.path.join(ROOT, "vtc_amd", "csrc", "attention.hip"), tmp_path)
    for types in ("t", "5f16_t"):
        for nt in (1, 2, 3, 4):
            hits = [k for k in meta if f"attn_kernelI{types}Li{nt}EEE" in k]
            assert len(hits) == 1, (types, nt, hits)
            assert meta[hits[0]]["spill"] == 0 and meta[hits[0]]["vgpr"] <= 128, (hits[0], meta[hits[0]])
