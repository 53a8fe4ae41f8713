"""Is the big GEMM power-bound?  Loop one 16-bit GEMM shape for a few seconds and sample `rocm-smi` (clocks, power) from a child
process meanwhile; print TFLOP/s for random-normal, constant and zero operands (the data a matrix pipe toggles sets its power).
Usage: python tools/gemm_clock_probe.py [M N K]"""
import subprocess
import sys
import time

import torch

from vtc_amd import ops


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:      # noqa: BLE001
        return [f"rocm-smi failed: {e}"]
    keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power", "power"))]
    return keep


def main():
    M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (402432, 2304, 768)
    dev = torch.device("cuda:0")
    print("idle:", smi(), flush=True)
    for name in ("normal", "ones", "zeros"):
        if name == "normal":
            a = torch.randn(M, K, device=dev).to(torch.bfloat16)
            w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        elif name == "ones":
            a = torch.ones(M, K, device=dev, dtype=torch.bfloat16)
            w = torch.ones(N, K, device=dev, dtype=torch.bfloat16)
        else:
            a = torch.zeros(M, K, device=dev, dtype=torch.bfloat16)
            w = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            ops.gemm(a, w, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 1500
        e0.record()
        for _ in range(reps):
            ops.gemm(a, w, out=out)
        e1.record()
        time.sleep(0.7)
        s1 = smi()
        time.sleep(0.5)
        s2 = smi()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{name}: {ms * 1e3:.1f} us = {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s", flush=True)
        print("   under load:", s1, flush=True)
        print("   under load:", s2, flush=True)


if __name__ == "__main__":
    main()
