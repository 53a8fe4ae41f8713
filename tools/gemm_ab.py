"""A/B of GEMM builds in ONE process, with a hipBLASLt yardstick.

    python tools/gemm_ab.py [--libs name=path,...] [--reps 20] [--rounds 3] [--smi] [--shapes square,tower]

Every library named (default: the product build; variants from tools/build_variant.sh, e.g.
`deep0=vtc_amd/lib/variants/libvtc_deep0.so`) is loaded side by side through ctypes and runs `vtc_gemm` on the SAME device
buffers, round-robin (the chip's clock under load drifts: interleaving gives every build the same conditions); torch.mm on the
same operands (rocBLAS / hipBLASLt behind PyTorch-ROCm -- tools only, never the product) is timed in the same rounds.  Random
normal operands (zeros and constants read 15-20 % high: guide 5.4 rule 25).  Outputs of the builds are compared bit for bit.
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

EPI = {"store": 0, "gelu": 1, "resid": 2}
F32, BF16 = 0, 1


def load(path):
    lib = C.CDLL(path)
    lib.vtc_gemm.restype = C.c_int
    lib.vtc_gemm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.vtc_last_error.restype = C.c_char_p
    return lib


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:      # noqa: BLE001
        return f"rocm-smi failed: {e}"
    keep = [ln.strip().split(":", 1)[-1].strip() for ln in out.splitlines() if any(k in ln for k in ("sclk", "Power", "power"))]
    return " | ".join(keep)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--smi", action="store_true")
    ap.add_argument("--shapes", default="square,tower")
    ap.add_argument("--no-blas", action="store_true")
    args = ap.parse_args()
    from vtc_amd import _lib as L
    libs = {"product": load(L.LIB_PATH)}
    for item in [x for x in args.libs.split(",") if x]:
        name, path = item.split("=", 1)
        libs[name] = load(os.path.join(ROOT, path) if not os.path.isabs(path) else path)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    shapes = []
    if "square" in args.shapes:
        shapes += [(4096, 4096, 4096, "store", "4096^3"), (8192, 8192, 8192, "store", "8192^3")]
    if "tower" in args.shapes:   # config 3 at B = 1024: 402 432 token rows
        shapes += [(402432, 2304, 768, "store", "tsf qkv"), (402432, 3072, 768, "gelu", "tsf c_fc"),
                   (402432, 768, 768, "resid", "tsf out_proj"), (402432, 768, 3072, "resid", "tsf c_proj")]
    if "text" in args.shapes:
        shapes += [(226000, 1536, 512, "store", "text qkv (ragged)"), (226000, 2048, 512, "gelu", "text c_fc"), (226000, 512, 2048, "resid", "text c_proj")]
    if "small" in args.shapes:
        shapes += [(19712, 2304, 768, "store", "tsf qkv B=50"), (19712, 768, 3072, "resid", "tsf c_proj B=50")]
    print(f"libs: {list(libs)}; reps {args.reps} x rounds {args.rounds}; random normal bf16 operands", flush=True)
    print("idle:", smi(), flush=True)
    for M, N, K, epi, label in shapes:
        g = torch.Generator(device=dev).manual_seed(M + N + K)
        a = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev, generator=g)
        odt = torch.float32 if epi == "resid" else torch.bfloat16
        outs = {n: torch.zeros(M, N, device=dev, dtype=odt) for n in libs}
        wt = w.t().contiguous() if not args.no_blas else None

        def run(name):
            lib = libs[name]
            rc = lib.vtc_gemm(a.data_ptr(), w.data_ptr(), bias.data_ptr(), outs[name].data_ptr(), M, N, K, BF16, EPI[epi],
                              F32 if epi == "resid" else BF16, 0, stream)
            if rc:
                raise RuntimeError(f"{name}: {lib.vtc_last_error().decode()}")

        names = list(libs) + ([] if args.no_blas else ["torch.mm (hipBLASLt/rocBLAS)", "torch.mm W^T contiguous"])
        blas_out = None if args.no_blas else torch.empty(M, N, device=dev, dtype=torch.bfloat16)

        def call(name):
            if name in libs:
                run(name)
            elif name.startswith("torch.mm W^T"):
                torch.mm(a, wt, out=blas_out)
            else:
                torch.mm(a, w.t(), out=blas_out)

        for n in names:
            call(n); call(n)
        torch.cuda.synchronize()
        # bit-for-bit between the builds (one launch each on zeroed outputs for the accumulating epilogue)
        if len(libs) > 1:
            for n in libs:
                outs[n].zero_()
                run(n)
            torch.cuda.synchronize()
            ref = outs["product"]
            for n in libs:
                if n != "product":
                    same = torch.equal(outs[n], ref)
                    print(f"    {label}: {n} == product bit for bit: {same}" + ("" if same else f"  max diff {(outs[n].float() - ref.float()).abs().max().item():.3e}"), flush=True)
        best = {n: 1e30 for n in names}
        tot = {n: 0.0 for n in names}
        for r in range(args.rounds):
            for n in names:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    call(n)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / args.reps
                best[n] = min(best[n], ms)
                tot[n] += ms
        fl = 2.0 * M * N * K
        line = f"{label:14s} M={M:6d} N={N:5d} K={K:5d} {epi:5s}"
        for n in names:
            line += f" | {n}: {fl / (tot[n] / args.rounds) / 1e9:7.1f} (best {fl / best[n] / 1e9:7.1f}) TF"
        print(line, flush=True)
        if args.smi:
            for _ in range(200):
                call("product")
            print("    under load:", smi(), flush=True)
            torch.cuda.synchronize()
        del a, w, outs, wt, blas_out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
