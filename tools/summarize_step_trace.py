"""usage: python tools/summarize_step_trace.py <rocprofv3 kernel-trace dir> <steps> <out.md>
Splits the kernel trace of tools/step_kernels.py into forwards -- a forward ends with the similarity GEMM (vtc_similarity:
gemm_kernel<float, 5, ...>) -- and lists, for the LAST <steps> forwards, every kernel between the first and the last launch of
the forward that is not libvtc_hip.so's: torch kernels (at::native::*) and the runtime's own copy / fill kernels."""
import collections, csv, glob, os, sys
d, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))


def kind(n):
    if n.startswith("__amd_rocclr_"):
        return "runtime"
    if "(anonymous namespace)::" in n or "vtcgemm" in n or n.startswith("nonfinite_flag_kernel"):
        return "vtc"
    return "foreign"


ends = [i for i, r in enumerate(rows) if "gemm_kernel<float, 5," in r["Kernel_Name"]]
lines = ["# kernels of a steady-state config-3 forward (rocprofv3 --kernel-trace of tools/step_kernels.py: B = 64, bf16 mode)", "",
         "A forward = everything after the previous forward's similarity GEMM up to and including its own.", ""]
bad = 0
for k in range(len(ends) - steps, len(ends)):
    seg = rows[ends[k - 1] + 1: ends[k] + 1]
    c = collections.Counter(kind(r["Kernel_Name"]) for r in seg)
    foreign = sorted(set(r["Kernel_Name"][:90] for r in seg if kind(r["Kernel_Name"]) == "foreign"))
    runtime = collections.Counter(r["Kernel_Name"] for r in seg if kind(r["Kernel_Name"]) == "runtime")
    bad += c["foreign"]
    dur = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
    lines.append(f"* forward {k - (len(ends) - steps)}: {len(seg)} launches in {dur:.3f} ms: {c['vtc']} of libvtc_hip.so, {c['runtime']} runtime copy/fill "
                 f"({dict(runtime)}), **{c['foreign']} torch / other kernels** {foreign if foreign else ''}")
lines += ["", f"**{bad} foreign kernels inside the last {steps} forwards.**  The runtime entries are the CAM's barrier-word memset "
          "(`hipMemsetAsync`, cam.hip) and the asynchronous D2H copy of the text tower's range-guard flag to pinned host memory (vtc_amd/towers.py)."]
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
