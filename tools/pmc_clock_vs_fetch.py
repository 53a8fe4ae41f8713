"""Held clock against fetched bytes: does the Infinity-Cache re-fetch of operand panels cost clock on a power-capped part?
Input: directories of `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d DIR -- python3 tools/gemm_super_pmc.py`
run under different VTC_GEMM_SUPER values (row tiles per super-row of the tile walk: fewer fetched bytes at larger values).
Per directory and GEMM kernel: launches, mean duration (kernel trace), cycles the chip ran (GRBM_GUI_ACTIVE / 8 XCDs), held clock =
cycles / duration, bytes fetched through the fabric (2 x FETCH_SIZE KiB on gfx950).
usage: python tools/pmc_clock_vs_fetch.py LABEL:DIR [LABEL:DIR ...]"""
import collections, csv, glob, os, sys

print("| tile walk | kernel | launches | duration us | cycles run | held clock GHz | fetched GB / launch | TFLOP/s |")
print("|---|---|---|---|---|---|---|---|")
for arg in sys.argv[1:]:
    label, d = arg.split(":", 1)
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            did = int(r.get("Dispatch_Id") or r.get("Correlation_Id") or 0)
            dur[did] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for did, c in cnt.items():
        if did not in dur or "gemm_phased_kernel" not in dur[did][0]:
            continue
        a = agg["gemm_phased_kernel (402432 x 2304 x 768, bf16 store)"]
        a[0] += 1; a[1] += dur[did][1]; a[2] += c.get("GRBM_GUI_ACTIVE", 0.0) / 8; a[3] += 2 * c.get("FETCH_SIZE", 0.0) * 1024
    for k, (n, ns, cyc, by) in agg.items():
        if n == 0:
            continue
        us = ns / n / 1e3
        print(f"| {label} | `{k}` | {n} | {us:.1f} | {cyc / n:,.0f} | {cyc / ns:.3f} | {by / n / 1e9:.2f} | {2 * 402432 * 2304 * 768 / (ns / n) / 1e3:.0f} |")
