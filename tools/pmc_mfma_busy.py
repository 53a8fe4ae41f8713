"""Per-kernel matrix-pipe occupancy from one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass.
busy fraction = (SQ_VALU_MFMA_BUSY_CYCLES / SIMDs) / (GRBM_GUI_ACTIVE / XCDs): the share of the cycles the chip ran during the
launch in which a SIMD's matrix pipe was executing (a 16x16x32 bf16 MFMA counts 32 busy cycles -- profiles/r04_mfma_busy.md).
usage: python tools/pmc_mfma_busy.py <rocprofv3 output dir> [min launches]"""
import csv, glob, os, sys, collections, re
d = sys.argv[1]
SIMDS, XCDS = 1024, 8
per = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, dispatch) -> counter -> sum
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"], int(r["Dispatch_Id"]))][r["Counter_Name"]] += float(r["Counter_Value"])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for (k, _), c in per.items():
    k = k.replace("(anonymous namespace)::", "")
    k = re.sub(r"^void ", "", k)
    depth = 0
    for i, ch in enumerate(k):                      # cut the parameter list: the first "(" outside the template arguments
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            k = k[:i]
            break
    a = agg[k]
    a[0] += 1; a[1] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); a[2] += c.get("GRBM_GUI_ACTIVE", 0.0)
rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
print("| kernel | launches | MFMA busy cycles / SIMD / launch | cycles run / launch (GRBM_GUI_ACTIVE / 8) | matrix pipe busy |")
print("|---|---|---|---|---|")
for k, (n, busy, gui) in rows:
    if gui <= 0 or n < (int(sys.argv[2]) if len(sys.argv) > 2 else 1):
        continue
    b, g = busy / SIMDS / n, gui / XCDS / n
    if g < 2000:
        continue
    print(f"| `{k[:110]}` | {n} | {b:,.0f} | {g:,.0f} | {b / g:.3f} |")
