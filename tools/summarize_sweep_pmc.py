"""usage: python tools/summarize_sweep_pmc.py <out.json> <commit> N:<fetch_dir>:<write_dir>:<sweeps> [N:...]
HBM bytes of ONE whole N x N EXACT sweep (every launch of the library between two sweeps) from separate rocprofv3 --pmc FETCH_SIZE /
--pmc WRITE_SIZE passes of tools/sweep_profile.py (MI355X_MICROARCH.md "HBM": counters in KiB, reads = 2 x FETCH_SIZE on gfx950, writes =
WRITE_SIZE), per kernel and summed, beside the 2 x 8 N^2 bytes a materialised fp32 matrix would cost.  bench.py reads the JSON."""
import csv, glob, json, os, re, sys, datetime
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", n).replace("unsigned short", "bf16")


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    per = defaultdict(lambda: [0, 0.0])
    if not f:
        return per
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            n = short(r["Kernel_Name"])
            if "at::native" in n or n.startswith("at::"):
                continue
            per[n][0] += 1
            per[n][1] += float(r["Counter_Value"])
    return per


out, commit = sys.argv[1], sys.argv[2]
res = {"commit": commit, "date": datetime.datetime.utcnow().strftime("%Y-%m-%d"),
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/sweep_profile.py; reads = 2 x FETCH_SIZE KiB "
                 "(gfx950), writes = WRITE_SIZE KiB; all launches of the library in the process divided by the number of sweeps", "sweeps": {}}
lines = []
for spec in sys.argv[3:]:
    N, fd, wd, sweeps = spec.split(":")
    N, sweeps = int(N), int(sweeps)
    fe, wr = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    rd = sum(2 * v[1] * 1024 for v in fe.values()) / sweeps
    wt = sum(v[1] * 1024 for v in wr.values()) / sweeps
    mat = 16.0 * N * N
    res["sweeps"][str(N)] = {"read_bytes": round(rd), "write_bytes": round(wt), "total_bytes": round(rd + wt),
                             "materialised_matrix_bytes": round(mat), "fraction_of_materialised": round((rd + wt) / mat, 4), "sweeps_in_pass": sweeps}
    lines += [f"## N = {N}: {rd / 1e6:.1f} MB read + {wt / 1e6:.1f} MB written per sweep = {(rd + wt) / 1e6:.1f} MB "
              f"({100 * (rd + wt) / mat:.1f} % of the {mat / 1e9:.2f} GB of a materialised fp32 matrix, both directions)", "",
              "| kernel | launches per sweep | read MB per sweep | write MB per sweep |", "|---|---|---|---|"]
    for n in sorted(set(fe) | set(wr), key=lambda n: -(2 * fe[n][1] + wr[n][1])):
        r_, w_ = 2 * fe[n][1] * 1024 / sweeps / 1e6, wr[n][1] * 1024 / sweeps / 1e6
        if r_ + w_ >= 0.05:
            lines.append(f"| `{n}` | {max(fe[n][0], wr[n][0]) / sweeps:.2f} | {r_:.2f} | {w_:.2f} |")
    lines.append("")
json.dump(res, open(out, "w"), indent=1)
md = out.replace(".json", ".md")
open(md, "w").write("# HBM bytes of the N x N EXACT sweep from PMC counters (round 5)\n\n" + res["method"] + "\n\n" + "\n".join(lines))
print(open(md).read())
