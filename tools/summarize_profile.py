"""Turn rocprofv3 output directories into the small summaries committed under profiles/.

usage: python tools/summarize_profile.py <stats_dir> <fetch_dir> <write_dir> <out_prefix> [steps_in_stats_run] [workload_tag] [commit]

  <stats_dir>  rocprofv3 --kernel-trace --stats --output-format csv
  <fetch_dir>  rocprofv3 --pmc FETCH_SIZE   (separate pass)
  <write_dir>  rocprofv3 --pmc WRITE_SIZE   (separate pass)
HBM bytes follow MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE reports half
the bytes of wide coalesced streams, so reads = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def find(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return f[0] if f else None


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    return name.replace("unsigned short", "bf16")


BF16 = "16-bit-operand GEMM (gemm_phased_kernel + gemm_kernel<bf16 | f16_t,...> + qkv_attn_kernel)"


def is_bf16_gemm(n):
    return n.startswith("gemm_phased_kernel") or n.startswith("gemm_kernel<bf16") or n.startswith("gemm_kernel<f16_t") or \
        n.startswith("qkv_attn_kernel")


def main():
    stats_dir, fetch_dir, write_dir, out = sys.argv[1:5]
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    workload = sys.argv[6] if len(sys.argv) > 6 else None
    commit = sys.argv[7] if len(sys.argv) > 7 else None
    rows = list(csv.DictReader(open(find(stats_dir, "*kernel_stats.csv"))))
    total = sum(int(r["TotalDurationNs"]) for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary", "",
             f"total kernel time {total/1e6:.3f} ms over {steps} steps = {total/1e6/steps:.3f} ms/step", "",
             "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    agg = defaultdict(lambda: [0, 0])
    for r in rows:
        n = short(r["Name"])
        if float(r["Percentage"]) >= 0.05:
            lines.append(f"| `{n}` | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
        cls = BF16 if is_bf16_gemm(n) else ("gemm_kernel<float,...>" if n.startswith("gemm_kernel<float") else None)
        if cls:
            agg[cls][0] += int(r["Calls"]); agg[cls][1] += int(r["TotalDurationNs"])
    lines += ["", "## aggregated over template instantiations", "", "| kernel | calls | total ms | avg us |", "|---|---|---|---|"]
    for k, (c, t) in agg.items():
        lines.append(f"| `{k}` | {c} | {t/1e6:.3f} | {t/1e3/c:.2f} |")

    def pmc(d, counter):
        f = find(d, "*counter_collection.csv")
        per = defaultdict(lambda: [0, 0.0])
        if not f:
            return per
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                n = short(r["Kernel_Name"])
                per[n][0] += 1; per[n][1] += float(r["Counter_Value"])
        return per
    fe, wr = pmc(fetch_dir, "FETCH_SIZE"), pmc(write_dir, "WRITE_SIZE")
    lines += ["", "## HBM traffic from PMC (separate passes; reads = 2 x FETCH_SIZE KiB on gfx950, writes = WRITE_SIZE KiB)", "",
              "| kernel | launches | read MB/launch | write MB/launch |", "|---|---|---|---|"]
    tot = defaultdict(lambda: [0, 0.0, 0.0])
    for n in sorted(set(fe) | set(wr)):
        c = max(fe[n][0], wr[n][0])
        rd = 2 * fe[n][1] * 1024 / max(1, fe[n][0]) / 1e6
        wt = wr[n][1] * 1024 / max(1, wr[n][0]) / 1e6
        if rd + wt > 1.0:
            lines.append(f"| `{n}` | {c} | {rd:.1f} | {wt:.1f} |")
        if is_bf16_gemm(n):
            tot[BF16][0] += c; tot[BF16][1] += 2 * fe[n][1] * 1024; tot[BF16][2] += wr[n][1] * 1024
    for k, (c, rd, wt) in tot.items():
        lines.append(f"| **{k} (all)** | {c} | {rd/c/1e6:.1f} | {wt/c/1e6:.1f} |")
        lines += ["", f"traffic per launch of `{k}`: {(rd+wt)/c/1e6:.1f} MB"]
        # bench.py reads this file for roofline.traffic (bytes per launch of the dominant kernel class)
        import json
        json.dump({"kernel": k, "launches": c, "read_bytes_per_launch": rd / c, "write_bytes_per_launch": wt / c,
                   "traffic_bytes_per_launch": (rd + wt) / c, "workload": workload, "commit": commit,
                   "date": __import__("datetime").datetime.utcnow().strftime("%Y-%m-%d"), "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes",
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --no-extra --no-cpu --no-sweep`; "
                             "reads = 2 x FETCH_SIZE KiB (gfx950 correction), writes = WRITE_SIZE KiB"},
                  open(out + "_traffic.json", "w"), indent=1)
    open(out + ".md", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
