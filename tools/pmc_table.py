"""Print per-dispatch sums of rocprofv3 counter_collection.csv files for kernels matching a substring.
usage: python tools/pmc_table.py <substring> <dir> [<dir> ...]"""
import csv, glob, os, sys, collections
sub = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if sub not in r["Kernel_Name"]:
                continue
            k = (int(r["Dispatch_Id"]), r["Grid_Size"], r["Counter_Name"])
            agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
        for (disp, grid, name), v in agg.items():
            print(f"{os.path.basename(d):12s} dispatch {disp:4d} grid {grid:>8s} {name:36s} {v:16.0f}")
