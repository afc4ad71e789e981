"""Aggregate rocprofv3 --pmc CSV passes (tools/pmc_run.sh) per (kernel, grid): mean counter value per dispatch.
usage: python tools/pmc_summary.py gpurun_out/pmc > profiles/xyz.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            key = (row["Kernel_Name"][:48], int(row.get("Grid_Size", 0) or 0))
            agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
print(f"# PMC summary (mean per dispatch) from {root}; FETCH_SIZE/WRITE_SIZE in KiB as reported (gfx950: FETCH_SIZE counts"
      f" 64 B per 128-B request -> double it for wide coalesced reads, MI355X_MICROARCH.md HBM section)")
for key in sorted(agg, key=lambda k: (k[0], k[1])):
    print(f"\n{key[0]}  grid={key[1]}")
    for cname, vals in sorted(agg[key].items()):
        print(f"   {cname:<36} {sum(vals) / len(vals):>18.1f}   (n={len(vals)})")
