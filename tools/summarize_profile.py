"""Condense rocprofv3 CSV output (kernel-trace stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
  return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace", "*kernel_stats.csv"):
  with open(f) as fh:
    rows = list(csv.DictReader(fh))
  for r in rows:
    print("  %-90s calls=%s total_ns=%s avg_ns=%s pct=%s" % (r.get("Name", "")[:90], r.get("Calls"), r.get("TotalDurationNs"),
                                                          r.get("AverageNs"), r.get("Percentage")))
for f in find("trace", "*kernel_trace.csv"):
  with open(f) as fh:
    rows = list(csv.DictReader(fh))
  by = defaultdict(list)
  meta = {}
  for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    by[r["Kernel_Name"]].append(d)
    meta[r["Kernel_Name"]] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
  print("== kernel trace (per dispatch) ==")
  for k, v in by.items():
    v2 = sorted(v)
    print("  %-90s n=%d avg_us=%.2f med_us=%.2f min_us=%.2f vgpr/sgpr/lds/wg/grid=%s" % (k[:90], len(v), sum(v) / len(v) / 1e3, v2[len(v2) // 2] / 1e3, v2[0] / 1e3, meta[k]))

print("== PMC (per dispatch averages) ==")
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
  for f in find(sub, "*counter_collection.csv"):
    with open(f) as fh:
      rows = list(csv.DictReader(fh))
    acc = defaultdict(lambda: defaultdict(list))
    for r in rows:
      acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
      for c, v in cs.items():
        print("  %-70s %-22s n=%d avg=%.6g" % (k[:70], c, len(v), sum(v) / len(v)))
