"""Condense rocprofv3 CSV output (kernel-trace stats + PMC passes) into a short text summary and a JSON record.

  python tools/summarize_profile.py OUT_DIR [STEPS]

STEPS = number of timed steps of the profiled bench command: the kernel-trace average over the LAST `STEPS` dispatches
of each kernel is the figure comparable with bench.py's `kernel_ms_avg` (earlier dispatches are pre-warm / warm-up).
Writes OUT_DIR/summary.json with per-kernel durations and PMC averages; HBM traffic per launch is derived as the
MI355X guide prescribes for gfx950: FETCH_SIZE (KB, doubled: the counter reports half of wide coalesced reads) +
WRITE_SIZE (KB), x 1024.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
record = defaultdict(dict)


def find(sub, pat):
  return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace", "*kernel_stats.csv"):
  with open(f) as fh:
    rows = list(csv.DictReader(fh))
  for r in rows:
    print("  %-90s calls=%s total_ns=%s avg_ns=%s pct=%s" % (r.get("Name", "")[:90], r.get("Calls"), r.get("TotalDurationNs"),
                                                          r.get("AverageNs"), r.get("Percentage")))
    record[r.get("Name", "")]["stats_calls"] = int(r.get("Calls") or 0)
    record[r.get("Name", "")]["stats_avg_ns"] = float(r.get("AverageNs") or 0)
for f in find("trace", "*kernel_trace.csv"):
  with open(f) as fh:
    rows = list(csv.DictReader(fh))
  rows.sort(key=lambda r: int(r["Start_Timestamp"]))
  by = defaultdict(list)
  meta = {}
  for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    by[r["Kernel_Name"]].append(d)
    meta[r["Kernel_Name"]] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
  print("== kernel trace (per dispatch, time order) ==")
  for k, v in by.items():
    v2 = sorted(v)
    line = "  %-90s n=%d avg_us=%.2f med_us=%.2f min_us=%.2f" % (k[:90], len(v), sum(v) / len(v) / 1e3, v2[len(v2) // 2] / 1e3, v2[0] / 1e3)
    record[k].update(trace_n=len(v), trace_avg_us=sum(v) / len(v) / 1e3, trace_med_us=v2[len(v2) // 2] / 1e3, trace_min_us=v2[0] / 1e3)
    if steps and len(v) >= steps:
      last = v[-steps:]
      line += "  last%d_avg_us=%.2f" % (steps, sum(last) / len(last) / 1e3)
      record[k]["trace_last_steps"] = steps
      record[k]["trace_last_avg_us"] = sum(last) / len(last) / 1e3
    print(line + "  vgpr/sgpr/lds/wg/grid=%s" % (meta[k],))

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
        record[k][c] = sum(v) / len(v)
for k, d in record.items():
  if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
    d["hbm_traffic_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
with open(os.path.join(out, "summary.json"), "w") as fh:
  json.dump(record, fh, indent=1, sort_keys=True)
