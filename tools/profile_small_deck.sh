#!/bin/bash
# rocprofv3 kernel trace of small-deck time steps (tools/profile_small_deck.py) -> gpurun_out/prof_small_deck_summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small_deck -- python3 tools/profile_small_deck.py > gpurun_out/prof_small_deck.log 2>&1 || { tail -20 gpurun_out/prof_small_deck.log; exit 1; }
f=$(ls gpurun_out/prof_small_deck/*/*kernel_stats.csv | head -1)
python3 - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("gpurun_out/prof_small_deck_summary.txt", "w") as out:
  out.write("# rocprofv3 --kernel-trace --stats -- python3 tools/profile_small_deck.py\n")
  for line in open("gpurun_out/prof_small_deck.log"):
    if line.startswith("bodies"):
      out.write("# " + line)
  out.write("# kernel | calls | total ms | average us | percent of GPU kernel time\n")
  for r in rows[:40]:
    name = r["Name"] if len(r["Name"]) < 110 else r["Name"][:107] + "..."
    out.write("%-110s %7s %9.3f %9.2f %6.2f\n" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print(open("gpurun_out/prof_small_deck_summary.txt").read())
PY
find gpurun_out/prof_small_deck -name "*kernel_trace.csv" -size +2M -delete
