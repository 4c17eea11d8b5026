#!/bin/bash
# node timeline of one graphed GMRES solve: rocprofv3 kernel trace of exp_gmres_nodes.py, last 80 kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NB=${1:-64}
rm -rf gpurun_out/prof_gmres_nodes
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gmres_nodes -- python3 tools/experiments/exp_gmres_nodes.py $NB > gpurun_out/prof_gmres_nodes.log 2>&1 || { tail -20 gpurun_out/prof_gmres_nodes.log; exit 1; }
f=$(ls gpurun_out/prof_gmres_nodes/*/*kernel_trace.csv | head -1)
python3 - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-70:]
print(open("gpurun_out/prof_gmres_nodes.log").read().strip().split("\n")[-1])
prev = None
for r in tail:
  s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
  name = r["Kernel_Name"].split("(")[0][-60:]
  print("%-62s dur %6.2f us  gap %6.2f us" % (name, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
  prev = e
PY
rm -rf gpurun_out/prof_gmres_nodes
