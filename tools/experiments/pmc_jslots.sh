#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of sym_kernel: default build + order 1 vs RMB_SYM_JSLOTS=4 build + order 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_jslots
for spec in "100000 4 1" "1000000 2 1"; do
  for cfg in "base.so 1 1024" "j4.so 2 1024" "j4.so 2 2048"; do
    set -- $spec $cfg
    for counter in FETCH_SIZE WRITE_SIZE; do
      d=gpurun_out/pmc_jslots/N$1_$4_o$5c$6_$counter
      RMB_AB_LIB=build/ab/$4 rocprofv3 --pmc $counter --output-format csv -d $d -- python3 tools/experiments/exp_jslots.py $1 -- $5:$6 > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
    done
  done
done
python3 - <<PY
import csv, glob
for N in (100000, 1000000):
  for cfg in ("base.so_o1c1024", "j4.so_o2c1024", "j4.so_o2c2048"):
    row = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
      vals = []
      for f in glob.glob("gpurun_out/pmc_jslots/N%d_%s_%s/**/*counter_collection.csv" % (N, cfg, counter), recursive=True):
        for r in csv.DictReader(open(f)):
          if r["Counter_Name"] == counter and "sym_kernel<0, true, false>" in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
      row[counter] = sum(vals) / max(len(vals), 1)
    print("N=%7d %-16s FETCH_SIZE %14.0f KB  WRITE_SIZE %14.0f KB  -> HBM bytes per launch %.3e" % (N, cfg, row["FETCH_SIZE"], row["WRITE_SIZE"], (2 * row["FETCH_SIZE"] + row["WRITE_SIZE"]) * 1024))
PY
find gpurun_out/pmc_jslots -name "*.csv" -size +1M -delete
