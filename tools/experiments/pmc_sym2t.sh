#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the dominant kernel: default kernels vs two target blobs per lane (sym_two_targets)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_sym2t
for spec in "10000 50 5" "100000 6 1" "1000000 2 1"; do
  for mode in 0 1; do
    set -- $spec
    for counter in FETCH_SIZE WRITE_SIZE; do
      d=gpurun_out/pmc_sym2t/N$1_m${mode}_$counter
      rocprofv3 --pmc $counter --output-format csv -d $d -- python3 bench.py --blobs $1 --steps $2 --warmup $3 --prewarm-ms 0 --no-sweep --no-cpu --no-host-surface --ctx-option sym_two_targets=$mode > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
    done
  done
done
python3 - <<PY
import csv, glob
for N in (10000, 100000, 1000000):
  for mode in (0, 1):
    row = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
      vals = []
      for f in glob.glob("gpurun_out/pmc_sym2t/N%d_m%d_%s/**/*counter_collection.csv" % (N, mode, counter), recursive=True):
        for r in csv.DictReader(open(f)):
          if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in ("sym_kernel<0, true, false>", "sym_coop_kernel<0, true, false>", "sym2t_kernel<0, true>")):
            vals.append(float(r["Counter_Value"]))
      row[counter] = sum(vals) / max(len(vals), 1)
    print("N=%7d sym_two_targets=%d  FETCH_SIZE %14.0f KB  WRITE_SIZE %14.0f KB  -> HBM bytes per launch %.3e" % (N, mode, row["FETCH_SIZE"], row["WRITE_SIZE"], (2 * row["FETCH_SIZE"] + row["WRITE_SIZE"]) * 1024))
PY
find gpurun_out/pmc_sym2t -name "*.csv" -size +1M -delete
