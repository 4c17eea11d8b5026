#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the dominant kernel for row-major + plain numbering vs blocked + XCD-aware numbering
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_order
for spec in "10000 50 5" "24576 20 3" "100000 6 1" "262144 3 1" "1000000 2 1"; do
  set -- $spec
  for cfg in "0 0 0" "1 1 1024"; do
    set -- $spec $cfg
    for counter in FETCH_SIZE WRITE_SIZE; do
      d=gpurun_out/pmc_order/N$1_o$4x$5_$counter
      rocprofv3 --pmc $counter --output-format csv -d $d -- python3 bench.py --blobs $1 --steps $2 --warmup $3 --prewarm-ms 0 --no-sweep --no-cpu --no-host-surface --ctx-option sym_order=$4 --ctx-option sym_xcd=$5 --ctx-option sym_chunk_steps=$6 > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
    done
  done
done
python3 - <<PY
import csv, glob, collections
for N in (10000, 24576, 100000, 262144, 1000000):
  for cfg in ("o0x0", "o1x1"):
    row = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
      vals = []
      for f in glob.glob("gpurun_out/pmc_order/N%d_%s_%s/**/*counter_collection.csv" % (N, cfg, counter), recursive=True):
        for r in csv.DictReader(open(f)):
          if r["Counter_Name"] == counter and ("sym_kernel<0, true, false>" in r["Kernel_Name"] or "sym_coop_kernel<0, true, false>" in r["Kernel_Name"]):
            vals.append(float(r["Counter_Value"]))
      row[counter] = sum(vals) / max(len(vals), 1)
    print("N=%7d %s  FETCH_SIZE %14.0f KB  WRITE_SIZE %14.0f KB  -> HBM bytes per launch %.3e" % (N, "row-major, plain numbering " if cfg == "o0x0" else "blocked, XCD-aware, chunks", row["FETCH_SIZE"], row["WRITE_SIZE"], (2 * row["FETCH_SIZE"] + row["WRITE_SIZE"]) * 1024))
PY
find gpurun_out/pmc_order -name "*.csv" -size +1M -delete
