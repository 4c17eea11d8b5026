#!/bin/bash
# Profiles the headline bench (run on the GPU box via gpurun).  Outputs under gpurun_out/prof_<tag>/.
#   1. rocprofv3 --kernel-trace --stats        -> per-kernel average duration
#   2. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ counters), as MI355X_MICROARCH.md prescribes
# The program itself follows `--` (python3 bench.py ...), never a wrapper.
set -o pipefail
TAG=${1:-r1}
OUT=gpurun_out/prof_${TAG}
ARGS="bench.py --steps 50 --warmup 5 --no-sweep --no-cpu ${BENCH_EXTRA}"
mkdir -p ${OUT}
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d ${OUT}/trace -- python3 ${ARGS} > ${OUT}/trace.log 2>&1 || { tail -20 ${OUT}/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d ${OUT}/pmc_fetch -- python3 ${ARGS} > ${OUT}/pmc_fetch.log 2>&1 || { tail -20 ${OUT}/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d ${OUT}/pmc_write -- python3 ${ARGS} > ${OUT}/pmc_write.log 2>&1 || { tail -20 ${OUT}/pmc_write.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d ${OUT}/pmc_sq -- python3 ${ARGS} > ${OUT}/pmc_sq.log 2>&1 || { tail -20 ${OUT}/pmc_sq.log; exit 1; }
find ${OUT} -name "*.csv" | head -40
python3 tools/summarize_profile.py ${OUT} > ${OUT}/summary.txt 2>&1
cat ${OUT}/summary.txt
