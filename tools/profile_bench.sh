#!/bin/bash
# Profiles the headline bench (run on the GPU box via gpurun).  Outputs under gpurun_out/prof_<tag>/.
#   1. rocprofv3 --kernel-trace --stats        -> per-kernel average duration
#   2. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ counters), as MI355X_MICROARCH.md prescribes
# The program itself follows `--` (python3 bench.py ...), never a wrapper.
#   tools/profile_bench.sh TAG [BLOBS] [STEPS] [WARMUP] [PREWARM_MS]
set -o pipefail
TAG=${1:-r2}
BLOBS=${2:-10000}
STEPS=${3:-50}
WARMUP=${4:-5}
PREWARM=${5:-0}
OUT=gpurun_out/prof_${TAG}
# BENCH_EXTRA: further bench.py arguments of this pass, e.g. BENCH_EXTRA="--ctx-option sym_coop=2"
ARGS="bench.py --blobs ${BLOBS} --steps ${STEPS} --warmup ${WARMUP} --prewarm-ms ${PREWARM} --no-sweep --no-cpu --no-host-surface ${BENCH_EXTRA}"
mkdir -p ${OUT}
export TMPDIR=/tmp
echo "python3 ${ARGS}" > ${OUT}/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d ${OUT}/trace -- python3 ${ARGS} > ${OUT}/trace.log 2>&1 || { tail -20 ${OUT}/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d ${OUT}/pmc_fetch -- python3 ${ARGS} > ${OUT}/pmc_fetch.log 2>&1 || { tail -20 ${OUT}/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d ${OUT}/pmc_write -- python3 ${ARGS} > ${OUT}/pmc_write.log 2>&1 || { tail -20 ${OUT}/pmc_write.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d ${OUT}/pmc_sq -- python3 ${ARGS} > ${OUT}/pmc_sq.log 2>&1 || { tail -20 ${OUT}/pmc_sq.log; exit 1; }
python3 tools/summarize_profile.py ${OUT} ${STEPS} > ${OUT}/summary.txt 2>&1
grep "^{" ${OUT}/trace.log | tail -1 > ${OUT}/bench_line_under_trace.json
# keep the merge-back small: the raw per-dispatch CSVs of the large passes are not needed once summarised
find ${OUT} -name "*.csv" -size +2M -delete
grep -E "sym_kernel|sym_coop_kernel|sym2t_kernel|ubench|sym_finalize" ${OUT}/summary.txt | head -40
