#!/bin/bash
# Collects rocprofv3 evidence for bench.py's dominant kernel on the GPU box:
#   kernel-trace/stats summary + PMC passes (each in its own run, --kernel-trace only).
# usage: profiles/collect_pmc.sh <tag> [bench.py args...]   (run from the repo root via gpurun)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
         "SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
         "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVES SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
         "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc$i -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/pmc$i.log 2>&1
done
python3 $R/profiles/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
