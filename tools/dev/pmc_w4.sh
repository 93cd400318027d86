#!/bin/bash
# SQ latency / wait counters of the F(4x4) kernels (development): three rocprofv3 --pmc passes over tools/dev/bench_wino4.py
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/w4pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY"; do
  i=$((i + 1))
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i -o pmc -- python $REPO/tools/dev/bench_wino4.py 16 c3 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
  DB=$(find $OUT/p$i -name '*.db' | head -1)
  [ -n "$DB" ] && python $REPO/tools/rocpd_pmc.py "$DB" $OUT/p$i.md | grep -E "kernel|wino4" | head -5
done
find $OUT -name '*.db' -size +20M -delete
