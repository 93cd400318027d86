#!/bin/bash
# One gpurun call for a field-kernel change: bit-identity / parity tests, interleaved A/B against a variant library, SQ LDS counters of the default build.
#   bash tools/dev/field_round.sh <variant-name[,variant...] or -> <tag> [pmc: 0 to skip]
cd "$(dirname "$0")/../.."
V=${1:--}; TAG=${2:-field}
REPO=$(pwd)
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "field or walk or renderer_strip or e2e_tiny or bench_batches or fp64" 2>&1 | tail -40 > gpurun_out/$TAG/tests.log
tail -3 gpurun_out/$TAG/tests.log
VS=$(echo $V | tr ',' ' ')
for v in default $VS default $VS default $VS; do
  [ "$v" == "-" ] && continue
  if [ $v == default ]; then L=default; else L=tools/dev/variants/$v.so; fi
  timeout 120 python tools/dev/with_lib.py $L tools/dev/bench_field.py 16 3 2>&1 | tail -1 | tee -a gpurun_out/$TAG/ab.log
done
[ "${3:-1}" == "0" ] && exit 0
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $REPO/gpurun_out/$TAG/sq -o pmc -- python $REPO/tools/dev/bench_field.py 16 1 > $REPO/gpurun_out/$TAG/sq.log 2>&1
echo "pmc rc=$?"
DB=$(find $REPO/gpurun_out/$TAG/sq -name '*.db' | head -1)
[ -n "$DB" ] && python $REPO/tools/rocpd_pmc.py "$DB" $REPO/gpurun_out/$TAG/sq.md | head -12
find $REPO/gpurun_out/$TAG -name '*.db' -size +20M -delete
