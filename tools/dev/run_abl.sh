#!/bin/bash
# A/B of field-kernel variant libraries inside one gpurun call: bash tools/dev/run_abl.sh default v1 v2 ...   (renderer-only timing, B = 16)
cd "$(dirname "$0")/../.."
for v in "$@"; do
  if [ $v == default ]; then L=default; else L=tools/dev/variants/$v.so; fi
  timeout 120 python tools/dev/with_lib.py $L tools/dev/bench_field.py 16 3 2>&1 | tail -1
done
