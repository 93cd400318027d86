#!/bin/bash
# A/B of library variants inside ONE gpurun call (box-to-box variance is +-2.5 %): bench.py per-kernel ms for each variant, interleaved.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for rep in 1 2; do
  for v in default "$@"; do
    if [ "$v" == default ]; then L=default; else L=tools/dev/variants/$v.so; fi
    timeout 200 python tools/dev/with_lib.py $L bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernels']
print('$v', d['value'], d['ms_per_step'], ' '.join(f\"{n.replace('_kernel','')}={k[n]['ms_per_step']:.3f}\" for n in ('conv_wino4_kernel','upconv_wino4_kernel','wino4_input_kernel','conv_wino_kernel','conv_mfma_kernel','upconv_mfma_kernel','torgb_mfma_kernel','fir_act_kernel','triplane_field_kernel') if n in k))"
  done
done
