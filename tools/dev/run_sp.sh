cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "merge_composite_equals or fused_chain or fused_fine or e2e or cut" 2>&1 | tail -3
for v in default base default base; do
  if [ $v == default ]; then L=default; else L=tools/dev/variants/$v.so; fi
  timeout 120 python tools/dev/with_lib.py $L tools/dev/bench_field.py 16 3 2>&1 | tail -1
done
