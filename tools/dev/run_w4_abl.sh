cd "$(dirname "$0")/../.."
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "winograd4" 2>&1 | tail -2
for v in default "$@"; do
  if [ $v == default ]; then L=default; else L=tools/dev/variants/$v.so; fi
  timeout 200 python tools/dev/with_lib.py $L tools/dev/bench_wino4.py 16 c3 2>&1 | tail -1
done
