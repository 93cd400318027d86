cd "$(dirname "$0")/../.."
for v in default ocmlsp default ocmlsp; do
  if [ $v == default ]; then L=default; else L=tools/dev/variants/$v.so; fi
  timeout 120 python tools/dev/with_lib.py $L tools/dev/bench_field.py 16 3 2>&1 | tail -1
done
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
