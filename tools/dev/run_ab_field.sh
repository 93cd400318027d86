# A/B of per-ray / field kernel variants inside one gpurun call (renderer-only timing, B = 16): bash tools/dev/run_ab_field.sh <variant> [pytest -k expression]
cd "$(dirname "$0")/../.."
V=$1
[ -n "${2:-}" ] && timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$2" 2>&1 | tail -2
for v in default $V default $V; do
  if [ $v == default ]; then L=default; else L=tools/dev/variants/$v.so; fi
  timeout 120 python tools/dev/with_lib.py $L tools/dev/bench_field.py 16 3 2>&1 | tail -1
done
