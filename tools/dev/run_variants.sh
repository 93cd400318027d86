# A/B of several variant libraries inside one gpurun call (renderer-only timing, B = 16): bash tools/dev/run_variants.sh v1 v2 ...   (default is always included, twice)
cd "$(dirname "$0")/../.."
for v in default "$@" default "$@"; do
  if [ $v == default ]; then L=default; else L=tools/dev/variants/$v.so; fi
  timeout 120 python tools/dev/with_lib.py $L tools/dev/bench_field.py 16 3 2>&1 | tail -1
done
