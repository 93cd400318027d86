cd "$(dirname "$0")/../.."
timeout 300 python tools/dev/bench_wino4.py 16 c4 2>&1 | tail -1
timeout 300 python tools/dev/bench_wino4.py 4 c4 2>&1 | tail -1
timeout 600 python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_c4_r04a.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_c4_r04a.json").read())
print('C4', d["value"], d["ms_per_step"], d["other_batches"]["4"]["value"])
for k,v in d["kernels"].items():
    if v["ms_per_step"]>0.1: print(' ', k, v["ms_per_step"], v["launches_per_step"], v.get("tflops"), v.get("frac_of_own_peak"))
PY
