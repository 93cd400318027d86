cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16 or c5" 2>&1 | tail -8
timeout 300 python bench.py --config c5 --steps 20 --warmup 5 --other-batches "" 2>&1 | tail -1 > gpurun_out/c5_n.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c5_n.json').read())
print(d['value'], d['ms_per_step'])
for k,v in sorted(d.get('kernels',{}).items(), key=lambda kv:-kv[1].get('ms_per_step',0))[:12]: print(k, v)
PY
