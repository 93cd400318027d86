#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r05cfg
for c in c1 c2 c4; do
  timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-host-probe > gpurun_out/r05cfg/$c.json 2> gpurun_out/r05cfg/$c.err; echo "$c rc=$?"
done
timeout 400 python bench.py --config c5 --steps 20 --warmup 5 --no-host-probe > gpurun_out/r05cfg/c5.json 2> gpurun_out/r05cfg/c5.err; echo "c5 rc=$?"
timeout 400 python bench.py --config c5 --cmax 1024 --steps 10 --warmup 3 --no-host-probe > gpurun_out/r05cfg/c5_cmax1024.json 2> gpurun_out/r05cfg/c5_cmax1024.err; echo "c5 1024 rc=$?"
for lanes in 2 4; do
  timeout 300 python bench.py --graph --fid-lanes $lanes --steps 16 --warmup 3 --no-cpu-baseline --no-host-probe --other-batches "" --batch-gen 4 > gpurun_out/r05cfg/fid_lanes$lanes.json 2> gpurun_out/r05cfg/fid_lanes$lanes.err; echo "lanes $lanes rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05cfg/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        fl=d.get('fid_loop') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], (d.get('other_batches') or {}).get('4',{}).get('value'), 'fid', fl.get('value'), fl.get('lanes'), (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'))
    except Exception as e: print(f, 'ERR', e)
PY
