# Round-end refresh on the GPU box: profile passes, the whole GPU suite, the other BASELINE configurations.   gpurun -- 'bash tools/dev/final_round.sh r03'
TAG=${1:-r03}
cd "$(dirname "$0")/../.."
bash tools/profile_round.sh $TAG
mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/$TAG/gpu_tests.log
tail -3 gpurun_out/$TAG/gpu_tests.log
for c in c5 c4 c1 c2; do
  timeout 400 python bench.py --config $c --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/$TAG/${c}_bench.json
  python -c "import json; d=json.load(open('gpurun_out/$TAG/${c}_bench.json')); print('$c', d['value'], d['ms_per_step'], d['config'].get('batch_per_gpu'))"
done
timeout 600 python bench.py --fid-loop --no-cpu-baseline --steps 24 --warmup 3 --other-batches "" 2>/dev/null | tail -1 > gpurun_out/$TAG/fid_bench.json
python -c "import json; d=json.load(open('gpurun_out/$TAG/fid_bench.json')); f=d['fid_loop']; print('fid', d['value'], f['value'], f['batch_gen'], {k: v['value'] for k, v in f['other_batch_gen'].items()})"
