cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02k; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "modconv or fused_layers or e2e or full_size or config_c or bf16" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  " $O/pytest.log | tail -5
run() { timeout 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --other-batches "" "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']
print('$*', d['value'], d['ms_per_step'], ' '.join(f\"{n.split('_')[0]}={k[n]['ms_per_step']:.2f}\" for n in k if k[n]['ms_per_step'] > 0.1))"; }
run --batch 16
run --batch 16 --config c5
run --batch 4
