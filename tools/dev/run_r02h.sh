cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02h; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "chunked or full_size or e2e" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  " $O/pytest.log | tail -4
run() { timeout 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --other-batches "" "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']
print('$*', d['value'], d['ms_per_step'], ' '.join(f\"{n.split('_')[0]}={k[n]['ms_per_step']:.2f}\" for n in ('conv_mfma_kernel','upconv_mfma_kernel','torgb_mfma_kernel','fir_act_kernel','triplane_field_kernel','merge_composite_kernel','importance_from_coarse_kernel')))"; }
run --batch 16 --chunk 0
run --batch 16 --chunk 2 --chunk-from 128
run --batch 16 --chunk 4 --chunk-from 128
run --batch 16 --chunk 8 --chunk-from 128
run --batch 16 --chunk 2 --chunk-from 256
run --batch 16 --chunk 4 --chunk-from 256
run --batch 16 --chunk 4 --chunk-from 64
run --batch 16 --chunk 2 --chunk-from 64
run --batch 16 --chunk 1 --chunk-from 256
run --batch 16 --chunk 0
run --batch 8 --chunk 0
run --batch 8 --chunk 2 --chunk-from 128
