cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "winograd or up2_folded or bench_batches or overlap" 2>&1 | tail -3
for rep in 1 2; do for o in 0 1; do
TDGP_OVERLAP_TORGB=$o timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']
print('overlap=$o', d['value'], d['ms_per_step'], d['other_batches'].get('4',{}).get('value'), 'kernel sum', d['whole_forward']['kernel_ms_sum'])"
done; done
