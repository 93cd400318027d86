cd "$(dirname "$0")/../.."
for rep in 1 2; do for f in 1 0; do for o in 1 0; do
TDGP_OVERLAP_TORGB=$o TDGP_FOLD_UP2=$f timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 ${1:-} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']
print('fold=$f overlap=$o', d['value'], d['ms_per_step'], d['other_batches'].get('4',{}).get('value'), 'kernel sum', d['whole_forward']['kernel_ms_sum'], ' '.join(f\"{n.replace('_kernel','')}={k[n]['ms_per_step']:.2f}\" for n in k if k[n]['ms_per_step'] > 0.25))"
done; done; done
