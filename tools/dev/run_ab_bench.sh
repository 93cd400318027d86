# tests + whole-forward A/B inside one gpurun call: bash tools/dev/run_ab_bench.sh <variant> "<pytest -k expression>"
cd "$(dirname "$0")/../.."
[ -n "${2:-}" ] && timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$2" 2>&1 | tail -2
bash tools/dev/ab_bench.sh $1 2>&1 | tail -4
