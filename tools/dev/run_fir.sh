cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "modconv or up or fir or e2e or backbone or noise" 2>&1 | tail -3
bash tools/dev/ab_bench.sh firold 2>&1 | tail -4
