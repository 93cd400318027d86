#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line, the rocprofv3 kernel trace of the same command, and the two HBM-traffic
# PMC passes (FETCH_SIZE / WRITE_SIZE need separate passes on gfx950).  Everything lands in gpurun_out/<tag>/.
#   gpurun -- 'bash tools/profile_bench.sh r01'
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 10 --warmup 2"

timeout 400 $BENCH > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"

timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $BENCH --no-cpu-baseline > "$OUT/trace.log" 2>&1
DB=$(find "$OUT/trace" -name '*.db' | head -1)
[ -n "$DB" ] && python "$REPO/tools/rocpd_summary.py" "$DB" "$OUT/kernel_stats"

for C in FETCH_SIZE WRITE_SIZE; do
    timeout 240 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_$C" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline > "$OUT/pmc_$C.log" 2>&1
    echo "pmc $C rc=$?"
    DB=$(find "$OUT/pmc_$C" -name '*.db' | head -1)
    [ -n "$DB" ] && python "$REPO/tools/rocpd_pmc.py" "$DB" "$OUT/pmc_$C.md" | head -12
done
find "$OUT" -name '*.db' -size +20M -delete
