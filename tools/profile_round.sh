#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line, the rocprofv3 kernel trace of the same command, and the three PMC passes
# (FETCH_SIZE / WRITE_SIZE need separate passes on gfx950; the matrix-pipe busy counter gets its own).  Everything lands in
# gpurun_out/<tag>/; tools/pmc_fold.py folds the passes into pmc.json (-> profiles/pmc_latest.json after review).
#   gpurun -- 'bash tools/profile_round.sh r02 [extra bench args]'
set -u
TAG=${1:-r02}
shift || true
EXTRA="$*"
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 20 --warmup 5 $EXTRA"

# (the trace and the counter passes time the HEADLINE batch only: --other-batches "" --no-fid-loop --no-host-probe keep every other launch shape
#  out of the per-kernel averages, which must agree with bench.py's own `roofline.avg_launch_ms`)
if [ -z "${SKIP_BENCH:-}" ]; then
timeout 600 $BENCH > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 900 "$OUT/bench.json"
fi

timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $BENCH --no-cpu-baseline --other-batches "" --no-fid-loop --no-host-probe > "$OUT/trace.log" 2>&1
DB=$(find "$OUT/trace" -name '*.db' | head -1)
[ -n "$DB" ] && python "$REPO/tools/rocpd_summary.py" "$DB" "$OUT/kernel_stats"

PMCBENCH="python $REPO/bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --other-batches \"\" $EXTRA"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum"; do
    N=$(echo $C | cut -d' ' -f1)
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_$N" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --other-batches "" --no-fid-loop --no-host-probe $EXTRA > "$OUT/pmc_$N.log" 2>&1
    echo "pmc $N rc=$?"
    DB=$(find "$OUT/pmc_$N" -name '*.db' | head -1)
    [ -n "$DB" ] && python "$REPO/tools/rocpd_pmc.py" "$DB" "$OUT/pmc_$N.md" | head -8
done
python "$REPO/tools/pmc_fold.py" "$OUT" "$OUT/pmc.json" --bench "$OUT/bench.json"
find "$OUT" -name '*.db' -size +20M -delete
