#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 --pmc passes over tools/bin/pmc_calibrate (known byte counts per access shape) ->
# gpurun_out/<tag>/cal_<pass>.md; tools/pmc_fold.py --calibration turns them into the per-access-shape FETCH_SIZE factors.
#   gpurun -- 'bash tools/profile_calibrate.sh cal [MiB]'
set -u
TAG=${1:-cal}
MIB=${2:-2048}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/p$i" -o cal -- "$REPO/tools/bin/pmc_calibrate" $MIB > "$OUT/p$i.log" 2>&1
    echo "pass $i ($C) rc=$?"
    DB=$(find "$OUT/p$i" -name '*.db' | head -1)
    [ -n "$DB" ] && python "$REPO/tools/rocpd_pmc.py" "$DB" "$OUT/cal_p$i.md"
done
find "$OUT" -name '*.db' -size +20M -delete
