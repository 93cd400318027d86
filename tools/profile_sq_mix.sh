#!/bin/bash
# Runs on the GPU box (through gpurun): SQ instruction-mix counters of the main kernels, three PMC passes (separate from the kernel trace
# / HBM passes of profile_round.sh; rocprofv3 --pmc only, no trace domains).   gpurun -- 'bash tools/profile_sq_mix.sh r03'
#   -> gpurun_out/<tag>/sq_mix.md  (per-launch averages, summed over XCDs / SEs)
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/sq_$i" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --other-batches "" --no-fid-loop --no-host-probe > "$OUT/sq_$i.log" 2>&1
    echo "sq pass $i rc=$?"
    DB=$(find "$OUT/sq_$i" -name '*.db' | head -1)
    [ -n "$DB" ] && python "$REPO/tools/rocpd_pmc.py" "$DB" "$OUT/sq_$i.md" | head -3
done
find "$OUT" -name '*.db' -size +20M -delete
