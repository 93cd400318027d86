#!/bin/bash
# Runs on the GPU box (through gpurun): one PMC pass with the matrix-pipe busy counter next to the GPU-active cycle counter.
#   gpurun -- 'bash tools/profile_mfma.sh r01'
# MFMA-busy % of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs) (the gfx94x MfmaUtil formula;
# ROCm 7.2 ships no gfx950 derived-counter section, MI355X_MICROARCH.md).
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d "$OUT/pmc_mfma" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline ${2:-} > "$OUT/pmc_mfma.log" 2>&1
echo "pmc mfma rc=$?"
DB=$(find "$OUT/pmc_mfma" -name '*.db' | head -1)
[ -n "$DB" ] && python "$REPO/tools/rocpd_pmc.py" "$DB" "$OUT/pmc_mfma.md" | head -14
find "$OUT" -name '*.db' -size +20M -delete
