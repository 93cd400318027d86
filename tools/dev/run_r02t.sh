cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/c3t; mkdir -p /tmp/c3t
timeout 300 rocprofv3 --kernel-trace -d /tmp/c3t -o trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --other-batches 4 ${EXTRA:-} > /tmp/c3t/log 2>&1
DB=$(find /tmp/c3t -name '*.db' | head -1)
for pat in ${PATS:-upconv fir_act torgb wino}; do python $R/tools/rocpd_by_grid.py $DB $pat | tail -n +3; done
