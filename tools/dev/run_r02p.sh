cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in default $VARIANTS; do
  if [ $v == default ]; then unset TDGP_LIB_PATH; else export TDGP_LIB_PATH=$R/tools/dev/variants/$v.so; fi
  rm -rf /tmp/c5t; mkdir -p /tmp/c5t
  timeout 300 rocprofv3 --kernel-trace -d /tmp/c5t -o trace -- python $R/bench.py --config c5 --steps 4 --warmup 2 --no-cpu-baseline --other-batches "" > /tmp/c5t/log 2>&1
  DB=$(find /tmp/c5t -name '*.db' | head -1)
  echo "== $v"; python $R/tools/rocpd_by_grid.py $DB ${PAT:-conv3_bf16} | tail -n +3
done
