cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_WAVES"; do
  i=$((i+1)); rm -rf /tmp/pm; mkdir -p /tmp/pm
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pm -o pmc -- python $R/bench.py --config ${CFG:-c5} --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --other-batches "" > /tmp/pm/log 2>&1
  echo "rc=$?"; tail -2 /tmp/pm/log | cut -c1-300
  DB=$(find /tmp/pm -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_pmc.py $DB | grep -E "kernel \||${PAT:-bf16_kernel}"
done
