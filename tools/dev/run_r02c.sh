cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "field or e2e or full_size or config_c or fused_chain or training_mode or importance_renderer" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 120 python tools/dev/bench_field.py 8 3 > $O/field_b8.log 2>&1; tail -1 $O/field_b8.log
TDGP_LIB_PATH=tools/dev/variants/field_old.so timeout 120 python tools/dev/bench_field.py 8 3 > $O/field_old_b8.log 2>&1; tail -1 $O/field_old_b8.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $O/counters.txt; wc -c $O/counters.txt
for V in new old; do
  if [ $V == old ]; then export TDGP_LIB_PATH=$GRAFT_REPO_ROOT/tools/dev/variants/field_old.so; else unset TDGP_LIB_PATH; fi
  for C in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    N=$(echo $C | cut -d' ' -f1)
    timeout 200 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_${V}_$N -o pmc -- python $GRAFT_REPO_ROOT/tools/dev/bench_field.py 8 1 > $O/pmc_${V}_$N.log 2>&1
    DB=$(find $O/pmc_${V}_$N -name '*.db' | head -1)
    [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py "$DB" $O/pmc_${V}_$N.md | grep "kernel\|triplane" 
  done
done
find $O -name '*.db' -delete
