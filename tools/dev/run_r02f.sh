cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02f; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "field or e2e or full_size or config_c or fused_chain or training_mode or importance_renderer" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log
for rep in 1 2; do
for v in default field_old; do
  if [ $v == default ]; then unset TDGP_LIB_PATH; else export TDGP_LIB_PATH=tools/dev/variants/$v.so; fi
  for B in 8 16 4; do timeout 120 python tools/dev/bench_field.py $B 3 2>&1 | tail -1 | sed 's/importance.*triplane_field_kernel/ field/'; done
done
done
