cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02d; mkdir -p $O
timeout 200 tools/dev/variants/ubench_overlap2 > $O/ubench.log 2>&1; cat $O/ubench.log
for v in default field_abl1 field_abl2 field_abl3 field_old; do
  if [ $v == default ]; then unset TDGP_LIB_PATH; else export TDGP_LIB_PATH=tools/dev/variants/$v.so; fi
  timeout 120 python tools/dev/bench_field.py 8 3 2>&1 | tail -1
done
