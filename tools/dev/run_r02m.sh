cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in default field_w3; do
  if [ $v == default ]; then unset TDGP_LIB_PATH; else export TDGP_LIB_PATH=tools/dev/variants/$v.so; fi
  for B in 16 8; do timeout 120 python tools/dev/bench_field.py $B 3 2>&1 | tail -1 | sed 's/importance.*triplane_field_kernel/ field/'; done
done; done
