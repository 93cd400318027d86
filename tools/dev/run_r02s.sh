cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in default $VARIANTS; do
  if [ $v == default ]; then unset TDGP_LIB_PATH; else export TDGP_LIB_PATH=tools/dev/variants/$v.so; fi
  for B in 16 8; do echo -n "$v B=$B: "; timeout 120 python tools/dev/bench_field.py $B 3 2>&1 | tail -1 | sed 's/importance.*triplane_field_kernel/ field/'; done
done; done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "field or e2e or full_size_renderer" 2>&1 | tail -2
