cd $GRAFT_REPO_ROOT
for v in default $VARIANTS; do
  if [ $v == default ]; then unset TDGP_LIB_PATH; else export TDGP_LIB_PATH=tools/dev/variants/$v.so; fi
  echo "== $v"; timeout 200 python tools/dev/bench_conv.py 16 2>&1 | grep "C=" | sed 's/direct.*wino/wino/'
done
