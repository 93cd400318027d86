cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  " $O/pytest.log | tail -8; grep "integer rows" $O/pytest.log
for B in 8 16; do timeout 120 python tools/dev/bench_field.py $B 3 2>&1 | tail -1; done
