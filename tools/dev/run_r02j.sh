cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02j; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  " $O/pytest.log | tail -12
timeout 200 python tools/dev/bench_upfirdn.py 2>&1 | grep -v amdgpu.ids | tee $O/upfirdn.log
