cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "bf16 or c5 or torgb" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  [a-zA-Z]" $O/pytest.log | tail -40; grep "bf16\|C5" $O/pytest.log | grep "^  " | head -30
timeout 300 python bench.py --config c5 --steps 10 --warmup 3 --other-batches "" --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench rc=$?"; tail -c 2500 $O/bench_c5.json; tail -3 $O/bench_c5.err
