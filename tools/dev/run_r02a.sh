cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/r02a/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --other-batches 4,8 > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/r02a/bench.json
TDGP_LIB_PATH=tools/dev/variants/field32.so timeout 120 python tools/dev/bench_field.py 1 1 > gpurun_out/r02a/field32.log 2>&1; tail -12 gpurun_out/r02a/field32.log
timeout 120 python tools/dev/bench_field.py 8 3 > gpurun_out/r02a/field_b8.log 2>&1; tail -3 gpurun_out/r02a/field_b8.log
