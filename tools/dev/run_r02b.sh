cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout 120 tools/dev/variants/ubench_overlap2 > gpurun_out/r02b/ubench.log 2>&1; cat gpurun_out/r02b/ubench.log
timeout 900 python -m pytest tests -m gpu -x -q -k "field or e2e or full_size or config_c or fused_chain or training_mode or importance_renderer" > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/r02b/pytest.log
timeout 120 python tools/dev/bench_field.py 8 3 > gpurun_out/r02b/field_b8.log 2>&1; tail -2 gpurun_out/r02b/field_b8.log
timeout 120 python tools/dev/bench_field.py 16 3 > gpurun_out/r02b/field_b16.log 2>&1; tail -2 gpurun_out/r02b/field_b16.log
timeout 120 python tools/dev/bench_field.py 1 3 > gpurun_out/r02b/field_b1.log 2>&1; tail -2 gpurun_out/r02b/field_b1.log
