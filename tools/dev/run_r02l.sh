cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02l; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "field or e2e or full_size or config_c" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  " $O/pytest.log | tail -3
for B in 16 8 4; do timeout 120 python tools/dev/bench_field.py $B 3 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc -o pmc -- python $GRAFT_REPO_ROOT/tools/dev/bench_field.py 16 1 > $O/pmc.log 2>&1
DB=$(find $O/pmc -name '*.db' | head -1); [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py "$DB" $O/pmc_fetch.md | grep "kernel\|triplane"
find $O -name '*.db' -delete
