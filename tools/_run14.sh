cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w14
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/w14/tests.log 2>&1; tail -4 gpurun_out/w14/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
