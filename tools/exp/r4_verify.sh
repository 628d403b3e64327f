cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r4/verify_smoke.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r4/verify_pytest.txt
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4/verify_bench.json
