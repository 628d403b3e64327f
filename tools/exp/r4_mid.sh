# round 4, mid-round artefacts: the whole GPU suite, the default bench line, the 2^26 multiexp's traffic
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4mid
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r4mid/pytest_gpu.txt
timeout 900 python bench.py 2>gpurun_out/r4mid/bench_stderr.txt | tail -1 > gpurun_out/r4mid/bench_default.json
bash tools/msm_pmc.sh r4mid 26
