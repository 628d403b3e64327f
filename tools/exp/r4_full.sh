# round 4: the whole GPU suite twice (default and with the staged partition forced everywhere), fuzz
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r4/full_pytest_gpu.txt
ZA_AMD_SORT_STAGED=3 ZA_AMD_PART_STAGE_MIN=0 timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_golden_1k.py tests/test_gpu_prove.py tests/test_gpu_outer.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r4/full_pytest_forced.txt
timeout 1300 python tools/fuzz_prove.py 1200 2>&1 | tail -4 > gpurun_out/r4/full_fuzz.txt
ZA_AMD_SORT_STAGED=2 ZA_AMD_QUAD_W=4 timeout 700 python tools/fuzz_prove.py 600 2>&1 | tail -4 > gpurun_out/r4/full_fuzz_staged_q4.txt
