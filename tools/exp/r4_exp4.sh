# round 4, experiment 4: staged three-level partition -- correctness with the path forced, then A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
ZA_AMD_SORT_STAGED=3 timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_golden_1k.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4/exp4_tests_forced.txt
ZA_AMD_SORT_STAGED=2 timeout 600 python tools/fuzz_prove.py 120 2>&1 | tail -5 > gpurun_out/r4/exp4_fuzz.txt
timeout 600 python tools/ab_env.py ZA_AMD_SORT_STAGED 0,1 22 6 6 > gpurun_out/r4/exp4_ab22.txt 2>&1
timeout 600 python tools/ab_env.py ZA_AMD_SORT_STAGED 0,1 22 3 4 0.0 > gpurun_out/r4/exp4_ab22_dense.txt 2>&1
ZA_AMD_SORT_STAGED=0 timeout 600 python tools/msm_trace_run.py 26 > gpurun_out/r4/exp4_msm26_old.txt 2>&1
timeout 600 python tools/msm_trace_run.py 26 > gpurun_out/r4/exp4_msm26_new.txt 2>&1
