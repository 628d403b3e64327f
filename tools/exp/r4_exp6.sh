# round 4, experiment 6: staged partition with the full-tile fast path and the single-pass last level
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
ZA_AMD_SORT_STAGED=3 timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_golden_1k.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4/exp6_tests_forced.txt
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_golden_1k.py tests/test_gpu_prove.py tests/test_gpu_curve.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4/exp6_tests.txt
ZA_AMD_SORT_STAGED=2 timeout 600 python tools/fuzz_prove.py 100 2>&1 | tail -3 > gpurun_out/r4/exp6_fuzz.txt
timeout 600 python tools/ab_env.py ZA_AMD_SORT_STAGED 0,1 22 6 6 > gpurun_out/r4/exp6_ab22.txt 2>&1
timeout 600 python tools/ab_env.py ZA_AMD_PART_SINGLE 0,1 22 6 6 > gpurun_out/r4/exp6_ab22_single.txt 2>&1
timeout 600 python tools/ab_env.py ZA_AMD_SORT_STAGED 0,1 22 3 4 0.0 > gpurun_out/r4/exp6_ab22_dense.txt 2>&1
timeout 600 python tools/ab_env.py ZA_AMD_PART_SINGLE 0,1 14 10 20 > gpurun_out/r4/exp6_ab14_single.txt 2>&1
timeout 600 python tools/msm_trace_run.py 26 > gpurun_out/r4/exp6_msm26.txt 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/r4/kt6_msm26 -o kt -- python tools/msm_trace_run.py 26 > gpurun_out/r4/exp6_msm26_prof.log 2>&1
rocprofv3 --kernel-trace -d gpurun_out/r4/kt6_p22 -o kt -- python tools/proof_run.py 22 8 > gpurun_out/r4/exp6_p22.log 2>&1
