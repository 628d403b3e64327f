# round 4, experiment 8: one-round staged tiles, staged only for dense-ish jobs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
python tools/exp/alloc_probe.py > gpurun_out/r4/exp8_alloc.txt 2>&1
ZA_AMD_SORT_STAGED=3 timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_golden_1k.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4/exp8_tests_forced.txt
ZA_AMD_SORT_STAGED=2 timeout 600 python tools/fuzz_prove.py 60 2>&1 | tail -3 > gpurun_out/r4/exp8_fuzz.txt
timeout 600 python tools/ab_env.py ZA_AMD_SORT_STAGED 0,1 22 6 6 > gpurun_out/r4/exp8_ab22.txt 2>&1
timeout 600 python tools/ab_env.py ZA_AMD_SORT_STAGED 0,1 22 3 4 0.0 > gpurun_out/r4/exp8_ab22_dense.txt 2>&1
timeout 600 python tools/ab_env.py ZA_AMD_SORT_STAGED 0,1 20 6 8 > gpurun_out/r4/exp8_ab20.txt 2>&1
timeout 600 python tools/msm_trace_run.py 26 > gpurun_out/r4/exp8_msm26.txt 2>&1
timeout 600 python tools/msm_trace_run.py 24 > gpurun_out/r4/exp8_msm24.txt 2>&1
ZA_AMD_SORT_STAGED=0 timeout 600 python tools/msm_trace_run.py 24 > gpurun_out/r4/exp8_msm24_old.txt 2>&1
