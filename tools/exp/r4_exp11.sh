# round 4, experiment 11: compact window table over the field-element positions; where the upload of the resident multiexp goes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_golden_1k.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4/exp11_tests.txt
ZA_AMD_SORT_STAGED=3 timeout 1200 python -m pytest tests/test_gpu_prove.py -m gpu -x -q -k "compact or subset" 2>&1 | tail -15 > gpurun_out/r4/exp11_tests_forced.txt
timeout 600 python tools/fuzz_prove.py 90 2>&1 | tail -3 > gpurun_out/r4/exp11_fuzz.txt
timeout 900 python tools/ab_keys.py ZA_AMD_QUAD_W 4,8 22 6 6 > gpurun_out/r4/exp11_ab_qw.txt 2>&1
timeout 900 python tools/ab_keys.py ZA_AMD_CW 0,1 22 6 6 > gpurun_out/r4/exp11_ab_cw.txt 2>&1
ZA_AMD_TRACE=1 timeout 600 python tools/proof_run.py 22 8 > gpurun_out/r4/exp11_trace.log 2>&1
timeout 600 python tools/msm_trace_run.py 26 host > gpurun_out/r4/exp11_msm26_host.txt 2>&1
HSA_ENABLE_SDMA=0 timeout 600 python tools/msm_trace_run.py 26 host > gpurun_out/r4/exp11_msm26_host_nosdma.txt 2>&1
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/r4/kt11_msm26h -o kt -- python tools/msm_trace_run.py 26 host > gpurun_out/r4/exp11_msm26_host_prof.log 2>&1
