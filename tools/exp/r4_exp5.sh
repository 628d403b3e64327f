# round 4, experiment 5: kernel traces of the staged partition
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
rocprofv3 --kernel-trace --stats -d gpurun_out/r4/kt5_msm26 -o kt -- python tools/msm_trace_run.py 26 > gpurun_out/r4/exp5_msm26.log 2>&1
rocprofv3 --kernel-trace -d gpurun_out/r4/kt5_p22 -o kt -- python tools/proof_run.py 22 8 > gpurun_out/r4/exp5_p22.log 2>&1
python -m pytest tests/test_gpu_prove.py tests/test_gpu_msm.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4/exp5_tests.txt
