# round 4, experiment 1: accumulation stream barred from R compute units (ZA_AMD_ACC_RESERVE)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_msm.py tests/test_gpu_golden_1k.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r4/exp1_tests.txt
python tools/ab_env.py ZA_AMD_ACC_RESERVE 0,8,16,32 22 6 6 > gpurun_out/r4/exp1_ab.txt 2>&1
ZA_AMD_ACC_RESERVE=16 python tools/ab_env.py ZA_AMD_H_PARTS 2,3,4 22 5 6 > gpurun_out/r4/exp1_ab_parts.txt 2>&1
ZA_AMD_ACC_RESERVE=16 rocprofv3 --kernel-trace -d gpurun_out/r4/kt_res16 -o kt -- python tools/proof_run.py 22 8 > gpurun_out/r4/exp1_trace.log 2>&1
rocprofv3 --kernel-trace -d gpurun_out/r4/kt_res0 -o kt -- python tools/proof_run.py 22 8 > gpurun_out/r4/exp1_trace0.log 2>&1
