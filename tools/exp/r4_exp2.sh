# round 4, experiment 2: device curve selftest; accumulate at two waves per SIMD (ZA_AMD_ACC_W2)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_curve.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4/exp2_curve.txt
python tools/ab_env.py ZA_AMD_ACC_W2 0,1 22 6 6 > gpurun_out/r4/exp2_ab_w2.txt 2>&1
ZA_AMD_ACC_W2=1 python tools/ab_env.py ZA_AMD_H_PARTS 2,3,4 22 5 6 > gpurun_out/r4/exp2_ab_w2_parts.txt 2>&1
python tools/ab_env.py ZA_AMD_PART_STAGE_MIN 16384,0 22 5 6 > gpurun_out/r4/exp2_ab_stage.txt 2>&1
ZA_AMD_TRACE=1 python tools/proof_run.py 22 8 > gpurun_out/r4/exp2_trace0.log 2>&1
ZA_AMD_TRACE=1 ZA_AMD_ACC_W2=1 python tools/proof_run.py 22 8 > gpurun_out/r4/exp2_trace1.log 2>&1
