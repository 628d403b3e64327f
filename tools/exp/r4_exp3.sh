# round 4, experiment 3: four-lane mixed addition in small G1 accumulations (ZA_AMD_QUAD_ACC)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_msm.py tests/test_gpu_golden_1k.py tests/test_gpu_curve.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4/exp3_tests.txt
python tools/ab_env.py ZA_AMD_QUAD_ACC 0,262144 14 10 20 > gpurun_out/r4/exp3_ab14.txt 2>&1
python tools/ab_env.py ZA_AMD_QUAD_ACC 0,262144 16 10 20 > gpurun_out/r4/exp3_ab16.txt 2>&1
python tools/ab_env.py ZA_AMD_QUAD_ACC 0,262144 18 10 20 > gpurun_out/r4/exp3_ab18.txt 2>&1
ZA_AMD_QUAD_ACC=0 python tools/proof_run.py eddsa 400 > gpurun_out/r4/exp3_eddsa0.txt 2>&1
python tools/proof_run.py eddsa 400 > gpurun_out/r4/exp3_eddsa1.txt 2>&1
ZA_AMD_QUAD_ACC=0 python tools/proof_run.py eddsa 400 > gpurun_out/r4/exp3_eddsa0b.txt 2>&1
python tools/proof_run.py eddsa 400 > gpurun_out/r4/exp3_eddsa1b.txt 2>&1
