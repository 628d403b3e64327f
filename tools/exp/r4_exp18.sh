cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python tools/ab_env.py ZA_AMD_PACK_SIMD 0,1 22 8 6 > gpurun_out/r4/exp18_ab_simd.txt 2>&1
ZA_AMD_TRACE=1 ZA_AMD_PACK_SIMD=0 python tools/proof_run.py 22 8 2>&1 | grep "pack " > gpurun_out/r4/exp18_trace0.txt
ZA_AMD_TRACE=1 python tools/proof_run.py 22 8 2>&1 | grep "pack " > gpurun_out/r4/exp18_trace1.txt
timeout 600 python -m pytest tests/test_gpu_prove.py -m gpu -x -q -k "pack or hand" 2>&1 | tail -3 > gpurun_out/r4/exp18_tests.txt
uptime > gpurun_out/r4/exp18_uptime.txt
