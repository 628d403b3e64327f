cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
tools/ubench_shoup > gpurun_out/r4/exp14_ubench_shoup.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_prove.py -m gpu -x -q -k "back_to_back or hands_over" 2>&1 | tail -5 > gpurun_out/r4/exp14_tests.txt
