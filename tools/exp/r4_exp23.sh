# the witness part of the combination beside the h multiexp (ZG_OPT_COMBINE_EARLY = 11)
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_prove.py tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/r4/exp23_tests.txt 2>&1
tail -3 gpurun_out/r4/exp23_tests.txt
uptime > gpurun_out/r4/exp23_ab.txt
for lg in 22 14 18; do
  echo "== 2^$lg" >> gpurun_out/r4/exp23_ab.txt
  timeout 600 python tools/ab_opt.py 11 0,1 $lg 8 8 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4/exp23_ab.txt
done
cat gpurun_out/r4/exp23_ab.txt
