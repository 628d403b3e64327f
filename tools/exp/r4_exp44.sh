mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_prove.py tests/test_gpu_golden_1k.py tests/test_gpu_outer.py -x -q -m gpu > gpurun_out/r4/exp44_tests.txt 2>&1; tail -2 gpurun_out/r4/exp44_tests.txt
out=gpurun_out/r4/exp44.txt; uptime > $out
for lg in 14 16 22; do echo "== 2^$lg" >> $out; timeout 600 python tools/ab_opt.py 11 0,1 $lg 8 8 2>&1 | grep -v amdgpu.ids >> $out; done
timeout 300 python tools/proof_run.py eddsa 60 2>&1 | grep -o "median [0-9.]*" >> $out
timeout 300 python tools/proof_run.py 14 60 2>&1 | grep -o "median [0-9.]*" >> $out
cat $out
