# hand-over consumed from both ends: copy engine from the front, packing threads from the back (cap = option 10)
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_prove.py -x -q -m gpu > gpurun_out/r4/exp25_tests.txt 2>&1
tail -3 gpurun_out/r4/exp25_tests.txt
uptime > gpurun_out/r4/exp25_ab.txt
timeout 600 python tools/ab_opt.py 10 0,200,400,600 22 8 8 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4/exp25_ab.txt
echo "== pack threads 4 (a slow packing pass)" >> gpurun_out/r4/exp25_ab.txt
ZA_AB_PRESET="3=4" timeout 600 python tools/ab_opt.py 10 0,200,400,600 22 6 8 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4/exp25_ab.txt
cat gpurun_out/r4/exp25_ab.txt
ZA_AMD_TRACE=1 timeout 300 python tools/proof_run.py 22 8 2>&1 | grep -E "pack |handed|upload_bytes" | tail -8
