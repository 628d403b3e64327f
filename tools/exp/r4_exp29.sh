# the latency chains of a small proof, each multiexp alone; then the h multiexp alone under the kernel trace
mkdir -p gpurun_out/r4
timeout 300 python tools/probe.py small-chains eddsa 2>&1 | grep -v amdgpu.ids > gpurun_out/r4/exp29_eddsa.txt
timeout 300 python tools/probe.py small-chains 14 2>&1 | grep -v amdgpu.ids > gpurun_out/r4/exp29_14.txt
cat gpurun_out/r4/exp29_eddsa.txt gpurun_out/r4/exp29_14.txt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4/kt29 -o kt -- python tools/probe.py small-chains eddsa > /dev/null 2>&1
