mkdir -p gpurun_out/r4
uptime > gpurun_out/r4/exp26_ab.txt
timeout 900 python tools/ab_env.py ZA_AMD_PACK_PREFETCH 0,256,512,768,1024,2048 22 8 8 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4/exp26_ab.txt
cat gpurun_out/r4/exp26_ab.txt
