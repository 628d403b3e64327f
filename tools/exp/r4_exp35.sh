mkdir -p gpurun_out/r4
for js in 3,2,4,1,0; do
  echo "== ZA_AMD_JSLOTS=$js" >> gpurun_out/r4/exp35.txt
  ZA_AMD_JSLOTS=$js timeout 600 python tools/exp/r4_exp35.py 22 2>&1 | grep "per proof" >> gpurun_out/r4/exp35.txt
done
cat gpurun_out/r4/exp35.txt
echo "== lone context, option 10 (ab_opt)" >> gpurun_out/r4/exp35.txt
timeout 600 python tools/ab_opt.py 10 0,400 22 8 8 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4/exp35.txt
tail -3 gpurun_out/r4/exp35.txt
