mkdir -p gpurun_out/r4
out=gpurun_out/r4/exp45.txt; uptime > $out
for rep in 1 2 3; do
for pq in 0 1; do
  echo -n "ZA_AMD_PAR_QUEUE=$pq " >> $out
  ZA_AMD_PAR_QUEUE=$pq timeout 300 python tools/exp/r4_exp31_child.py 2>&1 | grep median >> $out
done
done
ZA_AMD_PAR_QUEUE=1 ZA_AMD_TRACE=1 python tools/proof_run.py 14 30 2>&1 | sed -n "/traced proof/,\$p" | grep -v amdgpu | cut -c1-110 | head -30 >> $out
cat $out
