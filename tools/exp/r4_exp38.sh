mkdir -p gpurun_out/r4
out=gpurun_out/r4/exp38.txt
for rep in 1 2 3 4; do
  unset ZA_AMD_WPRIO ZA_AMD_HPRIO
  echo -n "witness streams high priority: " >> $out
  timeout 300 python tools/exp/r4_exp33_child.py 22 2>&1 | grep median >> $out
  echo -n "all streams equal priority:    " >> $out
  ZA_AMD_WPRIO=0 ZA_AMD_HPRIO=0 timeout 300 python tools/exp/r4_exp33_child.py 22 2>&1 | grep median >> $out
done
cat $out
