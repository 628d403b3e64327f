# the main stream (R1CS evaluation + transforms: the critical path until the h multiexp starts) at high priority?
mkdir -p gpurun_out/r4
out=gpurun_out/r4/exp51.txt; uptime > $out
for rep in 1 2; do
for mp in 0 -1; do
  echo "== ZA_AMD_MAIN_PRIO=$mp" >> $out
  ZA_AMD_MAIN_PRIO=$mp timeout 300 python tools/exp/r4_exp31_child.py 2>&1 | grep median >> $out
  ZA_AMD_MAIN_PRIO=$mp timeout 300 python tools/exp/r4_exp33_child.py 16 18 20 22 2>&1 | grep median >> $out
done
done
echo -n "three provers 2^22, main stream high: " >> $out
ZA_AMD_MAIN_PRIO=-1 timeout 300 python tools/exp/r4_exp35.py 22 2>&1 | grep "per proof" | awk '{print $(NF-3)}' | tr '\n' ' ' >> $out; echo >> $out
cat $out
