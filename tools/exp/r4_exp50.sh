# more hardware queues than the 8 the library asks for?
mkdir -p gpurun_out/r4
out=gpurun_out/r4/exp50.txt; uptime > $out
for q in 8 12 16 24; do
  echo "== GPU_MAX_HW_QUEUES=$q" >> $out
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/exp/r4_exp31_child.py 2>&1 | grep median >> $out
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/exp/r4_exp33_child.py 18 22 2>&1 | grep median >> $out
  echo -n "three provers 2^22: " >> $out
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/exp/r4_exp35.py 22 2>&1 | grep "per proof" | awk '{print $(NF-3)}' | tr '\n' ' ' >> $out; echo >> $out
done
cat $out
