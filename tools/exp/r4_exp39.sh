mkdir -p gpurun_out/r4
out=gpurun_out/r4/exp39.txt
for rep in 1 2 3; do
  echo -n "three provers, 2^22: " >> $out
  timeout 600 python tools/exp/r4_exp35.py 22 2>&1 | grep "per proof" | awk '{print $(NF-3)}' | tr '\n' ' ' >> $out; echo >> $out
done
echo -n "three provers, 2^16: " >> $out
timeout 600 python tools/exp/r4_exp35.py 16 2>&1 | grep "per proof" | awk '{print $(NF-3)}' | tr '\n' ' ' >> $out; echo >> $out
timeout 300 python tools/exp/r4_exp31_child.py 2>&1 | grep median >> $out
timeout 600 python tools/exp/r4_exp33_child.py 16 18 20 22 2>&1 | grep median >> $out
timeout 300 python tools/shard_model_run.py 8 2>&1 | tail -2 >> $out
cat $out
