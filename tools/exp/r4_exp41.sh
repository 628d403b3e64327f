mkdir -p gpurun_out/r4
out=gpurun_out/r4/exp41.txt
for rep in 1 2 3; do
for js in default 0,1,2,3,4 3,2,4,1,0; do
  echo -n "JSLOTS=$js: " >> $out
  if [ $js = default ]; then timeout 300 python tools/shard_model_run.py 8 2>&1 | tail -1 | cut -c1-118 >> $out
  else ZA_AMD_JSLOTS=$js timeout 300 python tools/shard_model_run.py 8 2>&1 | tail -1 | cut -c1-118 >> $out; fi
done
done
cat $out
