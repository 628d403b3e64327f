mkdir -p gpurun_out/r4
out=gpurun_out/r4/exp40.txt
for rep in 1 2; do
for cfg in "3,2,4,1,0 0 0" "3,2,4,1,0 -1 0" "0,1,2,3,4 -1 0" "0,1,2,3,4 0 0"; do
  set -- $cfg
  echo -n "JSLOTS=$1 WPRIO=$2 HPRIO=$3: " >> $out
  ZA_AMD_JSLOTS=$1 ZA_AMD_WPRIO=$2 ZA_AMD_HPRIO=$3 timeout 300 python tools/shard_model_run.py 8 2>&1 | tail -1 | cut -c1-120 >> $out
done
done
cat $out
