mkdir -p gpurun_out/r4
out=gpurun_out/r4/exp37.txt
for rep in 1 2; do
for cfg in "3,2,4,1,0 x x" "3,2,4,1,0 0 0" "0,1,2,3,4 x x"; do
  set -- $cfg
  if [ "$2" = "x" ]; then export -n ZA_AMD_WPRIO ZA_AMD_HPRIO; unset ZA_AMD_WPRIO ZA_AMD_HPRIO; else export ZA_AMD_WPRIO=$2 ZA_AMD_HPRIO=$3; fi
  echo -n "prio $2/$3: " >> $out
  ZA_AMD_JSLOTS=$1 timeout 300 python tools/exp/r4_exp31_child.py 2>&1 | grep median >> $out
  echo -n "prio $2/$3: " >> $out
  ZA_AMD_JSLOTS=$1 timeout 600 python tools/exp/r4_exp33_child.py 16 18 20 22 2>&1 | grep median >> $out
done
done
unset ZA_AMD_WPRIO ZA_AMD_HPRIO
echo "dense 2^20:" >> $out
for cfg in "3,2,4,1,0 x x" "3,2,4,1,0 0 0" "0,1,2,3,4 x x"; do
  set -- $cfg
  if [ "$2" = "x" ]; then unset ZA_AMD_WPRIO ZA_AMD_HPRIO; else export ZA_AMD_WPRIO=$2 ZA_AMD_HPRIO=$3; fi
  echo -n "$1 prio $2/$3: " >> $out
  ZA_AMD_JSLOTS=$1 timeout 300 python tools/proof_run.py dense:20 16 2>&1 | grep -o "median [0-9.]*" >> $out
done
cat $out
