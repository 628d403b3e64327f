# the stream assignment found for small proofs (h,l,a,b_g1,b_g2 -> side streams 1,3,4,2,0) against the old one at larger sizes
mkdir -p gpurun_out/r4
out=gpurun_out/r4/exp32.txt; rm -f $out
for rep in 1 2; do
for js in 0,1,2,3,4 1,3,4,2,0; do
  for w in 22 18 20 dense:20 poseidon:16; do
    echo -n "JSLOTS=$js $w: " >> $out
    ZA_AMD_JSLOTS=$js timeout 300 python tools/proof_run.py $w 24 2>&1 | grep -o "median [0-9.]*" >> $out
  done
done
done
cat $out
