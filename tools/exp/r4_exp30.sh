# do the small chains serialize because their hardware queues share a pipe?  (b_g2 + h: 0.92 ms together, 0.66 / 0.30 alone)
mkdir -p gpurun_out/r4
for ws in "1,2,3,4" "1,2,4,3" "1,4,3,2" "4,2,3,1"; do
  echo "== ZA_AMD_WSLOTS=$ws (streams of l, a, b_g1, b_g2)" >> gpurun_out/r4/exp30.txt
  ZA_AMD_WSLOTS=$ws python tools/probe.py small-chains eddsa 2>&1 | grep -E "^(h |b_g2|all five|zg_prove_dev|l \+)" >> gpurun_out/r4/exp30.txt
done
echo "== priorities equal (WPRIO=HPRIO=0)" >> gpurun_out/r4/exp30.txt
ZA_AMD_WPRIO=0 ZA_AMD_HPRIO=0 python tools/probe.py small-chains eddsa 2>&1 | grep -E "^(h |b_g2|all five|zg_prove_dev|l \+)" >> gpurun_out/r4/exp30.txt
cat gpurun_out/r4/exp30.txt
