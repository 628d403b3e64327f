# small proofs: where the 1.0 ms goes (EdDSA circuit, 2^14 config-4 shape)
mkdir -p gpurun_out/r4
for w in eddsa 14; do
  timeout 300 python tools/proof_run.py $w 40 2>&1 | grep "ms per proof" | sed 's/.*median/median/' | cut -c1-200 > gpurun_out/r4/exp28_$w.txt
  ZA_AMD_TRACE=1 timeout 300 python tools/proof_run.py $w 12 2> gpurun_out/r4/exp28_${w}_trace.txt >/dev/null
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4/kt28_$w -o kt -- python tools/proof_run.py $w 12 > /dev/null 2>&1
  python tools/timeline2.py gpurun_out/r4/kt28_$w/kt_kernel_trace.csv 1.6 > gpurun_out/r4/exp28_${w}_timeline.txt
  cat gpurun_out/r4/exp28_$w.txt
done
