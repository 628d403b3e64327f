# k_bits_level on lane pairs (G2) without its scratch frame: one call site of the bucket fetch
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prove.py tests/test_gpu_golden_1k.py -x -q -m gpu > gpurun_out/r4/exp47_tests.txt 2>&1; tail -2 gpurun_out/r4/exp47_tests.txt
out=gpurun_out/r4/exp47.txt; uptime > $out
python tools/probe.py small-chains eddsa 2>&1 | grep -E "^(b_g2 |h |all five|zg_prove_dev)" >> $out
python tools/probe.py small-chains 14 2>&1 | grep -E "^(b_g2 |h |all five|zg_prove_dev)" >> $out
for w in eddsa 14 16 dense:20 22; do echo -n "$w: " >> $out; python tools/proof_run.py $w 40 2>&1 | grep -o "median [0-9.]*" >> $out; done
cat $out
