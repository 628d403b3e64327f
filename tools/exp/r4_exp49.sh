mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_prove.py tests/test_gpu_golden_1k.py -x -q -m gpu > gpurun_out/r4/exp49_tests.txt 2>&1; tail -2 gpurun_out/r4/exp49_tests.txt
out=gpurun_out/r4/exp49.txt; uptime > $out
python tools/probe.py small-chains eddsa 2>&1 | grep -E "^(R1CS|h |all five|zg_prove_dev)" >> $out
python tools/probe.py small-chains 14 2>&1 | grep -E "^(R1CS|h |all five|zg_prove_dev)" >> $out
for w in eddsa 14 16 18; do echo -n "$w: " >> $out; python tools/proof_run.py $w 60 2>&1 | grep -o "median [0-9.]*" >> $out; done
cat $out
