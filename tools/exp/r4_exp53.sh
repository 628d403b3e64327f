# small-job plan with the short last digit penalised: 2^16-sized jobs
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_prove.py -x -q -m gpu > gpurun_out/r4/exp53_tests.txt 2>&1; tail -2 gpurun_out/r4/exp53_tests.txt
out=gpurun_out/r4/exp53.txt; uptime > $out
python tools/probe.py small-chains 16 2>&1 | grep -v amdgpu >> $out
for w in 16 poseidon:16 15 17 eddsa 14; do echo -n "$w: " >> $out; python tools/proof_run.py $w 40 2>&1 | grep -o "median [0-9.]*" >> $out; done
cat $out
