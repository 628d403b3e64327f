# round 4, experiment 12: compact table test; upload chunk ratios; packing threads / pinning
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_gpu_prove.py -m gpu -x -q -k "compact or subset" 2>&1 | tail -25 > gpurun_out/r4/exp12_tests.txt
ZA_AMD_SORT_STAGED=3 timeout 1200 python -m pytest tests/test_gpu_prove.py -m gpu -x -q -k "compact or subset" 2>&1 | tail -15 > gpurun_out/r4/exp12_tests_forced.txt
for r in 1,2,4,8 1,2,3,4 2,3,4,5 1,2,4,5 1,3,5,6 1,1,1,1 2,3,3,3; do
  echo "== $r" >> gpurun_out/r4/exp12_chunks.txt
  ZA_AMD_MSM_CHUNKS=$r timeout 600 python tools/msm_trace_run.py 26 host 2>&1 | grep msm >> gpurun_out/r4/exp12_chunks.txt
done
timeout 900 python tools/ab_opt.py 5 0,1 22 6 6 > gpurun_out/r4/exp12_ab_pin.txt 2>&1
timeout 900 python tools/ab_opt.py 3 4,8,16,32 22 6 6 > gpurun_out/r4/exp12_ab_threads.txt 2>&1
uptime > gpurun_out/r4/exp12_uptime.txt; nproc >> gpurun_out/r4/exp12_uptime.txt; numactl -H >> gpurun_out/r4/exp12_uptime.txt 2>&1
