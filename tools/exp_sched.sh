timeout 1200 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-msm --provers 2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d['proof_sha'], d['kernel_ms_per_step']['ntt_pass'], d.get('concurrent_provers'))"; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt7 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-msm --provers 1 --steps 5 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
grep "k_ntt_pass" $(find gpurun_out/kt7 -name "*kernel_stats.csv" | head -1) | cut -c1-40,150-260
python tools/timeline.py $(find gpurun_out/kt7 -name "*kernel_trace.csv" | head -1) | grep " q1 " | head -30
