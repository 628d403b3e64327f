# Round artefacts: GPU tests, default bench line, rocprofv3 kernel stats, timeline, PMC traffic passes.
# usage (on the GPU box, from the repo root): bash tools/final_artifacts.sh <tag>
tag=${1:-r01_e}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $out/pytest_gpu.txt
timeout 900 python bench.py 2>$out/bench_stderr.txt | tail -1 > $out/bench_default.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-msm --provers 1 > $out/bench_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-msm --provers 1 --steps 2 --warmup 1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-msm --provers 1 --steps 2 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py $(find $out/stats -name "*kernel_trace.csv" | head -1) > $out/timeline_last_proof.txt
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python tools/pmc_traffic.py $(find $out/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $out/pmc_write -name "*counter_collection.csv" | head -1) $out/pmc_traffic.json > $out/pmc_summary.txt 2>&1
rm -rf $out/stats $out/pmc_fetch $out/pmc_write
ls -la $out
