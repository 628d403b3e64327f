# end-of-round refresh: GPU tests, default bench line, bench under rocprofv3 (kernel stats), the small workloads' lines
out=gpurun_out/r4h; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $out/pytest_gpu.txt
timeout 900 python bench.py 2>$out/bench_stderr.txt | tail -1 > $out/bench_default.json
B=$GRAFT_REPO_ROOT/bench.py
cd /tmp && export TMPDIR=/tmp
PROF="--no-cpu-baseline --no-msm --no-dense --no-sha --provers 1 --shard-model 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -- python3 $B $PROF > $GRAFT_REPO_ROOT/$out/bench_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/timeline2.py $(find $out/stats -name "*kernel_trace.csv" | head -1) 60 > $out/timeline_last_proofs.txt
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/stats
SKIP_BIG=1 bash tools/bench_workloads.sh r4hw 2>&1 | tail -6 > $out/workloads.txt
for w in eddsa 14; do echo -n "$w " >> $out/workloads.txt; python tools/proof_run.py $w 100 2>&1 | grep -o "median [0-9.]*" >> $out/workloads.txt; done
cat $out/pytest_gpu.txt $out/workloads.txt; head -c 300 $out/bench_default.json
