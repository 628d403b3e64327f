cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/kt17_u24 -- python3 bench.py --no-msm --no-dense --no-sha --shard-model 0 --no-cpu-baseline --log-m 24 --bit-fraction 0 --seed 0x26 --synthetic-key --provers 1 --steps 4 --warmup 2 > gpurun_out/r4/exp17_u24.json 2>gpurun_out/r4/exp17_u24.err
cp $(find gpurun_out/r4/kt17_u24 -name "*kernel_stats.csv" | head -1) gpurun_out/r4/exp17_u24_kernel_stats.csv
python tools/timeline2.py $(find gpurun_out/r4/kt17_u24 -name "*kernel_trace.csv" | head -1) 140 > gpurun_out/r4/exp17_u24_timeline.txt
rm -rf gpurun_out/r4/kt17_u24
