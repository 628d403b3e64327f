# clean kernel timeline of single-context zg_prove proofs at 2^22 (memory copies included) after the hand-over / combination changes
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r4/kt24 -o kt -- python tools/proof_run.py 22 8 > gpurun_out/r4/exp24.log 2>&1
f=$(find gpurun_out/r4/kt24 -name '*kernel_trace.csv' | head -1)
python tools/timeline2.py $f 24 > gpurun_out/r4/exp24_timeline.txt
m=$(find gpurun_out/r4/kt24 -name '*memory_copy_trace.csv' | head -1)
cp $m gpurun_out/r4/exp24_memcpy.csv
find gpurun_out/r4/kt24 -name '*.csv' -size +3M -delete
tail -2 gpurun_out/r4/exp24.log
