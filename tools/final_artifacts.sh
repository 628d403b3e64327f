# Round artefacts: GPU tests, default bench line, rocprofv3 kernel stats, timeline, PMC passes (separate runs).
# usage (on the GPU box, from the repo root): bash tools/final_artifacts.sh <tag>      -> gpurun_out/<tag>/
# (SKIP_TESTS=1: without the GPU test suite)
tag=${1:-r02_x}
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$ROOT}
cd "$GRAFT_REPO_ROOT"
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p "$out"
B=$GRAFT_REPO_ROOT/bench.py
if [ -z "${SKIP_TESTS:-}" ]; then timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $out/pytest_gpu.txt; fi
timeout 900 python bench.py 2>$out/bench_stderr.txt | tail -1 > $out/bench_default.json
cd /tmp && export TMPDIR=/tmp
PROF="--no-cpu-baseline --no-msm --no-dense --no-sha --provers 1 --shard-model 0 --no-outer"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $B $PROF > $out/bench_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $B $PROF --steps 2 --warmup 1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $B $PROF --steps 2 --warmup 1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_sq -- python3 $B $PROF --steps 2 --warmup 1 > /dev/null 2>&1
# where the accumulation's launch time goes: cycles the SQs are busy vs the whole GPU, waves resident, VALU-active cycles
timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_busy -- python3 $B $PROF --steps 2 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline2.py $(find $out/stats -name "*kernel_trace.csv" | head -1) 60 > $out/timeline_last_proofs.txt
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python tools/pmc_traffic.py $(find $out/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $out/pmc_write -name "*counter_collection.csv" | head -1) $out/pmc_traffic.json 1.0 > $out/pmc_summary.txt 2>&1
sq=$(find $out/pmc_sq -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py --by-grid $sq $out/pmc_sq_counters.json k_accumulate k_ntt >> $out/pmc_summary.txt 2>&1
python tools/pmc_summary.py --by-grid $(find $out/pmc_busy -name "*counter_collection.csv" | head -1) $out/pmc_busy_counters.json k_accumulate k_ntt >> $out/pmc_summary.txt 2>&1
# the h multiexp of the 2^22 proof (evaluation form: 2^22 points) runs in two parts of 2^21 points x 13 digits
python tools/pmc_valu.py $sq $out/pmc_valu_instructions.json 27262976 >> $out/pmc_summary.txt 2>&1
rm -rf $out/stats $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/pmc_busy
ls -la $out
