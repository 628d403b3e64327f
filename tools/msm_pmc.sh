# HBM traffic of the kernels of a stand-alone 2^n G1 multiexp (resident bases, window table): two rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) -> gpurun_out/<tag>/pmc_traffic_msm_2p<n>.json
# usage (GPU box): bash tools/msm_pmc.sh <tag> [log_n]
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
tag=${1:-r0x}
lg=${2:-26}
out="$ROOT/gpurun_out/$tag"
mkdir -p "$out"
R="$ROOT/tools/msm_trace_run.py"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/msm_stats" -- python3 "$R" "$lg" > "$out/msm_2p${lg}_under_rocprof.txt" 2>/dev/null
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/msm_pmc_fetch" -- python3 "$R" "$lg" > /dev/null 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/msm_pmc_write" -- python3 "$R" "$lg" > /dev/null 2>&1
cd "$ROOT"
stats=$(find "$out/msm_stats" -name "*kernel_stats.csv" | head -1)
fetch=$(find "$out/msm_pmc_fetch" -name "*counter_collection.csv" | head -1)
write=$(find "$out/msm_pmc_write" -name "*counter_collection.csv" | head -1)
[ -n "$stats" ] && [ -n "$fetch" ] && [ -n "$write" ] || { echo "msm_pmc.sh: a rocprofv3 pass left no CSV under $out" >&2; exit 1; }
cp "$stats" "$out/kernel_stats_msm_2p${lg}.csv"
python3 tools/pmc_traffic.py "$fetch" "$write" "$out/pmc_traffic_msm_2p${lg}.json" 1.0 > "$out/pmc_msm_summary.txt" 2>&1
rm -rf "$out/msm_stats" "$out/msm_pmc_fetch" "$out/msm_pmc_write"
