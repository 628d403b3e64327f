# The other workloads' bench lines (one JSON line each): bash tools/bench_workloads.sh <tag>  -> gpurun_out/<tag>/bench_*.json
# (SKIP_BIG=1: without the 2^24 / 2^26 configurations, which take ten minutes of set-up; ONLY=<name>: that line alone)
tag=${1:-r0x_w}
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$ROOT}
cd "$GRAFT_REPO_ROOT"
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p "$out"
uptime > $out/uptime.txt
Q="--no-msm --no-dense --no-sha --shard-model 0 --no-cpu-baseline --no-outer"
run() { name=$1; shift; if [ -n "${ONLY:-}" ] && [ "$ONLY" != "$name" ]; then return; fi; timeout 1500 python bench.py $Q "$@" 2>$out/$name.err | tail -1 > $out/bench_$name.json; }
run eddsa_poseidon --workload eddsa --steps 100 --warmup 10
run poseidon_2p14 --workload poseidon --log-m 14 --steps 100 --warmup 10
run poseidon_2p20 --workload poseidon --log-m 20 --steps 30 --warmup 5
run sha256_merkle_d32 --workload sha256 --depth 32 --steps 30 --warmup 5
run sha256_merkle_d90 --workload sha256 --depth 90 --steps 20 --warmup 5
if [ -z "${SKIP_BIG:-}" ]; then
run uniform_2p24 --log-m 24 --bit-fraction 0 --seed 0x26 --synthetic-key --provers 1 --steps 5 --warmup 2
run uniform_2p26 --log-m 26 --bit-fraction 0 --seed 0x26 --synthetic-key --provers 1 --steps 3 --warmup 1
run bits_2p26 --log-m 26 --synthetic-key --provers 1 --steps 3 --warmup 1
fi
for f in $out/bench_*.json; do python -c "import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],3), 'ms', int(d['value']), d['unit'])" $f; done
find $out -name "*.err" -size 0 -delete
