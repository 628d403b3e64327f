# round 4, experiment 10: subset sums over 8 positions; geometric upload chunks of the resident multiexp
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_golden_1k.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4/exp10_tests.txt
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "2p26_points or config4" 2>&1 | tail -15 > gpurun_out/r4/exp10_tests_full.txt
timeout 600 python tools/fuzz_prove.py 90 2>&1 | tail -3 > gpurun_out/r4/exp10_fuzz.txt
timeout 900 python bench.py --no-cpu-baseline --no-dense --no-sha --provers 1 --shard-model 0 2>gpurun_out/r4/exp10_bench.err | tail -1 > gpurun_out/r4/exp10_bench.json
ZA_AMD_QUAD_W=4 timeout 900 python bench.py --no-cpu-baseline --no-dense --no-sha --no-msm --provers 1 --shard-model 0 2>gpurun_out/r4/exp10_bench4.err | tail -1 > gpurun_out/r4/exp10_bench_q4.json
timeout 900 python bench.py --no-cpu-baseline --no-dense --no-sha --no-msm --provers 1 --shard-model 0 2>gpurun_out/r4/exp10_bench8.err | tail -1 > gpurun_out/r4/exp10_bench_q8.json
ZA_AMD_QUAD_W=4 timeout 900 python bench.py --no-cpu-baseline --no-dense --no-sha --no-msm --provers 1 --shard-model 0 2>gpurun_out/r4/exp10_bench4.err | tail -1 > gpurun_out/r4/exp10_bench_q4b.json
timeout 900 python bench.py --no-cpu-baseline --no-dense --no-sha --no-msm --provers 1 --shard-model 0 2>gpurun_out/r4/exp10_bench8.err | tail -1 > gpurun_out/r4/exp10_bench_q8b.json
