# round 4, experiment 13: h parts and the range-gated start, re-measured with the staged partition
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python tools/ab_env.py ZA_AMD_H_PARTS 1,2,3 22 6 6 > gpurun_out/r4/exp13_ab_parts.txt 2>&1
timeout 900 python tools/ab_env.py ZA_AMD_H_PIPE 0,1 22 6 6 > gpurun_out/r4/exp13_ab_pipe.txt 2>&1
for p in 2 3 4; do echo "== parts $p" >> gpurun_out/r4/exp13_msm26_parts.txt; ZA_AMD_H_PARTS=$p timeout 600 python tools/msm_trace_run.py 26 2>&1 | grep msm >> gpurun_out/r4/exp13_msm26_parts.txt; done
timeout 900 python tools/proof_run.py eddsa 400 > gpurun_out/r4/exp13_eddsa.txt 2>&1
timeout 900 python tools/proof_run.py 14 400 > gpurun_out/r4/exp13_p14.txt 2>&1
timeout 900 python tools/shard_model_run.py 8 > gpurun_out/r4/exp13_shard.txt 2>&1
