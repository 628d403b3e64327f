cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for v in 1 0 1 0; do echo "== ZA_AMD_SORT_STAGED=$v" >> gpurun_out/r4/exp21_shard.txt; ZA_AMD_SORT_STAGED=$v python tools/shard_model_run.py 8 2>&1 | tail -2 >> gpurun_out/r4/exp21_shard.txt; done
timeout 600 python tools/ab_env.py ZA_AMD_SORT_STAGED 0,1 19 8 10 0.0 > gpurun_out/r4/exp21_ab19_dense.txt 2>&1
timeout 600 python tools/ab_env.py ZA_AMD_SORT_STAGED 0,1 20 8 10 > gpurun_out/r4/exp21_ab20.txt 2>&1
