# round 4: final artefacts
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/final_artifacts.sh r04_m > gpurun_out/r04_final_artifacts.log 2>&1
bash tools/msm_pmc.sh r04_m 26 >> gpurun_out/r04_final_artifacts.log 2>&1
bash tools/bench_workloads.sh r04_w >> gpurun_out/r04_final_artifacts.log 2>&1
python tools/probe.py witness-jobs > gpurun_out/r04_m/witness_jobs_alone.txt 2>&1
python tools/proof_run.py eddsa 400 > gpurun_out/r04_m/proof_run_eddsa.txt 2>&1
python tools/proof_run.py 14 400 > gpurun_out/r04_m/proof_run_2p14.txt 2>&1
python tools/shard_model_run.py 8 > gpurun_out/r04_m/shard_model_run.txt 2>&1
