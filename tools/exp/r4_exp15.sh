cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
ZA_AMD_TRACE=1 python tools/proof_run.py eddsa 30 > gpurun_out/r4/exp15_eddsa_trace.txt 2>&1
rocprofv3 --kernel-trace -d gpurun_out/r4/kt15_eddsa -o kt -- python tools/proof_run.py eddsa 30 > gpurun_out/r4/exp15_eddsa_prof.log 2>&1
