cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
ZA_AMD_TRACE=1 python tools/proof_run.py 22 6 2>&1 | grep -E "pack workers|pack scan" | tail -4 > gpurun_out/r4/exp20_pin.txt
timeout 900 python tools/ab_opt.py 5 0,1 22 8 6 >> gpurun_out/r4/exp20_pin.txt 2>&1
