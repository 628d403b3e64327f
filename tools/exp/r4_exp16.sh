cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
ZA_AMD_SMALL_LOG_N=0 timeout 700 python tools/fuzz_prove.py 600 2>&1 | tail -6 > gpurun_out/r4/exp16_fuzz_nosmall.txt
timeout 700 python tools/fuzz_prove.py 600 2>&1 | tail -6 > gpurun_out/r4/exp16_fuzz.txt
