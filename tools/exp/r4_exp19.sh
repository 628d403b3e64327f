cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for a in 0 1; do for r in 0 1; do
  echo "== alloc node $a run node $r" >> gpurun_out/r4/exp19_numa.txt
  ZA_AMD_TRACE=1 python tools/probe.py numa $a $r 2>&1 | grep -E "pack scan|nodes|last proof" | tail -6 >> gpurun_out/r4/exp19_numa.txt
done; done
cat /sys/class/drm/card*/device/numa_node > gpurun_out/r4/exp19_gpu_node.txt 2>&1
for i in 1 2 3 4; do ZA_AMD_TRACE=1 python tools/proof_run.py 22 6 2>&1 | grep "pack scan" | tail -2 >> gpurun_out/r4/exp19_repeat.txt; echo "--" >> gpurun_out/r4/exp19_repeat.txt; done
