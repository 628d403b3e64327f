# exclusive kernel times of the witness multiexps (each alone, then all four) of the 2^22 config-4 proof
mkdir -p gpurun_out/r4
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4/kt27 -o kt -- python tools/probe.py witness-jobs > gpurun_out/r4/exp27.log 2>&1
grep -E "^(l|a|b_g1|b_g2|all four|h) " gpurun_out/r4/exp27.log
