mkdir -p gpurun_out/r4
python - <<'PY' >> gpurun_out/r4/exp36b.txt
import torch
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,"priority_range") else "?")
PY
for rep in 1 2; do
for pr in "x x" "0 0" "-1 -1"; do
  set -- $pr
  echo "== ZA_AMD_JSLOTS=3,2,4,1,0 WPRIO=$1 HPRIO=$2" >> gpurun_out/r4/exp36b.txt
  if [ "$1" = "x" ]; then
  ZA_AMD_JSLOTS=3,2,4,1,0 timeout 600 python tools/exp/r4_exp35.py 22 2>&1 | grep "per proof" | awk '{print $(NF-3)}' | tr '\n' ' ' >> gpurun_out/r4/exp36b.txt
  else
  ZA_AMD_WPRIO=$1 ZA_AMD_HPRIO=$2 ZA_AMD_JSLOTS=3,2,4,1,0 timeout 600 python tools/exp/r4_exp35.py 22 2>&1 | grep "per proof" | awk '{print $(NF-3)}' | tr '\n' ' ' >> gpurun_out/r4/exp36b.txt
  fi
  echo >> gpurun_out/r4/exp36b.txt
done
done
cat gpurun_out/r4/exp36b.txt
