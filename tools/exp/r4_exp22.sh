# hybrid hand-over: head of the witness raw over PCIe while the rest is packed (ZG_OPT_PACK_RAW_PERMILLE = 10)
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_prove.py tests/test_gpu_bench_contract.py -x -q -m gpu > gpurun_out/r4/exp22_tests.txt 2>&1
tail -3 gpurun_out/r4/exp22_tests.txt
uptime > gpurun_out/r4/exp22_ab2.txt
timeout 600 python tools/ab_opt.py 10 0,200,250 22 8 8 >> gpurun_out/r4/exp22_ab2.txt 2>&1
tail -4 gpurun_out/r4/exp22_ab2.txt
timeout 600 python bench.py > gpurun_out/r4/exp22_bench.json 2> gpurun_out/r4/exp22_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/exp22_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d.get('upload_bytes_last_step'), d.get('stages_ms'))
PY
