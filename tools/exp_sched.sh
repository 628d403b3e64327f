timeout 600 python -m pytest tests/test_gpu_prove.py -m gpu -x -q -k "in_parts" 2>&1 | tail -3
for n in 2; do
ZA_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 3 --warmup 1 --dist-backend gloo --log-m 22 2>&1 | tail -1 | cut -c1-400
done
timeout 300 python bench.py --no-cpu-baseline --no-msm --provers 1 --steps 3 --warmup 1 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('single', d['ms_per_step'], d['proof_sha'])"
