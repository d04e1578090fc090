#!/bin/bash
# bench.py's N > 1 path rehearsed on the ONE GPU of a gpurun box: two ranks share device 0 (RCCL refuses that, so the run falls back — all ranks
# together — to torch's nccl reduce and then to gloo, and says so); what is checked is the new bookkeeping: frame_equals_1gpu, multi_gpu.per_rank, no hang.
O=gpurun_out/r04_multi; mkdir -p $O
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_n2.json 2> $O/bench_n2.err
echo "rc=$?"; tail -4 $O/bench_n2.err
python - <<PY
import json
b = json.loads([l for l in open("$O/bench_n2.json") if l.startswith("{")][-1])
print({k: b[k] for k in ("n_gpus", "ms_per_step", "value", "frame_equals_1gpu")})
print(b["config"]["parallelism"]); print(json.dumps(b["multi_gpu"], indent=1)[:900])
PY
