#!/bin/bash
# two ranks sharing the one GPU of the box: a rehearsal of the N > 1 code path (rendezvous, barriers, MAX, the per-rank report), not a scaling number
cd $GRAFT_REPO_ROOT
DOOMGPU_BENCH_DEVICE=0 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/r05/n2.json 2> gpurun_out/r05/n2.err
echo "rc $?"; tail -3 gpurun_out/r05/n2.err | cut -c1-200
python3 -c "
import json
d=json.loads(open('gpurun_out/r05/n2.json').read().strip().splitlines()[-1])
print('n_gpus', d['n_gpus'], 'value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'side', d.get('side_legs'), 'per_rank', [(r['rank'], round(r['frames_per_s'])) for r in d['per_rank']])"
