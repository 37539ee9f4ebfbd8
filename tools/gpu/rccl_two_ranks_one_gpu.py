#!/usr/bin/env python3
"""The RCCL branch of the band halo exchange, executed: TWO ranks, both on cuda:0 (the round's GPU boxes have one GPU), backend nccl (= RCCL on ROCm).
Each rank decodes its bands of a tall multi-LF-group fixture with jxl_coder_amd.shard.decode_sharded — the rank-border halos travel as device tensors through
dist.batch_isend_irecv — and checks its rows against a whole-frame decode.  Launch:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/gpu/rccl_two_ranks_one_gpu.py
RCCL may refuse two ranks on one device ("Duplicate GPU detected"): the script then says so and exits 3 — that, too, is a result."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import jxl_coder_amd as J
from jxl_coder_amd import shard

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
DEV = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()      # one GPU: both ranks on cuda:0; more: one each
torch.cuda.set_device(DEV)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{DEV}"))
    t = torch.ones(4, device=f"cuda:{DEV}") * (rank + 1)
    dist.all_reduce(t)                                   # the communicator is created here: a duplicate-GPU refusal shows now
    torch.cuda.synchronize()
except Exception as e:  # noqa: BLE001
    print(f"[rank {rank}] RCCL refused two ranks on one GPU: {str(e)[:300]}")
    sys.exit(3)
print(f"[rank {rank}] nccl communicator over one GPU up, all_reduce -> {t[0].item()}")
name = "vb520x4400_e7"          # 17 group rows = 3 LF-group rows
data = open(os.path.join(ROOT, "tests", "golden", name + ".jxl"), "rb").read()
whole, info = J.JxlDecoder(DEV).decode_one_shot(data)
for nbands in (2, 4):
    got = shard.decode_sharded(data, nbands=nbands, rank=rank, world=world, device=DEV)
    torch.cuda.synchronize()
    for (y0, y1, rows) in got:
        a = rows.cpu().numpy().reshape(y1 - y0, whole.shape[1], 4)
        assert np.array_equal(a, whole[y0:y1]), (rank, nbands, y0, y1)
    print(f"[rank {rank}] {nbands} bands over {world} ranks: rows {[(y0, y1) for y0, y1, _ in got]} == whole-frame decode (halos through dist.batch_isend_irecv on device tensors)")
dist.barrier()
dist.destroy_process_group()
