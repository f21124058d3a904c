"""One-rank RCCL sanity on a one-GPU box: the process group comes up and an all-reduce of a flat fp32 buffer returns."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.arange(330000, dtype=torch.float32, device="cuda")
dist.all_reduce(x)
dist.barrier()
torch.cuda.synchronize()
print("rccl ok", float(x[-1]))
dist.destroy_process_group()
