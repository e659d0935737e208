"""One-rank RCCL smoke on the GPU box: the exact torch.distributed calls bench.py makes when world_size > 1."""
import os, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29517')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=dev)
dist.barrier(); torch.cuda.synchronize()
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print('nccl ok', float(t[0]), dist.get_world_size())
dist.destroy_process_group()
