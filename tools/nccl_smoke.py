import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
a=torch.arange(1024, dtype=torch.float32, device="cuda"); g=torch.empty(1024, dtype=torch.float32, device="cuda")
dist.all_gather_into_tensor(g, a); dist.barrier(); torch.cuda.synchronize()
t=torch.tensor([1.5],dtype=torch.float64,device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("nccl ok", bool(torch.equal(g,a)), t.item())
dist.destroy_process_group()
