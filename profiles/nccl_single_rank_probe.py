import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29511")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
t=torch.ones(1,device="cuda:0"); dist.all_reduce(t); torch.cuda.synchronize()
x=torch.tensor([3.5],dtype=torch.float64,device="cuda:0"); dist.all_reduce(x,op=dist.ReduceOp.MAX)
parts=[torch.zeros_like(x)]; dist.all_gather(parts,x); dist.barrier(); print("nccl single rank ok", float(t.item()), float(x.item()), float(parts[0].item())); dist.destroy_process_group()
