import os, sys, time, socket
import torch, torch.distributed as dist, torch.multiprocessing as mp

def w(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    print(rank, "start", flush=True)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    print(rank, "init done", flush=True)
    t = torch.full((1000,), float(rank + 1), device="cuda:0")
    h = dist.all_reduce(t[10:500], async_op=True)
    print(rank, "issued", flush=True)
    h.wait()
    torch.cuda.synchronize()
    print(rank, "sync allreduce on cuda slice ->", float(t[10]), float(t[0]), flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(w, args=(2, port), nprocs=2, join=True)
    print("done", flush=True)
