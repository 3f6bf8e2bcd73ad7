import os, time, torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY","0")
for k,v in (("RANK","0"),("WORLD_SIZE","1"),("MASTER_ADDR","127.0.0.1"),("MASTER_PORT","29519")): os.environ.setdefault(k,v)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
t=torch.zeros(167444, device="cuda")
for _ in range(20): dist.all_reduce(t)
torch.cuda.synchronize()
for n,asyn in ((2000,False),(2000,True)):
    torch.cuda.synchronize(); s=time.perf_counter()
    for _ in range(n):
        w=dist.all_reduce(t, async_op=asyn)
        if asyn: w.wait()
    h=time.perf_counter()-s
    torch.cuda.synchronize(); e=time.perf_counter()-s
    print("async" if asyn else "sync", "host per call %.1f us, total per call %.1f us" % (h/n*1e6, e/n*1e6))
# slice views like the exchange uses
v=t[4+364:]
torch.cuda.synchronize(); s=time.perf_counter()
for _ in range(2000):
    w=dist.all_reduce(v, async_op=True); w.wait()
h=time.perf_counter()-s; torch.cuda.synchronize()
print("slice async host per call %.1f us" % (h/2000*1e6))
dist.destroy_process_group()
