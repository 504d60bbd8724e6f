import time, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
x = torch.zeros(1, device='cuda')
for _ in range(3):
    dist.barrier(); torch.cuda.synchronize()
ts = []
for _ in range(20):
    t0 = time.perf_counter(); dist.barrier(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print('barrier+sync ms: median', sorted(ts)[10]*1e3, 'min', min(ts)*1e3)
ts = []
for _ in range(20):
    t0 = time.perf_counter(); dist.all_reduce(x); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print('all_reduce(1)+sync ms: median', sorted(ts)[10]*1e3, 'min', min(ts)*1e3)
dist.destroy_process_group()
