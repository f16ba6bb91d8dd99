import torch, time
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for mb in (1,4,9,18,36,72,144,288,1024):
    n=mb*1024*1024//4
    a=torch.randn(n,device='cuda'); b=torch.empty_like(a)
    us=t(lambda: torch.add(a,1.0,out=b))
    us2=t(lambda: a.sum())
    print(f"{mb:5d} MB  add: {us:7.1f} us  {2*mb/1024/us*1e6/1024:6.2f} TB/s   sum: {us2:7.1f} us {mb/1024/us2*1e6/1024:6.2f} TB/s")
