import torch,time
x=torch.empty(66752*2048,device='cuda')
y=torch.randn(66752*2048,device='cuda')
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
us=t(lambda: x.fill_(1.0)); print(f'fill 547 MB: {us:.1f} us = {x.numel()*4/us/1e6:.2f} TB/s write')
us=t(lambda: x.copy_(y)); print(f'copy 547 MB: {us:.1f} us = {2*x.numel()*4/us/1e6:.2f} TB/s read+write')
us=t(lambda: torch.sum(y)); print(f'sum 547 MB: {us:.1f} us = {x.numel()*4/us/1e6:.2f} TB/s read')
