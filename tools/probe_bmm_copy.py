import torch
from torch.profiler import profile, ProfilerActivity
E, M, n_in, n_out = 8, 66752, 256, 256
y2 = torch.randn(M, E * n_in, device='cuda')
x3 = y2.view(M, E, n_in).transpose(0, 1)
W = torch.randn(E, n_in, n_out, device='cuda')
b = torch.randn(E, 1, n_out, device='cuda')
g = torch.randn(E, M, n_out, device='cuda')
def run(name, fn):
    fn(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn(); torch.cuda.synchronize()
    ks = [(e.key[:60], e.count, round(e.self_device_time_total)) for e in prof.key_averages() if e.self_device_time_total > 5]
    print(name, ks)
run('baddbmm strided x', lambda: torch.baddbmm(b, x3, W))
run('bmm strided x', lambda: torch.bmm(x3, W))
run('baddbmm contiguous x', lambda: torch.baddbmm(b, x3.contiguous(), W))
run('elu strided', lambda: torch.nn.functional.elu(x3))
e = torch.nn.functional.elu(x3); print('elu out strides', e.stride(), x3.stride())
run('dW = x^T g strided x', lambda: torch.bmm(x3.transpose(1, 2), g))
gs = torch.randn(M, E * n_out, device='cuda').view(M, E, n_out).transpose(0, 1)
run('dx = g W^T, g strided', lambda: torch.bmm(gs, W.transpose(1, 2)))
run('g strided -> [M, E*out] view + mm', lambda: torch.mm(gs.transpose(0, 1).reshape(-1, E * n_out), torch.randn(E * n_out, 384, device='cuda')))
run('g contiguous [E,M,out] -> reshape(M, E*out)', lambda: g.transpose(0, 1).reshape(-1, E * n_out))
