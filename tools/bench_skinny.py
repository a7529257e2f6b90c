import sys, os
ROOT='/root/repo'
sys.path[:0]=[ROOT, os.path.join(ROOT,'recurrent-offpolicy-rl_amd')]
import torch
from offpolicy_rnn.hip import ops
T=66752
def timeit(fn,n=20,warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
g=torch.Generator(device='cuda').manual_seed(0)
r=lambda *s: torch.randn(*s,device='cuda',generator=g)
# encode fwd: K=41 (lib) vs padded K=44 (mine)
x41,W41,b=r(T,41),r(384,41),r(384)
x44=torch.cat([x41,torch.zeros(T,3,device='cuda')],1); W44=torch.cat([W41,torch.zeros(384,3,device='cuda')],1)
print('encode fwd  lib K=41 %.1f us | mine K=44 %.1f us | cat cost %.1f us'%(timeit(lambda: torch.addmm(b,x41,W41.t())), timeit(lambda: ops.gemm_f32(x44,W44,True,True,b)), timeit(lambda: torch.cat([x41,torch.zeros(T,3,device='cuda')],1))))
dy=r(T,384)
print('encode wgrad lib [384,T]x[T,41] %.1f us | mine [384,T]x[T,44] %.1f us'%(timeit(lambda: dy.t()@x41), timeit(lambda: ops.gemm_f32(dy,x44,False,False))))
# dt_proj fwd K=16 N=512 (input is a strided slice of x_dbl [T,80])
xd=r(T,80); Wd=r(512,16)
print('dt fwd  lib %.1f us | mine %.1f us'%(timeit(lambda: xd[:,:16]@Wd.t()), timeit(lambda: ops.gemm_f32(xd[:,:16],Wd,True,True))))
dd=r(T,512)
print('dt dgrad [T,512]x[512,16] lib %.1f us | mine %.1f us'%(timeit(lambda: dd@Wd), timeit(lambda: ops.gemm_f32(dd,Wd,True,False))))
# x_proj dgrad with accumulate: dxc += dx_dbl @ xproj_w  ([T,80]x[80,512])
dx=r(T,80); Wx=r(80,512); acc=r(T,512)
print('xproj dgrad addmm_ lib %.1f us | mine gemm + add_ %.1f us'%(timeit(lambda: acc.addmm_(dx,Wx)), timeit(lambda: acc.add_(ops.gemm_f32(dx,Wx,True,False)))))
