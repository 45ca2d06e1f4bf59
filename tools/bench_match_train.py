import sys, time, torch
sys.path.insert(0,'/root/repo')
from witw_amd import cvig_fov
B=1024
ov=torch.randn(B,16,4,64,device='cuda',requires_grad=True); su=torch.randn(B,16,4,64,device='cuda',requires_grad=True)
for it in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    ori,d=cvig_fov.match(ov,su); loss=cvig_fov.triplet_loss(d)
    torch.cuda.synchronize(); t1=time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); t2=time.perf_counter()
    print('B=%d match+loss fwd %.2f ms, bwd %.2f ms'%(B,(t1-t0)*1e3,(t2-t1)*1e3))
