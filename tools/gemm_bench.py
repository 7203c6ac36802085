import sys, torch, time
sys.path.insert(0,'.')
from boosted_detr_amd import kernels as k
torch.cuda.set_device(0)
def bench(name, fn, flops, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/iters
    print(f"{name:46s} {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TF/s", flush=True)
shapes=[(25600,1024,1024),(25600,256,1024),(102400,512,128),(409600,256,64),(409600,64,256),(6400,2048,512),(8192,8192,1024)]
for (M,O,K) in shapes:
    x=torch.randn(M,K,device='cuda'); w=torch.randn(O,K,device='cuda'); b=torch.randn(O,device='cuda'); dy=torch.randn(M,O,device='cuda')
    bench(f"fwd  RCxRC M={M} O={O} K={K}", lambda: k.linear_fwd(x,w,b,0), 2*M*O*K)
    bench(f"bwdD RCxXC M={M} O={O} K={K}", lambda: k.linear_bwd_data(dy,w), 2*M*O*K)
    bench(f"bwdW XCxXC M={M} O={O} K={K}", lambda: k.linear_bwd_weight(dy,x), 2*M*O*K)
g=k.ConvGeom(16,40,40,256,256,3,3,1,1)
x=torch.randn(16,40,40,256,device='cuda'); w=torch.randn(256,3,3,256,device='cuda'); b=torch.randn(256,device='cuda'); dy=torch.randn(16,40,40,256,device='cuda')
fl=2*g.M*256*2304
bench("conv3x3 fwd 16x40x40x256", lambda: k.conv2d_fwd(x,w,b,g,0,True), fl)
bench("conv3x3 bwd-data", lambda: k.conv2d_bwd_data(dy,w,g), fl)
bench("conv3x3 bwd-weight", lambda: k.conv2d_bwd_weight(x,dy,g), fl)
