import sys, torch
sys.path.insert(0,'.')
from boosted_detr_amd import kernels as k
torch.cuda.set_device(0)
M,O,K=25600,256,1024
x=torch.randn(M,K,device='cuda'); w=torch.randn(O,K,device='cuda'); b=torch.randn(O,device='cuda')
for _ in range(10): k.linear_fwd(x,w,b,0)
g=k.ConvGeom(16,40,40,256,256,3,3,1,1)
x2=torch.randn(16,40,40,256,device='cuda'); w2=torch.randn(256,3,3,256,device='cuda')
for _ in range(10): k.conv2d_fwd(x2,w2,b,g,0,True)
torch.cuda.synchronize()
