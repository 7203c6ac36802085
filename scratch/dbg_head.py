import numpy as np, torch, sys
sys.path.insert(0,'.')
from oracle import detr_oracle as O
from boosted_detr_amd import kernels as k
cfg=O.CONFIG1
P=O.make_params(cfg,0)
batch=O.make_batch(cfg,2,20,seed=1234,num_objects=[3,7])
net=O.Net(cfg,P,dtype=torch.float64,requires_grad=True)
out=O.forward(net,batch,True)
dec=net.probes['DecoderBlock_0']; dec.retain_grad()
att=out.attribute_preds; att.retain_grad()
out.loss_vector.sum().backward()
d_att=att.grad.detach()            # [2,50,296] fp64
decv=dec.detach()
# standalone fp64 head
W1=net.p['AttributePredictionHead/Dense/kernel'].detach(); b1=net.p['AttributePredictionHead/Dense/bias'].detach()
g=net.p['AttributePredictionHead/BatchNorm/gamma'].detach(); be=net.p['AttributePredictionHead/BatchNorm/beta'].detach()
W2=net.p['AttributePredictionHead/DenseLinear/kernel'].detach(); b2=net.p['AttributePredictionHead/DenseLinear/bias'].detach()
x=decv.reshape(100,256).clone().requires_grad_(True)
h=(x@W1+b1); h.retain_grad(); r=h.relu(); r.retain_grad()
m=r.mean(0); v=r.var(0,unbiased=False); bn=(r-m)/torch.sqrt(v+1e-3)*g+be; bn.retain_grad()
lg=bn@W2+b2; lg.retain_grad(); p=torch.sigmoid(lg)
p.backward(d_att.reshape(100,296))
dev=lambda t: t.float().contiguous().cuda()
# GPU path
xg=dev(x.detach()); w1=dev(W1.T); w2=dev(W2.T)
hg=k.linear_fwd(xg,w1,dev(b1),1)
parts=k.colstats(hg); mm=torch.zeros(1024).cuda(); mv=torch.ones(1024).cuda()
mean,rstd=k.bn_stats(100,1024,parts,1e-3,0.99,False,mm,mv,like=mm)
bng=k.bn_apply(hg,mean,rstd,dev(g),dev(be),None,False)
lgg=k.linear_fwd(bng,w2,dev(b2),0)
pg=k.sigmoid_fwd(lgg)
def rel(a,b): 
    a=a.detach().cpu().double(); b=b.detach().double(); return float((a-b).norm()/b.norm())
print("fwd h",rel(hg,r),"bn",rel(bng,bn),"p",rel(pg,p))
dl=k.sigmoid_bwd(pg,dev(d_att.reshape(100,296)))
print("dlogits",rel(dl,lg.grad))
dbn=k.linear_bwd_data(dl,w2)
print("dbn",rel(dbn,bn.grad))
dr,dg,db,_=k.bn_bwd(dbn,None,hg,mean,rstd,dev(g),False,False)
print("dr",rel(dr,r.grad),"dgamma",rel(dg,(bn.grad*((r-m)/torch.sqrt(v+1e-3))).sum(0)),"dbeta",rel(db,bn.grad.sum(0)))
# same with exact fp64 dbn as input
dr2,_,_,_=k.bn_bwd(dev(bn.grad),None,hg,mean,rstd,dev(g),False,False)
print("dr (exact dbn in)",rel(dr2,r.grad))
dh=k.relu_bwd(hg,dr)
print("dh",rel(dh,h.grad))
print("dW1",rel(k.linear_bwd_weight(dh,xg),(x.detach().T@h.grad).T), "db1", rel(k.colsum(dh),h.grad.sum(0)))
# torch fp32 reference of BN backward for comparison
r32=r.detach().float().requires_grad_(True); m32=r32.mean(0); v32=r32.var(0,unbiased=False)
bn32=(r32-m32)/torch.sqrt(v32+1e-3)*g.float()+be.float(); bn32.backward(bn.grad.float())
print("torch fp32 dr", rel(r32.grad, r.grad))
print("min var", float(v.min()), "n dead", int((v==0).sum()), "rstd max", float(rstd.max()))
