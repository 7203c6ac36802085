import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from test_training_gpu import small_model, small_batch
from boosted_detr_amd.training import SGD
from oracle import detr_oracle as O
cfg,batch=small_batch()
res=[]
for mode in ("direct","temp"):
    m=small_model(dropout=0.1)
    m.compile(optimizer=SGD(1e-2,momentum=.9,nesterov=True,clipnorm=.1))
    m.forward_backward(batch); m.set_weights_dict(O.make_params(cfg,seed=1)); m.steps_done=0
    losses=[]
    for s in range(4):
        if mode=="temp":
            for v in m.variables: v.grad_buf=None
            if getattr(m.optimizer,'flat_grad',None) is not None: m.optimizer._built_for=None
        losses.append(m.logs_to_host(m.train_step(batch))["loss"])
    res.append((losses,{v.name:v.numpy() for v in m.variables}))
    print(mode, losses)
a,b=res[0][1],res[1][1]
worst=sorted(((np.abs(a[k]-b[k]).max()/(np.abs(b[k]).max()+1e-12),k) for k in a), reverse=True)[:6]
print(worst)
