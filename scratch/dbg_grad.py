import numpy as np, torch, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from oracle import detr_oracle as O
from test_model_gpu import build_model
cfg=O.CONFIG1
P=O.make_params(cfg,0)
batch=O.make_batch(cfg,2,20,seed=1234,num_objects=[3,7])
m=build_model(cfg); m.forward_backward(batch); m.set_weights_dict(P); m.forward_backward(batch)
_,g64=O.train_step_grads(cfg,P,batch,dtype=torch.float64)
_,g32=O.train_step_grads(cfg,P,batch,dtype=torch.float32)
for name in ['AttributePredictionHead/Dense/kernel','CategoryPredictionHead/DenseCateg/kernel','EncoderBackbone/resnet50/conv5_block3_3_conv/kernel']:
    v=[x for x in m.variables if x.name==name][0]
    got=v.grad_numpy().astype(np.float64); want=g64[name]; c32=g32[name].astype(np.float64)
    for lab,a in (('gpu',got),('cpu32',c32)):
        err=(a-want).reshape(-1,want.shape[-1])
        col=np.linalg.norm(err,axis=0); tot=np.linalg.norm(err)
        top=np.sort(col)[::-1][:5]
        frac=(np.abs(a-want)>2e-3*np.abs(want).max()).mean()
        print(name.split('/')[-2],lab,"relL2",tot/np.linalg.norm(want),"top cols share",(top**2).sum()/tot**2,"frac>2e-3max",frac)
