import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import detr_oracle as O
import test_model_gpu as T
from boosted_detr_amd import kernels as k
cfg = O.CONFIG1
batch = O.make_batch(cfg, 2, 20, seed=1234, num_objects=[3, 7])
params = O.make_params(cfg, seed=0)
net = O.Net(cfg, params)
ref = O.forward(net, {"image": batch["image"]}, training=False)
ids, hot = O.decode_predictions(ref.cat_preds, ref.attribute_preds)
rc = ref.cat_preds.detach().numpy()
top2 = np.sort(rc, -1)[..., -2:]
print("oracle: min top1-top2 gap", (top2[..., 1] - top2[..., 0]).min(), "max |cat|", np.abs(rc).max())
for mode in ["fp32", "split", "mixed"]:
    k.set_gemm_precision(mode)
    model = T.build_model(cfg, False)
    model.forward_backward(batch)
    model.set_weights_dict(params)
    # run pieces: full inference
    cat, att, box = model.predict_tensors({"image": batch["image"]}) if hasattr(model, "predict_tensors") else (None, None, None)
    category, attributes, boxes = model({"image": batch["image"]}, training=False)
    vocab = ["<PAD>", "<OOV>"] + model.vocab_dict["category"]
    want = np.array([[vocab[i] for i in row] for row in ids.numpy()])
    print(mode, "mismatched ids", int((category[..., 0] != want).sum()), "of", want.size, "box err", float(np.abs(boxes.cpu().numpy() - ref.box_preds.detach().numpy()).max()),
          "finite", bool(torch.isfinite(boxes).all()))
