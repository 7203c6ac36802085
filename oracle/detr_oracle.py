"""CPU oracle for the DETR training-step hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (``boosted_detr_amd``) never does:
it runs hand-written HIP kernels through the C ABI in ``include/bdetr.h`` and
fails loudly when that library is missing.

What this is
------------
A from-scratch CPU restatement (PyTorch-CPU ops, fp32 or fp64, autograd for the
gradients) of the reference's algorithm, following these reference files line by
line (citations are ``/root/reference/ModelComponents/<file>:<line>``):

* ``backbone.py:15-58``      EncoderBackbone  (ResNet-50 branch, 34-39)
* ``backbone.py:66-95``      BackboneNeck
* ``transformers.py:18-102`` MultiheadAttention (incl. the no-permute reshape at 100)
* ``transformers.py:112-151`` AttentionBlock, ``161-193`` FeedForwardBlock
* ``transformers.py:200-235`` EncoderBlock, ``244-315`` ImageEncoderAttention
* ``transformers.py:324-394`` DecoderBlock_NoSelfAttention / DecoderBlock
* ``transformers.py:397-450`` DecoderPrep
* ``prediction_heads.py:13-63, 72-131, 140-201`` the three heads
* ``losses_and_metrics.py:8-72`` loss primitives, ``75-161`` MatchingLoss,
  ``164-192`` MatchingMetric, ``195-251`` mask / cost array / scipy assignment
* ``tokenizers.py:40-82`` Tokenization (ids -> one-hot / multi-hot)
* ``model.py:145-233`` DETR.call, ``boosted_model.py:170-267`` BoostedDETR.call

PARITY STATUS
-------------
**parity unpinned** for the floating-point part: the reference ships no tests,
golden vectors or fixtures, and TensorFlow / tensorflow_addons / Keras are not
installed in the authoring container, so the reference itself cannot be run.
The third-party semantics this file assumes (Keras layer definitions, TFA GIoU
and focal loss, Keras ResNet-50 v1 topology) are listed in SURVEY.md section 8(c)
table S and restated in the docstrings below.

**pinned** for the matcher: the assignment is computed by the reference's real
dependency, ``scipy.optimize.linear_sum_assignment`` (scipy 1.15.3 here), called
exactly as ``losses_and_metrics.py:240-243`` does.
"""
from __future__ import annotations

import hashlib
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

# losses_and_metrics.py:8-11
DEFAULT_CATEGORY_WEIGHT = 1000.0
DEFAULT_BOX_WEIGHT = 1.0
DEFAULT_ATTRIBUTE_WEIGHT = 100.0
DEFAULT_EXIST_WEIGHT = 100.0

KERAS_EPS = 1e-7          # tf.keras.backend.epsilon()
RESNET_BN_EPS = 1.001e-5  # keras.applications.resnet: BatchNormalization(epsilon=1.001e-5)
KERAS_BN_EPS = 1e-3       # BatchNormalization default epsilon
LN_EPS = 1e-3             # transformers.py:137 (explicit) and LayerNormalization default
BN_MOMENTUM = 0.99

RESNET50_STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))  # (filters, blocks, stride1)
CAFFE_MEAN_BGR = (103.939, 116.779, 123.68)


# ----------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------
@dataclass
class Config:
    """Hyper-parameters of one DETR / BoostedDETR instance (model.py:30-35)."""
    image_size: Tuple[int, int] = (224, 224)
    num_object_preds: int = 50
    num_encoder_blocks: int = 1
    num_encoder_heads: int = 8
    encoder_dim: int = 256
    num_decoder_blocks: int = 1
    num_decoder_heads: int = 8
    decoder_dim: int = 256
    num_categories: int = 48      # vocab + <PAD> + <OOV>  (tokenizers.py:22-33)
    num_attributes: int = 296
    attribute_weight: float = 1.0
    classification_only: bool = False
    boosted: bool = False         # BoostedDETR wiring (boosted_model.py)
    dropout_rate: float = 0.0     # reference uses 0.1 (TF RNG, unreproducible) -> parity runs use 0
    stages: Tuple[Tuple[int, int, int], ...] = RESNET50_STAGES

    @property
    def feature_hw(self) -> Tuple[int, int]:
        h, w = self.image_size
        return (_resnet_out(h), _resnet_out(w))


def _resnet_out(s: int) -> int:
    s = (s + 6 - 7) // 2 + 1          # ZeroPad3 + conv7x7/2 valid
    s = (s + 2 - 3) // 2 + 1          # ZeroPad1 + maxpool3x3/2 valid
    for _ in range(3):
        s = (s - 1) // 2 + 1          # 1x1 stride-2 convs of stages 2..4
    return s


CONFIG1 = Config()  # BASELINE.json configs[0]: 2x224x224, 1+1 layers, 50 queries, Fashionpedia sizes
CONFIG2 = Config(image_size=(640, 640), num_object_preds=100, num_encoder_blocks=6,
                 num_decoder_blocks=6, num_categories=82, num_attributes=3, attribute_weight=0.0)


# ----------------------------------------------------------------------------
# seeded, procedural parameters (Keras layouts: Dense [in,out], Conv2D HWIO)
# ----------------------------------------------------------------------------
def _rng_for(name: str, seed: int) -> np.random.Generator:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.random.Generator(np.random.PCG64(int.from_bytes(h[:8], "little")))


def _randn(name, seed, shape, std):
    return (_rng_for(name, seed).standard_normal(shape) * std).astype(np.float32)


def param_specs(cfg: Config) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Ordered list of (name, shape, kind) for every variable of the model.

    kind in {conv_kernel, dense_kernel, bias, gamma, beta, moving_mean,
    moving_var, positional, queries}.  Names follow the Keras layer names of the
    reference (``model.py:68-115``) and of ``keras.applications.ResNet50``.
    """
    specs: List[Tuple[str, Tuple[int, ...], str]] = []

    def conv(name, kh, kw, cin, cout):
        specs.append((f"{name}/kernel", (kh, kw, cin, cout), "conv_kernel"))
        specs.append((f"{name}/bias", (cout,), "bias"))

    def bn(name, c):
        specs.append((f"{name}/gamma", (c,), "gamma"))
        specs.append((f"{name}/beta", (c,), "beta"))
        specs.append((f"{name}/moving_mean", (c,), "moving_mean"))
        specs.append((f"{name}/moving_variance", (c,), "moving_var"))

    def dense(name, cin, cout):
        specs.append((f"{name}/kernel", (cin, cout), "dense_kernel"))
        specs.append((f"{name}/bias", (cout,), "bias"))

    # --- ResNet-50 v1 ---
    bb = "EncoderBackbone/resnet50"
    conv(f"{bb}/conv1_conv", 7, 7, 3, 64)
    bn(f"{bb}/conv1_bn", 64)
    cin = 64
    for si, (f, nblocks, _stride) in enumerate(cfg.stages):
        for bi in range(nblocks):
            p = f"{bb}/conv{si + 2}_block{bi + 1}"
            if bi == 0:
                conv(f"{p}_0_conv", 1, 1, cin, 4 * f)
                bn(f"{p}_0_bn", 4 * f)
            conv(f"{p}_1_conv", 1, 1, cin, f)
            bn(f"{p}_1_bn", f)
            conv(f"{p}_2_conv", 3, 3, f, f)
            bn(f"{p}_2_bn", f)
            conv(f"{p}_3_conv", 1, 1, f, 4 * f)
            bn(f"{p}_3_bn", 4 * f)
            cin = 4 * f
    # --- neck (backbone.py:76-80) ---
    bn("BackboneNeck/batch_norm1", cin)
    conv("BackboneNeck/conv2d_downscaler", 1, 1, cin, cfg.encoder_dim)
    bn("BackboneNeck/batch_norm2", cfg.encoder_dim)

    D, Dd = cfg.encoder_dim, cfg.decoder_dim
    r, c = cfg.feature_hw

    def attention_block(p, dq):
        for proj in ("QueryProjection", "KeyProjection", "ValueProjection", "OutputProjection"):
            dense(f"{p}/AttentionLayer/{proj}", dq, dq)
        specs.append((f"{p}/LayerNorm/gamma", (dq,), "gamma"))
        specs.append((f"{p}/LayerNorm/beta", (dq,), "beta"))

    def ffn(p, dq):
        dense(f"{p}/DenseRelu", dq, dq)
        dense(f"{p}/DenseLinear", dq, dq)
        specs.append((f"{p}/LayerNorm/gamma", (dq,), "gamma"))
        specs.append((f"{p}/LayerNorm/beta", (dq,), "beta"))

    def encoder(name, nblocks):
        for i in range(nblocks):
            attention_block(f"{name}/EncoderBlock_{i}/SelfAttentionBlock", D)
            ffn(f"{name}/EncoderBlock_{i}/FeedForwardBlock", D)
        specs.append((f"{name}/positional_encoding", (r, c, D), "positional"))

    if cfg.boosted:
        for i in range(cfg.num_decoder_blocks):   # boosted_model.py:86-92
            encoder(f"ImageEncoderAttention_{i}", 1)
    else:
        encoder("ImageEncoderAttention", cfg.num_encoder_blocks)

    specs.append(("DecoderPrep/init_decoder_features", (cfg.num_object_preds, Dd), "queries"))
    for i in range(cfg.num_decoder_blocks):
        p = f"DecoderBlock_{i}"
        if i > 0:
            attention_block(f"{p}/SelfAttentionBlock", Dd)
        attention_block(f"{p}/JointAttentionBlock", Dd)
        ffn(f"{p}/FeedForwardBlock", Dd)

    def heads(suffix, hidden_cls):
        dense(f"CategoryPredictionHead{suffix}/DenseCateg", Dd, hidden_cls)
        bn(f"CategoryPredictionHead{suffix}/BatchNorm", hidden_cls)
        dense(f"CategoryPredictionHead{suffix}/DenseLogits", hidden_cls, cfg.num_categories)
        dense(f"AttributePredictionHead{suffix}/Dense", Dd, hidden_cls)
        bn(f"AttributePredictionHead{suffix}/BatchNorm", hidden_cls)
        dense(f"AttributePredictionHead{suffix}/DenseLinear", hidden_cls, cfg.num_attributes)
        dense(f"BoxPredictionHead{suffix}/Dense", Dd, Dd)
        bn(f"BoxPredictionHead{suffix}/BatchNorm", Dd)
        dense(f"BoxPredictionHead{suffix}/BoxCoords", Dd, 4)

    if cfg.boosted:
        for i in range(cfg.num_decoder_blocks):   # boosted_model.py:113-137: hidden = decoder_dim
            heads(f"_{i}", Dd)
    else:
        heads("", 4 * Dd)                          # model.py:99-115: hidden = 4*decoder_dim
    return specs


def positional_init(r: int, c: int, D: int) -> np.ndarray:
    """transformers.py:282-292: value depends on the *position* parity, computed
    in Python float (fp64) and stored as fp32."""
    out = np.empty((r * c, D), dtype=np.float64)
    for k in range(r * c):
        even, odd = k % 2, (k + 1) % 2
        for dim in range(D):
            denom = 2 * (1 + dim) / D
            out[k, dim] = even * math.sin(k / denom) + odd * math.cos(k / denom)
    return out.reshape(r, c, D).astype(np.float32)


def make_params(cfg: Config, seed: int = 0, perturb: bool = True) -> Dict[str, np.ndarray]:
    """Seeded synthetic weights.  ``perturb=True`` makes every tensor non-trivial
    (random gamma/beta/bias/moving stats/queries) so parity tests exercise all
    terms; it is *not* the Keras initialiser (that lives in the product's
    ``build()``).  Kernel std follows the variance-scaling rule of the layer's
    initialiser so activations stay O(1) through ~110 layers."""
    params: Dict[str, np.ndarray] = {}
    for name, shape, kind in param_specs(cfg):
        if kind == "conv_kernel":
            fan_in = shape[0] * shape[1] * shape[2]
            params[name] = _randn(name, seed, shape, math.sqrt(2.0 / fan_in))
        elif kind == "dense_kernel":
            params[name] = _randn(name, seed, shape, math.sqrt(2.0 / (shape[0] + shape[1])))
        elif kind == "bias":
            params[name] = _randn(name, seed, shape, 0.05) if perturb else np.zeros(shape, np.float32)
        elif kind == "gamma":
            params[name] = (1.0 + _randn(name, seed, shape, 0.1)) if perturb else np.ones(shape, np.float32)
        elif kind == "beta":
            params[name] = _randn(name, seed, shape, 0.1) if perturb else np.zeros(shape, np.float32)
        elif kind == "moving_mean":
            params[name] = _randn(name, seed, shape, 0.1) if perturb else np.zeros(shape, np.float32)
        elif kind == "moving_var":
            params[name] = (1.0 + np.abs(_randn(name, seed, shape, 0.2))) if perturb else np.ones(shape, np.float32)
        elif kind == "positional":
            params[name] = positional_init(*shape)
        elif kind == "queries":  # reference initialises zeros (transformers.py:428-431)
            params[name] = _randn(name, seed, shape, 0.5) if perturb else np.zeros(shape, np.float32)
        else:
            raise ValueError(kind)
        params[name] = np.ascontiguousarray(params[name], dtype=np.float32)
    return params


def trainable_names(cfg: Config) -> List[str]:
    return [n for n, _, k in param_specs(cfg) if k not in ("moving_mean", "moving_var")]


# ----------------------------------------------------------------------------
# seeded synthetic inputs
# ----------------------------------------------------------------------------
def make_batch(cfg: Config, batch: int, max_objects: int, seed: int = 1234,
               num_objects: Optional[Sequence[int]] = None, attrs_per_object: int = 3,
               image_hw: Optional[Tuple[int, int]] = None) -> Dict[str, np.ndarray]:
    """COCO-shaped synthetic batch with *integer* (pre-tokenised) labels.

    ``category`` int32 [B,M] (0=<PAD>, 1=<OOV>, vocab from 2), ``attribute`` int32
    [B,M,Amax] (0=<PAD>), ``bbox`` f32 [B,M,4] COCO [xmin,ymin,w,h] padded with
    -10 (pipeline.py:131-186), ``num_objects`` int32 [B]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    h, w = image_hw or cfg.image_size
    image = rng.random((batch, h, w, 3), dtype=np.float32)
    if num_objects is None:
        num_objects = np.clip(1 + rng.poisson(6.3, size=batch), 1, min(93, max_objects))
    num_objects = np.asarray(num_objects, dtype=np.int32)
    M = max_objects
    category = np.zeros((batch, M), np.int32)
    attribute = np.zeros((batch, M, attrs_per_object), np.int32)
    bbox = np.full((batch, M, 4), -10.0, np.float32)
    for b in range(batch):
        n = int(num_objects[b])
        category[b, :n] = rng.integers(2, cfg.num_categories, size=n)
        if cfg.num_attributes > 2:
            k = rng.integers(0, attrs_per_object + 1, size=n)
            for m in range(n):
                attribute[b, m, :k[m]] = rng.integers(2, cfg.num_attributes, size=k[m])
        bbox[b, :n, 0:2] = rng.uniform(0.0, 0.6, size=(n, 2))
        bbox[b, :n, 2:4] = rng.uniform(0.05, 0.4, size=(n, 2))
    return {"image": image, "category": category, "attribute": attribute,
            "bbox": bbox.astype(np.float32), "num_objects": num_objects}


# ----------------------------------------------------------------------------
# layer restatements
# ----------------------------------------------------------------------------
class Net:
    """Holds torch views of the parameters plus BN moving-stat updates and probes."""

    def __init__(self, cfg: Config, params: Dict[str, np.ndarray], dtype=torch.float32,
                 requires_grad: bool = False, frozen_bn: bool = False):
        self.cfg = cfg
        self.dtype = dtype
        self.frozen_bn = frozen_bn       # BN uses moving stats even when training (S18 / 8e test)
        self.p: Dict[str, torch.Tensor] = {}
        trainable = set(trainable_names(cfg))
        for k, v in params.items():
            t = torch.from_numpy(np.array(v)).to(dtype)
            if requires_grad and k in trainable:
                t.requires_grad_(True)
            self.p[k] = t
        self.new_moving: Dict[str, torch.Tensor] = {}
        self.probes: Dict[str, torch.Tensor] = {}
        self.dropout_masks: Dict[str, torch.Tensor] = {}

    # -- primitives -----------------------------------------------------------
    def conv(self, x, name, stride=1, padding=0):
        """Keras Conv2D, NHWC/HWIO, bias (S7).  x is NCHW internally."""
        w = self.p[f"{name}/kernel"].permute(3, 2, 0, 1)
        return F.conv2d(x, w, self.p[f"{name}/bias"], stride=stride, padding=padding)

    def bn(self, x, name, eps, training, channel_dim=1):
        """Keras BatchNormalization (S5): batch statistics + biased variance when
        training; moving stats updated with momentum .99 (fused 4-D path uses the
        Bessel-corrected variance for the moving estimate)."""
        g, b = self.p[f"{name}/gamma"], self.p[f"{name}/beta"]
        mm, mv = self.p[f"{name}/moving_mean"], self.p[f"{name}/moving_variance"]
        shape = [1] * x.dim()
        shape[channel_dim] = -1
        if training and not self.frozen_bn:
            dims = [d for d in range(x.dim()) if d != channel_dim]
            mean = x.mean(dim=dims)
            var = x.var(dim=dims, unbiased=False)
            n = x.numel() // x.shape[channel_dim]
            var_m = var * (n / max(n - 1, 1)) if x.dim() == 4 else var
            with torch.no_grad():
                self.new_moving[f"{name}/moving_mean"] = mm * BN_MOMENTUM + mean * (1 - BN_MOMENTUM)
                self.new_moving[f"{name}/moving_variance"] = mv * BN_MOMENTUM + var_m * (1 - BN_MOMENTUM)
        else:
            mean, var = mm, mv
        inv = torch.rsqrt(var + eps)
        return (x - mean.view(shape)) * (inv * g).view(shape) + b.view(shape)

    def dense(self, x, name):
        return x @ self.p[f"{name}/kernel"] + self.p[f"{name}/bias"]

    def layer_norm(self, x, name):
        """Keras LayerNormalization over the last axis, biased variance, eps 1e-3 (S6)."""
        g, b = self.p[f"{name}/gamma"], self.p[f"{name}/beta"]
        mean = x.mean(dim=-1, keepdim=True)
        var = x.var(dim=-1, unbiased=False, keepdim=True)
        return (x - mean) * torch.rsqrt(var + LN_EPS) * g + b

    def dropout(self, x, name, training):
        """Keras Dropout(.1): inverted dropout (S8).  The mask is drawn from a seeded
        numpy stream keyed by the layer name so a test can inject the same mask."""
        rate = self.cfg.dropout_rate
        if not training or rate == 0.0:
            return x
        if name not in self.dropout_masks:
            keep = _rng_for("dropout:" + name, 0).random(tuple(x.shape)) >= rate
            self.dropout_masks[name] = torch.from_numpy(keep)
        return x * self.dropout_masks[name].to(x.dtype) / (1.0 - rate)

    # -- backbone.py:49-58 ------------------------------------------------------
    def image_prep(self, image_nhwc: torch.Tensor) -> torch.Tensor:
        """clip -> Resizing (bilinear, identity at equal size; S2) -> x*255.5 truncated
        to uint8 then cast f32 (S1) -> caffe preprocess: RGB->BGR, minus mean (S3).
        Returns NHWC."""
        h, w = self.cfg.image_size
        x = image_nhwc.to(torch.float32).clamp(0.0, 1.0)
        if x.shape[1] != h or x.shape[2] != w:
            x = F.interpolate(x.permute(0, 3, 1, 2), size=(h, w), mode="bilinear",
                              align_corners=False, antialias=False).permute(0, 2, 3, 1)
        x = torch.floor(x * 255.5).clamp(0, 255)        # saturate_cast<uint8>(x*255.5)
        x = x.flip(-1) - torch.tensor(CAFFE_MEAN_BGR, dtype=torch.float32)
        return x.to(self.dtype)

    def resnet50(self, x_nhwc: torch.Tensor, training: bool) -> torch.Tensor:
        """Keras ResNet-50 v1, include_top=False (S4).  Stride sits on the first 1x1."""
        bb = "EncoderBackbone/resnet50"
        x = x_nhwc.permute(0, 3, 1, 2)
        x = self.conv(x, f"{bb}/conv1_conv", stride=2, padding=3)
        x = F.relu(self.bn(x, f"{bb}/conv1_bn", RESNET_BN_EPS, training))
        self.probes["conv1_relu"] = x
        x = F.max_pool2d(F.pad(x, (1, 1, 1, 1)), 3, 2)       # zero pad (post-ReLU => same as -inf pad)
        self.probes["pool1"] = x
        for si, (f, nblocks, stride1) in enumerate(self.cfg.stages):
            for bi in range(nblocks):
                p = f"{bb}/conv{si + 2}_block{bi + 1}"
                s = stride1 if bi == 0 else 1
                if bi == 0:
                    sc = self.bn(self.conv(x, f"{p}_0_conv", stride=s), f"{p}_0_bn", RESNET_BN_EPS, training)
                else:
                    sc = x
                y = F.relu(self.bn(self.conv(x, f"{p}_1_conv", stride=s), f"{p}_1_bn", RESNET_BN_EPS, training))
                y = F.relu(self.bn(self.conv(y, f"{p}_2_conv", padding=1), f"{p}_2_bn", RESNET_BN_EPS, training))
                y = self.bn(self.conv(y, f"{p}_3_conv"), f"{p}_3_bn", RESNET_BN_EPS, training)
                x = F.relu(sc + y)
            self.probes[f"conv{si + 2}_out"] = x
        return x  # NCHW

    def neck(self, x_nchw, training):
        """backbone.py:90-95: BN -> 1x1 conv + tanh -> BN."""
        x = self.bn(x_nchw, "BackboneNeck/batch_norm1", KERAS_BN_EPS, training)
        x = torch.tanh(self.conv(x, "BackboneNeck/conv2d_downscaler"))
        x = self.bn(x, "BackboneNeck/batch_norm2", KERAS_BN_EPS, training)
        x = x.permute(0, 2, 3, 1)  # NHWC
        self.probes["neck"] = x
        return x

    # -- transformers.py ------------------------------------------------------
    def mha(self, q, k, v, name, heads):
        """transformers.py:68-102.  Output of softmax(QK^T/sqrt(d))V is [B,h,q,d] and is
        reshaped to [B,q,h*d] WITHOUT permuting heads back (line 100) - reproduced."""
        B, nq, dq = q.shape
        d = dq // heads                                  # AttentionBlock: key_dim = query_dim // heads (129)
        Q = self.dense(q, f"{name}/QueryProjection").reshape(B, nq, heads, d).permute(0, 2, 1, 3)
        K = self.dense(k, f"{name}/KeyProjection").reshape(B, k.shape[1], heads, d).permute(0, 2, 3, 1)
        V = self.dense(v, f"{name}/ValueProjection").reshape(B, v.shape[1], heads, d).permute(0, 2, 1, 3)
        scale = torch.tensor(1.0 / math.sqrt(float(d)), dtype=torch.float32).to(self.dtype)
        s = torch.softmax((Q @ K) * scale, dim=-1)       # mask=None -> multiply by ones (92-94)
        o = (s @ V).reshape(B, nq, heads * d)            # row-major reinterpretation (F5)
        return self.dense(o, f"{name}/OutputProjection")

    def attention_block(self, q, k, v, name, heads, training):
        a = self.mha(q, k, v, f"{name}/AttentionLayer", heads)
        a = self.dropout(a, f"{name}/Dropout", training)
        return self.layer_norm(q + a, f"{name}/LayerNorm")

    def ffn(self, x, name, training):
        """transformers.py:182-193: hidden width == feature width (174-177)."""
        y = F.relu(self.dense(x, f"{name}/DenseRelu"))
        y = self.dense(y, f"{name}/DenseLinear")
        y = self.dropout(y, f"{name}/Dropout", training)
        return self.layer_norm(x + y, f"{name}/LayerNorm")

    def image_encoder(self, feat_nhwc, name, nblocks, training):
        """transformers.py:294-315."""
        B, r, c, D = feat_nhwc.shape
        pos = self.p[f"{name}/positional_encoding"].reshape(1, r * c, D).expand(B, -1, -1)
        x = feat_nhwc.reshape(B, r * c, D)
        for i in range(nblocks):
            p = f"{name}/EncoderBlock_{i}"
            qk = x + pos
            x = self.attention_block(qk, qk, x, f"{p}/SelfAttentionBlock", self.cfg.num_encoder_heads, training)
            x = self.ffn(x, f"{p}/FeedForwardBlock", training)
            self.probes[f"{name}/block{i}"] = x
        return x.reshape(B, r, c, D), pos.reshape(B, r, c, D)

    def decoder_prep(self, enc_nhwc, pos_nhwc):
        """transformers.py:433-450."""
        B, r, c, D = enc_nhwc.shape
        value = enc_nhwc.reshape(B, r * c, D)
        key = value + pos_nhwc.reshape(B, r * c, D)
        dec = self.p["DecoderPrep/init_decoder_features"].unsqueeze(0).expand(B, -1, -1)
        return value, dec, key

    def decoder_block(self, i, value, dec, key, training):
        """transformers.py:340-353 (block 0) / 374-394 (blocks >= 1)."""
        p, h = f"DecoderBlock_{i}", self.cfg.num_decoder_heads
        if i > 0:
            dec = self.attention_block(dec, dec, dec, f"{p}/SelfAttentionBlock", h, training)
        dec = self.attention_block(dec, key, value, f"{p}/JointAttentionBlock", h, training)
        dec = self.ffn(dec, f"{p}/FeedForwardBlock", training)
        self.probes[f"{p}"] = dec
        return dec

    # -- prediction_heads.py --------------------------------------------------
    def heads(self, dec, suffix, training):
        def trunk(x, dname, bname):
            x = F.relu(self.dense(x, dname))
            return self.bn(x, bname, KERAS_BN_EPS, training, channel_dim=2)

        # the pre-activation outputs (prediction_heads.py:111 softmax input, 180 sigmoid input, 44 box-sigmoid input) are kept as
        # probes "<head>/logits": the parity tests compare them as well as the activated predictions
        c = f"CategoryPredictionHead{suffix}"
        cat_logits = self.dense(trunk(dec, f"{c}/DenseCateg", f"{c}/BatchNorm"), f"{c}/DenseLogits")
        cat = torch.softmax(cat_logits, dim=-1)
        a = f"AttributePredictionHead{suffix}"
        att_logits = self.dense(trunk(dec, f"{a}/Dense", f"{a}/BatchNorm"), f"{a}/DenseLinear")
        att = torch.sigmoid(att_logits)
        b = f"BoxPredictionHead{suffix}"
        box_logits = self.dense(trunk(dec, f"{b}/Dense", f"{b}/BatchNorm"), f"{b}/BoxCoords")
        box = 3.0 * torch.sigmoid(box_logits / 100.0) - 1.0
        self.probes[f"{c}/logits"], self.probes[f"{a}/logits"], self.probes[f"{b}/logits"] = cat_logits, att_logits, box_logits
        return cat, att, box


# ----------------------------------------------------------------------------
# tokenizers.py:40-82 on integer ids
# ----------------------------------------------------------------------------
def tokens_to_hot(category_ids: np.ndarray, attribute_ids: np.ndarray, C: int, A: int, dtype=torch.float32):
    cat = F.one_hot(torch.from_numpy(category_ids.astype(np.int64)), C).to(dtype)          # [B,M,C]
    att = F.one_hot(torch.from_numpy(attribute_ids.astype(np.int64)), A).amax(dim=2).to(dtype)  # [B,M,A]
    return cat, att


# ----------------------------------------------------------------------------
# losses_and_metrics.py
# ----------------------------------------------------------------------------
def safe_clip(p):  # 26-27
    return p.clamp(0.001, 0.999)


def keras_bce(y_true, y_pred):
    """Keras binary_crossentropy, from_logits=False, mean over the last axis (S9)."""
    eps = KERAS_EPS
    out = y_pred.clamp(eps, 1.0 - eps)
    bce = y_true * torch.log(out + eps) + (1 - y_true) * torch.log(1 - out + eps)
    return (-bce).mean(dim=-1)


def category_cost(cat_true, cat_pred):
    """CostArray(CategoryLoss) (44-49, 222-225): [B,M,1,C] x [B,1,N,C] -> [B,M,N]."""
    yt = cat_true.unsqueeze(-2)
    yp = cat_pred.unsqueeze(-3)
    return keras_bce(yt, safe_clip(yp) * yt)


def attribute_cost(att_true, att_pred, alpha=0.25, gamma=2.0):
    """CostArray(AttributeLoss) (51-57) with TFA SigmoidFocalCrossEntropy (S10)."""
    yt = att_true.unsqueeze(-2).unsqueeze(-1)            # [B,M,1,A,1]
    yp = safe_clip(att_pred).unsqueeze(-3).unsqueeze(-1)  # [B,1,N,A,1]
    eps = KERAS_EPS
    out = yp.clamp(eps, 1.0 - eps)
    ce = -(yt * torch.log(out + eps) + (1 - yt) * torch.log(1 - out + eps))
    p_t = yt * yp + (1 - yt) * (1 - yp)
    alpha_f = yt * alpha + (1 - yt) * (1 - alpha)
    focal = (alpha_f * (1.0 - p_t) ** gamma * ce).sum(dim=-1)   # [B,M,N,A]
    return focal.mean(dim=-1)


def coco_to_tf(box):  # 59-66
    xmin, ymin, w, h = box[..., 0:1], box[..., 1:2], box[..., 2:3], box[..., 3:4]
    return torch.cat([ymin, xmin, ymin + h, xmin + w], dim=-1)


def _div_no_nan(a, b):
    safe = torch.where(b == 0, torch.ones_like(b), b)
    return torch.where(b == 0, torch.zeros_like(a), a / safe)


def tfa_giou(b1, b2, mode="giou"):
    """tensorflow_addons.losses.giou_loss's _calculate_giou (S11); boxes
    [ymin,xmin,ymax,xmax]; broadcasting pairwise."""
    zero = torch.zeros((), dtype=b1.dtype)
    b1_ymin, b1_xmin, b1_ymax, b1_xmax = b1.unbind(-1)
    b2_ymin, b2_xmin, b2_ymax, b2_xmax = b2.unbind(-1)
    b1_w = torch.maximum(zero, b1_xmax - b1_xmin)
    b1_h = torch.maximum(zero, b1_ymax - b1_ymin)
    b2_w = torch.maximum(zero, b2_xmax - b2_xmin)
    b2_h = torch.maximum(zero, b2_ymax - b2_ymin)
    b1_area, b2_area = b1_w * b1_h, b2_w * b2_h
    i_ymin, i_xmin = torch.maximum(b1_ymin, b2_ymin), torch.maximum(b1_xmin, b2_xmin)
    i_ymax, i_xmax = torch.minimum(b1_ymax, b2_ymax), torch.minimum(b1_xmax, b2_xmax)
    i_w = torch.maximum(zero, i_xmax - i_xmin)
    i_h = torch.maximum(zero, i_ymax - i_ymin)
    inter = i_w * i_h
    union = b1_area + b2_area - inter
    iou = _div_no_nan(inter, union)
    if mode == "iou":
        return iou
    e_ymin, e_xmin = torch.minimum(b1_ymin, b2_ymin), torch.minimum(b1_xmin, b2_xmin)
    e_ymax, e_xmax = torch.maximum(b1_ymax, b2_ymax), torch.maximum(b1_xmax, b2_xmax)
    e_w = torch.maximum(zero, e_xmax - e_xmin)
    e_h = torch.maximum(zero, e_ymax - e_ymin)
    enclose = e_w * e_h
    return iou - _div_no_nan(enclose - union, enclose)


def box_cost(box_true, box_pred, giou_weight=2.0, l2_weight=5.0):
    """CostArray(BoxLoss) (68-72): 2*(1-GIoU) + 5*mean4((10*delta)^2)."""
    yt = coco_to_tf(box_true).unsqueeze(-2)   # [B,M,1,4]
    yp = coco_to_tf(box_pred).unsqueeze(-3)   # [B,1,N,4]
    giou_loss = 1.0 - tfa_giou(yt, yp, "giou")
    l2 = ((10.0 * yt - 10.0 * yp) ** 2).mean(dim=-1)
    return giou_weight * giou_loss + l2_weight * l2


def scipy_assignment_mask(cost: np.ndarray, num_objects: np.ndarray):
    """MatchingAssignment.scipy_linear_assignment_mask (234-245), verbatim call
    pattern: fp32 slice cost[i, :n_i, :] handed to scipy.  Also returns the index
    lists (int64) per image."""
    masks = np.zeros_like(cost)
    matches = []
    for i in range(cost.shape[0]):
        n = int(num_objects[i])
        rows, cols = linear_sum_assignment(cost[i, :n, :])
        masks[i][rows, cols] = 1.0
        matches.append((rows.astype(np.int64), cols.astype(np.int64)))
    return masks, matches


@dataclass
class LossOut:
    total: torch.Tensor
    category: torch.Tensor
    attribute: torch.Tensor
    box: torch.Tensor
    exist: torch.Tensor
    iou: torch.Tensor
    cost_total: torch.Tensor
    cost_components: Tuple[torch.Tensor, torch.Tensor, torch.Tensor]
    mask: torch.Tensor
    matches: list


def matching_loss(cat_true, att_true, bbox, num_objects, cat_pred, att_pred, box_pred,
                  attribute_weight=DEFAULT_ATTRIBUTE_WEIGHT, box_weight=DEFAULT_BOX_WEIGHT,
                  category_weight=DEFAULT_CATEGORY_WEIGHT, exist_weight=DEFAULT_EXIST_WEIGHT) -> LossOut:
    """MatchingLoss.call (111-161) + MatchingMetric (176-192)."""
    dt = cat_pred.dtype
    c_cost = category_weight * category_cost(cat_true, cat_pred)
    a_cost = attribute_weight * attribute_cost(att_true, att_pred)
    b_cost = box_weight * box_cost(bbox, box_pred)
    total_cost = c_cost + b_cost + a_cost                       # summation order of line 130
    # the reference hands the fp32 tensor to numpy (tf.numpy_function, 249-250)
    cost_np = total_cost.detach().to(torch.float32).numpy()
    mask_np, matches = scipy_assignment_mask(cost_np, np.asarray(num_objects))
    mask = torch.from_numpy(mask_np).to(dt)
    assigned = mask.amax(dim=-2).unsqueeze(-1)                   # [B,N,1]  (206-207)
    c_m, a_m, b_m = mask * c_cost, mask * a_cost, mask * b_cost
    exist = exist_weight * keras_bce(1.0 - assigned, safe_clip(cat_pred[..., 0:1]))   # [B,N]
    tot = 1.0 + float(np.sum(num_objects))
    np1 = 1.0 + float(cat_pred.shape[1])
    cat_l = c_m.sum(dim=(-2, -1)) / tot
    att_l = a_m.sum(dim=(-2, -1)) / tot
    box_l = b_m.sum(dim=(-2, -1)) / tot
    exist_l = exist.mean(dim=-1) / np1
    total = cat_l + att_l + box_l + exist_l
    # metric: IoU on the RAW COCO-format boxes (188) - quirk reproduced
    iou = tfa_giou(bbox.unsqueeze(-2), box_pred.unsqueeze(-3), "iou")
    iou = (mask * iou).sum(dim=(1, 2)) / tot
    return LossOut(total, cat_l, att_l, box_l, exist_l, iou, total_cost,
                   (c_cost, a_cost, b_cost), mask, matches)


# ----------------------------------------------------------------------------
# model.py / boosted_model.py
# ----------------------------------------------------------------------------
@dataclass
class StepOut:
    cat_preds: torch.Tensor
    attribute_preds: torch.Tensor
    box_preds: torch.Tensor
    loss: Optional[LossOut] = None            # DETR: the single loss; Boosted: the last learner's
    loss_vector: Optional[torch.Tensor] = None  # [B] value handed to add_loss
    metrics: Dict[str, torch.Tensor] = field(default_factory=dict)
    learner_losses: List[LossOut] = field(default_factory=list)
    probes: Dict[str, torch.Tensor] = field(default_factory=dict)
    new_moving: Dict[str, torch.Tensor] = field(default_factory=dict)


def forward(net: Net, batch: Dict[str, np.ndarray], training: bool = True) -> StepOut:
    """DETR.call (model.py:145-233) or BoostedDETR.call (boosted_model.py:170-267)."""
    cfg = net.cfg
    image = torch.from_numpy(np.asarray(batch["image"]))
    x = net.image_prep(image)
    net.probes["image_prep"] = x
    x = net.resnet50(x, training)
    feat = net.neck(x, training)

    y_true = None
    if training:
        cat_t, att_t = tokens_to_hot(batch["category"], batch["attribute"], cfg.num_categories,
                                     cfg.num_attributes, net.dtype)
        bbox = torch.from_numpy(np.asarray(batch["bbox"])).to(net.dtype)
        y_true = (cat_t, att_t, bbox, np.asarray(batch["num_objects"]))
    box_w = 0.0 if cfg.classification_only else DEFAULT_BOX_WEIGHT

    def loss_of(cat, att, box):
        return matching_loss(y_true[0], y_true[1], y_true[2], y_true[3], cat, att, box,
                             attribute_weight=cfg.attribute_weight, box_weight=box_w)

    out = None
    if not cfg.boosted:
        enc, pos = net.image_encoder(feat, "ImageEncoderAttention", cfg.num_encoder_blocks, training)
        value, dec, key = net.decoder_prep(enc, pos)
        for i in range(cfg.num_decoder_blocks):
            dec = net.decoder_block(i, value, dec, key, training)
        cat, att, box = net.heads(dec, "", training)    # only at the last block (model.py:179,186)
        out = StepOut(cat, att, box)
        if training:
            lo = loss_of(cat, att, box)
            out.loss, out.loss_vector, out.learner_losses = lo, lo.total, [lo]
            out.metrics = {"Category_Loss": lo.category, "Attribute_Loss": lo.attribute,
                           "Box_Loss": lo.box, "Existence_Loss": lo.exist, "IOU": lo.iou}
    else:
        enc = feat
        cat_sum = att_sum = box_sum = None
        losses: List[LossOut] = []
        for i in range(cfg.num_decoder_blocks):
            enc, pos = net.image_encoder(enc, f"ImageEncoderAttention_{i}", 1, training)
            value, dec, key = net.decoder_prep(enc, pos)     # queries re-tiled each learner (210-211)
            dec = net.decoder_block(i, value, dec, key, training)
            cat_i, att_i, box_i = net.heads(dec, f"_{i}", training)
            if i == 0:                                       # learner 0 counted twice (222-229)
                cat_sum, att_sum, box_sum = cat_i, att_i, box_i
            cat_sum, att_sum, box_sum = cat_sum + cat_i, att_sum + att_i, box_sum + box_i
            if training:
                losses.append(loss_of(cat_sum, att_sum, box_sum))
        out = StepOut(cat_sum, att_sum, box_sum)
        if training:
            out.learner_losses = losses
            out.loss = losses[-1]
            out.loss_vector = sum(l.total for l in losses)
            out.metrics = {"Category_Loss": sum(l.category for l in losses),
                           "Attribute_Loss": sum(l.attribute for l in losses),
                           "Box_Loss": sum(l.box for l in losses),
                           "Existence_Loss": sum(l.exist for l in losses),
                           "IOU": losses[-1].iou}
    out.probes = net.probes
    out.new_moving = net.new_moving
    return out


def train_step_grads(cfg: Config, params: Dict[str, np.ndarray], batch, dtype=torch.float32,
                     frozen_bn: bool = False, loss_scale: float = 1.0):
    """Keras train_step with a [B] add_loss and no compiled loss (S14): the gradient
    is that of sum_b loss_b.  Returns (StepOut, {name: grad ndarray})."""
    net = Net(cfg, params, dtype=dtype, requires_grad=True, frozen_bn=frozen_bn)
    out = forward(net, batch, training=True)
    (out.loss_vector.sum() * loss_scale).backward()
    grads = {k: (v.grad.detach().numpy() if v.grad is not None else np.zeros(tuple(v.shape), np.float32))
             for k, v in net.p.items() if v.requires_grad}
    return out, grads


def decode_predictions(cat_preds: torch.Tensor, attribute_preds: torch.Tensor):
    """InverseTokenization.call (tokenizers.py:122-139) up to the id level: argmax
    category id (first max on ties, S13) and the >=0.5 attribute indicator."""
    return cat_preds.argmax(dim=-1), (attribute_preds >= 0.5)


# ----------------------------------------------------------------------------
# Keras optimizer restatement (S15) - used by optimizer parity tests
# ----------------------------------------------------------------------------
def sgd_nesterov_clipnorm(w, g, v, lr, momentum=0.9, clipnorm=0.1):
    """Per-tensor clip-by-norm then Keras SGD(nesterov=True):
    v <- m*v - lr*g ; w <- w + m*v - lr*g."""
    g = np.asarray(g, np.float32)
    norm = np.sqrt(np.sum(g.astype(np.float32) ** 2, dtype=np.float32))
    if norm > clipnorm:
        g = g * np.float32(clipnorm) / norm
    v_new = np.float32(momentum) * v - np.float32(lr) * g
    w_new = w + np.float32(momentum) * v_new - np.float32(lr) * g
    return w_new.astype(np.float32), v_new.astype(np.float32)


def cosine_decay_restarts(step, initial_lr=1e-3, first_decay_steps=4000, t_mul=2.0, m_mul=0.95, alpha=0.1):
    """tf.keras.optimizers.schedules.CosineDecayRestarts (notebook cell 26)."""
    completed = step / first_decay_steps
    if t_mul == 1.0:
        i_restart = math.floor(completed)
        completed -= i_restart
    else:
        i_restart = math.floor(math.log(1.0 - completed * (1.0 - t_mul)) / math.log(t_mul))
        sum_r = (1.0 - t_mul ** i_restart) / (1.0 - t_mul)
        completed = (completed - sum_r) / t_mul ** i_restart
    m_fac = m_mul ** i_restart
    cosine = 0.5 * m_fac * (1.0 + math.cos(math.pi * completed))
    return initial_lr * ((1 - alpha) * cosine + alpha)
