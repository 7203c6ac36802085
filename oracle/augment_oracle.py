"""CPU oracle for the input-step augmentations.  TEST INFRASTRUCTURE ONLY (tests/ may import it).

Restates /root/reference/ModelComponents/pipeline.py:274-341 with the tf.image semantics it relies on
(public TensorFlow documentation / kernels; cannot be run here - **parity unpinned**):
  * tf.image.resize(bilinear, antialias=False): half-pixel centres
  * tf.image.pad_to_bounding_box: zero padding
  * tf.image.adjust_contrast: (x - mean_c) * f + mean_c, mean over H,W per channel
  * tf.image.adjust_brightness: x + delta
  * tf.image.adjust_saturation: adjust_saturation_op.cc's rgb->hsv, s = clamp(s*f,0,1), hsv->rgb
  * tf.image.random_jpeg_quality (319-325): oracle/jpeg_oracle.py (that part IS pinned: bit-exact against libjpeg-turbo)
The random draws are inputs (TF's RNG is not reproducible)."""
import numpy as np
import torch
import torch.nn.functional as F


def downsize_with_pad(image: np.ndarray, new_h: int, new_w: int, off_h: int, off_w: int) -> np.ndarray:
    H, W, _ = image.shape
    x = torch.from_numpy(image.astype(np.float64)).permute(2, 0, 1)[None]
    if (new_h, new_w) != (H, W):
        x = F.interpolate(x, size=(new_h, new_w), mode="bilinear", align_corners=False, antialias=False)
    out = torch.zeros(1, 3, H, W, dtype=torch.float64)
    out[:, :, off_h:off_h + new_h, off_w:off_w + new_w] = x
    return out[0].permute(1, 2, 0).numpy()


def adjust_saturation(img: np.ndarray, scale: float) -> np.ndarray:
    r, g, b = img[..., 0], img[..., 1], img[..., 2]
    vmax, vmin = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    rng = vmax - vmin
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.where(vmax > 0, rng / vmax, 0.0)
        norm = 1.0 / (6.0 * rng)
        h = np.where(r == vmax, norm * (g - b), np.where(g == vmax, norm * (b - r) + 2.0 / 6.0, norm * (r - g) + 4.0 / 6.0))
    h = np.where(rng <= 0, 0.0, h)
    h = np.where(h < 0, h + 1.0, h)
    v = vmax
    s = np.clip(s * scale, 0.0, 1.0)
    c = s * v
    m = v - c
    dh = h * 6.0
    fm = dh.copy()
    fm = np.where(fm <= 0, fm + 2.0, fm)
    while (fm >= 2.0).any():
        fm = np.where(fm >= 2.0, fm - 2.0, fm)
    x = c * (1.0 - np.abs(fm - 1.0))
    cat = dh.astype(np.int64)
    z = np.zeros_like(c)
    rr = np.select([cat == 0, cat == 1, cat == 4, cat == 5], [c, x, x, c], z)
    gg = np.select([cat == 0, cat == 1, cat == 2, cat == 3], [x, c, c, x], z)
    bb = np.select([cat == 2, cat == 3, cat == 4, cat == 5], [x, c, c, x], z)
    return np.stack([rr + m, gg + m, bb + m], axis=-1)


def augment(image: np.ndarray, p: dict, b: int) -> np.ndarray:
    """One image [H,W,3] through downsizer -> contrast -> brightness -> [jpeg quality] -> saturation (fp64)."""
    x = downsize_with_pad(image, int(p["new_h"][b]), int(p["new_w"][b]), int(p["off_h"][b]), int(p["off_w"][b]))
    mean = x.mean(axis=(0, 1), keepdims=True)
    x = (x - mean) * float(p["contrast"][b]) + mean
    x = x + float(p["brightness"][b])
    if "jpeg_quality" in p:
        from . import jpeg_oracle
        x = jpeg_oracle.adjust_jpeg_quality(x, int(p["jpeg_quality"][b])).astype(np.float64)
    return adjust_saturation(x, float(p["saturation"][b]))


def adjust_boxes(bbox: np.ndarray, p: dict, H: int, W: int) -> np.ndarray:
    """pipeline.py:302-313 (quirks reproduced: x divided by the height factor, offsets added to w/h too)."""
    out = bbox.astype(np.float64).copy()
    for b in range(bbox.shape[0]):
        rh, rw = float(p["rand_val"][b, 0]), float(p["rand_val"][b, 1])
        out[b] = out[b] / np.array([rh, rw, rh, rw]) + np.array([p["off_h"][b] / H, p["off_w"][b] / W, p["off_h"][b] / H, p["off_w"][b] / W])
    return out
