"""Golden vectors of the JPEG quality round trip (tf.image.adjust_jpeg_quality's codec part), made with a REAL libjpeg-turbo: Pillow's
encoder (quality q, 4:2:0 chroma, no Huffman optimisation) and decoder (libjpeg defaults: slow integer IDCT, fancy up-sampling) - the
same library family TensorFlow bundles.  Run here (Pillow is installed); writes tests/golden/jpeg_quality.npz (uint8 inputs, qualities,
uint8 outputs).  oracle/jpeg_oracle.py must reproduce every vector bit for bit (tests/test_jpeg_quality.py)."""
import io
import os

import numpy as np
from PIL import Image

rng = np.random.default_rng(20261004)
out = {}
cases = [(16, 16, 70), (40, 33, 75), (24, 24, 83), (17, 23, 90), (64, 48, 99), (2, 3, 71), (1, 1, 95), (33, 40, 100), (31, 16, 88), (9, 7, 70)]
for i, (H, W, q) in enumerate(cases):
    if i % 2:
        img = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    else:
        yy, xx = np.mgrid[0:H, 0:W]
        img = np.clip(np.stack([128 + 100 * np.sin(xx / 7.0 + yy / 11.0), 128 + 90 * np.cos(xx / 5.0), 128 + 80 * np.sin(yy / 3.0)], -1)
                      + rng.normal(0, 12, (H, W, 3)), 0, 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="JPEG", quality=q, subsampling=2, optimize=False)
    buf.seek(0)
    out[f"in{i}"], out[f"q{i}"], out[f"out{i}"] = img, np.int32(q), np.asarray(Image.open(buf).convert("RGB"))
out["n"] = np.int32(len(cases))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "jpeg_quality.npz"), **out)
print("wrote", len(cases), "vectors")
