"""tf.image.random_jpeg_quality (reference pipeline.py:319-325).  The codec is a third-party dependency (libjpeg-turbo inside TensorFlow):
the oracle restates its lossy stages and is PINNED bit for bit against a real libjpeg-turbo - by committed golden vectors
(tests/golden/jpeg_quality.npz, made with Pillow by tests/golden/make_jpeg_golden.py) and, where Pillow is importable, live on random
and ragged images; the HIP kernels are compared bit for bit against the oracle."""
import io
import os

import numpy as np
import pytest

SIZES = [(16, 16), (40, 33), (24, 24), (17, 23), (64, 48), (2, 3), (1, 1), (33, 100), (31, 16), (9, 7), (96, 80)]


def test_oracle_reproduces_the_libjpeg_golden_vectors():
    from oracle import jpeg_oracle as J
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "jpeg_quality.npz"))
    for i in range(int(g["n"])):
        got = J.jpeg_quality_roundtrip_u8(g[f"in{i}"], int(g[f"q{i}"]))
        assert np.array_equal(got, g[f"out{i}"]), (i, g[f"in{i}"].shape, int(g[f"q{i}"]))


def test_oracle_matches_a_live_libjpeg_turbo():
    Image = pytest.importorskip("PIL.Image")
    from oracle import jpeg_oracle as J
    rng = np.random.default_rng(3)
    for (H, W) in SIZES:
        for q in (70, 77, 85, 92, 99, 100, 35):
            img = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
            buf = io.BytesIO()
            Image.fromarray(img).save(buf, format="JPEG", quality=q, subsampling=2, optimize=False)
            buf.seek(0)
            want = np.asarray(Image.open(buf).convert("RGB"))
            assert np.array_equal(J.jpeg_quality_roundtrip_u8(img, q), want), (H, W, q)


def test_float_conversion_saturates_like_convert_image_dtype():
    from oracle import jpeg_oracle as J
    x = np.array([[[-0.2, 0.0, 0.001], [0.5, 1.0, 1.7]]], np.float32)            # below 0, exact ends, above 1
    u8 = np.clip(np.floor(x.astype(np.float64) * 255.5), 0, 255).astype(np.uint8)
    assert u8.tolist() == [[[0, 0, 0], [127, 255, 255]]]
    out = J.adjust_jpeg_quality(x, 100)
    assert out.dtype == np.float32 and out.min() >= 0.0 and out.max() <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", SIZES)
def test_gpu_jpeg_quality_is_bit_exact_against_the_oracle(cuda, H, W):
    import torch
    from boosted_detr_amd import kernels as K
    from oracle import jpeg_oracle as J
    rng = np.random.default_rng(H * 1000 + W)
    B = 3
    u8 = rng.integers(0, 256, (B, H, W, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    u8[1] = np.clip(np.stack([128 + 100 * np.sin(xx / 7.0 + yy / 11.0), 128 + 90 * np.cos(xx / 5.0), 128 + 80 * np.sin(yy / 3.0)], -1)
                    + rng.normal(0, 12, (H, W, 3)), 0, 255).astype(np.uint8)
    image = ((u8.astype(np.float64) + 0.5) / 255.5).astype(np.float32)          # floor(x * 255.5) recovers u8 robustly
    quality = np.array([70, 86, 99], np.int32)
    got = K.jpeg_quality(torch.from_numpy(image).to(cuda), torch.from_numpy(quality).to(cuda)).cpu().numpy()
    for b in range(B):
        want = J.jpeg_quality_roundtrip_u8(u8[b], int(quality[b]))
        got_u8 = np.rint(got[b].astype(np.float64) * 255.0).astype(np.int64)
        assert np.array_equal(got_u8, want.astype(np.int64)), (b, np.abs(got_u8 - want).max())
        assert np.array_equal(got[b], (want.astype(np.float32) / np.float32(255.0)))                 # and the float conversion itself


@pytest.mark.gpu
def test_gpu_jpeg_quality_saturates_out_of_range_input(cuda):
    import torch
    from boosted_detr_amd import kernels as K
    from oracle import jpeg_oracle as J
    rng = np.random.default_rng(9)
    image = (rng.random((2, 32, 48, 3)) * 1.6 - 0.3).astype(np.float32)         # brightness / contrast leave [0, 1]
    q = np.array([75, 95], np.int32)
    got = K.jpeg_quality(torch.from_numpy(image).to(cuda), torch.from_numpy(q).to(cuda)).cpu().numpy()
    for b in range(2):
        assert np.array_equal(got[b], J.adjust_jpeg_quality(image[b], int(q[b])))
