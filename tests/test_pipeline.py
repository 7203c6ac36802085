"""Input step: annotation packing on the host (CPU tests) and the GPU augmentation
kernels against the fp64 oracle restatement of the tf.image ops."""
import numpy as np
import pytest
import torch


def test_coco_records_and_padding_contract():
    from boosted_detr_amd import pipeline
    coco = {"images": [{"id": 7, "width": 200, "height": 100, "file_name": "a.jpg"}, {"id": 9, "width": 50, "height": 50}],
            "categories": [{"id": 1, "name": "person"}, {"id": 18, "name": "dog"}],
            "annotations": [{"image_id": 7, "bbox": [20, 10, 100, 50], "category_id": 18},
                            {"image_id": 7, "bbox": [0, 0, 200, 100], "category_id": 1}]}
    recs = pipeline.coco_records(coco)
    assert recs[0]["bbox"][0] == [0.1, 0.1, 0.5, 0.5] and recs[0]["category"] == [["dog"], ["person"]] and recs[1]["bbox"] == []
    batch = pipeline.pad_annotations(recs, max_objects=4)
    assert batch["category"].shape == (2, 4, 1) and batch["category"][0, :, 0].tolist() == ["dog", "person", "<PAD>", "<PAD>"]
    assert batch["bbox"].shape == (2, 4, 4) and (batch["bbox"][0, 2:] == -10).all() and (batch["bbox"][1] == -10).all()
    assert batch["num_objects"].tolist() == [2, 0] and batch["attribute"].shape == (2, 4, 1)


def test_box_adjustment_quirks_match_oracle():
    from boosted_detr_amd.pipeline import Augmentations
    from oracle import augment_oracle as AO
    aug = Augmentations(seed=3)
    p = aug.draw(4, 64, 96)
    assert (p["rand_val"] >= 1).all() and (p["off_h"] + p["new_h"] <= 64).all() and (p["off_w"] + p["new_w"] <= 96).all()
    bbox = np.random.default_rng(0).random((4, 5, 4)).astype(np.float32)
    bbox[:, 3:] = -10.0
    got = Augmentations.adjust_boxes(bbox, p, 64, 96)
    assert np.allclose(got, AO.adjust_boxes(bbox, p, 64, 96), atol=1e-6)


@pytest.mark.gpu
def test_gpu_augment_matches_oracle(cuda):
    from boosted_detr_amd.pipeline import Augmentations
    from oracle import augment_oracle as AO
    rng = np.random.default_rng(1)
    B, H, W = 4, 48, 80
    image = rng.random((B, H, W, 3), dtype=np.float32)
    aug = Augmentations(seed=5, jpeg_quality=False)
    p = aug.draw(B, H, W)
    quality = p.pop("jpeg_quality")                                 # first without the JPEG round trip: fp32 against fp64 to 2e-5
    p["rand_val"][0] = [1.0, 1.0]; p["new_h"][0], p["new_w"][0], p["off_h"][0], p["off_w"][0] = H, W, 0, 0     # identity geometry
    p["rand_val"][1] = [1.7, 1.3]; p["new_h"][1], p["new_w"][1] = int(np.float32(H) / np.float32(1.7)), int(np.float32(W) / np.float32(1.3))
    p["off_h"][1], p["off_w"][1] = 5, 11
    batch = {"image": image, "bbox": rng.random((B, 3, 4)).astype(np.float32)}
    out = aug.apply(batch, params=p)
    got = out["image"].cpu().numpy().astype(np.float64)
    for b in range(B):
        want = AO.augment(image[b], p, b)
        err = np.abs(got[b] - want).max()
        assert err < 2e-5, (b, err)
    assert np.allclose(out["bbox"], AO.adjust_boxes(batch["bbox"], p, H, W), atol=1e-6)
    # with the JPEG round trip (between brightness and saturation, pipeline.py:364-383): a last-bit difference of the fp32 chain in front
    # of it can move a pixel across a uint8 boundary, and the lossy codec spreads that over its 8x8 block - so this is statistical; the
    # codec itself is compared bit for bit in tests/test_jpeg_quality.py
    p["jpeg_quality"] = quality
    got = Augmentations(seed=5).apply(batch, params=p)["image"].cpu().numpy().astype(np.float64)
    for b in range(B):
        d = np.abs(got[b] - AO.augment(image[b], p, b))
        assert d.mean() < 0.5 / 255 and (d <= 4.0 / 255).mean() > 0.99, (b, d.mean(), d.max())
    # the generator form maps a stream of batches
    outs = list(Augmentations(seed=1).apply_image_augmentations([batch, batch]))
    assert len(outs) == 2 and tuple(outs[0]["image"].shape) == (B, H, W, 3)
