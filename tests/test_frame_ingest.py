"""Frame ingest (resize + pad + channel order): the oracle's restatement of Pillow's 8-bit bicubic resampler is
pinned bit for bit against Pillow itself (when installed) and against the Pillow-generated golden vectors; the
OpenCV bilinear restatement is unpinned (cv2 is not in this image) and only sanity-checked."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import frame_ingest as fi  # noqa: E402
from make_golden import INGEST_CASES, ingest_frame  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "frame_ingest.npz")


def test_geometry_rule():
    assert fi.resize_geometry(1280, 720, 384) == (384, 216, (0, 84, 0, 84))
    assert fi.resize_geometry(720, 1280, 384) == (216, 384, (84, 0, 84, 0))
    assert fi.resize_geometry(500, 385, 384) == (384, 295, (0, 44, 0, 45))      # odd leftover: extra row at the bottom
    assert fi.resize_geometry(640, 640, 336) == (336, 336, (0, 0, 0, 0))


def test_pillow_restatement_matches_golden_vectors():
    gold = np.load(GOLD)
    for i, (S, h, w) in enumerate(INGEST_CASES):
        got = fi.demo_frame_to_canvas(ingest_frame(i, h, w), S)
        assert got.dtype == np.uint8 and np.array_equal(got, gold[f"canvas_{i}"]), (i, S, h, w)


def test_pillow_restatement_matches_installed_pillow():
    PIL = pytest.importorskip("PIL")
    from PIL import Image, ImageOps
    rng = np.random.default_rng(3)
    geoms = [(720, 1280), (1280, 720), (480, 640), (100, 60), (37, 91), (385, 383), (2, 5), (384, 384), (1080, 1920)]
    for S in (384, 336):
        for h, w in geoms:
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            new_w, new_h, border = fi.resize_geometry(w, h, S)
            ref = np.array(ImageOps.expand(Image.fromarray(img).resize((new_w, new_h)), border=border, fill=(0, 0, 0)))
            assert np.array_equal(fi.demo_frame_to_canvas(img, S), ref.transpose(2, 0, 1)), (S, h, w, PIL.__version__)
    # smooth content exercises the negative bicubic lobes and the clip at 0 / 255
    yy, xx = np.mgrid[0:300, 0:500]
    img = np.stack([(xx % 256), (255 * (yy > 150)), ((xx + yy) % 7 == 0) * 255], -1).astype(np.uint8)
    new_w, new_h, border = fi.resize_geometry(500, 300, 384)
    ref = np.array(ImageOps.expand(Image.fromarray(img).resize((new_w, new_h)), border=border, fill=(0, 0, 0)))
    assert np.array_equal(fi.demo_frame_to_canvas(img, 384), ref.transpose(2, 0, 1))


def test_opencv_restatement_is_a_bilinear_resize():
    """unpinned method: it must at least agree with float bilinear (half-pixel centres) to within fixed-point rounding,
    copy same-size input, swap B and R, and pad with zeros"""
    rng = np.random.default_rng(5)
    for h, w in [(90, 160), (160, 90), (50, 50), (300, 200)]:
        bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        S = 64
        got = fi.benchmark_frame_to_canvas(bgr, S)
        new_w, new_h, (left, top, _, _) = fi.resize_geometry(w, h, S)
        x = torch.from_numpy(bgr[:, :, ::-1].copy()).permute(2, 0, 1)[None].float()
        ref = torch.nn.functional.interpolate(x, size=(new_h, new_w), mode="bilinear", align_corners=False)[0].numpy()
        inner = got[:, top:top + new_h, left:left + new_w].astype(np.float64)
        assert np.abs(inner - ref).max() <= 1.0 + 1e-6
        mask = np.ones((S, S), bool)
        mask[top:top + new_h, left:left + new_w] = False
        assert got[:, mask].max(initial=0) == 0
    same = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    assert np.array_equal(fi.benchmark_frame_to_canvas(same, 64), same[:, :, ::-1].transpose(2, 0, 1))
