"""TEST INFRASTRUCTURE (oracle) -- CPU restatement of the reference's frame ingest: aspect-preserving resize +
centred zero pad to the model resolution + channel order + CHW, uint8 in, uint8 out.  Never imported by the product.

Two paths exist in the reference and they use different resamplers:

* demo / live path, `LiveInferForDemo.load_one_frame` (test/live_infer_for_video.py:98-121):
  `PIL.Image.resize((new_w, new_h))` (pillow==10.4.0 in requirements.txt:36; default resample = BICUBIC) followed by
  `ImageOps.expand(border=(left, top, right, bottom), fill=0)`, `np.array`, HWC->CHW.
  Pillow's resampler (src/libImaging/Resample.c, third-party, not under /root/reference) is restated here:
  per axis, `precompute_coeffs` (double) -> `normalize_coeffs_8bpc` (22-bit fixed point, round half away from zero)
  -> horizontal pass to a uint8 image -> vertical pass, each `clip8((1 << 21) + sum(pixel * k) >> 22)`.
  PINNED: tests/test_frame_ingest.py compares it bit for bit with the Pillow installed here (12.2.0; the 8-bit
  resampler is unchanged since 10.4) on random images over many geometries, and tests/golden/frame_ingest.npz holds
  Pillow-generated vectors for the GPU box (tests/make_golden.py).

* benchmark path, `load_video_for_testing` / `load_video` (test/inference.py:538-562,
  test/live_infer_for_video.py:49-71): `cv2.resize(frame, (new_w, new_h))` (opencv-python==4.10.0.84,
  requirements.txt:33; default INTER_LINEAR), `cv2.copyMakeBorder(..., BORDER_CONSTANT, 0)`, `cvtColor(BGR2RGB)`,
  HWC->CHW.  OpenCV's 8-bit bilinear (modules/imgproc/src/resize.cpp: HResizeLinear / VResizeLinear with
  INTER_RESIZE_COEF_BITS = 11, generic path; the IPP path is not taken for 8-bit linear unless useIPP_NotExact) is
  restated from the published algorithm.  cv2 is not installed in this image and the reference holds no fixture for
  it: PARITY UNPINNED for this method.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c


def resize_geometry(width, height, resolution):
    """new size and (left, top, right, bottom) border, exactly the integer/float arithmetic of
    test/live_infer_for_video.py:108-119 (and test/inference.py:538-555)."""
    if width > height:
        new_w, new_h = resolution, int((height / width) * resolution)
    else:
        new_h, new_w = resolution, int((width / height) * resolution)
    left, right = (resolution - new_w) // 2, (resolution - new_w + 1) // 2
    top, bottom = (resolution - new_h) // 2, (resolution - new_h + 1) // 2
    return new_w, new_h, (left, top, right, bottom)


# ---- Pillow BICUBIC -----------------------------------------------------------------------------------
def _bicubic(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc for box (0, in_size), bicubic (support 2.0).
    Returns ksize, bounds int32 [out,2] = (xmin, count), kk int32 [out, ksize]."""
    scale = float(in_size) / out_size               # (double)(in1 - in0) / outSize
    filterscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        for x, w in enumerate(k):
            v = w * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w < 0 else int(0.5 + v)        # C (int) truncates toward zero
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)       # arithmetic shift, then the lookup's clamp


def _pil_pass(img, out_size, axis):
    """one resampling pass along `axis` (1 = horizontal, 0 = vertical) of a uint8 [h,w,c] image"""
    in_size = img.shape[axis]
    _, bounds, kk = pil_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        acc += np.tensordot(kk[xx, :n].astype(np.int64), src[xmin:xmin + n], axes=(0, 0))
        out[xx] = _clip8(acc)
    return np.moveaxis(out, 0, axis)


def pil_resize_bicubic(img_hwc, new_w, new_h):
    """PIL.Image.resize((new_w, new_h)) on an RGB uint8 image: identity sizes are a copy (Image.resize returns
    self.copy()), otherwise ImagingResample = horizontal pass (if widths differ) then vertical pass (if heights differ)."""
    h, w, _ = img_hwc.shape
    out = img_hwc
    if new_w != w:
        out = _pil_pass(out, new_w, 1)
    if new_h != h:
        out = _pil_pass(out, new_h, 0)
    return out.copy()


def demo_frame_to_canvas(img_hwc_rgb, resolution):
    """load_one_frame (test/live_infer_for_video.py:98-121): uint8 [h,w,3] RGB -> uint8 [3,S,S]"""
    h, w, _ = img_hwc_rgb.shape
    new_w, new_h, (left, top, _, _) = resize_geometry(w, h, resolution)
    canvas = np.zeros((resolution, resolution, 3), np.uint8)
    canvas[top:top + new_h, left:left + new_w] = pil_resize_bicubic(img_hwc_rgb, new_w, new_h)
    return np.ascontiguousarray(canvas.transpose(2, 0, 1))


# ---- OpenCV INTER_LINEAR, 8-bit -----------------------------------------------------------------------
_COEF_BITS = 11
_COEF_SCALE = 1 << _COEF_BITS


def _cv_round_short(v):
    """saturate_cast<short>(float): cvRound = round half to even (lrint), then saturate"""
    return int(max(-32768, min(32767, np.rint(np.float32(v)))))


def _cv_axis_tables(src_size, dst_size):
    """per output index: first source index, fixed-point pair (a0, a1); float arithmetic as in resize.cpp:
    scale = 1. / ((double)dst / src); f = (float)((d + 0.5) * scale - 0.5); s = cvFloor(f); f -= s."""
    scale = 1.0 / (float(dst_size) / src_size)
    idx = np.zeros(dst_size, np.int32)
    coef = np.zeros((dst_size, 2), np.int32)
    frac = np.zeros(dst_size, np.float32)
    for d in range(dst_size):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(math.floor(f))
        f = np.float32(f - np.float32(s))
        idx[d], frac[d] = s, f
    return idx, frac


def cv2_resize_linear(img_hwc, new_w, new_h):
    """cv2.resize(img, (new_w, new_h)) with the default INTER_LINEAR on uint8, generic fixed-point path."""
    h, w, c = img_hwc.shape
    if (new_w, new_h) == (w, h):
        return img_hwc.copy()
    sx, fx = _cv_axis_tables(w, new_w)
    sy, fy = _cv_axis_tables(h, new_h)
    # horizontal tables: at the borders the x fraction is zeroed and the index clamped (resize.cpp, the
    # "sx < 0" / "sx >= ssize.width - 1" branches); dx >= xmax reads a single pixel with weight ONE
    x0 = np.zeros(new_w, np.int64); x1 = np.zeros(new_w, np.int64)
    a0 = np.zeros(new_w, np.int64); a1 = np.zeros(new_w, np.int64)
    for d in range(new_w):
        s, f = int(sx[d]), np.float32(fx[d])
        if s < 0:
            s, f = 0, np.float32(0)
        if s >= w - 1:
            s, f = w - 1, np.float32(0)
        x0[d], x1[d] = s, min(s + 1, w - 1)
        a0[d] = _cv_round_short(np.float32(np.float32(1.0) - f) * _COEF_SCALE)
        a1[d] = _cv_round_short(f * _COEF_SCALE)
    src = img_hwc.astype(np.int64)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]          # [h, new_w, c] int
    out = np.empty((new_h, new_w, c), np.uint8)
    for d in range(new_h):
        s, f = int(sy[d]), np.float32(fy[d])
        b0 = _cv_round_short(np.float32(np.float32(1.0) - f) * _COEF_SCALE)
        b1 = _cv_round_short(f * _COEF_SCALE)
        r0 = min(max(s, 0), h - 1)                       # vertical: indices clamp, coefficients stay
        r1 = min(max(s + 1, 0), h - 1)
        v = (((b0 * (rows[r0] >> 4)) >> 16) + ((b1 * (rows[r1] >> 4)) >> 16) + 2) >> 2
        out[d] = np.clip(v, 0, 255).astype(np.uint8)
    return out


def benchmark_frame_to_canvas(frame_hwc_bgr, resolution):
    """the per-frame body of load_video_for_testing (test/inference.py:538-562): uint8 [h,w,3] BGR -> uint8 [3,S,S] RGB"""
    h, w, _ = frame_hwc_bgr.shape
    new_w, new_h, (left, top, _, _) = resize_geometry(w, h, resolution)
    canvas = np.zeros((resolution, resolution, 3), np.uint8)
    canvas[top:top + new_h, left:left + new_w] = cv2_resize_linear(frame_hwc_bgr, new_w, new_h)
    return np.ascontiguousarray(canvas[:, :, ::-1].transpose(2, 0, 1))


def sample_frame_indices(input_fps, frame_count, output_fps, max_num_frames=None, floor_total=False):
    """Which decoded frames load_video_for_testing keeps (test/inference.py:509-571; floor_total=True is load_video,
    test/live_infer_for_video.py:42-43,74-75): frame i is kept when the running clock `cur_time` (accumulated
    1/input_fps per decoded frame, in floating point, as the reference does) has reached the next i/output_fps."""
    video_duration = frame_count / input_fps
    output_fps = output_fps if output_fps > 0 else max_num_frames / video_duration
    total = math.floor(video_duration * output_fps) if floor_total else math.ceil(video_duration * output_fps)
    frame_sec = [i / output_fps for i in range(total)]
    keep, cur_time, frame_index = [], 0, 0
    for true_index in range(int(frame_count)):
        if frame_index < len(frame_sec) and cur_time >= frame_sec[frame_index]:
            keep.append(true_index)
            frame_index += 1
        if max_num_frames and len(keep) >= max_num_frames:
            break
        cur_time += 1 / input_fps
    return keep, output_fps, video_duration
