"""ORACLE (test infrastructure, CPU, numpy): `LetterBox` - resize with unchanged aspect ratio + constant padding - as the
predictor applies it to every frame before the model (ultralytics/data/augment.py:1544-1700, called from
engine/predictor.py:151-173 `pre_transform`).

Two parts:
  * the geometry (scale ratio, rounded unpadded size, padding split, `round(d -/+ 0.1)` borders) restates
    LetterBox.__call__ line by line (augment.py:1640-1682); it is PINNED: oracle/gen_golden.py runs the reference's own
    class on procedural frames and stores inputs / outputs in tests/golden/letterbox.npz;
  * `cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR)` on uint8 (augment.py:1664).  OpenCV (opencv-python >= 4.6.0
    in the reference's requirements) is NOT installed in this image, so its arithmetic is restated here from the published
    algorithm (OpenCV 4.x modules/imgproc/src/resize.cpp: resizeGeneric_ with HResizeLinear / VResizeLinear<uchar, int,
    short, ...>, fixed point with INTER_RESIZE_COEF_BITS = 11; the IPP path is not taken for 8-bit linear unless
    `useIPP_NotExact`).  **Parity of this part is unpinned against the real cv2**: the fixtures were produced with THIS
    function plugged into the reference's class in place of cv2.resize; its known answers (identity, constants, exact 2x
    averaging) are in tests/test_letterbox.py.
"""

from __future__ import annotations

import numpy as np

INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


def _coeffs(dsize: int, ssize: int):
    """Source index and the two fixed-point weights of every destination index (resize.cpp, resizeGeneric_ set-up loop):
    f = (d + 0.5) * scale - 0.5 in double then float; s = floor(f); f -= s; weights saturate_cast<short>(w * 2048)."""
    scale = 1.0 / (float(dsize) / float(ssize))
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _fix(w: np.ndarray) -> np.ndarray:
    """saturate_cast<short>(float * 2048): round half to even (cvRound = lrint), clamp to int16."""
    return np.clip(np.rint(w.astype(np.float32) * np.float32(INTER_RESIZE_COEF_SCALE)), -32768, 32767).astype(np.int64)


def cv2_resize_linear_u8(img: np.ndarray, dsize) -> np.ndarray:
    """cv2.resize(img, dsize=(new_w, new_h), interpolation=cv2.INTER_LINEAR) for uint8 (h, w, c) images."""
    assert img.dtype == np.uint8 and img.ndim == 3
    sh, sw, _ = img.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    # horizontal: at the borders the x set-up pins the index AND zeroes the fraction (resize.cpp: "if (sx < 0) fx = 0, sx = 0",
    # "if (sx >= ssize.width - 1) fx = 0, sx = ssize.width - 1")
    sx, fx = _coeffs(dw, sw)
    lo, hi = sx < 0, sx >= sw - 1
    fx = np.where(lo | hi, np.float32(0), fx).astype(np.float32)
    sx = np.where(lo, 0, np.where(hi, sw - 1, sx))
    a0, a1 = _fix(np.float32(1) - fx), _fix(fx)
    sx1 = np.minimum(sx + 1, sw - 1)
    src = img.astype(np.int64)
    rows = src[:, sx, :] * a0[None, :, None] + src[:, sx1, :] * a1[None, :, None]  # (sh, dw, c) int, <= 255 * 2048
    # vertical: the fraction is kept, the two source rows are clipped into the image (clip(sy + k, 0, ssize.height))
    sy, fy = _coeffs(dh, sh)
    b0, b1 = _fix(np.float32(1) - fy), _fix(fy)
    y0 = np.clip(sy, 0, sh - 1)
    y1 = np.clip(sy + 1, 0, sh - 1)
    s0, s1 = rows[y0], rows[y1]  # (dh, dw, c)
    out = (((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox_geometry(shape, new_shape=(640, 640), auto=False, scale_fill=False, scaleup=True, center=True, stride=32):
    """(new_unpad (w, h), (top, bottom, left, right), ratio) of LetterBox.__call__ (augment.py:1640-1668)."""
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    ratio = r, r
    new_unpad = round(shape[1] * r), round(shape[0] * r)
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    elif scale_fill:
        dw, dh = 0.0, 0.0
        new_unpad = (new_shape[1], new_shape[0])
        ratio = new_shape[1] / shape[1], new_shape[0] / shape[0]
    if center:
        dw /= 2
        dh /= 2
    top, bottom = round(dh - 0.1) if center else 0, round(dh + 0.1)
    left, right = round(dw - 0.1) if center else 0, round(dw + 0.1)
    return (int(new_unpad[0]), int(new_unpad[1])), (int(top), int(bottom), int(left), int(right)), ratio


def letterbox(img: np.ndarray, new_shape=(640, 640), auto=False, scale_fill=False, scaleup=True, center=True, stride=32,
              padding_value=114) -> np.ndarray:
    """LetterBox(...)(image=img) for a uint8 (h, w, 3) frame (augment.py:1617-1682)."""
    shape = img.shape[:2]
    new_unpad, (top, bottom, left, right), _ = letterbox_geometry(shape, new_shape, auto, scale_fill, scaleup, center, stride)
    if shape[::-1] != new_unpad:
        img = cv2_resize_linear_u8(img, new_unpad)
    h, w, c = img.shape
    out = np.full((h + top + bottom, w + left + right, c), padding_value, dtype=img.dtype)
    out[top:top + h, left:left + w] = img
    return out
