"""I/O formats either side of the render_step path, without OpenCV / imageio:

    load_hdr / load_hdri_2k      the test-time environment map: Radiance RGBE `.hdr` -> float32 [1024, 2048, 3]
                                 (datasets/animation.py:195-204: cv2.imread(ANYDEPTH | COLOR) -> RGB -> INTER_AREA resize to 2k)
    save_hdr                     flat RGBE writer (fixtures, round trips)
    rgb_image_u8 / grayscale_image_u8 / image_grid_u8 / save_image_grid
                                 the panels the reference's SaverMixin writes per validation / test image
                                 (utils/mixins.py:43-58, 87-122, 124-155), as RGB uint8 arrays / PNG files via PIL

`.exr` needs OpenEXR, which this image does not have: load_hdri raises for it.  Checkpoint key layout and SMPL kinematics are
covered in checkpoint-facing modules (fields.py, smpl.py)."""
import os
import re
from typing import List, Optional, Sequence

import numpy as np


# ----------------------------------------------------------------------------- Radiance RGBE
def _rgbe_to_float(rgbe: np.ndarray) -> np.ndarray:
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(1.0, e - 136), 0.0).astype(np.float32)          # 2^(e-128) / 256
    return rgbe[..., :3].astype(np.float32) * scale[..., None]


def load_hdr(path: str) -> np.ndarray:
    """Radiance `.hdr` / `.pic` (32-bit_rle_rgbe, -Y H +X W; flat or new-style run-length encoded scanlines) -> float32 [H,W,3] RGB."""
    data = open(path, "rb").read()
    if not data.startswith(b"#?"):
        raise ValueError(f"{path}: not a Radiance HDR file")
    head_end = data.index(b"\n\n")
    header = data[:head_end].decode("ascii", "replace")
    if "32-bit_rle_xyze" in header:
        raise NotImplementedError("XYZE-encoded Radiance files are not supported")
    rest = data[head_end + 2:]
    line_end = rest.index(b"\n")
    m = re.match(rb"-Y\s+(\d+)\s+\+X\s+(\d+)", rest[:line_end])
    if not m:
        raise NotImplementedError(f"{path}: only the standard -Y H +X W orientation is supported")
    H, W = int(m.group(1)), int(m.group(2))
    buf = np.frombuffer(rest[line_end + 1:], dtype=np.uint8)
    out = np.empty((H, W, 4), np.uint8)
    pos = 0
    if W < 8 or W > 0x7FFF or not (buf[0] == 2 and buf[1] == 2 and (int(buf[2]) << 8 | int(buf[3])) == W):
        out[:] = buf[: H * W * 4].reshape(H, W, 4)                                     # flat pixels
        return _rgbe_to_float(out)
    for y in range(H):
        if not (buf[pos] == 2 and buf[pos + 1] == 2 and (int(buf[pos + 2]) << 8 | int(buf[pos + 3])) == W):
            raise ValueError(f"{path}: bad scanline header at row {y}")
        pos += 4
        for c in range(4):
            x = 0
            while x < W:
                n = int(buf[pos]); pos += 1
                if n > 128:
                    n -= 128
                    out[y, x:x + n, c] = buf[pos]; pos += 1
                else:
                    out[y, x:x + n, c] = buf[pos:pos + n]; pos += n
                x += n
    return _rgbe_to_float(out)


def save_hdr(path: str, img: np.ndarray) -> None:
    """float32 [H,W,3] RGB -> flat (uncompressed) Radiance RGBE."""
    img = np.ascontiguousarray(img, np.float32)
    H, W, _ = img.shape
    m = img.max(-1)
    mant, exp = np.frexp(m)
    scale = np.where(m > 1e-32, mant * 256.0 / np.maximum(m, 1e-38), 0.0)
    rgbe = np.zeros((H, W, 4), np.uint8)
    rgbe[..., :3] = np.clip(img * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(m > 1e-32, exp + 128, 0).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {H} +X {W}\n".encode())
        f.write(rgbe.tobytes())


def area_resize(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.INTER_AREA for down-scaling: box average over the source footprint of every target pixel (exact for integer
    ratios, area-weighted otherwise), float32 [H,W,C]."""
    def axis(a, n_out, ax):
        n_in = a.shape[ax]
        if n_in == n_out:
            return a
        edges = np.linspace(0, n_in, n_out + 1)
        w = np.zeros((n_out, n_in), np.float64)
        for i in range(n_out):
            lo, hi = edges[i], edges[i + 1]
            j0, j1 = int(np.floor(lo)), int(np.ceil(hi))
            for j in range(j0, min(j1, n_in)):
                w[i, j] = min(hi, j + 1) - max(lo, j)
            w[i] /= (hi - lo)
        return np.moveaxis(np.tensordot(w, np.moveaxis(a, ax, 0), axes=(1, 0)), 0, ax)
    return axis(axis(img.astype(np.float64), out_h, 0), out_w, 1).astype(np.float32)


def load_hdri_2k(path: str) -> np.ndarray:
    """the test-time light of datasets/animation.py:195-204: environment map resized to [1024, 2048, 3] float32."""
    ext = os.path.splitext(path)[1].lower()
    if ext in (".hdr", ".pic"):
        img = load_hdr(path)
    elif ext == ".exr":
        raise NotImplementedError("OpenEXR is not available in this image: convert the environment map to Radiance .hdr")
    elif ext == ".npy":
        img = np.load(path).astype(np.float32)
    else:
        raise ValueError(f"unsupported environment map format: {ext}")
    return img if img.shape[:2] == (1024, 2048) else area_resize(img, 1024, 2048)


# ----------------------------------------------------------------------------- SaverMixin panels (RGB uint8)
def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def rgb_image_u8(img, data_format: str = "HWC", data_range=(0, 1)) -> np.ndarray:
    """utils/mixins.py:43-54 (channels beyond 3 become further panels to the right), RGB order."""
    img = _np(img)
    assert data_format in ("CHW", "HWC")
    if data_format == "CHW":
        img = img.transpose(1, 2, 0)
    img = img.clip(min=data_range[0], max=data_range[1])
    img = ((img - data_range[0]) / (data_range[1] - data_range[0]) * 255.0).astype(np.uint8)
    panels = [img[..., s:s + 3] for s in range(0, img.shape[-1], 3)]
    panels = [p if p.shape[-1] == 3 else np.concatenate([p, np.zeros(p.shape[:2] + (3 - p.shape[2],), p.dtype)], -1) for p in panels]
    return np.concatenate(panels, axis=1)


def grayscale_image_u8(img, data_range=None, cmap: Optional[str] = None) -> np.ndarray:
    """utils/mixins.py:87-118 for cmap None / 'jet' / 'magma' (matplotlib colour maps; 'jet' here is matplotlib's, OpenCV's
    COLORMAP_JET differs by rounding)."""
    img = np.nan_to_num(_np(img)).astype(np.float64)
    if data_range is None:
        img = (img - img.min()) / max(img.max() - img.min(), 1e-12)
    else:
        img = (img.clip(data_range[0], data_range[1]) - data_range[0]) / (data_range[1] - data_range[0])
    assert cmap in (None, "jet", "magma")
    if cmap is None:
        return np.repeat((img * 255.0).astype(np.uint8)[..., None], 3, axis=2)
    from matplotlib import colormaps
    if cmap == "magma":
        img = 1.0 - img
    lut = colormaps[cmap](np.linspace(0, 1, 256))[:, :3]
    a = np.floor(img * 255.0)
    b = (a + 1).clip(max=255.0)
    f = img * 255.0 - a
    a, b = a.astype(np.uint16).clip(0, 255), b.astype(np.uint16).clip(0, 255)
    return ((lut[a] + (lut[b] - lut[a]) * f[..., None]) * 255.0).astype(np.uint8)


def image_grid_u8(imgs: Sequence) -> np.ndarray:
    """utils/mixins.py:124-145: a row (list of {'type','img','kwargs'}) or a list of rows -> one RGB uint8 image."""
    if isinstance(imgs[0], list):
        return np.concatenate([image_grid_u8(row) for row in imgs], axis=0)
    cols: List[np.ndarray] = []
    for col in imgs:
        kw = dict(col.get("kwargs", {}))
        if col["type"] == "rgb":
            cols.append(rgb_image_u8(col["img"], **kw))
        elif col["type"] == "grayscale":
            cols.append(grayscale_image_u8(col["img"], **kw))
        else:
            raise NotImplementedError(f"panel type {col['type']!r} (only 'rgb' and 'grayscale' are written on the render_step path)")
    return np.concatenate(cols, axis=1)


def save_image_grid(path: str, imgs: Sequence) -> np.ndarray:
    from PIL import Image
    img = image_grid_u8(imgs)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(img).save(path)
    return img
