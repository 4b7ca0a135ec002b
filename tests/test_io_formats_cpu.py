"""CPU: environment-map loading (Radiance RGBE, flat and run-length encoded) and the image panels of the reference's saver."""
import numpy as np

from intrinsicavatar_amd import io_formats as IO


def _rle_hdr(path, img):
    """new-style RLE Radiance writer (test-side, to exercise the reader's RLE branch)."""
    H, W, _ = img.shape
    m = img.max(-1)
    mant, exp = np.frexp(m)
    scale = np.where(m > 1e-32, mant * 256.0 / np.maximum(m, 1e-38), 0.0)
    rgbe = np.zeros((H, W, 4), np.uint8)
    rgbe[..., :3] = np.clip(img * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(m > 1e-32, exp + 128, 0).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {H} +X {W}\n".encode())
        for y in range(H):
            f.write(bytes([2, 2, W >> 8, W & 255]))
            for c in range(4):
                row, x = rgbe[y, :, c], 0
                while x < W:
                    run = 1
                    while x + run < W and run < 127 and row[x + run] == row[x]:
                        run += 1
                    if run >= 4:
                        f.write(bytes([128 + run, int(row[x])])); x += run
                    else:
                        n = 1
                        while x + n < W and n < 128 and not (x + n + 3 < W and row[x + n] == row[x + n + 1] == row[x + n + 2] == row[x + n + 3]):
                            n += 1
                        f.write(bytes([n]) + row[x:x + n].tobytes()); x += n
    return rgbe


def test_radiance_hdr_roundtrip_flat_and_rle(tmp_path):
    rng = np.random.default_rng(0)
    img = (rng.random((24, 64, 3)) ** 4 * 50).astype(np.float32)
    img[5:9, 10:40] = 3.5                                  # runs
    img[0, 0] = 0.0
    IO.save_hdr(str(tmp_path / "flat.hdr"), img)
    a = IO.load_hdr(str(tmp_path / "flat.hdr"))
    _rle_hdr(str(tmp_path / "rle.hdr"), img)
    b = IO.load_hdr(str(tmp_path / "rle.hdr"))
    np.testing.assert_array_equal(a, b)
    assert (np.abs(a - img) <= img.max(-1, keepdims=True) / 128 + 1e-6).all()      # 8-bit mantissas, one exponent per pixel
    assert a[0, 0].max() == 0


def test_hdri_area_resize_and_panels(tmp_path):
    rng = np.random.default_rng(1)
    big = rng.random((2048, 4096, 3)).astype(np.float32)
    np.save(tmp_path / "env.npy", big)
    out = IO.load_hdri_2k(str(tmp_path / "env.npy"))
    assert out.shape == (1024, 2048, 3) and out.dtype == np.float32
    np.testing.assert_allclose(out, big.reshape(1024, 2, 2048, 2, 3).mean((1, 3)), rtol=1e-5)      # INTER_AREA at ratio 2 = 2x2 box
    rgb = rng.random((8, 12, 3)).astype(np.float32) * 1.4 - 0.2
    u8 = IO.rgb_image_u8(rgb)
    np.testing.assert_array_equal(u8, (rgb.clip(0, 1) * 255.0).astype(np.uint8))
    six = IO.rgb_image_u8(np.concatenate([rgb, rgb[..., :1]], -1))       # 4 channels -> two panels
    assert six.shape == (8, 24, 3) and (six[:, 12:, 1:] == 0).all()
    g = IO.grayscale_image_u8(rgb[..., 0], data_range=(0, 1))
    assert g.shape == (8, 12, 3) and (g[..., 0] == g[..., 2]).all()
    grid = IO.save_image_grid(str(tmp_path / "it0-test" / "0.png"),
                              [{"type": "rgb", "img": rgb, "kwargs": {"data_format": "HWC"}},
                               {"type": "grayscale", "img": rgb[..., 0], "kwargs": {"data_range": None, "cmap": "jet"}},
                               {"type": "grayscale", "img": rgb[..., 1], "kwargs": {}}])
    assert grid.shape == (8, 36, 3)
    from PIL import Image
    np.testing.assert_array_equal(np.asarray(Image.open(tmp_path / "it0-test" / "0.png")), grid)
