"""On-disk formats: cameras_sphere npz -> intrinsics / pose, PNG + mask / RGBA decoding with the reference's
compositing and channel order, wavelet-feature normalisation. The reference's loader needs OpenCV (absent), so these pin the
restatement by construction: projection matrices built from known K, R, c must decompose back to them."""
import os

import numpy as np
import pytest
import torch

from vdn_train import dataset, synth


def _K(f=1111.0, w=80, h=60, skew=0.0):
    return np.array([[f, skew, (w - 1) / 2.0], [0, 1.03 * f, (h - 1) / 2.0], [0, 0, 1.0]])


@pytest.mark.parametrize("scale", [1.0, -2.5, 1e-3])
def test_projection_matrix_decomposition_recovers_camera(scale):
    cams = synth.make_cameras(3)
    K = _K(skew=0.7)
    for c2w in cams[:6]:
        w2c = np.linalg.inv(c2w)
        P = scale * (K @ w2c[:3, :4])
        intr, pose = dataset.load_K_Rt_from_P(P)
        np.testing.assert_allclose(intr[:3, :3], K, rtol=1e-9, atol=1e-7)
        np.testing.assert_allclose(pose, c2w, rtol=0, atol=2e-6)            # pose is float32 like the reference's
        assert intr.shape == (4, 4) and pose.dtype == np.float32 and abs(np.linalg.det(pose[:3, :3]) - 1) < 1e-5


def test_opencv_restatement_properties():
    """oracle/opencv_decompose.py restates cv2.decomposeProjectionMatrix (OpenCV 4.5.2, absent here): M = K R, K upper-triangular
    with its first two diagonal entries positive, R a proper rotation, the returned position in P's null space - for P of either
    sign and type, and for left blocks that need each of the routine's three 180-degree fixes."""
    from oracle import opencv_decompose as cvd
    rng = np.random.default_rng(11)
    cams = synth.make_cameras(3)
    cases = [s * (_K(skew=0.7) @ np.linalg.inv(c)[:3, :4]) for c in cams[:4] for s in (1.0, -2.5)]
    for flip in (np.diag([-1.0, -1.0, 1.0]), np.diag([-1.0, 1.0, -1.0]), np.diag([1.0, -1.0, -1.0])):       # K with negative entries
        cases.append((_K() @ flip) @ np.linalg.inv(cams[1])[:3, :4])
    cases += [rng.standard_normal((3, 4)) for _ in range(8)]
    for P in cases:
        for dt, tol in ((np.float64, 1e-12), (np.float32, 2e-6)):
            K, R, t = cvd.decompose_projection_matrix(P.astype(dt))
            assert K.dtype == R.dtype == t.dtype == dt and t.shape == (4, 1)
            scale = np.abs(P[:, :3]).max()
            assert np.abs(K.astype(np.float64) @ R.astype(np.float64) - P[:, :3]).max() < tol * 10 * scale
            assert K[1, 0] == 0 and K[2, 0] == 0 and K[2, 1] == 0 and K[0, 0] > 0 and K[1, 1] > 0
            assert np.abs(R.astype(np.float64) @ R.astype(np.float64).T - np.eye(3)).max() < tol * 10
            assert abs(np.linalg.det(R.astype(np.float64)) - 1.0) < tol * 10
            assert np.sign(K[2, 2]) == np.sign(np.linalg.det(P[:, :3]))          # the last diagonal entry keeps det(M)'s sign
            assert np.abs(P @ t.astype(np.float64)).max() < tol * 10 * np.abs(P).max()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_load_K_Rt_from_P_equals_the_opencv_route(dtype):
    """dataset.py:14-35 = cv.decomposeProjectionMatrix + IDR's post-processing, on the restated OpenCV routine, against this
    package's load_K_Rt_from_P (scipy RQ + explicit signs) on what the reference feeds it: P = (world_mat @ scale_mat)[:3, :4] of
    DTU-style cameras (dataset.py:84-90: float32; positive scale, skewed and unskewed K, off-centre principal points)."""
    from oracle import opencv_decompose as cvd
    cams = synth.make_cameras(5)
    scale_mat = np.diag([1.7, 1.7, 1.7, 1.0])
    scale_mat[:3, 3] = [0.2, -0.1, 0.05]
    n = 0
    for c2w in cams[:8]:
        for K in (_K(), _K(skew=0.7), _K(f=2892.3, w=1600, h=1200), _K(f=20.0, w=16, h=12)):
            world = np.eye(4)
            world[:3, :4] = K @ np.linalg.inv(c2w)[:3, :4]
            P = (world @ scale_mat)[:3, :4].astype(dtype)
            ref_intr, ref_pose = cvd.load_K_Rt_from_P_reference(P)
            intr, pose = dataset.load_K_Rt_from_P(P)
            tol = 1e-9 if dtype == np.float64 else 3e-6              # (float32: OpenCV hands K, R, t back rounded to P's type)
            assert np.abs(intr - ref_intr).max() <= tol * np.abs(ref_intr).max()
            assert np.abs(pose.astype(np.float64) - ref_pose.astype(np.float64)).max() <= max(tol, 2e-6)
            assert intr.dtype == ref_intr.dtype == np.float64 and pose.dtype == ref_pose.dtype == np.float32
            n += 1
    assert n == 32


def _write_scene(root, rgba, with_depth, n=3, H=12, W=16):
    from PIL import Image
    rng = np.random.default_rng(5)
    os.makedirs(os.path.join(root, "image", "mask"), exist_ok=True)
    os.makedirs(os.path.join(root, "image", "wavelet_feats", "0"), exist_ok=True)
    cams = synth.make_cameras(1)[:n]
    names = ["%03d" % i for i in range(n)]
    K4 = np.eye(4)
    K4[:3, :3] = _K(f=20.0, w=W, h=H)
    scale_mat = np.diag([1.7, 1.7, 1.7, 1.0])
    scale_mat[:3, 3] = [0.2, -0.1, 0.05]
    world = [K4 @ np.linalg.inv(c) @ np.linalg.inv(scale_mat) for c in cams]      # so that world @ scale = K @ w2c
    dataset.write_cameras_npz(os.path.join(root, "cameras_sphere.npz"), names, world, [scale_mat] * n)
    raw = {}
    for nm in names:
        rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        m = (rng.random((H, W)) > 0.4).astype(np.uint8) * 255
        if rgba:
            Image.fromarray(np.concatenate([rgb, m[:, :, None]], 2), "RGBA").save(os.path.join(root, "image", nm + ".png"))
        else:
            Image.fromarray(rgb, "RGB").save(os.path.join(root, "image", nm + ".png"))
            Image.fromarray(np.repeat(m[:, :, None], 3, 2), "RGB").save(os.path.join(root, "image", "mask", nm + ".png"))
        f = rng.normal(0.3, 2.0, (1, 96, H // 2, W // 2)).astype(np.float32)
        if with_depth:
            np.save(os.path.join(root, "image", "wavelet_feats", "0", nm + ".npy"), f)
        raw[nm] = (rgb, m, f)
    return names, cams, K4, raw


@pytest.mark.parametrize("rgba", [False, True])
def test_scene_loader_matches_reference_semantics(tmp_path, rgba):
    names, cams, K4, raw = _write_scene(str(tmp_path), rgba, with_depth=True)
    sc = dataset.SceneData(str(tmp_path), with_depth=True)
    assert sc.n_images == 3 and (sc.H, sc.W) == (12, 16) and sc.names == names
    np.testing.assert_allclose(sc.pose_all, np.stack(cams), atol=3e-6)
    np.testing.assert_allclose(sc.intrinsics_all[1], K4, rtol=1e-5, atol=1e-4)
    assert abs(sc.focal - 20.0) < 1e-4
    for i, nm in enumerate(names):
        rgb, m, _ = raw[nm]
        a = (m / 255.0)[:, :, None]
        want = (rgb[:, :, ::-1] / 255.0) * a + (1 - a)               # BGR like cv.imread, composited on white (poses.py:117-127)
        np.testing.assert_allclose(sc.images[i], want, atol=1e-6)
        np.testing.assert_allclose(sc.masks[i][..., :1], a, atol=1e-6)
    assert sc.masks.shape[-1] == (1 if rgba else 3)
    # wavelet features: global statistics over the whole stack, sigmoid, bilinear x2 (poses.py:133-146)
    stack = np.stack([raw[nm][2][0] for nm in names])
    z = torch.sigmoid(torch.from_numpy((stack - stack.mean()) / stack.std()))
    want = torch.nn.Upsample(size=(12, 16), mode="bilinear")(z).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(sc.depth_feats, want, atol=1e-6)
    assert sc.depth_feats.shape == (3, 12, 16, 96)
    np.testing.assert_allclose(sc.object_bbox_min, [-1.01] * 3, atol=1e-6)
    assert sc.depth_from_sdf_path(2).endswith(os.path.join("image", "depth_from_sdf", "sdf_002.npy"))
    gen = sc.rays_generator(device="cpu")                             # tensors only; launching needs the GPU
    assert gen.images.shape == (3, 12, 16, 3) and gen.C == 96 and gen.intrin_inv.shape == (3, 3, 3)


def test_scene_loader_errors(tmp_path):
    with pytest.raises(FileNotFoundError):
        dataset.SceneData(str(tmp_path))


def test_image_metrics_formula():
    from vdn_train import validate
    rng = np.random.default_rng(0)
    img, gt = rng.random((6, 5, 3)).astype(np.float32), rng.random((6, 5, 3)).astype(np.float32)
    mask = (rng.random((6, 5, 1)) > 0.5).astype(np.float32)
    l1, psnr = validate.image_metrics(img, gt, mask)
    ms = mask.sum() + 1e-5
    assert abs(l1 - np.abs((img - gt) * mask).sum() / ms) < 1e-6
    assert abs(psnr - 20 * np.log10(1 / np.sqrt((((img - gt) ** 2) * mask).sum() / (ms * 3)))) < 1e-5


@pytest.mark.parametrize("mode", ["1", "P", "L", "RGBA"])
def test_mask_files_decode_like_imread_color(tmp_path, mode):
    """cv.imread(mask) (poses.py:125) always yields 8-bit 3-channel: 1-bit, palette, grey and RGBA mask files must all
    come out as {0, 255} x 3 (a 1-bit mask read raw would be {0, 1} and composite the object almost entirely to white)."""
    from PIL import Image
    rng = np.random.default_rng(2)
    m = (rng.random((9, 7)) > 0.5)
    base = Image.fromarray((m * 255).astype(np.uint8), "L")
    im = {"1": base.convert("1"), "P": base.convert("P"), "L": base, "RGBA": base.convert("RGBA")}[mode]
    path = os.path.join(tmp_path, "mask_%s.png" % mode)
    im.save(path)
    a = dataset._read_png(path, unchanged=False)
    assert a.shape == (9, 7, 3)
    np.testing.assert_array_equal(a, np.repeat((m * 255.0)[:, :, None], 3, axis=2))


def test_palette_image_is_expanded_like_imread_unchanged(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 4, (6, 5, 3), dtype=np.uint8) * 80
    path = os.path.join(tmp_path, "pal.png")
    Image.fromarray(rgb, "RGB").convert("P", palette=Image.ADAPTIVE, colors=64).save(path)
    a = dataset._read_png(path)                      # cv.imread(path, -1) expands palettes to BGR
    np.testing.assert_array_equal(a, rgb[:, :, ::-1].astype(np.float64))


def _rays_fx():
    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rays.npz")))


def test_image_compositing_and_feature_normalisation_equal_the_reference_constructor():
    """tests/golden/rays.npz holds what the REFERENCE's RaysGenerator.__init__ (poses.py:96-152, run by make_golden.py with a cv2
    stub whose imread hands back the arrays below) keeps as images / masks / depth_feats: bit for bit."""
    fx = _rays_fx()
    img, msk = dataset.composite_on_white(fx["bgra"])                            # poses.py:117-122
    assert np.array_equal(img, fx["rgba__images"]) and np.array_equal(msk, fx["rgba__masks"])
    img, msk = dataset.composite_on_white(fx["bgr"], fx["mask_files"])           # poses.py:123-127
    assert np.array_equal(img, fx["rgbmask__images"]) and np.array_equal(msk, fx["rgbmask__masks"])
    stack = np.stack([np.squeeze(f) for f in fx["feat_files"]])                  # poses.py:135
    feats = dataset.normalise_depth_feats(stack, img.shape[1:3]).numpy()
    assert feats.shape == fx["rgba__depth_feats"].shape
    assert np.abs(feats - fx["rgba__depth_feats"]).max() <= 1e-7                 # (sigmoid + bilinear: same torch ops)


@pytest.mark.parametrize("rgba", [False, True])
def test_scene_files_decode_to_the_reference_constructor_arrays(tmp_path, rgba):
    """The same arrays written as real PNG / npy files (RGB order on disk, as any PNG holds them) and read back through
    SceneData: the decoder's BGR flip and 8-bit handling land on what cv.imread would have handed the reference."""
    from PIL import Image
    fx = _rays_fx()
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "image", "mask"))
    os.makedirs(os.path.join(root, "image", "wavelet_feats", "0"))
    n = fx["bgr"].shape[0]
    names = ["%03d" % i for i in range(n)]
    K4 = fx["intrinsics_all"][0].astype(np.float64)
    dataset.write_cameras_npz(os.path.join(root, "cameras_sphere.npz"), names, [K4 @ np.linalg.inv(c.astype(np.float64)) for c in fx["pose_all"]],
                              [np.eye(4)] * n)
    for i, nm in enumerate(names):
        if rgba:
            a = fx["bgra"][i]
            Image.fromarray(np.concatenate([a[:, :, 2::-1], a[:, :, 3:]], 2), "RGBA").save(os.path.join(root, "image", nm + ".png"))
        else:
            Image.fromarray(np.ascontiguousarray(fx["bgr"][i][:, :, ::-1]), "RGB").save(os.path.join(root, "image", nm + ".png"))
            Image.fromarray(fx["mask_files"][i], "RGB").save(os.path.join(root, "image", "mask", nm + ".png"))
        np.save(os.path.join(root, "image", "wavelet_feats", "0", nm + ".npy"), fx["feat_files"][i])
    sc = dataset.SceneData(root, with_depth=rgba)
    tag = "rgba" if rgba else "rgbmask"
    assert np.array_equal(sc.images, fx[tag + "__images"]) and np.array_equal(sc.masks, fx[tag + "__masks"])
    if rgba:
        assert np.abs(sc.depth_feats - fx["rgba__depth_feats"]).max() <= 1e-7
    # cameras: K [R|t] written, decomposed back (load_K_Rt_from_P against OpenCV's restated routine: test_load_K_Rt_from_P_equals_the_opencv_route)
    np.testing.assert_allclose(sc.pose_all, fx["pose_all"], atol=3e-6)
    np.testing.assert_allclose(sc.intrinsics_all[:, :3, :3], fx["intrinsics_all"][:, :3, :3], rtol=1e-5, atol=1e-4)
