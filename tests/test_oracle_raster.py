"""Known-answer tests that pin oracle/gsplat_raster.py (the restatement of gsplat 1.4.0's rasterisation; the wheel is absent, so
these analytic checks are the anchor) and the host-side camera-path / video code of vist3a_amd/misc/image_io.py."""
import math
import struct

import numpy as np
import torch

from oracle import gsplat_raster as G


def test_sh_basis_is_orthonormal_to_degree_4():
    # Gauss-Legendre in cos(theta) x uniform phi integrates products of degree-4 harmonics exactly
    xs, ws = np.polynomial.legendre.leggauss(16)
    phi = (np.arange(32) + 0.5) * (2 * math.pi / 32)
    ct, ph = np.meshgrid(xs, phi, indexing="ij")
    st = np.sqrt(1 - ct ** 2)
    d = torch.tensor(np.stack([st * np.cos(ph), st * np.sin(ph), ct], -1).reshape(-1, 3), dtype=torch.float64)
    w = torch.tensor(np.repeat(ws, 32) * (2 * math.pi / 32), dtype=torch.float64)
    Y = G.sh_basis(d, 4)
    gram = (Y * w[:, None]).T @ Y
    assert Y.shape[1] == 25
    assert torch.allclose(gram, torch.eye(25, dtype=torch.float64), atol=1e-12)


def test_sh_sign_convention_and_colour_offset():
    d = torch.tensor([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    Y = G.sh_basis(d, 1)
    c1 = 0.48860251190292
    # rows: direction +z, +x, +y; columns: Y_1^-1 = -C1 y, Y_1^0 = C1 z, Y_1^1 = -C1 x
    assert torch.allclose(Y[:, 1:], torch.tensor([[0, c1, 0], [0, 0, -c1], [-c1, 0, 0]], dtype=torch.float32), atol=1e-7)
    sh = torch.zeros(1, 25, 3)
    sh[0, 0] = torch.tensor([1.0, -5.0, 0.0])
    col = G.sh_colors(torch.tensor([[0.0, 0.0, 2.0]]), torch.zeros(3), sh, 4)
    assert torch.allclose(col, torch.tensor([[0.2820947917738781 + 0.5, 0.0, 0.5]]), atol=1e-6)  # clamp_min(. + 0.5, 0)


def _cam(W, H, f):
    K = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]])
    return torch.eye(4), K


def test_single_isotropic_gaussian_closed_form():
    W, H, f, z, s = 64, 48, 50.0, 4.0, 0.2
    view, K = _cam(W, H, f)
    means = torch.tensor([[0.0, 0.0, z]])
    cov = (s * s * torch.eye(3))[None]
    pr = G.project(means, cov, view, K, W, H)
    v = (f / z) ** 2 * s * s + 0.3
    assert torch.allclose(pr["means2d"], torch.tensor([[W / 2, H / 2]]))
    assert torch.allclose(pr["conics"], torch.tensor([[1 / v, 0.0, 1 / v]]), rtol=1e-5)
    assert int(pr["radii"][0]) == math.ceil(3 * math.sqrt(v + 0.1))  # b + sqrt(max(0.01, b^2 - det)) with b^2 == det
    sh = torch.zeros(1, 25, 3)
    sh[0, 0] = (torch.tensor([0.9, 0.3, 0.1]) - 0.5) / 0.2820947917738781
    op = torch.tensor([0.8])
    img, alpha, meta = G.rasterization(means, cov, op, sh, view, K, W, H, background=torch.ones(3))
    y, x = 24, 32  # pixel centre (32.5, 24.5): offset (0.5, 0.5) px from the mean
    a = 0.8 * math.exp(-0.5 * (0.25 + 0.25) / v)
    assert abs(alpha[y, x].item() - a) < 1e-6
    exp_rgb = torch.tensor([0.9, 0.3, 0.1]) * a + (1 - a)
    assert torch.allclose(img[y, x, :3], exp_rgb, atol=1e-5)
    assert abs(img[y, x, 3].item() - a * z) < 1e-5  # depth channel: no background, not normalised
    far_px = img[0, 0]
    assert torch.allclose(far_px[:3], torch.ones(3)) and alpha[0, 0] == 0  # outside 3 sigma / below 1/255


def test_front_to_back_compositing_and_saturation():
    W, H, f = 32, 32, 40.0
    view, K = _cam(W, H, f)
    means = torch.tensor([[0.0, 0.0, 5.0], [0.0, 0.0, 2.0], [0.0, 0.0, 9.0]])  # index 1 is nearest
    cov = (torch.tensor([1.0, 0.5, 2.0]) ** 2)[:, None, None] * torch.eye(3)[None]
    sh = torch.zeros(3, 25, 3)
    cols = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
    sh[:, 0] = (cols - 0.5) / 0.2820947917738781
    op = torch.tensor([0.5, 0.25, 1.0])
    img, alpha, meta = G.rasterization(means, cov, op, sh, view, K, W, H, background=torch.zeros(3))
    ids = meta["flatten_ids"][meta["tile_offsets"][0]:meta["tile_offsets"][1]].tolist()
    assert ids == [1, 0, 2]  # depth order inside the tile
    c = meta["conics"]
    px = torch.tensor([15.5, 15.5])
    al = []
    for g in (1, 0, 2):
        d = meta["means2d"][g] - px
        sig = 0.5 * (c[g, 0] * d[0] ** 2 + c[g, 2] * d[1] ** 2) + c[g, 1] * d[0] * d[1]
        al.append(min(0.999, float(op[g] * torch.exp(-sig))))
    T1, T2 = 1 - al[0], (1 - al[0]) * (1 - al[1])
    exp = torch.tensor([al[1] * T1, al[0], al[2] * T2])
    assert torch.allclose(img[15, 15, :3], exp, atol=1e-6)
    assert abs(alpha[15, 15].item() - (1 - T2 * (1 - al[2]))) < 1e-6
    # an opaque wall in front: transmittance would fall below 1e-4 -> the Gaussian that would cross it and all after are dropped
    op2 = torch.tensor([1.0, 1.0, 1.0])
    means2 = torch.tensor([[0.0, 0.0, 2.0], [0.0, 0.0, 2.5], [0.0, 0.0, 3.0]])
    img2, alpha2, _ = G.rasterization(means2, cov, op2, sh, view, K, W, H, background=torch.zeros(3))
    a = 0.999
    assert abs(alpha2[15, 15].item() - (1 - (1 - a))) < 1e-6          # second would give T = 1e-6 <= 1e-4: stop before it
    assert torch.allclose(img2[15, 15, :3], torch.tensor([a, 0.0, 0.0]), atol=1e-6)


def test_culling_rules():
    W, H, f = 64, 64, 60.0
    view, K = _cam(W, H, f)
    means = torch.tensor([[0.0, 0.0, -1.0],     # behind the camera
                          [0.0, 0.0, 0.0],      # z = 0 < near plane 1e-10
                          [50.0, 0.0, 1.0],     # far off screen to the right
                          [0.0, 0.0, 1.0]])     # visible
    cov = (0.01 * torch.eye(3))[None].repeat(4, 1, 1)
    pr = G.project(means, cov, view, K, W, H)
    assert pr["radii"].tolist()[:3] == [0, 0, 0] and pr["radii"][3] > 0
    # a degenerate (negative-definite) covariance gives det <= 0 after the blur only if large: radius 0
    cov_bad = (-100.0 * torch.eye(3))[None]
    assert int(G.project(means[3:], cov_bad, view, K, W, H)["radii"][0]) == 0


def test_binning_invariants_random_scene():
    g = torch.Generator().manual_seed(0)
    U, W, H = 300, 80, 72  # 5 x 5 tiles, last row/column partial
    means = torch.randn(U, 3, generator=g) * torch.tensor([1.5, 1.5, 1.0]) + torch.tensor([0.0, 0.0, 4.0])
    A = torch.randn(U, 3, 3, generator=g) * 0.15
    cov = A @ A.transpose(1, 2) + 1e-4 * torch.eye(3)
    view, K = _cam(W, H, 70.0)
    pr = G.project(means, cov, view, K, W, H)
    offs, ids = G.bin_and_sort(pr["means2d"], pr["radii"], pr["depths"], W, H)
    tmin, tmax, tw, th = G.tile_bounds(pr["means2d"], pr["radii"], W, H)
    assert (tw, th) == (5, 5) and offs[0] == 0 and offs[-1] == len(ids) and bool((offs[1:] >= offs[:-1]).all())
    want = int(((tmax - tmin)[:, 0] * (tmax - tmin)[:, 1]).sum())
    assert len(ids) == want
    for t in range(tw * th):
        seg = ids[offs[t]:offs[t + 1]]
        d = pr["depths"][seg]
        assert bool((d[1:] >= d[:-1]).all())
        ty, tx = divmod(t, tw)
        inb = (tmin[seg, 0] <= tx) & (tx < tmax[seg, 0]) & (tmin[seg, 1] <= ty) & (ty < tmax[seg, 1])
        assert bool(inb.all()) and len(set(seg.tolist())) == len(seg)


def test_camera_path_interpolation():
    from vist3a_amd.misc.image_io import interpolate_camera_path

    def rotz(a):
        c, s = math.cos(a), math.sin(a)
        m = torch.eye(4)
        m[:2, :2] = torch.tensor([[c, -s], [s, c]])
        return m

    ex = torch.stack([rotz(0.0), rotz(0.6), rotz(1.0)])[None].clone()
    ex[0, 1, :3, 3] = torch.tensor([1.0, 2.0, 3.0])
    ix = torch.eye(3)[None, None].repeat(1, 3, 1, 1).clone()
    ix[0, 1, 0, 0] = 2.0
    a, b = interpolate_camera_path(ex, ix, 1, t=3)
    ao, bo = G.interpolate_camera_path(ex, ix, t=3)
    assert a.shape == (1, 8, 4, 4) and b.shape == (1, 8, 3, 3)  # (V-1)(t+1); the last view is not rendered (reference quirk)
    assert torch.allclose(a, ao, atol=1e-6) and torch.allclose(b, bo, atol=1e-6)
    assert torch.equal(a[0, 0], ex[0, 0]) and torch.equal(a[0, 4], ex[0, 1])
    R = a[0, :, :3, :3]
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(8, 3, 3), atol=1e-6)
    assert torch.allclose(torch.linalg.det(R), torch.ones(8), atol=1e-6)
    mid = rotz(0.3)
    assert torch.allclose(a[0, 2, :3, :3], mid[:3, :3], atol=1e-6)  # blend of two rotations about one axis -> half angle
    assert torch.allclose(a[0, 2, :3, 3], torch.tensor([0.5, 1.0, 1.5]), atol=1e-6)
    assert abs(b[0, 2, 0, 0].item() - 1.5) < 1e-6


def test_mjpeg_avi_container(tmp_path):
    from vist3a_amd.misc.image_io import save_video
    v = torch.rand(5, 3, 32, 48)
    p = tmp_path / "x.avi"
    save_video(v, p, fps=20)
    raw = p.read_bytes()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"AVI " and struct.unpack("<I", raw[4:8])[0] == len(raw) - 8
    assert raw.count(b"00dc") == 10 and b"MJPG" in raw and b"idx1" in raw  # 5 frame chunks + 5 index entries
    i = raw.index(b"avih") + 8
    us_per_frame, _, _, _, nframes = struct.unpack("<IIIII", raw[i:i + 20])
    w, h = struct.unpack("<II", raw[i + 32:i + 40])
    assert (us_per_frame, nframes, w, h) == (50000, 5, 48, 32)
    from PIL import Image
    import io
    j = raw.index(b"00dc") + 8
    n = struct.unpack("<I", raw[j - 4:j])[0]
    im = Image.open(io.BytesIO(raw[j:j + n]))
    assert im.size == (48, 32)
