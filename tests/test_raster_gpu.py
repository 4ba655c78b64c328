"""GPU parity: HIP 3D-Gaussian rasteriser (csrc/raster.hip through the C ABI) vs oracle/gsplat_raster.py.

  * index work (tile ranges, per-tile composite order) is BIT-EXACT when both sides bin the same projected Gaussians;
  * projection floats: 2e-5 relative (fp32, different contraction/order); radii may differ by one pixel only where
    3*sqrt(lambda) lands within float noise of an integer (counted, must be < 0.2 %);
  * pixels: 2e-5 absolute when both composite the same projected inputs; end to end 99.9 % of pixels within 1e-3
    (a one-pixel radius flip or an alpha straddling 1/255 moves isolated pixels), mean abs error < 2e-5;
  * at full size (448^2, 1M Gaussians) size-independent properties: determinism, alpha range, white-minus-black = 1 - alpha,
    zero-opacity Gaussians are invisible."""
import math

import pytest
import torch

from oracle import gsplat_raster as G

pytestmark = pytest.mark.gpu


def _scene(U, seed, spread=1.5, scale=0.15, z0=4.0):
    g = torch.Generator().manual_seed(seed)
    means = torch.randn(U, 3, generator=g) * torch.tensor([spread, spread, 1.0]) + torch.tensor([0.0, 0.0, z0])
    A = torch.randn(U, 3, 3, generator=g) * scale
    cov = A @ A.transpose(1, 2) + 1e-4 * torch.eye(3)
    sh = torch.randn(U, 3, 25, generator=g) * 0.3  # Gaussians.harmonics layout [U,3,K]
    op = torch.rand(U, generator=g)
    return means, cov, sh, op


def _camera(W, H, f, yaw=0.2, t=(0.1, -0.2, 0.3)):
    c, s = math.cos(yaw), math.sin(yaw)
    view = torch.eye(4)
    view[:3, :3] = torch.tensor([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    view[:3, 3] = torch.tensor(t)
    K = torch.tensor([[f, 0, W / 2], [0, f * 1.1, H / 2], [0, 0, 1.0]])
    return view, K


def _project_gpu(ops, means, cov, sh, view, K, W, H):
    campos = torch.linalg.inv(view)[:3, 3].contiguous()
    return ops.gs_project(means.cuda(), cov.cuda().contiguous(), sh.cuda().contiguous(), view.cuda().contiguous(), campos.cuda(),
                          K.cuda().contiguous(), W, H)


@pytest.mark.parametrize("U,W,H", [(4000, 96, 80), (1, 16, 16), (700, 130, 50)])
def test_projection_matches_oracle(hip_lib, U, W, H):
    from vist3a_amd import ops
    means, cov, sh, op = _scene(U, 1)
    view, K = _camera(W, H, 70.0)
    pr = _project_gpu(ops, means, cov, sh, view, K, W, H)
    ref = G.project(means, cov, view, K, W, H)
    col = G.sh_colors(means, torch.linalg.inv(view)[:3, 3], sh.permute(0, 2, 1), 4)
    rg, rr = pr["radii"].cpu(), ref["radii"]
    diff = (rg != rr)
    assert int((rg - rr).abs().max()) <= 1 and diff.float().mean().item() < 2e-3
    vis = (rg > 0) & (rr > 0)
    assert torch.allclose(pr["depths"].cpu(), ref["depths"], rtol=2e-6, atol=1e-6)
    assert torch.allclose(pr["means2d"].cpu()[vis], ref["means2d"][vis], rtol=2e-5, atol=2e-4)
    assert torch.allclose(pr["conics"].cpu()[vis], ref["conics"][vis], rtol=2e-4, atol=1e-6)
    assert torch.allclose(pr["colors"].cpu()[vis][:, :3], col[vis], rtol=1e-5, atol=2e-6)
    assert torch.equal(pr["colors"].cpu()[:, 3], pr["depths"].cpu())
    assert bool((pr["colors"].cpu()[~(rg > 0)][:, :3] == 0).all())  # masked SH evaluation


def test_sh_layouts_agree(hip_lib):
    from vist3a_amd import ops
    means, cov, sh, op = _scene(500, 2)
    view, K = _camera(64, 64, 60.0)
    campos = torch.linalg.inv(view)[:3, 3].contiguous().cuda()
    a = ops.gs_project(means.cuda(), cov.cuda(), sh.cuda(), view.cuda(), campos, K.cuda(), 64, 64, sh_layout=1)
    b = ops.gs_project(means.cuda(), cov.cuda(), sh.permute(0, 2, 1).contiguous().cuda(), view.cuda(), campos, K.cuda(), 64, 64, sh_layout=0)
    assert torch.equal(a["colors"], b["colors"])
    for deg in (0, 2):
        c = ops.gs_project(means.cuda(), cov.cuda(), sh.cuda(), view.cuda(), campos, K.cuda(), 64, 64, sh_degree=deg)["colors"].cpu()
        ref = G.sh_colors(means, campos.cpu(), sh.permute(0, 2, 1), deg)
        v = a["radii"].cpu() > 0
        assert torch.allclose(c[v][:, :3], ref[v], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("U,W,H,seed", [(3000, 96, 80, 3), (150, 33, 17, 4), (40, 160, 48, 5)])
def test_binning_bit_exact_and_pixels_on_same_inputs(hip_lib, U, W, H, seed):
    from vist3a_amd import ops
    means, cov, sh, op = _scene(U, seed, scale=0.25)
    view, K = _camera(W, H, 60.0)
    ref_img, ref_alpha, meta = G.rasterization(means, cov, op, sh.permute(0, 2, 1), view, K, W, H, background=torch.ones(3))
    pr = {k: meta[k].cuda().contiguous() for k in ("radii", "means2d", "depths", "conics", "colors")}
    out = ops.gs_rasterize(pr, op.cuda(), W, H, background=torch.ones(3, device="cuda"), clamp_rgb=False, return_order=True)
    assert out["n_isect"] == len(meta["flatten_ids"])
    assert torch.equal(out["tile_offsets"].cpu().long(), meta["tile_offsets"])
    assert torch.equal(out["flatten_ids"].cpu().long(), meta["flatten_ids"])
    assert torch.allclose(out["color"].cpu(), ref_img[..., :3], atol=2e-5, rtol=1e-5)
    assert torch.allclose(out["depth"].cpu(), ref_img[..., 3], atol=1e-4, rtol=1e-5)
    assert torch.allclose(out["alpha"].cpu(), ref_alpha, atol=2e-5)


def test_equal_depth_ties_keep_gaussian_order(hip_lib):
    """Stable sort: coincident depths composite in ascending Gaussian index, as gsplat's radix sort does."""
    from vist3a_amd import ops
    U, W, H = 64, 32, 32
    means = torch.zeros(U, 3)
    means[:, 2] = 3.0
    means[:, 0] = torch.linspace(-0.2, 0.2, U)
    cov = (0.05 * torch.eye(3))[None].repeat(U, 1, 1)
    sh = torch.randn(U, 3, 25, generator=torch.Generator().manual_seed(0)) * 0.2
    op = torch.full((U,), 0.3)
    view, K = torch.eye(4), torch.tensor([[30.0, 0, 16], [0, 30.0, 16], [0, 0, 1]])
    _, _, meta = G.rasterization(means, cov, op, sh.permute(0, 2, 1), view, K, W, H)
    pr = {k: meta[k].cuda().contiguous() for k in ("radii", "means2d", "depths", "conics", "colors")}
    out = ops.gs_rasterize(pr, op.cuda(), W, H, return_order=True)
    assert torch.equal(out["flatten_ids"].cpu().long(), meta["flatten_ids"])
    seg = out["flatten_ids"][out["tile_offsets"][0]:out["tile_offsets"][1]].cpu()
    assert bool((seg[1:] > seg[:-1]).all())


def test_empty_and_fully_culled(hip_lib):
    from vist3a_amd import ops
    means, cov, sh, op = _scene(50, 6)
    means[:, 2] = -5.0  # everything behind the camera
    view, K = torch.eye(4), torch.tensor([[50.0, 0, 24], [0, 50.0, 24], [0, 0, 1]])
    pr = _project_gpu(ops, means, cov, sh, view, K, 48, 48)
    assert int(pr["radii"].abs().sum()) == 0
    bg = torch.tensor([0.2, 0.4, 0.6], device="cuda")
    out = ops.gs_rasterize(pr, op.cuda(), 48, 48, background=bg, return_order=True)
    assert out["n_isect"] == 0 and int(out["tile_offsets"].abs().sum()) == 0
    assert torch.allclose(out["color"], bg.expand(48, 48, 3)) and float(out["alpha"].abs().max()) == 0 and float(out["depth"].abs().max()) == 0
    out2 = ops.gs_rasterize(pr, op.cuda(), 48, 48, background=None)
    assert float(out2["color"].abs().max()) == 0


def test_workspace_grows_when_intersections_exceed_capacity(hip_lib):
    from vist3a_amd import ops
    means, cov, sh, op = _scene(2000, 7, scale=0.6)
    view, K = _camera(256, 256, 200.0)
    pr = _project_gpu(ops, means, cov, sh, view, K, 256, 256)
    ws = ops.GsWorkspace()
    ws.get(2000, 1, 256, 256, 1)
    ws.cap = 64  # pretend the scratch was sized for a tiny scene
    a = ops.gs_rasterize(pr, op.cuda(), 256, 256, workspace=ws)
    assert a["n_isect"] > 64 and ws.cap >= a["n_isect"]
    b = ops.gs_rasterize(pr, op.cuda(), 256, 256)
    assert torch.equal(a["color"], b["color"]) and torch.equal(a["alpha"], b["alpha"])


def test_camera_batch_equals_single_camera_calls(hip_lib):
    """C cameras in one launch (camera-major keys) == C one-camera launches, bit for bit; composite ids are c*U + g."""
    from vist3a_amd import ops
    U, W, H = 3000, 100, 70
    means, cov, sh, op = _scene(U, 11, scale=0.2)
    cams = [_camera(W, H, 65.0 + 5 * i, yaw=0.1 * i - 0.1, t=(0.05 * i, 0.0, 0.1 * i)) for i in range(3)]
    view = torch.stack([c[0] for c in cams]).cuda()
    K = torch.stack([c[1] for c in cams]).cuda()
    campos = torch.stack([torch.linalg.inv(c[0])[:3, 3] for c in cams]).contiguous().cuda()
    m, c, s_, o = means.cuda(), cov.cuda(), sh.cuda(), op.cuda()
    bg = torch.tensor([1.0, 0.5, 0.0], device="cuda")
    prb = ops.gs_project(m, c, s_, view, campos, K, W, H)
    outb = ops.gs_rasterize(prb, o, W, H, background=bg, return_order=True)
    ntiles = ((W + 15) // 16) * ((H + 15) // 16)
    tot = 0
    for j in range(3):
        pr1 = ops.gs_project(m, c, s_, view[j].contiguous(), campos[j].contiguous(), K[j].contiguous(), W, H)
        for k in pr1:
            assert torch.equal(pr1[k], prb[k][j])
        o1 = ops.gs_rasterize(pr1, o, W, H, background=bg, return_order=True)
        for k in ("color", "depth", "alpha"):
            assert torch.equal(o1[k], outb[k][j])
        lo, hi = int(outb["tile_offsets"][j * ntiles]), int(outb["tile_offsets"][(j + 1) * ntiles])
        assert hi - lo == o1["n_isect"]
        assert torch.equal(outb["flatten_ids"][lo:hi] - j * U, o1["flatten_ids"])
        tot += o1["n_isect"]
    assert tot == outb["n_isect"]


def test_decoder_end_to_end_vs_oracle(hip_lib):
    """DecoderSplattingCUDA.forward (c2w extrinsics, normalised intrinsics, white background, clamp) on 3 cameras."""
    from vist3a_amd.models.decoder_splatting import DecoderSplattingCUDA
    from vist3a_amd.models.types import Gaussians
    U, W, H = 2500, 112, 96
    means, cov, sh, op = _scene(U, 8, scale=0.2)
    g = Gaussians(means=means[None].cuda(), covariances=cov[None].cuda(), harmonics=sh[None].cuda(), opacities=op[None].cuda(),
                  scales=torch.ones(1, U, 3).cuda(), rotations=torch.zeros(1, U, 4).cuda())
    views = [_camera(W, H, 80.0, yaw=a, t=(0.1 * i, 0.0, 0.2))[0] for i, a in enumerate((0.0, 0.15, -0.2))]
    c2w = torch.stack([torch.linalg.inv(v) for v in views])[None]
    Kn = torch.tensor([[80.0 / W, 0, 0.5], [0, 88.0 / H, 0.5], [0, 0, 1.0]])[None, None].repeat(1, 3, 1, 1)
    dec = DecoderSplattingCUDA(background_color=(1.0, 1.0, 1.0), camera_batch=2)  # 3 views -> batches of 2 + 1
    out = dec.forward(g, c2w.cuda(), Kn.cuda(), torch.full((1, 3), 0.1).cuda(), torch.full((1, 3), 100.0).cuda(), (H, W))
    assert out.color.shape == (1, 3, 3, H, W) and out.depth.shape == (1, 3, H, W) and out.alpha.shape == (1, 3, H, W)
    for j in range(3):
        w2c = torch.linalg.inv(c2w[0, j])
        K = Kn[0, j].clone()
        K[0] *= W
        K[1] *= H
        img, alpha, _ = G.rasterization(means, cov, op, sh.permute(0, 2, 1), w2c, K, W, H, background=torch.ones(3))
        ref = img[..., :3].clamp(0, 1).permute(2, 0, 1)
        err = (out.color[0, j].cpu() - ref).abs()
        assert err.mean().item() < 2e-5 and (err.amax(0) < 1e-3).float().mean().item() > 0.999
        assert (out.alpha[0, j].cpu() - alpha).abs().mean().item() < 2e-5
        assert (out.depth[0, j].cpu() - img[..., 3]).abs().mean().item() < 1e-4


def test_full_size_properties(hip_lib):
    """448^2, 1M Gaussians (the production shape): determinism, ranges, background linearity, zero-opacity invisibility."""
    from vist3a_amd import ops
    U, W, H = 1_000_000, 448, 448
    means, cov, sh, op = _scene(U, 9, spread=2.0, scale=0.02, z0=5.0)
    view, K = _camera(W, H, 400.0)
    pr = _project_gpu(ops, means, cov, sh, view, K, W, H)
    opc = op.cuda()
    ws = ops.GsWorkspace()
    white = ops.gs_rasterize(pr, opc, W, H, background=torch.ones(3, device="cuda"), clamp_rgb=False, workspace=ws, return_order=True)
    again = ops.gs_rasterize(pr, opc, W, H, background=torch.ones(3, device="cuda"), clamp_rgb=False, workspace=ws)
    black = ops.gs_rasterize(pr, opc, W, H, background=torch.zeros(3, device="cuda"), clamp_rgb=False, workspace=ws)
    assert white["n_isect"] > U // 4
    for k in ("color", "depth", "alpha"):
        assert torch.equal(white[k], again[k])
    a = white["alpha"]
    assert float(a.min()) >= 0 and float(a.max()) <= 1 and torch.isfinite(white["color"]).all()
    assert torch.allclose(white["color"] - black["color"], (1 - a)[..., None].expand(-1, -1, 3), atol=1e-5)
    assert torch.equal(white["depth"], black["depth"])
    # per-tile depth order of the composite list
    offs, ids = white["tile_offsets"].long(), white["flatten_ids"].long()
    d = pr["depths"][ids]
    same_tile = torch.ones(len(ids) - 1, dtype=torch.bool, device="cuda")
    same_tile[(offs[1:-1] - 1).clamp(0, len(ids) - 2)] = False
    assert bool(((d[1:] >= d[:-1]) | ~same_tile).all())
    # Gaussians with zero opacity contribute nothing (alpha < 1/255 is skipped)
    half = opc.clone()
    half[::2] = 0
    pr2 = dict(pr)
    r2 = pr["radii"].clone()
    r2[::2] = 0
    pr2["radii"] = r2
    x = ops.gs_rasterize(pr, half, W, H, clamp_rgb=False, workspace=ws)
    y = ops.gs_rasterize(pr2, half, W, H, clamp_rgb=False, workspace=ws)
    assert torch.allclose(x["color"], y["color"], atol=1e-6) and torch.allclose(x["alpha"], y["alpha"], atol=1e-6)
