"""Direct kernel-level parity (`-m gpu`, through the C ABI) of the six small kernels that round 2 only exercised through the width-64
engine golden: qknorm_rope2d (16 heads), rownorm (+SiLU), softmax_rows, depth_unproject, linear_f32, attention_small_f32 - each
against a plain PyTorch fp32 computation of the reference's formula on the same inputs, at the production shapes of the
reconstruction / VAE.  Tolerances are <= 2x the error measured on MI355X (recorded through the `parity` fixture).

Reference formulas:
  qknorm_rope2d       vggt/layers/attention.py:57-63 (LayerNorm over head_dim on q and k) + rope.py:154-188 (2-D RoPE, base 100)
  rownorm             utils/wan_utils.py:150-184 (WanRMS_norm = F.normalize * sqrt(d) * gamma + bias), :370-372 (SiLU)
  softmax_rows        utils/wan_utils.py:428-475 (single-head attention of the VAE mid block)
  depth_unproject     vggt/heads/head_act.py:61-112 (exp / 1+exp) + vggt/utils/geometry.py:10-58 (unprojection)
  linear_f32 / attention_small_f32   vggt/heads/camera_head.py:87-170 (fp32 trunk over S camera tokens)
"""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import recon as R

pytestmark = pytest.mark.gpu
dev = "cuda"
bf16, f32 = torch.bfloat16, torch.float32


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


@pytest.mark.parametrize("S,hp,wp,H", [(13, 32, 32, 16), (3, 2, 3, 2), (21, 32, 32, 16)], ids=["13_views", "tiny_nonsquare", "21_views"])
def test_qknorm_rope2d_multi_head(hip_lib, parity, S, hp, wp, H):
    """q / k LayerNorm over each 64-wide head + 2-D RoPE on the padded token layout of the engine (Pp rows per view, 5 special tokens
    at position 0, filler rows ignored) vs oracle.recon.rope2d / F.layer_norm on the [B, H, N, 64] view the reference uses."""
    from vist3a_amd import ops
    from vist3a_amd.recon.engine import rope2d_table
    g = torch.Generator().manual_seed(S * 100 + H)
    C, nsp = H * 64, 5
    P = hp * wp + nsp
    Pp = (P + 7) // 8 * 8
    qk = torch.randn(S * Pp, 2 * C, generator=g).to(bf16)
    w = [1 + 0.2 * torch.randn(64, generator=g) for _ in range(2)]
    b = [0.1 * torch.randn(64, generator=g) for _ in range(2)]
    pos = R.patch_positions(S, hp, wp, nsp)                      # [S, P, 2], zeros for the specials
    outs = []
    for i in range(2):                                           # the reference's arithmetic, fp32, on the bf16 inputs
        t = qk[:, i * C:(i + 1) * C].float().view(S, Pp, H, 64)[:, :P].permute(0, 2, 1, 3)     # [S, H, P, 64]
        t = F.layer_norm(t, (64,), w[i], b[i], 1e-5)
        outs.append(R.rope2d(t, pos).permute(0, 2, 1, 3).reshape(S, P, C))
    x = qk.clone().to(dev)
    ops.qknorm_rope2d(x, C, w[0].to(dev), b[0].to(dev), w[1].to(dev), b[1].to(dev), rope2d_table(max(hp, wp) + 2).to(dev), Pp, nsp, P, wp, 1e-5)
    torch.cuda.synchronize()
    got = x.cpu().view(S, Pp, 2 * C)[:, :P]
    rq, rk = relerr(got[..., :C], outs[0]), relerr(got[..., C:], outs[1])
    parity("qknorm_rope2d", S=S, hp=hp, wp=wp, heads=H, rel_q=rq, rel_k=rk)
    print(f"qknorm_rope2d S={S} {hp}x{wp} H={H}: q {rq:.2e} k {rk:.2e}")
    assert rq < 3.3e-3 and rk < 3.3e-3     # measured 1.66e-3 = the bf16 rounding of the output itself (every element correctly rounded, below)
    # ... so compare before the final rounding as well: the bf16 result must be the correctly rounded fp32 reference almost everywhere
    exact = (got[..., :C] == outs[0].to(bf16)).float().mean().item()
    parity("qknorm_rope2d_bits", S=S, heads=H, fraction_correctly_rounded=exact)
    assert exact > 0.999, exact      # measured 1.0


@pytest.mark.parametrize("M,d,mode,act", [(13 * 64 * 64, 384, 1, True), (13 * 128 * 128, 192, 1, True), (4 * 512 * 512, 96, 1, True),
                                           (1000, 96, 0, False), (777, 1536, 0, False)])
def test_rownorm_act_matches_wan_rms_norm(hip_lib, parity, M, d, mode, act):
    from vist3a_amd import lib as L
    from vist3a_amd import ops
    g = torch.Generator(device=dev).manual_seed(M % 1000 + d)
    x = (torch.randn(M, d, device=dev, generator=g) * 1.7).to(bf16)
    w = 1 + 0.2 * torch.randn(d, device=dev, generator=g)
    b = 0.1 * torch.randn(d, device=dev, generator=g) if mode == 1 else None
    y = ops.rownorm_act(x, w, bias=b, mode=mode, act=L.ACT_SILU if act else L.ACT_NONE, eps=1e-6)
    xf = x.float()
    if mode == 1:    # WanRMS_norm: F.normalize(x, dim=channel) * sqrt(d) * gamma + bias
        ref = F.normalize(xf, dim=-1) * math.sqrt(d) * w + b
    else:            # RMSNorm: x * rsqrt(mean(x^2) + eps) * w
        ref = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * w
    if act:
        ref = F.silu(ref)
    r = relerr(y, ref)
    exact = (y == ref.to(bf16)).float().mean().item()
    parity("rownorm_act", M=M, d=d, mode=mode, silu=act, rel_vs_fp32=r, fraction_correctly_rounded=exact)
    assert r < 3.4e-3 and exact > 0.999, (r, exact)    # measured 1.68e-3 (= bf16 output rounding) / 1.0


@pytest.mark.parametrize("M,N,scale", [(4096, 4096, 384 ** -0.5), (1000, 1024, 0.05), (3, 8, 1.0)])
def test_softmax_rows_matches_torch(hip_lib, parity, M, N, scale):
    from vist3a_amd import ops
    g = torch.Generator(device=dev).manual_seed(N)
    s = torch.randn(M, N, device=dev, generator=g) * 30
    s[0, 3] = 400.0          # a dominant entry: everything else underflows
    p = ops.softmax_rows(s, scale)
    ref = torch.softmax(s * scale, -1)
    r = relerr(p, ref)
    rows = (p.float().sum(-1) - 1).abs().max().item()
    parity("softmax_rows", M=M, N=N, rel_vs_fp32=r, max_row_sum_error=rows)
    assert r < 3.2e-3 and rows < 2.7e-3, (r, rows)     # measured 1.6e-3 / 1.3e-3


def test_depth_unproject_matches_geometry(hip_lib, parity):
    """exp / 1+exp head activation + pixel -> camera -> world unprojection at 13 x 448 x 448 vs oracle.recon.unproject."""
    from vist3a_amd import ops
    from vist3a_amd.recon.engine import pose_encoding_to_extri_intri
    S, H, W = 13, 448, 448
    g = torch.Generator().manual_seed(9)
    raw = torch.randn(S * H * W, 8, generator=g) * 0.7
    pose = torch.cat([torch.randn(S, 3, generator=g) * 0.3, F.normalize(torch.randn(S, 4, generator=g), dim=-1),
                      0.8 + 0.2 * torch.rand(S, 2, generator=g)], -1)
    ext, K = R.pose_encoding_to_extri_intri(pose[None], (H, W))
    depth_ref = raw[:, 0].exp().view(1, S, H, W)
    pts_ref = R.unproject(depth_ref, ext, K)[0]
    e2, K2 = pose_encoding_to_extri_intri(pose.to(dev), (H, W))
    Rt = e2[:, :, :3].transpose(1, 2)
    tinv = -(Rt * e2[:, None, :, 3]).sum(-1)
    cam = torch.cat([K2[:, 0, 0:1], K2[:, 1, 1:2], K2[:, 0, 2:3], K2[:, 1, 2:3], Rt.reshape(S, 9), tinv], 1).contiguous()
    depth, conf, pts = ops.depth_unproject(raw.to(dev), cam, S, H, W)
    rd, rc, rp = relerr(depth.cpu(), depth_ref[0]), relerr(conf.cpu(), (1 + raw[:, 1].exp()).view(S, H, W)), relerr(pts.cpu(), pts_ref)
    parity("depth_unproject", S=S, H=H, W=W, rel_depth=rd, rel_conf=rc, rel_pts=rp)
    print(f"depth_unproject: depth {rd:.1e} conf {rc:.1e} pts {rp:.1e}")
    assert rd < 5e-8 and rc < 4e-8 and rp < 1.6e-7, (rd, rc, rp)    # measured 2.2e-8 / 1.7e-8 / 8.0e-8
    assert torch.allclose(e2.cpu(), ext[0], atol=1e-6) and torch.allclose(K2.cpu(), K[0], rtol=1e-6)


@pytest.mark.parametrize("M,N,K,act,extra", [(13, 6144, 2048, "none", False), (13, 2048, 2048, "none", True), (13, 8192, 2048, "gelu", False),
                                              (21, 2048, 8192, "none", True), (13, 2048, 12, "silu", False), (1, 9, 1024, "none", False),
                                              (32, 1024, 2048, "gelu", False)])
def test_linear_f32_matches_torch(hip_lib, parity, M, N, K, act, extra):
    """The fp32 camera-head linear at its production shapes (2048-wide trunk, S = 13 / 21 tokens): bias, GELU(erf) / SiLU,
    LayerScale gamma + residual."""
    from vist3a_amd import lib as L
    from vist3a_amd import ops
    g = torch.Generator(device=dev).manual_seed(N + K)
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)
    b = torch.randn(N, device=dev, generator=g) * 0.1
    res = torch.randn(M, N, device=dev, generator=g) if extra else None
    gam = 0.3 * (1 + 0.2 * torch.randn(N, device=dev, generator=g)) if extra else None
    y = ops.linear_f32(x, w, b, act={"none": L.ACT_NONE, "gelu": L.ACT_GELU_ERF, "silu": L.ACT_SILU}[act], residual=res, gamma=gam)
    ref = (x.double() @ w.double().t() + b.double())
    ref = {"none": lambda t: t, "gelu": F.gelu, "silu": F.silu}[act](ref)
    if extra:
        ref = res.double() + gam.double() * ref
    r = relerr(y.double(), ref)
    parity("linear_f32", M=M, N=N, K=K, act=act, ls_residual=extra, rel_vs_fp64=r)
    assert r < 4e-7, r       # measured 4e-8 .. 2e-7


@pytest.mark.parametrize("S,H", [(13, 16), (21, 16), (2, 4), (32, 16)])
def test_attention_small_f32_matches_sdpa(hip_lib, parity, S, H):
    from vist3a_amd import ops
    hd = 128
    g = torch.Generator(device=dev).manual_seed(S)
    qkv = torch.randn(S, 3 * H * hd, device=dev, generator=g)
    out = ops.attention_small_f32(qkv, H)
    q, k, v = qkv.double().view(S, 3, H, hd).permute(1, 2, 0, 3)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1) @ v).permute(1, 0, 2).reshape(S, H * hd)
    r = relerr(out.double(), ref)
    parity("attention_small_f32", S=S, H=H, rel_vs_fp64=r)
    assert r < 6e-7, r       # measured 1.3e-7 .. 3.0e-7
