"""Kernel-level parity at PRODUCTION shapes, through the C ABI (`-m gpu`).

Every hand-written kernel that carries a scene is compared with a plain PyTorch fp32 computation of the same op on the same
bf16 inputs with the same rounding points (tolerances next to each test, set at <= 2x the error measured on MI355X and
recorded in profiles/r2/parity.json through the `parity` fixture); GEMM tile shapes are additionally asserted BIT-identical to
each other (same ascending-k chain of 32x32x16 MFMAs in every main loop).
north_star's 1e-3 relative: met by GEMM/conv (<= 1e-4 wherever both sides round at the same points), norms (<= 1e-5 fp32 /
bf16-rounding-limited otherwise), resize; NOT met by flash attention (bf16 P operand of the PV MFMA, 2..3e-3 like every
flash kernel incl. the reference's own SDPA)."""
import math
from pathlib import Path

import pytest
import torch
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu
dev = "cuda"
bf16, f32 = torch.bfloat16, torch.float32
G = Path(__file__).parent / "golden"


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def gemm_ref(L, a, w, bias, act, residual, scale, rpb, round_after_scale, bias_row, out_f32):
    v = a.float() @ w.float().t()
    if bias is not None:
        v = v + (bias[:, None] if bias_row else bias[None, :])
    v = v.to(bf16).float()
    F = torch.nn.functional
    if act == L.ACT_GELU_TANH:
        v = F.gelu(v, approximate="tanh").to(bf16).float()
    elif act == L.ACT_GELU_ERF:
        v = F.gelu(v).to(bf16).float()
    elif act == L.ACT_SILU:
        v = F.silu(v).to(bf16).float()
    elif act == L.ACT_RELU:
        v = torch.relu(v)
    if scale is not None:
        v = v * (scale.repeat_interleave(rpb, dim=0)[: v.shape[0]] if scale.dim() == 2 else scale[None])
        if round_after_scale:
            v = v.to(bf16).float()
    if residual is not None:
        v = v + residual.float()
    return v if out_f32 else v.to(bf16)


def _tile_names(hip_lib):
    return [hip_lib.v3a_gemm_tile_name(t).decode() for t in range(hip_lib.v3a_gemm_num_tiles())]


# the DiT's five projection shapes at 13 views, CFG batch 2 (M = 8192 rows), with the epilogue each one carries in the model
DIT_SHAPES = [
    ("attn_out+gate+res", 8192, 1536, 1536, dict(bias=True, res="bf16", scale="batch", rpb=4096)),
    ("qk_proj", 8192, 3072, 1536, dict(bias=True)),
    ("ffn1+gelu", 8192, 8960, 1536, dict(bias=True, act="gelu_tanh")),
    ("ffn2+gate+res", 8192, 1536, 8960, dict(bias=True, res="bf16", scale="batch", rpb=4096)),
    ("vT_proj", 1536, 8192, 1536, dict(bias=True, bias_row=True)),
    # BASELINE config #4: Wan-14B (d = 5120 = 26.67 x 192, ffn 13824 = 72 x 192: ragged / multi-round tilings at M = 8192)
    ("14b_attn_out+gate+res", 8192, 5120, 5120, dict(bias=True, res="bf16", scale="batch", rpb=4096)),
    ("14b_qk_proj", 8192, 10240, 5120, dict(bias=True)),
    ("14b_ffn1+gelu", 8192, 13824, 5120, dict(bias=True, act="gelu_tanh")),
    ("14b_ffn2+gate+res", 8192, 5120, 13824, dict(bias=True, res="bf16", scale="batch", rpb=4096)),
    ("14b_vT_proj", 5120, 8192, 5120, dict(bias=True, bias_row=True)),
]


@pytest.mark.parametrize("name,M,N,K,opt", DIT_SHAPES, ids=[s[0] for s in DIT_SHAPES])
def test_gemm_production_shapes_every_tile(hip_lib, parity, name, M, N, K, opt):
    """All tile shapes (lockstep 0-5, ping-pong 6-8) at the DiT's real shapes: <= 1.2e-4 vs
    fp32 torch with identical rounding points, and bit-equal to each other."""
    from vist3a_amd import lib as L, ops
    g = torch.Generator(device=dev).manual_seed(hash(name) % 1000)
    a = torch.randn(M, K, device=dev, generator=g).to(bf16)
    w = (torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).to(bf16)
    bias = torch.randn(M if opt.get("bias_row") else N, device=dev, generator=g) if opt.get("bias") else None
    res = torch.randn(M, N, device=dev, generator=g).to(bf16) if opt.get("res") else None
    rpb = opt.get("rpb", 0)
    scale = torch.randn((M + rpb - 1) // rpb, N, device=dev, generator=g) if opt.get("scale") == "batch" else None
    act = dict(gelu_tanh=L.ACT_GELU_TANH).get(opt.get("act"), L.ACT_NONE)
    kw = dict(act=act, residual=res, scale=scale, rows_per_batch=rpb, bias_row=opt.get("bias_row", False))
    ref = gemm_ref(L, a, w, bias, act, res, scale, rpb, False, kw["bias_row"], False)
    names = _tile_names(hip_lib)
    auto = hip_lib.v3a_gemm_pick_tile(M, N)
    first, worst = None, 0.0
    for t, tn in enumerate(names):
        out = ops.gemm(a, w, bias, tile=t, **kw)
        torch.cuda.synchronize()
        r = relerr(out, ref)
        worst = max(worst, r)
        parity("gemm_production", shape=name, M=M, N=N, K=K, tile=tn, auto=(t == auto), rel_vs_fp32=r)
        assert math.isfinite(r) and r < 1.2e-4, (tn, r)   # measured <= 5.5e-5 (profiles/r2/parity.json)
        if first is None:
            first = out
        else:
            assert torch.equal(out, first), f"tile {tn} differs bitwise from {names[0]}"
    print(f"{name}: worst rel {worst:.2e} over {len(names)} tiles (auto = {names[auto]}), all bit-identical")


def test_gemm_epilogue_flags_every_tile(hip_lib, parity):
    """Small / ragged problems x every epilogue flag x every tile."""
    from vist3a_amd import lib as L, ops
    g = torch.Generator(device=dev).manual_seed(0)
    cases = [
        dict(M=512, N=768, K=256),
        dict(M=300, N=200, K=128),
        dict(M=1029, N=1024, K=1024, act=L.ACT_GELU_ERF),
        dict(M=512, N=384, K=256, act=L.ACT_GELU_TANH, bias=True),
        dict(M=640, N=256, K=128, bias=True, res="bf16", scale="batch", rpb=320),
        dict(M=520, N=256, K=128, bias=True, res="f32", scale="col", round_after_scale=True, out_f32=True),
        dict(M=384, N=1000, K=64, bias=True, bias_row=True),
        dict(M=257, N=264, K=64, act=L.ACT_SILU, bias=True),
        dict(M=2100, N=520, K=192, act=L.ACT_RELU, bias=True, res="bf16"),
    ]
    names = _tile_names(hip_lib)
    for c in cases:
        M, N, K = c["M"], c["N"], c["K"]
        a = torch.randn(M, K, device=dev, generator=g).to(bf16)
        w = (torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).to(bf16)
        bias = torch.randn(M if c.get("bias_row") else N, device=dev, generator=g) if c.get("bias") else None
        res = torch.randn(M, N, device=dev, generator=g).to(bf16 if c.get("res") == "bf16" else f32) if c.get("res") else None
        rpb = c.get("rpb", 0)
        scale = None
        if c.get("scale") == "batch":
            scale = torch.randn((M + rpb - 1) // rpb, N, device=dev, generator=g)
        elif c.get("scale") == "col":
            scale = torch.randn(N, device=dev, generator=g)
        kw = dict(act=c.get("act", 0), residual=res, scale=scale, rows_per_batch=rpb,
                  round_after_scale=c.get("round_after_scale", False), out_f32=c.get("out_f32", False), bias_row=c.get("bias_row", False))
        ref = gemm_ref(L, a, w, bias, kw["act"], res, scale, rpb, kw["round_after_scale"], kw["bias_row"], kw["out_f32"])
        first = None
        for t, tn in enumerate(names):
            out = ops.gemm(a, w, bias, tile=t, **kw)
            torch.cuda.synchronize()
            r = relerr(out, ref)
            parity("gemm_flags", case={k: v for k, v in c.items()}, tile=tn, rel_vs_fp32=r)
            # erf/tanh GELU: device transcendental vs torch's differ by an ulp before the bf16 rounding (measured <= 6e-4)
            assert math.isfinite(r) and r < (1.5e-3 if kw["act"] in (L.ACT_GELU_ERF, L.ACT_GELU_TANH, L.ACT_SILU) else 1.2e-4), (tn, c, r)
            if first is None:
                first = out
            else:
                assert torch.equal(out, first), (tn, c)


@pytest.mark.parametrize("M,N,K,S,opt", [(1024, 1536, 8960, 4, dict(res=True, scale=True)), (512, 1536, 8960, 2, dict(act="gelu")),
                                         (1000, 520, 1024, 4, dict(bias_row=True, out_f32=True)), (2048, 5120, 13824, 2, {})])
def test_gemm_split_k_matches_unsplit(hip_lib, parity, M, N, K, S, opt):
    """split_k (a sequence-parallel shard's long-K FFN2): S K-slices side by side + the finishing launch against the fp32 reference
    with the documented rounding points and against the one-launch kernel (one extra bf16 rounding per partial apart); deterministic."""
    from vist3a_amd import lib as L, ops
    g = torch.Generator(device=dev).manual_seed(M + S)
    a = torch.randn(M, K, device=dev, generator=g).to(bf16)
    w = (torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).to(bf16)
    bias_row, out_f32 = opt.get("bias_row", False), opt.get("out_f32", False)
    bias = torch.randn(M if bias_row else N, device=dev, generator=g)
    act = L.ACT_GELU_TANH if opt.get("act") == "gelu" else L.ACT_NONE
    res = torch.randn(M, N, device=dev, generator=g).to(bf16) if opt.get("res") else None
    scale = torch.randn(2, N, device=dev, generator=g) if opt.get("scale") else None
    rpb = M // 2 if scale is not None else 0
    kw = dict(act=act, residual=res, scale=scale, rows_per_batch=rpb, bias_row=bias_row, out_f32=out_f32)
    one = ops.gemm(a, w, bias, **kw)
    s1 = ops.gemm(a, w, bias, split_k=S, **kw)
    s2 = ops.gemm(a, w, bias, split_k=S, **kw)
    ref = gemm_ref(L, a, w, bias, act, res, scale, rpb, False, bias_row, out_f32)
    r, r1 = relerr(s1, ref), relerr(s1, one)
    parity("gemm_split_k", M=M, N=N, K=K, S=S, rel_vs_fp32=r, rel_vs_one_launch=r1)
    assert torch.equal(s1, s2)
    assert r < 4e-3 and r1 < 4e-3, (r, r1)   # bf16 partials: ~sqrt(2) x one output rounding (2.2e-3)
    with pytest.raises(RuntimeError):
        ops.gemm(a[:, :192], w[:, :192], None, split_k=2)   # K % (64 * split_k) != 0


def test_gemm_rejects_unknown_flag_bits(hip_lib):
    from vist3a_amd import ops
    import ctypes as C
    from vist3a_amd import lib as L
    a = torch.zeros(64, 64, device=dev, dtype=bf16)
    out = torch.empty(64, 64, device=dev, dtype=bf16)
    for bit in (27, 28, 29, 30):
        args = L.GemmArgs(a.data_ptr(), a.data_ptr(), out.data_ptr(), None, None, None, 64, 64, 64, 64, 64, 64, 0,
                          0, 0, 0, 1 << bit, -1, None, 0, 0, 0, 0, 0)
        assert hip_lib.v3a_gemm_bf16_nt(C.byref(args), C.c_void_p(torch.cuda.current_stream().cuda_stream)) == -1  # V3A_ERR_ARG


# ---------------------------------------------------------------- attention ----
# vs exact fp32 softmax attention: bounded by the bf16 P operand (2.2-2.4e-3 measured on every shape; every flash kernel incl. the
# reference's CUDA SDPA has this term).  vs the kernel's own rounding contract (bf16 P emulated): fp32 round-off + the bf16 flips of the
# output it causes - THIS is the gate that shows whether the kernel computes what it says (north_star: 1e-3).
TOL_ATTN_EXACT = 4.5e-3      # measured 1.7e-3 .. 2.36e-3
TOL_ATTN_CONTRACT = 3e-4     # measured 2e-5 .. 1.5e-4 (north_star asks for 1e-3)
def _attn_inputs(B, H, Nq, Nk, D, g, scale=1.0):
    q = (torch.randn(B, Nq, H * D, device=dev, generator=g) * scale).to(bf16)
    k = (torch.randn(B, Nk, H * D, device=dev, generator=g) * scale).to(bf16)
    v = torch.randn(B, Nk, H * D, device=dev, generator=g).to(bf16)
    nkp = (Nk + 63) // 64 * 64
    vt = torch.zeros(H * D, B * nkp, device=dev, dtype=bf16)
    vt.view(H * D, B, nkp)[:, :, :Nk] = v.permute(2, 0, 1)
    return q, k, v, vt, nkp


def _attn_ref(q, k, v, B, H, D, bias=None, mask=None):
    """fp32 softmax(q k^T / sqrt(D) + bias) v per head (chunked over heads to bound memory)."""
    qf = q.float().view(B, -1, H, D).transpose(1, 2)
    kf = k.float().view(B, -1, H, D).transpose(1, 2)
    vf = v.float().view(B, -1, H, D).transpose(1, 2)
    outs = []
    for h in range(H):
        s = (qf[:, h] @ kf[:, h].transpose(-1, -2)) * D ** -0.5
        if bias is not None:
            s = s + bias[:, None, :]
        if mask is not None:
            s = s.masked_fill(~mask[None, None, :], float("-inf"))
        outs.append(torch.softmax(s, -1) @ vf[:, h])
    return torch.stack(outs, 2).reshape(B, -1, H * D)


def _attn_emu(q, k, v, B, H, D, bias=None, mask=None):
    """The kernel's CONTRACT (oracle.wan_dit.attention_flash_emulated: bf16 P per 64-key tile against the deferred per-wave exponent
    reference, fp32 everywhere else), evaluated with torch on the GPU, output rounded to bf16 like the kernel's."""
    from oracle.wan_dit import attention_flash_emulated
    hs = lambda t: t.float().view(B, -1, H, D).transpose(1, 2)
    kb = None if bias is None else bias[:, None, :]
    o = attention_flash_emulated(hs(q), hs(k), hs(v), D ** -0.5, key_bias=kb, key_mask=mask)
    return o.transpose(1, 2).reshape(B, -1, H * D).to(bf16)


def _run_attn(q, k, vt, nkp, B, H, Nq, Nk, D, **kw):
    from vist3a_amd import ops
    out = torch.empty(B * Nq, H * D, device=dev, dtype=bf16)
    ops.attention(q.view(B * Nq, H * D), k.view(B * Nk, H * D), vt, out, B=B, H=H, Nq=Nq, Nk=Nk, D=D,
                  q_batch_stride=Nq * H * D, k_batch_stride=Nk * H * D, vt_batch_stride=nkp, o_batch_stride=Nq * H * D, **kw)
    torch.cuda.synchronize()
    return out.view(B, Nq, H * D)


def test_attention_hd128_dit_self_attention_shape(hip_lib, parity):
    """The DiT self-attention launch exactly as the model issues it: B=2 (CFG), 12 heads, 4096 x 4096, hd 128."""
    B, H, N, D = 2, 12, 4096, 128
    g = torch.Generator(device=dev).manual_seed(11)
    q, k, v, vt, nkp = _attn_inputs(B, H, N, N, D, g)
    out = _run_attn(q, k, vt, nkp, B, H, N, N, D)
    r, re = relerr(out, _attn_ref(q, k, v, B, H, D)), relerr(out, _attn_emu(q, k, v, B, H, D))
    parity("attention_hd128_self", B=B, H=H, Nq=N, Nk=N, rel_vs_fp32=r, rel_vs_contract=re)
    print(f"hd128 4096x4096 rel vs exact softmax {r:.2e}, vs the kernel's rounding contract {re:.2e}")
    assert r < TOL_ATTN_EXACT and re < TOL_ATTN_CONTRACT, (r, re)


@pytest.mark.parametrize("B,H,N", [(2, 12, 4096), (1, 3, 2048)])
def test_attention_plain_kernel_is_bit_identical_to_the_general_kernel(hip_lib, parity, B, H, N):
    """The DiT self-attention runs on its own hand-scheduled kernel (attn_fwd_plain_kernel, csrc/attention.hip) when Nk % 64 == 0 and no
    mask / bias / segment option is set; the same launch described as ONE key segment (kv_seg = Nk) takes the general, hipcc-scheduled
    attn_fwd_kernel.  Same arithmetic, same order, explicit FMAs: every output bit must agree (the sequence-parallel forward, which reads
    gathered slabs through kv_seg, relies on it), and both must be deterministic."""
    D = 128
    g = torch.Generator(device=dev).manual_seed(N + H)
    q, k, v, vt, nkp = _attn_inputs(B, H, N, N, D, g)
    plain, plain2 = _run_attn(q, k, vt, nkp, B, H, N, N, D), _run_attn(q, k, vt, nkp, B, H, N, N, D)
    if B == 1:
        general = _run_attn(q, k, vt, nkp, B, H, N, N, D, kv_seg=N, k_seg_stride=N * H * D, vt_seg_stride=nkp)
    else:   # segments are per batch item: compare item by item
        general = torch.cat([_run_attn(q[b:b + 1], k[b:b + 1], vt[:, b * nkp:(b + 1) * nkp], nkp, 1, H, N, N, D, kv_seg=N, k_seg_stride=N * H * D,
                                       vt_seg_stride=nkp) for b in range(B)], 0)
    same = bool(torch.equal(plain, general))
    parity("attention_plain_vs_general_kernel", B=B, H=H, N=N, bit_identical=same, deterministic=bool(torch.equal(plain, plain2)))
    assert torch.equal(plain, plain2)
    assert same, f"{int((plain != general).sum())} of {plain.numel()} outputs differ"


@pytest.mark.parametrize("B,H,Nq,Nk,S", [(1, 12, 1024, 4096, 8), (2, 12, 512, 4096, 5), (1, 40, 1000, 4096 + 37, 3), (2, 3, 130, 200, 4)])
def test_attention_key_split_matches_unsplit_and_fp32(hip_lib, parity, B, H, Nq, Nk, S):
    """kv_split (sequence-parallel shards: few query rows against all keys): partial softmaxes over S key ranges merged by the second
    launch - against fp32 attention and against the unsplit kernel (bf16 rounding of the output apart), run to run deterministic,
    ragged key counts and a split count that does not divide the tile count included."""
    D = 128
    g = torch.Generator(device=dev).manual_seed(Nq + S)
    q, k, v, vt, nkp = _attn_inputs(B, H, Nq, Nk, D, g)
    one = _run_attn(q, k, vt, nkp, B, H, Nq, Nk, D)
    a = _run_attn(q, k, vt, nkp, B, H, Nq, Nk, D, kv_split=S)
    b = _run_attn(q, k, vt, nkp, B, H, Nq, Nk, D, kv_split=S)
    ref = _attn_ref(q, k, v, B, H, D)
    r, r1 = relerr(a, ref), relerr(a, one)
    parity("attention_hd128_key_split", B=B, H=H, Nq=Nq, Nk=Nk, S=S, rel_vs_fp32=r, rel_vs_unsplit=r1)
    assert torch.equal(a, b)
    assert r < 5e-3 and r1 < 6e-3, (r, r1)   # two bf16-P flash results differ by ~sqrt(2) x their own 2.3e-3


def test_attention_hd128_cross_attention_merged_padding_keys(hip_lib, parity):
    """Cross-attention as run in production: 88 keys (87 real + one merged padding key carrying log(count) as key bias)."""
    B, H, Nq, Nk, D = 2, 12, 4096, 88, 128
    g = torch.Generator(device=dev).manual_seed(12)
    q, k, v, vt, nkp = _attn_inputs(B, H, Nq, Nk, D, g)
    kb = torch.zeros(B, nkp, device=dev)
    kb[:, Nk - 1] = math.log(512 - (Nk - 1))
    out = _run_attn(q, k, vt, nkp, B, H, Nq, Nk, D, key_bias=kb, key_bias_first=Nk - 1)
    r, re = relerr(out, _attn_ref(q, k, v, B, H, D, bias=kb[:, :Nk])), relerr(out, _attn_emu(q, k, v, B, H, D, bias=kb[:, :Nk]))
    parity("attention_hd128_cross_keybias", B=B, H=H, Nq=Nq, Nk=Nk, rel_vs_fp32=r, rel_vs_contract=re)
    assert r < TOL_ATTN_EXACT and re < TOL_ATTN_CONTRACT, (r, re)


def test_attention_hd64_global_attention_production_mask(hip_lib, parity):
    """Reconstruction global attention at 13 views: one batch over 13 x 1032 padded rows, only (row % 1032) < 1029 are keys
    (the kv_period > 64 branch of the kernel, previously oracle-checked only at period 16)."""
    S, Pp, P, H, D = 13, 1032, 1029, 16, 64
    N = S * Pp
    g = torch.Generator(device=dev).manual_seed(13)
    q, k, v, vt, nkp = _attn_inputs(1, H, N, N, D, g)
    out = _run_attn(q, k, vt, nkp, 1, H, N, N, D, kv_period=Pp, kv_valid=P)
    mask = (torch.arange(N, device=dev) % Pp) < P
    ref = _attn_ref(q, k, v, 1, H, D, mask=mask)
    r, re = relerr(out[:, mask], ref[:, mask]), relerr(out[:, mask], _attn_emu(q, k, v, 1, H, D, mask=mask)[:, mask])
    parity("attention_hd64_global_masked", S=S, Pp=Pp, valid=P, H=H, rel_vs_fp32=r, rel_vs_contract=re)
    print(f"hd64 global masked rel vs exact softmax {r:.2e}, vs contract {re:.2e}")
    assert r < TOL_ATTN_EXACT and re < TOL_ATTN_CONTRACT, (r, re)


def test_attention_hd64_frame_attention_production_shape(hip_lib, parity):
    S, P, H, D = 13, 1029, 16, 64
    g = torch.Generator(device=dev).manual_seed(14)
    q, k, v, vt, nkp = _attn_inputs(S, H, P, P, D, g)
    out = _run_attn(q, k, vt, nkp, S, H, P, P, D)
    r, re = relerr(out, _attn_ref(q, k, v, S, H, D)), relerr(out, _attn_emu(q, k, v, S, H, D))
    parity("attention_hd64_frame", B=S, H=H, N=P, rel_vs_fp32=r, rel_vs_contract=re)
    assert r < TOL_ATTN_EXACT and re < TOL_ATTN_CONTRACT, (r, re)


def test_attention_spiked_scores_force_running_max_jumps(hip_lib, parity):
    """A few keys aligned with a query make the running max jump by > 50 between 64-key tiles (rescale branch)."""
    B, H, Nq, Nk, D = 1, 2, 256, 1024, 128
    g = torch.Generator(device=dev).manual_seed(15)
    q, k, v, vt, nkp = _attn_inputs(B, H, Nq, Nk, D, g)
    k[0, 700, :D] = (q[0, 5, :D].float() * 4).to(bf16)
    k[0, 70, D:] = (q[0, 77, D:].float() * 3).to(bf16)
    out = _run_attn(q, k, vt, nkp, B, H, Nq, Nk, D)
    r, re = relerr(out, _attn_ref(q, k, v, B, H, D)), relerr(out, _attn_emu(q, k, v, B, H, D))
    parity("attention_spiked", rel_vs_fp32=r, rel_vs_contract=re)
    assert r < TOL_ATTN_EXACT and re < TOL_ATTN_CONTRACT, (r, re)


# ---------------------------------------------------------------- convolution ----
CONV_CASES = [
    # name, Cin, Cout, k, T,H,W, stride, pad(leading T,H,W), causal, ups2, replicate, zero_trailing_only
    ("causal3x3x3_96", 96, 96, (3, 3, 3), 5, 24, 20, (1, 1, 1), (2, 1, 1), True, False, False),
    ("causal3x3x3_384_192", 384, 192, (3, 3, 3), 3, 16, 16, (1, 1, 1), (2, 1, 1), True, False, False),
    ("conv1x1_192_384", 192, 384, (1, 1, 1), 2, 16, 16, (1, 1, 1), (0, 0, 0), False, False, False),
    ("ups2_conv3x3_192_96", 192, 96, (1, 3, 3), 3, 16, 12, (1, 1, 1), (0, 1, 1), False, True, False),
    ("timeconv_384_768", 384, 768, (3, 1, 1), 4, 8, 8, (1, 1, 1), (2, 0, 0), True, False, False),
    ("stitch_replicate_16_1024", 16, 1024, (5, 3, 3), 13, 16, 16, (1, 2, 2), (2, 1, 1), False, False, True),
    ("conv7x7_3_128", 3, 128, (1, 7, 7), 2, 28, 28, (1, 1, 1), (0, 3, 3), False, False, False),
    ("conv3x3_96_3", 96, 3, (3, 3, 3), 3, 32, 32, (1, 1, 1), (2, 1, 1), True, False, False),
    ("conv3x3_stride2_256", 256, 256, (1, 3, 3), 2, 32, 32, (1, 2, 2), (0, 1, 1), False, False, False),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_variants_match_torch_conv3d(hip_lib, parity, case):
    from vist3a_amd import ops
    F = torch.nn.functional
    (name, Cin, Cout, k, T, H, W, st, pd, causal, ups2, repl) = case
    g = torch.Generator(device=dev).manual_seed(9)
    w = torch.randn(Cout, Cin, *k, device=dev, generator=g) / math.sqrt(Cin * k[0] * k[1] * k[2])
    b = torch.randn(Cout, device=dev, generator=g)
    x = torch.randn(1, Cin, T, H, W, device=dev, generator=g).to(bf16)
    cw = ops.ConvWeight(w.to(bf16), b)
    xcl = torch.zeros(T, H, W, cw.CinP, device=dev, dtype=bf16)
    xcl[..., :Cin] = x[0].permute(1, 2, 3, 0)
    xin = x.float()
    if ups2:
        xin = F.interpolate(xin.transpose(1, 2).reshape(T, Cin, H, W), scale_factor=2.0, mode="nearest-exact").view(1, T, Cin, 2 * H, 2 * W).transpose(1, 2)
    mode = "replicate" if repl else "constant"
    xp = F.pad(xin, (pd[2], pd[2], pd[1], pd[1], pd[0], 0 if causal else pd[0]), mode=mode)
    ref = F.conv3d(xp, w.to(bf16).float(), b, stride=st)
    res = torch.randn(ref.shape[2], ref.shape[3], ref.shape[4], cw.CoutP, device=dev, generator=g).to(bf16)
    y = ops.conv(xcl, cw, stride=st, pad=pd, ups2=ups2, replicate=repl, residual=res)
    torch.cuda.synchronize()
    refcl = (ref[0].permute(1, 2, 3, 0).to(bf16).float() + res[..., :Cout].float()).to(bf16)
    r = relerr(y[..., :Cout], refcl)
    parity("conv_variant", name=name, rel_vs_fp32=r)
    assert tuple(y.shape[:3]) == tuple(ref.shape[2:])
    assert r < 1.5e-4, (name, r)   # measured <= 6.2e-5
    if cw.CoutP != Cout:
        assert (y[..., Cout:].float() - res[..., Cout:].float()).abs().max() == 0


HALO_CASES = [
    # name, Cin, Cout, kT, T, H, W (output extent), ups2, act / residual variants follow the VAE decoder's layers
    ("halo_causal3x3x3_96", 96, 96, 3, 4, 32, 64, False),        # the 512^2 stage's layer at a small extent: all three dt cases (t = 0, 1, >= 2)
    ("halo_causal3x3x3_192", 192, 192, 3, 3, 32, 32, False),     # two N tiles, four channel chunks
    ("halo_ups2_conv3x3_192_96", 192, 96, 1, 3, 64, 64, True),   # WanResample: nearest-exact 2x + Conv2d, stored input 32 x 32
    ("halo_conv3x3_384_192", 384, 192, 1, 2, 16, 32, True),      # single tile rows / columns: every halo edge is an image edge
    ("halo_one_frame_96", 96, 96, 3, 1, 16, 32, False),          # T = 1: only dt = 2 exists (one step per chunk)
]


@pytest.mark.parametrize("case", HALO_CASES, ids=[c[0] for c in HALO_CASES])
def test_conv_halo_tile_kernel_matches_torch_and_implicit_gemm(hip_lib, parity, case):
    """csrc/conv_halo.hip (input patch staged once in LDS, nine taps as shifted views) forced with tile=-2: against torch conv3d in fp32 on
    the same bf16 operands (same bar as the implicit GEMM), against the implicit-GEMM kernel (tile=-3: differs only in fp32 summation
    order), with bias + residual fused, and run to run."""
    from vist3a_amd import ops
    F = torch.nn.functional
    name, Cin, Cout, kT, T, H, W, ups2 = case
    g = torch.Generator(device=dev).manual_seed(31)
    w = torch.randn(Cout, Cin, kT, 3, 3, device=dev, generator=g) / math.sqrt(Cin * kT * 9)
    b = torch.randn(Cout, device=dev, generator=g)
    sh, sw = (H // 2, W // 2) if ups2 else (H, W)
    x = torch.randn(1, Cin, T, sh, sw, device=dev, generator=g).to(bf16)
    cw = ops.ConvWeight(w.to(bf16), b)
    assert cw.w_halo is not None
    xcl = x[0].permute(1, 2, 3, 0).contiguous()
    xin = x.float()
    if ups2:
        xin = F.interpolate(xin.transpose(1, 2).reshape(T, Cin, sh, sw), scale_factor=2.0, mode="nearest-exact").view(1, T, Cin, H, W).transpose(1, 2)
    ref = F.conv3d(F.pad(xin, (1, 1, 1, 1, kT - 1, 0)), w.to(bf16).float(), b)
    res = torch.randn(T, H, W, Cout, device=dev, generator=g).to(bf16)
    pad = (kT - 1, 1, 1)
    y = ops.conv(xcl, cw, pad=pad, ups2=ups2, residual=res, tile=-2)
    y2 = ops.conv(xcl, cw, pad=pad, ups2=ups2, residual=res, tile=-2)
    yi = ops.conv(xcl, cw, pad=pad, ups2=ups2, residual=res, tile=-3)
    torch.cuda.synchronize()
    refcl = (ref[0].permute(1, 2, 3, 0).to(bf16).float() + res.float()).to(bf16)
    r, ri = relerr(y, refcl), relerr(y, yi)
    parity("conv_halo", name=name, rel_vs_fp32=r, rel_vs_implicit_gemm=ri)
    assert tuple(y.shape) == (T, H, W, Cout) and torch.equal(y, y2)
    assert r < 1.5e-4 and ri < 1.5e-4, (name, r, ri)
    # SiLU-free activation path of the epilogue + no residual (conv1 of a residual block feeds rownorm, conv_out has neither)
    y3 = ops.conv(xcl, cw, pad=pad, ups2=ups2, tile=-2)
    r3 = relerr(y3, ref[0].permute(1, 2, 3, 0).to(bf16))
    assert r3 < 1.5e-4, (name, r3)


def test_conv_halo_dispatch_rules(hip_lib):
    """v3a_conv_bf16 takes the halo kernel only for layers of its form that fill the chip (>= 512 tiles), tile=-3 forbids it, tile=-2 on a
    layer that is not of its form is an error, and a weight without the second packing never takes it."""
    import ctypes as C
    from vist3a_amd import lib as L, ops
    g = torch.Generator(device=dev).manual_seed(5)
    w = torch.randn(96, 96, 3, 3, 3, device=dev, generator=g) * 0.05
    cw = ops.ConvWeight(w.to(bf16), None)
    x = torch.randn(2, 20, 36, 96, device=dev, generator=g).to(bf16)     # 20 x 36: not a multiple of 16 x 32
    with pytest.raises(RuntimeError):
        ops.conv(x, cw, pad=(2, 1, 1), tile=-2)
    ops.conv(x, cw, pad=(2, 1, 1))                                          # falls back to the implicit GEMM
    w5 = torch.randn(96, 96, 1, 5, 5, device=dev, generator=g) * 0.05
    assert ops.ConvWeight(w5.to(bf16), None).w_halo is None
    w64 = torch.randn(64, 96, 1, 3, 3, device=dev, generator=g) * 0.05
    assert ops.ConvWeight(w64.to(bf16), None).w_halo is None


def test_conv_stride2_trailing_zero_pad_only(hip_lib, parity):
    """WanResample downsample2d: ZeroPad2d((0,1,0,1)) + stride-2 conv = out_size override, taps past the edge read the zero page."""
    from vist3a_amd import ops
    F = torch.nn.functional
    g = torch.Generator(device=dev).manual_seed(21)
    Cin = Cout = 96
    T, H, W = 3, 32, 48
    w = torch.randn(Cout, Cin, 1, 3, 3, device=dev, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, device=dev, generator=g)
    x = torch.randn(1, Cin, T, H, W, device=dev, generator=g).to(bf16)
    cw = ops.ConvWeight(w.to(bf16), b)
    xcl = x[0].permute(1, 2, 3, 0).contiguous()
    y = ops.conv(xcl, cw, stride=(1, 2, 2), pad=(0, 0, 0), out_size=(T, H // 2, W // 2))
    ref = F.conv3d(F.pad(x.float(), (0, 1, 0, 1, 0, 0)), w.to(bf16).float(), b, stride=(1, 2, 2))
    r = relerr(y, ref[0].permute(1, 2, 3, 0).to(bf16))
    parity("conv_stride2_zero_page", rel_vs_fp32=r)
    assert r < 3e-4, r


def test_stitching_conv_matches_reference_golden(hip_lib, parity):
    """The HIP T-upsample + replicate-padded stitching Conv3d directly against tests/golden/stitch_tiny.safetensors, which holds
    the output of the reference's own `parse_conv_spec(...).build()` layer (tests/golden/make_golden.py::stitch_tiny)."""
    from vist3a_amd import ops
    gold = load_file(str(G / "stitch_tiny.safetensors"))
    lat, w, b, ref = gold["latent"], gold["weight"], gold["bias"], gold["out"]
    lat_cl = ops.latent_upsample_t_cl(lat[0].to(dev).contiguous())       # [T', h, w, 16] bf16
    cw = ops.ConvWeight(w, b)
    y = ops.conv(lat_cl, cw, stride=(1, 2, 2), pad=(2, 1, 1), replicate=True, out_f32=True)
    torch.cuda.synchronize()
    got = y.permute(3, 0, 1, 2)[None].float().cpu()
    r = relerr(got, ref)
    parity("stitch_conv_vs_reference_golden", rel_vs_reference=r)
    print(f"stitch conv vs reference golden rel {r:.2e}")
    assert got.shape == ref.shape
    assert r < 6e-3, r   # bf16 latent clip and bf16 weights against the reference's fp32 layer


def test_dilated_and_grouped_stitching_conv_matches_reference_golden(hip_lib, parity):
    """/root/reference/models/stitching_layer_builder.py:21-42 builds dilated (`_d..` of the spec grammar) and grouped (`build(groups=)`) layers;
    tests/golden/stitch_dilated_grouped.safetensors holds the outputs of two such layers of the reference's own builder.  Ours: the same
    spec strings through `parse_conv_spec(..).build(..)`, the checkpoint's [Cout, Cin / groups, *k] parameters assigned into the holder, the
    packed weight the stitched model makes of it (dilated taps in the K-chunk table, block-diagonal dense weight), v3a_conv_bf16."""
    from vist3a_amd import ops
    from vist3a_amd.models.stitching_layer_builder import parse_conv_spec
    gold = load_file(str(G / "stitch_dilated_grouped.safetensors"))
    lat_cl = ops.latent_upsample_t_cl(gold["latent"][0].to(dev).contiguous())
    errs = {}
    for name, spec_s, groups in (("dil", "conv3d_k3x3x3_o64_s1x2x2_p1x2x2_d1x2x2", 1), ("dilgrp", "conv3d_k3x3x3_o64_s1x1x1_p2x2x2_d2x2x2", 4)):
        layer = parse_conv_spec(spec_s).build(in_channels=16, groups=groups)
        assert tuple(layer.weight.shape) == tuple(gold[f"{name}_weight"].shape)              # nn.Conv3d's parameter layout
        layer.weight.data, layer.bias.data = gold[f"{name}_weight"], gold[f"{name}_bias"]
        cw = ops.ConvWeight(layer.dense_weight(), layer.bias.detach(), dilation=layer.dilation3)
        y = ops.conv(lat_cl, cw, stride=layer.stride3, pad=layer.padding3, replicate=True, out_f32=True)
        got = y.permute(3, 0, 1, 2)[None].float().cpu()
        assert got.shape == gold[f"{name}_out"].shape, (got.shape, gold[f"{name}_out"].shape)
        errs[name] = relerr(got, gold[f"{name}_out"])
    parity("stitch_conv_dilated_grouped_vs_reference_golden", **errs)
    print("dilated / grouped stitching conv vs reference golden", errs)
    assert max(errs.values()) < 6e-3, errs   # bf16 latent clip and bf16 weights against the reference's fp32 layer (stitch_tiny: the same bar)


# ---------------------------------------------------------------- resize / norms ----
def test_bilinear_cl_512_to_448_matches_interpolate(hip_lib, parity):
    """V8: F.interpolate(size=448, mode="bilinear", align_corners=False) on the decoded 13 x 512^2 clip (t23d.py)."""
    from vist3a_amd import ops
    g = torch.Generator(device=dev).manual_seed(17)
    x = (torch.rand(13, 512, 512, 8, device=dev, generator=g) * 2 - 1).to(bf16)
    y = ops.bilinear_cl(x, (448, 448), align_corners=False)
    ref = torch.nn.functional.interpolate(x.float().permute(0, 3, 1, 2), size=(448, 448), mode="bilinear", align_corners=False)
    ref = ref.permute(0, 2, 3, 1)
    r = relerr(y, ref.to(bf16))
    mx = (y.float() - ref).abs().max().item()
    parity("bilinear_cl_512_448", rel_vs_fp32=r, max_abs=mx)
    assert tuple(y.shape) == (13, 448, 448, 8)
    assert r < 1e-3 and mx < 8e-3, (r, mx)   # one bf16 ulp at |x| <= 1 is 3.9e-3
    # align_corners=True (DPT head) on an odd size
    x2 = torch.randn(2, 37, 37, 16, device=dev, generator=g).to(bf16)
    y2 = ops.bilinear_cl(x2, (74, 74), align_corners=True)
    ref2 = torch.nn.functional.interpolate(x2.float().permute(0, 3, 1, 2), size=(74, 74), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    r2 = relerr(y2, ref2.to(bf16))
    parity("bilinear_cl_align_corners", rel_vs_fp32=r2)
    assert r2 < 1e-3, r2


def test_norm_kernels_production_shapes(hip_lib, parity):
    from vist3a_amd import ops
    g = torch.Generator(device=dev).manual_seed(4)
    for (M, d, rpb) in [(8192, 1536, 4096), (13 * 1032, 1024, 0), (100, 2048, 50), (64, 5120, 32)]:
        x = (torch.randn(M, d, device=dev, generator=g) * 2 + 0.5).to(bf16)
        nb = (M + rpb - 1) // rpb if rpb else 1
        sc = torch.randn(nb, d, device=dev, generator=g) * 0.3
        sh = torch.randn(nb, d, device=dev, generator=g) * 0.3
        w = torch.randn(d, device=dev, generator=g)
        b = torch.randn(d, device=dev, generator=g)
        y = ops.layernorm(x, scale=sc, shift=sh, rows_per_batch=rpb or M, eps=1e-6)
        ln = torch.nn.functional.layer_norm(x.float(), (d,), eps=1e-6)
        idx = torch.arange(M, device=dev) // (rpb or M)
        r = relerr(y, (ln * (1 + sc[idx]) + sh[idx]).to(bf16))
        parity("layernorm_adaln", M=M, d=d, rel_vs_fp32=r)
        assert r < 4e-5, r   # measured 1.5e-5 (bf16 output rounding flips)
        xf = x.float() * 1.37
        y = ops.layernorm(xf, weight=w, bias=b, eps=1e-5, out_dtype=f32)
        r = relerr(y, torch.nn.functional.layer_norm(xf, (d,), w, b, eps=1e-5))
        parity("layernorm_affine_f32", M=M, d=d, rel_vs_fp32=r)
        assert r < 1e-5, r
    for (B, N, H, hd) in [(2, 4096, 12, 128), (1, 300, 3, 128)]:
        d, M = H * hd, B * N
        x = torch.randn(M, d, device=dev, generator=g).to(bf16)
        w = torch.randn(d, device=dev, generator=g)
        ang = torch.rand(N, hd // 2, device=dev, generator=g, dtype=torch.float64) * 6.28
        rope = torch.stack([ang.cos(), ang.sin()], -1).float().contiguous()
        y = ops.rmsnorm_rope(x, w, rope=rope, head_dim=hd, tokens_per_batch=N, eps=1e-6)
        xf = x.float()
        n = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * w
        nc = torch.view_as_complex(n.double().view(B, N, H, hd // 2, 2))
        fc = torch.polar(torch.ones_like(ang), ang)[None, :, None, :]
        r = relerr(y, torch.view_as_real(nc * fc).reshape(M, d).to(bf16))
        parity("rmsnorm_rope", B=B, N=N, rel_vs_fp32=r)
        assert r < 6e-5, r   # measured 2.2e-5
        # q | k of the fused projection in ONE launch (weight2) == the two separate launches
        qk = torch.randn(M, 2 * d, device=dev, generator=g).to(bf16)
        w2 = torch.randn(d, device=dev, generator=g)
        sep = torch.cat([ops.rmsnorm_rope(qk[:, :d], w, rope=rope, head_dim=hd, tokens_per_batch=N, eps=1e-6),
                         ops.rmsnorm_rope(qk[:, d:], w2, rope=rope, head_dim=hd, tokens_per_batch=N, eps=1e-6)], 1)
        both = qk.clone()
        ops.rmsnorm_rope(both, w, out=both, rope=rope, head_dim=hd, tokens_per_batch=N, eps=1e-6, weight2=w2)
        assert torch.equal(both, sep)
        # ... and the e4m3 operand form (fp8 attention): == the bf16 result pushed through v3a_quantize_fp8
        q8 = torch.empty(M, 2 * d, device=dev, dtype=torch.uint8)
        ops.rmsnorm_rope(qk, w, out=q8, rope=rope, head_dim=hd, tokens_per_batch=N, eps=1e-6, weight2=w2, fp8_scale=0.5)
        assert torch.equal(q8, ops.quantize_fp8(sep, 0.5))


@pytest.mark.parametrize("spread", [0.03, 0.25], ids=["7_points_per_voxel", "1_point_per_voxel"])
def test_voxel_fusion_production_size_matches_scatter_reference(hip_lib, parity, spread):
    """R13 at 13 x 448^2 points x 84 columns: voxel ids bit-exact vs torch.unique on the rounded coordinates; fused features and
    positions vs a torch scatter-softmax (scatter_max / exp / scatter_add, anysplat.py:298-335) to fp32 round-off."""
    from vist3a_amd import ops
    g = torch.Generator(device=dev).manual_seed(3)
    M, C = 13 * 448 * 448, 83
    pts = (torch.randn(M, 3, device=dev, generator=g) * spread).contiguous()
    feat = torch.randn(M, C + 1, device=dev, generator=g).contiguous()
    v = ops.voxelize_fuse(pts, feat, C, C, 0.002)
    U, inv = v["keys"].shape[0], v["inverse"].long()
    # tensor / tensor = IEEE division, what the reference's CPU path computes (torch's GPU tensor / python-scalar kernel multiplies by
    # the reciprocal instead and differs in the last bit for a handful of points)
    keys = (pts / torch.full_like(pts, 0.002)).round().int()
    uq, uinv, ucnt = torch.unique(keys, dim=0, return_inverse=True, return_counts=True)
    assert torch.equal(v["keys"], uq) and torch.equal(inv, uinv) and torch.equal(v["counts"].long(), ucnt)
    conf = feat[:, C]
    mx = torch.full((U,), -float("inf"), device=dev).scatter_reduce(0, inv, conf, "amax")
    ex = torch.exp(conf - mx[inv])
    w = ex / (torch.zeros(U, device=dev).index_add_(0, inv, ex) + 1e-6)[inv]
    rf = torch.zeros(U, C, device=dev).index_add_(0, inv, feat[:, :C] * w[:, None])
    rp = torch.zeros(U, 3, device=dev).index_add_(0, inv, pts * w[:, None])
    ef, ep = relerr(v["voxel_feat"][:, :C], rf), relerr(v["voxel_pts"], rp)
    parity("voxel_fusion_production", M=M, U=U, rel_feat=ef, rel_pts=ep)
    assert ef < 1e-6 and ep < 1e-6, (ef, ep)
    again = ops.voxelize_fuse(pts, feat, C, C, 0.002)
    assert torch.equal(again["voxel_feat"], v["voxel_feat"]) and torch.equal(again["voxel_pts"], v["voxel_pts"])   # fixed summation order


# ---------------------------------------------------------------- fp8 attention (BASELINE config #4) ----
def _fp8_inputs(B, H, Nq, Nk, g, qk_std=1.0):
    D = 128
    q = (torch.randn(B, Nq, H * D, device=dev, generator=g) * qk_std).to(bf16)
    k = (torch.randn(B, Nk, H * D, device=dev, generator=g) * qk_std).to(bf16)
    v = torch.randn(B, Nk, H * D, device=dev, generator=g).to(bf16)
    nkp = (Nk + 63) // 64 * 64
    vt = torch.zeros(H * D, B * nkp, device=dev, dtype=bf16)
    vt.view(H * D, B, nkp)[:, :, :Nk] = v.permute(2, 0, 1)
    return q, k, v, vt, nkp


@pytest.mark.parametrize("B,H,Nq,Nk", [(1, 2, 256, 320), (2, 3, 200, 333), (1, 40, 512, 512)], ids=["small", "ragged", "14B_heads"])
def test_attention_fp8_matches_e4m3_emulation(hip_lib, parity, B, H, Nq, Nk):
    """The fp8 kernel against the oracle's e4m3 emulation with the SAME rounding points (per-tensor scales, P rounded per 64-key
    tile against the running maximum): agreement to fp32 round-off + bf16 output rounding; and against exact fp32 attention to show
    what the precision mode itself costs."""
    from oracle import wan_dit as O
    from vist3a_amd import ops
    D = 128
    g = torch.Generator(device=dev).manual_seed(31 + Nk)
    q, k, v, vt, nkp = _fp8_inputs(B, H, Nq, Nk, g)
    qs, ks, vs = 0.5, 0.25, 2.0
    q8 = ops.quantize_fp8(q.view(B * Nq, H * D), qs)
    k8 = ops.quantize_fp8(k.view(B * Nk, H * D), ks)
    vt8 = ops.quantize_fp8(vt, vs)
    ref8 = q.view(B * Nq, -1).float().div(qs).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(q8, ref8), "quantiser is not torch's RNE e4m3 rounding"
    out = torch.empty(B * Nq, H * D, device=dev, dtype=bf16)
    ops.attention_fp8(q8, k8, vt8, out, B=B, H=H, Nq=Nq, Nk=Nk, q_batch_stride=Nq * H * D, k_batch_stride=Nk * H * D,
                      vt_batch_stride=nkp, o_batch_stride=Nq * H * D, q_scale=qs, k_scale=ks, v_scale=vs)
    torch.cuda.synchronize()
    out = out.view(B, Nq, H, D).float().cpu()
    sh = lambda t, n: t.float().cpu().view(B, n, H, D).transpose(1, 2)
    emu = O.attention_fp8_emulated(sh(q, Nq), sh(k, Nk), sh(v, Nk), D ** -0.5, qs, ks, vs).transpose(1, 2)
    exact = torch.softmax(sh(q, Nq) @ sh(k, Nk).transpose(-1, -2) * D ** -0.5, -1) @ sh(v, Nk)
    r_emu, r_exact = relerr(out, emu), relerr(out, exact.transpose(1, 2))
    parity("attention_fp8", B=B, H=H, Nq=Nq, Nk=Nk, rel_vs_e4m3_emulation=r_emu, rel_vs_exact_fp32=r_exact)
    print(f"fp8 attention: rel vs e4m3 emulation {r_emu:.2e}, vs exact fp32 {r_exact:.2e}")
    assert r_emu < 3e-3, r_emu          # bf16 output rounding + exp2 ulp flips at e4m3 rounding boundaries of P
    assert r_exact < 6e-2, r_exact      # what e4m3 operands cost (3 mantissa bits)


@pytest.mark.parametrize("B,H,Nq,Nk,S", [(1, 40, 512, 4096, 5), (2, 4, 300, 1000, 3)])
def test_attention_fp8_key_split_and_slabs(hip_lib, parity, B, H, Nq, Nk, S):
    """The e4m3 kernel on a sequence-parallel shard's launch: kv_split merges partial softmaxes (deterministic; within the e4m3 P
    rounding of the unsplit kernel - a split changes which running maximum P is rounded against), and kv_seg reads K / V^T from
    per-rank slabs bit-identically to the contiguous layout."""
    from vist3a_amd import ops
    D = 128
    g = torch.Generator(device=dev).manual_seed(S + Nk)
    q, k, v, vt, nkp = _fp8_inputs(B, H, Nq, Nk, g)
    q8, k8, vt8 = ops.quantize_fp8(q.view(B * Nq, H * D)), ops.quantize_fp8(k.view(B * Nk, H * D)), ops.quantize_fp8(vt)
    kw = dict(B=B, H=H, Nq=Nq, Nk=Nk, q_batch_stride=Nq * H * D, k_batch_stride=Nk * H * D, vt_batch_stride=nkp, o_batch_stride=Nq * H * D)
    run = lambda **e: ops.attention_fp8(q8, k8, vt8, torch.empty(B * Nq, H * D, device=dev, dtype=bf16), **kw, **e)
    one, a, b = run(), run(kv_split=S), run(kv_split=S)
    exact = _attn_ref(q, k, v, B, H, D).reshape(B * Nq, H * D)
    r, re1, reS = relerr(a, one), relerr(one, exact), relerr(a, exact)
    parity("attention_fp8_key_split", B=B, H=H, Nq=Nq, Nk=Nk, S=S, rel_vs_unsplit=r, unsplit_vs_fp32=re1, split_vs_fp32=reS)
    # P is rounded to e4m3 against the running maximum of ITS key range, so split and unsplit round differently: each stays at the
    # e4m3 mode's distance from exact attention (5e-2), and they differ from each other by about the same
    assert torch.equal(a, b) and reS < 6e-2 and reS < 1.2 * re1 and r < 6e-2, (r, re1, reS)
    if Nk % 128 == 0 and B == 1:   # two slabs of Nk / 2 keys each, laid out like the all-gather buffer: [K_0 | VT_0][K_1 | VT_1]
        seg, d = Nk // 2, H * D
        slab = torch.empty(2, 2 * seg * d, device=dev, dtype=torch.uint8)
        for r_ in range(2):
            slab[r_, :seg * d] = k8[r_ * seg:(r_ + 1) * seg].reshape(-1)
            slab[r_, seg * d:] = vt8[:, r_ * seg:(r_ + 1) * seg].reshape(-1)
        o = torch.empty(B * Nq, d, device=dev, dtype=bf16)
        ops.attention_fp8(q8, slab[0, :seg * d].view(seg, d), slab[0, seg * d:].view(d, seg), o, B=B, H=H, Nq=Nq, Nk=Nk, q_batch_stride=Nq * d,
                          k_batch_stride=seg * d, vt_batch_stride=seg, o_batch_stride=Nq * d, kv_seg=seg, k_seg_stride=2 * seg * d,
                          vt_seg_stride=2 * seg * d)
        assert torch.equal(o, one)


def test_attention_fp8_production_shape_and_speed(hip_lib, parity):
    """Wan-14B self-attention launch (B=1 per CFG branch, 40 heads, 4096 x 4096): finite, deterministic, close to bf16 attention."""
    from vist3a_amd import ops
    B, H, N, D = 1, 40, 4096, 128
    g = torch.Generator(device=dev).manual_seed(77)
    q, k, v, vt, nkp = _fp8_inputs(B, H, N, N, g)
    q8, k8, vt8 = ops.quantize_fp8(q.view(N, H * D)), ops.quantize_fp8(k.view(N, H * D)), ops.quantize_fp8(vt)
    outs = []
    for _ in range(2):
        o = torch.empty(N, H * D, device=dev, dtype=bf16)
        ops.attention_fp8(q8, k8, vt8, o, B=B, H=H, Nq=N, Nk=N, q_batch_stride=N * H * D, k_batch_stride=N * H * D, vt_batch_stride=nkp,
                          o_batch_stride=N * H * D)
        outs.append(o)
    ob = _run_attn(q, k, vt, nkp, B, H, N, N, D).view(N, H * D)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.isfinite(outs[0].float()).all()
    r = relerr(outs[0], ob)
    parity("attention_fp8_vs_bf16_kernel_14B_shape", rel=r)
    assert r < 6e-2, r


# --------------------------------------------------------------------------------------------------------------------
# e4m3 GEMM (BASELINE config #4: "MFMA bf16/fp8 GEMMs for the attention/FFN contractions")
def _quant_rows_emu(x):
    """torch restatement of v3a_quantize_fp8_rows (a tensor / python-scalar division would multiply by a reciprocal on the GPU)."""
    amax = x.float().abs().amax(dim=1)
    sc = amax.clamp_min(1e-12) / torch.full_like(amax, 448.0)
    q = (x.float() * (torch.ones_like(sc) / sc)[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q, sc


def test_quantize_fp8_rows_is_bit_exact(hip_lib):
    from vist3a_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    for rows, cols in ((1000, 5120), (257, 13824), (64, 128), (5, 8200)):
        x = (torch.randn(rows, cols, device=dev, generator=g) * torch.rand(rows, 1, device=dev, generator=g) * 8).to(bf16)
        x[rows // 2] = 0          # an all-zero row quantises to zeros with the floor scale
        x[0, 3] = 3.0e4           # one outlier sets its row's scale
        q, sc = ops.quantize_fp8_rows(x)
        rq, rsc = _quant_rows_emu(x)
        assert torch.equal(sc, rsc) and torch.equal(q, rq.view(torch.uint8)), (rows, cols)
        assert float(sc[rows // 2]) == pytest.approx(1e-12 / 448.0) and int(q[rows // 2].max()) == 0


@pytest.mark.parametrize("name,M,N,K,opt", [
    ("14b_ffn1", 8192, 13824, 5120, dict(act="gelu")),
    ("14b_ffn2", 8192, 5120, 13824, dict(res_f32=False, scale=True)),
    ("14b_qk", 8192, 10240, 5120, {}),
    ("14b_vt", 5120, 8192, 5120, dict(bias_row=True)),
    ("ragged", 1000, 520, 384, dict(act="gelu")),
])
def test_gemm_fp8_every_tile_matches_e4m3_emulation(hip_lib, parity, name, M, N, K, opt):
    """v3a_gemm_fp8_nt against the same arithmetic in torch (e4m3 values held in fp32, fp32 matmul, scale product, the bf16 GEMM's
    epilogue): every tile within bf16 output rounding of it and bit-identical to the others."""
    from vist3a_amd import lib as L, ops
    lib = L.load()
    g = torch.Generator(device=dev).manual_seed(K + N)
    a = torch.randn(M, K, device=dev, generator=g).to(bf16)
    w = (torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).to(bf16)
    bias_row = opt.get("bias_row", False)
    bias = torch.randn(M if bias_row else N, device=dev, generator=g)
    act = L.ACT_GELU_TANH if opt.get("act") == "gelu" else L.ACT_NONE
    res = torch.randn(M, N, device=dev, generator=g).to(bf16) if "res_f32" in opt else None
    scale = torch.randn(2, N, device=dev, generator=g) if opt.get("scale") else None
    rpb = M // 2 if scale is not None else 0
    a8, sa = ops.quantize_fp8_rows(a)
    w8, sw = ops.quantize_fp8_rows(w)
    qa, qw = a8.view(torch.float8_e4m3fn).float(), w8.view(torch.float8_e4m3fn).float()
    v = (qa @ qw.T) * (sa[:, None] * sw[None, :]) + (bias[:, None] if bias_row else bias[None, :])
    v = v.to(bf16).float()
    if act:
        v = torch.nn.functional.gelu(v, approximate="tanh").to(bf16).float()
    if scale is not None:
        v = v * scale.repeat_interleave(rpb, dim=0)
    if res is not None:
        v = v + res.float()
    ref = v.to(bf16)
    exact = a.float() @ w.float().T
    outs = {}
    for t in range(lib.v3a_gemm_fp8_num_tiles()):
        nm = lib.v3a_gemm_fp8_tile_name(t).decode()
        outs[nm] = ops.gemm(a8, w8, bias, act=act, residual=res, scale=scale, rows_per_batch=rpb, bias_row=bias_row, a_scale=sa,
                            w_scale=sw, tile=t)
    auto = ops.gemm(a8, w8, bias, act=act, residual=res, scale=scale, rows_per_batch=rpb, bias_row=bias_row, a_scale=sa, w_scale=sw)
    first = next(iter(outs.values()))
    worst = max(relerr(o, ref) for o in outs.values())
    quant_cost = relerr((qa @ qw.T) * (sa[:, None] * sw[None, :]), exact)
    parity(f"gemm_fp8_{name}", M=M, N=N, K=K, rel_vs_e4m3_emulation=worst, rel_e4m3_operands_vs_bf16_operands=quant_cost)
    assert all(torch.equal(o, first) for o in outs.values()) and torch.equal(auto, first)
    assert worst < 1.2e-3, worst            # bf16 output rounding flips (fp32 accumulation order differs from torch's)
    assert quant_cost < 5e-2, quant_cost    # what 3 mantissa bits on both operands cost


def test_gemm_fp8_rejects_bad_arguments(hip_lib):
    from vist3a_amd import lib as L, ops
    a8 = torch.zeros(256, 192, device=dev, dtype=torch.uint8)       # K % 128 != 0
    w8 = torch.zeros(64, 192, device=dev, dtype=torch.uint8)
    s = torch.ones(256, device=dev)
    with pytest.raises(RuntimeError):
        ops.gemm(a8, w8, None, a_scale=s, w_scale=s[:64].contiguous())
    with pytest.raises(ValueError):
        ops.gemm(a8[:, :128].contiguous(), w8[:, :128].contiguous(), None, a_scale=s[:5].contiguous(), w_scale=s[:64].contiguous())


def test_layernorm_fp8_output_equals_separate_quantisation_pass(hip_lib):
    """v3a_layernorm with y_fp8_scale: bit-identical to LayerNorm -> bf16 followed by v3a_quantize_fp8_rows (AdaLN and affine forms,
    fp32 and bf16 inputs, a width that is not a multiple of 512)."""
    from vist3a_amd import ops
    g = torch.Generator(device=dev).manual_seed(9)
    for (M, d, rpb, xf32) in [(8192, 5120, 4096, True), (1000, 1536, 500, False), (37, 1000, 37, True)]:
        x = (torch.randn(M, d, device=dev, generator=g) * 3 + 0.25)
        x = x if xf32 else x.to(bf16)
        nb = (M + rpb - 1) // rpb
        sc, sh = torch.randn(nb, d, device=dev, generator=g) * 0.3, torch.randn(nb, d, device=dev, generator=g) * 0.3
        w, b = torch.randn(d, device=dev, generator=g), torch.randn(d, device=dev, generator=g)
        for kw in (dict(scale=sc, shift=sh, rows_per_batch=rpb), dict(weight=w, bias=b)):
            y = ops.layernorm(x, eps=1e-6, **kw)
            q_ref, s_ref = ops.quantize_fp8_rows(y)
            q = torch.empty(M, d, device=dev, dtype=torch.uint8)
            s = torch.empty(M, device=dev)
            ops.layernorm(x, out=q, fp8_scale=s, eps=1e-6, **kw)
            assert torch.equal(s, s_ref) and torch.equal(q, q_ref), (M, d, list(kw))


# ---------------------------------------------------------------- cached-context cross-attention (csrc/xattn_probs.hip) ----
XATTN_CASES = [
    ("dit_1_3b_88_keys", 2, 12, 4096, 88, 96, True),      # production: 80 / 64 real tokens + merged padding key, padded to 96 per head
    ("one_key_tile", 2, 12, 1000, 40, 48, True),          # <= 64 keys: the one-tile instantiation; ragged last query block
    ("full_128_keys", 1, 40, 512, 128, 128, False),       # Wan-14B head count, both key tiles full, no bias
    ("two_heads_pad32", 2, 2, 96, 24, 32, True),          # the test models' geometry (H = 2: keys padded to 32 so that H * Lkp % 64 == 0)
    ("pad_beyond_key_tiles", 1, 5, 200, 10, 64, True),    # an odd head count pads 10 keys to 64: the zero columns lie beyond the last 32-key sub-tile that holds a key
    ("two_units_per_wave", 2, 12, 16424, 73, 80, True),   # > 2048 workgroups of one unit per wave: every wave walks two 32-query units (ragged end)
]


def test_gemm_row_sumsq_by_product_is_tile_independent(hip_lib):
    """v3a_gemm_args.row_sumsq: per (row, 32-column block) sum of squares of bf16(acc + bias), emitted by the epilogue of every tile shape
    with the same bits, and equal to the squares of the stored bf16 output summed in fp32."""
    from vist3a_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    M, N, K = 8192, 1536, 1536
    a = torch.randn(M, K, device=dev, generator=g).to(bf16)
    w = (torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).to(bf16)
    bias = torch.randn(N, device=dev, generator=g)
    names = _tile_names(hip_lib)
    ref_sq = ref_out = None
    for t in range(len(names)):
        sq = torch.zeros(M, N // 32, device=dev)
        out = ops.gemm(a, w, bias, tile=t, row_sumsq=sq)
        if ref_sq is None:
            ref_sq, ref_out = sq, out
            want = out.float().view(M, N // 32, 32).pow(2).sum(-1)
            assert ((sq - want).abs() / want).max().item() < 1e-5
        assert torch.equal(out, ref_out) and torch.equal(sq, ref_sq), names[t]
    assert torch.equal(ops.gemm(a, w, bias), ref_out)     # and the by-product does not disturb the output
    # the statistics describe bf16(acc + bias) only: any epilogue stage behind it is refused, not silently mis-described
    sq = torch.zeros(M, N // 32, device=dev)
    from vist3a_amd import lib as L
    for kw in (dict(residual=ref_out), dict(relu_out=True), dict(act=L.ACT_RELU), dict(out=torch.empty(2 * M, N, device=dev, dtype=bf16), out_rows=(64, 64, 0))):
        with pytest.raises(ValueError):
            ops.gemm(a, w, bias, row_sumsq=sq, **kw)
    with pytest.raises(ValueError):     # batched problems write batch * M rows of statistics
        ops.gemm(a, w, bias, row_sumsq=sq, batch=(2, 0, 0, M * N), out=torch.empty(2 * M, N, device=dev, dtype=bf16))


@pytest.mark.parametrize("name,B,H,Nq,Nk,Lkp,biased", XATTN_CASES, ids=[c[0] for c in XATTN_CASES])
def test_xattn_probs_matches_fp32_softmax(hip_lib, parity, name, B, H, Nq, Nk, Lkp, biased):
    """v3a_xattn_probs_bf16: P[m, h * Lkp + j] = bf16(softmax_j(q_h . k_hj / sqrt(128) + bias_j)) for j < Nk, exact zeros in the padding
    columns, rows of every head summing to 1 within bf16 rounding; against fp32 softmax evaluated by torch on the same bf16 q, k."""
    from vist3a_amd import ops
    D = 128
    g = torch.Generator(device=dev).manual_seed(Nk * 7 + H)
    q = torch.randn(B * Nq, H * D, device=dev, generator=g).to(bf16)
    Lt = Nk + 5                                            # the key buffer has more rows than keys (row stride of the batch item)
    k = torch.randn(B * Lt, H * D, device=dev, generator=g).to(bf16)
    bias = None
    if biased:
        bias = torch.zeros(B, 128, device=dev)
        bias[:, Nk - 1] = math.log(512 - (Nk - 1))         # the merged zero-padding key
    Kp = H * Lkp
    out = torch.full((B * Nq, Kp + 8), 7.0, device=dev, dtype=bf16)   # row stride > H * Lkp; the guard columns must stay untouched
    p = out[:, :Kp]
    ops.xattn_probs(q, k, p, B=B, H=H, Nq=Nq, Nk=Nk, Lkp=Lkp, q_batch_stride=Nq * H * D, k_batch_stride=Lt * H * D,
                    p_batch_stride=Nq * (Kp + 8), key_bias=bias, key_bias_first=Nk - 1)
    torch.cuda.synchronize()
    qf = q.float().view(B, Nq, H, D).transpose(1, 2)
    kf = k.float().view(B, Lt, H, D)[:, :Nk].transpose(1, 2)
    s = qf @ kf.transpose(-1, -2) * D ** -0.5
    if bias is not None:
        s = s + bias[:, None, None, :Nk]
    ref = torch.softmax(s, -1)                              # [B, H, Nq, Nk] fp32
    got = p.float().view(B, Nq, H, Lkp).permute(0, 2, 1, 3)
    assert torch.equal(out[:, Kp:], torch.full_like(out[:, Kp:], 7.0))
    assert (got[..., Nk:] == 0).all()
    r = ((got[..., :Nk] - ref).norm() / ref.norm()).item()
    exact = (got[..., :Nk] == ref.to(bf16).float()).float().mean().item()   # share of elements that ARE the correctly rounded fp32 value
    rowsum = (got.sum(-1) - 1).abs().max().item()
    # q_row_sumsq: the same probabilities from an UN-normalised q plus its row statistics (the RMS factor applied to the scores)
    qs = (q.float() * 3.0).to(bf16)                                               # rows with rms ~3
    sq = qs.float().view(B * Nq, H * D // 32, 32).pow(2).sum(-1).contiguous()
    eps = 1e-6
    out2 = torch.zeros_like(out)
    ops.xattn_probs(qs, k, out2[:, :Kp], B=B, H=H, Nq=Nq, Nk=Nk, Lkp=Lkp, q_batch_stride=Nq * H * D, k_batch_stride=Lt * H * D,
                    p_batch_stride=Nq * (Kp + 8), key_bias=bias, key_bias_first=Nk - 1, q_row_sumsq=sq, q_eps=eps)
    rq = torch.rsqrt(qs.float().pow(2).mean(-1) + eps).view(B, 1, Nq, 1)
    s2 = (qs.float().view(B, Nq, H, D).transpose(1, 2) @ kf.transpose(-1, -2)) * rq * D ** -0.5
    if bias is not None:
        s2 = s2 + bias[:, None, None, :Nk]
    ref2 = torch.softmax(s2, -1)
    got2 = out2[:, :Kp].float().view(B, Nq, H, Lkp).permute(0, 2, 1, 3)
    r2 = ((got2[..., :Nk] - ref2).norm() / ref2.norm()).item()
    parity("xattn_probs", name=name, rel_vs_fp32=r, correctly_rounded_share=exact, max_rowsum_error=rowsum, rel_vs_fp32_folded_q_norm=r2)
    assert r < 3e-3 and exact > 0.999 and rowsum < 5e-3, (r, exact, rowsum)   # measured: rel 1.3-1.7e-3 = the bf16 rounding of P itself; 99.99 % of the elements ARE the correctly rounded fp32 value
    assert r2 < 3e-3 and (got2[..., Nk:] == 0).all(), r2


def test_gemm_batched_operands_equal_separate_launches(hip_lib):
    """v3a_gemm_args.batch: `batch` equally shaped problems in one launch (per-head / per-prompt operands) == the same problems launched one
    by one, bit for bit, with bias + residual epilogue, for a ping-pong tile shape and for the tiny per-head V.Wo^T shape."""
    from vist3a_amd import ops
    g = torch.Generator(device=dev).manual_seed(77)
    # (1) two CFG items, one B operand each: [4096, 1152] x [1536, 1152]^T + bias + residual (the cross-attention's finishing GEMM)
    nb, M, N, K = 2, 4096, 1536, 1152
    a = torch.randn(nb * M, K, device=dev, generator=g).to(bf16)
    w = (torch.randn(nb, N, K, device=dev, generator=g) / math.sqrt(K)).to(bf16)
    bias = torch.randn(N, device=dev, generator=g)
    x = torch.randn(nb * M, N, device=dev, generator=g).to(bf16)
    want = torch.cat([ops.gemm(a[z * M:(z + 1) * M], w[z], bias, residual=x[z * M:(z + 1) * M]) for z in range(nb)], 0)
    got = x.clone()
    ops.gemm(a[:M], w[0], bias, out=got[:M], residual=got[:M], batch=(nb, M * K, N * K, M * N))
    assert torch.equal(got, want)
    # (2) twelve heads: out[:, h * 96:(h + 1) * 96] = Wo[:, h-th 128 columns] . V[:, h-th 128 columns]^T  (column-offset strides, ldc = 12 * 96)
    H, d, Lkp = 12, 1536, 96
    wo = torch.randn(d, d, device=dev, generator=g).to(bf16)
    v = torch.randn(Lkp, d, device=dev, generator=g).to(bf16)
    out = torch.zeros(d, H * Lkp, device=dev, dtype=bf16)
    ops.gemm(wo[:, :128], v[:, :128], out=out[:, :Lkp], batch=(H, 128, 128, Lkp))
    ref = torch.cat([(wo[:, h * 128:(h + 1) * 128].float() @ v[:, h * 128:(h + 1) * 128].float().t()).to(bf16) for h in range(H)], 1)
    one = torch.cat([ops.gemm(wo[:, h * 128:(h + 1) * 128], v[:, h * 128:(h + 1) * 128]) for h in range(H)], 1)
    assert torch.equal(out, one)
    assert ((out.float() - ref.float()).norm() / ref.float().norm()).item() < 3e-3



def test_gemm_transposed_tail_is_bit_identical_to_two_launches(hip_lib):
    """v3a_gemm_args.C_t: the fused q | k | v projection of a DiT block - columns below t_col0 row-major into C, the rest TRANSPOSED into C_t
    (what the flash kernel reads as V^T) - against the q | k GEMM and the V^T = Wv . X^T GEMM (bias by row) it replaces, at the production
    shape (8192 x (3072 + 1536) x 1536: 768 tiles) and at ragged / small ones: bit for bit."""
    from vist3a_amd import ops
    g = torch.Generator(device=dev).manual_seed(17)
    for (M, d, K) in ((8192, 1536, 1536), (4096, 768, 512), (1000, 192, 256), (8, 384, 64)):
        x = torch.randn(M, K, device=dev, generator=g).to(bf16)
        w = (torch.randn(3 * d, K, device=dev, generator=g) / math.sqrt(K)).to(bf16)
        b = torch.randn(3 * d, device=dev, generator=g)
        qk_ref = ops.gemm(x, w[: 2 * d], b[: 2 * d].contiguous())
        vt_ref = torch.zeros(d, M + 64, device=dev, dtype=bf16)
        ops.gemm(w[2 * d:], x, b[2 * d:].contiguous(), out=vt_ref[:, :M], bias_row=True)
        qk = torch.zeros(M, 2 * d, device=dev, dtype=bf16)
        vt = torch.zeros(d, M + 64, device=dev, dtype=bf16)
        ops.gemm(x, w, b, out=qk, t_out=vt, t_col0=2 * d)
        assert torch.equal(qk, qk_ref), (M, d, K)
        assert torch.equal(vt, vt_ref), (M, d, K, (vt.float() - vt_ref.float()).abs().max().item())
    with pytest.raises(ValueError):
        ops.gemm(x, w, b, out=qk, t_out=vt, t_col0=100)          # not whole 192-column tiles
    with pytest.raises(RuntimeError):
        ops.gemm(x, w, b, out=qk, t_out=vt, t_col0=2 * d, act=1)  # bias-only epilogue
