"""Host-side logic that needs no GPU: conv-spec grammar, CLI flags, LoRA merge, ply writer, prompt sharding over a
world_size-2 gloo process group (the only multi-GPU mechanism the reference's inference path has: data parallel prompts)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


def _free_port() -> str:
    """a TCP port the kernel just handed out (fixed rendezvous ports collide with a previous run's sockets in TIME_WAIT)"""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return str(so.getsockname()[1])


def test_conv_spec_grammar():
    from vist3a_amd.models.stitching_layer_builder import ConvSpec, parse_conv_spec
    s = parse_conv_spec("conv3d_k5x3x3_o1024_s1x2x2_p2x1x1")
    assert s == ConvSpec(dim=3, out_channels=1024, kernel_size=(5, 3, 3), stride=(1, 2, 2), padding=(2, 1, 1), dilation=1)
    assert parse_conv_spec("conv2d_k3_o64") == ConvSpec(2, 64, 3, 1, 0, 1)
    assert parse_conv_spec("CONV3D_k3x3x3_o32_s2_p1").stride == 2
    for bad in ("conv4d_k3_o1", "conv3d_o3", "conv3d_k3x3x3", "foo"):
        with pytest.raises(ValueError):
            parse_conv_spec(bad)
    layer = s.build(in_channels=16)
    assert tuple(layer.weight.shape) == (1024, 16, 5, 3, 3) and layer.padding_mode == "replicate"


def test_grouped_dilated_stitching_layer_holder():
    """`ConvSpec.build(in_channels, groups=)` (stitching_layer_builder.py:21-42): nn.Conv's parameter layout [Cout, Cin / groups, *k], its
    divisibility error, and a dense block-diagonal form that is the same linear map as the grouped convolution."""
    from vist3a_amd.models.stitching_layer_builder import parse_conv_spec
    spec = parse_conv_spec("conv3d_k3x3x3_o32_s1x2x2_p2x2x2_d2x2x2")
    assert spec.dilation == (2, 2, 2)
    layer = spec.build(in_channels=16, groups=4)
    assert tuple(layer.weight.shape) == (32, 4, 3, 3, 3) and layer.groups == 4 and layer.dilation3 == (2, 2, 2) and layer.padding_mode == "replicate"
    x = torch.randn(1, 16, 5, 9, 9, generator=torch.Generator().manual_seed(0))
    F = torch.nn.functional
    xp = F.pad(x, (2, 2, 2, 2, 2, 2), mode="replicate")
    want = F.conv3d(xp, layer.weight, layer.bias, stride=(1, 2, 2), dilation=2, groups=4)
    got = F.conv3d(xp, layer.dense_weight(), layer.bias, stride=(1, 2, 2), dilation=2)
    assert torch.allclose(got, want, atol=1e-5)
    plain = spec.build(16)
    assert plain.dense_weight() is not None and torch.equal(plain.dense_weight(), plain.weight.detach())      # groups = 1: the weight itself
    with pytest.raises(ValueError):
        spec.build(in_channels=18, groups=4)


def test_cli_flags_match_reference_surface():
    from vist3a_amd.utils.argument import inference_vist3a_argument, parse_lora_mode
    a = inference_vist3a_argument().parse_args(["--checkpoint_path", "c", "--transformer_lora_path", "l", "--input_texts_path", "t"])
    assert (a.model_id, a.num_frames, a.flow_shift, a.cfg_scale, a.resolution, a.feedforward_resolution) == \
        ("Wan-AI/Wan2.1-T2V-1.3B-Diffusers", 13, 5, "7.5", 512, 448)
    assert a.stitching_layer_location == "enc_blocks_2" and a.lora_config == "r8,a16,d0.05,f0"
    assert a.feedforward_model == "anysplat" and a.video_model == "wan" and a.output_dir == "inference_vist3a_results"
    assert parse_lora_mode("r64,a32,d0.0,f0") == (64, 32)
    with pytest.raises(SystemExit):
        inference_vist3a_argument().parse_args([])


def test_lora_merge_is_the_unmerged_map():
    from vist3a_amd.wan.dit import merge_lora_into_state_dict
    torch.manual_seed(0)
    W, A, B = torch.randn(6, 5), torch.randn(2, 5), torch.randn(6, 2)
    sd = {"blocks.0.attn1.to_q.weight": W.clone()}
    n = merge_lora_into_state_dict(sd, {"base_model.model.blocks.0.attn1.to_q.lora_A.weight": A,
                                        "base_model.model.blocks.0.attn1.to_q.lora_B.weight": B}, alpha=16, r=2)
    x = torch.randn(3, 5)
    assert n == 1 and torch.allclose(x @ sd["blocks.0.attn1.to_q.weight"].t(), x @ W.t() + 8.0 * (x @ A.t()) @ B.t(), atol=1e-5)


def test_ply_roundtrip(tmp_path):
    from vist3a_amd.utils.ply_export import export_ply
    U = 7
    g = torch.Generator().manual_seed(1)
    q = torch.randn(U, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    means, scales, sh, op = torch.randn(U, 3, generator=g), torch.rand(U, 3, generator=g) + 0.1, torch.randn(U, 3, 25, generator=g), torch.rand(U, generator=g)
    p = tmp_path / "g.ply"
    export_ply(means, scales, q, sh, op, p)
    raw = p.read_bytes()
    head, body = raw.split(b"end_header\n")
    assert b"element vertex 7" in head and head.count(b"property float") == 17 and b"binary_little_endian" in head
    arr = np.frombuffer(body, dtype="<f4").reshape(U, 17)
    assert np.allclose(arr[:, :3], means.numpy()) and np.allclose(arr[:, 6:9], sh[..., 0].numpy()) and np.allclose(arr[:, 9], op.numpy())
    assert np.allclose(arr[:, 10:13], scales.log().numpy(), atol=1e-6)
    wxyz = arr[:, 13:]
    same = np.allclose(wxyz, q[:, [3, 0, 1, 2]].numpy(), atol=1e-5) or np.allclose(np.abs(wxyz), np.abs(q[:, [3, 0, 1, 2]].numpy()), atol=1e-5)
    assert same


_WORKER = '''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from vist3a_amd.utils.dist_util import setup_dist, shard_prompts, is_main_process
setup_dist("gloo")
r, w = dist.get_rank(), dist.get_world_size()
mine = shard_prompts([f"p{i}" for i in range(7)], r, w)
out = [None] * w
dist.all_gather_object(out, mine)
dist.barrier()
if is_main_process():
    print(json.dumps(out))
dist.destroy_process_group()
'''


def test_prompt_sharding_world_size_2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", _free_port(), str(script), str(ROOT)], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("[[")][-1]
    shards = json.loads(line)
    assert shards == [["p0", "p2", "p4", "p6"], ["p1", "p3", "p5"]]


_SP_WORKER = '''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from vist3a_amd.utils.dist_util import setup_dist
from vist3a_amd.wan.seqpar import DenoisePlan
setup_dist("gloo")
r, w = dist.get_rank(), dist.get_world_size()
plan = DenoisePlan.from_dist()
ok = plan.sp is None and plan.cfg is not None and plan.cfg.rank == r and plan.cfg.world == 2
# the cond/uncond exchange of one step: each rank contributes its branch, both see [cond, uncond]
mine = torch.full((1, 4, 2, 3, 3), float(r + 1), dtype=torch.bfloat16)
pair = torch.empty((2,) + tuple(mine.shape), dtype=torch.bfloat16)
plan.cfg.all_gather(pair, mine).wait()
ok = ok and bool((pair[0] == 1).all()) and bool((pair[1] == 2).all())
# K|V^T slab exchange as the DiT does it (B=2 batch items, Nl=8 local tokens, d=4), reassembled to [B,N,d] / [d,B,N]
from vist3a_amd.wan.seqpar import DistGroup
spg = DistGroup([0, 1], r, dist.group.WORLD)
B, Nl, d, P = 2, 8, 4, 2
N = P * Nl
K = torch.arange(B * N * d, dtype=torch.float32).view(B, N, d)
Vt = -torch.arange(d * B * N, dtype=torch.float32).view(d, B, N)
kl = K[:, r * Nl:(r + 1) * Nl].reshape(B * Nl, d)
vtl = Vt[:, :, r * Nl:(r + 1) * Nl].reshape(d, B * Nl)
pack = torch.cat([kl.reshape(-1), vtl.reshape(-1)])
gbuf = torch.empty(P, pack.numel())
spg.all_gather(gbuf, pack).wait()
Ml = B * Nl
kfull = gbuf[:, :Ml * d].view(P, B, Nl, d).permute(1, 0, 2, 3).reshape(B, N, d)
vfull = gbuf[:, Ml * d:].view(P, d, B, Nl).permute(1, 2, 0, 3).reshape(d, B, N)
ok = ok and torch.equal(kfull, K) and torch.equal(vfull, Vt)
res = [None] * w
dist.all_gather_object(res, bool(ok))
if r == 0:
    print(json.dumps(res))
dist.destroy_process_group()
'''


def test_denoise_plan_world_size_2_gloo(tmp_path):
    """wan/seqpar.py over a real 2-process group: plan layout, the per-step CFG exchange and the K|V^T slab all-gather
    with the exact reassembly permutations WanDiT.forward applies."""
    script = tmp_path / "sp.py"
    script.write_text(_SP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29633")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", _free_port(), str(script), str(ROOT)], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("[")][-1]
    assert json.loads(line) == [True, True]


_SP_SHAPE_WORKER = '''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from vist3a_amd.utils.dist_util import setup_dist
from vist3a_amd.wan.seqpar import DistGroup
setup_dist("gloo")
r, P = dist.get_rank(), dist.get_world_size()
grp = DistGroup(list(range(P)), r, dist.group.WORLD)
bf16 = torch.bfloat16
B, N, H, hd = 2, 4096, 12, 128            # Wan-1.3B, 13 views: the shard shape of BASELINE configs #3 / #4 at P = 2
d, Nl = H * hd, N // P
Ml = B * Nl
g = torch.Generator().manual_seed(7)       # same stream on every rank: everybody can build the unsharded reference
K = torch.randn(B, N, d, generator=g).to(bf16)
V = torch.randn(B, N, d, generator=g).to(bf16)
Q = torch.randn(B, N, d, generator=g).to(bf16)
# what WanDiT.forward packs on this rank: K rows [B*Nl, d] then V^T [d, B*Nl] of ITS tokens
kl = K[:, r * Nl:(r + 1) * Nl].reshape(Ml, d)
vtl = V[:, r * Nl:(r + 1) * Nl].reshape(Ml, d).t().contiguous()
pack = torch.cat([kl.reshape(-1), vtl.reshape(-1)])
gbuf = torch.empty(P, pack.numel(), dtype=bf16)
grp.all_gather(gbuf, pack).wait()
# the addressing ops.attention is given (wan/dit.py): K = gbuf[0, :Ml*d] as [Ml, d], V^T = gbuf[0, Ml*d:] as [d, Ml], kv_seg = Nl,
# k_seg_stride = vt_seg_stride = 2*Ml*d elements, k_batch_stride = Nl*d, vt_batch_stride = Nl: key kk of batch item b is row
# b*Nl + kk % Nl of segment kk // Nl.  Read every key through exactly that arithmetic on the flat buffer:
flat = gbuf.view(-1)
kk = torch.arange(N)
seg, loc = kk // Nl, kk % Nl
ok = True
for b in range(B):
    k_off = seg * (2 * Ml * d) + b * (Nl * d) + loc * d                       # element offset of key row kk
    Kslab = flat[(k_off[:, None] + torch.arange(d)[None]).reshape(-1)].view(N, d)
    v_off = seg * (2 * Ml * d) + Ml * d + b * Nl + loc                         # element offset of V^T[0, kk]; channel stride Ml
    Vslab = flat[(v_off[:, None] + (torch.arange(d) * Ml)[None]).reshape(-1)].view(N, d)
    ok = ok and torch.equal(Kslab, K[b]) and torch.equal(Vslab, V[b])
    # and the local query rows attend to ALL keys: 32 of this rank's rows, every head, against the unsharded result
    rows = r * Nl + torch.arange(0, Nl, Nl // 32)
    q = Q[b, rows].float().view(-1, H, hd).transpose(0, 1)
    o_slab = torch.softmax(q @ Kslab.float().view(N, H, hd).permute(1, 2, 0) / hd ** 0.5, -1) @ Vslab.float().view(N, H, hd).transpose(0, 1)
    o_full = torch.softmax(q @ K[b].float().view(N, H, hd).permute(1, 2, 0) / hd ** 0.5, -1) @ V[b].float().view(N, H, hd).transpose(0, 1)
    ok = ok and torch.equal(o_slab, o_full)
res = [None] * P
dist.all_gather_object(res, bool(ok))
if r == 0:
    print(json.dumps(res))
dist.destroy_process_group()
'''


def test_seq_parallel_slab_addressing_at_production_shard_shape_gloo(tmp_path):
    """Two real processes (gloo), Wan-1.3B width, 4096 tokens: DistGroup all-gathers the [K | V^T] packs of the ranks' 2048-token
    shards (25 MB each) and every key of every batch item is then read from the gathered buffer through the segment / batch / row strides
    WanDiT.forward hands the flash kernel (`kv_seg`, `k_seg_stride`, `vt_seg_stride`): the slabs must be the unsharded K and V, and the
    local query rows' attention over them the unsharded attention.  The bookkeeping of the RCCL path beyond the tiny-shape test above;
    the kernel side of the same addressing is tests/test_dit_gpu.py::test_seq_parallel_production_width_reads_gathered_slabs_in_place."""
    script = tmp_path / "sps.py"
    script.write_text(_SP_SHAPE_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", _free_port(), str(script), str(ROOT)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("[")][-1]
    assert json.loads(line) == [True, True]


def test_denoise_plan_layouts():
    from vist3a_amd.wan.seqpar import DenoisePlan, ThreadWorld
    assert [DenoisePlan.layout(w) for w in (1, 2, 3, 4, 8)] == [(1, 1), (2, 1), (1, 3), (2, 2), (2, 4)]
    plans = DenoisePlan.from_threads(8)
    assert [(p.cfg.rank, p.sp.rank, p.sp.world) for p in plans] == [(c, s, 4) for c in range(2) for s in range(4)]
    w = ThreadWorld(3)
    out = w.run(lambda r: [int(v) for v in _gather_cpu(w.group(r), r)])
    assert out == [[0, 10, 20]] * 3


def _gather_cpu(group, r):
    o = torch.empty(group.world, 1)
    group.all_gather(o, torch.tensor([10.0 * r])).wait()
    return o.view(-1)


def test_merge_lora_matches_reference_eval_merge_golden():
    """recon/weights.merge_lora against tests/golden/lora_tiny.safetensors: the `lora` dict written by the reference's
    lora_state_dict(bias="lora_only") and the weights its add_lora + load_state_dict + .eval() produce (make_golden_lora.py)."""
    from pathlib import Path
    from safetensors import safe_open
    from vist3a_amd.recon.weights import merge_lora
    f = safe_open(str(Path(__file__).parent / "golden" / "lora_tiny.safetensors"), "pt")
    meta = f.metadata()
    grp = lambda p: {k[len(p):]: f.get_tensor(k) for k in f.keys() if k.startswith(p)}
    base, lora, merged, io = grp("base/"), grp("lora/"), grp("merged/"), grp("io/")
    assert any(k.endswith(".bias") for k in lora), "the checkpoint layout carries trained biases"
    sd = {k: v.clone() for k, v in base.items()}
    n = merge_lora(sd, lora, float(meta["alpha"]), int(meta["r"]))
    assert n == 6 and set(sd) == set(merged)
    for k in merged:
        assert torch.allclose(sd[k], merged[k], atol=1e-6), k
    changed = [k for k in merged if k.endswith(".bias") and not torch.equal(base[k], merged[k])]
    assert len(changed) == 4   # qkv x2 and both convs: the biases a weight-only merge would silently drop
    # the merged weights reproduce the reference module's outputs
    y = torch.nn.functional.linear(torch.nn.functional.linear(io["x"], sd["blocks.0.qkv.weight"], sd["blocks.0.qkv.bias"])[..., :12], sd["blocks.1.proj.weight"])
    z = torch.nn.functional.conv2d(torch.relu(torch.nn.functional.conv2d(io["img"], sd["head.0.weight"], sd["head.0.bias"], padding=1)), sd["head.2.weight"], sd["head.2.bias"])
    assert torch.allclose(y, io["y"], atol=1e-5) and torch.allclose(z, io["z"], atol=1e-5)
    # a key of a layer this engine HAS but cannot take (wrong shape) must not be dropped silently ...
    import pytest
    with pytest.raises(KeyError):
        merge_lora({k: v.clone() for k, v in base.items()}, {**lora, "blocks.0.qkv.bias": torch.zeros(5)}, 16.0, 4)
    # ... adapters of layers that are not part of the model at all raise as well (a stray `module.` / `stitched_3d_model.` prefix, a
    # renamed block: merging nothing and reconstructing with un-adapted weights would have no other symptom) ...
    extra = {"blocks.7.qkv.bias": torch.zeros(36), "blocks.7.qkv.lora_A": torch.zeros(4, 12), "blocks.7.qkv.lora_B": torch.zeros(36, 4)}
    with pytest.raises(KeyError):
        merge_lora({k: v.clone() for k, v in base.items()}, {**lora, **extra}, float(meta["alpha"]), int(meta["r"]))
    with pytest.raises(KeyError):
        merge_lora({k: v.clone() for k, v in base.items()}, {"module." + k: v for k, v in lora.items()}, float(meta["alpha"]), int(meta["r"]))
    # ... except under the explicit list of modules the reference wraps but the inference forward never runs (one warning)
    sd2 = {k: v.clone() for k, v in base.items()}
    unused = {"encoder.point_head." + k: v for k, v in extra.items()}
    with pytest.warns(UserWarning, match="skipped 3 adapter tensors"):
        n2 = merge_lora(sd2, {**lora, **unused}, float(meta["alpha"]), int(meta["r"]))
    assert n2 == 6 and all(torch.equal(sd2[k], sd[k]) for k in sd)
    with pytest.raises(KeyError), pytest.warns(UserWarning):    # adapter matrices present, none merged
        merge_lora({k: v.clone() for k, v in base.items()}, unused, 16.0, 4)


def test_parse_lora_mode_matches_reference_parser_golden():
    import json
    from pathlib import Path
    from safetensors import safe_open
    from vist3a_amd.utils.argument import parse_lora_mode
    meta = safe_open(str(Path(__file__).parent / "golden" / "lora_tiny.safetensors"), "pt").metadata()
    for spec, (r, alpha) in json.loads(meta["specs"]).items():
        if ",f1" in spec:
            continue
        assert parse_lora_mode(spec) == (r, alpha), spec
    import pytest
    for bad in ("x3", "r", "bfoo"):
        with pytest.raises(ValueError):
            parse_lora_mode(bad)


def test_load_peft_lora_folder_layout(tmp_path):
    """`lora_ema/` as /root/reference/train_vdm.py:32-97 writes it through peft (adapter_config.json + adapter_model.safetensors with
    `base_model.model.<module>.lora_{A,B}.weight`) and /root/reference/inference_t23d.py:74-77 reads it.  peft itself is not
    installed here (parity unpinned): the folder is written by hand in the published adapter format."""
    import json
    from safetensors.torch import save_file
    from vist3a_amd.wan.weights import load_peft_lora
    g = torch.Generator().manual_seed(5)
    d, r, alpha = 16, 4, 8
    sd = {f"blocks.{i}.attn1.{n}.weight": torch.randn(d, d, generator=g).to(torch.bfloat16) for i in range(2) for n in ("to_q", "to_k", "to_v", "to_out.0")}
    ref = {k: v.float().clone() for k, v in sd.items()}
    ad = {}
    for k in list(sd)[:5]:
        mod = k[: -len(".weight")]
        A, B = torch.randn(r, d, generator=g) * 0.1, torch.randn(d, r, generator=g) * 0.1
        ad[f"base_model.model.{mod}.lora_A.weight"], ad[f"base_model.model.{mod}.lora_B.weight"] = A, B
        ref[k] = ref[k] + (alpha / r) * (B @ A)
    save_file(ad, str(tmp_path / "adapter_model.safetensors"))
    (tmp_path / "adapter_config.json").write_text(json.dumps({"peft_type": "LORA", "r": r, "lora_alpha": alpha,
                                                               "target_modules": ["to_q", "to_k", "to_v", "to_out.0"]}))
    assert load_peft_lora(str(tmp_path), sd) == 5
    for k in sd:
        assert torch.allclose(sd[k].float(), ref[k], atol=1e-6), k          # merged in fp32, not re-rounded to the checkpoint dtype
    bad = dict(ad)
    bad["base_model.model.blocks.9.attn1.to_q.lora_A.weight"] = torch.zeros(r, d)
    bad["base_model.model.blocks.9.attn1.to_q.lora_B.weight"] = torch.zeros(d, r)
    save_file(bad, str(tmp_path / "adapter_model.safetensors"))
    import pytest
    with pytest.raises(KeyError):
        load_peft_lora(str(tmp_path), sd)


def test_full_upstream_checkpoint_drops_first_k_dino_blocks_and_reindexes_all():
    """convert_model_to_stitched_model (/root/reference/models/anysplat_stitched.py:158-165): a checkpoint with ALL DINO blocks loses
    blocks [0, k) and block i becomes block i - k - for every block (a one-pass in-place rename collides at i - k == j)."""
    from oracle import recon as R
    from vist3a_amd.models.anysplat_stitched import AnySplatStitched, AnySplatWeights
    from vist3a_amd.recon.engine import ReconCfg
    kw = dict(C=64, heads=1, n_dino=22, depth=2, cam_heads=2, cam_trunk=1, features=32, oc=(16, 32, 64, 64))
    ssd = R.make_recon_weights(R.ReconCfg(**kw), seed=1)
    pe = "encoder.aggregator.patch_embed.blocks."
    full = {}
    for k, v in ssd.items():
        if k.startswith(pe):
            i, rest = k[len(pe):].split(".", 1)
            full[f"{pe}{int(i) + 2}.{rest}"] = v
            if int(i) < 2:
                full[f"{pe}{i}.{rest}"] = torch.full_like(v, 7.0)
        else:
            full[k] = v
    m = AnySplatStitched(AnySplatWeights(full, ReconCfg(**kw), n_total_dino_blocks=24), "enc_blocks_2", "cpu")
    got = {k: v for k, v in m._sd.items() if k.startswith(pe)}
    assert set(got) == {k for k in ssd if k.startswith(pe)}
    assert all(torch.equal(got[k], ssd[k]) for k in got)
    import pytest
    with pytest.raises(ValueError):   # a checkpoint that is neither stitched (22) nor full (24)
        AnySplatStitched(AnySplatWeights({k: v for k, v in full.items() if ".blocks.23." not in k}, ReconCfg(**kw), n_total_dino_blocks=24), "enc_blocks_2", "cpu")


def test_oracle_digest_cache_roundtrip(tmp_path, monkeypatch):
    """tests/oracle_cache.py: the digest of a tensor is a fixed strided sample, the sampled relative error tracks the full one, the
    fixture is keyed to an exact checksum of the inputs (a different input is rejected, not silently compared)."""
    import oracle_cache as OC
    g = torch.Generator().manual_seed(0)
    ref = torch.randn(7, 1029, 2048, generator=g)
    x = ref + 1e-2 * torch.randn(ref.shape, generator=g)
    full = ((x - ref).norm() / ref.norm()).item()
    d = OC.digest(ref)
    assert 8192 <= d.numel() <= 2 * 16384 + 1 and torch.equal(d, ref.reshape(-1)[OC.sample_index(ref.numel())])
    assert abs(OC.rel(x, d) / full - 1) < 0.03
    assert OC.checksum(ref) == OC.checksum(ref.clone()) and OC.checksum(ref) != OC.checksum(x) and OC.checksum(ref.bfloat16()) != OC.checksum(x.bfloat16())
    monkeypatch.setattr(OC, "GOLD", tmp_path)
    monkeypatch.setenv("V3A_WRITE_ORACLE", "1")
    monkeypatch.delenv("V3A_LIVE_ORACLE", raising=False)
    monkeypatch.delenv("V3A_ORACLE_OUT", raising=False)
    calls = []
    comp = lambda: (calls.append(1), dict(a=ref, n=12345, s=1.5))[1]
    d1, live1 = OC.oracle("unit", OC.checksum(ref), comp)
    d2, live2 = OC.oracle("unit", OC.checksum(ref), comp)
    assert live1 and not live2 and len(calls) == 1 and torch.equal(d1["a"], d2["a"]) and int(d2["n"].item()) == 12345 and float(d2["s"].item()) == 1.5
    import pytest
    with pytest.raises(AssertionError):
        OC.oracle("unit", OC.checksum(x), comp)


def test_bench_parses_the_rocm_smi_clock_and_power_sample():
    """bench.py's untimed clock extra: shader clock and package power out of `rocm-smi --showclocks --showpower --json` as the MI355X boxes
    print it; anything else (no GPU, other key names, an error text) gives None and the bench line carries null."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("bench_mod", Path(__file__).resolve().parents[1] / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sample = ('{"card0": {"fclk clock speed:": "(1250Mhz)", "fclk clock level:": "0", "mclk clock speed:": "(2000Mhz)", "mclk clock level:": "0", '
              '"sclk clock speed:": "(1977Mhz)", "sclk clock level:": "1", "socclk clock speed:": "(38Mhz)", '
              '"Current Socket Graphics Package Power (W)": "1307.0"}}')
    assert bench.parse_rocm_smi(sample) == (1977, 1307.0)
    assert bench.parse_rocm_smi('{"card0": {"sclk clock level:": "S"}}') is None
    assert bench.parse_rocm_smi("ERROR: no GPU") is None and bench.parse_rocm_smi("") is None
    assert bench.parse_rocm_smi('{"card0": {"sclk clock speed:": "(95Mhz)", "Average Graphics Package Power (W)": "N/A"}}') is None


def test_committed_oracle_digests_match_the_current_oracle_sources():
    """tests/oracle_cache.py: every committed full-size digest carries the sha256 of the oracle sources that computed it; an edit of
    oracle/recon.py / oracle/wan_dit.py (or of the case definitions) without regenerating the digests fails HERE, on the CPU, not only
    when the GPU suite next reads them."""
    import sys
    from pathlib import Path
    from safetensors import safe_open
    sys.path.insert(0, str(Path(__file__).parent))
    import fullsize_cases as FC
    import oracle_cache as OC
    files = sorted(OC.GOLD.glob("oracle_*.safetensors"))
    assert {f.stem[len("oracle_"):] for f in files} == set(FC.SOURCES)
    for f in files:
        meta = safe_open(str(f), "pt").metadata() or {}
        name = f.stem[len("oracle_"):]
        assert meta.get("oracle_sources_sha256") == OC.source_hash(*FC.SOURCES[name]), f"{f.name} is stale: re-run tests/golden/make_fullsize_oracle.py"
        assert meta.get("generated_without_gpu") == "True", f"{f.name} was not written by the CPU-only generator"


_VIEW_SHARD_WORKER = '''
import sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from vist3a_amd.utils.dist_util import setup_dist
from vist3a_amd.wan.seqpar import DenoisePlan
from vist3a_amd.recon.engine import ReconEngine as E, OUT_COLS
setup_dist("gloo")
r, w = dist.get_rank(), dist.get_world_size()
plan = DenoisePlan.from_dist()
grp = plan.world
ok = grp is not None and grp.world == 2 and grp.rank == r
S, Pp, C = 13, 16, 8
views = E.shard_views(S, w)
ok = ok and views == [(0, 7), (7, 13)]
v0, v1 = views[r]
Sl, maxS = v1 - v0, 7
Mx = maxS * Pp
# a global block's exchange: every rank contributes the K rows / V^T columns of ITS views, everybody reassembles all 13 views in order
K = torch.arange(S * Pp * C, dtype=torch.float32).view(S * Pp, C)
Vt = -torch.arange(C * S * Pp, dtype=torch.float32).view(C, S * Pp)
pack, gbuf = torch.zeros(2 * Mx * C), torch.zeros(w, 2 * Mx * C)
E.kv_pack(pack, K[v0 * Pp: v1 * Pp], Vt[:, v0 * Pp: v1 * Pp], Mx)
grp.all_gather(gbuf, pack).wait()
kf, vtf = torch.zeros(S * Pp, C), torch.zeros(C, S * Pp + 64)
E.kv_unpack(gbuf, views, Pp, Mx, kf, vtf)
ok = ok and torch.equal(kf, K) and torch.equal(vtf[:, : S * Pp], Vt) and bool((vtf[:, S * Pp:] == 0).all())
# camera-token rows and the per-pixel maps: padded per-rank blocks -> rows of all views in view order
HW = 6
full = torch.arange(S * HW * OUT_COLS, dtype=torch.float32).view(S * HW, OUT_COLS)
op, og = torch.zeros(maxS * HW * OUT_COLS), torch.zeros(w, maxS * HW * OUT_COLS)
op.view(maxS * HW, OUT_COLS)[: Sl * HW] = full[v0 * HW: v1 * HW]
grp.all_gather(og, op).wait()
ok = ok and torch.equal(E.gather_rows(og, views, HW, OUT_COLS, maxS), full)
res = [None] * w
dist.all_gather_object(res, bool(ok))
if r == 0:
    print(json.dumps(res))
dist.destroy_process_group()
'''


def test_view_sharded_reconstruction_exchange_world_size_2_gloo(tmp_path):
    """SURVEY 8(e) "Recon under SP": the 13 -> 7 / 6 view split of `ReconEngine.forward_sharded` over two real processes (gloo): the
    DenoisePlan's world group, the K | V^T pack / all-gather / reassembly of a global block and the final per-pixel gather, through the
    very helpers the GPU path calls (kv_pack / kv_unpack / gather_rows) - ragged shards included."""
    script = tmp_path / "vs.py"
    script.write_text(_VIEW_SHARD_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", _free_port(), str(script), str(ROOT)], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("[")][-1]
    assert json.loads(line) == [True, True]


def test_shard_views_layout():
    from vist3a_amd.recon.engine import ReconEngine as E
    for S in (1, 5, 13, 21):
        for w in (1, 2, 3, 4, 8):
            v = E.shard_views(S, w)
            assert len(v) == w and v[0][0] == 0 and v[-1][1] == S and all(a[1] == b[0] for a, b in zip(v, v[1:]))
            sizes = [b - a for a, b in v]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert E.shard_views(13, 4) == [(0, 4), (4, 7), (7, 10), (10, 13)] and E.shard_views(21, 8)[0] == (0, 3)


def test_bench_started_without_a_launcher_spawns_its_own_ranks_gloo():
    """`python bench.py --gpus 2` the way the driver calls `--gpus 1` (no torch.distributed.run around it): bench.py must become the launcher of
    its own two ranks (127.0.0.1 rendezvous on a free port), run the barrier / max-over-ranks skeleton and print exactly ONE JSON line that
    says how many ranks it really ran on.  `--launch-self-test` swaps the scene for a stub step on gloo / CPU; everything around it is the
    code path of the real multi-GPU run (spawn_ranks, timed_steps)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--launch-self-test"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["config"]["rccl"]["world_size"] == 2 and "bench.py itself" in line["config"]["rccl"]["launched_by"]
    assert line["config"]["steps_run_per_rank_incl_warmup"] == [4, 4]          # W untimed + EXACTLY K timed steps on every rank
    # the stub step sleeps 10 ms x (rank + 1): the reported time is the MAX over ranks (>= 3 x 20 ms), not rank 0's own 30 ms
    assert line["ms_per_step"] >= 19.0, line


def test_bench_rejects_a_launcher_with_the_wrong_rank_count():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--launch-self-test"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)
