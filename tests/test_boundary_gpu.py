"""GPU: the drop-in boundary objects (StitchVAE3D.forward_with_latent / AnySplatStitched.forward / Text23DGS.generate) against
the oracle on seeded inputs, at reduced width, plus full-size size-independent properties."""
import pytest
import torch

from oracle import recon as R

pytestmark = pytest.mark.gpu
RECON_TINY = dict(C=64, heads=1, n_dino=22, depth=24, cam_heads=2, cam_trunk=2, features=32, oc=(16, 32, 64, 64))


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def test_forward_with_latent_matches_oracle(hip_lib):
    from vist3a_amd.models.anysplat_stitched import AnySplatWeights
    from vist3a_amd.models.stitched_model import StitchVAE3D
    from vist3a_amd.models.stitching_layer_builder import parse_conv_spec
    from vist3a_amd.models.types import EncoderOutput
    from vist3a_amd.recon.engine import ReconCfg
    ocfg = R.ReconCfg(**RECON_TINY)
    sd = R.make_recon_weights(ocfg, seed=7)
    model = StitchVAE3D(None, AnySplatWeights(dict(sd), ReconCfg(**RECON_TINY)), "cuda", "enc_blocks_2",
                        parse_conv_spec("conv3d_k5x3x3_o64_s1x2x2_p2x1x1"), resolution=32)
    g = torch.Generator().manual_seed(8)
    w = torch.randn(64, 16, 5, 3, 3, generator=g) * 0.08
    b = torch.randn(64, generator=g) * 0.1
    model.stitching_layer.weight.data, model.stitching_layer.bias.data = w, b   # assigned the way nvs_eval.py:51-52 does
    Tl, S, H = 2, 5, 28
    lat = torch.randn(1, 16, Tl, 4, 4, generator=g)
    img = torch.rand(1, 3, S, H, H, generator=g) * 2 - 1
    out = model.forward_with_latent(lat.cuda(), img.cuda(), train=False)
    assert isinstance(out, EncoderOutput) and out.last_pred_pose_enc.shape == (1, S, 9)
    with torch.no_grad():
        feat = R.stitch_conv(R.upsample_T(lat), w, b, (1, 2, 2), (2, 1, 1))
        ora = R.recon_forward(sd, ocfg, feat, img)
    r_pose = _rel(out.last_pred_pose_enc, ora["pred_pose_enc_list"][-1])
    r_depth = _rel(out.depth_dict["depth"], ora["depth"])
    r_c2w = _rel(out.pred_context_pose["extrinsic"], ora["pred_context_pose"]["extrinsic"])
    print(f"boundary: pose {r_pose:.2e} depth {r_depth:.2e} c2w {r_c2w:.2e} U {out.gaussians.means.shape[1]} vs {ora['gaussians']['means'].shape[1]}")
    assert r_pose < 4e-2 and r_depth < 8e-3 and r_c2w < 1e-1  # measured 2.0e-2 / 4.1e-3 / 5.4e-2 (c2w = inverse(w2c) amplifies the pose noise at width 64;
    #                                                            at production width: 2.8e-3 / 3.9e-3 / 4.2e-3, tests/test_fullsize_gpu.py)
    assert out.gaussians.means.shape[0] == 1 and out.gaussians.covariances.shape[-2:] == (3, 3) and out.gaussians.harmonics.shape[-2:] == (3, 25)
    U, Uo = out.gaussians.means.shape[1], ora["gaussians"]["means"].shape[1]
    assert abs(U - Uo) <= 0.08 * Uo
    # train=True returns the 4-tuple of anysplat_stitched.py:501-514
    eo, anchor, conf, dconf = model.forward_with_latent(lat.cuda(), img.cuda(), train=True)
    assert anchor.shape == (1, S, 83, H, H) and conf.shape == (1, S, H, H) and dconf.shape == (1, S, H, H)
    # pre_upsample_layer == trilinear align_corners=True
    up = model.pre_upsample_layer(lat.cuda()).cpu()
    assert torch.allclose(up, R.upsample_T(lat).to(torch.bfloat16).float(), atol=1e-6)
    # voxelize switch (model_stitching_training.py:331) is honoured
    model.stitched_3d_model.encoder.cfg.voxelize = False
    out2 = model.forward_with_latent(lat.cuda(), img.cuda())
    assert out2.gaussians.means.shape[1] == S * H * H


def test_forward_with_latent_batch_of_scenes(hip_lib):
    """`StitchVAE3D.forward_with_latent` is batch-agnostic in the reference (stitched_model.py:165-173; the (b v) fold of
    anysplat_stitched.py:174-202 reconstructs every scene on its own views): B = 2 equals the two B = 1 forwards scene by scene - depth,
    poses and raw maps bit for bit, each scene's Gaussians in the leading rows of the padded batch, padding rows with opacity 0 - and
    the feed-forward image stays fp32 through (x + 1) / 2 (:174-175): an image that is not bf16-representable gives a different
    result from its bf16 rounding."""
    from vist3a_amd.models.anysplat_stitched import AnySplatWeights
    from vist3a_amd.models.stitched_model import StitchVAE3D
    from vist3a_amd.models.stitching_layer_builder import parse_conv_spec
    from vist3a_amd.recon.engine import ReconCfg
    sd = R.make_recon_weights(R.ReconCfg(**RECON_TINY), seed=7)
    model = StitchVAE3D(None, AnySplatWeights(dict(sd), ReconCfg(**RECON_TINY)), "cuda", "enc_blocks_2",
                        parse_conv_spec("conv3d_k5x3x3_o64_s1x2x2_p2x1x1"), resolution=32)
    g = torch.Generator().manual_seed(18)
    model.stitching_layer.weight.data = torch.randn(64, 16, 5, 3, 3, generator=g) * 0.08
    model.stitching_layer.bias.data = torch.randn(64, generator=g) * 0.1
    S, H = 5, 28
    lat = torch.randn(2, 16, 2, 4, 4, generator=g).cuda()
    img = (torch.rand(2, 3, S, H, H, generator=g) * 2 - 1).cuda()
    eo, anchor, conf, dconf = model.forward_with_latent(lat, img, train=True)
    assert eo.gaussians.means.shape[0] == 2 and anchor.shape == (2, S, 83, H, H) and dconf.shape == (2, S, H, H)
    assert eo.pred_pose_enc_list[-1].shape == (2, S, 9) and eo.depth_dict["depth"].shape == (2, S, H, H, 1)
    U = eo.gaussians.means.shape[1]
    counts = []
    for b in range(2):
        e1, a1, c1, d1 = model.forward_with_latent(lat[b:b + 1], img[b:b + 1], train=True)
        n = e1.gaussians.means.shape[1]
        counts.append(n)
        assert torch.equal(eo.depth_dict["depth"][b], e1.depth_dict["depth"][0]) and torch.equal(anchor[b], a1[0]) and torch.equal(dconf[b], d1[0])
        assert torch.equal(eo.pred_pose_enc_list[-1][b], e1.pred_pose_enc_list[-1][0])
        assert torch.equal(eo.gaussians.means[b, :n], e1.gaussians.means[0]) and torch.equal(eo.gaussians.harmonics[b, :n], e1.gaussians.harmonics[0])
        assert (eo.gaussians.opacities[b, n:] == 0).all()          # padded rows: density sigmoid(-1e10) = 0 (anysplat_stitched.py:440-453)
    assert U == max(counts)
    # fp32 image path: a perturbation below the bf16 resolution of every pixel (|x| >= 0.25: half an ulp is >= 2^-10) must still reach the
    # Gaussian head's input_merger - it would vanish if the image were rounded to bf16 before (x + 1) / 2
    sign = torch.where(torch.rand(1, 3, S, H, H, generator=g) < 0.5, -1.0, 1.0)
    img1 = (sign * (0.25 + 0.75 * torch.rand(1, 3, S, H, H, generator=g))).to(torch.bfloat16).float().cuda()
    img2 = img1 + 2.0 ** -12
    assert torch.equal(img2.to(torch.bfloat16), img1.to(torch.bfloat16)) and not torch.equal(img2, img1)
    d_a = model.forward_with_latent(lat[:1], img1, train=True)[1]
    d_b = model.forward_with_latent(lat[:1], img2, train=True)[1]
    assert not torch.equal(d_a, d_b)
    assert torch.equal(d_a, model.forward_with_latent(lat[:1], img1, train=True)[1])    # (and the forward itself is deterministic)


def test_image_conditioned_forward_equals_encode_then_forward_with_latent(hip_lib):
    """StitchVAE3D.forward (stitched_model.py:139-163): VAE-encode the views, sample the posterior, forward_with_latent."""
    from oracle import wan_vae as OV
    from vist3a_amd.models.anysplat_stitched import AnySplatWeights
    from vist3a_amd.models.stitched_model import StitchVAE3D
    from vist3a_amd.models.stitching_layer_builder import parse_conv_spec
    from vist3a_amd.recon.engine import ReconCfg
    from vist3a_amd.wan.vae import WanVAEConfig, WanVAEDecoder
    vcfg = OV.WanVAEConfig(base_dim=16)
    vsd = dict(OV.make_weights(vcfg, seed=5))
    vsd.update(OV.make_encoder_weights(vcfg, seed=6))
    vae = WanVAEDecoder(WanVAEConfig(base_dim=16), vsd)
    sd = R.make_recon_weights(R.ReconCfg(**RECON_TINY), seed=7)
    model = StitchVAE3D(vae, AnySplatWeights(dict(sd), ReconCfg(**RECON_TINY)), "cuda", "enc_blocks_2",
                        parse_conv_spec("conv3d_k5x3x3_o64_s1x2x2_p2x1x1"), resolution=32)
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        model.stitching_layer.weight.copy_(torch.randn(64, 16, 5, 3, 3, generator=g) * 0.08)
    S, H = 5, 28
    images = (torch.rand(1, 3, S, 32, 32, generator=g) * 2 - 1).cuda()   # 5 views @ 32^2 -> latent [1,16,2,4,4]
    ff = (torch.rand(1, 3, S, H, H, generator=g) * 2 - 1).cuda()
    out = model.forward(images, ff, train=False, generator=torch.Generator().manual_seed(1))
    lat, none = model.vae_encoder_forward(images, decode=False, generator=torch.Generator().manual_seed(1))
    assert none is None and lat.shape == (1, 16, 2, 4, 4)
    ref = model.forward_with_latent(lat, ff, train=False)
    assert torch.equal(out.last_pred_pose_enc, ref.last_pred_pose_enc) and torch.equal(out.gaussians.means, ref.gaussians.means)
    # the latent really is the oracle's posterior sample of the oracle's encoding (bf16 conv stack: 2e-2)
    noise = torch.randn(lat.shape, generator=torch.Generator().manual_seed(1))
    want = OV.posterior_sample(OV.encode(vsd, vcfg, images.cpu()), noise)
    assert _rel(lat, want) < 2e-2
    lat2, dec = model.vae_encoder_forward(images, decode=True, generator=torch.Generator().manual_seed(1))
    assert dec.shape == (1, 3, S, 32, 32)


def test_scene_pipeline_properties_reduced(hip_lib):
    """Text23DGS.generate end to end at reduced DiT/recon width but production 512^2/448^2 geometry: determinism, finiteness,
    size-independent invariants (unit quaternions, symmetric PSD covariances, opacity in (0,1), scale clamp, sorted voxel keys)."""
    from vist3a_amd.recon.engine import ReconCfg
    from vist3a_amd.t23d import Text23DGS
    from vist3a_amd.wan.dit import WanDiTConfig
    from vist3a_amd.wan.vae import WanVAEConfig
    dit = WanDiTConfig(num_attention_heads=2, attention_head_dim=128, ffn_dim=512, num_layers=2, text_dim=256, freq_dim=64)
    m = Text23DGS.synthetic(dit, seed=0, device="cuda", recon_cfg=ReconCfg(**RECON_TINY), vae_cfg=WanVAEConfig(base_dim=16),
                            stitch_spec="conv3d_k5x3x3_o64_s1x2x2_p2x1x1")
    g = torch.Generator().manual_seed(1)
    pe = torch.randn(1, 64, 256, generator=g) * 0.1
    ne = torch.randn(1, 64, 256, generator=g) * 0.1
    lat0 = torch.randn(1, 16, 2, 64, 64, generator=g)
    o1, lat, clip = m.generate(pe.cuda(), ne.cuda(), latents=lat0.clone(), num_frames=5, num_inference_steps=3, guidance_scale=7.5)
    o2, _, _ = m.generate(pe.cuda(), ne.cuda(), latents=lat0.clone(), num_frames=5, num_inference_steps=3, guidance_scale=7.5)
    gs = o1.gaussians
    assert clip.shape == (5, 512, 512, 8) and clip.abs().max() <= 1.0
    assert torch.equal(gs.means, o2.gaussians.means) and torch.equal(gs.harmonics, o2.gaussians.harmonics)
    for t in (gs.means, gs.covariances, gs.harmonics, gs.opacities, gs.scales, gs.rotations):
        assert torch.isfinite(t).all()
    assert torch.allclose(gs.rotations.norm(dim=-1), torch.ones_like(gs.opacities), atol=1e-4)
    assert (gs.opacities > 0).all() and (gs.opacities < 1).all() and (gs.scales > 0).all() and gs.scales.max() <= 0.3 + 1e-6
    cov = gs.covariances[0]
    assert torch.allclose(cov, cov.transpose(-1, -2), atol=1e-9) and (torch.linalg.eigvalsh(cov.double().cpu()) > -1e-9).all()
    assert gs.means.shape[1] <= 5 * 448 * 448


def test_rccl_backend_single_rank_all_gather(hip_lib, tmp_path):
    """`DistGroup` on the real RCCL backend (one rank is all a 1-GPU box offers): process-group bring-up via setup_dist,
    asynchronous all_gather_into_tensor on device tensors, wait() ordering the compute stream."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    script = tmp_path / "rccl1.py"
    script.write_text(f"""
import sys
sys.path.insert(0, {str(root)!r})
import torch, torch.distributed as dist
from vist3a_amd.utils.dist_util import setup_dist
from vist3a_amd.wan.seqpar import DenoisePlan, DistGroup
setup_dist()
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
plan = DenoisePlan.from_dist()
assert plan.sp is None and plan.cfg is None          # one rank: nothing to shard
g = DistGroup([0], 0, dist.group.WORLD)
x = torch.arange(1 << 20, device="cuda", dtype=torch.float32).to(torch.bfloat16)
out = torch.empty(1, x.numel(), device="cuda", dtype=torch.bfloat16)
h = g.all_gather(out, x)
h.wait()
y = out[0].float().sum()                              # consumer on the compute stream, ordered after the collective
torch.cuda.synchronize()
assert torch.equal(out[0], x) and float(y) == float(x.float().sum())
dist.destroy_process_group()
print("rccl ok")
""")
    import os
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stderr[-2000:]


_RCCL_SCENE_WORKER = '''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
local = int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
from vist3a_amd.wan.dit import WanDiT, WanDiTConfig
from vist3a_amd.wan.pipeline import WanT2VPipeline
from vist3a_amd.wan.scheduler import UniPCMultistepScheduler
from vist3a_amd.wan.seqpar import DenoisePlan
from vist3a_amd.wan.weights import random_dit_state_dict
cfg = WanDiTConfig(num_attention_heads=2, attention_head_dim=128, ffn_dim=512, num_layers=2, text_dim=128, freq_dim=64)
model = WanDiT(cfg, random_dit_state_dict(cfg, seed=3, device="cuda"), device="cuda")
g = torch.Generator().manual_seed(9)
pe = torch.randn(1, 32, cfg.text_dim, generator=g) * 0.5
ne = torch.randn(1, 32, cfg.text_dim, generator=g) * 0.5
lat0 = torch.randn(1, 16, 2, 32, 32, generator=g)     # 512 tokens: 64-key multiples for up to 8 token shards
kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, height=256, width=256, num_frames=5, num_inference_steps=4, guidance_scale=6.0, latents=lat0)
ref = WanT2VPipeline(model, UniPCMultistepScheduler(flow_shift=5.0))(**kw)["frames"].clone()
out = WanT2VPipeline(model, UniPCMultistepScheduler(flow_shift=5.0), plan=DenoisePlan.from_dist())(**kw)["frames"].clone()
torch.cuda.synchronize()
# the same denoise with every rank's sharded forward (its RCCL all-gathers included) replayed from a captured hipGraph
from vist3a_amd.wan.dit import GraphedWanDiT
gout = WanT2VPipeline(GraphedWanDiT(model, capture_sp=True), UniPCMultistepScheduler(flow_shift=5.0), plan=DenoisePlan.from_dist())(**kw)["frames"].clone()
torch.cuda.synchronize()
ok = bool(torch.equal(out, ref)) and bool(torch.equal(gout, ref))
# the two stages behind the denoise, sharded over ALL ranks of the scene (round 5): VAE decode by H-strips with exchanged halo rows,
# reconstruction by views with one K | V^T all-gather per global block - each must return the unsharded result on every rank
wg = DenoisePlan.from_dist().world
if wg is not None:
    from vist3a_amd.t23d import random_vae_decoder_state_dict
    from vist3a_amd.wan.vae import WanVAEConfig, WanVAEDecoder
    vcfg = WanVAEConfig(base_dim=16)
    vae = WanVAEDecoder(vcfg, random_vae_decoder_state_dict(vcfg, 1, "cuda"))
    z = torch.randn(1, 16, 2, 16, 16, generator=g).cuda()
    ok = ok and bool(torch.equal(vae.decode_cl_sharded(z, wg), vae.decode_cl(z)))
    from vist3a_amd.models.anysplat_stitched import AnySplatWeights
    from vist3a_amd.models.stitched_model import StitchVAE3D
    from vist3a_amd.models.stitching_layer_builder import parse_conv_spec
    from vist3a_amd.recon.engine import ReconCfg
    from vist3a_amd.recon.weights import random_recon_state_dict
    rk = dict(C=64, heads=1, n_dino=22, depth=24, cam_heads=2, cam_trunk=2, features=32, oc=(16, 32, 64, 64))
    st = StitchVAE3D(None, AnySplatWeights(random_recon_state_dict(ReconCfg(**rk), seed=4, device="cuda", scene_like=True), ReconCfg(**rk)), "cuda",
                     "enc_blocks_2", parse_conv_spec("conv3d_k5x3x3_o64_s1x2x2_p2x1x1"), resolution=64)
    with torch.no_grad():
        st.stitching_layer.weight.copy_(torch.randn(st.stitching_layer.weight.shape, generator=g) * 0.05)
    lat = torch.randn(1, 16, 3, 8, 8, generator=g).cuda()
    img = (torch.rand(1, 3, 9, 56, 56, generator=g) * 2 - 1).cuda()
    a = st.forward_with_latent(lat, img, train=False)
    b = st.forward_with_latent(lat, img, train=False, recon_group=wg)
    ok = ok and bool(torch.equal(a.gaussians.means, b.gaussians.means)) and bool(torch.equal(a.depth_dict["depth"], b.depth_dict["depth"]))
    ok = ok and bool(torch.equal(a.gaussians.harmonics, b.gaussians.harmonics)) and bool(torch.equal(a.pred_pose_enc_list[-1], b.pred_pose_enc_list[-1]))
    torch.cuda.synchronize()
res = [None] * dist.get_world_size()
dist.all_gather_object(res, ok)
if dist.get_rank() == 0:
    print(json.dumps(res))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_rccl_scene_parallel_denoise_matches_single_gpu(hip_lib, tmp_path, world):
    """`bench.py --parallel scene` / `inference_t23d.py --scene_parallel` in miniature over REAL ranks: one process per GPU, RCCL
    over xGMI, DenoisePlan.from_dist() (CFG-parallel x sequence-parallel, K|V^T slabs read in place) - the 4-step CFG denoise must
    reproduce the single-GPU latents bit for bit on every rank, and (world > 1) the strip-sharded VAE decode and the view-sharded
    reconstruction must return the unsharded clip / scene on every rank.  Needs `world` GPUs (skipped on the 1-GPU dev box; the same
    decomposition is covered there by the virtual-rank tests in test_dit_gpu.py and the 2-process gloo test)."""
    import os, subprocess, sys, json
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, {torch.cuda.device_count()} visible")
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    script = tmp_path / "scene.py"
    script.write_text(_RCCL_SCENE_WORKER)
    port = str(29700 + world)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", port, str(script), str(root)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("[")][-1]
    assert json.loads(line) == [True] * world


_RCCL_CAPTURE_WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
from vist3a_amd.wan.seqpar import DistGroup
grp = DistGroup([0], 0, None)
inp = torch.arange(1 << 16, device="cuda", dtype=torch.float32)
out = torch.zeros(1, 1 << 16, device="cuda")
grp.all_gather(out, inp).wait()          # eager warm-up: communicator set-up must not happen under capture
torch.cuda.synchronize()
import time
time.sleep(0.3)                          # the RCCL watchdog retires the eager collective (it polls every ~100 ms) ...
out.zero_()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):   # ... and may keep polling while THIS thread captures (GraphedWanDiT does the same)
    grp.all_gather(out, inp).wait()
    y = out * 2
ok = []
for k in range(3):
    inp.add_(1.0)
    g.replay()
    torch.cuda.synchronize()
    ok.append(bool(torch.equal(out[0], inp)) and bool(torch.equal(y[0], inp * 2)))
print(json.dumps(ok))
dist.destroy_process_group()
'''


def test_rccl_all_gather_is_capturable_in_a_hipgraph(hip_lib, tmp_path):
    """DistGroup.all_gather (async RCCL all_gather_into_tensor + wait) inside a captured hipGraph on a one-rank communicator: the replay
    must re-run the collective on the current contents of its input.  (Captured thread-locally after the watchdog retired the warm-up
    collective: under the default global capture mode the watchdog thread's event poll aborted 2 of 12 runs.)  What GraphedWanDiT(capture_sp=True) relies on; the multi-rank
    form runs inside test_rccl_scene_parallel_denoise_matches_single_gpu whenever more than one GPU is visible."""
    import os, subprocess, sys, json
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    script = tmp_path / "cap.py"
    script.write_text(_RCCL_CAPTURE_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script), str(root)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("[")][-1]
    assert json.loads(line) == [True, True, True]


def _tiny_dit():
    from oracle import wan_dit as O
    from vist3a_amd.wan.dit import WanDiT, WanDiTConfig
    kw = dict(num_attention_heads=2, attention_head_dim=128, ffn_dim=512, num_layers=2, text_dim=128, freq_dim=64)
    sd = {k: v.to(torch.bfloat16).float() for k, v in O.make_weights(O.WanDiTConfig(**kw), seed=3).items()}
    return WanDiT(WanDiTConfig(**kw), sd, device="cuda")


@pytest.mark.parametrize("steps,guidance,shape", [(7, 7.5, (1, 16, 2, 16, 16)), (3, 1.0, (1, 16, 1, 8, 12)), (50, 6.0, (1, 16, 1, 8, 8))])
def test_fused_denoise_loop_is_bit_identical_to_tensor_ops(hip_lib, steps, guidance, shape):
    """One launch per step (csrc/denoise_step.hip: unpatchify + classifier-free guidance + UniPC corrector / predictor + next-step
    patchify, time conditioning of the schedule computed up front) against the loop of PyTorch tensor ops that restates diffusers'
    WanPipeline / UniPCMultistepScheduler.step op by op: the latents after the whole schedule must agree bit for bit (first step without
    corrector, order-1 and order-2 corrector, order-2 predictor, the final order-1 step, guided and unguided)."""
    from vist3a_amd.wan.pipeline import WanT2VPipeline
    from vist3a_amd.wan.scheduler import UniPCMultistepScheduler
    model = _tiny_dit()
    g = torch.Generator().manual_seed(steps)
    pe = (torch.randn(1, 24, 128, generator=g) * 0.5)
    ne = (torch.randn(1, 24, 128, generator=g) * 0.5)
    lat0 = torch.randn(shape, generator=g)
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, height=shape[3] * 8, width=shape[4] * 8, num_frames=(shape[2] - 1) * 4 + 1,
              num_inference_steps=steps, guidance_scale=guidance)
    outs = {}
    for fused in (False, True):
        pipe = WanT2VPipeline(model, UniPCMultistepScheduler(flow_shift=5.0))
        pipe.fused = fused
        outs[fused] = pipe(latents=lat0.clone(), **kw)["frames"].clone()
    torch.cuda.synchronize()
    assert torch.isfinite(outs[True]).all()
    d = (outs[True] - outs[False]).abs().max().item()
    assert torch.equal(outs[True], outs[False]), f"fused loop differs from the tensor-op loop: max abs {d}"


def test_fused_step_kernel_matches_tensor_ops_for_every_order(hip_lib):
    """The step kernel alone, on random fp32 / bf16 inputs, for every (corrector order, predictor order) pair the scheduler can ask for:
    the PyTorch expression sequence of scheduler.py::_correct / _predict, evaluated on the GPU, bit for bit."""
    from vist3a_amd import ops
    from vist3a_amd.wan.scheduler import UniPCMultistepScheduler
    g = torch.Generator(device="cuda").manual_seed(5)
    C, T, H, W = 16, 2, 8, 12
    N = T * (H // 2) * (W // 2)
    sch = UniPCMultistepScheduler(flow_shift=5.0)
    sch.set_timesteps(9)
    cur = torch.randn(1, C, T, H, W, device="cuda", generator=g)
    last = torch.empty_like(cur)
    m_a, m_b = torch.empty_like(cur), torch.empty_like(cur)
    tok = torch.empty(2 * N, 4 * C, device="cuda", dtype=torch.bfloat16)
    ref = UniPCMultistepScheduler(flow_shift=5.0)
    ref.set_timesteps(9, device="cuda")
    x_ref = cur.clone()
    m1 = m2 = None
    seen = set()
    for i, t in enumerate(ref.timesteps):
        out_tok = torch.randn(2 * N, 4 * C, device="cuda", generator=g).to(torch.bfloat16)
        # tensor-op reference: unpatchify, guidance in bf16, scheduler.step
        o = out_tok.view(2, T, H // 2, W // 2, 1, 2, 2, C).permute(0, 7, 1, 4, 2, 5, 3, 6).reshape(2, C, T, H, W)
        noise = o[1:2] + 7.5 * (o[0:1] - o[1:2])
        x_ref = ref.step(noise, t, x_ref)[0]
        c = sch.plan_step()
        seen.add((c["corr_order"], c["pred_order"]))
        m_out = m_b if m1 is m_a else m_a
        ops.unipc_cfg_step(out_tok, tok, cur, last if c["corr_order"] else None, m1, m2, m_out, last, cur, batch=2, guidance=7.5, coeffs=c)
        m2, m1 = m1, m_out
        assert torch.equal(cur, x_ref), (i, c["corr_order"], c["pred_order"], (cur - x_ref).abs().max().item())
        want_tok = x_ref.to(torch.bfloat16).expand(2, -1, -1, -1, -1).view(2, C, T, 1, H // 2, 2, W // 2, 2).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(2 * N, 4 * C)
        assert torch.equal(tok, want_tok)
    assert seen == {(0, 1), (1, 2), (2, 2), (2, 1)}, seen
