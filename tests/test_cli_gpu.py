"""GPU: the CLI + loader contract executed end to end.

`inference_t23d.py` is run as a subprocess the way a user runs the reference's script (/root/reference/inference_t23d.py:51-171,
Readme.md:271-286) on checkpoints written to disk in the reference's own layouts:
  * `<model_id>/transformer/{config.json, diffusion_pytorch_model.safetensors}` - a diffusers WanTransformer3DModel folder
    (reduced num_layers / width), read by `load_dit_config` + `load_dit_state_dict`;
  * `<model_id>/vae/{config.json, diffusion_pytorch_model.safetensors}` - AutoencoderKLWan (utils/wan_utils.py:904-1000);
  * `<lora>/adapter_config.json + adapter_model.safetensors` - the peft `lora_ema/` folder of train_vdm.py:32-97, read by
    `load_peft_lora` (inference_t23d.py:74-77);
  * the stitching checkpoint: torch-pickle dict {lora, stitching_layer:{weight,bias}, mask_token, cls_token, register_tokens}
    (model_stitching_training.py:59-72), read by `load_stitching_model` (nvs_eval.py:21-63);
  * an AnySplat hub-snapshot folder (config.json + model.safetensors) holding the FULL upstream encoder (all DINO blocks: the
    first two are dropped at load, anysplat_stitched.py:158-165).
Checked: the output directory contract (`prompt.txt`, `gaussians.ply` header + vertex count + finite payload, `gs.*`, `depth.*`),
that every loaded piece really reaches the result (an in-process run with the adapter / checkpoint pieces withheld differs), and
the reference's raise-on-existing-directory behaviour (`os.makedirs` without exist_ok, inference_t23d.py:125-126)."""
import json
import os
import subprocess
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import recon as R
from oracle import wan_dit as O
from oracle import wan_vae as OV

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
RECON_TINY = dict(C=64, heads=1, n_dino=22, depth=24, cam_heads=2, cam_trunk=2, features=32, oc=(16, 32, 64, 64))
DIT_TINY = dict(num_attention_heads=2, attention_head_dim=128, ffn_dim=512, num_layers=2, text_dim=4096, freq_dim=64)
SPEC = "conv3d_k5x3x3_o64_s1x2x2_p2x1x1"
LORA_CFG = "r4,a8,d0.0,f0"


def _write_assets(tmp: Path) -> SimpleNamespace:
    from safetensors.torch import save_file
    g = torch.Generator().manual_seed(123)
    rn = lambda *s, std=1.0: torch.randn(*s, generator=g) * std
    # ---- diffusers model folder: transformer + vae
    model = tmp / "Wan2.1-T2V-tiny-Diffusers"
    (model / "transformer").mkdir(parents=True)
    (model / "vae").mkdir()
    ocfg = O.WanDiTConfig(**DIT_TINY)
    dsd = {k: v.to(torch.bfloat16).contiguous() for k, v in O.make_weights(ocfg, seed=3).items()}
    save_file(dsd, str(model / "transformer" / "diffusion_pytorch_model.safetensors"))
    (model / "transformer" / "config.json").write_text(json.dumps(dict(
        _class_name="WanTransformer3DModel", _diffusers_version="0.33.1", patch_size=[1, 2, 2], in_channels=16, out_channels=16,
        cross_attn_norm=True, qk_norm="rms_norm_across_heads", eps=1e-6, image_dim=None, added_kv_proj_dim=None, rope_max_seq_len=1024,
        **DIT_TINY)))
    vcfg = OV.WanVAEConfig(base_dim=16)
    vsd = {k: v.contiguous() for k, v in OV.make_weights(vcfg, seed=5).items()}
    save_file(vsd, str(model / "vae" / "diffusion_pytorch_model.safetensors"))
    (model / "vae" / "config.json").write_text(json.dumps(dict(_class_name="AutoencoderKLWan", base_dim=16, z_dim=16, dim_mult=[1, 2, 4, 4],
                                                                num_res_blocks=2, attn_scales=[], temperal_downsample=[False, True, True], dropout=0.0)))
    # ---- peft adapter folder (lora_ema/): r=4 adapters on attn1.to_q / attn2.to_out.0 of block 0 and attn1.to_v of block 1
    lora = tmp / "lora_ema"
    lora.mkdir()
    d = ocfg.num_attention_heads * ocfg.attention_head_dim
    psd = {}
    for name in ("blocks.0.attn1.to_q", "blocks.0.attn2.to_out.0", "blocks.1.attn1.to_v"):
        psd[f"base_model.model.{name}.lora_A.weight"] = rn(4, d, std=0.3)
        psd[f"base_model.model.{name}.lora_B.weight"] = rn(d, 4, std=0.3)
    save_file(psd, str(lora / "adapter_model.safetensors"))
    (lora / "adapter_config.json").write_text(json.dumps(dict(peft_type="LORA", r=4, lora_alpha=8, lora_dropout=0.0,
                                                              target_modules=["to_q", "to_k", "to_v", "to_out.0"], base_model_name_or_path=str(model))))
    # ---- AnySplat hub snapshot: the FULL upstream encoder = two leading DINO blocks + patch-embed conv in front of the stitched weights
    rcfg = R.ReconCfg(**RECON_TINY)
    ssd = R.make_recon_weights(rcfg, seed=7)
    pe = "encoder.aggregator.patch_embed.blocks."
    full = {}
    for k, v in ssd.items():
        if k.startswith(pe):
            i, rest = k[len(pe):].split(".", 1)
            full[f"{pe}{int(i) + 2}.{rest}"] = v
            if int(i) < 2:   # two blocks of the same shapes in front (dropped at load)
                full[f"{pe}{i}.{rest}"] = rn(*v.shape, std=0.5)
        else:
            full[k] = v
    full["encoder.aggregator.patch_embed.patch_embed.proj.weight"] = rn(64, 3, 14, 14, std=0.02)
    full["encoder.aggregator.patch_embed.patch_embed.proj.bias"] = torch.zeros(64)
    snap = tmp / "anysplat"
    snap.mkdir()
    save_file({k: v.contiguous() for k, v in full.items()}, str(snap / "model.safetensors"))
    (snap / "config.json").write_text(json.dumps(dict(recon_cfg=dict(RECON_TINY, oc=list(RECON_TINY["oc"])))))
    # ---- stitching checkpoint in the reference's dict layout
    a = "encoder.aggregator."
    lsd = {}
    for layer, (o, i) in ((a + "frame_blocks.0.attn.qkv", (192, 64)), (a + "global_blocks.3.mlp.fc1", (256, 64)),
                          (a + "patch_embed.blocks.5.attn.proj", (64, 64)), ("encoder.camera_head.trunk.0.mlp.fc2", (128, 512))):
        lsd[layer + ".lora_A"] = rn(4, i, std=0.3)
        lsd[layer + ".lora_B"] = rn(o, 4, std=0.3)
        lsd[layer + ".bias"] = rn(o, std=0.05)
    ckpt = dict(lora=lsd, stitching_layer=dict(weight=rn(64, 16, 5, 3, 3, std=0.08), bias=rn(64, std=0.1)),
                mask_token=rn(1, 64, std=0.02), cls_token=rn(1, 1, 64, std=0.5), register_tokens=rn(1, 4, 64, std=0.5))
    torch.save(ckpt, tmp / "stitched_model_epoch_1.pth")
    prompts = tmp / "prompts.txt"
    prompts.write_text("a red chair in a white room\nan old/worn wooden boat on a beach\n")
    return SimpleNamespace(model=model, lora=lora, snap=snap, ckpt=tmp / "stitched_model_epoch_1.pth", prompts=prompts, stitched_sd=ssd,
                           lora_sd=lsd, ckpt_dict=ckpt, dit_sd=dsd, peft_sd=psd)


def _cli(a: SimpleNamespace, out: Path, extra=()):
    cmd = [sys.executable, str(ROOT / "inference_t23d.py"), "--model_id", str(a.model), "--checkpoint_path", str(a.ckpt),
           "--transformer_lora_path", str(a.lora), "--input_texts_path", str(a.prompts), "--output_dir", str(out),
           "--anysplat_weights", str(a.snap), "--stitching_layer_config", SPEC, "--lora_config", LORA_CFG, "--num_frames", "5",
           "--num_inference_steps", "2", "--synthetic_text", *extra]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)


def _read_ply(path: Path):
    raw = path.read_bytes()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    head = raw[:end].decode("ascii").split("\n")
    n = int(next(l for l in head if l.startswith("element vertex")).split()[-1])
    props = [l.split()[-1] for l in head if l.startswith("property float")]
    data = np.frombuffer(raw[end:], dtype="<f4").reshape(n, len(props))
    return head, props, data


def test_cli_runs_on_reference_layout_checkpoints(hip_lib, tmp_path):
    a = _write_assets(tmp_path)
    out = tmp_path / "results"
    r = _cli(a, out)
    assert r.returncode == 0, r.stderr[-3000:]
    dirs = sorted(p.name for p in out.iterdir())
    assert dirs == ["a red chair in a white room", "an oldworn wooden boat on a beach"]   # prompt[:100] with '/' removed (:125)
    counts = {}
    for dname, prompt in zip(dirs, a.prompts.read_text().splitlines()):
        dd = out / dname
        assert (dd / "prompt.txt").read_text() == prompt
        files = sorted(p.name for p in dd.iterdir())
        assert files[1] == "gaussians.ply" and files[3] == "prompt.txt", files
        assert files[0] in ("depth.mp4", "depth.avi") and files[2] in ("gs.mp4", "gs.avi"), files
        for v in (files[0], files[2]):
            blob = (dd / v).read_bytes()
            assert len(blob) > 10_000 and (blob[:4] == b"RIFF" or blob[4:8] == b"ftyp")
        head, props, data = _read_ply(dd / "gaussians.ply")
        assert head[0] == "ply" and head[1] == "format binary_little_endian 1.0"
        assert props == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1", "scale_2",
                         "rot_0", "rot_1", "rot_2", "rot_3"]                                            # ply_export.py:26-74, DC only
        assert 0 < data.shape[0] <= 5 * 448 * 448 and np.isfinite(data).all()
        assert np.allclose(np.linalg.norm(data[:, 13:17], axis=1), 1.0, atol=1e-4)                      # wxyz unit quaternions
        assert (data[:, 10:13] <= np.log(0.3) + 1e-5).all()                                             # log-scales, clamp 0.3
        counts[dname] = data.shape[0]
        assert f"{data.shape[0]} gaussians" in r.stdout
    # the reference raises when the output directory of a prompt exists (os.makedirs without exist_ok)
    r2 = _cli(a, out)
    assert r2.returncode != 0 and "FileExistsError" in r2.stderr
    # --overwrite (an extra of this CLI) reuses it and is deterministic: same seed, same prompts -> same Gaussian counts
    r3 = _cli(a, out, extra=("--overwrite", "--no_video"))
    assert r3.returncode == 0, r3.stderr[-3000:]
    for dname in dirs:
        assert _read_ply(out / dname / "gaussians.ply")[2].shape[0] == counts[dname]


def test_loaders_apply_every_checkpoint_piece(hip_lib, tmp_path):
    """`load_stitching_model` / `load_dit_state_dict` / `load_peft_lora` in process: what they build equals a model assembled by hand
    from the same tensors (LoRA merged by the golden-pinned merge functions), and differs from one without the trained pieces."""
    from vist3a_amd.models.loading import load_stitching_model
    from vist3a_amd.models.stitching_layer_builder import parse_conv_spec
    from vist3a_amd.wan.dit import WanDiT, merge_lora_into_state_dict
    from vist3a_amd.wan.weights import load_dit_config, load_dit_state_dict, load_peft_lora
    from vist3a_amd.wan.dit import WAN_1_3B
    a = _write_assets(tmp_path)
    args = SimpleNamespace(feedforward_model="anysplat", video_model="wan", stitching_layer_location="enc_blocks_2",
                           stitching_layer_config=parse_conv_spec(SPEC), resolution=64, initialization_weight_path=None, lora_config=LORA_CFG,
                           checkpoint_path=str(a.ckpt), anysplat_weights=str(a.snap), model_id=str(a.model))
    model = load_stitching_model(args)
    assert model.diffusion_vae.cfg.base_dim == 16 and model.stitched_3d_model._cfg.C == 64
    ck = a.ckpt_dict
    assert torch.equal(model.stitching_layer.weight.data, ck["stitching_layer"]["weight"])
    pe = model.stitched_3d_model.encoder.aggregator.patch_embed
    assert torch.equal(pe.cls_token.data, ck["cls_token"]) and torch.equal(pe.register_tokens.data, ck["register_tokens"])
    sd = model.stitched_3d_model._sd
    # the two leading DINO blocks were dropped and the rest re-indexed: block 0 of the engine is block 2 of the snapshot
    assert torch.equal(sd["encoder.aggregator.patch_embed.blocks.0.norm1.weight"], a.stitched_sd["encoder.aggregator.patch_embed.blocks.0.norm1.weight"])
    # any other stitching location drops that many leading blocks of the SAME checkpoint (anysplat_stitched.py:158-165): n_dino follows
    m3 = load_stitching_model(SimpleNamespace(**{**vars(args), "stitching_layer_location": "enc_blocks_3"}))
    assert m3.stitched_3d_model._cfg.n_dino == model.stitched_3d_model._cfg.n_dino - 1
    assert torch.equal(m3.stitched_3d_model._sd["encoder.aggregator.patch_embed.blocks.0.norm1.weight"],
                       a.stitched_sd["encoder.aggregator.patch_embed.blocks.1.norm1.weight"])
    del m3
    # LoRA: W + (alpha / r) B A, trained bias replaces the base bias
    k = "encoder.aggregator.frame_blocks.0.attn.qkv"
    want = a.stitched_sd[k + ".weight"] + (a.lora_sd[k + ".lora_B"] @ a.lora_sd[k + ".lora_A"]) * (8 / 4)
    assert torch.allclose(sd[k + ".weight"].float(), want, atol=1e-6) and torch.equal(sd[k + ".bias"].float(), a.lora_sd[k + ".bias"])
    # forward: tokens / LoRA from the checkpoint change the result
    g = torch.Generator().manual_seed(4)
    lat = torch.randn(1, 16, 2, 8, 8, generator=g)
    img = torch.rand(1, 3, 5, 56, 56, generator=g) * 2 - 1
    out = model.forward_with_latent(lat.cuda(), img.cuda())
    ocfg = R.ReconCfg(**RECON_TINY)
    ref_sd = {kk: v.clone() for kk, v in a.stitched_sd.items()}
    from vist3a_amd.recon.weights import merge_lora
    assert merge_lora(ref_sd, a.lora_sd, 8.0, 4) == 4
    ref_sd["encoder.aggregator.patch_embed.cls_token"], ref_sd["encoder.aggregator.patch_embed.register_tokens"] = ck["cls_token"], ck["register_tokens"]
    with torch.no_grad():
        feat = R.stitch_conv(R.upsample_T(lat), ck["stitching_layer"]["weight"], ck["stitching_layer"]["bias"], (1, 2, 2), (2, 1, 1))
        ora = R.recon_forward(ref_sd, ocfg, feat, img)
        ora0 = R.recon_forward(a.stitched_sd, ocfg, feat, img)
    rel = lambda x, y: ((x.float().cpu() - y.float().cpu()).norm() / y.float().cpu().norm()).item()
    e, e0 = rel(out.depth_dict["depth"], ora["depth"]), rel(out.depth_dict["depth"], ora0["depth"])
    print(f"loader: depth vs oracle with the checkpoint pieces {e:.2e}, without {e0:.2e}")
    assert e < 1e-2 and e0 > 3 * e
    # DiT: config.json + safetensors + peft folder
    cfg = load_dit_config(str(a.model), WAN_1_3B)
    assert cfg.num_layers == 2 and cfg.num_attention_heads == 2 and cfg.ffn_dim == 512 and tuple(cfg.patch_size) == (1, 2, 2)
    dsd = load_dit_state_dict(str(a.model))
    assert set(dsd) == set(a.dit_sd) and all(torch.equal(dsd[kk], a.dit_sd[kk]) for kk in dsd)
    base = {kk: v.clone() for kk, v in dsd.items()}
    assert load_peft_lora(str(a.lora), dsd) == 3
    kq = "blocks.0.attn1.to_q"
    want = base[kq + ".weight"].float() + 2.0 * (a.peft_sd[f"base_model.model.{kq}.lora_B.weight"] @ a.peft_sd[f"base_model.model.{kq}.lora_A.weight"])
    assert torch.allclose(dsd[kq + ".weight"].float(), want, atol=1e-6)
    dit = WanDiT(cfg, dsd, device="cuda")
    x = torch.randn(2, 16, 2, 16, 16, generator=g).to(torch.bfloat16)
    text = (torch.randn(2, 32, 4096, generator=g) * 0.1).to(torch.bfloat16)
    t = torch.tensor([500, 500])
    y = dit(x.cuda(), t.cuda(), text.cuda())[0].float().cpu()
    ocf = O.WanDiTConfig(**DIT_TINY)
    fsd = {kk: v.float() for kk, v in base.items()}
    fsd.update({kk.replace("base_model.model.", ""): v for kk, v in a.peft_sd.items()})   # the oracle runs the adapter UNMERGED (peft form)
    fsd["lora_scaling"] = torch.tensor(8 / 4)
    ref = O.dit_forward(fsd, ocf, x.float(), t, text.float(), emulate_bf16=True)
    ref0 = O.dit_forward({kk: v.float() for kk, v in base.items()}, ocf, x.float(), t, text.float(), emulate_bf16=True)
    ed, ed0 = rel(y, ref), rel(y, ref0)
    print(f"loader: DiT vs oracle with the adapter {ed:.2e}, without {ed0:.2e}")
    assert ed < 6e-3 and ed0 > 3 * ed


def test_bench_emits_the_contract_line(hip_lib):
    """`bench.py` end to end on a shortened schedule (2 denoise steps, one scene, no CPU baseline leg): exactly one JSON line on stdout with
    the driver's fields, the BASELINE metric / workload, the `roofline` object of the dominant kernel with a live-measured fraction, and a
    value consistent with ms_per_step."""
    import json, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "1", "--warmup", "0", "--denoise-steps", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=str(root))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["unit"] == "scenes/s" and j["n_gpus"] == 1 and j["steps"] == 1 and j["warmup"] == 0 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "bf16" and "synthetic" in j["data"]
    assert "Wan-1.3B" in j["config"]["workload"] and j["config"]["views"] == 13 and j["config"]["denoise_steps"] == 2 and "model" not in j["config"]
    assert abs(j["value"] * j["ms_per_step"] / 1e3 - 1.0) < 1e-6
    rf = j["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0 and 0.2 < rf["frac"] < 0.8 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] is None or rf["traffic"] > 0
    assert "cpu_baseline" not in j      # (--no-cpu-baseline; the default run adds the object: profiles/rNN/bench_default_run.json)


def test_graft_entry_build_then_smoke_in_one_process(hip_lib):
    """`python __graft_entry__.py smoke` = build() followed by smoke() in ONE process: the library must come up on torch's HIP runtime even
    though build() maps it before anything touches the GPU (lib.load imports torch first; with the system runtime mapped first the first
    launch failed with V3A_ERR_LAUNCH)."""
    import subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, str(root / "__graft_entry__.py"), "smoke"], capture_output=True, text=True, timeout=900, cwd=str(root))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "[smoke] DiT rel err vs oracle" in r.stdout and "voxel keys bit-exact" in r.stdout
