"""Headline benchmark: 3D-Gaussian scenes/sec — 50-step CFG denoise (Wan-1.3B DiT + UniPC) -> Wan VAE decode -> 512->448
resize -> stitched Conv3d -> AnySplat reconstruction (13 views @512), synthetic inputs and seeded random weights of the
production shapes (BASELINE.json configs[1]; no checkpoint is reachable offline).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus N ...        (started as a plain process it re-launches ITSELF under torch.distributed.run, 127.0.0.1 rendezvous,
                                         a free port, one rank per GPU, and still prints the one JSON line - `spawn_ranks` below)

One "step" = one complete scene.  N>1 = data parallel over prompts exactly like the reference (prompt_list[rank::world],
/root/reference/inference_t23d.py:62): every rank owns whole scenes, no data-path collective (weak scaling).
Rank 0 prints ONE JSON line (contract in the task statement) incl. `roofline` for the dominant kernel and `cpu_baseline`."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import torch

# tile name (v3a_gemm_tile_name) -> kernel symbol as rocprofv3 prints it
SYMBOL_OF_TILE = {"pp_np3_ratrue_l5": "gemm_pp_kernel<3, true, 5, false, false>", "pp_np3_rafalse_l5": "gemm_pp_kernel<3, false, 5, false, false>",
                  "pp_np4_ratrue_l7": "gemm_pp_kernel<4, true, 7, false, false>", "pp_np3_ratrue_l5_tt": "gemm_pp_kernel<3, true, 5, false, true>"}
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
FP8_MFMA_PEAK_TFLOPS = 5000.0   # dense fp8 (--dtype fp8 only)


def pmc_traffic(kernel_symbol: str):
    """HBM bytes per launch of the dominant kernel from the LATEST committed PMC pass (profiles/rNN/pmc_traffic.json, produced by
    tools/pmc_traffic.sh: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied).  bench.py cannot run rocprofv3
    around itself, so this is the value of the same command profiled at round time.  Returns (bytes, source file) - and (None, reason)
    when the newest file does not carry exactly the symbol this run's dominant tile resolves to: a stale figure for another tile must
    not be printed beside a new kernel."""
    files = sorted(ROOT.glob("profiles/r*/pmc_traffic.json"), key=lambda f: int("".join(c for c in f.parent.name if c.isdigit()) or 0))
    if not files:
        return None, "no profiles/rNN/pmc_traffic.json"
    f = files[-1]
    try:
        k = json.loads(f.read_text())["kernels"]
    except Exception as e:  # noqa: BLE001
        return None, f"{f}: {e}"
    want = kernel_symbol.replace(" ", "")
    for name, v in k.items():
        if name.replace(" ", "") == want:
            return v["hbm_bytes_per_launch"], str(f.relative_to(ROOT))
    return None, f"{f.relative_to(ROOT)} holds no row for {kernel_symbol} (profiled before the dominant tile changed: re-run tools/pmc_traffic.sh)"


def dit_flops_per_forward(N, d, ffn, L, ctx=512):
    """projections + attention + FFN, multiply-add = 2."""
    return L * (8 * N * d * d + 4 * N * N * d + (4 * N * d * d + 4 * ctx * d * d) + 4 * N * ctx * d + 4 * N * d * ffn)


def cpu_baseline(cfg, mode: str, repeats: int = 3, dit_blocks: int = 6):
    """The CPU oracle (fp32 restatement of the reference path, `kind: "port"`) timed on this host's cores, stage by stage.  The DiT
    forward is 97 % of a scene's CPU time, and single shots of it swung 1.7x between boxes: it is timed `repeats` times on a BOUNDED
    sample - the full-size forward truncated to `dit_blocks` of its 30 identical blocks - and the minimum is scaled by 30 / dit_blocks
    (patchify / head are < 0.1 %).  VAE decode and reconstruction (3 % together) are single full-size shots.  All at
    the production shapes of config #1/#2: one full 30-block DiT forward (N = 4096 tokens, B = 1), one Wan VAE
    decode (latent [1,16,4,64,64] -> 13 x 512^2) and one stitched reconstruction forward (13 views @448, 22 DINO + 48 aggregator
    blocks at width 1024, heads, voxel fusion).  A scene is 2 x steps DiT forwards + one decode + one reconstruction: only that
    multiplication is extrapolated.  mode = "full" (about 2-3 minutes of CPU) | "dit" (the DiT forward only, ~40 s)."""
    from oracle import recon as R
    from oracle import wan_dit as O
    from oracle import wan_vae as V
    n = torch.get_num_threads()
    g = torch.Generator().manual_seed(0)
    ocfg = O.WanDiTConfig(num_attention_heads=cfg.num_attention_heads, attention_head_dim=cfg.attention_head_dim, ffn_dim=cfg.ffn_dim,
                          num_layers=cfg.num_layers, text_dim=cfg.text_dim, freq_dim=cfg.freq_dim)
    sd = O.make_weights(ocfg, seed=0)
    lat = torch.randn(1, 16, 4, 64, 64, generator=g)
    text = torch.randn(1, 512, cfg.text_dim, generator=g) * 0.1
    t = torch.tensor([900])
    stages, runs = {}, {}

    def timed(name, fn, n):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        stages[name], runs[name] = min(ts), [round(x, 2) for x in ts]
        return r

    with torch.no_grad():
        O.dit_forward(sd, ocfg, lat[:, :, :1, :16, :16], t, text, num_layers=1)  # warm the thread pool on a small clip
        nb = min(dit_blocks, cfg.num_layers)
        timed("dit_sample_s", lambda: O.dit_forward(sd, ocfg, lat, t, text, num_layers=nb), repeats)
        stages["dit_forward_s"] = stages["dit_sample_s"] * cfg.num_layers / nb
        del sd
        if mode == "full":
            vcfg = V.WanVAEConfig()
            vsd = V.make_weights(vcfg, seed=1)
            img = timed("vae_decode_s", lambda: V.decode(vsd, vcfg, lat), 1)
            del vsd
            rcfg = R.ReconCfg()
            rsd = R.make_recon_weights(rcfg, seed=2)
            w = torch.randn(rcfg.C, 16, 5, 3, 3, generator=g) * 0.02
            img448 = torch.nn.functional.interpolate(img[0].permute(1, 0, 2, 3).clamp(-1, 1), size=(448, 448), mode="bilinear", align_corners=False)
            img448 = img448.permute(1, 0, 2, 3)[None]
            timed("stitch_recon_s", lambda: R.recon_forward(rsd, rcfg, R.stitch_conv(R.upsample_T(lat), w, torch.zeros(rcfg.C), (1, 2, 2), (2, 1, 1)), img448), 1)
            del rsd
    tail = stages.get("vae_decode_s", 0.0) + stages.get("stitch_recon_s", 0.0)
    scene50 = 100 * stages["dit_forward_s"] + tail
    scene10 = 20 * stages["dit_forward_s"] + tail
    what = (f"DiT: min of {repeats} runs of {nb} of {cfg.num_layers} blocks at full size, x {cfg.num_layers}/{nb}; "
            + ("one VAE decode + one reconstruction forward, single shots" if mode == "full" else "VAE decode + reconstruction (3 % of the time) not timed"))
    return dict(value=1.0 / scene50, unit="scenes/s", cores=n, kind="port",
                sample=f"oracle fp32 on {n} threads, production shapes ({what}); scene = 100 x DiT forward + decode + recon",
                stage_seconds={k: round(v, 2) for k, v in stages.items()}, stage_runs_seconds=runs,
                scene_seconds_50_steps=round(scene50, 1),
                config1_10_steps={"scene_seconds": round(scene10, 1), "scenes_per_s": 1.0 / scene10,
                                  "note": "BASELINE config #1 (10 denoise steps) = 20 x DiT + decode + recon"})


def dit_flops_executed(N, d, ffn, L, ctx_keys: int, heads_x_padded_keys: int = 0):
    """FLOPs the product actually executes per forward: the text context K / V^T projections are cached per prompt (not per step),
    and cross-attention runs over the `ctx_keys` real + merged-padding keys instead of 512.  `heads_x_padded_keys` > 0: the
    cached-context form (WanDiT.ctx_vo) - to_q, scores, then one GEMM with K = heads x padded keys instead of P.V + the to_out GEMM."""
    if heads_x_padded_keys:
        cross = 2 * N * d * d + 2 * N * ctx_keys * d + 2 * N * heads_x_padded_keys * d
    else:
        cross = 4 * N * d * d + 4 * N * ctx_keys * d
    return L * (8 * N * d * d + 4 * N * N * d + cross + 4 * N * d * ffn)


def coop_requested(a, world):
    return a.parallel == "scene" and world > 1


def parse_rocm_smi(text):
    """(sclk MHz, socket power W) from `rocm-smi --showclocks --showpower --json` (first card of the output), or None."""
    import re
    try:
        card = next(iter(json.loads(text).values()))
    except Exception:
        return None
    mhz = [int(re.search(r"(\d+)\s*Mhz", str(v), re.I).group(1)) for k, v in card.items() if "sclk" in k.lower() and re.search(r"\d+\s*Mhz", str(v), re.I)]
    watts = []
    for k, v in card.items():
        if "power (w)" in k.lower():
            try:
                watts.append(float(v))
            except (TypeError, ValueError):
                pass
    return (mhz[0], watts[0]) if mhz and watts else None


def sustained_clock(load, device_index, samples=4):
    """Median shader clock / socket power rocm-smi reports while `load()` (one untimed scene) is repeated on the GPU."""
    import subprocess
    import threading
    got = []

    def sampler():
        time.sleep(0.4)   # the scene is in its denoise loop by then
        for _ in range(samples):
            try:
                r = subprocess.run(["rocm-smi", "-d", str(device_index), "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20)
                one = parse_rocm_smi(r.stdout)
                if one:
                    got.append(one)
            except Exception:
                pass

    th = threading.Thread(target=sampler)
    th.start()
    n = 0
    while th.is_alive() and n < 8:
        load()
        torch.cuda.synchronize()
        n += 1
    th.join()
    if not got:
        return None
    got.sort()
    mhz, watts = got[len(got) // 2][0], sorted(w for _, w in got)[len(got) // 2]
    return {"sclk_mhz": mhz, "socket_power_w": watts, "samples": len(got), "scenes_run": n,
            "note": "untimed extra; the MFMA roof at this clock is 2500 x sclk / 2400 TFLOP/s"}


def spawn_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` started WITHOUT a launcher: run this very command line under torch.distributed.run (one rank per GPU of
    this node, rendezvous on 127.0.0.1 - the container hostname may not resolve - on a port the kernel just handed out), pass the children's
    stdout / stderr through, return their exit code.  Rank 0 of the children prints the one JSON line."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL / tensor sharing across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    return subprocess.run(cmd, env=env).returncode


def timed_steps(step, steps: int, warmup: int, sync, world: int, reduce_max):
    """The measurement contract shared by the real run and the launch self-test: `warmup` untimed steps, then EXACTLY `steps` steps between
    two (barrier + device synchronise) brackets; the time reported is the MAXIMUM over ranks."""
    for i in range(warmup):
        step(-1 - i, False)
    sync()
    t0 = time.perf_counter()
    out = None
    for i in range(steps):
        out = step(i, i == steps - 1)
    sync()
    dt = time.perf_counter() - t0
    return (reduce_max(dt) if world > 1 else dt), dt, out


def launch_self_test(a, world: int, rank: int) -> None:
    """--launch-self-test: the rendezvous / barrier / max-over-ranks / one-line skeleton of this file on the CPU (gloo) around a stub step
    that only sleeps - what tests/test_host_logic.py runs with --gpus 2 to prove that a launcher-less `python bench.py --gpus N` comes
    up as N ranks and prints ONE line.  Not a measurement: the line says so."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")

    def sync():
        if world > 1:
            dist.barrier()

    def reduce_max(dt):
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item()

    seen = []
    dt, _, _ = timed_steps(lambda i, last: (time.sleep(0.01 * (1 + rank)), seen.append(i))[1], a.steps, a.warmup, sync, world, reduce_max)
    counts = [None] * world
    if world > 1:
        dist.all_gather_object(counts, len(seen))
    else:
        counts = [len(seen)]
    if rank == 0:
        print(json.dumps({"metric": "launch self-test (stub step, no model, no GPU)", "value": world * a.steps / dt, "unit": "stub steps/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none (launch self-test)",
                          "config": {"workload": "stub", "steps_run_per_rank_incl_warmup": counts,
                                     "rccl": {"backend": dist.get_backend() if world > 1 else "none", "world_size": world,
                                              "launched_by": os.environ.get("V3A_BENCH_LAUNCHED_BY", "external launcher")}}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--denoise-steps", type=int, default=50)
    ap.add_argument("--num-frames", type=int, default=13)
    ap.add_argument("--cpu-baseline", choices=["full", "dit", "none"], default="full",
                    help="CPU oracle timed once per stage on this host (full: ~2-3 min of CPU after the timed GPU region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--model", choices=["1.3b", "14b"], default="1.3b",
                    help="14b = BASELINE config #4 geometry (Wan-14B, 40 x 128 heads, FFN 13824, 40 blocks); NOT the headline config")
    ap.add_argument("--dtype", choices=["bf16", "fp8", "fp8-attn"], default="bf16",
                    help="fp8: config #4's precision mode - DiT self-attention AND the block projection / FFN GEMMs on the e4m3 MFMA "
                         "(per-token / per-channel scales); fp8-attn: the attention only, GEMMs stay bf16")
    ap.add_argument("--sp-graph", action="store_true",
                    help="--parallel scene only: replay every rank's sharded DiT forward (RCCL all-gathers included) from a captured hipGraph")
    ap.add_argument("--parallel", choices=["dp", "scene"], default="dp",
                    help="dp: one prompt per GPU, no data-path collective (the reference's split; the headline metric). "
                         "scene: all ranks cooperate on ONE scene (CFG-parallel x sequence-parallel DiT over RCCL; latency mode)")
    ap.add_argument("--launch-self-test", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a plain `python bench.py --gpus N` (the way the driver calls --gpus 1): become the launcher of our own N ranks
        os.environ["V3A_BENCH_LAUNCHED_BY"] = "bench.py itself (no launcher in the environment)"
        raise SystemExit(spawn_ranks(a.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks: use --nproc-per-node {a.gpus}")
    if a.launch_self_test:
        return launch_self_test(a, world, rank)
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device(dev))

    from vist3a_amd import ops
    from vist3a_amd import lib
    from vist3a_amd.t23d import SceneTimes, Text23DGS, synthetic_text_embeddings
    from vist3a_amd.wan.dit import WAN_1_3B, WAN_14B
    lib.load()
    cfg = WAN_14B if a.model == "14b" else WAN_1_3B
    model = Text23DGS.synthetic(cfg, seed=0, device=dev)
    model.transformer.attn_dtype = "bf16" if a.dtype == "bf16" else "fp8"   # (sharded runs: e4m3 K | V^T slabs are what is all-gathered)
    if a.dtype == "fp8":
        model.transformer.enable_fp8_gemm()
    pe, ne = synthetic_text_embeddings(dev)
    coop = a.parallel == "scene" and world > 1
    if coop:
        from vist3a_amd.wan.seqpar import DenoisePlan
        model.pipe.plan = DenoisePlan.from_dist()
        model.stitched_decoder.recon_group = model.pipe.plan.world      # reconstruction split by views over all ranks of the scene
        if a.sp_graph:
            from vist3a_amd.wan.dit import GraphedWanDiT
            model.pipe.transformer = GraphedWanDiT(model.transformer, capture_sp=True)
    Tl = (a.num_frames - 1) // 4 + 1
    N = Tl * 32 * 32

    def scene(i, timings=None, stage_flops=None):
        # the reference seeds once per process and strides prompts over ranks: scene i of this rank = global prompt i*world+rank
        g = torch.Generator().manual_seed(12413 + (i if coop else i * world + rank))
        lat0 = torch.randn(1, 16, Tl, 64, 64, generator=g)
        out, _, _ = model.generate(pe, ne, latents=lat0, num_frames=a.num_frames, num_inference_steps=a.denoise_steps,
                                   guidance_scale=7.5, timings=timings, stage_flops=stage_flops)
        return out

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # dominant kernel = the bf16 GEMM tile every N=1536/3072/8960-wide projection resolves to (one kernel symbol)
    f8 = a.dtype == "fp8"   # then the dominant kernel is the e4m3 form of the same tile
    dom_tile = lib.load().v3a_gemm_fp8_pick_tile(2 * N, cfg.dim) if f8 else lib.load().v3a_gemm_pick_tile(2 * N, cfg.dim)
    # every 7th launch of the symbol is bracketed by events.  Per DiT block the symbol runs FIVE times - self-attention out-projection,
    # cross-attention to_q, the batched cached-context GEMM (two per-prompt operands in one launch), FFN2 and the two-round q|k
    # projection - and 7 is coprime to 5, so every shape is sampled equally: 30 blocks x 5 x 50 (CFG-batched) forwards / 7 = ~1075 samples per scene,
    # and the event pairs do not cost the probed scene 3 % of its time as bracketing every launch did
    probe = ops.GemmProbe(dom_tile, fp8=f8, stride=7)
    ops.set_gemm_probe(probe)
    stage = SceneTimes()

    def step(i, last):
        probe.active = last and i >= 0   # HIP-event pairs around the sampled launches of the dominant kernel during the last timed scene
        return scene(i, stage if (last and i >= 0) else None)

    def reduce_max(t):
        tt = torch.tensor([t], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return tt.item()

    dt_max, dt, out = timed_steps(step, a.steps, a.warmup, sync, world, reduce_max)
    probe.active = False
    ops.set_gemm_probe(None)
    per_rank = rccl_info = None
    if world > 1:
        # what torch.distributed actually ran on: a SCALE run shows the rank count / backend it saw without a code change
        rccl_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "gpus_visible_to_rank0": torch.cuda.device_count(),
                     "mode": "scene (CFG x sequence parallel, RCCL all-gathers on the data path)" if coop else "dp (no data-path collective)",
                     "launched_by": os.environ.get("V3A_BENCH_LAUNCHED_BY", "external launcher (torch.distributed.run)")}
        # every rank's wall time and stage times of its last scene (not part of the timed region): shows WHICH stage or rank is slow
        mine = torch.tensor([dt, stage.denoise_ms, stage.vae_ms, stage.recon_ms], device=dev, dtype=torch.float64)
        allr = torch.empty(world, 4, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allr.view(-1), mine)
        per_rank = [{"rank": r, "wall_s": round(v[0], 3), "denoise_ms": round(v[1], 1), "vae_ms": round(v[2], 1), "recon_ms": round(v[3], 1)}
                    for r, v in enumerate(allr.cpu().tolist())]
        if coop:   # the view-sharded reconstruction of the last scene, stage by stage, on every rank (the Amdahl term of the latency mode)
            st = getattr(model.stitched_decoder, "recon_shard_times", None) or {}
            mine = torch.tensor([st.get("views", [0, 0])[0], st.get("backbone_ms", 0.0), st.get("camera_ms", 0.0), st.get("heads_ms", 0.0),
                                 st.get("gather_and_tail_ms", 0.0)], device=dev, dtype=torch.float64)
            allr = torch.empty(world, 5, device=dev, dtype=torch.float64)
            dist.all_gather_into_tensor(allr.view(-1), mine)
            for pr, v in zip(per_rank, allr.cpu().tolist()):
                pr["recon_view_shard"] = {"views": int(v[0]), "backbone_ms": round(v[1], 1), "camera_replicated_ms": round(v[2], 1),
                                          "heads_ms": round(v[3], 1), "gather_and_tail_ms": round(v[4], 1)}
    dt = dt_max
    comm = None
    if coop:
        # RCCL time of the sequence-parallel all-gathers on their own: two extra denoise steps, untimed, with every all-gather run
        # synchronously between events (in the timed region they overlap the Q projection)
        plan = model.pipe.plan
        groups = [g for g in (plan.sp, plan.cfg) if g is not None]
        for g in groups:
            g.profile = True
        gq = torch.Generator().manual_seed(1)
        model.generate(pe, ne, latents=torch.randn(1, 16, Tl, 64, 64, generator=gq), num_frames=a.num_frames, num_inference_steps=2, guidance_scale=7.5)
        names = [n for n, g in (("sequence_parallel_kv", plan.sp), ("cfg_pair_noise", plan.cfg)) if g is not None]
        comm = {n: g.profile_summary() for n, g in zip(names, groups)}
        for n in comm:
            comm[n] = {"calls_per_step": comm[n]["calls"] // 2, "ms_per_step": round(comm[n]["ms"] / 2, 3),
                       "gathered_MB_per_step": round(comm[n]["gathered_bytes"] / 2 / 1e6, 2)}
        for g in groups:
            g.profile = False
    # the step after the path, reported beside the metric, never inside it: orbit render of the last scene
    render_ms = None
    if rank == 0:
        from vist3a_amd.misc.image_io import interpolate_camera_path
        ex, ix = interpolate_camera_path(out.pred_context_pose["extrinsic"], out.pred_context_pose["intrinsic"], 1, 10)
        dec = model.stitched_decoder.stitched_3d_model.decoder
        ones = torch.ones(1, ex.shape[1], device=dev)
        for _ in range(2):
            torch.cuda.synchronize()
            tr = time.perf_counter()
            dec.forward(out.gaussians, ex, ix.float(), ones * 0.1, ones * 100, (448, 448))
            torch.cuda.synchronize()
            render_ms = (time.perf_counter() - tr) * 1e3
    # R13 / R15 / rasteriser at a REALISTIC point distribution (untimed extra): seeded synthetic weights collapse the 2.6 M points of
    # a scene into ~35 k voxels (near-constant depth, near-identical poses); real checkpoints keep 1-2 M.  Same kernels, a spread cloud.
    tail = None
    if rank == 0:
        from vist3a_amd.models.types import Gaussians
        gen = torch.Generator(device=dev).manual_seed(7)
        M = a.num_frames * 448 * 448
        pts = torch.randn(M, 3, device=dev, generator=gen) * 0.25
        pts[:, 2] += 1.5
        raw = torch.randn(M, 84, device=dev, generator=gen) * 0.5
        eng = model.stitched_decoder.stitched_3d_model.engine()

        def timed(fn, reps=3):
            fn()
            torch.cuda.synchronize()
            tq = time.perf_counter()
            for _ in range(reps):
                r = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - tq) / reps * 1e3, r

        vox_ms, v = timed(lambda: ops.voxelize_fuse(pts, raw, 83, 83, 0.002))
        ad_ms, gs = timed(lambda: ops.gaussian_adapter(v["voxel_pts"], v["voxel_feat"], eng.sh_mask, 4, 1.0))
        gsp = Gaussians(**{k: t[None] for k, t in gs.items()})
        rd_ms, _ = timed(lambda: dec.forward(gsp, ex, ix.float(), ones * 0.1, ones * 100, (448, 448)), reps=1)
        tail = {"points": M, "voxels": int(v["keys"].shape[0]), "voxelize+fuse_ms": round(vox_ms, 2), "gaussian_adapter_ms": round(ad_ms, 2),
                "orbit_render_ms_132_cameras_448": round(rd_ms, 1)}
        del pts, raw, v, gs, gsp
    # boxes of one pool differ by up to 5 % on identical binaries (profiles/r3/README.md): two fixed launches of the product library,
    # untimed extras, so that lines from different boxes can be put side by side
    box = None
    if rank == 0:
        gb = torch.Generator(device=dev).manual_seed(1)
        Mb = 8192
        xa, xw = torch.randn(Mb, 1536, device=dev, generator=gb).bfloat16(), (torch.randn(1536, 1536, device=dev, generator=gb) / 39.0).bfloat16()
        qb_, kb_ = (torch.randn(Mb, 1536, device=dev, generator=gb) * 0.5).bfloat16(), (torch.randn(Mb, 1536, device=dev, generator=gb) * 0.5).bfloat16()
        vtb, ob = torch.randn(1536, Mb, device=dev, generator=gb).bfloat16(), torch.empty(Mb, 1536, device=dev, dtype=torch.bfloat16)

        def best_us(fn, n=20, rounds=4):
            for _ in range(5):
                fn()
            bt = 1e9
            for _ in range(rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                bt = min(bt, e0.elapsed_time(e1) / n * 1e3)
            return bt
        g_us = best_us(lambda: ops.gemm(xa, xw, None, out=ob))
        a_us = best_us(lambda: ops.attention(qb_, kb_, vtb, ob, B=2, H=12, Nq=4096, Nk=4096, D=128, q_batch_stride=4096 * 1536, k_batch_stride=4096 * 1536,
                                            vt_batch_stride=4096, o_batch_stride=4096 * 1536))
        box = {"gemm_8192x1536x1536_us": round(g_us, 1), "self_attention_2x12x4096_us": round(a_us, 1),
               "note": "untimed extras: best of 4 x 20 back-to-back launches on seeded data; the same launches read 38-41 / 184-192 us across the boxes of round 3"}
        del xa, xw, qb_, kb_, vtb, ob
        # what the boxes differ in: the shader clock the firmware sustains under THIS load (the roofline's 2.5 PFLOP/s is quoted at 2.4 GHz).
        # rocm-smi is sampled from a thread while untimed extra scenes run; any failure leaves the field null.
        box["clock_under_load"] = None if coop else sustained_clock(lambda: scene(a.steps), local)
    # algorithmic FLOPs of every matrix-pipe launch of one scene, stage by stage (an untimed extra scene with ops.FlopMeter installed; the
    # denoise loop is eager unless V3A_GRAPH=1, so its launches are counted too): the stage-level roofline fractions beside the kernel-level one
    stage_flops = {}
    if rank == 0 and not coop:
        scene(a.steps + 1, None, stage_flops)
        torch.cuda.synchronize()
    if rank == 0:
        ps = probe.summary()
        ach = ps["flops_per_launch"] / (ps["avg_ms"] * 1e-3) / 1e12 if ps["launches"] else 0.0
        fwd_flops = dit_flops_per_forward(N, cfg.dim, cfg.ffn_dim, cfg.num_layers)
        ctx_keys = (64 + 1 + 80 + 1) // 2   # synthetic prompts: 64 / 80 real tokens + one merged padding key each (cond / uncond)
        tr = model.transformer
        ent = next(iter(tr._ctx.values()))[1] if getattr(tr, "_ctx", None) else None
        hxk = cfg.num_attention_heads * ent[8] if (ent is not None and ent[7] is not None) else 0   # cached-context form active: heads x padded keys
        U = int(out.gaussians.means.shape[1])
        dom_symbol = SYMBOL_OF_TILE.get(lib.load().v3a_gemm_tile_name(dom_tile).decode()) if not f8 else None
        traffic, traffic_src = (None, "fp8 mode: not profiled") if f8 else (pmc_traffic(dom_symbol) if dom_symbol else (None, "unknown tile symbol"))
        line = {
            "metric": "3D Gaussian scenes/sec (50-step denoise, 512^2, 13 views)",
            "value": (1 if coop else world) * a.steps / dt, "unit": "scenes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong" if coop else "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp8-attn": "fp8 (e4m3 self-attention operands; bf16 GEMMs)",
                      "fp8": "fp8 (e4m3 self-attention and block projection / FFN GEMM operands, fp32 accumulation; bf16 elsewhere)"}[a.dtype],
            "data": "synthetic (seeded random weights of production shapes, synthetic text embeddings)",
            "config": {"workload": f"Wan-{'14B' if a.model == '14b' else '1.3B'} stitched, {a.denoise_steps}-step CFG denoise (batch-2 cond/uncond), {a.num_frames} views @512, "
                                   "VAE decode, 448^2 AnySplat enc_blocks_2 reconstruction with voxel fusion; "
                                   + ("one scene over all GPUs (CFG-parallel x sequence-parallel)" if coop else "1 prompt per GPU (data parallel)"),
                       "denoise_steps": a.denoise_steps, "views": a.num_frames, "dit_tokens": N, "gaussians_last_scene": U,
                       "precision": {"dit / vae / recon backbone": "bf16 MFMA, fp32 accumulate (the reference's autocast dtype)",
                                     "camera head": "fp32",
                                     "depth + gaussian DPT heads": ("fp32-equivalent (split-bf16 MFMA: 3 products, fp32 accumulate, fp32 activations as bf16 pairs) - the reference "
                                                                    "runs them with autocast off" if model.stitched_decoder.stitched_3d_model.engine().cfg.dpt_precision == "f32"
                                                                    else "bf16 (opt-in deviation)")},
                       **({"per_rank_last_scene": per_rank, "rccl": rccl_info} if per_rank else {}),
                       **({"scene_parallel": {
                           "layout": dict(zip(("cfg_degree", "sp_degree"), DenoisePlan.layout(world))),
                           "sp_mode": ("exact (sp_kv_split = 1): bit-identical to the single-GPU forward" if model.transformer.sp_kv_split == 1 else
                                       "default (sp_kv_split = None): key-split attention + split-K FFN2 on the shards - deterministic, within bf16 "
                                       "rounding (5e-3) of the single-GPU forward, NOT bit-identical"),
                           "graph": ("hipGraph replay of each rank's sharded forward incl. its RCCL all-gathers (--sp-graph)" if a.sp_graph else
                                     "eager (pass --sp-graph to replay the sharded forward from a captured hipGraph)"),
                           "rccl_all_gather (2 untimed profiled steps, collectives serialised)": comm}} if coop else {}),
                       "stage_ms_last_scene": {"denoise": round(stage.denoise_ms, 1), "vae_decode+resize": round(stage.vae_ms, 1),
                                               "stitch+recon": round(stage.recon_ms, 1)},
                       "orbit_render_ms_132_cameras_448 (untimed extra)": round(render_ms, 1),
                       "recon_tail_on_spread_cloud (untimed extra)": tail,
                       "box_speed_probe (untimed extra)": box,
                       "dit_model_tflops_per_s": round(2 * a.denoise_steps * fwd_flops / (stage.denoise_ms * 1e-3) / 1e12, 1),
                       "dit_executed_tflops_per_s": round(2 * a.denoise_steps * dit_flops_executed(N, cfg.dim, cfg.ffn_dim, cfg.num_layers, ctx_keys, hxk)
                                                          / (stage.denoise_ms * 1e-3) / 1e12, 1),
                       "dit_flops_note": f"model = BASELINE.md §2 formula (512 text keys, context K/V projected every step); executed = "
                                         f"{ctx_keys} cross-attention keys after merging the zero-padding keys, context K/V cached per prompt"
                                         + (f", cross-attention in the cached-context form (to_out folded into a per-prompt V.Wo^T: one GEMM with K = {hxk})" if hxk else "")},
            "roofline": {"bound": "mfma",
                         "kernel": (f"gemm_pp_kernel<.., F8> = tile {lib.load().v3a_gemm_fp8_tile_name(dom_tile).decode()} (e4m3 MFMA 32x32x64 f8f6f4, ping-pong)" if f8 else
                                    f"{dom_symbol} = tile {lib.load().v3a_gemm_tile_name(dom_tile).decode()} (bf16 MFMA 32x32x16, ping-pong 256x192)"),
                         "achieved": round(ach, 1), "peak": FP8_MFMA_PEAK_TFLOPS if f8 else BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / (FP8_MFMA_PEAK_TFLOPS if f8 else BF16_MFMA_PEAK_TFLOPS), 4),
                         "traffic": traffic, "traffic_unit": "bytes/launch (PMC FETCH_SIZE + WRITE_SIZE of this symbol)", "traffic_source": traffic_src,
                         "launches_timed": ps["launches"], "launch_sampling": "every 7th launch of the symbol in the last timed scene", "avg_launch_ms": round(ps["avg_ms"], 4),
                         "flops_per_launch": ps["flops_per_launch"]},
        }
        pk_tf = FP8_MFMA_PEAK_TFLOPS if f8 else BF16_MFMA_PEAK_TFLOPS

        def srow(flops, ms, note, kinds=None):
            tf = flops / (ms * 1e-3) / 1e12 if ms else 0.0
            r = {"flops": flops, "ms": round(ms, 2), "achieved_tflops": round(tf, 1), "frac_of_mfma_peak": round(tf / pk_tf, 4), "what": note}
            if kinds:
                r["flops_by_kind"] = {k: round(v, 0) for k, v in sorted(kinds.items())}
            return r
        line["stage_roofline"] = {
            "peak_tflops": pk_tf,
            "dit_model": srow(2 * a.denoise_steps * fwd_flops, stage.denoise_ms, "BASELINE.md section 2 formula x 2 CFG branches x denoise steps / the denoise stage's time"),
            "dit_executed": srow(2 * a.denoise_steps * dit_flops_executed(N, cfg.dim, cfg.ffn_dim, cfg.num_layers, ctx_keys, hxk), stage.denoise_ms,
                                 "the FLOPs the product executes (merged padding keys, per-prompt context cache, cached-context cross-attention)"),
        }
        names = {"denoise": ("dit_metered", stage.denoise_ms), "vae": ("vae", stage.vae_ms), "recon": ("recon", stage.recon_ms)}
        for st_name, kinds in stage_flops.items():
            key, ms = names[st_name]
            line["stage_roofline"][key] = srow(sum(kinds.values()), ms, "sum of the algorithmic FLOPs of every GEMM / convolution / attention launch of the stage "
                                               "(ops.FlopMeter, one untimed extra scene; fp32-equivalent convolutions counted once, not as their three bf16 products) "
                                               "/ the stage's time in the last timed scene", kinds)
        clk = (box or {}).get("clock_under_load")
        if clk:   # the same achieved rate against the roof at the clock the box actually held (information beside `frac`, which stays on the guide's peak)
            pk = (FP8_MFMA_PEAK_TFLOPS if f8 else BF16_MFMA_PEAK_TFLOPS) * clk["sclk_mhz"] / 2400.0
            line["roofline"]["frac_at_sustained_clock"] = round(ach / pk, 4)
            line["roofline"]["sustained_clock_note"] = f"sclk {clk['sclk_mhz']} MHz at {clk['socket_power_w']:.0f} W under the scene's load: roof {pk:.0f} TFLOP/s"
        if world == 1 and not a.no_cpu_baseline and a.cpu_baseline != "none":
            line["cpu_baseline"] = cpu_baseline(cfg, a.cpu_baseline)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
