"""Wan-2.1 DiT forward on MI355X — host orchestration over the C-ABI kernels (vist3a_amd.ops).

Mirrors the call surface the reference drives through diffusers==0.33.1 (not vendored under /root/reference):
    transformer(hidden_states[B,16,T,H,W], timestep[B], encoder_hidden_states[B,L,4096], return_dict=False)[0]
(call sites /root/reference/inference_t23d.py:94-103 via WanPipeline, /root/reference/train_vdm.py:598-603) and the
diffusers state-dict layout (`blocks.{i}.attn1.to_q.weight`, `…ffn.net.0.proj…`, `condition_embedder.*`,
`patch_embedding`, `proj_out`, `scale_shift_table`) so real checkpoints load unchanged.

MI355X-first choices (none of them inherited from the PyTorch module graph):
  * one fused [to_q;to_k] projection, and V produced directly TRANSPOSED (V^T = Wv·X^T) by the same NT GEMM so the
    attention kernel's PV operand is a plain 16-byte LDS read;
  * AdaLN gates / residual adds / GELU live in GEMM epilogues; LayerNorm+modulation and RMSNorm+RoPE are single
    read-once/write-once row kernels;
  * cross-attention K / V^T of the (step-invariant) text context are computed once per prompt and kept resident;
  * all activations live in a preallocated workspace (no allocator traffic inside the denoise loop).
"""
from __future__ import annotations

import math
import os
import threading
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from .. import lib as L
from .. import ops

bf16, f32 = torch.bfloat16, torch.float32


@dataclass
class WanDiTConfig:
    patch_size: tuple = (1, 2, 2)
    num_attention_heads: int = 12
    attention_head_dim: int = 128
    in_channels: int = 16
    out_channels: int = 16
    text_dim: int = 4096
    freq_dim: int = 256
    ffn_dim: int = 8960
    num_layers: int = 30
    eps: float = 1e-6
    rope_max_seq_len: int = 1024

    @property
    def dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


WAN_1_3B = WanDiTConfig()
WAN_14B = WanDiTConfig(num_attention_heads=40, ffn_dim=13824, num_layers=40)


def rope_table(cfg: WanDiTConfig, ppf: int, pph: int, ppw: int, device) -> torch.Tensor:
    """(cos, sin) table [N, head_dim/2, 2] f32, float64 angles, split t/h/w = (hd/2 - 2*(hd//6), hd//6, hd//6)."""
    hd = cfg.attention_head_dim
    h_dim = w_dim = 2 * (hd // 6)
    t_dim = hd - h_dim - w_dim
    parts = []
    for dim, n, shape in ((t_dim, ppf, (ppf, 1, 1)), (h_dim, pph, (1, pph, 1)), (w_dim, ppw, (1, 1, ppw))):
        inv = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float64)[: dim // 2] / dim))
        ang = torch.outer(torch.arange(n, dtype=torch.float64), inv)
        parts.append(ang.view(*shape, -1).expand(ppf, pph, ppw, -1))
    ang = torch.cat(parts, dim=-1).reshape(ppf * pph * ppw, hd // 2)
    return torch.stack([ang.cos(), ang.sin()], dim=-1).to(f32).contiguous().to(device)


def merge_lora_into_state_dict(sd: Dict[str, torch.Tensor], lora_sd: Dict[str, torch.Tensor], alpha: float, r: int) -> int:
    """Fold a peft LoRA adapter (`…lora_A.weight` [r,in], `…lora_B.weight` [out,r]) into the base weights:
    W += (alpha/r)·B·A.  The reference keeps the adapter unmerged (inference_t23d.py:74-77); merging is the same map
    in exact arithmetic and removes 16 skinny GEMMs per block.  Returns the number of merged matrices."""
    n = 0
    for key, A in lora_sd.items():
        if "lora_A" not in key:
            continue
        kb = key.replace("lora_A", "lora_B")
        base = key.split("lora_A")[0].rstrip(".")
        for pref in ("base_model.model.", "transformer."):
            if base.startswith(pref):
                base = base[len(pref):]
        wname = base + ".weight"
        if wname not in sd or kb not in lora_sd:
            raise KeyError(f"LoRA target {wname} not found in base state dict")
        # kept in fp32: WanDiT._load rounds to bf16 exactly once (rounding to a bf16/fp16 checkpoint dtype first would quantise the
        # rank-r delta at the ulp of W twice)
        sd[wname] = sd[wname].float() + (alpha / r) * (lora_sd[kb].float() @ A.float())
        n += 1
    return n


class _Workspace:
    """Activation buffers for B batch items of Nl LOCAL tokens attending to N keys (Nl == N unless sequence-parallel)."""

    def __init__(self, B: int, Nl: int, N: int, P: int, cfg: WanDiTConfig, device):
        d, M = cfg.dim, B * Nl
        e = lambda *s, dt=bf16: torch.empty(*s, device=device, dtype=dt)
        self.B, self.N, self.M = B, Nl, M
        self.a8 = self.sa = self.h8 = self.sh = None   # e4m3 activations + per-token scales, allocated on first use of the fp8 GEMM mode
        self.x = e(M, d)
        self.n = e(M, d)
        self.q2 = e(M, d)
        self.p2 = self.q2sq = None   # cached-context cross-attention: probabilities [M, heads * padded keys] and the row statistics of q
        self.ao = e(M, d)
        self.h = e(M, cfg.ffn_dim)
        self.tok = e(M, cfg.in_channels * math.prod(cfg.patch_size))
        self.out = e(M, cfg.out_channels * math.prod(cfg.patch_size))
        if P == 1:
            self.qk = e(M, 2 * d)
            self.vt = torch.zeros(d, B * ((N + 63) // 64 * 64), device=device, dtype=bf16)
            self.qk8 = self.vt8 = None   # e4m3 copies, allocated on first use of the fp8 attention mode
        else:  # sequence-parallel: local q, packed local [K | V^T] to send, gathered slabs, full K / V^T
            self.q = e(M, d)
            self.pack = e(2 * M * d)
            self.gbuf = e(P, 2 * M * d)
            self.q8 = self.pack8 = self.gbuf8 = None   # e4m3 forms, allocated on first use of the fp8 attention mode
            self.kfull = e(B * N, d)
            self.vt = torch.zeros(d, B * N + 64, device=device, dtype=bf16)
            self.ogather = e(P, M * self.out.shape[1])


class WanDiT:
    """Drop-in for diffusers' WanTransformer3DModel on the inference path (forward only, bf16 compute)."""

    def __init__(self, cfg: WanDiTConfig, state_dict: Dict[str, torch.Tensor], device="cuda"):
        L.load()  # fail loudly before touching weights if the HIP library is missing
        self.cfg = cfg
        self.device = torch.device(device)
        self.dtype = bf16
        self._ws: Dict[tuple, _Workspace] = {}
        self._rope: Dict[tuple, torch.Tensor] = {}
        # (B, L, thread) -> (prompt key, persistent cross-attention K / V^T buffers), least recently used first.  A slot is a prompt context's
        # BUFFERS (per block: K | V rows, V^T, and with ctx_vo the [B, d, H * 128] V.Wo^T operand - 283 MB per slot at Wan-1.3B, B = 2;
        # 4.2 GB at Wan-14B), so the cache is bounded: at most `max_ctx_slots` per host thread, and the slots of threads that no longer
        # exist (seqpar.ThreadWorld ranks of a finished call) are dropped.  An evicted slot's hipGraphs die with it (GraphedWanDiT keys
        # its graphs by the slot's serial number).
        self._ctx: "OrderedDict[tuple, tuple]" = OrderedDict()
        self.max_ctx_slots = 2
        self._ctx_serial = 0
        self.sp_split_k = os.environ.get("V3A_SP_SPLIT_K", "1") != "0"   # sequence-parallel shards: FFN2 as split-K partials + finish (see _sp_ksplit)
        self._ctx_lock = threading.Lock()   # virtual ranks (threads) look up / evict concurrently
        self.merge_padding_keys = True      # see _context
        # "fp8": self-attention on the block-scaled fp8 MFMA (BASELINE config #4; csrc/attention_fp8.hip): q / k (after RMSNorm + RoPE)
        # and V^T are rounded to e4m3 with the unit scales below.  "bf16" (default) is the reference's precision.
        self.attn_dtype = "bf16"
        self.fp8_scales = (1.0, 1.0, 1.0)
        # "fp8" (set by enable_fp8_gemm): the block projections of latent tokens (q/k/v/o, cross q/o, FFN) on e4m3 operands
        # (v3a_gemm_fp8_nt), activations quantised per token and weights per output channel (token-local: the sequence-parallel
        # path runs them on its shard unchanged).
        self.gemm_dtype = "bf16"
        # sequence-parallel self-attention: a rank has N / P query rows but every workgroup still walks all N keys, so the launch has
        # too few workgroups to fill the chip or hide their latency.  None = divide the keys among enough workgroups (kv_split of
        # v3a_attention_fwd_bf16; deterministic, within bf16 rounding of the unsharded forward) and split the K of its FFN2 GEMM;
        # 1 = never (bit-identical to the unsharded forward).
        self.sp_kv_split: Optional[int] = None
        # Cross-attention in its cached-context form  attn2(x) = sum_h softmax_h(q K^T) (V_h Wo_h^T) + bo  (csrc/xattn_probs.hip): K, V
        # depend on the prompt only, so (V_h Wo_h^T) is built once per prompt in `_context`; per step one probabilities kernel and ONE GEMM
        # with K = heads * padded keys (1152 at Wan-1.3B) replace the flash kernel's P.V MFMAs and the K = d to_out projection.  Taken when
        # head_dim = 128, the (merged) key count is <= 128 and the block GEMMs run in bf16; False = flash attention + to_out GEMM.
        self.ctx_vo = True
        # q | k | V^T of a block's self-attention from ONE GEMM launch (transposed tail, csrc/gemm_bf16.hip tile 13) instead of a q | k and a V^T
        # launch: bit-identical.  OFF by default: measured 114.8 against 119.1 us in isolation but 118.4 against 117.5 us inside the model and
        # 27.03-27.20 against 27.02-27.11 ms on the step (every ROUND of a launch pays the per-tile fixed cost: DESIGN.md section 9, round 5 (d))
        self.fused_qkv = False
        self._load(state_dict)

    def _sp_ksplit(self, M: int, N: int, K: int) -> int:
        """split-K factor of a shard's long-K projection (FFN2): few output tiles, 140 K tiles each - unless the exact mode is on"""
        if self.sp_kv_split == 1 or M > 2048 or K < 4096 or not self.sp_split_k:
            return 1
        for S in (4, 2):
            if K % (64 * S) == 0 and (M // 64) * (N // 64) * S <= 2048:
                return S
        return 1

    def _sp_split(self, B: int, Nq: int, Nk: int) -> int:
        if self.sp_kv_split is not None:
            return max(1, int(self.sp_kv_split))
        wgs = B * self.cfg.num_attention_heads * ((Nq + 127) // 128)
        return max(1, min(-(-768 // wgs), (Nk // 64) // 8))   # ~3 workgroups per CU, at least 8 key tiles each

    _FP8_WEIGHTS = ("wqk", "wv", "wo", "wq2", "wo2", "w1", "w2")

    def enable_fp8_gemm(self) -> "WanDiT":
        """Quantise the block projection weights to e4m3 with one scale per output channel (kept beside the bf16 weights, which the
        per-prompt text K / V projections still use) and switch the block GEMMs to v3a_gemm_fp8_nt."""
        if self.cfg.dim % 128 or self.cfg.ffn_dim % 128:
            raise ValueError("fp8 GEMMs need K % 128 == 0")
        for b in self.blocks:
            for k in self._FP8_WEIGHTS:
                if k + "8" not in b:
                    b[k + "8"], b["s" + k] = ops.quantize_fp8_rows(b[k])
        self.gemm_dtype = "fp8"
        return self

    # ---------------------------------------------------------------- weights
    def _load(self, sd):
        cfg, dev = self.cfg, self.device
        d = cfg.dim
        W = lambda k: sd[k].to(device=dev, dtype=bf16).contiguous()
        Fv = lambda k: sd[k].to(device=dev, dtype=f32).contiguous()
        self.patch_w = sd["patch_embedding.weight"].reshape(d, -1).to(device=dev, dtype=bf16).contiguous()
        self.patch_b = Fv("patch_embedding.bias")
        ce = "condition_embedder."
        self.te1_w, self.te1_b = W(ce + "time_embedder.linear_1.weight"), Fv(ce + "time_embedder.linear_1.bias")
        self.te2_w, self.te2_b = W(ce + "time_embedder.linear_2.weight"), Fv(ce + "time_embedder.linear_2.bias")
        self.tp_w, self.tp_b = W(ce + "time_proj.weight"), Fv(ce + "time_proj.bias")
        self.tx1_w, self.tx1_b = W(ce + "text_embedder.linear_1.weight"), Fv(ce + "text_embedder.linear_1.bias")
        self.tx2_w, self.tx2_b = W(ce + "text_embedder.linear_2.weight"), Fv(ce + "text_embedder.linear_2.bias")
        self.blocks = []
        tables = []
        # cross-attention K | V projections of ALL blocks stacked [L, 2, d, d] (k rows then v rows per block): the per-prompt context is one
        # weight-streaming GEMM per batch item over the stack instead of two tiny launches per block; b["wk2"] / b["wv2"] are views of it
        nl = cfg.num_layers
        self.ctx_w = torch.empty(nl, 2, d, d, device=dev, dtype=bf16)
        self.ctx_b = torch.empty(nl, 2, d, device=dev, dtype=f32)
        for i in range(cfg.num_layers):
            p = f"blocks.{i}."
            b = {}
            # to_q | to_k | to_v stacked: ONE tensor, so that the fused projection (v3a_gemm_args.C_t: q | k row-major, V^T transposed, 768
            # tiles = three full rounds of one launch) and the separate q | k / V^T GEMMs of the sharded and e4m3 paths read the same memory
            b["wqkv"] = torch.cat([W(p + "attn1.to_q.weight"), W(p + "attn1.to_k.weight"), W(p + "attn1.to_v.weight")], 0).contiguous()
            b["bqkv"] = torch.cat([Fv(p + "attn1.to_q.bias"), Fv(p + "attn1.to_k.bias"), Fv(p + "attn1.to_v.bias")], 0).contiguous()
            b["wqk"], b["bqk"] = b["wqkv"][: 2 * d], b["bqkv"][: 2 * d]
            b["wv"], b["bv"] = b["wqkv"][2 * d:], b["bqkv"][2 * d:]
            b["wo"], b["bo"] = W(p + "attn1.to_out.0.weight"), Fv(p + "attn1.to_out.0.bias")
            b["nq"], b["nk"] = Fv(p + "attn1.norm_q.weight"), Fv(p + "attn1.norm_k.weight")
            b["n2w"], b["n2b"] = Fv(p + "norm2.weight"), Fv(p + "norm2.bias")
            b["wq2"], b["bq2"] = W(p + "attn2.to_q.weight"), Fv(p + "attn2.to_q.bias")
            self.ctx_w[i, 0].copy_(sd[p + "attn2.to_k.weight"]); self.ctx_b[i, 0].copy_(sd[p + "attn2.to_k.bias"])
            self.ctx_w[i, 1].copy_(sd[p + "attn2.to_v.weight"]); self.ctx_b[i, 1].copy_(sd[p + "attn2.to_v.bias"])
            b["wk2"], b["bk2"], b["wv2"], b["bv2"] = self.ctx_w[i, 0], self.ctx_b[i, 0], self.ctx_w[i, 1], self.ctx_b[i, 1]
            b["wo2"], b["bo2"] = W(p + "attn2.to_out.0.weight"), Fv(p + "attn2.to_out.0.bias")
            b["nq2"], b["nk2"] = Fv(p + "attn2.norm_q.weight"), Fv(p + "attn2.norm_k.weight")
            b["w1"], b["b1"] = W(p + "ffn.net.0.proj.weight"), Fv(p + "ffn.net.0.proj.bias")
            b["w2"], b["b2"] = W(p + "ffn.net.2.weight"), Fv(p + "ffn.net.2.bias")
            tables.append(sd[p + "scale_shift_table"].to(device=dev, dtype=f32).reshape(6, d))
            self.blocks.append(b)
        self.nq2_all = torch.stack([b["nq2"] for b in self.blocks], 0).contiguous()   # [L, d]: folded into the cached keys (ctx_vo)
        self.sst = torch.stack(tables, 0).contiguous()  # [L,6,d]
        self.out_sst = sd["scale_shift_table"].to(device=dev, dtype=f32).reshape(2, d).contiguous()
        self.po_w, self.po_b = W("proj_out.weight"), Fv("proj_out.bias")
        half = cfg.freq_dim // 2
        self._tfreq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=f32) / half).to(dev)

    # ---------------------------------------------------------------- context (per prompt)
    def _context(self, text: torch.Tensor):
        """text [B,L,4096] -> per-block cross-attention K [B*L,d] and V^T [d,B*Lp] (+ key bias).  The results live in buffers that
        persist per (B, L) (so a captured hipGraph of `forward` stays valid across prompts) and are recomputed only when `text`
        changes.

        Zero-padded prompts (the reference pads every prompt to 512 rows with zeros: wan_utils.py:52-59 / diffusers
        `_get_t5_prompt_embeds`): all-zero embedding rows give bit-identical keys and values, so the trailing run of `c` padding
        rows is represented by ONE key with an additive score bias log(c) — softmax over {k_1..k_n, c copies of k_pad} is
        softmax over {k_1..k_n, k_pad + log c} in exact arithmetic.  Cross-attention then runs over n+1 instead of 512 keys."""
        B, Lt, _ = text.shape
        slot = (B, Lt, threading.get_ident())  # per thread: virtual ranks (seqpar.ThreadWorld) hold different prompts
        # identity of the tensor OBJECT (kept alive by the cache entry, so its address cannot be recycled for another prompt's
        # embeddings while the entry is live) + its in-place version counter
        with self._ctx_lock:
            ent = self._ctx.get(slot)
            if ent is not None:
                self._ctx.move_to_end(slot)
        mode = (self.ctx_vo, self.gemm_dtype)   # the cached-context form stores keys with the query norm's weight folded in
        if ent is not None and ent[0][0] is text and ent[0][1] == text._version and ent[0][2] == mode:
            return ent[1]
        key = (text, text._version, mode)
        cfg = self.cfg
        d = cfg.dim
        Lp = (Lt + 63) // 64 * 64
        H, hd = cfg.num_attention_heads, cfg.attention_head_dim
        nl = len(self.blocks)
        if ent is None:
            with self._ctx_lock:
                self._evict_ctx_slots(slot[2])
                self._ctx_serial += 1
                serial = self._ctx_serial
            # K | V rows of every block in ONE buffer [B * Lt, L * 2d] (row = context token, columns = (block, k | v, channel)): what
            # the stacked projection writes; ks[l] is the strided [B * Lt, d] view of block l's keys
            kv = torch.zeros(B * Lt, nl * 2 * d, device=self.device, dtype=bf16)
            ks = [kv[:, 2 * d * l: 2 * d * l + d] for l in range(nl)]
            vts = [torch.zeros(d, B * Lp, device=self.device, dtype=bf16) for _ in self.blocks]
            kbias = torch.zeros(B, Lp, device=self.device, dtype=f32)
            vwo_store = None
        else:
            ks, vts, kbias, vwo_store, kv, serial = ent[1][0], ent[1][1], ent[1][4], ent[1][9], ent[1][10], ent[1][11]
        if vwo_store is None and self.ctx_vo and hd == 128 and self.gemm_dtype == "bf16":
            # [B, d, H * 128] per block: (V_h Wo_h^T) of the cached-context form, viewed [B, d, H * Lkp] for the prompt's key count
            vwo_store = [torch.empty(B * d * H * 128, device=self.device, dtype=bf16) for _ in self.blocks]
        # trailing all-zero rows (one host sync per prompt)
        nz = (text != 0).any(dim=-1)                                   # [B, Lt]
        last = torch.where(nz.any(dim=1), Lt - 1 - nz.flip(1).float().argmax(dim=1), torch.full((B,), -1, device=text.device))
        n_real = int(last.max().item()) + 1                            # rows [n_real, Lt) are zero for every batch item
        Lk = (n_real + 1 + 7) // 8 * 8                                  # real rows + explicit padding rows up to a multiple of 8
        merged = self.merge_padding_keys and Lk + 1 < Lt               # ... the last of which stands for all remaining ones
        if not merged:
            Lk = Lt                                                    # keys per batch item actually attended to
        kbias.zero_()
        if merged:
            kbias[:, Lk - 1] = math.log(Lt - (Lk - 1))
        t2 = text.reshape(B * Lt, -1).to(bf16).contiguous()
        c = ops.gemm(t2, self.tx1_w, self.tx1_b, act=L.ACT_GELU_TANH)
        c = ops.gemm(c, self.tx2_w, self.tx2_b)
        gran = max(16, 64 // math.gcd(H, 64))    # keys per head padded so that the GEMM's K = H * Lkp is a multiple of 64 (16 at 12 / 40 heads)
        Lkp = (Lk + gran - 1) // gran * gran
        use_vo = vwo_store is not None and self.ctx_vo and self.gemm_dtype == "bf16" and Lkp <= min(128, Lt)
        # K and V rows of ALL blocks for the first Lk rows of every batch item (merged: row Lk-1 represents all padding rows from there on):
        # one weight-streaming launch per batch item over the stacked [L * 2d, d] projection (120 two-launch skinny GEMMs before: 4 ms per prompt)
        wall, ball = self.ctx_w.view(nl * 2 * d, d), self.ctx_b.view(-1)
        for bi in range(B):
            ops.gemm(c[bi * Lt: bi * Lt + Lk], wall, ball, out=kv[bi * Lt: bi * Lt + Lk])
            if use_vo and Lkp > Lk:
                kv[bi * Lt + Lk: bi * Lt + Lkp].zero_()        # the padded key rows of V feed the V.Wo^T GEMM (their probabilities are exactly 0)
        for li, (b, k) in enumerate(zip(self.blocks, ks)):
            for bi in range(B):
                rows = slice(bi * Lt, bi * Lt + Lk)
                ops.rmsnorm_rope(k[rows], b["nk2"], out=k[rows], eps=cfg.eps)
        # cached-context cross-attention: VWo[b][n, h * Lkp + j] = sum_c Wo[n, h hd + c] V[j, h hd + c], one batched GEMM over the heads per
        # (block, batch item) straight from the V rows of the projection buffer; fp32 accumulation, rounded to bf16 once
        vwos = None
        if use_vo:
            Kp = H * Lkp
            vwos = [st[: B * d * Kp].view(B, d, Kp) for st in vwo_store]
            for li, (b, vwo) in enumerate(zip(self.blocks, vwos)):
                for bi in range(B):
                    vrows = kv[bi * Lt: bi * Lt + Lkp, 2 * d * li + d: 2 * d * li + d + hd]     # [Lkp, hd] of head 0; heads hd columns apart
                    ops.gemm(b["wo2"][:, :hd], vrows, out=vwo[bi][:, :Lkp], batch=(H, hd, hd, Lkp))
            # the query RMS norm's per-column weight moves onto the (already normalised) keys, all blocks at once; its per-row factor is
            # applied to the scores by the probabilities kernel from the to_q projection's row statistics: q is never normalised in memory
            k5 = kv.view(B, Lt, nl, 2, d)[:, :Lk, :, 0]
            k5.copy_((k5.float() * self.nq2_all).to(bf16))
        else:
            Lkp = 0
            for li, (b, vt) in enumerate(zip(self.blocks, vts)):   # flash form: V^T per batch item, each at its 64-padded column block
                for bi in range(B):
                    ops.gemm(b["wv2"], c[bi * Lt: bi * Lt + Lk], b["bv2"], out=vt[:, bi * Lp: bi * Lp + Lk], bias_row=True)
        val = (ks, vts, Lt, Lp, kbias, Lk, merged, vwos, Lkp, vwo_store, kv, serial)
        with self._ctx_lock:
            self._ctx[slot] = (key, val)
            self._ctx.move_to_end(slot)
        return val

    def _evict_ctx_slots(self, thread_id: int) -> None:
        """Make room for one new context slot of `thread_id`: drop the slots of host threads that no longer exist, then the least recently
        used slots of this thread until fewer than `max_ctx_slots` remain.  (Buffers still referenced by in-flight launches stay alive
        until the stream has passed them: the caching allocator frees stream-ordered.)"""
        alive = {t.ident for t in threading.enumerate()}
        for k in [k for k in self._ctx if k[2] not in alive]:
            del self._ctx[k]
        mine = [k for k in self._ctx if k[2] == thread_id]
        for k in mine[: max(0, len(mine) - (self.max_ctx_slots - 1))]:
            del self._ctx[k]

    def live_ctx_serials(self) -> set:
        with self._ctx_lock:
            return {e[1][11] for e in self._ctx.values()}

    # ---------------------------------------------------------------- forward
    @torch.no_grad()
    def forward(self, hidden_states: Optional[torch.Tensor], timestep, encoder_hidden_states: torch.Tensor,
                return_dict: bool = False, num_layers: Optional[int] = None, sp=None, tokens_in: bool = False, tokens_out: bool = False,
                latent_shape: Optional[tuple] = None, time_table: Optional[tuple] = None, hidden_in: Optional[torch.Tensor] = None,
                first_layer: int = 0, return_hidden: bool = False):
        """`sp` (wan/seqpar.py group) shards the latent tokens over sp.world ranks: every rank passes the SAME full
        `hidden_states` and gets the full prediction back; only N/P token rows are computed locally.
        Fused denoise loop (wan/pipeline.py): `tokens_in` = the patchified input already sits in `token_buffers()[0]` (written by
        ops.unipc_cfg_step; `hidden_states` may be None, `latent_shape` gives [B, C, T, H, W]); `tokens_out` = return the raw output
        tokens (`token_buffers()[1]`) instead of the un-patchified tensor; `time_table` = this step's (temb, mod) from `time_tables()`.
        Teacher forcing (tests): `hidden_in` [B, N, d] replaces the residual stream in front of block `first_layer`, blocks
        first_layer .. num_layers - 1 run, and `return_hidden` returns the residual stream behind the last of them ([B, N, d] bf16 copy)
        instead of the prediction (single GPU only)."""
        cfg = self.cfg
        B, C, Fr, Hh, Ww = hidden_states.shape if hidden_states is not None else latent_shape
        pt, ph, pw = cfg.patch_size
        ppf, pph, ppw = Fr // pt, Hh // ph, Ww // pw
        N, d, H, hd = ppf * pph * ppw, cfg.dim, cfg.num_attention_heads, cfg.attention_head_dim
        P, rk = (1, 0) if sp is None else (sp.world, sp.rank)
        if N % P or (P > 1 and (N // P) % 8):
            raise ValueError(f"{N} tokens do not split into {P} shards of a multiple of 8 rows")
        Nl = N // P  # local tokens [rk*Nl, (rk+1)*Nl) of every batch item
        wkey = (B, N, P, rk, threading.get_ident())  # per thread: virtual ranks (seqpar.ThreadWorld) must not share buffers
        ws = self._ws.get(wkey)
        if ws is None:
            ws = self._ws[wkey] = _Workspace(B, Nl, N, P, cfg, self.device)
        rope = self._rope.get((ppf, pph, ppw))
        if rope is None:
            rope = self._rope[(ppf, pph, ppw)] = rope_table(cfg, ppf, pph, ppw, self.device)
        rope = rope[rk * Nl:(rk + 1) * Nl]
        ks, vts, Lt, Lp, kbias, Lk, merged, vwos, Lkp = self._context(encoder_hidden_states)[:9]
        kbs = Lt * ks[0].stride(0)      # keys of a batch item: Lt rows of the stacked projection buffer

        # patchify: Conv3d(k=s=(1,2,2)) == GEMM over (c,pt,ph,pw)-major patches
        if not tokens_in:
            x5 = hidden_states.to(bf16).view(B, C, ppf, pt, pph, ph, ppw, pw).permute(0, 2, 4, 6, 1, 3, 5, 7)
            if P == 1:
                ws.tok.view(B, ppf, pph, ppw, C, pt, ph, pw).copy_(x5)
            else:
                ws.tok.view(B, Nl, -1).copy_(x5.reshape(B, N, -1)[:, rk * Nl:(rk + 1) * Nl])
        elif P != 1:
            raise ValueError("tokens_in is the single-GPU fused loop's path")
        x = ops.gemm(ws.tok, self.patch_w, self.patch_b, out=ws.x)

        # time conditioning (M = B rows: latency-only work)
        if time_table is not None:
            temb, mod = time_table
        else:
            temb, mod = self._time_conditioning(timestep.to(device=self.device, dtype=f32), B)

        nl = cfg.num_layers if num_layers is None else num_layers
        Ml = B * Nl
        if hidden_in is not None or return_hidden or first_layer:
            if P != 1:
                raise ValueError("hidden_in / first_layer / return_hidden: single GPU only")
            if hidden_in is not None:
                x.copy_(hidden_in.reshape(x.shape))
        if P == 1:
            q, k = ws.qk[:, :d], ws.qk[:, d:]
            vbs = ws.vt.shape[1] // B
        else:
            kl, vtl = ws.pack[:Ml * d].view(Ml, d), ws.pack[Ml * d:].view(d, Ml)
            vt_full = ws.vt[:, :B * N]
        g8 = self.gemm_dtype == "fp8"
        if g8:
            if ws.a8 is None:
                ws.a8, ws.sa = torch.empty(Ml, d, device=self.device, dtype=torch.uint8), torch.empty(Ml, device=self.device, dtype=f32)
                ws.h8, ws.sh = torch.empty(Ml, cfg.ffn_dim, device=self.device, dtype=torch.uint8), torch.empty(Ml, device=self.device, dtype=f32)

        def lin(a, b, w, bias, **kw):
            """a @ b[w]^T (+ epilogue): bf16, or e4m3 after a per-token quantisation pass over `a`"""
            if not g8:
                return ops.gemm(a, b[w], bias, **kw)
            a8, sa = (ws.h8, ws.sh) if a is ws.h else (ws.a8, ws.sa)
            if a is not lin.last:   # the q/k and V^T projections share one quantised copy of the normed tokens
                ops.quantize_fp8_rows(a, a8, sa)
                lin.last = a
            return ops.gemm(a8, b[w + "8"], bias, a_scale=sa, w_scale=b["s" + w], **kw)
        lin.last = None

        def norm(**kw):
            """ws.n = LN(x) (bf16), or in the fp8 mode its per-token e4m3 image straight from the LayerNorm kernel"""
            if g8:
                ops.layernorm(x, out=ws.a8, fp8_scale=ws.sa, eps=cfg.eps, **kw)
                lin.last = ws.n
            else:
                ops.layernorm(x, out=ws.n, eps=cfg.eps, **kw)

        for li in range(first_layer, nl):
            b, m = self.blocks[li], mod[li]
            # --- self attention
            norm(scale=m[:, 1], shift=m[:, 0], rows_per_batch=Nl)
            if P == 1:
                fused_qkv = self.fused_qkv and not g8 and vbs == N and (2 * d) % 192 == 0 and Ml % 8 == 0
                if fused_qkv:
                    ops.gemm(ws.n, b["wqkv"], b["bqkv"], out=ws.qk, t_out=ws.vt, t_col0=2 * d)
                else:
                    lin(ws.n, b, "wqk", b["bqk"], out=ws.qk)
                if fused_qkv:
                    pass
                elif g8:   # V^T = Wv . X^T: the weight rows are the GEMM's A side
                    for bi in range(B if vbs != N else 1):
                        r0, r1 = (0, Ml) if vbs == N else (bi * N, (bi + 1) * N)
                        ops.gemm(b["wv8"], ws.a8[r0:r1], b["bv"], out=ws.vt if vbs == N else ws.vt[:, bi * vbs: bi * vbs + N], bias_row=True,
                                 a_scale=b["swv"], w_scale=ws.sa[r0:r1])
                elif vbs == N:  # no per-item padding: V^T of the whole batch is one [d, B*N] GEMM (256 tiles of 192x256 at 1.3B)
                    ops.gemm(b["wv"], ws.n, b["bv"], out=ws.vt, bias_row=True)
                else:
                    for bi in range(B):
                        ops.gemm(b["wv"], ws.n[bi * N:(bi + 1) * N], b["bv"], out=ws.vt[:, bi * vbs: bi * vbs + N], bias_row=True)
                if self.attn_dtype == "fp8":
                    if ws.qk8 is None:
                        ws.qk8 = torch.empty(ws.qk.shape, device=ws.qk.device, dtype=torch.uint8)
                        ws.vt8 = torch.zeros(ws.vt.shape, device=ws.vt.device, dtype=torch.uint8)
                    qs, ksc, vs = self.fp8_scales
                    if qs == ksc:   # RMSNorm + RoPE emit the e4m3 operands themselves (= bf16 result -> v3a_quantize_fp8, in one pass)
                        ops.rmsnorm_rope(ws.qk, b["nq"], out=ws.qk8, rope=rope, head_dim=hd, tokens_per_batch=N, eps=cfg.eps, weight2=b["nk"],
                                         fp8_scale=qs)
                    else:
                        ops.rmsnorm_rope(ws.qk, b["nq"], out=ws.qk, rope=rope, head_dim=hd, tokens_per_batch=N, eps=cfg.eps, weight2=b["nk"])
                        ops.quantize_fp8(q, qs, out=ws.qk8[:, :d])
                        ops.quantize_fp8(k, ksc, out=ws.qk8[:, d:])
                    ops.quantize_fp8(ws.vt, vs, out=ws.vt8)
                    ops.attention_fp8(ws.qk8[:, :d], ws.qk8[:, d:], ws.vt8, ws.ao, B=B, H=H, Nq=N, Nk=N, q_batch_stride=N * 2 * d,
                                      k_batch_stride=N * 2 * d, vt_batch_stride=vbs, o_batch_stride=N * d, q_scale=qs, k_scale=ksc, v_scale=vs)
                else:
                    ops.rmsnorm_rope(ws.qk, b["nq"], out=ws.qk, rope=rope, head_dim=hd, tokens_per_batch=N, eps=cfg.eps, weight2=b["nk"])
                    ops.attention(q, k, ws.vt, ws.ao, B=B, H=H, Nq=N, Nk=N, D=hd, q_batch_stride=N * 2 * d,
                                  k_batch_stride=N * 2 * d, vt_batch_stride=vbs, o_batch_stride=N * d)
            else:
                # K and V^T of the local tokens first, so their all-gather rides under the Q projection
                if g8:   # (norm() left the e4m3 tokens and their scales in ws.a8 / ws.sa)
                    ops.gemm(ws.a8, b["wqk8"][d:], b["bqk"][d:], out=kl, a_scale=ws.sa, w_scale=b["swqk"][d:])
                else:
                    ops.gemm(ws.n, b["wqk"][d:], b["bqk"][d:], out=kl)
                ops.rmsnorm_rope(kl, b["nk"], out=kl, rope=rope, head_dim=hd, tokens_per_batch=Nl, eps=cfg.eps)
                if g8:
                    ops.gemm(b["wv8"], ws.a8, b["bv"], out=vtl, bias_row=True, a_scale=b["swv"], w_scale=ws.sa)
                else:
                    ops.gemm(b["wv"], ws.n, b["bv"], out=vtl, bias_row=True)
                a8m = self.attn_dtype == "fp8"
                if a8m:   # e4m3 K | V^T slabs: half the all-gather bytes, attention on the block-scaled MFMA over the gathered slabs
                    if Nl % 64:
                        raise NotImplementedError("fp8 attention over sequence-parallel shards needs 64 | tokens per rank")
                    if ws.pack8 is None:
                        u8 = lambda *sh: torch.empty(*sh, device=self.device, dtype=torch.uint8)
                        ws.q8, ws.pack8, ws.gbuf8 = u8(Ml, d), u8(2 * Ml * d), u8(P, 2 * Ml * d)
                    qs, ksc, vs = self.fp8_scales
                    ops.quantize_fp8(kl, ksc, out=ws.pack8[:Ml * d].view(Ml, d))
                    ops.quantize_fp8(vtl, vs, out=ws.pack8[Ml * d:].view(d, Ml))
                    pending = sp.all_gather(ws.gbuf8, ws.pack8)
                else:
                    pending = sp.all_gather(ws.gbuf, ws.pack)
                if g8:
                    ops.gemm(ws.a8, b["wqk8"][:d], b["bqk"][:d], out=ws.q, a_scale=ws.sa, w_scale=b["swqk"][:d])
                else:
                    ops.gemm(ws.n, b["wqk"][:d], b["bqk"][:d], out=ws.q)
                if a8m:
                    ops.rmsnorm_rope(ws.q, b["nq"], out=ws.q8, rope=rope, head_dim=hd, tokens_per_batch=Nl, eps=cfg.eps, fp8_scale=qs)
                else:
                    ops.rmsnorm_rope(ws.q, b["nq"], out=ws.q, rope=rope, head_dim=hd, tokens_per_batch=Nl, eps=cfg.eps)
                pending.wait()
                if a8m:
                    ops.attention_fp8(ws.q8, ws.gbuf8[0, :Ml * d].view(Ml, d), ws.gbuf8[0, Ml * d:].view(d, Ml), ws.ao, B=B, H=H, Nq=Nl, Nk=N,
                                      q_batch_stride=Nl * d, k_batch_stride=Nl * d, vt_batch_stride=Nl, o_batch_stride=Nl * d,
                                      q_scale=qs, k_scale=ksc, v_scale=vs, kv_seg=Nl, k_seg_stride=2 * Ml * d, vt_seg_stride=2 * Ml * d,
                                      kv_split=self._sp_split(B, Nl, N))
                elif Nl % 64 == 0:
                    # the flash kernel walks the gathered slabs in place: rank r's [K | V^T] pack is segment r (keys r*Nl .. (r+1)*Nl)
                    ops.attention(ws.q, ws.gbuf[0, :Ml * d].view(Ml, d), ws.gbuf[0, Ml * d:].view(d, Ml), ws.ao, B=B, H=H, Nq=Nl, Nk=N,
                                  D=hd, q_batch_stride=Nl * d, k_batch_stride=Nl * d, vt_batch_stride=Nl, o_batch_stride=Nl * d,
                                  kv_seg=Nl, k_seg_stride=2 * Ml * d, vt_seg_stride=2 * Ml * d, kv_split=self._sp_split(B, Nl, N))
                else:  # ragged shard: reassemble K [B*N, d] and V^T [d, B*N] (two copies per block)
                    ws.kfull.view(B, P, Nl, d).copy_(ws.gbuf[:, :Ml * d].view(P, B, Nl, d).permute(1, 0, 2, 3))
                    vt_full.view(d, B, P, Nl).copy_(ws.gbuf[:, Ml * d:].view(P, d, B, Nl).permute(1, 2, 0, 3))
                    ops.attention(ws.q, ws.kfull, ws.vt, ws.ao, B=B, H=H, Nq=Nl, Nk=N, D=hd, q_batch_stride=Nl * d,
                                  k_batch_stride=N * d, vt_batch_stride=N, o_batch_stride=Nl * d, kv_split=self._sp_split(B, Nl, N))
            lin.last = None
            lin(ws.ao, b, "wo", b["bo"], out=x, residual=x, scale=m[:, 2], rows_per_batch=Nl)
            # --- cross attention
            norm(weight=b["n2w"], bias=b["n2b"])
            if vwos is not None and not g8:
                # cached-context form: q stays the projection's raw output (its row statistics come out of the GEMM epilogue), the
                # probabilities kernel applies the RMS factor and writes P [B Nl, H Lkp], then ONE GEMM against the prompt's
                # (V_h Wo_h^T), one B operand per batch item
                Kp = H * Lkp
                if ws.p2 is None:
                    ws.p2 = torch.empty(Ml * H * 128, device=self.device, dtype=bf16)
                    ws.q2sq = torch.empty(Ml, d // 32, device=self.device, dtype=f32)
                p2 = ws.p2[: Ml * Kp].view(Ml, Kp)
                ops.gemm(ws.n, b["wq2"], b["bq2"], out=ws.q2, row_sumsq=ws.q2sq)
                ops.xattn_probs(ws.q2, ks[li], p2, B=B, H=H, Nq=Nl, Nk=Lk, Lkp=Lkp, q_batch_stride=Nl * d, k_batch_stride=kbs,
                                p_batch_stride=Nl * Kp, key_bias=kbias if merged else None, key_bias_first=Lk - 1,
                                q_row_sumsq=ws.q2sq, q_eps=cfg.eps)
                ops.gemm(p2[:Nl], vwos[li][0], b["bo2"], out=x[:Nl], residual=x[:Nl], batch=(B, Nl * Kp, d * Kp, Nl * d))
            else:
                lin(ws.n, b, "wq2", b["bq2"], out=ws.q2)
                ops.rmsnorm_rope(ws.q2, b["nq2"], out=ws.q2, eps=cfg.eps)
                ops.attention(ws.q2, ks[li], vts[li], ws.ao, B=B, H=H, Nq=Nl, Nk=Lk, D=hd, q_batch_stride=Nl * d,
                              k_batch_stride=kbs, vt_batch_stride=Lp, o_batch_stride=Nl * d, key_bias=kbias if merged else None,
                              key_bias_first=Lk - 1)
                lin.last = None
                lin(ws.ao, b, "wo2", b["bo2"], out=x, residual=x)
            # --- feed forward
            norm(scale=m[:, 4], shift=m[:, 3], rows_per_batch=Nl)
            lin(ws.n, b, "w1", b["b1"], out=ws.h, act=L.ACT_GELU_TANH)
            if P > 1 and not g8 and self._sp_ksplit(Ml, d, cfg.ffn_dim) > 1:
                ops.gemm(ws.h, b["w2"], b["b2"], out=x, residual=x, scale=m[:, 5], rows_per_batch=Nl, split_k=self._sp_ksplit(Ml, d, cfg.ffn_dim))
            else:
                lin(ws.h, b, "w2", b["b2"], out=x, residual=x, scale=m[:, 5], rows_per_batch=Nl)

        if return_hidden:
            return x.view(B, Nl, d).clone()
        om = (self.out_sst[None] + temb.float()[:, None]).contiguous()  # [B,2,d]
        ops.layernorm(x, out=ws.n, scale=om[:, 1], shift=om[:, 0], rows_per_batch=Nl, eps=cfg.eps)
        ops.gemm(ws.n, self.po_w, self.po_b, out=ws.out)
        if tokens_out:
            if P != 1:
                raise ValueError("tokens_out is the single-GPU fused loop's path")
            return ws.out if return_dict else (ws.out,)
        if P == 1:
            tokens = ws.out
        else:
            sp.all_gather(ws.ogather, ws.out).wait()
            tokens = ws.ogather.view(P, B, Nl, -1).permute(1, 0, 2, 3)
        o = tokens.reshape(B, ppf, pph, ppw, pt, ph, pw, cfg.out_channels).permute(0, 7, 1, 4, 2, 5, 3, 6)
        o = o.reshape(B, cfg.out_channels, Fr, Hh, Ww)
        return o if return_dict else (o,)

    __call__ = forward

    def _time_conditioning(self, t: torch.Tensor, B: int):
        """timesteps [R] f32 (R = B per-item values, or the distinct steps of a whole schedule) -> (temb [R, d] bf16, mod): with R == B the
        per-block modulation table [L, B, 6, d] f32 of ONE step; rows are independent, so a schedule's R steps can go through the three
        skinny GEMMs as one batch (`time_tables`)."""
        d = self.cfg.dim
        te = t[:, None] * self._tfreq[None]
        te = torch.cat([te.cos(), te.sin()], -1).to(bf16)
        t1 = ops.gemm(te, self.te1_w, self.te1_b, act=L.ACT_SILU)
        temb = ops.gemm(t1, self.te2_w, self.te2_b)
        tproj = ops.gemm(torch.nn.functional.silu(temb.float()).to(bf16), self.tp_w, self.tp_b)  # [R,6d]
        mod = (self.sst[:, None] + tproj.float().view(1, t.shape[0], 6, d)).contiguous()  # [L,R,6,d] f32
        return temb, mod

    @torch.no_grad()
    def time_tables(self, timesteps: torch.Tensor, B: int):
        """The time conditioning of a whole schedule at once: it depends on the timestep only, so the fused denoise loop computes it
        for all S steps in ONE pass of the embedder (three skinny GEMMs over S rows) instead of S passes over B identical rows.
        -> list of S (temb [B, d], mod [L, B, 6, d]) pairs, bit-identical per step to `_time_conditioning(t.expand(B))` (asserted at the
        1.3B / 14B widths by tests/test_dit_gpu.py::test_time_tables_bit_identical_at_production_width).  The modulation of the whole
        schedule is ONE [L, S, 6, d] f32 tensor (55 MB at 1.3B x 50 steps); a step's table is a stride-0 view of it over the batch
        dimension - the LayerNorm / GEMM kernels take the per-batch row stride of the scale / shift / gate operands, so nothing is
        materialised per step or per batch item."""
        t = timesteps.to(device=self.device, dtype=f32)
        out = []
        for s0 in range(0, t.shape[0], 64):   # (<= 128 rows per skinny GEMM)
            temb, mod = self._time_conditioning(t[s0:s0 + 64], B)
            for i in range(temb.shape[0]):
                out.append((temb[i:i + 1].expand(B, -1), mod[:, i:i + 1].expand(-1, B, -1, -1)))
        return out

    def token_buffers(self, B: int, latent_shape: tuple):
        """(input tokens [B N, 64] bf16, output tokens [B N, 64] bf16) of the workspace `forward` uses for this shape on this thread."""
        cfg = self.cfg
        _, C, Fr, Hh, Ww = latent_shape
        pt, ph, pw = cfg.patch_size
        N = (Fr // pt) * (Hh // ph) * (Ww // pw)
        wkey = (B, N, 1, 0, threading.get_ident())
        ws = self._ws.get(wkey)
        if ws is None:
            ws = self._ws[wkey] = _Workspace(B, N, N, 1, cfg, self.device)
        return ws.tok, ws.out


class GraphedWanDiT:
    """hipGraph replay of `WanDiT.forward` (torch.cuda.CUDAGraph = hipGraph on ROCm): one denoise step is ~600 launches, a third of
    them latency-bound (time embedding, norms at M = B rows, patchify); replaying a captured graph removes the host launch
    path from the 50-step loop.  Inputs are copied into static buffers, the prompt context lives in WanDiT's persistent
    buffers, so ONE capture per latent shape serves every step of every prompt.  While a GemmProbe is active (bench.py's
    roofline leg brackets individual launches with events) the call runs eagerly.
    `capture_sp=True` also captures the SEQUENCE-PARALLEL forward (wan/seqpar.py) - its per-block all-gathers included - for groups that
    declare `graph_safe` (DistGroup: RCCL collectives enqueue on the capturing stream; the thread-backed ThreadGroup of the 1-GPU tests
    synchronises through the host and is never captured).  A rank's sharded forward is 17 launches of ~10 us per block
    (tools/sp_rank_time.py), exactly where the host launch path shows.  One graph per (shape, group, rank)."""

    def __init__(self, dit: WanDiT, capture_sp: bool = False):
        self.dit, self.cfg, self.device, self.dtype = dit, dit.cfg, dit.device, dit.dtype
        self.capture_sp = capture_sp
        self._graphs: Dict[tuple, tuple] = {}

    @torch.no_grad()
    def __call__(self, hidden_states, timestep, encoder_hidden_states, return_dict: bool = False, num_layers=None, sp=None):
        pr = ops._probe
        sp_ok = sp is None or (self.capture_sp and getattr(sp, "graph_safe", False) and not getattr(sp, "profile", False))
        if not sp_ok or num_layers is not None or (pr is not None and pr.active):
            return self.dit.forward(hidden_states, timestep, encoder_hidden_states, return_dict, num_layers, sp)
        text = encoder_hidden_states
        cx = self.dit._context(text)  # eager: refreshes the persistent K / V^T buffers when the prompt changed
        lk, serial = cx[5], cx[11]
        live = self.dit.live_ctx_serials()
        for k in [k for k in self._graphs if k[-1] not in live]:   # graphs over an evicted context slot replay freed buffers: drop them
            del self._graphs[k]
        d = self.dit   # the precision modes are baked into a capture: a flipped mode must not replay the old-precision graph
        key = (tuple(hidden_states.shape), tuple(text.shape), lk, threading.get_ident(), d.attn_dtype, d.gemm_dtype, tuple(d.fp8_scales),
               d.merge_padding_keys, d.ctx_vo, d.fused_qkv, None if sp is None else (id(sp), sp.world, sp.rank, d.sp_kv_split), serial)
        ent = self._graphs.get(key)
        if ent is None:
            sx = torch.empty(hidden_states.shape, device=self.device, dtype=bf16)
            st = torch.empty(timestep.shape, device=self.device, dtype=timestep.dtype)
            sx.copy_(hidden_states)
            st.copy_(timestep)
            self.dit.forward(sx, st, text, sp=sp)  # eager warm-up: workspaces, kernel attributes, rope tables (and the communicator)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            kw = {}
            if sp is not None:
                # torch.distributed's RCCL watchdog thread polls the events of the eager warm-up's collectives every ~100 ms.  Under the
                # default GLOBAL capture mode such a query from another thread while this thread captures aborts the process ("operation not
                # permitted on an event last recorded in a capturing stream": 2 of 12 runs on MI355X / torch 2.10).  THREAD-LOCAL capture makes
                # the other thread's query legal by definition (it concerns this thread's calls only): 0 of 12 on its own, no timing assumption
                # (a 0.3 s sleep that let the watchdog retire the warm-up's work first used to sit here as well; it is gone).
                kw = dict(capture_error_mode="thread_local")
            with torch.cuda.graph(g, **kw):
                out = self.dit.forward(sx, st, text, sp=sp)[0]
            ent = self._graphs[key] = (g, sx, st, out, sp)   # (the group stays referenced: its id is part of the key)
        g, sx, st, out = ent[:4]
        sx.copy_(hidden_states)
        st.copy_(timestep)
        g.replay()
        return out if return_dict else (out,)

    forward = __call__
