// Non-causal flash attention forward for gfx950, bf16 in / fp32 softmax+accumulate / bf16 out.
// Replaces F.scaled_dot_product_attention at:
//   DiT   : diffusers==0.33.1 WanAttnProcessor2_0 (self-attn N=4096/6144 hd=128, cross-attn Nk=512)
//   recon : /root/reference/third_party_model/anysplat/src/model/encoder/vggt/layers/attention.py:64-69
//           (hd=64; frame attention 1029 keys, global attention 13377/21609 keys)
//   UMT5  : transformers T5/UMT5 self-attention with additive relative-position bias (RELB variant, hd=64, <=512 tokens)
//   VAE   : /root/reference/utils/wan_utils.py:460 (single head, 4096 tokens, C=384 -> handled as 3x128? no:
//           the VAE mid-block attention uses the GEMM path)
//
// Data layout (chosen for the MFMA operand shapes, not inherited from the reference):
//   Q, K : [batch][token][head*D + d]  (row stride ldq/ldk elements)           d contiguous
//   V^T  : [head*D + d][batch*ldvt_batch + key]  (row stride ldvt elements)     KEYS contiguous
//          produced directly by the projection GEMM in its V^T = Wv.X^T form, zero padded to a
//          multiple of 64 keys.
//   O    : [batch][token][head*D + d]
// One kernel template, two staging schedules of the same arithmetic (template flag V1, see attn_fwd_kernel): K and V^T
// double-buffered with two workgroups per CU (hd = 64, relative-bias variant), or V^T single-buffered with THREE workgroups per CU
// (hd = 128: 706 -> 807 TFLOP/s on the DiT self-attention launch, bit-identical output).
// Per workgroup: NW waves x 32 queries.  Per 64-key tile each wave computes
//   S^T[key][q]  = K_tile . Q^T        (A = K rows from LDS, B = Q fragments held in registers)
//   O^T[d][q]   += V^T_tile . P^T      (A = V^T rows from LDS, B = P straight from the S^T registers)
// The row->key assignment of the S^T tile is permuted (key = (i&3) + 4*(i>>3) + 16*((i>>2)&1)) so that
// every lane ends up holding 16 CONSECUTIVE keys of one query: P feeds the second MFMA as-is (no
// cross-lane traffic, no LDS round trip) and V^T is read with plain 16-byte ds_read_b128.
// K / V^T tiles arrive by 16-byte LDS-DMA into a 2-deep ring; bank swizzles are applied on the DMA
// source address and mirrored on the fragment reads.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "../../include/vist3a_hip.h"

#ifndef V3A_ATTN_PRIO
#define V3A_ATTN_PRIO 3   // bit 0: s_setprio around the S phase, bit 1: around the PV phase (MFMAs + interleaved softmax).  One box: 194.7 us without, 193.8 S only,
                          // 189.4 PV only, 191.2 both at level 1; another: 193.4 PV only, 191.5 with S at level 1 and PV at level 2 (the default), 195.9 the other way round
#endif
#ifndef V3A_ATTN_PRIO_S_LEVEL
#define V3A_ATTN_PRIO_S_LEVEL 1
#endif
#ifndef V3A_ATTN_PRIO_PV_LEVEL
#define V3A_ATTN_PRIO_PV_LEVEL 2
#endif
#ifndef V3A_ATTN_PF
#define V3A_ATTN_PF 3   // LDS fragment reads ahead of their MFMA (plain kernel): 189.6 / 186.4 / 184.9 / 192.6 us for 1 / 2 / 3 / 4 (4 spills)
#endif

namespace {

struct AttnP {
  const char* q; const char* k; const char* vt; char* o;
  long q_bs, k_bs, vt_bs, o_bs;   // batch strides (elements)
  int ldq, ldk, ldvt, ldo;        // row strides (elements)
  int H, Nq, Nk;
  float scale_log2e;              // softmax scale * log2(e)
  int kv_period, kv_valid;        // kv_period > 0: key k takes part only if (k % kv_period) < kv_valid
  const float* relb;              // RELB: additive bias by relative position, [H][relb_stride], index key - query + relb_center
  int relb_stride, relb_center;
  float inv_scale;                // bias is added to the raw score as bias / softmax_scale
  const float* kbias;             // KBIAS: additive per-key bias [B][kbias_stride] (log-multiplicity of merged identical keys)
  int kbias_stride, kbias_first;  // keys below kbias_first have zero bias: tiles entirely below it skip the loads
  int kv_seg;                     // > 0: keys live in segments of kv_seg (a multiple of 64) keys, one per all-gathered rank slab:
  long k_seg, vt_seg;             //      key kk of a batch item is row (kk % kv_seg) of segment kk / kv_seg, segments k_seg / vt_seg elements apart
  int kv_split, B;                // > 1: the key tiles are divided among kv_split workgroups per query block, each writing an
  float* ws_o; float* ws_ml;      //      UNNORMALISED fp32 partial O [S][B][Nq][H*D] and its (reference, sum) [S][B][Nq][H][2]
  float out_mul;                  // combine: factor on the merged result (1 for bf16; 256 * v_scale for the e4m3 kernel's units)
};

// One source, two staging schedules of the same arithmetic (V1):
//   V1 = false: K and V^T tiles double-buffered (64 KB at hd = 128), two workgroups per CU - hd = 64 and the relative-bias variant;
//   V1 = true : K double-, V^T SINGLE-buffered (48 KB) and fetched under S and the softmax of its own tile, THREE workgroups per CU -
//               the DiT's 98304 query rows are exactly 12 waves of 32 per CU (3 x 4 waves is one full round where 2 x 4 leaves a
//               half-empty second one); one extra barrier per tile orders the V^T landing before the PV MFMAs.
template <int D, int NW, bool RELB, bool KBIAS, bool V1>
__global__ __launch_bounds__(NW * 64, V1 ? 3 : 2) void attn_fwd_kernel(const AttnP p) {
  constexpr int KSTRIDE = V1 ? 64 * D * 2 : 64 * D * 2 + D * 128;   // distance between the two K buffers
  constexpr int KV = 64;                 // keys per tile
  constexpr int KROWB = D * 2;           // bytes per K row in LDS
  constexpr int KCPR = KROWB / 16;       // 16-B chunks per K row (16 or 8)
  constexpr int KRPI = 64 / KCPR;        // K rows per DMA instruction (4 or 8)
  constexpr int KTILE = KV * KROWB;      // bytes
  constexpr int VTILE = D * 128;         // D rows x 64 keys x 2 B
  constexpr int STAGE = KTILE + VTILE;
  constexpr int KINS = KTILE / 1024 / NW;  // DMA instructions per wave per tile (K)
  constexpr int VINS = VTILE / 1024 / NW;
  constexpr int KS = D / 16;             // k-steps of QK^T
  constexpr int DT = D / 32;             // 32-row d tiles of O^T
  constexpr int OPITCH = D * 2 + 8;
  static_assert(KTILE % (1024 * NW) == 0 && VTILE % (1024 * NW) == 0, "tile/wave split");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;

  // block -> (batch, head, query block); consecutive blocks of one (batch, head) share K/V in L2:
  // hardware places block b on XCD b%8, so make the q-block index vary slowest across XCD lanes.
  const int S = p.kv_split > 1 ? p.kv_split : 1;
  const int nqb = (p.Nq + NW * 32 - 1) / (NW * 32);
  const int bid0 = xcd_remap(blockIdx.x, gridDim.x);
  const int split = bid0 % S, bid = bid0 / S;
  const int bh = bid / nqb, qb = bid % nqb;
  const int b = bh / p.H, h = bh % p.H;
  const int q0 = qb * (NW * 32) + wave * 32;

  const char* Qb = p.q + ((size_t)b * p.q_bs + (size_t)h * D) * 2;
  const char* Kb = p.k + ((size_t)b * p.k_bs + (size_t)h * D) * 2;
  const char* Vb = p.vt + ((size_t)h * D * p.ldvt + (size_t)b * p.vt_bs) * 2;

  // ---- Q fragments (B operand): lane (q = l31, hi) slot j <-> d = 16*ks + 8*hi + j ----
  bf16x8 qf[KS];
  {
    int qr = q0 + l31;
    qr = qr < p.Nq ? qr : p.Nq - 1;
    const char* qp = Qb + (size_t)qr * p.ldq * 2 + hi * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 32);
  }

  // ---- DMA sources ----
  const char* kp[KINS];
  const char* vp[VINS];
  int krow[KINS];
#pragma unroll
  for (int j = 0; j < KINS; ++j) {
    const int g = j * NW + wave;
    const int r = g * KRPI + lane / KCPR;
    const int c = lane % KCPR;
    const int f = (KCPR == 16) ? (r & 15) : ((r >> 1) & 7);
    krow[j] = r;
    kp[j] = Kb + (size_t)(c ^ f) * 16;  // + row*ldk*2 added per tile (row clamp depends on tile)
  }
#pragma unroll
  for (int j = 0; j < VINS; ++j) {
    const int g = j * NW + wave;
    const int r = g * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    vp[j] = Vb + ((size_t)r * p.ldvt + c * 8) * 2;
  }
  // with kv_seg (sequence-parallel: K / V^T are read straight from the all-gathered per-rank slabs, no reassembly copy) a 64-key
  // tile lies inside one segment: its wave-uniform base moves by (k_seg - kv_seg * ldk) / (vt_seg - kv_seg) per segment crossed
  auto stage_k = [&](int s, int kt) {
    char* sb = smem + s * KSTRIDE;
    const size_t so = p.kv_seg > 0 ? (size_t)((kt * KV) / p.kv_seg) * (size_t)(p.k_seg - (long)p.kv_seg * p.ldk) * 2 : 0;
#pragma unroll
    for (int j = 0; j < KINS; ++j) {
      int row = kt * KV + krow[j];
      row = row < p.Nk ? row : p.Nk - 1;
      glds16(kp[j] + (size_t)row * p.ldk * 2 + so, sb + (j * NW + wave) * 1024);
    }
  };
  auto stage_v = [&](int s, int kt) {
    char* sv = V1 ? smem + 2 * KTILE : smem + s * KSTRIDE + KTILE;
    const size_t so = p.kv_seg > 0 ? (size_t)((kt * KV) / p.kv_seg) * (size_t)(p.vt_seg - p.kv_seg) * 2 : 0;
#pragma unroll
    for (int j = 0; j < VINS; ++j) glds16(vp[j] + (size_t)kt * KV * 2 + so, sv + (j * NW + wave) * 1024);
  };

  // ---- fragment read offsets ----
  // K (A operand of QK^T): MFMA row i = l31 <-> key pi(i) inside the 32-key sub-tile
  const int pi = (l31 & 3) + 4 * (l31 >> 3) + 16 * ((l31 >> 2) & 1);
  int kfo[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int cc = 2 * ks + hi;
    const int f = (KCPR == 16) ? (pi & 15) : ((pi >> 1) & 7);  // (pi+32)&15 == pi&15, ((pi+32)>>1)&7 == (pi>>1)&7
    kfo[ks] = pi * KROWB + ((cc ^ f) << 4);
  }
  // V^T (A operand of PV): row d = l31 (+32*dt), chunk = 4*t32 + 2*hi + ks2
  int vfo[4];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) {
    const int t32 = c4 >> 1, ks2 = c4 & 1;
    const int cc = 4 * t32 + 2 * hi + ks2;
    vfo[c4] = l31 * 128 + ((cc ^ ((l31 >> 1) & 7)) << 4);
  }

  f32x16 oacc[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float c = p.scale_log2e;

  const int nkt = (p.Nk + KV - 1) / KV;
  const int kt0 = (int)((long)split * nkt / S), kt1 = (int)((long)(split + 1) * nkt / S);   // this workgroup's key tiles
  stage_k(0, kt0);
  if constexpr (!V1) stage_v(0, kt0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    if constexpr (V1) {
      stage_v(0, kt);                               // V^T of this tile: lands under S and the softmax
      if (kt + 1 < kt1) stage_k(cur ^ 1, kt + 1);   // K of the next tile
    } else if (kt + 1 < kt1) {
      stage_k(cur ^ 1, kt + 1);
      stage_v(cur ^ 1, kt + 1);
    }
    const char* sK = smem + cur * KSTRIDE;
    const char* sV = V1 ? smem + 2 * KTILE : sK + KTILE;

    // S^T = K . Q^T  (two 32-key sub-tiles)
    f32x16 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(sK + t * 32 * KROWB + kfo[ks]);
        s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[t], 0, 0, 0);
      }
    }
    // lane (q = l31, hi) now holds keys kt*64 + 32*t + 16*hi + r, r = 0..15
    if (KBIAS && (kt + 1) * KV > p.kbias_first) {  // per-key additive bias (cross-attention over a zero-padded prompt: the identical padding keys are
                            // merged into ONE key carrying log(count)); 16 consecutive entries per sub-tile and lane
      const float* kb = p.kbias + (size_t)b * p.kbias_stride + kt * KV + 16 * hi;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt * KV + 32 * t + 16 * hi + r < p.Nk) s[t][r] += kb[32 * t + r] * p.inv_scale;
    }
    if constexpr (RELB) {  // T5-style relative position bias (UMT5 text encoder): 16 consecutive table entries per sub-tile
      const float* tb = p.relb + (size_t)h * p.relb_stride + p.relb_center + (kt * KV + 16 * hi) - min(q0 + l31, p.Nq - 1);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * KV + 32 * t + 16 * hi + r;
          if (key < p.Nk) s[t][r] += tb[32 * t + r] * p.inv_scale;
        }
    }
    if (kt == nkt - 1 && (p.Nk & (KV - 1))) {
      const int kb = kt * KV + 16 * hi;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb + 32 * t + r >= p.Nk) s[t][r] = -1e30f;
    }
    // padded multi-frame token layout: mask the per-frame filler rows (not in the bias instantiations - v3a_attention_fwd_bf16 rejects the
    // combination: beside the bias loops this block sent hipcc into a 12 k-instruction, 3 KB-scratch body)
    if (!RELB && !KBIAS && p.kv_period > 0) {
      const int pos = (kt * KV) % p.kv_period;
      if (pos + KV > p.kv_valid) {  // some key of this tile is filler (always true for periods < 64)
        // the filler rows [kv_valid, kv_period) of each frame the tile meets are tile-relative keys [st, st + n), wave-uniform: one unsigned
        // compare per key and range (one range when kv_valid >= 64; round 3: the per-key modulo form cost the reconstruction's global
        // attention 6 %, this one 2 %)
        const unsigned n = (unsigned)(p.kv_period - p.kv_valid);
        for (int st = p.kv_valid - pos; st < KV; st += p.kv_period) {
          const int lo = st - 16 * hi;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if ((unsigned)(32 * t + r - lo) < n) s[t][r] = -1e30f;
        }
      }
    }
    float mx = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
    // max over the two lane halves: v_permlane32_swap instead of the ds_bpermute a shuffle compiles to (see attn_fwd_plain_kernel) - NOT in
    // the two bias instantiations: with the asm statement beside their bias loops hipcc 7.2 spills 3.3 KB per lane (the cross-attention
    // fallback launch went from 21 us to ~950 us; same fragility as the key-period mask below)
#ifndef V3A_ATTN_BPERM
    if constexpr (!RELB && !KBIAS) {
      float ma = mx, mb = mx;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ma), "+v"(mb));
      mx = fmaxf(ma, mb);
    } else
#endif
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // Deferred rescale: m_run is the REFERENCE of the exponentials, not necessarily the running maximum.  It moves (and l / O are
    // rescaled) only when some lane's maximum outgrew it by more than 2^DEFER; until then p = 2^((s - m_run) c) <= 2^DEFER, which
    // costs bf16 no precision, and the 64 accumulator multiplies per tile are skipped (on random scores: every tile but the first
    // few).  O / l at the end is invariant to the reference.
    constexpr float DEFER = 8.0f;
    const float m_new = fmaxf(m_run, mx);
    const bool rescale = __builtin_amdgcn_ballot_w64((m_new - m_run) * c > DEFER) != 0;
    float alpha = 1.0f;
    if (rescale) {
      alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
    }
    const float mc = m_run * c;
    float psum = 0.f;
    bf16x8 pf[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pv[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], c, -mc));   // an explicit FMA: left to contraction hipcc fused 31 of the 32 and
                                                                            // the hand-scheduled plain kernel all 32 - the two must agree bit for bit
        psum += pv[r];
      }
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        u32x4 pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = pack_bf16x2(pv[8 * ks2 + 2 * e], pv[8 * ks2 + 2 * e + 1]);
        pf[2 * t + ks2] = __builtin_bit_cast(bf16x8, pk);
      }
    }
    l_run = l_run * alpha + psum;
    if (rescale) {
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    }
    if constexpr (V1) {
      // V^T pieces were issued before the K pieces: leave the K prefetch in flight
      if (kt + 1 < kt1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KINS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave's V^T pieces have landed
    }
    // O^T += V^T . P^T
    // (s_setprio 1 around this loop, which pays in the plain kernel, measured -0.5 % here on the reconstruction's hd = 64 launch)
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
#pragma unroll
      for (int i = 0; i < DT; ++i) {
        const bf16x8 vf = *(const bf16x8*)(sV + i * 4096 + vfo[c4]);
        oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[c4], oacc[i], 0, 0, 0);
      }
    }
    // lgkmcnt(0) is essential: hipcc sinks the last PV MFMA (and the wait for its V^T fragment read) BELOW the barrier, so a
    // wave would pass it with an LDS read of the single V^T buffer still in flight while a faster wave's next-tile DMA
    // overwrites that buffer (seen as run-to-run differences in whole 32-row groups)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- finish: combine the two key halves of l, normalise, park O as [q][d] in LDS, store rows ----
  l_run += __shfl_xor(l_run, 32, 64);
  if (S > 1) {   // split keys: unnormalised partial O (fp32) + its reference and sum; attn_combine_kernel finishes the softmax
    const int qr = q0 + l31;
    if (qr < p.Nq) {
      const size_t row = ((size_t)split * p.B + b) * p.Nq + qr;
      float* po = p.ws_o + (row * p.H + h) * D;
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = oacc[i][g * 4 + e];
          *(f32x4*)(po + i * 32 + g * 8 + hi * 4) = v;
        }
      if (hi == 0) {
        float* pm = p.ws_ml + (row * p.H + h) * 2;
        pm[0] = m_run; pm[1] = l_run;
      }
    }
    return;
  }
  const float inv = 1.0f / l_run;
  char* reg = smem + wave * (32 * OPITCH);
#pragma unroll
  for (int i = 0; i < DT; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = i * 32 + g * 8 + hi * 4;
      u32x2 pk;
      pk[0] = pack_bf16x2(oacc[i][g * 4 + 0] * inv, oacc[i][g * 4 + 1] * inv);
      pk[1] = pack_bf16x2(oacc[i][g * 4 + 2] * inv, oacc[i][g * 4 + 3] * inv);
      *(u32x2*)(reg + l31 * OPITCH + d * 2) = pk;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  constexpr int CH = D / 8;  // 16-B chunks per output row
  char* Ob = p.o + ((size_t)b * p.o_bs + (size_t)h * D) * 2;
#pragma unroll
  for (int it = 0; it < 32 * CH / 64; ++it) {
    const int idx = it * 64 + lane;
    const int ql = idx / CH, ch = idx % CH;
    const u32x2 lo = *(const u32x2*)(reg + ql * OPITCH + ch * 16);
    const u32x2 hi2 = *(const u32x2*)(reg + ql * OPITCH + ch * 16 + 8);
    const int qr = q0 + ql;
    if (qr < p.Nq) {
      u32x4 v;
      v[0] = lo[0]; v[1] = lo[1]; v[2] = hi2[0]; v[3] = hi2[1];
      *(u32x4*)(Ob + ((size_t)qr * p.ldo) * 2 + ch * 16) = v;
    }
  }
}


// The DiT self-attention's own instantiation ("plain": hd = 128, Nk % 64 == 0, no key mask / bias / segments; K double-, V^T single-buffered,
// three workgroups per CU) - the same arithmetic as attn_fwd_kernel<128, NW, false, false, true> above with the loop scheduled by hand
// (round 3): wave-uniform DMA bases + one 32-bit lane offset, LDS fragment reads PF ahead of their MFMA, sub-tile
// 0's row maximum behind the MFMAs of sub-tile 1, P in four 16-key chunks exponentiated behind the PV MFMAs of the previous chunk,
// s_setprio 1 around the PV phase, cross-half maximum by v_permlane32_swap.  Everything else (ragged / masked / biased / segmented keys,
// hd = 64, split keys' general forms) stays on attn_fwd_kernel, whose loop hipcc schedules: pinning it there cost registers (Q fragments
// in scratch at 168 VGPRs, hd = 64 lost a wave per SIMD) and kept the bias-table loads from being hoisted (cross-attention 21.6 -> 26.5 us).
template <int D, int NW>
__global__ __launch_bounds__(NW * 64, 3) void attn_fwd_plain_kernel(const AttnP p) {
  constexpr bool V1 = true;
  static_assert(D == 128 && NW == 4, "the DiT self-attention form");
  constexpr int KSTRIDE = V1 ? 64 * D * 2 : 64 * D * 2 + D * 128;   // distance between the two K buffers
  constexpr int KV = 64;                 // keys per tile
  constexpr int KROWB = D * 2;           // bytes per K row in LDS
  constexpr int KCPR = KROWB / 16;       // 16-B chunks per K row (16 or 8)
  constexpr int KRPI = 64 / KCPR;        // K rows per DMA instruction (4 or 8)
  constexpr int KTILE = KV * KROWB;      // bytes
  constexpr int VTILE = D * 128;         // D rows x 64 keys x 2 B
  constexpr int STAGE = KTILE + VTILE;
  constexpr int KINS = KTILE / 1024 / NW;  // DMA instructions per wave per tile (K)
  constexpr int VINS = VTILE / 1024 / NW;
  constexpr int KS = D / 16;             // k-steps of QK^T
  constexpr int DT = D / 32;             // 32-row d tiles of O^T
  constexpr int OPITCH = D * 2 + 8;
  static_assert(KTILE % (1024 * NW) == 0 && VTILE % (1024 * NW) == 0, "tile/wave split");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  constexpr int PF = V3A_ATTN_PF;   // plain instantiation: LDS fragment reads in flight ahead of the MFMA that consumes them

  // block -> (batch, head, query block); consecutive blocks of one (batch, head) share K/V in L2:
  // hardware places block b on XCD b%8, so make the q-block index vary slowest across XCD lanes.
  const int S = p.kv_split > 1 ? p.kv_split : 1;
  const int nqb = (p.Nq + NW * 32 - 1) / (NW * 32);
  const int bid0 = xcd_remap(blockIdx.x, gridDim.x);
  const int split = bid0 % S, bid = bid0 / S;
  const int bh = bid / nqb, qb = bid % nqb;
  const int b = bh / p.H, h = bh % p.H;
  const int q0 = qb * (NW * 32) + wave * 32;

  const char* Qb = p.q + ((size_t)b * p.q_bs + (size_t)h * D) * 2;
  const char* Kb = p.k + ((size_t)b * p.k_bs + (size_t)h * D) * 2;
  const char* Vb = p.vt + ((size_t)h * D * p.ldvt + (size_t)b * p.vt_bs) * 2;

  // ---- Q fragments (B operand): lane (q = l31, hi) slot j <-> d = 16*ks + 8*hi + j ----
  bf16x8 qf[KS];
  {
    int qr = q0 + l31;
    qr = qr < p.Nq ? qr : p.Nq - 1;
    const char* qp = Qb + (size_t)qr * p.ldq * 2 + hi * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 32);
  }

  // ---- DMA sources ----
  // A DMA address = wave-uniform 64-bit base (SGPRs: batch / head slab + tile + piece) + ONE per-lane 32-bit byte offset (row inside the
  // piece, swizzled 16-byte chunk): one v_lshl_add_u64 per piece and two VGPRs per operand for the whole loop.  (attn_fwd_kernel keeps KINS +
  // VINS per-lane 64-bit pointers and rebuilds row / clamp / row x ldk per piece - it has ragged last tiles to clamp: 47 VALU instructions
  // per tile incl. eight quarter-rate v_mul_lo_u32.)
  static_assert((NW * KRPI) % 16 == 0, "the swizzle term must not depend on the piece");
  unsigned kvo, vvo;
  {
    const int r0 = wave * KRPI + lane / KCPR, c = lane % KCPR;
    kvo = (unsigned)((c ^ (r0 & 15)) * 16) + (unsigned)r0 * (unsigned)p.ldk * 2u;
    const int rv = wave * 8 + (lane >> 3);                                // V^T piece stride NW * 8 rows: (rv >> 1) & 7 is piece-invariant
    vvo = ((unsigned)rv * (unsigned)p.ldvt + (unsigned)(((lane & 7) ^ ((rv >> 1) & 7)) * 8)) * 2u;
  }
  auto stage_k = [&](int s, int kt) {
    char* sb = smem + s * KSTRIDE;
    const size_t base = (size_t)kt * ((size_t)KV * p.ldk * 2), piece = (size_t)(NW * KRPI) * p.ldk * 2;
#pragma unroll
    for (int j = 0; j < KINS; ++j) glds16(Kb + (base + j * piece) + (size_t)kvo, sb + (j * NW + wave) * 1024);
  };
  auto stage_v = [&](int s, int kt) {
    char* sv = smem + 2 * KTILE;
    const size_t base = (size_t)kt * KV * 2, piece = (size_t)(NW * 8) * p.ldvt * 2;
#pragma unroll
    for (int j = 0; j < VINS; ++j) glds16(Vb + (base + j * piece) + (size_t)vvo, sv + (j * NW + wave) * 1024);
  };

  // ---- fragment read offsets ----
  // K (A operand of QK^T): MFMA row i = l31 <-> key pi(i) inside the 32-key sub-tile
  const int pi = (l31 & 3) + 4 * (l31 >> 3) + 16 * ((l31 >> 2) & 1);
  int kfo[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int cc = 2 * ks + hi;
    const int f = (KCPR == 16) ? (pi & 15) : ((pi >> 1) & 7);  // (pi+32)&15 == pi&15, ((pi+32)>>1)&7 == (pi>>1)&7
    kfo[ks] = pi * KROWB + ((cc ^ f) << 4);
  }
  // V^T (A operand of PV): row d = l31 (+32*dt), chunk = 4*t32 + 2*hi + ks2
  int vfo[4];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) {
    const int t32 = c4 >> 1, ks2 = c4 & 1;
    const int cc = 4 * t32 + 2 * hi + ks2;
    vfo[c4] = l31 * 128 + ((cc ^ ((l31 >> 1) & 7)) << 4);
  }

  f32x16 oacc[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float c = p.scale_log2e;

  const int nkt = (p.Nk + KV - 1) / KV;
  const int kt0 = (int)((long)split * nkt / S), kt1 = (int)((long)(split + 1) * nkt / S);   // this workgroup's key tiles
  stage_k(0, kt0);
  if constexpr (!V1) stage_v(0, kt0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    if constexpr (V1) {
      stage_v(0, kt);                               // V^T of this tile: lands under S and the softmax
      if (kt + 1 < kt1) stage_k(cur ^ 1, kt + 1);   // K of the next tile
    } else if (kt + 1 < kt1) {
      stage_k(cur ^ 1, kt + 1);
      stage_v(cur ^ 1, kt + 1);
    }
    const char* sK = smem + cur * KSTRIDE;
    const char* sV = V1 ? smem + 2 * KTILE : sK + KTILE;

    // S^T = K . Q^T  (two 32-key sub-tiles)
    f32x16 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
    constexpr bool PLAIN = true;   // nothing touches S between the MFMAs and the row maximum
    float mx0 = 0.f;
    if constexpr (V3A_ATTN_PRIO & 1) __builtin_amdgcn_s_setprio(V3A_ATTN_PRIO_S_LEVEL);
    {   // sub-tile 0 first; its row maximum (8 v_max3) rides behind the MFMAs of sub-tile 1, one per MFMA
      constexpr int NF = 2 * KS;
      auto ldk = [&](int j) { return *(const bf16x8*)(sK + (j / KS) * 32 * KROWB + kfo[j % KS]); };
      bf16x8 kf[NF];
#pragma unroll
      for (int j = 0; j < PF; ++j) kf[j] = ldk(j);
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        if (j + PF < NF) kf[j + PF] = ldk(j + PF);
        s[j / KS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[j], qf[j % KS], s[j / KS], 0, 0, 0);
        if (j > KS) {   // (not behind the first MFMA of sub-tile 1: sub-tile 0's last MFMA has just been issued)
          static_assert(!PLAIN || KS == 8, "max schedule");
          constexpr int FIRST[8] = {0, 3, 6, 8, 10, 12, 14, 16};   // 16 values over the 7 remaining MFMAs
          const int e = j - KS - 1;
#pragma unroll
          for (int r = FIRST[e]; r < FIRST[e + 1]; ++r) mx0 = (r == 0) ? s[0][0] : fmaxf(mx0, s[0][r]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (V3A_ATTN_PRIO & 1) __builtin_amdgcn_s_setprio(0);
    // lane (q = l31, hi) now holds keys kt*64 + 32*t + 16*hi + r, r = 0..15; no bias, no mask in this instantiation
    float mx = mx0;
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
    {   // max over the two lane halves (the two 16-key halves of a query's row) without the LDS round trip of ds_bpermute in the middle of
        // the tile: v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second, so two copies of mx
        // become [lo, lo] and [hi, hi].  As an asm statement: through __builtin_amdgcn_permlane32_swap hipcc (ROCm 7.2) dropped the second
        // result - max(r0, r0) in the disassembly, i.e. a reference that ignores the upper half (tools/probe/permlane_probe.hip shows the
        // instruction itself is fine).  s_nop 1 on both sides: VALU write -> permlane read, and permlane write -> the consuming v_max
        // (an asm statement is invisible to hipcc's hazard recogniser, so the wait states are spelled out; cost: not measurable).
#ifdef V3A_ATTN_BPERM
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#else
      float ma = mx, mb = mx;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ma), "+v"(mb));
      mx = fmaxf(ma, mb);
#endif
    }
    // Deferred rescale: m_run is the REFERENCE of the exponentials, not necessarily the running maximum.  It moves (and l / O are
    // rescaled) only when some lane's maximum outgrew it by more than 2^DEFER; until then p = 2^((s - m_run) c) <= 2^DEFER, which
    // costs bf16 no precision, and the 64 accumulator multiplies per tile are skipped (on random scores: every tile but the first
    // few).  O / l at the end is invariant to the reference.
    constexpr float DEFER = 8.0f;
    const float m_new = fmaxf(m_run, mx);
    const bool rescale = __builtin_amdgcn_ballot_w64((m_new - m_run) * c > DEFER) != 0;
    float alpha = 1.0f;
    if (rescale) {
      alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
    }
    const float mc = m_run * c;
    float psum = 0.f;
    {
      // P in four 16-key chunks (= the four K-slices of the PV product), summed in key order: chunk c + 1 is exponentiated in the issue
      // slots behind the MFMAs of chunk c (a wave issues in order: VALU work right behind an MFMA runs in its shadow)
      u32x4 pw[4];
      auto pquarter = [&](int c4, int q) {   // keys 2q, 2q + 1 of chunk c4: 2 x (fma, exp, add) + one pack = 7 VALU instructions
        const int t = c4 >> 1, r0 = 8 * (c4 & 1) + 2 * q;
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r0], c, -mc)), p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r0 + 1], c, -mc));
        psum += p0;
        psum += p1;
        pw[c4][q] = pack_bf16x2(p0, p1);
      };
      auto pchunk = [&](int c4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) pquarter(c4, q);
      };
      if (rescale) {
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
      }
      pchunk(0);
      if constexpr (V1) {
        // V^T pieces were issued before the K pieces: leave the K prefetch in flight
        if (kt + 1 < kt1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KINS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave's V^T pieces have landed
      }
      // O^T += V^T . P^T
      if constexpr (V3A_ATTN_PRIO & 2) __builtin_amdgcn_s_setprio(V3A_ATTN_PRIO_PV_LEVEL);
      {
        constexpr int NF = 4 * DT;
        auto ldv = [&](int j) { return *(const bf16x8*)(sV + (j % DT) * 4096 + vfo[j / DT]); };
        bf16x8 vf[NF];
#pragma unroll
        for (int j = 0; j < PF; ++j) vf[j] = ldv(j);
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          if (j + PF < NF) vf[j + PF] = ldv(j + PF);
          oacc[j % DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[j], __builtin_bit_cast(bf16x8, pw[j / DT]), oacc[j % DT], 0, 0, 0);
          static_assert(!PLAIN || DT == 4, "one quarter chunk per MFMA");
          if (j / DT + 1 < 4) pquarter(j / DT + 1, j % DT);
          __builtin_amdgcn_sched_barrier(0);   // exactly this order: read ahead, MFMA, its share of the next chunk's softmax
        }
      }
      if constexpr (V3A_ATTN_PRIO & 2) __builtin_amdgcn_s_setprio(0);
      l_run = l_run * alpha + psum;
    }
    // lgkmcnt(0) is essential: hipcc sinks the last PV MFMA (and the wait for its V^T fragment read) BELOW the barrier, so a
    // wave would pass it with an LDS read of the single V^T buffer still in flight while a faster wave's next-tile DMA
    // overwrites that buffer (seen as run-to-run differences in whole 32-row groups)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- finish: combine the two key halves of l, normalise, park O as [q][d] in LDS, store rows ----
  l_run += __shfl_xor(l_run, 32, 64);
  if (S > 1) {   // split keys: unnormalised partial O (fp32) + its reference and sum; attn_combine_kernel finishes the softmax
    const int qr = q0 + l31;
    if (qr < p.Nq) {
      const size_t row = ((size_t)split * p.B + b) * p.Nq + qr;
      float* po = p.ws_o + (row * p.H + h) * D;
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = oacc[i][g * 4 + e];
          *(f32x4*)(po + i * 32 + g * 8 + hi * 4) = v;
        }
      if (hi == 0) {
        float* pm = p.ws_ml + (row * p.H + h) * 2;
        pm[0] = m_run; pm[1] = l_run;
      }
    }
    return;
  }
  const float inv = 1.0f / l_run;
  char* reg = smem + wave * (32 * OPITCH);
#pragma unroll
  for (int i = 0; i < DT; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = i * 32 + g * 8 + hi * 4;
      u32x2 pk;
      pk[0] = pack_bf16x2(oacc[i][g * 4 + 0] * inv, oacc[i][g * 4 + 1] * inv);
      pk[1] = pack_bf16x2(oacc[i][g * 4 + 2] * inv, oacc[i][g * 4 + 3] * inv);
      *(u32x2*)(reg + l31 * OPITCH + d * 2) = pk;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  constexpr int CH = D / 8;  // 16-B chunks per output row
  char* Ob = p.o + ((size_t)b * p.o_bs + (size_t)h * D) * 2;
#pragma unroll
  for (int it = 0; it < 32 * CH / 64; ++it) {
    const int idx = it * 64 + lane;
    const int ql = idx / CH, ch = idx % CH;
    const u32x2 lo = *(const u32x2*)(reg + ql * OPITCH + ch * 16);
    const u32x2 hi2 = *(const u32x2*)(reg + ql * OPITCH + ch * 16 + 8);
    const int qr = q0 + ql;
    if (qr < p.Nq) {
      u32x4 v;
      v[0] = lo[0]; v[1] = lo[1]; v[2] = hi2[0]; v[3] = hi2[1];
      *(u32x4*)(Ob + ((size_t)qr * p.ldo) * 2 + ch * 16) = v;
    }
  }
}


// Finish a key-split attention: one wave per (batch, query, head) merges the S partial softmaxes,
//   O = sum_s 2^((m_s - M) c) O_s / sum_s 2^((m_s - M) c) l_s,   M = max_s m_s,
// in a fixed order (deterministic), and writes the bf16 row.
template <int D>
__global__ __launch_bounds__(256) void attn_combine_kernel(const AttnP p) {
  const int lane = threadIdx.x & 63;
  const long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);   // (b, q, h)
  const long total = (long)p.B * p.Nq * p.H;
  if (idx >= total) return;
  const int h = (int)(idx % p.H);
  const long bq = idx / p.H;
  const int q = (int)(bq % p.Nq), b = (int)(bq / p.Nq);
  const int S = p.kv_split;
  const long sstride = (long)p.B * p.Nq * p.H;   // (b, q, h) entries per split
  float M = -3.0e38f;
  for (int s = 0; s < S; ++s) M = fmaxf(M, p.ws_ml[(s * sstride + idx) * 2]);
  constexpr int E = D / 64;
  float acc[E] = {};
  float L = 0.f;
  for (int s = 0; s < S; ++s) {
    const float* ml = p.ws_ml + (s * sstride + idx) * 2;
    const float w = __builtin_amdgcn_exp2f((ml[0] - M) * p.scale_log2e);
    L += w * ml[1];
    const float* po = p.ws_o + (s * sstride + idx) * D + lane * E;
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] += w * po[e];
  }
  const float inv = p.out_mul / L;
  char* o = p.o + ((size_t)b * p.o_bs + (size_t)q * p.ldo + (size_t)h * D + lane * E) * 2;
  if constexpr (E == 2) *(unsigned*)o = pack_bf16x2(acc[0] * inv, acc[1] * inv);
  else *(unsigned short*)o = (unsigned short)(pack_bf16x2(acc[0] * inv, 0.f) & 0xffffu);
}

template <int D, int NW, bool RELB, bool KBIAS, bool V1, bool PLAIN = false>
int launch_attn(const AttnP& p, int B, void* stream) {
  constexpr int KT = 64 * D * 2, VT = D * 128;
  constexpr int RING = V1 ? 2 * KT + VT : 2 * (KT + VT);
  constexpr int OBYTES = NW * 32 * (D * 2 + 8);
  constexpr int LDS = RING > OBYTES ? RING : OBYTES;
  static bool attr = false;
  void (*fn)(const AttnP) = attn_fwd_kernel<D, NW, RELB, KBIAS, V1>;
  if constexpr (PLAIN) {
    static_assert(!PLAIN || (D == 128 && !RELB && !KBIAS && V1), "plain = the DiT self-attention form");
    fn = attn_fwd_plain_kernel<D, NW>;
  }
  if (!attr) {
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return V3A_ERR_LAUNCH;
    attr = true;
  }
  const int nqb = (p.Nq + NW * 32 - 1) / (NW * 32);
  const int S = p.kv_split > 1 ? p.kv_split : 1;
  hipLaunchKernelGGL(fn, dim3((unsigned)(nqb * B * p.H * S)), dim3(NW * 64), LDS, (hipStream_t)stream, p);
  if (S > 1) {
    const long waves = (long)B * p.Nq * p.H;
    AttnP pc = p;
    pc.out_mul = 1.0f;
    hipLaunchKernelGGL(attn_combine_kernel<D>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, pc);
  }
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

}  // namespace

// shared with attention_fp8.hip
int v3a_attn_combine_launch(const float* ws_o, const float* ws_ml, void* o, long o_bs, int ldo, int B, int Nq, int H, int D, int S,
                            float scale_log2e, float out_mul, void* stream) {
  if (D != 128) return V3A_ERR_SHAPE;
  AttnP p = {};
  p.ws_o = const_cast<float*>(ws_o); p.ws_ml = const_cast<float*>(ws_ml); p.o = (char*)o; p.o_bs = o_bs; p.ldo = ldo;
  p.B = B; p.Nq = Nq; p.H = H; p.kv_split = S; p.scale_log2e = scale_log2e; p.out_mul = out_mul;
  const long waves = (long)B * Nq * H;
  hipLaunchKernelGGL(attn_combine_kernel<128>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

extern "C" size_t v3a_attention_split_workspace_bytes(int B, int H, int Nq, int D, int kv_split) {
  if (B <= 0 || H <= 0 || Nq <= 0 || D <= 0 || kv_split <= 1) return 0;
  return (size_t)kv_split * B * Nq * H * (D + 2) * sizeof(float);
}

extern "C" int v3a_attention_fwd_bf16(const v3a_attn_args* a, void* stream) {
  if (!a || !a->q || !a->k || !a->vt || !a->o) return V3A_ERR_ARG;
  if (a->B <= 0 || a->H <= 0 || a->Nq <= 0 || a->Nk <= 0) return V3A_ERR_SHAPE;
  if (a->D != 128 && a->D != 64) return V3A_ERR_SHAPE;
  if (a->ldq % 8 || a->ldk % 8 || a->ldvt % 8 || a->ldo % 8) return V3A_ERR_SHAPE;
  if (a->q_batch_stride % 8 || a->k_batch_stride % 8 || a->vt_batch_stride % 8 || a->o_batch_stride % 8)
    return V3A_ERR_SHAPE;
  // V^T rows must be readable (and finite, ideally zero) up to the next multiple of 64 keys
  if (!a->kv_seg && a->vt_batch_stride && a->vt_batch_stride < a->Nk && a->B > 1) return V3A_ERR_SHAPE;
  if (a->kv_seg > 0 && a->vt_batch_stride && a->vt_batch_stride < a->kv_seg && a->B > 1) return V3A_ERR_SHAPE;
  if (a->kv_period < 0 || (a->kv_period > 0 && (a->kv_valid <= 0 || a->kv_valid > a->kv_period))) return V3A_ERR_ARG;
  if (a->kv_period > 0 && (a->rel_bias || a->key_bias)) return V3A_ERR_ARG;   // the key-period mask is not compiled into the bias kernels
  AttnP p = {};   // every optional feature off unless set below (the relative-bias launch returns before most of them)
  p.q = (const char*)a->q; p.k = (const char*)a->k; p.vt = (const char*)a->vt; p.o = (char*)a->o;
  p.q_bs = a->q_batch_stride; p.k_bs = a->k_batch_stride; p.vt_bs = a->vt_batch_stride; p.o_bs = a->o_batch_stride;
  p.ldq = a->ldq; p.ldk = a->ldk; p.ldvt = a->ldvt; p.ldo = a->ldo;
  p.H = a->H; p.Nq = a->Nq; p.Nk = a->Nk;
  p.scale_log2e = a->scale * 1.4426950408889634f;
  p.kv_period = a->kv_period; p.kv_valid = a->kv_valid;
  p.relb = a->rel_bias; p.relb_stride = a->rel_bias_stride; p.relb_center = a->rel_bias_center;
  p.inv_scale = 1.0f / a->scale;
  p.B = a->B; p.kv_split = 1;
  if (a->rel_bias) {  // needs table entries for every (key - query) in [-(Nq-1), Nk-1]
    if (a->D != 64 || a->rel_bias_center < a->Nq - 1 || a->rel_bias_stride < a->rel_bias_center + a->Nk) return V3A_ERR_SHAPE;
    return launch_attn<64, 4, true, false, false>(p, a->B, stream);
  }
  p.kbias = a->key_bias; p.kbias_stride = a->key_bias_stride; p.kbias_first = a->key_bias_first > 0 ? a->key_bias_first : 0;
  p.kv_seg = a->kv_seg; p.k_seg = a->k_seg_stride; p.vt_seg = a->vt_seg_stride;
  if (a->kv_seg < 0 || (a->kv_seg > 0 && (a->kv_seg % 64 || a->Nk % a->kv_seg || a->D != 128 || a->rel_bias || a->key_bias ||
                                         a->k_seg_stride % 8 || a->vt_seg_stride % 8))) return V3A_ERR_SHAPE;
  p.B = a->B; p.kv_split = a->kv_split > 1 ? a->kv_split : 1; p.ws_o = nullptr; p.ws_ml = nullptr;
  if (a->kv_split < 0) return V3A_ERR_ARG;
  if (p.kv_split > 1) {   // workspace = [S][B][Nq][H][D] fp32 partial O, then [S][B][Nq][H][2] (reference, sum)
    if (a->D != 128 || a->rel_bias || !a->workspace || p.kv_split > (a->Nk + 63) / 64) return V3A_ERR_ARG;
    p.ws_o = (float*)a->workspace;
    p.ws_ml = p.ws_o + (size_t)p.kv_split * a->B * a->Nq * a->H * a->D;
  }
  if (a->key_bias) {
    if (a->D != 128 || a->rel_bias || a->key_bias_stride < a->Nk) return V3A_ERR_SHAPE;
    return launch_attn<128, 4, false, true, true>(p, a->B, stream);
  }
  if (a->D == 128) {
    // a sequence-parallel shard (Nq = N / P queries against all N keys) has too few 128-query blocks to occupy 256 CUs: 64-query
    // workgroups double the count; per-wave arithmetic and key order are unchanged, so the output stays bit-identical
    const long wgs4 = (long)a->B * a->H * ((a->Nq + 127) / 128) * p.kv_split;
    if (wgs4 < 256) return launch_attn<128, 2, false, false, true>(p, a->B, stream);
    if (a->Nk % 64 == 0 && !a->kv_period && !a->kv_seg) return launch_attn<128, 4, false, false, true, true>(p, a->B, stream);
    return launch_attn<128, 4, false, false, true>(p, a->B, stream);
  }
  return launch_attn<64, 4, false, false, false>(p, a->B, stream);  // recon (hd = 64): the 3-per-CU schedule measured no gain there
}
