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
#ifndef V3A_PW_ABL
#define V3A_PW_ABL 0     // experiment builds only (tools/abl_build.sh)
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
        // instruction itself is fine).  s_nop 1: VALU write -> permlane read.
#ifdef V3A_ATTN_BPERM
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#else
      float ma = mx, mb = mx;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
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

// ==============================================================================================================================
// EXPERIMENTAL (compiled only with -DV3A_ATTN_EXPERIMENTAL; tools/abl_build.sh builds it): round-3 study of the one-wave-per-SIMD,
// 64-queries-per-wave structure for the DiT self-attention.  NOT used by the product; kept because the measurements below decide
// what the next attempt has to look like ("round 3"):
//   attn_fwd64_kernel  (kernel 2): S -> softmax -> PV in sequence.  Bit-identical to the ROUND-2 production kernel, deterministic (the
//                      round-3 production loop computes all 32 exponent arguments of a tile as fused multiply-adds where hipcc had left one
//                      of them as a packed multiply + subtract: 88 of 25 M outputs of the DiT launch move by one bf16 ulp).
//                      B=2 H=16 N=4096 (two full rounds of 256-query workgroups): 270 us vs 280 us; the DiT's H=12 (1.5 rounds): 266 vs 203.
//   attn_fwd64p_kernel (kernel 3): software-pipelined (S^T double-buffered in registers, finish-softmax(t) beside the S MFMAs of t+1,
//                      start-softmax(t+1) beside the PV MFMAs of t).  251-258 us at H=16.  With the row-max exchange through
//                      ds_bpermute and MFMA fences at the phase ends it is bit-identical on small grids; at 512 workgroups the
//                      single-MFMA-slot form still shows run-to-run differences (an unresolved register hazard) - do not ship.
// What the stamps (s_memtime per phase) showed: a lone wave issues in order, ~5.4 cycles per instruction; VALU work overlaps an MFMA
// only when it sits directly behind it (four MFMAs followed by twenty VALU instructions cost 128 + 20 x 5 cycles); the softmax of 64
// queries is ~430 instructions per 64 MFMAs, so even the perfectly interleaved stream is issue-bound at ~45 cycles per MFMA.
// ==============================================================================================================================
#ifdef V3A_ATTN_EXPERIMENTAL
// ------------------------------------------------------------------------------------------------------------------------------
// Round 3: the DiT self-attention (hd = 128, no bias / mask) with ONE wave per SIMD and 64 queries per wave.
//   * a workgroup = 4 waves = 256 queries; every K / V^T tile in LDS serves 256 queries (half the LDS-DMA pieces per query of the
//     128-query kernel above) and every fragment read from LDS feeds two MFMAs (two 32-query blocks A and B per wave);
//   * register files are assigned by hand through inline-asm operand classes, because hipcc left to itself shuttles MFMA results
//     between the two files (measured on the QB = 2 instantiation of the kernel above: ~480 v_accvgpr moves per 64 MFMAs):
//       O^T accumulators (128 registers) and the Q fragments (64) live in the ACCUMULATOR file for the whole kernel,
//       S^T (64), P (32), the K / V^T fragments in flight and the softmax temporaries in the architectural VGPRs;
//   * MFMAs are asm statements, so hipcc neither pads their result hazards nor moves them: the `mfma_fence` statements below carry the
//     wait states an MFMA result needs before a VALU instruction may read it (8-pass XDL op: 12 states; two s_nop 15 are issued).
// Same arithmetic per query row as attn_fwd_kernel (same ascending-k MFMA chains, same per-32-row rescale decision): bit-identical.
__device__ __forceinline__ void mfma_s0(f32x16& d, const bf16x8& a, const bf16x8& b) {   // d = a . b          (S^T, first k step)
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_s(f32x16& d, const bf16x8& a, const bf16x8& b) {    // d += a . b         (S^T in VGPRs, Q in AGPRs)
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_o(f32x16& d, const bf16x8& a, const bf16x8& b) {    // d += a . b         (O^T in AGPRs)
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}
// LDS-DMA that hipcc does not see: behind the builtin it orders the next LDS read after the copy (s_waitcnt vmcnt(0) straight after
// the issue - with one wave per SIMD that puts the whole L2 latency of every tile on the critical path).  Here the copy is counted by
// the explicit s_waitcnt vmcnt at the end of the tile.  M0 = wave-uniform LDS byte address of the 1 KB piece (lane i lands at +16 i).
__device__ __forceinline__ void glds16_raw(const void* gsrc, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void mfma_fence_v(f32x16& a, f32x16& b) {   // MFMA results in VGPRs -> VALU readers
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void mfma_fence_a(f32x16& a, f32x16& b, f32x16& c, f32x16& d) {   // MFMA results in AGPRs -> v_accvgpr_read
  asm volatile("s_nop 15\n\ts_nop 15" : "+a"(a), "+a"(b), "+a"(c), "+a"(d));
}

// online softmax of one 32-query block over one 64-key tile: S^T (two 32-key sub-tiles) -> P fragments; returns the rescale factor
// (1 when the exponent reference did not move).  Identical to the in-line code of attn_fwd_kernel.
__device__ __forceinline__ bool softmax_block(const f32x16& s0, const f32x16& s1, float c, float& m_run, float& l_run, bf16x8 (&pf)[4],
                                              float& alpha) {
  constexpr float DEFER = 8.0f;
  float mx = s0[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s0[r]);
#pragma unroll
  for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s1[r]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float m_new = fmaxf(m_run, mx);
  const bool rescale = __builtin_amdgcn_ballot_w64((m_new - m_run) * c > DEFER) != 0;
  alpha = 1.0f;
  if (rescale) {
    alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    m_run = m_new;
  }
  const float mc = m_run * c;
  float psum = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pv[r] = __builtin_amdgcn_exp2f((t ? s1[r] : s0[r]) * c - mc);
      psum += pv[r];
    }
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      u32x4 pk;
#pragma unroll
      for (int e = 0; e < 4; ++e) {   // a VECTOR fptrunc: selected as one v_cvt_pk_bf16_f32 (two scalar casts became 2 x cvt + v_perm here)
        const f32x2 pr = {pv[8 * ks2 + 2 * e], pv[8 * ks2 + 2 * e + 1]};
        pk[e] = __builtin_bit_cast(unsigned int, __builtin_convertvector(pr, v3a_bf16x2));
      }
      pf[2 * t + ks2] = __builtin_bit_cast(bf16x8, pk);
    }
  }
  l_run = l_run * alpha + psum;
  return rescale;
}

__global__ __launch_bounds__(256, 1) void attn_fwd64_kernel(const AttnP p) {
  constexpr int D = 128, NW = 4, KV = 64, KROWB = 256, KTILE = KV * KROWB, VTILE = D * 128, STAGE = KTILE + VTILE;
  constexpr int KINS = KTILE / 1024 / NW, VINS = VTILE / 1024 / NW, KS = 8, DT = 4, OPITCH = D * 2 + 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int nqb = (p.Nq + 255) / 256;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = bid / nqb, qb = bid % nqb;
  const int b = bh / p.H, h = bh % p.H;
  const int q0 = qb * 256 + wave * 64;
  const char* Qb = p.q + ((size_t)b * p.q_bs + (size_t)h * D) * 2;
  const char* Kb = p.k + ((size_t)b * p.k_bs + (size_t)h * D) * 2;
  const char* Vb = p.vt + ((size_t)h * D * p.ldvt + (size_t)b * p.vt_bs) * 2;

  bf16x8 qf[2][KS];
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) {
    int qr = q0 + qq * 32 + l31;
    qr = qr < p.Nq ? qr : p.Nq - 1;
    const char* qp = Qb + (size_t)qr * p.ldq * 2 + hi * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[qq][ks] = *(const bf16x8*)(qp + ks * 32);
  }
  // DMA sources (same LDS image and swizzles as attn_fwd_kernel at hd = 128)
  const char* kp[KINS];
  const char* vp[VINS];
#pragma unroll
  for (int j = 0; j < KINS; ++j) {
    const int r = (j * NW + wave) * 4 + lane / 16, cch = lane % 16;
    kp[j] = Kb + (size_t)r * p.ldk * 2 + (size_t)(cch ^ (r & 15)) * 16;
  }
#pragma unroll
  for (int j = 0; j < VINS; ++j) {
    const int r = (j * NW + wave) * 8 + (lane >> 3);
    const int cch = (lane & 7) ^ ((r >> 1) & 7);
    vp[j] = Vb + ((size_t)r * p.ldvt + cch * 8) * 2;
  }
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(LDS_AS char*)smem);
  auto stage = [&](int s, int kt) {
    const unsigned sb = lds0 + s * STAGE + wave * 1024;
#pragma unroll
    for (int j = 0; j < KINS; ++j) glds16_raw(kp[j] + (size_t)kt * KV * p.ldk * 2, sb + j * NW * 1024);
#pragma unroll
    for (int j = 0; j < VINS; ++j) glds16_raw(vp[j] + (size_t)kt * KV * 2, sb + KTILE + j * NW * 1024);
  };
  const int pi = (l31 & 3) + 4 * (l31 >> 3) + 16 * ((l31 >> 2) & 1);
  int kfo[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) kfo[ks] = pi * KROWB + (((2 * ks + hi) ^ (pi & 15)) << 4);
  int vfo[4];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) vfo[c4] = l31 * 128 + (((4 * (c4 >> 1) + 2 * hi + (c4 & 1)) ^ ((l31 >> 1) & 7)) << 4);

  f32x16 oa[DT], ob[DT];    // O^T of q-block A / B
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { oa[i][r] = 0.f; ob[i][r] = 0.f; }
  float ma = -1e30f, mb = -1e30f, la = 0.f, lb = 0.f;
  const float c = p.scale_log2e;
  const int nkt = p.Nk / KV;

  stage(0, 0);
  // a wait hipcc SEES (the builtin, not an asm string): it retires the Q loads in the compiler's own bookkeeping - otherwise every
  // first use of a Q register inside the loop gets a compiler-inserted s_waitcnt vmcnt(7..0) that drains the hidden LDS-DMA stream
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_s_barrier();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);
    const char* sK = smem + cur * STAGE;
    const char* sV = sK + KTILE;
    f32x16 sa0, sa1, sb0, sb1;
    {   // S^T = K . Q^T: 16 K fragments (2 sub-tiles x 8 k steps), each feeding both q-blocks; fragment reads run two ahead of their MFMAs
        // (the MFMA statements are volatile asm: hipcc does not move LDS reads across them, so the distance is set here)
      bf16x8 kf[16];
      auto ldk = [&](int i) { return *(const bf16x8*)(sK + (i & 1) * 32 * KROWB + kfo[i >> 1]); };   // i = 2 ks + sub-tile
#pragma unroll
      for (int i = 0; i < 4; ++i) kf[i] = ldk(i);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {   // four accumulators in rotation (a dependent MFMA two slots behind its producer stalls)
        if (ks + 2 < KS) { kf[2 * ks + 4] = ldk(2 * ks + 4); kf[2 * ks + 5] = ldk(2 * ks + 5); }
        if (ks == 0) { mfma_s0(sa0, kf[0], qf[0][0]); mfma_s0(sb0, kf[0], qf[1][0]); mfma_s0(sa1, kf[1], qf[0][0]); mfma_s0(sb1, kf[1], qf[1][0]); }
        else {
          mfma_s(sa0, kf[2 * ks], qf[0][ks]); mfma_s(sb0, kf[2 * ks], qf[1][ks]);
          mfma_s(sa1, kf[2 * ks + 1], qf[0][ks]); mfma_s(sb1, kf[2 * ks + 1], qf[1][ks]);
        }
      }
    }
    mfma_fence_v(sa0, sa1);
    mfma_fence_v(sb0, sb1);
    bf16x8 pa[4], pb[4];
    float al;
    if (softmax_block(sa0, sa1, c, ma, la, pa, al)) {
      mfma_fence_a(oa[0], oa[1], oa[2], oa[3]);
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oa[i][r] *= al;
    }
    if (softmax_block(sb0, sb1, c, mb, lb, pb, al)) {
      mfma_fence_a(ob[0], ob[1], ob[2], ob[3]);
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[i][r] *= al;
    }
    {   // O^T += V^T . P^T: 16 V^T fragments (4 key chunks x 4 d tiles), each feeding both q-blocks, reads two ahead
      bf16x8 vf[16];
      auto ldv = [&](int j) { return *(const bf16x8*)(sV + (j & 3) * 4096 + vfo[j >> 2]); };
      vf[0] = ldv(0);
      vf[1] = ldv(1);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j + 2 < 16) vf[j + 2] = ldv(j + 2);
        if ((j & 3) == 0) asm volatile("s_nop 1" : "+v"(pa[j >> 2]), "+v"(pb[j >> 2]));   // VALU-written P -> MFMA operand
        mfma_o(oa[j & 3], vf[j], pa[j >> 2]);
        mfma_o(ob[j & 3], vf[j], pb[j >> 2]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  mfma_fence_a(oa[0], oa[1], oa[2], oa[3]);
  mfma_fence_a(ob[0], ob[1], ob[2], ob[3]);
  char* reg = smem + wave * (32 * OPITCH);
  char* Ob = p.o + ((size_t)b * p.o_bs + (size_t)h * D) * 2;
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) {
    const float lx = qq ? lb : la;
    const float inv = 1.0f / (lx + __shfl_xor(lx, 32, 64));
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const f32x16& o = qq ? ob[i] : oa[i];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 pk;
        pk[0] = pack_bf16x2(o[g * 4 + 0] * inv, o[g * 4 + 1] * inv);
        pk[1] = pack_bf16x2(o[g * 4 + 2] * inv, o[g * 4 + 3] * inv);
        *(u32x2*)(reg + l31 * OPITCH + (i * 32 + g * 8 + hi * 4) * 2) = pk;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = it * 64 + lane;
      const int ql = idx / 16, ch = idx % 16;
      const u32x2 lo = *(const u32x2*)(reg + ql * OPITCH + ch * 16);
      const u32x2 hi2 = *(const u32x2*)(reg + ql * OPITCH + ch * 16 + 8);
      const int qr = q0 + qq * 32 + ql;
      if (qr < p.Nq) {
        u32x4 v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = hi2[0]; v[3] = hi2[1];
        *(u32x4*)(Ob + ((size_t)qr * p.ldo) * 2 + ch * 16) = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- the software-pipelined form of attn_fwd64_kernel ---------------------------------------------------------------------------------
// With ONE wave per SIMD nothing else fills the matrix pipe while the wave runs its softmax (measured on the kernel above: 4400 cycles per
// 64-key tile for 2048 cycles of MFMAs - a lone wave issues one instruction per ~4 cycles, and the softmax of 64 queries is ~430 of them).
// Here the instruction stream itself interleaves the two: every pair of MFMAs is followed by its share of VALU work that belongs to a
// DIFFERENT tile (S^T is double-buffered in registers),
//   phase A (32 MFMAs):  S^T(t+1) = K(t+1) . Q^T        beside  finish-softmax(t): p = 2^x, row sums, bf16 packing  -> P(t)
//   phase B (32 MFMAs):  O^T     += V^T(t) . P(t)^T     beside  start-softmax(t+1): row max, rescale decision, x = s c - m c (in place)
// pinned by __builtin_amdgcn_sched_barrier(0) between the steps.  A rescale decided in phase B of tile t is applied to O right before the
// PV MFMAs of tile t+1 and to l in its finish-softmax.  Arithmetic per query row is unchanged (bit-identical to both kernels above).
struct SSet { f32x16 a0, a1, b0, b1; };   // S^T of q-block A / B, key sub-tiles 0 / 1 (after start-softmax: x = s c - m c)

__device__ __forceinline__ float max8(float m, const f32x16& s, int r0) {
#pragma unroll
  for (int r = 0; r < 8; ++r) m = fmaxf(m, s[r0 + r]);
  return m;
}
__device__ __forceinline__ float half_swap_max(float mx) {   // max over the two lane halves (the two 16-key halves of a query's row)
  // (v_permlane32_swap on two copies of mx would avoid the LDS round trip, but measured NOT equivalent here: ~40 % of the rows got a
  //  reference below their maximum - kept behind the experiment switch until understood)
  if constexpr (!(V3A_PW_ABL & 4)) return fmaxf(mx, __shfl_xor(mx, 32, 64));
  const unsigned u = __builtin_bit_cast(unsigned, mx);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}
// rescale decision of one q-block (same rule as attn_fwd_kernel): returns whether the wave moves its exponent reference
__device__ __forceinline__ bool softmax_decide(float mx, float c, float& m_run, float& alpha) {
  constexpr float DEFER = 8.0f;
  const float m_new = fmaxf(m_run, mx);
  const bool rescale = __builtin_amdgcn_ballot_w64((m_new - m_run) * c > DEFER) != 0;
  alpha = 1.0f;
  if (rescale) {
    alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    m_run = m_new;
  }
  return rescale;
}
__device__ __forceinline__ void fma8(f32x16& s, int r0, float c, float mc) {
  // asm: left to itself hipcc SLP-packs these into v_pk_fma_f32, which is slower than two v_fma_f32 beside MFMAs (MI355X_MICROARCH.md)
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float x = s[r0 + r];
    asm("v_fma_f32 %0, %1, %2, -%3" : "=v"(x) : "v"(x), "v"(c), "v"(mc));
    s[r0 + r] = x;
  }
}
// finish-softmax of four values: p = 2^x, psum += p (in order), two packed bf16 words
__device__ __forceinline__ void exp4(const f32x16& x, int r0, float& psum, unsigned& w0, unsigned& w1) {
  float pv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    pv[r] = (V3A_PW_ABL & 2) ? x[r0 + r] : __builtin_amdgcn_exp2f(x[r0 + r]);
    psum += pv[r];
  }
  const f32x2 p0 = {pv[0], pv[1]}, p1 = {pv[2], pv[3]};
  w0 = __builtin_bit_cast(unsigned int, __builtin_convertvector(p0, v3a_bf16x2));
  w1 = __builtin_bit_cast(unsigned int, __builtin_convertvector(p1, v3a_bf16x2));
}

// Value anchors: an empty volatile asm that "modifies" the values.  The MFMAs are volatile asm statements, which keep their order, so a
// computation whose result passes through an anchor cannot be SUNK by the IR optimiser below the MFMAs that follow the anchor (it sinks
// the finish-softmax to its first use otherwise: all 64 exponentials end up in one block in front of the PV MFMAs).
#define PIN1(a) asm volatile("" : "+v"(a))
#define PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define PIN3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
// O <- O alpha for one accumulator tile living in the accumulator file, through 16 temporaries at a time (the cold rescale path must not
// raise the register pressure of the loop: with all 128 values in flight hipcc spilled loop-invariant addresses to scratch)
__device__ __forceinline__ void scale_acc(f32x16& o, float alpha) {
  f32x16 t = o;
#pragma unroll
  for (int r = 0; r < 16; ++r) t[r] *= alpha;
  asm volatile("" : "+v"(t));
  o = t;
  asm volatile("s_nop 1" : "+a"(o));
}

#if V3A_PW_ABL & 32
__device__ long long g_pw_dbg[64];   // experiment builds: s_memtime stamps of one tile of block 0 / wave 0
extern "C" int v3a_debug_read(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pw_dbg), n * 8); }
#define STAMP(i) do { if (blockIdx.x == 0 && tid == 0 && kt == 20) g_pw_dbg[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ void pw_fence_s(SSet& n) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(n.a0), "+v"(n.a1), "+v"(n.b0), "+v"(n.b1)); }
__device__ __forceinline__ void pw_fence_o(f32x16 (&a)[4], f32x16 (&b)[4]) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+a"(a[0]), "+a"(a[1]), "+a"(a[2]), "+a"(a[3]), "+a"(b[0]), "+a"(b[1]), "+a"(b[2]), "+a"(b[3]));
}

// finish-softmax of two values: p = 2^x, psum += p (in order), one packed bf16 word
__device__ __forceinline__ unsigned exp2w(const f32x16& x, int r0, float& psum) {
  const float p0 = __builtin_amdgcn_exp2f(x[r0]), p1 = __builtin_amdgcn_exp2f(x[r0 + 1]);
  psum += p0;
  psum += p1;
  const f32x2 pr = {p0, p1};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(pr, v3a_bf16x2));
}

__global__ __launch_bounds__(256, 1) void attn_fwd64p_kernel(const AttnP p) {
  constexpr int D = 128, NW = 4, KV = 64, KROWB = 256, KTILE = KV * KROWB, VTILE = D * 128, STAGE = KTILE + VTILE;
  constexpr int KINS = KTILE / 1024 / NW, VINS = VTILE / 1024 / NW, KS = 8, DT = 4, OPITCH = D * 2 + 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int nqb = (p.Nq + 255) / 256;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = bid / nqb, qb = bid % nqb;
  const int b = bh / p.H, h = bh % p.H;
  const int q0 = qb * 256 + wave * 64;
  const char* Qb = p.q + ((size_t)b * p.q_bs + (size_t)h * D) * 2;
  const char* Kb = p.k + ((size_t)b * p.k_bs + (size_t)h * D) * 2;
  const char* Vb = p.vt + ((size_t)h * D * p.ldvt + (size_t)b * p.vt_bs) * 2;

  bf16x8 qf[2][KS];
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) {
    int qr = q0 + qq * 32 + l31;
    qr = qr < p.Nq ? qr : p.Nq - 1;
    const char* qp = Qb + (size_t)qr * p.ldq * 2 + hi * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[qq][ks] = *(const bf16x8*)(qp + ks * 32);
  }
  // DMA sources: piece j of a tile = rows (j NW + wave) 4 .. + 3 (K) / (j NW + wave) 8 .. + 7 (V^T); the swizzle term does not depend on j
  // (16 j and 32 j rows further on), so ONE per-lane pointer per operand and wave-uniform piece offsets do
  const char* kp0;
  const char* vp0;
  {
    const int r = wave * 4 + lane / 16, cch = lane % 16;
    kp0 = Kb + (size_t)r * p.ldk * 2 + (size_t)(cch ^ (r & 15)) * 16;
    const int rv = wave * 8 + (lane >> 3);
    vp0 = Vb + ((size_t)rv * p.ldvt + ((lane & 7) ^ ((rv >> 1) & 7)) * 8) * 2;
  }
  // LDS: K tile t in stage t & 1 at offset 0, V^T tile t in stage t & 1 at offset KTILE
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(LDS_AS char*)smem) + wave * 1024;
  const size_t kstep = (size_t)KV * p.ldk * 2, kpiece = (size_t)16 * p.ldk * 2, vpiece = (size_t)32 * p.ldvt * 2;
  const int last = p.Nk / KV - 1;
  // (tile indices past the end are clamped: the copy lands in a stage nobody reads any more - keeps the tile body branch-free)
  auto dma_k = [&](int j, int kt) {
    if constexpr (V3A_PW_ABL & 1) { if (kt > 1) return; }
    glds16_raw(kp0 + (size_t)min(kt, last) * kstep + j * kpiece, lds0 + (kt & 1) * STAGE + j * NW * 1024);
  };
  auto dma_v = [&](int j, int kt) {
    if constexpr (V3A_PW_ABL & 1) { if (kt > 1) return; }
    glds16_raw(vp0 + (size_t)min(kt, last) * KV * 2 + j * vpiece, lds0 + (kt & 1) * STAGE + KTILE + j * NW * 1024);
  };

  const int pi = (l31 & 3) + 4 * (l31 >> 3) + 16 * ((l31 >> 2) & 1);
  int kfo[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) kfo[ks] = pi * KROWB + (((2 * ks + hi) ^ (pi & 15)) << 4);
  int vfo[4];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) vfo[c4] = l31 * 128 + (((4 * (c4 >> 1) + 2 * hi + (c4 & 1)) ^ ((l31 >> 1) & 7)) << 4);

  f32x16 oa[DT], ob[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { oa[i][r] = 0.f; ob[i][r] = 0.f; }
  float ma = -1e30f, mb = -1e30f, la = 0.f, lb = 0.f;
  float al_a = 1.f, al_b = 1.f;      // rescale factors decided by the last start-softmax, pending for O and l
  bool rs_a = false, rs_b = false;
  const float c = p.scale_log2e;
  const int nkt = p.Nk / KV;
  u32x4 pa[4], pb[4];                // P(t) fragments (bf16 pairs) of q-block A / B

  auto ldk = [&](const char* sK, int j) {
    if constexpr (V3A_PW_ABL & 8) { bf16x8 z = {1, 2, 3, 4, 5, 6, 7, (short)j}; asm volatile("" : "+v"(z)); return z; }
    return *(const bf16x8*)(sK + (j >> 3) * 32 * KROWB + kfo[j & 7]);
  };
  auto ldv = [&](const char* sV, int j) {
    if constexpr (V3A_PW_ABL & 8) { bf16x8 z = {1, 2, 3, 4, 5, 6, 7, (short)j}; asm volatile("" : "+v"(z)); return z; }
    return *(const bf16x8*)(sV + (j & 3) * 4096 + vfo[j >> 2]);
  };
  // one S^T step = k step ks of BOTH key sub-tiles on both q-blocks: four MFMAs on four different accumulators.  A dependent MFMA
  // is only cheap straight behind its producer or >= 4 MFMAs later: with two accumulators in rotation (a0 b0 a0 b0 ..) every MFMA waited
  // for the write-back of the one before last - measured 60 cycles per MFMA with nothing else in the loop.
  auto s_step = [&](SSet& n, const bf16x8& k0, const bf16x8& k1, int ks) {
    if (ks == 0) { mfma_s0(n.a0, k0, qf[0][0]); mfma_s0(n.b0, k0, qf[1][0]); mfma_s0(n.a1, k1, qf[0][0]); mfma_s0(n.b1, k1, qf[1][0]); }
    else { mfma_s(n.a0, k0, qf[0][ks]); mfma_s(n.b0, k0, qf[1][ks]); mfma_s(n.a1, k1, qf[0][ks]); mfma_s(n.b1, k1, qf[1][ks]); }
  };
  // fragment order of the S^T phase: index 2 ks + t  (k step ks, sub-tile t)
  auto ldk2 = [&](const char* sK, int i) { return ldk(sK, (i & 1) * 8 + (i >> 1)); };
  // start-softmax of both q-blocks, not interleaved (prologue only)
  auto start_softmax = [&](SSet& n) {
    float mxa = n.a0[0], mxb = n.b0[0];
    mxa = max8(max8(max8(max8(mxa, n.a0, 0), n.a0, 8), n.a1, 0), n.a1, 8);
    mxb = max8(max8(max8(max8(mxb, n.b0, 0), n.b0, 8), n.b1, 0), n.b1, 8);
    rs_a = softmax_decide(half_swap_max(mxa), c, ma, al_a);
    rs_b = softmax_decide(half_swap_max(mxb), c, mb, al_b);
    const float mca = ma * c, mcb = mb * c;
    fma8(n.a0, 0, c, mca); fma8(n.a0, 8, c, mca); fma8(n.a1, 0, c, mca); fma8(n.a1, 8, c, mca);
    fma8(n.b0, 0, c, mcb); fma8(n.b0, 8, c, mcb); fma8(n.b1, 0, c, mcb); fma8(n.b1, 8, c, mcb);
  };

  // ---- prologue: K(0), V(0), K(1) in flight; S(0); start-softmax(0) ----
#pragma unroll
  for (int j = 0; j < KINS; ++j) dma_k(j, 0);
#pragma unroll
  for (int j = 0; j < VINS; ++j) dma_v(j, 0);
#pragma unroll
  for (int j = 0; j < KINS; ++j) dma_k(j, 1);
  __builtin_amdgcn_s_waitcnt(0);     // (a wait hipcc sees: retires the Q loads in its own bookkeeping, see attn_fwd64_kernel)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  SSet X, Y;
  {
    bf16x8 kf[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) kf[i] = ldk2(smem, i);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + 2 < 8) { kf[2 * ks + 4] = ldk2(smem, 2 * ks + 4); kf[2 * ks + 5] = ldk2(smem, 2 * ks + 5); }
      s_step(X, kf[2 * ks], kf[2 * ks + 1], ks);
    }
  }
  mfma_fence_v(X.a0, X.a1);
  mfma_fence_v(X.b0, X.b1);
  start_softmax(X);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();      // every wave has read K(0): its stage may take K(2)

  // one tile: `cur` holds x(t) (start-softmax done), `nxt` receives S(t+1)
  auto tile = [&](auto more_tag, SSet& cur, SSet& nxt, int kt) {
    constexpr bool more = decltype(more_tag)::value;
    const char* sKn = smem + ((kt + 1) & 1) * STAGE;
    const char* sV = smem + (kt & 1) * STAGE + KTILE;
    // ---------------- phase A: S(t+1) MFMAs | finish-softmax(t) | DMA issue of K(t+2), V(t+1) ----------------
    STAMP(0);
    float psa = 0.f, psb = 0.f;
    bf16x8 kf[16];
    if constexpr (more) {
#pragma unroll
      for (int i = 0; i < 4; ++i) kf[i] = ldk2(sKn, i);
    }
    // 32 slots: ONE MFMA, then the finish-softmax of two values (2 exp, 2 add, 1 packed word).  A lone wave issues in order, so VALU work
    // only overlaps an MFMA if it sits right behind it: four MFMAs followed by twenty VALU instructions ran as 128 + 20 x 5 cycles.
#pragma unroll
    for (int sl = 0; sl < 32; ++sl) {
      const int ks = sl >> 2, w = sl & 3;      // k step, (q-block, sub-tile) of this slot: a0 b0 a1 b1
      if constexpr (more) {
        if (w == 0 && ks + 2 < 8) kf[2 * ks + 4] = ldk2(sKn, 2 * ks + 4);
        if (w == 2 && ks + 2 < 8) kf[2 * ks + 5] = ldk2(sKn, 2 * ks + 5);
        f32x16& acc = w == 0 ? nxt.a0 : w == 1 ? nxt.b0 : w == 2 ? nxt.a1 : nxt.b1;
        const bf16x8& kfr = kf[2 * ks + (w >> 1)];
        if (ks == 0) mfma_s0(acc, kfr, qf[w & 1][0]);
        else mfma_s(acc, kfr, qf[w & 1][ks]);
      }
      {   // word wi (two values) of q-block A for even slots, B for odd slots; words in ascending (sub-tile, row) order per q-block
        const int wi = sl >> 1, t = wi >> 3, r0 = (wi & 7) * 2, f = 2 * t + (r0 >> 3), e = (r0 & 7) >> 1;
        unsigned word = 0x3f803f80u;
        if constexpr (!(V3A_PW_ABL & 16)) {
          if ((sl & 1) == 0) { word = exp2w(t ? cur.a1 : cur.a0, r0, psa); PIN2(word, psa); }
          else { word = exp2w(t ? cur.b1 : cur.b0, r0, psb); PIN2(word, psb); }
        }
        if ((sl & 1) == 0) pa[f][e] = word;
        else pb[f][e] = word;
      }
      if ((sl & 3) == 3) STAMP(8 + (sl >> 2));
      __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(1);
    // hipcc may copy S^T / O^T registers at the block boundaries that follow (they differ between the unrolled copies of this body):
    // the copies are plain VALU / v_accvgpr reads, which need the MFMA results WRITTEN - fence here, while the matrix pipe still works
    // off the last four MFMAs (found as wrong d-tile 3 of q-block A for even tile counts only)
    if constexpr (more) pw_fence_s(nxt);
    la = la * al_a + psa;
    lb = lb * al_b + psb;
    // ---------------- the rescale decided for THIS tile: O <- O alpha before its PV MFMAs ----------------
    if (rs_a) {
      mfma_fence_a(oa[0], oa[1], oa[2], oa[3]);
#pragma unroll
      for (int i = 0; i < DT; ++i) scale_acc(oa[i], al_a);
    }
    if (rs_b) {
      mfma_fence_a(ob[0], ob[1], ob[2], ob[3]);
#pragma unroll
      for (int i = 0; i < DT; ++i) scale_acc(ob[i], al_b);
    }
    asm volatile("s_nop 3" :::);   // v_accvgpr_write'n O / VALU-written P -> MFMA operands
    // ---------------- phase B: PV(t) MFMAs | start-softmax(t+1) ----------------
    STAMP(2);
    float mxa = 0.f, mxb = 0.f, mca = 0.f, mcb = 0.f;
    bf16x8 vf[16];
    vf[0] = ldv(sV, 0);
    vf[1] = ldv(sV, 1);
    // 32 slots: one PV MFMA, then its share of start-softmax(t+1): slots 4..19 row max (two values of A and B each, v_max3),
    // slot 20 the two rescale decisions, slots 21..28 (odd/even: 8 values each) x = s c - m c, DMA pieces in slots 24..31
#pragma unroll
    for (int sl = 0; sl < 32; ++sl) {
      const int j = sl >> 1;
      if ((sl & 1) == 0 && j + 2 < 16) vf[j + 2] = ldv(sV, j + 2);
      if ((sl & 1) == 0) mfma_o(oa[j & 3], vf[j], __builtin_bit_cast(bf16x8, pa[j >> 2]));
      else mfma_o(ob[j & 3], vf[j], __builtin_bit_cast(bf16x8, pb[j >> 2]));
      if constexpr (more && !(V3A_PW_ABL & 16)) {
        if (sl == 3) { mxa = nxt.a0[0]; mxb = nxt.b0[0]; }
        if (sl >= 4 && sl < 20) {
          const int u = sl - 4, r0 = (u & 7) * 2;       // slots 4..11: sub-tile 0, 12..19: sub-tile 1
          const f32x16& xa = (u >> 3) ? nxt.a1 : nxt.a0;
          const f32x16& xb = (u >> 3) ? nxt.b1 : nxt.b0;
          mxa = fmaxf(fmaxf(mxa, xa[r0]), xa[r0 + 1]);
          mxb = fmaxf(fmaxf(mxb, xb[r0]), xb[r0 + 1]);
          PIN2(mxa, mxb);
        }
        if (sl == 20) {
          rs_a = softmax_decide(half_swap_max(mxa), c, ma, al_a);
          rs_b = softmax_decide(half_swap_max(mxb), c, mb, al_b);
          mca = ma * c;
          mcb = mb * c;
          PIN2(mca, mcb);
        }
        if (sl >= 21 && sl < 29) {
          const int u = sl - 21;
          f32x16& xs = u < 4 ? ((u >> 1) ? nxt.a1 : nxt.a0) : (((u - 4) >> 1) ? nxt.b1 : nxt.b0);
          fma8(xs, (u & 1) * 8, c, u < 4 ? mca : mcb);
          PIN1(xs);
        }
      }
      if constexpr (more) {   // LDS-DMA of K(t+2) / V(t+1)
        if (sl >= 24 && sl < 28) dma_k(sl - 24, kt + 2);
        else if (sl >= 28) dma_v(sl - 28, kt + 1);
      }
      if (sl & 1) STAMP(16 + j);
      __builtin_amdgcn_sched_barrier(0);
    }
    pw_fence_o(oa, ob);
    // K(t+2) and V(t+1) have landed, every wave is done with K(t+1) and V(t)
    STAMP(3);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    STAMP(4);
    __builtin_amdgcn_s_barrier();
    STAMP(5);
  };
  constexpr std::true_type MORE{};
  constexpr std::false_type LAST{};
  int kt = 0;
  for (; kt + 2 <= nkt - 1; kt += 2) {
    tile(MORE, X, Y, kt);
    tile(MORE, Y, X, kt + 1);
  }
  if (kt < nkt - 1) {
    tile(MORE, X, Y, kt);
    tile(LAST, Y, X, kt + 1);
  } else {
    tile(LAST, X, Y, kt);
  }

  mfma_fence_a(oa[0], oa[1], oa[2], oa[3]);
  mfma_fence_a(ob[0], ob[1], ob[2], ob[3]);
  char* reg = smem + wave * (32 * OPITCH);
  char* Ob = p.o + ((size_t)b * p.o_bs + (size_t)h * D) * 2;
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) {
    const float lx = qq ? lb : la;
    const float inv = 1.0f / (lx + __shfl_xor(lx, 32, 64));
#pragma unroll
    for (int i = 0; i < DT; ++i) {
      const f32x16& o = qq ? ob[i] : oa[i];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 pk;
        pk[0] = pack_bf16x2(o[g * 4 + 0] * inv, o[g * 4 + 1] * inv);
        pk[1] = pack_bf16x2(o[g * 4 + 2] * inv, o[g * 4 + 3] * inv);
        *(u32x2*)(reg + l31 * OPITCH + (i * 32 + g * 8 + hi * 4) * 2) = pk;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = it * 64 + lane;
      const int ql = idx / 16, ch = idx % 16;
      const u32x2 lo = *(const u32x2*)(reg + ql * OPITCH + ch * 16);
      const u32x2 hi2 = *(const u32x2*)(reg + ql * OPITCH + ch * 16 + 8);
      const int qr = q0 + qq * 32 + ql;
      if (qr < p.Nq) {
        u32x4 v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = hi2[0]; v[3] = hi2[1];
        *(u32x4*)(Ob + ((size_t)qr * p.ldo) * 2 + ch * 16) = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

int launch_attn64(const AttnP& p, int B, void* stream, bool pipelined) {
  constexpr int LDS = 2 * (64 * 256 + 128 * 128);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)attn_fwd64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return V3A_ERR_LAUNCH;
    if (hipFuncSetAttribute((const void*)attn_fwd64p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return V3A_ERR_LAUNCH;
    attr = true;
  }
  const int nqb = (p.Nq + 255) / 256;
  if (pipelined) hipLaunchKernelGGL(attn_fwd64p_kernel, dim3((unsigned)(nqb * B * p.H)), dim3(256), LDS, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(attn_fwd64_kernel, dim3((unsigned)(nqb * B * p.H)), dim3(256), LDS, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

#endif   // V3A_ATTN_EXPERIMENTAL

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

static int g_attn_kernel = 0;
extern "C" int v3a_attention_set_kernel(int which) {
  const int prev = g_attn_kernel;
  if (which >= 0 && which <= 3) g_attn_kernel = which;
  return prev;
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
  static const bool two_per_cu = getenv("V3A_ATTN_OCC2") != nullptr;  // A/B switch for the older 2-workgroup schedule
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
#ifdef V3A_ATTN_EXPERIMENTAL
    if (g_attn_kernel >= 2 && p.kv_split == 1 && !a->kv_seg && a->Nk % 64 == 0 && !a->kv_period && a->Nq >= 256)
      return launch_attn64(p, a->B, stream, g_attn_kernel == 3);
#endif
    if (two_per_cu) return launch_attn<128, 4, false, false, false>(p, a->B, stream);
    // a sequence-parallel shard (Nq = N / P queries against all N keys) has too few 128-query blocks to occupy 256 CUs: 64-query
    // workgroups double the count; per-wave arithmetic and key order are unchanged, so the output stays bit-identical
    const long wgs4 = (long)a->B * a->H * ((a->Nq + 127) / 128) * p.kv_split;
    if (wgs4 < 256) return launch_attn<128, 2, false, false, true>(p, a->B, stream);
    if (a->Nk % 64 == 0 && !a->kv_period && !a->kv_seg) return launch_attn<128, 4, false, false, true, true>(p, a->B, stream);
    return launch_attn<128, 4, false, false, true>(p, a->B, stream);
  }
  return launch_attn<64, 4, false, false, false>(p, a->B, stream);  // recon (hd = 64): the 3-per-CU schedule measured no gain there
}
