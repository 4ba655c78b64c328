// Shared by the bf16 GEMM main loops (gemm_bf16.hip) and the halo-tile convolution (conv_halo.hip): the problem descriptor and the
// fused epilogue (bias -> activation -> bf16 park in LDS -> whole-row re-read -> gate / residuals -> 16-byte coalesced stores).
#pragma once
#include "common.h"
#include "../../include/vist3a_hip.h"

namespace {

struct GemmP {
  const char* A;
  const char* B;
  char* C;
  const float* bias;
  const char* res;
  const float* scale;
  int M, N, K;
  int lda, ldb, ldc, ldr;  // in elements
  int rpb, sstride;
  int act, flags;
  const char* res2;        // optional second residual (bf16), added after `res`
  int ldr2, res_mod;       // res_mod > 0: residual row = m % res_mod (broadcast table, e.g. positional embedding)
  int orow_group, orow_skip, orow_off;  // orow_group > 0: output row = m + (m / group) * skip + off
  // implicit-GEMM convolution (CONV instantiations only): A is a channels-last activation [T][H][W][Cin]
  const int* ktab;        // one packed entry per 8-channel K chunk: cin | dw<<16 | dh<<20 | dt<<24 | valid<<31
  int cT, cH, cW, cCin;   // input extent
  int oH, oW;             // output extent (M = oT*oH*oW)
  int sT, sH, sW;         // stride
  int pT, pH, pW;         // leading pad (trailing implied by the output extent)
  int ups;                // 1: taps address a nearest-exact 2x (H,W) upsample of the stored input
  int replicate;          // 1: clamp out-of-range taps (padding_mode="replicate"), 0: zero
  // e4m3 operands (gemm_pp_kernel<.., F8 = true> only): per-row dequantisation scales of A and B (fp32), applied to the accumulators
  const float* a_scale;   // [M]
  const float* b_scale;   // [N]
  // blockIdx.y = z selects one of several equally shaped problems (split-K slices, batched operands): byte offsets of A, B, C, res per z
  long az, bz, cz, rz;
  float* rowsq;           // optional by-product: sum of squares of every (row, 32-column block) of bf16(acc + bias), [M][N / 32] f32
  // split-bf16 convolution (CONV == 2 instantiations only): the lo planes of the (hi, lo) bf16 pairs; hi planes are A / C / res / res2
  const char* A_lo; char* C_lo; const char* res_lo; const char* res2_lo;
  // transposed tail (gemm_pp_kernel<3, true, .., TT = true> only): output columns n >= tcol0 go to Ct[(n - tcol0) * ldct + m] (bf16) instead
  // of C - the V^T third of a fused q | k | v projection
  char* Ct; int ldct, tcol0;
};

__device__ const uint4 g_zero16 = {0u, 0u, 0u, 0u};

// Under load (every CU streaming) a global load issued in the epilogue takes MICROSECONDS to come back, and nothing is left to hide
// it: an exposed bias load cost FFN1 30 us, more than the rest of its epilogue.  The ping-pong kernel therefore DMAs the tile's
// bias / gate-scale slices into LDS before its K loop (the oldest vector-memory operations of the wave: the counted waits of the
// main loop cover them) and the epilogue reads them from there.
struct EpiAux {
  const char* lds_bias = nullptr;    // f32 [tile columns] (or [tile rows] with BIAS_ROW), indexed relative to n0 (m0)
  const char* lds_scale = nullptr;   // f32 [tile columns] of the tile's (single) batch row of `scale`
  int m0 = 0, n0 = 0;
  int gstride = 32;                  // output rows between the wave's consecutive 32-row groups (halo conv: one image row = W pixels)
};

// two 16-byte LDS fragments -> the 8-VGPR operand of the f8f6f4 MFMA
__device__ __forceinline__ i32x8 frag8(const u32x4 a, const u32x4 b) {
  i32x8 r;
  r[0] = (int)a[0]; r[1] = (int)a[1]; r[2] = (int)a[2]; r[3] = (int)a[3];
  r[4] = (int)b[0]; r[5] = (int)b[1]; r[6] = (int)b[2]; r[7] = (int)b[3];
  return r;
}

// e4m3 GEMM: acc *= a_scale[m] * b_scale[n] (the product is formed first, in fp32), ahead of the bias.
template <int MT, int NTL>
__device__ __forceinline__ void gemm_dequant(const GemmP& p, f32x16 (&acc)[MT][NTL], int lane, int mw0, int nw) {
  const int hi = lane >> 5, l31 = lane & 31;
  float sa[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) sa[i] = p.a_scale[min(mw0 + i * 32 + l31, p.M - 1)];
#pragma unroll
  for (int j = 0; j < NTL; ++j) {   // one 32-column block at a time: with all NTL blocks' scales live the 256x256 tile spills
    f32x4 sb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) sb[g] = *(const f32x4*)(p.b_scale + min(nw + j * 32 + g * 8 + hi * 4, p.N - 4));
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][g * 4 + e] *= sa[i] * sb[g][e];
  }
}

// Epilogue shared by every main loop: the wave owns an (MT*32) x (NTL*32) output tile whose 32x32 blocks sit in acc[i][j] in
// D^T orientation (lane (l31, hi) holds rows m = l31, 4 consecutive columns per accumulator quad).  Per 32-row group:
//   phase 1: acc (bias already added by gemm_add_bias) -> bf16 -> this wave's private
//            LDS region (32 rows x WTN)
//   phase 2: whole-row re-read, fused elementwise, 16-byte coalesced stores
// The epilogue is latency-, not bandwidth-bound when written naively (a dependent global load per quad / per row chunk:
// measured 7 us of a 38 us FFN1 tile), so every global load is issued ahead of its use: the bias in one batch before phase 1, and
// residual / gate loads in batches of PFB row chunks, with EARLY the first batch of a group BEFORE phase 1 so that its latency
// hides under the convert-and-park work (not with a 128-register accumulator, where it would spill).
template <int ACT, int MT, int NTL, int PFB, bool EARLY, int DBG>
__device__ __forceinline__ void gemm_epilogue_act(const GemmP& p, f32x16 (&acc)[MT][NTL], char* smem, int wave, int lane,
                                                  int mw0, int nw, const EpiAux& aux) {
  constexpr int WTN = NTL * 32, PITCH = WTN * 2 + 8;
  if constexpr (DBG & 2) {   // development ablation: no epilogue at all (accumulators kept live)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTL; ++j) { f32x16 t = acc[i][j]; asm volatile("" : "+v"(t)); acc[i][j] = t; }
    return;
  }
  const int hi = lane >> 5, l31 = lane & 31;
  char* reg = smem + wave * (32 * PITCH);
  constexpr int CH = WTN / 8;
  constexpr int ITERS = 32 * CH / 64;
  static_assert((32 * CH) % 64 == 0, "epilogue chunking");
  constexpr int PB = PFB;
  static_assert(PFB > 0 && ITERS % PB == 0, "prefetch batch");
  const int flags = p.flags;
  const bool res_f32 = (flags & V3A_GEMM_RES_F32) != 0;
  const bool plain = !p.scale && !p.res && !p.res2 && !(flags & (V3A_GEMM_RELU_OUT | V3A_GEMM_OUT_F32)) && p.orow_group <= 0;

  // prefetched operands of one row chunk (8 consecutive columns of one output row)
  u32x4 pr0[PB], pr1[PB];   // residual: bf16 x8 in pr0, or f32 x8 in pr0|pr1
  f32x4 ps0[PB], ps1[PB];   // scale
  auto coords = [&](int mw, int it, int& m, int& n, int& ml, int& ch) {
    const int idx = it * 64 + lane;
    ml = idx / CH; ch = idx % CH;
    m = mw + ml; n = nw + ch * 8;
  };
  auto fetch = [&](int mw, int it, int s) {
    int m, n, ml, ch;
    coords(mw, it, m, n, ml, ch);
    m = min(m, p.M - 1); n = min(n, p.N - 8);   // out-of-range chunks read a valid address and are dropped at the store: no
    if (p.scale && !aux.lds_scale) {            // divergent branch around the loads (hipcc serialises loads under exec masks)
      const float* sp = p.scale + ((flags & V3A_GEMM_SCALE_PER_BATCH) ? (size_t)(m / p.rpb) * p.sstride : 0) + n;
      ps0[s] = *(const f32x4*)sp; ps1[s] = *(const f32x4*)(sp + 4);
    }
    if (p.res) {
      const int mr = p.res_mod > 0 ? m % p.res_mod : m;
      if (res_f32) {
        const float* rp = (const float*)p.res + (size_t)mr * p.ldr + n;
        pr0[s] = *(const u32x4*)rp; pr1[s] = *(const u32x4*)(rp + 4);
      } else {
        pr0[s] = *(const u32x4*)(p.res + ((size_t)mr * p.ldr + n) * 2);
      }
    }
  };
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int mw = mw0 + i * aux.gstride;
    if constexpr (EARLY) {
#pragma unroll
      for (int s = 0; s < PB; ++s) fetch(mw, s, s);
    }
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[i][j][g * 4 + e];
          if constexpr (ACT != V3A_ACT_NONE) {   // act(bf16(acc + bias)), applied here on the registers: phase 2 then only copies
            float x = round_bf16(v[e]);
            if constexpr (ACT == V3A_ACT_GELU_TANH) x = gelu_tanh(x);
            else if constexpr (ACT == V3A_ACT_GELU_ERF) x = gelu_erf(x);
            else if constexpr (ACT == V3A_ACT_SILU) x = silu(x);
            else x = fmaxf(x, 0.f);
            v[e] = x;
          }
        }
        u32x2 pk;
        pk[0] = pack_bf16x2(v[0], v[1]);
        pk[1] = pack_bf16x2(v[2], v[3]);
        *(u32x2*)(reg + l31 * PITCH + (j * 32 + g * 8 + hi * 4) * 2) = pk;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if constexpr (DBG & 4) continue;   // development ablation: park only
#pragma unroll
    for (int it0 = 0; it0 < ITERS; it0 += PB) {
      if (!EARLY || it0 > 0) {
#pragma unroll
        for (int s = 0; s < PB; ++s) fetch(mw, it0 + s, s);
      }
#pragma unroll
      for (int s = 0; s < PB; ++s) {
        int m, n, ml, ch;
        coords(mw, it0 + s, m, n, ml, ch);
        const u32x2 lo = *(const u32x2*)(reg + ml * PITCH + ch * 16);
        const u32x2 hi2 = *(const u32x2*)(reg + ml * PITCH + ch * 16 + 8);
        const bool inside = m < p.M && n < p.N;
        u32x4 raw;
        raw[0] = lo[0]; raw[1] = lo[1]; raw[2] = hi2[0]; raw[3] = hi2[1];
        if (plain) {   // nothing left to fuse: the parked bf16 row chunk IS the output
          if constexpr (!(DBG & 1)) {
            if (inside) *(u32x4*)(p.C + ((size_t)m * p.ldc + n) * 2) = raw;
          }
          continue;
        }
        float v[8];
        unpack_bf16x8(raw, v);
        if (p.scale) {
          if (aux.lds_scale) {
            const char* sp = aux.lds_scale + (min(n, p.N - 8) - aux.n0) * 4;
            ps0[s] = *(const f32x4*)sp; ps1[s] = *(const f32x4*)(sp + 16);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] *= ps0[s][e]; v[4 + e] *= ps1[s][e]; }
          if (flags & V3A_GEMM_ROUND_AFTER_SCALE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = round_bf16(v[e]);
          }
        }
        if (p.res) {
          if (res_f32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += __uint_as_float(pr0[s][e]); v[4 + e] += __uint_as_float(pr1[s][e]); }
          } else {
            float rf[8];
            unpack_bf16x8(pr0[s], rf);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rf[e];
          }
        }
        if (p.res2) {
          const u32x4 rr = *(const u32x4*)(p.res2 + ((size_t)min(m, p.M - 1) * p.ldr2 + min(n, p.N - 8)) * 2);
          float rf[8];
          unpack_bf16x8(rr, rf);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rf[e];
        }
        if (flags & V3A_GEMM_RELU_OUT) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        const size_t mo = p.orow_group > 0 ? (size_t)m + (size_t)(m / p.orow_group) * p.orow_skip + p.orow_off : (size_t)m;
        if constexpr (DBG & 1) {   // development ablation: everything but the global stores
          if (v[0] == 123456.789f) *(float*)p.C = v[1];
          continue;
        }
        if (flags & V3A_GEMM_OUT_F32) {
          float* cp = (float*)p.C + mo * p.ldc + n;
          f32x4 o0, o1;
#pragma unroll
          for (int e = 0; e < 4; ++e) { o0[e] = v[e]; o1[e] = v[4 + e]; }
          if (inside) {
            *(f32x4*)cp = o0;
            *(f32x4*)(cp + 4) = o1;
          }
        } else {
          const u32x4 o = pack_bf16x8(v);
          if (inside) *(u32x4*)(p.C + (mo * p.ldc + n) * 2) = o;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this group's reads returned before the next group overwrites the region
    __builtin_amdgcn_wave_barrier();
  }
}

// The activation is resolved ONCE per launch: with the switch inside the element loops the epilogue grew to 16 k instructions
// of scalar branches (five activations x 8 elements x every row chunk) and thrashed the instruction cache - any activation,
// even ReLU, cost FFN1 +70 us.
template <int MT, int NTL, int PFB, bool EARLY, int DBG = 0>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x16 (&acc)[MT][NTL], char* smem, int wave, int lane,
                                              int mw0, int nw, const EpiAux& aux = EpiAux{}) {
  switch (p.act) {
    case V3A_ACT_GELU_TANH: gemm_epilogue_act<V3A_ACT_GELU_TANH, MT, NTL, PFB, EARLY, DBG>(p, acc, smem, wave, lane, mw0, nw, aux); break;
    case V3A_ACT_GELU_ERF: gemm_epilogue_act<V3A_ACT_GELU_ERF, MT, NTL, PFB, EARLY, DBG>(p, acc, smem, wave, lane, mw0, nw, aux); break;
    case V3A_ACT_SILU: gemm_epilogue_act<V3A_ACT_SILU, MT, NTL, PFB, EARLY, DBG>(p, acc, smem, wave, lane, mw0, nw, aux); break;
    case V3A_ACT_RELU: gemm_epilogue_act<V3A_ACT_RELU, MT, NTL, PFB, EARLY, DBG>(p, acc, smem, wave, lane, mw0, nw, aux); break;
    default: gemm_epilogue_act<V3A_ACT_NONE, MT, NTL, PFB, EARLY, DBG>(p, acc, smem, wave, lane, mw0, nw, aux); break;
  }
}

// acc += bias (fp32), ahead of the epilogue: all bias loads of the wave tile are issued together (one exposed latency instead of
// one per accumulator quad) and only once, since the column bias is the same for every 32-row group.
template <int MT, int NTL>
__device__ __forceinline__ void gemm_add_bias(const GemmP& p, f32x16 (&acc)[MT][NTL], int lane, int mw0, int nw,
                                              const EpiAux& aux = EpiAux{}) {
  const int hi = lane >> 5, l31 = lane & 31;
  if (!p.bias) return;
  if (p.flags & V3A_GEMM_BIAS_ROW) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = min(mw0 + i * 32 + l31, p.M - 1);
      const float b = aux.lds_bias ? *(const float*)(aux.lds_bias + (m - aux.m0) * 4) : p.bias[m];
#pragma unroll
      for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += b;
    }
  } else {
    f32x4 bq[NTL][4];
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // N % 8 == 0 and n % 4 == 0: a quad is either entirely inside [0, N) or entirely outside (and then never stored), so
        // the load is made branch-free by clamping its address.  (With a per-quad `if` hipcc wrapped every load in exec-mask
        // control flow and waited for each one in turn: 12 serial L2 round trips, 5 us per tile.)
        const int n = min(nw + j * 32 + g * 8 + hi * 4, p.N - 4);
        bq[j][g] = aux.lds_bias ? *(const f32x4*)(aux.lds_bias + (n - aux.n0) * 4) : *(const f32x4*)(p.bias + n);
      }
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][g * 4 + e] += bq[j][g][e];
  }
}

// By-product for a norm FOLDED into the consumer (v3a_gemm_args.row_sumsq): the sum of squares of the bf16-rounded outputs of every
// (row, 32-column block) - rowsq[m * (N / 32) + n / 32], fp32.  Tile-shape independent: every wave tile is made of whole 32 x 32
// blocks, and a block's 32 values are added in the same order by every tile (16 per lane half in accumulator order, then the two
// halves), so the sequence-parallel forward (small tiles) and the unsharded one (ping-pong tiles) produce the same bits.
template <int MT, int NTL>
__device__ __forceinline__ void gemm_row_sumsq(const GemmP& p, const f32x16 (&acc)[MT][NTL], int lane, int mw0, int nw) {
  if (!p.rowsq) return;
  const int hi = lane >> 5, l31 = lane & 31;
  const int nb = p.N / 32;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = round_bf16(acc[i][j][r]);
        s = fmaf(v, v, s);
      }
      s += __shfl_xor(s, 32, 64);
      const int m = mw0 + i * 32 + l31, n = nw + j * 32;
      if (hi == 0 && m < p.M && n < p.N) p.rowsq[(size_t)m * nb + (n >> 5)] = s;
    }
}

// Transposed store of a wave's (MT*32) x (NTL*32) tile (D^T orientation, bias already added): Ct[(n - tcol0) * ldct + m] = bf16(acc).  The wave
// writes its tile TRANSPOSED into its private LDS region with 2-byte stores (lane (l31, hi) owns token row l31: consecutive lanes write
// consecutive tokens of one column, 64 contiguous bytes), then reads whole columns back - MT*32 tokens = 128 contiguous bytes each at MT = 2 -
// as 16-byte pieces and stores them coalesced.  TPITCH = 136 bytes per column: the two lane halves (columns 4 apart) land on disjoint banks.
template <int MT, int NTL>
__device__ __forceinline__ void gemm_store_transposed(const GemmP& p, f32x16 (&acc)[MT][NTL], char* smem, int wave, int lane, int mw0, int nw) {
  constexpr int ROWS = MT * 32, COLS = NTL * 32, TPITCH = ROWS * 2 + 8;
  const int hi = lane >> 5, l31 = lane & 31;
  char* reg = smem + wave * (COLS * TPITCH);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          *(unsigned short*)(reg + (j * 32 + g * 8 + hi * 4 + e) * TPITCH + (i * 32 + l31) * 2) = f32_to_bf16(acc[i][j][g * 4 + e]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  constexpr int CPC = ROWS / 8;                 // 16-byte pieces per column
  constexpr int ITERS = COLS * CPC / 64;
  static_assert((COLS * CPC) % 64 == 0, "transposed epilogue chunking");
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = it * 64 + lane;
    const int c = idx / CPC, pc = idx % CPC;
    const int n = nw + c, m = mw0 + pc * 8;
    const u32x2 a = *(const u32x2*)(reg + c * TPITCH + pc * 16), b = *(const u32x2*)(reg + c * TPITCH + pc * 16 + 8);
    u32x4 v;
    v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
    if (n < p.N && m + 8 <= p.M) *(u32x4*)(p.Ct + ((size_t)(n - p.tcol0) * p.ldct + m) * 2) = v;   // (M % 8 == 0 is required by the entry point)
  }
}

// x -> (hi, lo) with hi = bf16(x) and lo = bf16(x - hi): x = hi + lo to 2^-17 relative (16 significand bits + the sign of lo).
__device__ __forceinline__ void split_bf16x8(const float* v, u32x4& hi, u32x4& lo) {
  hi = pack_bf16x8(v);
  float h[8], r[8];
  unpack_bf16x8(hi, h);
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = v[e] - h[e];
  lo = pack_bf16x8(r);
}
__device__ __forceinline__ void add_pair8(const u32x4 hi, const u32x4 lo, float* v) {
  float a[8], b[8];
  unpack_bf16x8(hi, a);
  unpack_bf16x8(lo, b);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] += a[e] + b[e];
}

// Epilogue of the split-bf16 convolution (v3a_conv_split): nothing is rounded to bf16 on the way.  The wave parks one 32-row group of
// fp32 accumulators (bias added, ReLU applied) in its private LDS region, re-reads whole rows and finishes 8 columns per lane:
//   v = act(acc + bias) + residual (f32 table, or a (hi, lo) pair) + residual2 (pair) ; [ReLU] ; store f32, or split into the (hi, lo) planes.
template <int MT, int NTL>
__device__ __forceinline__ void gemm_epilogue_f32(const GemmP& p, f32x16 (&acc)[MT][NTL], char* smem, int wave, int lane, int mw0, int nw,
                                                  int gstride = 32) {   // gstride: output rows between the wave's 32-row groups (halo tiles: W)
  constexpr int WTN = NTL * 32, PITCH = WTN * 4 + 16;
  constexpr int CH = WTN / 8, ITERS = 32 * CH / 64;
  static_assert((32 * CH) % 64 == 0, "epilogue chunking");
  const int hi = lane >> 5, l31 = lane & 31;
  char* reg = smem + wave * (32 * PITCH);
  const int flags = p.flags;
  const bool relu_in = p.act == V3A_ACT_RELU, res_f32 = (flags & V3A_GEMM_RES_F32) != 0;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int mw = mw0 + i * gstride;
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = relu_in ? fmaxf(acc[i][j][g * 4 + e], 0.f) : acc[i][j][g * 4 + e];
        *(f32x4*)(reg + l31 * PITCH + (j * 32 + g * 8 + hi * 4) * 4) = v;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = it * 64 + lane;
      const int ml = idx / CH, ch = idx % CH;
      const int m = mw + ml, n = nw + ch * 8;
      const bool inside = m < p.M && n < p.N;
      const int mc = min(m, p.M - 1), nc = min(n, p.N - 8);   // clamped addresses: loads stay branch-free, the store is predicated
      u32x4 r0 = {}, r1 = {}, q0 = {}, q1 = {};
      if (p.res) {
        const int mr = p.res_mod > 0 ? mc % p.res_mod : mc;
        if (res_f32) {
          const float* rp = (const float*)p.res + (size_t)mr * p.ldr + nc;
          r0 = *(const u32x4*)rp; r1 = *(const u32x4*)(rp + 4);
        } else {
          r0 = *(const u32x4*)(p.res + ((size_t)mr * p.ldr + nc) * 2);
          r1 = *(const u32x4*)(p.res_lo + ((size_t)mr * p.ldr + nc) * 2);
        }
      }
      if (p.res2) {
        q0 = *(const u32x4*)(p.res2 + ((size_t)mc * p.ldr2 + nc) * 2);
        q1 = *(const u32x4*)(p.res2_lo + ((size_t)mc * p.ldr2 + nc) * 2);
      }
      const f32x4 a = *(const f32x4*)(reg + ml * PITCH + ch * 32), b = *(const f32x4*)(reg + ml * PITCH + ch * 32 + 16);
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
      if (p.res) {
        if (res_f32) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] += __uint_as_float(r0[e]); v[4 + e] += __uint_as_float(r1[e]); }
        } else {
          add_pair8(r0, r1, v);
        }
      }
      if (p.res2) add_pair8(q0, q1, v);
      if (flags & V3A_GEMM_RELU_OUT) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      const size_t mo = p.orow_group > 0 ? (size_t)m + (size_t)(m / p.orow_group) * p.orow_skip + p.orow_off : (size_t)m;
      if (flags & V3A_GEMM_OUT_F32) {
        float* cp = (float*)p.C + mo * p.ldc + n;
        f32x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o0[e] = v[e]; o1[e] = v[4 + e]; }
        if (inside) { *(f32x4*)cp = o0; *(f32x4*)(cp + 4) = o1; }
      } else {
        u32x4 oh, ol;
        split_bf16x8(v, oh, ol);
        if (inside) {
          *(u32x4*)(p.C + (mo * p.ldc + n) * 2) = oh;
          *(u32x4*)(p.C_lo + (mo * p.ldc + n) * 2) = ol;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

// the implicit-GEMM view of a convolution (M = output pixels, N = Cout, K = Kpad) + the epilogue operands
inline GemmP conv_gemm_params(const v3a_conv_args* a) {
  GemmP p = {};
  p.A = (const char*)a->x; p.B = (const char*)a->w; p.C = (char*)a->y;
  p.bias = a->bias; p.res = (const char*)a->residual; p.scale = a->scale;
  p.M = a->oT * a->oH * a->oW; p.N = a->Cout; p.K = a->Kpad;
  p.lda = 0; p.ldb = a->Kpad; p.ldc = a->ldy; p.ldr = a->ldr;
  p.rpb = 1; p.sstride = 0;
  p.act = a->act; p.flags = a->flags & ~V3A_GEMM_SCALE_PER_BATCH;
  p.ktab = a->ktab;
  p.cT = a->T; p.cH = a->H; p.cW = a->W; p.cCin = a->Cin;
  p.oH = a->oH; p.oW = a->oW;
  p.sT = a->sT; p.sH = a->sH; p.sW = a->sW;
  p.pT = a->pT; p.pH = a->pH; p.pW = a->pW;
  p.ups = a->ups2 ? 1 : 0; p.replicate = a->replicate ? 1 : 0;
  p.res2 = (const char*)a->residual2; p.ldr2 = a->ldr2; p.res_mod = a->res_row_mod;
  p.orow_group = a->out_row_group; p.orow_skip = a->out_row_skip; p.orow_off = a->out_row_off;
  return p;
}

}  // namespace
