// bf16 NT GEMM for gfx950:  C[M,N] = epilogue(A[M,K] . B[N,K]^T), fp32 accumulation on
// v_mfma_f32_32x32x16_bf16.  Replaces nn.Linear under autocast on the DiT / ViT rows of SURVEY.md §8
// (A3,A4,A6,A7,A8,A9,R4,R6,R7,R9).
//
// Structure (one workgroup = one BMxBN output tile, 64-wide waves in a WMxWN grid):
//   * K is walked in BK=64 slabs.  A and B slabs are copied global->LDS with 16-byte LDS-DMA
//     (global_load_lds_dwordx4): one wave instruction moves 8 rows x 128 B.  The LDS image is
//     lane-linear, so the bank swizzle  chunk' = chunk ^ ((row>>1)&7)  is applied to the per-lane
//     SOURCE address and again on the fragment read (both sides or neither).
//   * two LDS stages; the DMA for slab t+1 is issued before the MFMAs of slab t and retired by a
//     counted wait + one s_barrier per slab.
//   * MFMA operands are issued as (B-fragment, A-fragment) so each lane ends up with 4 CONSECUTIVE
//     output columns per accumulator quad (D^T orientation): the epilogue packs them to bf16, parks
//     the wave's tile in LDS, and re-reads whole rows so that bias / activation / gate / residual and
//     the global stores are all 16-byte coalesced.
//   * workgroup ids are remapped so that each XCD (private 4 MiB L2) owns a contiguous run of tiles.
#include <cstdlib>

#include "common.h"
#include "../../include/vist3a_hip.h"
#include <type_traits>

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
};

__device__ const uint4 g_zero16 = {0u, 0u, 0u, 0u};

template <int BM, int BN, int WM, int WN, int BK, int NS>
struct TileCfg {
  static constexpr int NW = WM * WN;
  static constexpr int NTHR = NW * 64;
  static constexpr int WTM = BM / WM, WTN = BN / WN;
  static constexpr int MT = WTM / 32, NTL = WTN / 32;
  static constexpr int ROWS = BM + BN;
  static constexpr int RB = BK * 2;            // bytes per tile row in LDS
  static constexpr int CPR = RB / 16;          // 16-B chunks per row (8 or 4)
  static constexpr int RPI = 64 / CPR;         // rows covered by one 1-KiB DMA instruction (8 or 16)
  static constexpr int STAGE = ROWS * RB;
  static constexpr int NINS = ROWS / RPI;      // DMA instructions per stage (whole workgroup)
  static constexpr int NL = (NINS + NW - 1) / NW;   // ... per wave (upper bound)
  static constexpr int NLMIN = NINS / NW;           // ... per wave (lower bound: counted vmcnt uses this)
  static constexpr int NLA = BM / RPI / NW;    // (CONV) instructions of every wave that fall in the A tile
  static constexpr int KSTEPS = BK / 16;
  static constexpr int PITCH = WTN * 2 + 8;
  static constexpr int EPI_BYTES = NW * 32 * PITCH;   // the epilogue parks one 32-row group per wave at a time
  static constexpr int LDS_BYTES = (NS * STAGE > EPI_BYTES) ? NS * STAGE : EPI_BYTES;
  static constexpr bool CONV_OK = (BM / RPI) % NW == 0 && NINS % NW == 0;
  static_assert(BK == 32 || BK == 64, "BK");
  static_assert(NS >= 2, "ring depth");
  static_assert(ROWS % RPI == 0 && BM % RPI == 0, "staging instruction must not straddle A/B");
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of 32x32");
  static_assert(BM % 32 == 0, "swizzle assumes B rows start at a multiple of 32");
};

// STG = 0: tiles arrive by LDS-DMA (global_load_lds) into an NS-deep ring.
// STG = 1: tiles are staged through registers (global_load_dwordx4 -> ds_write_b128, 2 LDS buffers): the loads of slab t+2
//          are issued, and the slab t+1 registers written to LDS, BETWEEN the MFMA groups of slab t.  An LDS-DMA instruction
//          occupies its wave for ~100+ cycles at issue and every wave of the workgroup issues them at the same point, so
//          DMA staging leaves the matrix pipe idle for ~40 % of each slab (measured: 1468 TF without refill vs 830 with).
// OCC = workgroups the kernel is compiled to co-reside per CU (register cap = 512 / (OCC * waves per SIMD)): with OCC = 2
// one workgroup's epilogue / DMA-issue bubbles are filled by the other's MFMAs.
template <int BM, int BN, int WM, int WN, int BK, int NS, bool CONV, int STG, int OCC = 1>
__global__ __launch_bounds__(WM* WN * 64, (OCC * WM * WN + 3) / 4) void gemm_nt_kernel(const GemmP p) {
  using T = TileCfg<BM, BN, WM, WN, BK, NS>;
  constexpr int NW = T::NW, MT = T::MT, NTL = T::NTL, NL = T::NL, STAGE = T::STAGE;
  constexpr int WTM = T::WTM, WTN = T::WTN, PITCH = T::PITCH;
  constexpr int RB = T::RB, CPR = T::CPR, RPI = T::RPI, NINS = T::NINS, KSTEPS = T::KSTEPS;
  static_assert(!CONV || T::CONV_OK, "conv needs an even A/B instruction split");
  static_assert(STG == 0 || NS == 2, "register staging uses two LDS buffers");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int hi = lane >> 5, l31 = lane & 31;

  const int tilesN = (p.N + BN - 1) / BN, tilesM = (p.M + BM - 1) / BM;
  const int t = xcd_remap(blockIdx.x, tilesM * tilesN);
  // Grouped raster: consecutive tiles walk down a band of GM row-tiles before moving to the next column, so the ~64 tiles in
  // flight on one XCD (and the XCD's whole contiguous chunk) cover a near-square patch: 8 A row-panels + 8 W panels per 64
  // tiles instead of 1-2 A panels + every W panel (FFN1: 535 MB -> ~260 MB of L2 fills per launch).
  constexpr int GM = (OCC * 32 * BN / BM >= 36) ? 8 : 4;  // ~sqrt(tiles in flight per XCD x BN/BM): squarest in-flight patch
  int tm, tn;
  if (p.flags & (1 << 27)) {  // A/B switch: plain row-major order
    tm = t / tilesN; tn = t % tilesN;
  } else {
    const int gsz = GM * tilesN, gid = t / gsz, first = gid * GM;
    const int gm = min(tilesM - first, GM), r = t - gid * gsz;
    tm = first + r % gm; tn = r / gm;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-lane staging sources (advance by RB bytes per K slab) ----
  constexpr int NLA = T::NLA;
  const char* gp[NL];
  int ldso[STG == 1 ? NL : 1];
  u32x4 rg[STG == 1 ? NL : 1];
  int cvt[CONV ? NLA : 1], cvh[CONV ? NLA : 1], cvw[CONV ? NLA : 1], cvc[CONV ? NLA : 1];
  int* ktl = (int*)(smem + T::LDS_BYTES);  // CONV: K-chunk table copied behind the tile ring
  auto swz = [](int row) { return CPR == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int g = j * NW + wave;  // wave-uniform DMA instruction index = RPI-row group
    const int R = g * RPI + lane / CPR;
    // LDS-DMA writes lane-linear, so the swizzle goes on the SOURCE chunk; register staging reads linear and swizzles the
    // ds_write address instead.  Either way LDS position (R, c') holds global chunk c' ^ swz(R).
    const int c = STG == 0 ? ((lane % CPR) ^ swz(R)) : (lane % CPR);
    if (STG == 1) ldso[j] = R * RB + (((lane % CPR) ^ swz(R)) << 4);
    if (g * RPI < BM) {
      int row = m0 + R;
      row = row < p.M ? row : p.M - 1;
      if constexpr (CONV) {
        if (j < NLA) {
          const int ohw = p.oH * p.oW;
          const int t = row / ohw, rem = row - t * ohw;
          const int h = rem / p.oW;
          cvt[j] = t * p.sT - p.pT; cvh[j] = h * p.sH - p.pH; cvw[j] = (rem - h * p.oW) * p.sW - p.pW;
          cvc[j] = c;
        }
        gp[j] = p.A;
      } else {
        gp[j] = p.A + ((size_t)row * p.lda) * 2 + c * 16;
      }
    } else {
      int row = n0 + (R - BM);
      row = row < p.N ? row : p.N - 1;
      gp[j] = p.B + ((size_t)row * p.ldb) * 2 + c * 16;
    }
  }
  if constexpr (CONV) {
    for (int i = tid; i < p.K / 8; i += T::NTHR) ktl[i] = p.ktab[i];
    __syncthreads();
  }
  int kslab = 0;
  // issues this wave's DMA instructions j with j % nparts == part of the slab into ring slot s (the K-loop spreads the
  // parts between its MFMA groups so that VMEM issue overlaps matrix-pipe time instead of preceding it)
  auto stage = [&](int s, int part, int nparts) {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      if (j % nparts != part) continue;
      const int g = j * NW + wave;
      if (NINS % NW != 0 && g >= NINS) continue;  // ragged last round: wave-uniform skip
      if constexpr (CONV) {
        if (j < NLA) {  // (BM/RPI) % NW == 0: instruction j < NLA is an A-tile instruction for every wave
          const int e = ktl[kslab * CPR + cvc[j]];
          int tt = cvt[j] + ((e >> 24) & 15), hh = cvh[j] + ((e >> 20) & 15), ww = cvw[j] + ((e >> 16) & 15);
          const int eH = p.ups ? p.cH * 2 : p.cH, eW = p.ups ? p.cW * 2 : p.cW;
          bool ok = e < 0;  // bit 31 = valid chunk
          if (p.replicate) {
            tt = tt < 0 ? 0 : (tt >= p.cT ? p.cT - 1 : tt);
            hh = hh < 0 ? 0 : (hh >= eH ? eH - 1 : hh);
            ww = ww < 0 ? 0 : (ww >= eW ? eW - 1 : ww);
          } else {
            ok = ok && tt >= 0 && tt < p.cT && hh >= 0 && hh < eH && ww >= 0 && ww < eW;
          }
          if (p.ups) { hh >>= 1; ww >>= 1; }
          const char* src = p.A + (((size_t)(tt * p.cH + hh) * p.cW + ww) * p.cCin + (e & 0xffff)) * 2;
          glds16(ok ? src : (const char*)&g_zero16, smem + s * STAGE + g * 1024);
          continue;
        }
      }
      glds16(gp[j], smem + s * STAGE + g * 1024);
      gp[j] += RB;
    }
    if (part == nparts - 1) ++kslab;
  };
  auto load_regs = [&]() {  // STG == 1: this wave's share of the next slab -> registers (compiler-counted vmcnt)
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int g = j * NW + wave;
      if (NINS % NW != 0 && g >= NINS) continue;
      if constexpr (CONV) {
        if (j < NLA) {
          const int e = ktl[kslab * CPR + cvc[j]];
          int tt = cvt[j] + ((e >> 24) & 15), hh = cvh[j] + ((e >> 20) & 15), ww = cvw[j] + ((e >> 16) & 15);
          const int eH = p.ups ? p.cH * 2 : p.cH, eW = p.ups ? p.cW * 2 : p.cW;
          bool ok = e < 0;
          if (p.replicate) {
            tt = tt < 0 ? 0 : (tt >= p.cT ? p.cT - 1 : tt);
            hh = hh < 0 ? 0 : (hh >= eH ? eH - 1 : hh);
            ww = ww < 0 ? 0 : (ww >= eW ? eW - 1 : ww);
          } else {
            ok = ok && tt >= 0 && tt < p.cT && hh >= 0 && hh < eH && ww >= 0 && ww < eW;
          }
          if (p.ups) { hh >>= 1; ww >>= 1; }
          const char* src = p.A + (((size_t)(tt * p.cH + hh) * p.cW + ww) * p.cCin + (e & 0xffff)) * 2;
          rg[j] = *(const u32x4*)(ok ? src : (const char*)&g_zero16);
          continue;
        }
      }
      rg[j] = *(const u32x4*)gp[j];
      gp[j] += RB;
    }
    ++kslab;
  };
  auto write_lds = [&](int s) {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int g = j * NW + wave;
      if (NINS % NW != 0 && g >= NINS) continue;
      *(u32x4*)(smem + s * STAGE + ldso[j]) = rg[j];
    }
  };

  f32x16 acc[MT][NTL];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int sw = swz(l31);
  int koff[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) koff[ks] = l31 * RB + (((2 * ks + hi) ^ sw) << 4);
  const int aoff = (wm * WTM) * RB, boff = (BM + wn * WTN) * RB;

  const int nk = p.K / BK;
  if constexpr (STG == 0) {
  // NS-deep LDS ring fed by LDS-DMA.  Slabs kt+1 .. kt+NS-2 stay in flight ACROSS the barrier (counted vmcnt, raw
  // s_barrier): the DMA latency (~1 us under load) is covered by NS-2 slabs of MFMA work instead of one.
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0)
    if (s0 < nk) stage(s0, 0, 1);
  int slot = 0, fill = NS - 1;
  // one K slab: wait for its DMA (counted), barrier, then MFMA groups with the refill DMA spread between them
  auto slab = [&](auto more_tag, auto drain_tag) {
    constexpr bool MORE = decltype(more_tag)::value, DRAIN = decltype(drain_tag)::value;
    // lgkmcnt(0): every fragment read of the previous slab has RETURNED before this wave lets others refill that slot (the
    // compiler is free to sink the last MFMAs, and the waits for their operands, below a bare s_barrier)
    if constexpr (DRAIN) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(T::NLMIN * (NS - 2)) : "memory");
    __builtin_amdgcn_s_barrier();  // slab visible to all waves; everyone is done reading the slot about to be refilled
    if constexpr (MORE) stage(fill, 0, 1);  // measured: issuing the refill up front beats spreading it between MFMA groups
    const char* sA = smem + slot * STAGE + aoff;
    const char* sB = smem + slot * STAGE + boff;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      bf16x8 a[MT], b[NTL];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = *(const bf16x8*)(sA + i * 32 * RB + koff[ks]);
#pragma unroll
      for (int j = 0; j < NTL; ++j) b[j] = *(const bf16x8*)(sB + j * 32 * RB + koff[ks]);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    slot = slot + 1 == NS ? 0 : slot + 1;
    fill = fill + 1 == NS ? 0 : fill + 1;
  };
  using TT = std::true_type;
  using FF = std::false_type;
  int kt = 0;
  for (; kt + NS - 1 < nk; ++kt) slab(TT{}, FF{});   // steady state: refill in flight, NS-2 later slabs outstanding
  for (; kt < nk; ++kt) slab(FF{}, TT{});            // tail: nothing left to issue, drain
  } else {
    // register-staged pipeline, two LDS buffers
    auto mma_step = [&](const char* sA, const char* sB, int ks) {
      bf16x8 a[MT], b[NTL];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = *(const bf16x8*)(sA + i * 32 * RB + koff[ks]);
#pragma unroll
      for (int j = 0; j < NTL; ++j) b[j] = *(const bf16x8*)(sB + j * 32 * RB + koff[ks]);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    };
    load_regs();
    write_lds(0);
    if (nk > 1) load_regs();
    __syncthreads();
    auto slab = [&](int cur, auto write_tag, auto load_tag) {
      const char* sA = smem + cur * STAGE + aoff;
      const char* sB = smem + cur * STAGE + boff;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        mma_step(sA, sB, ks);
        if (ks == KSTEPS - 2) {
          // slab t+1: registers -> the buffer read during slab t-1 (the loads had a whole slab of MFMA time to land) ...
          if constexpr (decltype(write_tag)::value) write_lds(cur ^ 1);
          // ... and at once re-issue the same registers for slab t+2 (consumed one slab from now)
          if constexpr (decltype(load_tag)::value) { load_regs(); __builtin_amdgcn_sched_barrier(0); }
        }
      }
      __syncthreads();
    };
    using TT = std::true_type;
    using FF = std::false_type;
    int kt = 0;
    for (; kt + 2 < nk; ++kt) slab(kt & 1, TT{}, TT{});
    if (kt + 1 < nk) { slab(kt & 1, TT{}, FF{}); ++kt; }
    if (kt < nk) slab(kt & 1, FF{}, FF{});
  }
  if constexpr (STG == 0) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // all fragment reads retired before the ring is reused by the epilogue
  }

  if (p.flags & (1 << 29)) {  // DEBUG/profiling only: skip the epilogue (accumulators kept live by a never-taken store)
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    if (sum == 123456.789f) ((float*)p.C)[tid] = sum;
    return;
  }
  // ---- epilogue: per 32-row group of the wave tile:
  //   phase 1: acc + bias -> bf16 -> this wave's private LDS region (32 rows x WTN)
  //   phase 2: whole-row re-read, fused elementwise, 16-byte coalesced stores
  char* reg = smem + wave * (32 * PITCH);
  const int mw0 = m0 + wm * WTM, nw = n0 + wn * WTN;
  const bool bias_row = (p.flags & V3A_GEMM_BIAS_ROW) != 0;
  constexpr int CH = WTN / 8;
  constexpr int ITERS = 32 * CH / 64;
  static_assert((32 * CH) % 64 == 0, "epilogue chunking");
  const int act = p.act, flags = p.flags;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int mw = mw0 + i * 32;
    {
      const int ml = l31;
      float brow = 0.f;
      if (p.bias && bias_row) {
        int m = mw + ml;
        brow = p.bias[m < p.M ? m : p.M - 1];
      }
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = j * 32 + g * 8 + hi * 4;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][g * 4 + e];
          if (p.bias) {
            if (bias_row) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += brow;
            } else {
              const int n = nw + nl;
              if (n + 3 < p.N) {
                const f32x4 bv = *(const f32x4*)(p.bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bv[e];
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (n + e < p.N) v[e] += p.bias[n + e];
              }
            }
          }
          u32x2 pk;
          pk[0] = pack_bf16x2(v[0], v[1]);
          pk[1] = pack_bf16x2(v[2], v[3]);
          *(u32x2*)(reg + ml * PITCH + nl * 2) = pk;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll 2
    for (int it = 0; it < ITERS; ++it) {
    const int idx = it * 64 + lane;
    const int ml = idx / CH, ch = idx % CH;
    const int m = mw + ml, n = nw + ch * 8;
    const u32x2 lo = *(const u32x2*)(reg + ml * PITCH + ch * 16);
    const u32x2 hi2 = *(const u32x2*)(reg + ml * PITCH + ch * 16 + 8);
    if (m >= p.M || n >= p.N) continue;
    if (p.flags & (1 << 30)) continue;  // DEBUG/profiling only: phase 2 without global traffic
    u32x4 raw;
    raw[0] = lo[0]; raw[1] = lo[1]; raw[2] = hi2[0]; raw[3] = hi2[1];
    float v[8];
    unpack_bf16x8(raw, v);
    if (act != V3A_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = v[e];
        if (act == V3A_ACT_GELU_TANH) x = gelu_tanh(x);
        else if (act == V3A_ACT_GELU_ERF) x = gelu_erf(x);
        else if (act == V3A_ACT_SILU) x = silu(x);
        else x = fmaxf(x, 0.f);
        v[e] = round_bf16(x);
      }
    }
    if (p.scale) {
      const float* sp = p.scale + ((flags & V3A_GEMM_SCALE_PER_BATCH) ? (size_t)(m / p.rpb) * p.sstride : 0) + n;
      const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] *= s0[e]; v[4 + e] *= s1[e]; }
      if (flags & V3A_GEMM_ROUND_AFTER_SCALE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = round_bf16(v[e]);
      }
    }
    if (p.res) {
      const int mr = p.res_mod > 0 ? m % p.res_mod : m;
      if (flags & V3A_GEMM_RES_F32) {
        const float* rp = (const float*)p.res + (size_t)mr * p.ldr + n;
        const f32x4 r0 = *(const f32x4*)rp, r1 = *(const f32x4*)(rp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
      } else {
        const u32x4 rr = *(const u32x4*)(p.res + ((size_t)mr * p.ldr + n) * 2);
        float rf[8];
        unpack_bf16x8(rr, rf);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rf[e];
      }
    }
    if (p.res2) {
      const u32x4 rr = *(const u32x4*)(p.res2 + ((size_t)m * p.ldr2 + n) * 2);
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
    if (flags & V3A_GEMM_OUT_F32) {
      float* cp = (float*)p.C + mo * p.ldc + n;
      f32x4 o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o0[e] = v[e]; o1[e] = v[4 + e]; }
      *(f32x4*)cp = o0;
      *(f32x4*)(cp + 4) = o1;
    } else {
      *(u32x4*)(p.C + (mo * p.ldc + n) * 2) = pack_bf16x8(v);
    }
  }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this group's reads returned before the next group overwrites the region
    __builtin_amdgcn_wave_barrier();
  }
}

typedef void (*gemm_fn)(const GemmP);
struct TileEntry {
  const char* name;
  int BM, BN, nthr, lds;
  gemm_fn fn, conv_fn;
};

template <int BM, int BN, int WM, int WN, int BK, int NS, int STG>
constexpr gemm_fn conv_kernel_or_null() {
  if constexpr (TileCfg<BM, BN, WM, WN, BK, NS>::CONV_OK) return (gemm_fn)gemm_nt_kernel<BM, BN, WM, WN, BK, NS, true, STG>;
  else return nullptr;
}
#define TILE_ENTRY_S(BM, BN, WM, WN, BK, NS, STG)                                                   \
  { #BM "x" #BN "_w" #WM "x" #WN "_k" #BK "s" #NS "_stg" #STG, BM, BN, TileCfg<BM, BN, WM, WN, BK, NS>::NTHR, \
    TileCfg<BM, BN, WM, WN, BK, NS>::LDS_BYTES, (gemm_fn)gemm_nt_kernel<BM, BN, WM, WN, BK, NS, false, STG>, \
    conv_kernel_or_null<BM, BN, WM, WN, BK, NS, STG>() }
#define TILE_ENTRY(BM, BN, WM, WN, BK, NS) TILE_ENTRY_S(BM, BN, WM, WN, BK, NS, 0)
#define TILE_ENTRY_OCC(BM, BN, WM, WN, BK, NS, OCC)                                                 \
  { #BM "x" #BN "_w" #WM "x" #WN "_k" #BK "s" #NS "_occ" #OCC, BM, BN, TileCfg<BM, BN, WM, WN, BK, NS>::NTHR, \
    TileCfg<BM, BN, WM, WN, BK, NS>::LDS_BYTES, (gemm_fn)gemm_nt_kernel<BM, BN, WM, WN, BK, NS, false, 0, OCC>, nullptr }

const TileEntry kTiles[] = {
    TILE_ENTRY(256, 192, 4, 2, 64, 2),  // 0: N % 192 == 0 shapes (d=1536): 8192x1536 -> exactly 256 tiles
    TILE_ENTRY(192, 256, 2, 4, 64, 2),  // 1: transposed role of 0 (V^T = Wv . X^T)
    TILE_ENTRY(256, 256, 2, 4, 64, 2),  // 2
    TILE_ENTRY(128, 256, 2, 4, 64, 2),  // 3
    TILE_ENTRY(256, 128, 4, 2, 64, 2),  // 4
    TILE_ENTRY(128, 128, 2, 2, 64, 2),  // 5: small / ragged problems, 2 workgroups per CU
    // two (or four) co-resident workgroups per CU: one's epilogue / DMA-issue bubbles are filled by the others' MFMAs
    TILE_ENTRY_OCC(256, 192, 4, 2, 32, 2, 2),  // 6: 56 KiB ring; +14 % on shapes with >= 2 tiles per CU (QK, FFN1)
    TILE_ENTRY_OCC(128, 192, 2, 2, 32, 2, 4),  // 7: 40 KiB, 4 waves, up to 4 per CU
    TILE_ENTRY_OCC(128, 192, 2, 2, 64, 2, 2),  // 8: 80 KiB, 4 waves, 2 per CU
    TILE_ENTRY_OCC(192, 128, 2, 2, 32, 2, 4),  // 9
    TILE_ENTRY_OCC(192, 256, 2, 4, 32, 2, 2),  // 10: transposed role of 6
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);
int g_attr_lds[kNumTiles][2] = {};

constexpr int kAutoTiles = 7;  // tiles the heuristic may choose from (the rest are explicit / tuning variants)
int pick_tile(int M, int N, bool conv = false) {
  // minimise (#rounds over 256 CUs) x (tile area incl. padding waste); prefer bigger tiles on ties.
  double best = 1e30;
  int bi = 5;
  for (int i = 0; i < kAutoTiles; ++i) {
    const TileEntry& e = kTiles[i];
    if (conv && !e.conv_fn) continue;
    long tm = (M + e.BM - 1) / e.BM, tn = (N + e.BN - 1) / e.BN;
    long tiles = tm * tn;
    int per_cu = (e.lds <= 80 * 1024) ? 2 : 1;
    long slots = 256L * per_cu;
    long rounds = (tiles + slots - 1) / slots;
    // co-resident small tiles run ~concurrently: cost per round ~ per_cu tiles' area, small efficiency bonus for large tiles
    double eff = (e.BM * e.BN >= 256 * 192) ? 1.0 : (e.BM * e.BN >= 128 * 256 ? 0.9 : 0.8);
    if (i == 6) eff = 1.06;  // 256x192 two-per-CU: measured +5..14 % over tile 0 once every CU holds two workgroups
    double cost = (double)rounds * per_cu * e.BM * e.BN / eff;
    if (cost < best - 1e-9) { best = cost; bi = i; }
  }
  return bi;
}

int launch(const GemmP& p, int ti, bool conv, void* stream) {
  if (ti < 0 || ti >= kNumTiles) ti = pick_tile(p.M, p.N, conv);
  const TileEntry& e = kTiles[ti];
  const gemm_fn fn = conv ? e.conv_fn : e.fn;
  if (!fn) return V3A_ERR_ARG;  // this tile shape has no conv instantiation
  const int lds = e.lds + (conv ? p.K / 8 * 4 : 0);
  if (lds > 160 * 1024) return V3A_ERR_SHAPE;
  if (g_attr_lds[ti][conv] < lds) {
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return V3A_ERR_LAUNCH;
    g_attr_lds[ti][conv] = lds;
  }
  const long tiles = (long)((p.M + e.BM - 1) / e.BM) * ((p.N + e.BN - 1) / e.BN);
  hipLaunchKernelGGL(fn, dim3((unsigned)tiles), dim3(e.nthr), lds, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

}  // namespace

extern "C" int v3a_gemm_num_tiles(void) { return kNumTiles; }
extern "C" int v3a_gemm_pick_tile(int M, int N) { return (M > 0 && N > 0) ? pick_tile(M, N) : V3A_ERR_SHAPE; }
extern "C" const char* v3a_gemm_tile_name(int t) { return (t >= 0 && t < kNumTiles) ? kTiles[t].name : ""; }

extern "C" int v3a_gemm_bf16_nt(const v3a_gemm_args* a, void* stream) {
  if (!a || !a->A || !a->B || !a->C) return V3A_ERR_ARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return V3A_ERR_SHAPE;
  if (a->K % 64 || a->lda % 8 || a->ldb % 8 || a->ldc % 8 || a->N % 8) return V3A_ERR_SHAPE;
  if (a->residual && (a->ldr % 8)) return V3A_ERR_SHAPE;
  if ((a->flags & V3A_GEMM_SCALE_PER_BATCH) && a->scale && a->rows_per_batch <= 0) return V3A_ERR_ARG;
  if (a->flags & V3A_GEMM_NO_ROUND_ACC) return V3A_ERR_ARG;  // not implemented: accumulators are parked as bf16
  GemmP p = {};
  p.A = (const char*)a->A; p.B = (const char*)a->B; p.C = (char*)a->C;
  p.bias = a->bias; p.res = (const char*)a->residual; p.scale = a->scale;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldr = a->ldr;
  p.rpb = a->rows_per_batch > 0 ? a->rows_per_batch : 1; p.sstride = a->scale_stride;
  p.act = a->act; p.flags = a->flags;
  { static const bool rowmajor = getenv("V3A_GEMM_ROWMAJOR") != nullptr; if (rowmajor) p.flags |= 1 << 27; }  // A/B switch
  p.res2 = (const char*)a->residual2; p.ldr2 = a->ldr2; p.res_mod = a->res_row_mod;
  p.orow_group = a->out_row_group; p.orow_skip = a->out_row_skip; p.orow_off = a->out_row_off;
  if (a->residual2 && (a->ldr2 % 8)) return V3A_ERR_SHAPE;
  return launch(p, a->tile, false, stream);
}

extern "C" int v3a_conv_bf16(const v3a_conv_args* a, void* stream) {
  if (!a || !a->x || !a->w || !a->y || !a->ktab) return V3A_ERR_ARG;
  if (a->T <= 0 || a->H <= 0 || a->W <= 0 || a->Cin <= 0 || a->oT <= 0 || a->oH <= 0 || a->oW <= 0 || a->Cout <= 0)
    return V3A_ERR_SHAPE;
  if (a->Cin % 8 || a->Cout % 8 || a->Kpad % 64 || a->ldy % 8 || a->Kpad / 8 > 4096) return V3A_ERR_SHAPE;
  if (a->residual && (a->ldr % 8)) return V3A_ERR_SHAPE;
  if ((long)a->oT * a->oH * a->oW > 0x7fffffffL) return V3A_ERR_SHAPE;
  if (a->flags & (V3A_GEMM_BIAS_ROW | V3A_GEMM_NO_ROUND_ACC)) return V3A_ERR_ARG;
  GemmP p = {};
  p.A = (const char*)a->x; p.B = (const char*)a->w; p.C = (char*)a->y;
  p.bias = a->bias; p.res = (const char*)a->residual; p.scale = a->scale;
  p.M = a->oT * a->oH * a->oW; p.N = a->Cout; p.K = a->Kpad;
  p.lda = 0; p.ldb = a->Kpad; p.ldc = a->ldy; p.ldr = a->ldr;
  p.rpb = 1; p.sstride = 0;
  p.act = a->act; p.flags = a->flags & ~V3A_GEMM_SCALE_PER_BATCH;
  { static const bool rowmajor = getenv("V3A_GEMM_ROWMAJOR") != nullptr; if (rowmajor) p.flags |= 1 << 27; }  // A/B switch
  p.ktab = a->ktab;
  p.cT = a->T; p.cH = a->H; p.cW = a->W; p.cCin = a->Cin;
  p.oH = a->oH; p.oW = a->oW;
  p.sT = a->sT; p.sH = a->sH; p.sW = a->sW;
  p.pT = a->pT; p.pH = a->pH; p.pW = a->pW;
  p.ups = a->ups2 ? 1 : 0; p.replicate = a->replicate ? 1 : 0;
  p.res2 = (const char*)a->residual2; p.ldr2 = a->ldr2; p.res_mod = a->res_row_mod;
  p.orow_group = a->out_row_group; p.orow_skip = a->out_row_skip; p.orow_off = a->out_row_off;
  if (a->residual2 && (a->ldr2 % 8)) return V3A_ERR_SHAPE;
  return launch(p, a->tile, true, stream);
}
