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

#include "gemm_epilogue.h"
#include <type_traits>

namespace {

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
  static constexpr int EPI_F32_BYTES = NW * 32 * (WTN * 4 + 16);   // split-bf16 convolution: the same park in fp32
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
// CONV = 0: plain GEMM; 1: implicit-GEMM convolution (A gathered through the K-chunk table); 2: the same with fp32-equivalent
//        arithmetic (v3a_conv_split): A is a (hi, lo) bf16 pair of planes, bit 28 of a table entry selects the plane, the weight
//        matrix carries the three partial products side by side along K, and the epilogue stays in fp32 (gemm_epilogue_f32).
template <int BM, int BN, int WM, int WN, int BK, int NS, int CONV, int STG>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN + 3) / 4) void gemm_nt_kernel(const GemmP pin) {
  GemmP p = pin;
  if (gridDim.y > 1) { p.A += blockIdx.y * p.az; p.B += blockIdx.y * p.bz; p.C += blockIdx.y * p.cz; if (p.res) p.res += blockIdx.y * p.rz; if (p.rowsq) p.rowsq += (size_t)blockIdx.y * p.M * (p.N / 32); }
  using T = TileCfg<BM, BN, WM, WN, BK, NS>;
  constexpr int NW = T::NW, MT = T::MT, NTL = T::NTL, NL = T::NL, STAGE = T::STAGE;
  constexpr int WTM = T::WTM, WTN = T::WTN, PITCH = T::PITCH;
  constexpr int RB = T::RB, CPR = T::CPR, RPI = T::RPI, NINS = T::NINS, KSTEPS = T::KSTEPS;
  static_assert(!CONV || T::CONV_OK, "conv needs an even A/B instruction split");
  static_assert(CONV != 2 || T::EPI_F32_BYTES <= T::LDS_BYTES, "fp32 park must fit the tile ring");
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
  constexpr int GM = (32 * BN / BM >= 36) ? 8 : 4;  // ~sqrt(tiles in flight per XCD x BN/BM): squarest in-flight patch
  int tm, tn;
  {
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
          const char* src = ((CONV == 2 && (e & (1 << 28))) ? p.A_lo : p.A) + (((size_t)(tt * p.cH + hh) * p.cW + ww) * p.cCin + (e & 0xffff)) * 2;
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
          const char* src = ((CONV == 2 && (e & (1 << 28))) ? p.A_lo : p.A) + (((size_t)(tt * p.cH + hh) * p.cW + ww) * p.cCin + (e & 0xffff)) * 2;
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

  gemm_add_bias<MT, NTL>(p, acc, lane, m0 + wm * WTM, n0 + wn * WTN);
  if constexpr (!CONV) gemm_row_sumsq<MT, NTL>(p, acc, lane, m0 + wm * WTM, n0 + wn * WTN);
  if constexpr (CONV == 2) gemm_epilogue_f32<MT, NTL>(p, acc, smem, wave, lane, m0 + wm * WTM, n0 + wn * WTN);
  else gemm_epilogue<MT, NTL, (NTL < 3 ? 2 : NTL), (MT * NTL < 8)>(p, acc, smem, wave, lane, m0 + wm * WTM, n0 + wn * WTN);
}

// =====================================================================================================================
// Ping-pong ("8-phase") main loop: 8 waves, one workgroup per CU, 256 x (64*NP) output tile (or its transpose), BK = 64.
//
// The tile's operands are an R side (256 rows: 4 waves x 64 rows, fragments RESIDENT in registers for a whole K tile) and an
// S side (64*NP rows: 2 waves x NP blocks of 32 rows, STREAMED one block per phase).  A K tile is NP phases; in phase c a wave
// multiplies its two resident R blocks with S block c: 2 x 4 k-steps = 8 v_mfma_f32_32x32x16_bf16 (256 matrix-pipe cycles).
// The two waves that share a SIMD (wave w and w+4) sit in different groups that order the SAME phase differently, with one
// s_barrier per phase (interval k = the time between barriers k-1 and k):
//      G0, interval k : MFMA(k)  (fragments read in interval k-1)   -> read fragments(k+1) -> issue DMA chunk -> counted wait
//      G1, interval k : read fragments(k) -> issue DMA chunk -> lgkmcnt(0) -> MFMA(k) -> counted wait
// so on every SIMD one wave's 8 MFMAs run while its partner reads LDS and issues the LDS-DMA refill, and the matrix pipe is
// handed from G0 to G1 inside the interval without a barrier in between (the previous two-barriers-per-phase form measured
// 1.87 PF with neither reads nor DMA: each hand-over through a barrier idles the pipe ~70 cycles).
// K tiles live in two LDS buffers.  Per wave a tile is J = (256 + 64 NP)/64 DMA instructions of 8 rows x 128 B, issued as a
// stream ordered by first use  [S0, R0..R3, S1, .., S(NP-1)]  cut into NP chunks; chunk q is issued in interval q - LEAD and
// stays in flight across barriers (counted s_waitcnt vmcnt, raw s_barrier; never drained in the steady state).
//   RAW: the data of phase p is read by G0 in interval p-1 and by G1 in interval p; every wave waits for ITS share of it inside
//        interval p-2, and barrier p-2 orders all shares before the first read.
//   WAR: a block last read in phase m has been read (and the read retired by lgkmcnt(0)) by G0 before its MFMA(m) and by G1
//        before its MFMA(m), both inside interval m, so its LDS may be refilled from interval m+1 on.  R is read in phase 0,
//        S_c in phase c; the refill of tile t+2 into tile t's buffer honours  issue interval >= m + 1  iff  LEAD <= 2 NP - 1.
// Accumulation order per output element is the same ascending-k chain of 32x32x16 MFMAs as gemm_nt_kernel: bit-identical results.
template <int NP>
struct PPCfg {
  static constexpr int RROWS = 256, SROWS = 64 * NP, RB = 128;
  static constexpr int STAGE = (RROWS + SROWS) * RB;
  static constexpr int J = (RROWS + SROWS) / 64;                  // DMA instructions per wave per K tile (7 / 8)
  static constexpr int cnt(int c) { return NP == 3 ? (c == 0 ? 3 : 2) : 2; }   // instructions in chunk c
  static constexpr int cum(int k) { int s = 0; for (int i = 0; i < k; ++i) s += cnt(i); return s; }
  static constexpr int need(int c) { return 5 + c; }              // leading instructions of a tile that phase c reads
  // vmcnt that guarantees the data of phase c of a tile with `rem` tiles left (incl. itself; >= 3 = steady state) has landed,
  // when waited two intervals ahead, i.e. right after chunk (phase - 2 + LEAD) was issued
  static constexpr int allowed(int c, int rem, int LEAD) {
    const int x = c - 2 + LEAD;
    const int steady = J * (x / NP) + cum(x % NP + 1) - need(c);
    const int tail = J * rem - need(c);
    return (rem >= 3 || steady < tail) ? steady : tail;
  }
  static constexpr int EPI_BYTES = 8 * 32 * (128 * 2 + 8);
  static constexpr int AUX = 2 * STAGE;              // 1 KiB bias slice + 1 KiB gate-scale slice of the tile, behind the ring
  static constexpr int LDS_BYTES = 2 * STAGE + 2048;
  static_assert(J == cum(NP), "chunking");
  static_assert(LDS_BYTES >= EPI_BYTES, "epilogue parks in the ring");
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// (A persistent variant - min(#tiles, #CUs) workgroups walking the tiles with the next tile's first K tile prefetched under the
// epilogue - measured 2-5 % SLOWER than one workgroup per tile: it keeps all CUs' store bursts in lockstep and costs 50 VGPRs.)
//
// ABL = -DV3A_GEMM_ABL=<mask> development builds only (every ping-pong tile is ablated, results are garbage; the shipped library has
// ABL = 0 and none of this code): bit 0 = no LDS-DMA in the K loop, bit 1 = no fragment reads, bit 2 = no MFMAs, bit 3 = no s_setprio,
// bits 4-7 = the epilogue's DBG mask.
#ifndef V3A_GEMM_ABL
#define V3A_GEMM_ABL 0
#endif
constexpr int ABL = V3A_GEMM_ABL;

// TT = true (NP = 3, RA = true only): tiles whose columns lie at or beyond p.tcol0 are stored transposed into p.Ct (gemm_store_transposed) - a
// separate instantiation so that the dominant symbol's code is untouched.
template <int NP, bool RA, int LEAD, bool F8 = false, bool TT = false>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const GemmP pin) {
  GemmP p = pin;
  if (gridDim.y > 1) { p.A += blockIdx.y * p.az; p.B += blockIdx.y * p.bz; p.C += blockIdx.y * p.cz; if (p.res) p.res += blockIdx.y * p.rz; if (p.rowsq) p.rowsq += (size_t)blockIdx.y * p.M * (p.N / 32); }
  using T = PPCfg<NP>;
  constexpr int RB = T::RB, STAGE = T::STAGE, J = T::J;
  constexpr int BM = RA ? 256 : 64 * NP, BN = RA ? 64 * NP : 256;
  constexpr int MT = RA ? 2 : NP, NTL = RA ? NP : 2;
  static_assert(LEAD >= 2 && LEAD <= 2 * NP - 1, "refill would overtake the readers of the buffer");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;          // waves w and w+4 share a SIMD: opposite groups
  const int wr = wave & 3, ws = wave >> 2;
  const int hi = lane >> 5, l31 = lane & 31;

  const int tilesN = (p.N + BN - 1) / BN, tilesM = (p.M + BM - 1) / BM;
  const int ntiles = tilesM * tilesN;
  // R / S side views of the two operands
  const char* Rp = RA ? p.A : p.B;
  const char* Sp = RA ? p.B : p.A;
  const int ldR = RA ? p.lda : p.ldb, ldS = RA ? p.ldb : p.lda;
  const int Rn = RA ? p.M : p.N, Sn = RA ? p.N : p.M;

  // ---- per-lane DMA sources in stream order; LDS image row-major 128-B rows, chunk' = chunk ^ ((row >> 1) & 7) ----
  const char* gp[J];
  int ldst[J];  // wave-uniform LDS byte offset inside a stage
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const bool isR = (j >= 1 && j <= 4);
    const int row = isR ? ((j - 1) * 8 + wave) * 8 : (wave >> 2) * (32 * NP) + (j == 0 ? 0 : j - 4) * 32 + (wave & 3) * 8;
    ldst[j] = (isR ? 0 : T::RROWS * RB) + row * RB;
  }
  // output tile `id` -> its origin; DMA sources rewound to K = 0
  auto setup = [&](int id, int& m0, int& n0) {
    const int tl = xcd_remap(id, ntiles);
    constexpr int GM = RA ? 4 : 8;    // grouped raster: near-square patch of tiles in flight per XCD
    const int gsz = GM * tilesN, gid = tl / gsz, first = gid * GM;
    const int gm = min(tilesM - first, GM), r = tl - gid * gsz;
    m0 = (first + r % gm) * BM; n0 = (r / gm) * BN;
    const int r0 = RA ? m0 : n0, s0 = RA ? n0 : m0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const bool isR = (j >= 1 && j <= 4);
      const int row = isR ? ((j - 1) * 8 + wave) * 8 : (wave >> 2) * (32 * NP) + (j == 0 ? 0 : j - 4) * 32 + (wave & 3) * 8;
      const int rl = row + (lane >> 3);
      const int c = (lane & 7) ^ ((rl >> 1) & 7);
      int grow = (isR ? r0 : s0) + rl;
      const int lim = isR ? Rn : Sn;
      grow = grow < lim ? grow : lim - 1;
      gp[j] = (isR ? Rp : Sp) + ((size_t)grow * (isR ? ldR : ldS)) * 2 + c * 16;
    }
  };
  // chunk cc of the next not-yet-issued K tile -> stage `buf`
  auto issue = [&](auto cc_tag, int buf) {
    constexpr int cc = decltype(cc_tag)::value;
#pragma unroll
    for (int j = T::cum(cc); j < T::cum(cc + 1); ++j) {
      glds16(gp[j], smem + buf * STAGE + ldst[j]);
      gp[j] += RB;
    }
  };
  const int nk = p.K / 64;
  // prologue chunks [Q0, Q1) of an output tile (chunk q belongs to K tile q / NP)
  auto prologue = [&](auto q0_tag, auto q1_tag) {
    constexpr int Q0 = decltype(q0_tag)::value, Q1 = decltype(q1_tag)::value;
    auto pro = [&](auto q_tag) {
      constexpr int q = decltype(q_tag)::value;
      if constexpr (q >= Q0 && q < Q1) {
        if (q / NP < nk) issue(std::integral_constant<int, q % NP>{}, (q / NP) & 1);
      }
    };
    pro(std::integral_constant<int, 0>{}); pro(std::integral_constant<int, 1>{}); pro(std::integral_constant<int, 2>{});
    pro(std::integral_constant<int, 3>{}); pro(std::integral_constant<int, 4>{}); pro(std::integral_constant<int, 5>{});
    pro(std::integral_constant<int, 6>{});
  };

  const int sw = (l31 >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = l31 * RB + (((2 * ks + hi) ^ sw) << 4);
  const int roff = (wr * 64) * RB, soff = (T::RROWS + ws * 32 * NP) * RB;

  f32x16 acc[MT][NTL];
  bf16x8 rf[2][4] = {}, sf[4] = {};
  i32x8 rf8[2][2] = {}, sf8[2] = {};   // F8 form of the same fragments
  // counted wait for the data of phase nc of a tile with nrem tiles left (including itself); exact in the tail, where fewer
  // instructions than the steady-state count are outstanding behind the needed ones
  auto wait_phase = [&](auto nc_tag, int nrem) {
    constexpr int nc = decltype(nc_tag)::value;
    constexpr int a3 = T::allowed(nc, 3, LEAD), a2 = T::allowed(nc, 2, LEAD), a1 = T::allowed(nc, 1, LEAD);
    if (nrem >= 3 || (a2 == a3 && nrem == 2)) wait_vmcnt<a3>();
    else if (nrem == 2) wait_vmcnt<a2>();
    else if (nrem == 1) wait_vmcnt<a1>();
  };
  // fragments of phase c of the tile in stage BUF
  auto read_frags = [&](auto buf_tag, auto c_tag) {
    constexpr int BUF = decltype(buf_tag)::value, c = decltype(c_tag)::value;
    if constexpr (ABL & 2) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 x0 = sf[ks], x1 = rf[0][ks], x2 = rf[1][ks];
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2));
        sf[ks] = x0; rf[0][ks] = x1; rf[1][ks] = x2;
      }
    } else {
      const char* sR = smem + BUF * STAGE + roff;
      const char* sS = smem + BUF * STAGE + soff;
      if constexpr (F8) {   // the same eight 16-byte reads, landing pairwise in the 8-VGPR operands of the f8f6f4 MFMA
        if constexpr (c == 0) {
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
              rf8[b][s2] = frag8(*(const u32x4*)(sR + b * 32 * RB + koff[2 * s2]), *(const u32x4*)(sR + b * 32 * RB + koff[2 * s2 + 1]));
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          sf8[s2] = frag8(*(const u32x4*)(sS + c * 32 * RB + koff[2 * s2]), *(const u32x4*)(sS + c * 32 * RB + koff[2 * s2 + 1]));
      } else if constexpr (c == 0) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) rf[b][ks] = *(const bf16x8*)(sR + b * 32 * RB + koff[ks]);
      }
      if constexpr (!F8) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) sf[ks] = *(const bf16x8*)(sS + c * 32 * RB + koff[ks]);
      }
    }
  };
  auto mfma_phase = [&](auto c_tag) {
    constexpr int c = decltype(c_tag)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(ABL & 8)) __builtin_amdgcn_s_setprio(1);
    if constexpr (F8) {
      // e4m3 operands: the same 128-byte rows now hold 128 k values; one v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales) eats
      // two of the bf16 loop's 16-byte fragments per lane.  A and B agree on which k each (lane half, byte) carries, which is all
      // the instruction needs, so the LDS image, swizzle and fragment reads are those of the bf16 loop.
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if constexpr (RA) acc[b][c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(sf8[s2], rf8[b][s2], acc[b][c], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
          else acc[c][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(rf8[b][s2], sf8[s2], acc[c][b], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
      // the scaled-MFMA intrinsic is a pure call to the optimiser, which otherwise sinks whole phases of them to the next use of the
      // accumulator (several barriers later) and keeps every fragment alive until then: pin the results here
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if constexpr (RA) asm volatile("" : "+v"(acc[b][c]));
        else asm volatile("" : "+v"(acc[c][b]));
      }
    } else if constexpr (!(ABL & 4)) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if constexpr (RA) acc[b][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sf[ks], rf[b][ks], acc[b][c], 0, 0, 0);
          else acc[c][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rf[b][ks], sf[ks], acc[c][b], 0, 0, 0);
        }
    }
    if constexpr (!(ABL & 8)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // refill (chunk interval + LEAD), the counted wait for the data of phase interval + 2, then the interval's barrier
  auto refill_wait_barrier = [&](auto buf_tag, auto c_tag, const int rem) {
    constexpr int BUF = decltype(buf_tag)::value, c = decltype(c_tag)::value;
    if (!(ABL & 1) && (c + LEAD) / NP < rem) issue(std::integral_constant<int, (c + LEAD) % NP>{}, BUF ^ (((c + LEAD) / NP) & 1));
    constexpr int nc = (c + 2) % NP;
    wait_phase(std::integral_constant<int, nc>{}, c + 2 < NP ? rem : rem - 1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  int m0, n0;
  setup(blockIdx.x, m0, n0);
  // ---- bias / gate-scale slices of this tile -> LDS (oldest DMAs of the wave: every later counted wait covers them) ----
  EpiAux aux;
  aux.m0 = m0; aux.n0 = n0;
  {
    const bool brow = (p.flags & V3A_GEMM_BIAS_ROW) != 0;
    const int blim = brow ? p.M : p.N;
    if (p.bias && blim % 4 == 0) {   // (a ragged BIAS_ROW extent keeps the global loads: a clamped 16-B piece would shift it)
      aux.lds_bias = smem + T::AUX;
      // 256 floats cover a tile edge; lanes clamped at the end of the vector hold columns / rows that are never stored
      if (wave == 0) glds16(p.bias + min((brow ? m0 : n0) + lane * 4, blim - 4), smem + T::AUX);
    }
    if (p.scale) {
      const int b0 = (p.flags & V3A_GEMM_SCALE_PER_BATCH) ? m0 / p.rpb : 0;
      const int b1 = (p.flags & V3A_GEMM_SCALE_PER_BATCH) ? min(m0 + BM - 1, p.M - 1) / p.rpb : 0;
      if (b0 == b1) {   // the tile lies inside one batch item (always, when rows_per_batch % BM == 0): one 1-KiB slice
        aux.lds_scale = smem + T::AUX + 1024;
        if (wave == 1) glds16(p.scale + (size_t)b0 * p.sstride + min(n0 + lane * 4, p.N - 4), smem + T::AUX + 1024);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  {
    // ---- prologue: chunks 0 .. LEAD-1, then the data of phases 0 and 1 ----
    prologue(I0{}, std::integral_constant<int, LEAD>{});
    wait_phase(I1{}, nk);   // interval "-1": chunk LEAD-1 was the last one issued
    __builtin_amdgcn_s_barrier();
    if (grp == 0) {
      // ---------------- group 0: MFMA first, then next phase's fragments + refill ----------------
      auto tile = [&](auto buf_tag, const int rem) {
        constexpr int BUF = decltype(buf_tag)::value;
        auto phase = [&](auto c_tag) {
          constexpr int c = decltype(c_tag)::value;
          mfma_phase(c_tag);
          if constexpr (c + 1 < NP) read_frags(buf_tag, std::integral_constant<int, c + 1>{});
          else if (rem > 1) read_frags(std::integral_constant<int, BUF ^ 1>{}, I0{});
          refill_wait_barrier(buf_tag, c_tag, rem);
        };
        phase(std::integral_constant<int, 0>{});
        phase(std::integral_constant<int, 1>{});
        phase(std::integral_constant<int, 2>{});
        if constexpr (NP > 3) phase(std::integral_constant<int, 3>{});
      };
      read_frags(I0{}, I0{});
      int t = 0;
      for (; t + 1 < nk; t += 2) {
        tile(I0{}, nk - t);
        tile(I1{}, nk - t - 1);
      }
      if (t < nk) tile(I0{}, 1);
    } else {
      // ---------------- group 1: fragments + refill first, then MFMA ----------------
      auto tile = [&](auto buf_tag, const int rem) {
        auto phase = [&](auto c_tag) {
          read_frags(buf_tag, c_tag);
          constexpr int BUF = decltype(buf_tag)::value, c = decltype(c_tag)::value;
          if (!(ABL & 1) && (c + LEAD) / NP < rem) issue(std::integral_constant<int, (c + LEAD) % NP>{}, BUF ^ (((c + LEAD) / NP) & 1));
          __builtin_amdgcn_sched_barrier(0);
          mfma_phase(c_tag);
          constexpr int nc = (c + 2) % NP;
          wait_phase(std::integral_constant<int, nc>{}, c + 2 < NP ? rem : rem - 1);
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
        };
        phase(std::integral_constant<int, 0>{});
        phase(std::integral_constant<int, 1>{});
        phase(std::integral_constant<int, 2>{});
        if constexpr (NP > 3) phase(std::integral_constant<int, 3>{});
      };
      int t = 0;
      for (; t + 1 < nk; t += 2) {
        tile(I0{}, nk - t);
        tile(I1{}, nk - t - 1);
      }
      if (t < nk) tile(I0{}, 1);
    }
    // every fragment read has returned (lgkmcnt(0) precedes the last MFMAs) and no DMA is in flight (the tail waits reach 0):
    // both stages are free
    const int em = m0 + (RA ? wr * 64 : ws * 32 * NP), en = n0 + (RA ? ws * 32 * NP : wr * 64);
    if constexpr (F8) gemm_dequant<MT, NTL>(p, acc, lane, em, en);
    if constexpr (!(ABL & 32) && !(ABL & 128)) gemm_add_bias<MT, NTL>(p, acc, lane, em, en, aux);
    if constexpr (TT) {
      static_assert(RA && NP == 3 && !F8, "transposed tail: the 256 x 192 bf16 tile");
      if (p.Ct && n0 >= p.tcol0) {   // (tcol0 % 192 == 0: a tile is all normal or all transposed)
        gemm_store_transposed<MT, NTL>(p, acc, smem, wave, lane, em, en);
        return;
      }
    }
    gemm_row_sumsq<MT, NTL>(p, acc, lane, em, en);
    gemm_epilogue<MT, NTL, (NTL < 3 ? 2 : NTL), (MT * NTL < 8), (ABL >> 4) & 15>(p, acc, smem, wave, lane, em, en, aux);
  }
}


// =====================================================================================================================
// One wave per SIMD ("w4"): 4 waves x 512 registers, 256 x 192 tile, BK = 64, hand-scheduled K loop (gemm_w4_loop.inc, generated and
// documented by tools/gen_gemm_w4.py).  The structure of the vendor library's assembly kernels: a wave owns 128 x 96 outputs (192
// accumulators in AGPRs), so a 16-byte fragment feeds 3-4 MFMAs instead of 2-3 (112 instead of 160 LDS reads per K tile and CU), and the
// LDS-DMA pieces / fragment reads are placed one by one between the MFMAs of the single instruction stream each SIMD runs.
// Same LDS image (128-byte rows, chunk ^ ((row >> 1) & 7)), same D^T accumulator orientation, same ascending-k MFMA chain per output
// element as every other tile: bit-identical results.  Stage 1 sits 64 KiB above stage 0 so that one xor moves an address between them.
struct W4Cfg {
  static constexpr int BM = 256, BN = 192, STAGE = (BM + BN) * 128, STAGE1 = 65536;
  static constexpr int AUX = STAGE1 + STAGE;            // 1 KiB bias + 1 KiB gate-scale slice of the tile (as in the ping-pong kernel)
  static constexpr int LDS_BYTES = AUX + 2048;
  static_assert(4 * 32 * (96 * 2 + 8) <= STAGE1, "epilogue park fits stage 0");
};
#define W4_A8(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w4_kernel(const GemmP pin) {
  GemmP p = pin;
  if (gridDim.y > 1) { p.A += blockIdx.y * p.az; p.B += blockIdx.y * p.bz; p.C += blockIdx.y * p.cz; if (p.res) p.res += blockIdx.y * p.rz; if (p.rowsq) p.rowsq += (size_t)blockIdx.y * p.M * (p.N / 32); }
  using T = W4Cfg;
  constexpr int BM = T::BM, BN = T::BN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;
  const int tilesN = (p.N + BN - 1) / BN, tilesM = (p.M + BM - 1) / BM;
  int m0, n0;
  {
    const int tl = xcd_remap(blockIdx.x, tilesM * tilesN);
    constexpr int GM = 4;    // grouped raster: near-square patch of tiles in flight per XCD (as gemm_pp_kernel)
    const int gsz = GM * tilesN, gid = tl / gsz, first = gid * GM;
    const int gm = min(tilesM - first, GM), r = tl - gid * gsz;
    m0 = (first + r % gm) * BM; n0 = (r / gm) * BN;
  }
  EpiAux aux;
  aux.m0 = m0; aux.n0 = n0;
  {
    const bool brow = (p.flags & V3A_GEMM_BIAS_ROW) != 0;
    const int blim = brow ? p.M : p.N;
    if (p.bias && blim % 4 == 0) {
      aux.lds_bias = smem + T::AUX;
      if (wave == 0) glds16(p.bias + min((brow ? m0 : n0) + lane * 4, blim - 4), smem + T::AUX);
    }
    if (p.scale) {
      const int b0 = (p.flags & V3A_GEMM_SCALE_PER_BATCH) ? m0 / p.rpb : 0;
      const int b1 = (p.flags & V3A_GEMM_SCALE_PER_BATCH) ? min(m0 + BM - 1, p.M - 1) / p.rpb : 0;
      if (b0 == b1) {
        aux.lds_scale = smem + T::AUX + 1024;
        if (wave == 1) glds16(p.scale + (size_t)b0 * p.sstride + min(n0 + lane * 4, p.N - 4), smem + T::AUX + 1024);
      }
    }
  }
  f32x16 acc[4][3];
  {
    const unsigned lds0 = (unsigned)(size_t)(LDS_AS char*)smem;
    const int sw = (l31 >> 1) & 7;
    const unsigned vrowA = m0 + 8 * wave + (lane >> 3), vrowB = n0 + 8 * wave + (lane >> 3);
    const unsigned vchunk = ((lane & 7) ^ ((4 * wave + (lane >> 4)) & 7)) << 4;
    const unsigned vk0 = lds0 + l31 * 128 + (((0 + hi) ^ sw) << 4), vk1 = lds0 + l31 * 128 + (((2 + hi) ^ sw) << 4);
    const unsigned vk2 = lds0 + l31 * 128 + (((4 + hi) ^ sw) << 4), vk3 = lds0 + l31 * 128 + (((6 + hi) ^ sw) << 4);
    const unsigned sdma = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
    const unsigned sAoff = __builtin_amdgcn_readfirstlane(wm * 16384), sBoff = __builtin_amdgcn_readfirstlane(32768 + wn * 12288);
    const int sM1 = p.M - 1, sN1 = p.N - 1, slda2 = p.lda * 2, sldb2 = p.ldb * 2, snk = p.K / 64;
    const char* sA = p.A;
    const char* sB = p.B;
#ifndef W4_LOOP_INC
#define W4_LOOP_INC "gemm_w4_loop.inc"
#endif
    asm volatile(
#include W4_LOOP_INC
        : "=&{v[0:15]}"(acc[0][0]), "=&{v[16:31]}"(acc[0][1]), "=&{v[32:47]}"(acc[0][2]), "=&{v[48:63]}"(acc[1][0]), "=&{v[64:79]}"(acc[1][1]),
          "=&{v[80:95]}"(acc[1][2]), "=&{v[96:111]}"(acc[2][0]), "=&{v[112:127]}"(acc[2][1]), "=&{v[128:143]}"(acc[2][2]),
          "=&{v[144:159]}"(acc[3][0]), "=&{v[160:175]}"(acc[3][1]), "=&{v[176:191]}"(acc[3][2])
        : "s"(sA), "s"(sB), "v"(vrowA), "v"(vrowB), "v"(vchunk), "s"(sM1), "s"(sN1), "s"(slda2), "s"(sldb2), "s"(snk), "s"(sdma), "v"(vk0), "v"(vk1),
          "v"(vk2), "v"(vk3), "s"(sAoff), "s"(sBoff)
        : "memory", "m0", "scc", "vcc", "s60", "s61", "s62", "s63", "s64", "s65", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9",
          W4_A8(1), W4_A8(2), W4_A8(3), W4_A8(4), W4_A8(5), W4_A8(6), W4_A8(7), W4_A8(8), W4_A8(9), W4_A8(10), W4_A8(11), W4_A8(12), W4_A8(13),
          W4_A8(14), W4_A8(15), W4_A8(16), W4_A8(17), W4_A8(18), "a190", "a191");
  }
  // (every fragment read returned before the loop's last barrier and no LDS-DMA is in flight: the stages are free for the epilogue's park)
  const int em = m0 + wm * 128, en = n0 + wn * 96;
  gemm_add_bias<4, 3>(p, acc, lane, em, en, aux);
  gemm_row_sumsq<4, 3>(p, acc, lane, em, en);
#ifndef W4_EPI_DBG
#define W4_EPI_DBG 0
#endif
  gemm_epilogue<4, 3, 3, false, W4_EPI_DBG>(p, acc, smem, wave, lane, em, en, aux);
}

typedef void (*gemm_fn)(const GemmP);
struct TileEntry {
  const char* name;
  int BM, BN, nthr, lds;
  gemm_fn fn, conv_fn, split_fn;
};
template <int BM, int BN, int WM, int WN, int BK, int NS, int STG>
constexpr gemm_fn conv_kernel_or_null() {
  if constexpr (TileCfg<BM, BN, WM, WN, BK, NS>::CONV_OK) return (gemm_fn)gemm_nt_kernel<BM, BN, WM, WN, BK, NS, 1, STG>;
  else return nullptr;
}
template <int BM, int BN, int WM, int WN, int BK, int NS, int STG>
constexpr gemm_fn split_kernel_or_null() {
  if constexpr (TileCfg<BM, BN, WM, WN, BK, NS>::CONV_OK) return (gemm_fn)gemm_nt_kernel<BM, BN, WM, WN, BK, NS, 2, STG>;
  else return nullptr;
}
#define TILE_ENTRY_S(BM, BN, WM, WN, BK, NS, STG)                                                   \
  { #BM "x" #BN "_w" #WM "x" #WN "_k" #BK "s" #NS "_stg" #STG, BM, BN, TileCfg<BM, BN, WM, WN, BK, NS>::NTHR, \
    TileCfg<BM, BN, WM, WN, BK, NS>::LDS_BYTES, (gemm_fn)gemm_nt_kernel<BM, BN, WM, WN, BK, NS, 0, STG>, \
    conv_kernel_or_null<BM, BN, WM, WN, BK, NS, STG>(), split_kernel_or_null<BM, BN, WM, WN, BK, NS, STG>() }
#define TILE_ENTRY(BM, BN, WM, WN, BK, NS) TILE_ENTRY_S(BM, BN, WM, WN, BK, NS, 0)
#define PP_ENTRY(NP, RA, LEAD)                                                                      \
  { "pp_np" #NP "_ra" #RA "_l" #LEAD, (RA) ? 256 : 64 * NP, (RA) ? 64 * NP : 256, 512, PPCfg<NP>::LDS_BYTES, \
    (gemm_fn)gemm_pp_kernel<NP, RA, LEAD>, nullptr, nullptr }

const TileEntry kTiles[] = {
    TILE_ENTRY(256, 192, 4, 2, 64, 2),  // 0: lockstep main loop (also the implicit-GEMM convolution): N % 192 == 0 shapes
    TILE_ENTRY(192, 256, 2, 4, 64, 2),  // 1: transposed role of 0
    TILE_ENTRY(256, 256, 2, 4, 64, 2),  // 2
    TILE_ENTRY(128, 256, 2, 4, 64, 2),  // 3
    TILE_ENTRY(256, 128, 4, 2, 64, 2),  // 4
    TILE_ENTRY(128, 128, 2, 2, 64, 2),  // 5: small / ragged problems, 2 workgroups per CU
    // ping-pong main loop (gemm_pp_kernel), one 8-wave workgroup per CU
    PP_ENTRY(3, true, 5),    // 6: 256x192 (d = 1536 = 8 x 192: 8192x1536 -> exactly 256 tiles)
    PP_ENTRY(4, true, 7),    // 7: 256x256
    PP_ENTRY(3, false, 5),   // 8: 192x256 (V^T = Wv . X^T)
    // small tiles for problems that cannot fill 256 CUs with the big ones (sequence-parallel shards: M = N_tokens / P rows)
    TILE_ENTRY(64, 128, 2, 2, 64, 2),   // 9
    TILE_ENTRY(128, 64, 2, 2, 64, 2),   // 10
    TILE_ENTRY(64, 64, 2, 2, 64, 2),    // 11
    // 12: the VAE decoder's full-resolution 96 -> 96 channel convolutions (13 x 512 x 512 pixels): no padding of N to 128; two workgroups
    //     per CU.  3.21 -> 2.84 ms per layer against 256x128 (256x96 and 2-wave forms measured 4.5 ms)
    TILE_ENTRY(128, 96, 4, 1, 64, 2),
    // 13: tile 6 with the transposed tail (v3a_gemm_args.C_t: the fused q | k | v projection of a DiT block - 512 + 256 tiles = three full rounds
    //     in ONE launch instead of a two-round and a one-round launch); identical to tile 6 when C_t is NULL
    { "pp_np3_ratrue_l5_tt", 256, 192, 512, PPCfg<3>::LDS_BYTES, (gemm_fn)gemm_pp_kernel<3, true, 5, false, true>, nullptr, nullptr },
    // 14: one wave per SIMD, hand-scheduled K loop (gemm_w4_kernel): 256x192, 4 waves x 512 registers
    { "w4_256x192", 256, 192, 256, W4Cfg::LDS_BYTES, (gemm_fn)gemm_w4_kernel, nullptr, nullptr },
    // 15-17: the small tiles with a FOUR-deep LDS ring.  A sequence-parallel shard's projection (512-2048 rows) is one short chain of 24 K slabs per
    // workgroup with few workgroups per CU: with two stages every slab exposes its LDS-DMA latency (0.6 us per slab whatever the load - a 512-row launch
    // takes as long as a 1024-row one); with slabs kt+1 .. kt+2 in flight across the barrier the chain runs at the tile's LDS / MFMA rate
    TILE_ENTRY(64, 64, 2, 2, 64, 4),    // 15
    TILE_ENTRY(128, 64, 2, 2, 64, 4),   // 16
    TILE_ENTRY(64, 128, 2, 2, 64, 4),   // 17
    // 18-19: the 128 x 256 / 256 x 128 lockstep tiles with a THREE-deep ring (147 KB: one workgroup per CU) - the Wan-14B shard projections
    // (1024 x 5120 x 5120: 160 workgroups, 80 K slabs each, 52 MB of weights cold per launch)
    TILE_ENTRY(128, 256, 2, 4, 64, 3),  // 18
    TILE_ENTRY(256, 128, 4, 2, 64, 3),  // 19
    // 20: 128 x 192, four waves of 64 x 96, three-deep ring: the Wan-14B shard projections again - 8 x 27 = 216 workgroups instead of the 160 of 128 x 256
    TILE_ENTRY(128, 192, 2, 2, 64, 3),
    TILE_ENTRY(128, 128, 2, 2, 64, 3),  // 21: 128 x 128 with a three-deep ring (96 KB: one workgroup per CU) - 2048-row shard projections (192 workgroups)
};
constexpr int kTileTT = 13;
constexpr int kTileW4 = 14;
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);
// e4m3 forms of the ping-pong tiles (K tile = 128 elements: the same bytes per row, twice the matrix rate)
#define PP8_ENTRY(NP, RA, LEAD)                                                                     \
  { "pp8_np" #NP "_ra" #RA "_l" #LEAD, (RA) ? 256 : 64 * NP, (RA) ? 64 * NP : 256, 512, PPCfg<NP>::LDS_BYTES, \
    (gemm_fn)gemm_pp_kernel<NP, RA, LEAD, true>, nullptr, nullptr }
// (the 256x256 form spills in its epilogue and measured slower on every Wan-14B / 1.3B shape: not instantiated)
const TileEntry kTilesF8[] = {PP8_ENTRY(3, true, 5), PP8_ENTRY(3, false, 5)};
constexpr int kNumTilesF8 = 2;
int g_attr_lds_f8[kNumTilesF8] = {};
int g_attr_lds[kNumTiles][3] = {};

// tiles the heuristic may choose from (the rest are explicit / tuning variants); the ping-pong tiles have no conv form
constexpr int kAutoList[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
int pick_tile(int M, int N, bool conv = false, int mult = 1, int act = 0) {
  // Launch time ~ (tiles the busiest CU runs one after another or side by side) x (tile area incl. padding waste) / (measured
  // efficiency of the tile family; fitted to tools/gemm_sweep.py over the production and the sequence-parallel shard shapes).
  // convolutions (fitted to tools/conv_sweep_scene.py): narrow outputs take the narrow tiles - no N padded to 128 / 192 -
  // and only small feature maps (the DPT pyramid's 16^2 / 32^2 levels) may use the small tiles of the cost model below
  if (conv && N <= 64 && M >= 256 * 256) return 10;
  if (conv && N <= 96 && N > 64 && M >= 256 * 256) return 12;
  double best = 1e30;
  int bi = 5;
  for (int i : kAutoList) {
    const TileEntry& e = kTiles[i];
    if (conv && (!e.conv_fn || (i >= 9 && M > 16384))) continue;
    long tm = (M + e.BM - 1) / e.BM, tn = (N + e.BN - 1) / e.BN;
    long tiles = tm * tn * mult;   // (mult: split-K slices launched side by side)
    const bool pp = i >= 6 && i <= 8;   // (13, the transposed-tail form of 6, is never an automatic choice)
    const long area = (long)e.BM * e.BN;
    int per_cu = pp ? 1 : (e.lds <= 40 * 1024 ? 4 : (e.lds <= 80 * 1024 ? 2 : 1));
    long per_busiest = (tiles + 255) / 256;                       // tiles the busiest CU gets (the dispatcher spreads workgroups)
    long rounds = (per_busiest + per_cu - 1) / per_cu;            // ... of which per_cu run side by side, sharing the CU
    long side = per_busiest < per_cu ? per_busiest : per_cu;
    double eff = area >= 256 * 192 ? 1.0 : (area >= 128 * 256 ? 0.9 : (area >= 128 * 128 ? 0.8 : (area >= 64 * 128 ? 0.5 : 0.55)));
    if (pp) eff = (i == 7) ? 1.20 : 1.14;  // ping-pong main loop: measured +8..14 % (256x192 / 192x256), more at 256x256
    double cost = (double)rounds * side * area / eff;
    // with a transcendental activation in the epilogue the 192x256 form (three 32-row groups of 64 columns per wave) measured 1.4 %
    // faster than 256x192 on FFN1 (-1.0 % on the whole DiT step); without one it is 3 % slower: break the tie by the activation
    if (i == 8 && act != V3A_ACT_NONE && act != V3A_ACT_RELU) cost *= 0.999;
    if (cost < best - 1e-9) { best = cost; bi = i; }
  }
  // A small-tile launch with at most two (64 x 64) / one (128 x 64, 64 x 128) workgroups per CU is a sequence-parallel shard's projection or
  // a similar short problem: each workgroup is one chain of K slabs over weights nobody has touched since the last forward, and with two LDS
  // stages every slab exposes its DMA latency.  The four-deep ring (tiles 15-17, bit-identical) keeps two more slabs in flight: measured inside
  // the P = 4 sharded DiT forward 9.27 -> 8.61 ms (profiles/r6/sp_tile_ab.log); back to back on hot weights it is neutral, and with more
  // workgroups per CU than its 64 / 96 KB of LDS admit it loses (2048 x 1536 x 1536: 18.2 -> 22.5 us) - hence the bound.
  if (!conv) {
    const long wgs = (long)((M + kTiles[bi].BM - 1) / kTiles[bi].BM) * ((N + kTiles[bi].BN - 1) / kTiles[bi].BN) * mult;
    if (bi == 11 && wgs <= 512) bi = 15;
    else if (bi == 10 && wgs <= 256) bi = 16;
    else if (bi == 9 && wgs <= 256) bi = 17;
    // the same for the 128 x 256 / 256 x 128 tiles at one workgroup per CU (a Wan-14B shard's 1024 x 5120 x 5120 projections: 160 workgroups of
    // 80 K slabs over 52 MB of cold weights): three-deep ring, P = 4 forward 48.8 -> 43.5 ms (profiles/r6/sp_tile_ab_14b.log)
    else if (bi == 5 && wgs <= 256) bi = 21;   // 2048-row shard projections (P = 2): 13.5 -> 12.9 ms per rank forward (profiles/r6/sp_tile_ab_p2.log)
    else if (bi == 3 && wgs <= 256) bi = 18;
    else if (bi == 4 && wgs <= 256) bi = 19;
    if (bi == 18 || bi == 19) {   // ... and the 128 x 192 form of it when that puts more CUs to work without a second round (1024 x 5120: 216 workgroups instead of 160)
      const long w20 = (long)((M + 127) / 128) * ((N + 191) / 192) * mult;
      if (w20 <= 256 && w20 > wgs) bi = 20;
    }
  }
  return bi;
}

int launch(const GemmP& p, int ti, int conv, void* stream, int nz = 1) {   // conv: 0 GEMM, 1 convolution, 2 split-bf16 convolution
  if (p.Ct) {
    if (ti >= 0 && ti != kTileTT) return V3A_ERR_ARG;   // an explicit tile must be the one that runs: the transposed tail exists on tile 13 only
    ti = kTileTT;
  } else if (ti < 0 || ti >= kNumTiles) {
    ti = pick_tile(p.M, p.N, conv != 0, nz, p.act);
    // split-bf16 convolutions walk a 3x longer K: a layer that cannot give every CU a big tile (the 16^2 / 32^2 levels of the DPT pyramid:
    // < 128 tiles of 256 x 256) is latency-bound per tile and wants MANY small co-resident tiles (tools/conv_split_sweep.py, 13 views:
    // 1024 -> 1024 stride 2 at 32^2 0.56 -> 0.40 ms, 1024 -> 256 at 32^2 0.55 -> 0.39, 256 -> 256 at 32^2 0.148 -> 0.100)
    if (conv == 2 && (long)((p.M + 255) / 256) * ((p.N + 255) / 256) < 128) ti = (p.M <= 4096 && p.N <= 256) ? 11 : 9;
  }
  const TileEntry& e = kTiles[ti];
  const gemm_fn fn = conv == 2 ? e.split_fn : conv ? e.conv_fn : e.fn;
  if (!fn) return V3A_ERR_ARG;  // this tile shape has no conv instantiation
  // the hand-scheduled loop addresses an operand as 64-bit base + 32-bit per-lane byte offset (row * ld * 2 + chunk), 24-bit multiplicands
  if (ti == kTileW4 && ((double)p.M * p.lda * 2 + 256 >= 4294967296.0 || (double)p.N * p.ldb * 2 + 256 >= 4294967296.0 || p.M >= (1 << 24) ||
                        p.N >= (1 << 24) || p.lda >= (1 << 23) || p.ldb >= (1 << 23))) return V3A_ERR_SHAPE;
  const int lds = e.lds + (conv ? p.K / 8 * 4 : 0);
  if (lds > 160 * 1024) return V3A_ERR_SHAPE;
  if (g_attr_lds[ti][conv] < lds) {
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return V3A_ERR_LAUNCH;
    g_attr_lds[ti][conv] = lds;
  }
  const long tiles = (long)((p.M + e.BM - 1) / e.BM) * ((p.N + e.BN - 1) / e.BN);
  hipLaunchKernelGGL(fn, dim3((unsigned)tiles, (unsigned)nz), dim3(e.nthr), lds, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

int pick_tile_f8(int M, int N) {
  double best = 1e30;
  int bi = 0;
  for (int i = 0; i < kNumTilesF8; ++i) {
    const TileEntry& e = kTilesF8[i];
    const long tiles = (long)((M + e.BM - 1) / e.BM) * ((N + e.BN - 1) / e.BN);
    const double cost = (double)((tiles + 255) / 256) * e.BM * e.BN;
    if (cost < best - 1e-9) { best = cost; bi = i; }
  }
  return bi;
}

int launch_f8(const GemmP& p, int ti, void* stream) {
  if (ti < 0 || ti >= kNumTilesF8) ti = pick_tile_f8(p.M, p.N);
  const TileEntry& e = kTilesF8[ti];
  if (g_attr_lds_f8[ti] < e.lds) {
    if (hipFuncSetAttribute((const void*)e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, e.lds) != hipSuccess) return V3A_ERR_LAUNCH;
    g_attr_lds_f8[ti] = e.lds;
  }
  const long tiles = (long)((p.M + e.BM - 1) / e.BM) * ((p.N + e.BN - 1) / e.BN);
  hipLaunchKernelGGL(e.fn, dim3((unsigned)tiles), dim3(e.nthr), e.lds, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

// Second launch of a split-K GEMM: C = epilogue(sum_z partial_z), partials bf16 [S][M][N] summed in fp32 in the order z = 0 .. S-1
// (deterministic), then the documented epilogue order of v3a_gemm_bf16_nt.  One thread per 8 columns.
struct SplitFinP {
  const char* part; char* C; const float* bias; const char* res; const float* scale; const char* res2;
  int M, N, S, ldc, ldr, ldr2, rpb, sstride, act, flags, res_mod, og, os, oo;
};
__global__ __launch_bounds__(256) void gemm_splitk_finish_kernel(const SplitFinP p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int cpr = p.N / 8;
  if (idx >= (long)p.M * cpr) return;
  const int m = (int)(idx / cpr), n = (int)(idx % cpr) * 8;
  float v[8] = {};
  for (int z = 0; z < p.S; ++z) {
    float f[8];
    unpack_bf16x8(*(const u32x4*)(p.part + (((size_t)z * p.M + m) * p.N + n) * 2), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += f[e];
  }
  if (p.bias) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += (p.flags & V3A_GEMM_BIAS_ROW) ? p.bias[m] : p.bias[n + e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float x = round_bf16(v[e]);
    if (p.act != V3A_ACT_NONE) {
      x = p.act == V3A_ACT_GELU_TANH ? gelu_tanh(x) : p.act == V3A_ACT_GELU_ERF ? gelu_erf(x) : p.act == V3A_ACT_SILU ? silu(x) : fmaxf(x, 0.f);
      x = round_bf16(x);
    }
    v[e] = x;
  }
  if (p.scale) {
    const float* sp = p.scale + ((p.flags & V3A_GEMM_SCALE_PER_BATCH) ? (size_t)(m / p.rpb) * p.sstride : 0) + n;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= sp[e];
    if (p.flags & V3A_GEMM_ROUND_AFTER_SCALE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = round_bf16(v[e]);
    }
  }
  if (p.res) {
    const int mr = p.res_mod > 0 ? m % p.res_mod : m;
    if (p.flags & V3A_GEMM_RES_F32) {
      const float* rp = (const float*)p.res + (size_t)mr * p.ldr + n;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rp[e];
    } else {
      float f[8];
      unpack_bf16x8(*(const u32x4*)(p.res + ((size_t)mr * p.ldr + n) * 2), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += f[e];
    }
  }
  if (p.res2) {
    float f[8];
    unpack_bf16x8(*(const u32x4*)(p.res2 + ((size_t)m * p.ldr2 + n) * 2), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += f[e];
  }
  if (p.flags & V3A_GEMM_RELU_OUT) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  const size_t mo = p.og > 0 ? (size_t)m + (size_t)(m / p.og) * p.os + p.oo : (size_t)m;
  if (p.flags & V3A_GEMM_OUT_F32) {
    float* cp = (float*)p.C + mo * p.ldc + n;
#pragma unroll
    for (int e = 0; e < 8; ++e) cp[e] = v[e];
  } else {
    *(u32x4*)(p.C + (mo * p.ldc + n) * 2) = pack_bf16x8(v);
  }
}

constexpr int kKnownFlags = V3A_GEMM_BIAS_ROW | V3A_GEMM_SCALE_PER_BATCH | V3A_GEMM_ROUND_AFTER_SCALE | V3A_GEMM_RES_F32 |
                            V3A_GEMM_OUT_F32 | V3A_GEMM_NO_ROUND_ACC | V3A_GEMM_RELU_OUT;

}  // namespace

extern "C" int v3a_gemm_num_tiles(void) { return kNumTiles; }
extern "C" int v3a_gemm_pick_tile(int M, int N) { return (M > 0 && N > 0) ? pick_tile(M, N) : V3A_ERR_SHAPE; }
extern "C" int v3a_gemm_pick_tile_act(int M, int N, int act) { return (M > 0 && N > 0) ? pick_tile(M, N, false, 1, act) : V3A_ERR_SHAPE; }
// the tile a launch with these properties REALLY runs on (what `tile = -1` resolves to in v3a_gemm_bf16_nt): `mult` = batch count or
// split-K slices side by side on blockIdx.y, `has_tail` = v3a_gemm_args.C_t set (always the transposed-tail tile)
extern "C" int v3a_gemm_pick_tile_ex(int M, int N, int act, int mult, int has_tail) {
  if (M <= 0 || N <= 0 || mult <= 0) return V3A_ERR_SHAPE;
  return has_tail ? kTileTT : pick_tile(M, N, false, mult, act);
}
extern "C" const char* v3a_gemm_tile_name(int t) { return (t >= 0 && t < kNumTiles) ? kTiles[t].name : ""; }

extern "C" int v3a_gemm_bf16_nt(const v3a_gemm_args* a, void* stream) {
  if (!a || !a->A || !a->B || !a->C) return V3A_ERR_ARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return V3A_ERR_SHAPE;
  if (a->K % 64 || a->lda % 8 || a->ldb % 8 || a->ldc % 8 || a->N % 8) return V3A_ERR_SHAPE;
  if (a->residual && (a->ldr % 8)) return V3A_ERR_SHAPE;
  if ((a->flags & V3A_GEMM_SCALE_PER_BATCH) && a->scale && a->rows_per_batch <= 0) return V3A_ERR_ARG;
  if (a->flags & ~kKnownFlags) return V3A_ERR_ARG;          // a stray bit must not silently change behaviour
  if (a->flags & V3A_GEMM_NO_ROUND_ACC) return V3A_ERR_ARG;  // not implemented: accumulators are parked as bf16
  GemmP p = {};
  p.A = (const char*)a->A; p.B = (const char*)a->B; p.C = (char*)a->C;
  p.bias = a->bias; p.res = (const char*)a->residual; p.scale = a->scale;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldr = a->ldr;
  p.rpb = a->rows_per_batch > 0 ? a->rows_per_batch : 1; p.sstride = a->scale_stride;
  p.act = a->act; p.flags = a->flags;
  p.res2 = (const char*)a->residual2; p.ldr2 = a->ldr2; p.res_mod = a->res_row_mod;
  if (a->row_sumsq) {   // by-product of the plain bias epilogue only: the squares are those of bf16(acc + bias)
    // (... and of nothing else: with a residual, a ReLU or a row scatter the statistics would not describe what is stored)
    if (a->N % 32 || a->act != V3A_ACT_NONE || a->scale || a->split_k > 1 || (a->flags & (V3A_GEMM_BIAS_ROW | V3A_GEMM_RELU_OUT)) ||
        a->residual || a->residual2 || a->out_row_group > 0) return V3A_ERR_ARG;
    p.rowsq = a->row_sumsq;
  }
  p.orow_group = a->out_row_group; p.orow_skip = a->out_row_skip; p.orow_off = a->out_row_off;
  if (a->residual2 && (a->ldr2 % 8)) return V3A_ERR_SHAPE;
  if (a->C_t) {   // transposed tail: plain bias epilogue on both sides of t_col0, whole 192-column tiles, 8-row pieces
    if (a->t_col0 <= 0 || a->t_col0 % 192 || a->t_col0 >= a->N || a->ldct % 8 || a->M % 8 || a->act != V3A_ACT_NONE || a->scale || a->residual ||
        a->residual2 || a->out_row_group > 0 || a->split_k > 1 || a->batch > 1 || a->row_sumsq ||
        (a->flags & (V3A_GEMM_BIAS_ROW | V3A_GEMM_OUT_F32 | V3A_GEMM_RELU_OUT))) return V3A_ERR_ARG;
    p.Ct = (char*)a->C_t; p.ldct = a->ldct; p.tcol0 = a->t_col0;
  }
  if (a->split_k < 0 || a->batch < 0 || (a->batch > 1 && a->split_k > 1) || a->batch > 65535) return V3A_ERR_ARG;
  if (a->batch > 1) {     // equally shaped problems side by side on blockIdx.y (per-head / per-prompt operands)
    if (a->a_batch_stride % 8 || a->b_batch_stride % 8 || a->c_batch_stride % 8 || a->res_batch_stride % 8) return V3A_ERR_SHAPE;
    p.az = a->a_batch_stride * 2; p.bz = a->b_batch_stride * 2;
    p.cz = a->c_batch_stride * ((a->flags & V3A_GEMM_OUT_F32) ? 4 : 2);
    p.rz = a->res_batch_stride * ((a->flags & V3A_GEMM_RES_F32) ? 4 : 2);
    return launch(p, a->tile, 0, stream, a->batch);
  }
  if (a->split_k > 1) {   // S equally long K slices side by side (blockIdx.y), bf16 partials, then the epilogue in a second launch
    const int S = a->split_k;
    if (!a->workspace || a->K % (64 * S)) return V3A_ERR_ARG;
    GemmP q = {};
    q.A = p.A; q.B = p.B; q.C = (char*)a->workspace;
    q.M = p.M; q.N = p.N; q.K = p.K / S; q.lda = p.lda; q.ldb = p.ldb; q.ldc = p.N; q.rpb = 1;
    q.az = (long)q.K * 2; q.bz = (long)q.K * 2; q.cz = (long)p.M * p.N * 2;
    const int rc = launch(q, a->tile, 0, stream, S);
    if (rc != V3A_OK) return rc;
    SplitFinP f = {(const char*)a->workspace, p.C, p.bias, p.res, p.scale, p.res2, p.M, p.N, S, p.ldc, p.ldr, p.ldr2, p.rpb, p.sstride,
                   p.act, p.flags, p.res_mod, p.orow_group, p.orow_skip, p.orow_off};
    const long n = (long)p.M * (p.N / 8);
    hipLaunchKernelGGL(gemm_splitk_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, f);
    return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
  }
  return launch(p, a->tile, 0, stream);
}

extern "C" size_t v3a_gemm_split_workspace_bytes(int M, int N, int split_k) {
  return (M > 0 && N > 0 && split_k > 1) ? (size_t)split_k * M * N * 2 : 0;
}

extern "C" int v3a_gemm_fp8_num_tiles(void) { return kNumTilesF8; }
extern "C" int v3a_gemm_fp8_pick_tile(int M, int N) { return (M > 0 && N > 0) ? pick_tile_f8(M, N) : V3A_ERR_SHAPE; }
extern "C" const char* v3a_gemm_fp8_tile_name(int t) { return (t >= 0 && t < kNumTilesF8) ? kTilesF8[t].name : ""; }

extern "C" int v3a_gemm_fp8_nt(const v3a_gemm_fp8_args* f, void* stream) {
  if (!f) return V3A_ERR_ARG;
  const v3a_gemm_args* a = &f->g;
  if (!a->A || !a->B || !a->C || !f->a_scale || !f->b_scale) return V3A_ERR_ARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return V3A_ERR_SHAPE;
  if (a->K % 128 || a->lda % 16 || a->ldb % 16 || a->ldc % 8 || a->N % 8) return V3A_ERR_SHAPE;
  if (a->residual && (a->ldr % 8)) return V3A_ERR_SHAPE;
  if (a->residual2 && (a->ldr2 % 8)) return V3A_ERR_SHAPE;
  if ((a->flags & V3A_GEMM_SCALE_PER_BATCH) && a->scale && a->rows_per_batch <= 0) return V3A_ERR_ARG;
  if (a->flags & ~kKnownFlags) return V3A_ERR_ARG;
  if (a->flags & V3A_GEMM_NO_ROUND_ACC) return V3A_ERR_ARG;
  if (a->split_k > 1) return V3A_ERR_ARG;   // (bf16 entry only)
  GemmP p = {};
  p.A = (const char*)a->A; p.B = (const char*)a->B; p.C = (char*)a->C;
  p.bias = a->bias; p.res = (const char*)a->residual; p.scale = a->scale;
  // the main loop addresses operands in 2-byte units: an e4m3 row of K elements is a bf16 row of K / 2
  p.M = a->M; p.N = a->N; p.K = a->K / 2;
  p.lda = a->lda / 2; p.ldb = a->ldb / 2; p.ldc = a->ldc; p.ldr = a->ldr;
  p.rpb = a->rows_per_batch > 0 ? a->rows_per_batch : 1; p.sstride = a->scale_stride;
  p.act = a->act; p.flags = a->flags;
  p.res2 = (const char*)a->residual2; p.ldr2 = a->ldr2; p.res_mod = a->res_row_mod;
  p.orow_group = a->out_row_group; p.orow_skip = a->out_row_skip; p.orow_off = a->out_row_off;
  p.a_scale = f->a_scale; p.b_scale = f->b_scale;
  return launch_f8(p, a->tile, stream);
}

int v3a_conv_halo_launch(const v3a_conv_args* a, void* stream);   // conv_halo.hip

extern "C" int v3a_conv_bf16(const v3a_conv_args* a, void* stream) {
  if (!a || !a->x || !a->w || !a->y || !a->ktab) return V3A_ERR_ARG;
  if (a->T <= 0 || a->H <= 0 || a->W <= 0 || a->Cin <= 0 || a->oT <= 0 || a->oH <= 0 || a->oW <= 0 || a->Cout <= 0)
    return V3A_ERR_SHAPE;
  if (a->Cin % 8 || a->Cout % 8 || a->Kpad % 64 || a->ldy % 8 || a->Kpad / 8 > 4096) return V3A_ERR_SHAPE;
  if (a->residual && (a->ldr % 8)) return V3A_ERR_SHAPE;
  if ((long)a->oT * a->oH * a->oW > 0x7fffffffL) return V3A_ERR_SHAPE;
  if (a->flags & ~kKnownFlags) return V3A_ERR_ARG;
  if (a->flags & (V3A_GEMM_BIAS_ROW | V3A_GEMM_NO_ROUND_ACC)) return V3A_ERR_ARG;
  GemmP p = conv_gemm_params(a);
  if (a->residual2 && (a->ldr2 % 8)) return V3A_ERR_SHAPE;
  // halo-tile kernel (conv_halo.hip) for the wide-image 3x3(x3) layers: chosen when the layer fills the chip with 16 x 32 tiles
  if (a->w_halo && a->tile != -3 && (a->tile == -2 || (a->tile < 0 && v3a_conv_halo_tiles(a) >= 512))) {
    const int rc = v3a_conv_halo_launch(a, stream);
    if (rc != V3A_ERR_SHAPE || a->tile == -2) return rc;
  }
  return launch(p, a->tile < 0 ? -1 : a->tile, 1, stream);
}

// fp32-equivalent convolution on the bf16 matrix pipe (include/vist3a_hip.h: v3a_conv_split): the implicit-GEMM main loop over the
// K-concatenated partial products, fp32 epilogue.
int v3a_conv_split_halo_launch(const v3a_conv_split_args* s, void* stream);   // conv_halo_split.hip

extern "C" int v3a_conv_split(const v3a_conv_split_args* s, void* stream) {
  if (!s) return V3A_ERR_ARG;
  const v3a_conv_args* a = &s->c;
  if (!a->x || !s->x_lo || !a->w || !a->y || !a->ktab) return V3A_ERR_ARG;
  if (a->T <= 0 || a->H <= 0 || a->W <= 0 || a->Cin <= 0 || a->oT <= 0 || a->oH <= 0 || a->oW <= 0 || a->Cout <= 0) return V3A_ERR_SHAPE;
  if (a->Cin % 8 || a->Cout % 8 || a->Kpad % 64 || a->ldy % 8 || a->Kpad / 8 > 4096) return V3A_ERR_SHAPE;
  if ((long)a->oT * a->oH * a->oW > 0x7fffffffL) return V3A_ERR_SHAPE;
  if (a->flags & ~(V3A_GEMM_RES_F32 | V3A_GEMM_OUT_F32 | V3A_GEMM_RELU_OUT)) return V3A_ERR_ARG;
  if (a->act != V3A_ACT_NONE && a->act != V3A_ACT_RELU) return V3A_ERR_ARG;
  if (a->scale) return V3A_ERR_ARG;
  if (!(a->flags & V3A_GEMM_OUT_F32) && !s->y_lo) return V3A_ERR_ARG;
  if (a->residual && ((a->ldr % 8) || (!(a->flags & V3A_GEMM_RES_F32) && !s->residual_lo))) return V3A_ERR_ARG;
  if (a->residual2 && ((a->ldr2 % 8) || !s->residual2_lo)) return V3A_ERR_ARG;
  GemmP p = conv_gemm_params(a);
  p.A_lo = (const char*)s->x_lo; p.C_lo = (char*)s->y_lo; p.res_lo = (const char*)s->residual_lo; p.res2_lo = (const char*)s->residual2_lo;
  // halo-tile form (conv_halo_split.hip) for the wide-image 3x3 layers: chosen when the layer fills at least half the chip
  if (a->w_halo && a->tile != -3 && (a->tile == -2 || (a->tile < 0 && v3a_conv_split_halo_tiles(s) >= 128))) {
    const int rc = v3a_conv_split_halo_launch(s, stream);
    if (rc != V3A_ERR_SHAPE || a->tile == -2) return rc;
  }
  return launch(p, a->tile < 0 ? -1 : a->tile, 2, stream);
}
