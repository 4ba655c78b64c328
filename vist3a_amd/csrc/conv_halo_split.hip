// Halo-tile 3x3 convolution in fp32-EQUIVALENT ("split bf16") arithmetic for gfx950: the wide-image layers of the two DPT heads, which the
// reference runs with autocast off (/root/reference/models/anysplat_stitched.py:335; dpt_head.py:185-309 scratch.layer*_rn, refinenet*,
// output_conv1 / output_conv2; vggt_dpt_gs_head.py:122-176).  Numerics: include/vist3a_hip.h, v3a_conv_split - x and w travel as (hi, lo)
// bf16 pairs, acc += xl.wh + xh.wl + xh.wh on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, fp32 epilogue (gemm_epilogue_f32).
//
// The K-concatenated implicit GEMM (gemm_nt_kernel<.., CONV = 2>) reads every operand plane once per PRODUCT - xh and wh twice - and gathers
// each K slab of the im2col matrix from global memory: 750 TFLOP/s executed = 250 useful on the 256-channel 128^2 layers.  Here, as in
// conv_halo.hip, the INPUT PATCH of an output tile is staged once in LDS (both planes) and the nine taps are shifted views of it; a fragment
// is read once and used by all three products:
//   * one workgroup (8 waves) = 16 x 32 output pixels of one frame x BN = 32 NJ output channels (NJ = 4, 2, 1); wave w owns image rows 2w,
//     2w + 1: 2 x NJ accumulators.  Per tap and wave: 2 (2 + NJ) fragment reads feed 6 NJ MFMAs (0.5 reads per MFMA at NJ = 4; the bf16
//     halo kernel's 2 x 3 wave tile has 0.83);
//   * K is walked as steps of CK = 16 input channels x 9 taps (one k-step of the 32x32x16 MFMA per tap).  A step's patch = 18 x 34 halo
//     pixels x 16 channels x 2 planes (40 KB), double buffered; per tap the (hi, lo) weight slab [2][BN][16] (8 KB at BN = 128, packed on
//     the host in its LDS image) streams through a 3-deep ring, two taps ahead.  Everything arrives by 16-byte LDS-DMA: per tap every wave
//     issues at most one slab piece and (taps 0..4) one patch piece of the next step, so one counted s_waitcnt vmcnt + one raw s_barrier
//     per tap orders the whole pipeline.  (The last step re-issues its own patch and the last taps re-issue the last slab into free slots:
//     2 % redundant traffic for wait counts that never change.)
//   * a pixel / weight row is 32 B = two 16-byte chunks; chunk c of row p sits at position c ^ ((p >> 3) & 1), so the sixteen lanes of a
//     ds_read_b128 group cover all sixteen 16-byte bank slots for every tap shift.
#include "gemm_epilogue.h"
#include <type_traits>

namespace {

constexpr int CK = 16;
constexpr int PXB = CK * 2;             // 32 bytes per pixel / weight row and plane
constexpr int TH = 16, TW = 32, HH = TH + 2, HW = TW + 2, NPIX = HH * HW;   // 612 halo pixels
constexpr int NPP = (NPIX * 2 + 63) / 64;     // 20 one-KiB DMA pieces per plane
constexpr int PLANE = NPP * 1024;             // 20480
constexpr int PATCH = 2 * PLANE;              // hi | lo
static_assert(2 * NPP == 5 * 8, "five patch pieces per wave and step");

struct HaloSplitP {
  GemmP g;                     // epilogue descriptor
  const char* xh; const char* xl;   // channels-last [T][H][W][Cin] bf16 planes
  const char* w;               // [Cout/BN][Cin/16][9][2][BN][16] bf16, chunks rotated
  int T, H, W, Cin;
  int tilesH, tilesW, nN;
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int NJ>
__global__ __launch_bounds__(512, 2) void conv_halo_split_kernel(const HaloSplitP P) {
  constexpr int BN = 32 * NJ, SLAB = 2 * BN * PXB, NBI = SLAB / 1024;   // slab pieces: 8 / 4 / 2
  constexpr int RING = 2 * PATCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;

  const int nsp = P.T * P.tilesH * P.tilesW;
  int id = xcd_remap(blockIdx.x, nsp * P.nN);
  const int nt = id % P.nN;
  id /= P.nN;
  const int tw = id % P.tilesW, th = (id / P.tilesW) % P.tilesH, t = id / (P.tilesW * P.tilesH);
  const int h0 = th * TH, w0 = tw * TW;
  const int nsteps = P.Cin / CK;
  const size_t frame = (size_t)t * P.H * P.W * P.Cin * 2;
  const char* wp = P.w + (size_t)nt * nsteps * 9 * SLAB + lane * 16;

  // ---- this wave's five patch pieces per step: piece g = tp * 8 + wave; g < 20 hi plane, else lo plane ----
  int poff[5];
  unsigned pok = 0;
#pragma unroll
  for (int tp = 0; tp < 5; ++tp) {
    const int g = tp * 8 + wave, gp = g % NPP;
    const int q = gp * 64 + lane;
    const int px = q >> 1, s = q & 1;
    const int ph = px / HW, pw = px - ph * HW;
    const int hh = h0 - 1 + ph, ww = w0 - 1 + pw;
    const bool ok = px < NPIX && hh >= 0 && hh < P.H && ww >= 0 && ww < P.W;
    const int c = s ^ ((px >> 3) & 1);
    poff[tp] = ok ? ((hh * P.W + ww) * P.Cin + c * 8) * 2 : 0;
    pok |= (ok ? 1u : 0u) << tp;
  }
  // piece tp of the patch of step `st` -> patch buffer `buf`
  auto issue_patch = [&](auto tp_tag, int st, int buf) {
    constexpr int tp = decltype(tp_tag)::value;
    const int g = tp * 8 + wave;                      // wave-uniform
    const char* org = (g < NPP ? P.xh : P.xl) + frame + st * (CK * 2);
    glds16((pok >> tp) & 1 ? org + poff[tp] : (const char*)&g_zero16, smem + buf * PATCH + g * 1024);
  };
  const int nslab = nsteps * 9;
  auto issue_b = [&](int sl, int slot) {
    if (wave < NBI) glds16(wp + (size_t)min(sl, nslab - 1) * SLAB + wave * 1024, smem + RING + slot * SLAB + wave * 1024);
  };

  const int brot = (l31 >> 3) & 1;
  const int boff = l31 * PXB + ((hi ^ brot) << 4);
  const int pbase = (2 * wave) * HW + l31;

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issue_patch(std::integral_constant<int, 0>{}, 0, 0); issue_patch(std::integral_constant<int, 1>{}, 0, 0);
  issue_patch(std::integral_constant<int, 2>{}, 0, 0); issue_patch(std::integral_constant<int, 3>{}, 0, 0);
  issue_patch(std::integral_constant<int, 4>{}, 0, 0);
  issue_b(0, 0);
  issue_b(1, 1);

  auto phase = [&](auto tap_tag, auto first_tag, int st) {
    constexpr int tap = decltype(tap_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    if constexpr (FIRST && tap == 0) wait_vm<0>();
    else {
      // still in flight: only what this wave issued in the previous phase (one slab piece if it carries one, one patch piece in taps 0..4)
      constexpr int pp = ((tap + 8) % 9) < 5 ? 1 : 0;
      if (wave < NBI) wait_vm<pp + 1>();
      else wait_vm<pp>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every fragment read of the previous phase returned
    __builtin_amdgcn_s_barrier();
    issue_b(st * 9 + tap + 2, (tap + 2) % 3);
    if constexpr (tap < 5) issue_patch(tap_tag, min(st + 1, nsteps - 1), (st + 1) & 1);   // (last step: a redundant copy into the free buffer)
    __builtin_amdgcn_sched_barrier(0);
    constexpr int dh = tap / 3, dw = tap % 3;
    const char* pb = smem + (st & 1) * PATCH;
    const char* sb = smem + RING + (tap % 3) * SLAB + boff;
    bf16x8 ah[2], al[2], bh[NJ], bl[NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int px = pbase + (i + dh) * HW + dw;
      const int a = px * PXB + ((hi ^ ((px >> 3) & 1)) << 4);
      ah[i] = *(const bf16x8*)(pb + a);
      al[i] = *(const bf16x8*)(pb + PLANE + a);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      bh[j] = *(const bf16x8*)(sb + j * 32 * PXB);
      bl[j] = *(const bf16x8*)(sb + BN * PXB + j * 32 * PXB);
    }
    // small terms first; consecutive MFMAs write different accumulators
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
  };
  auto step = [&](auto first_tag, int st) {
    phase(std::integral_constant<int, 0>{}, first_tag, st);
    phase(std::integral_constant<int, 1>{}, first_tag, st);
    phase(std::integral_constant<int, 2>{}, first_tag, st);
    phase(std::integral_constant<int, 3>{}, first_tag, st);
    phase(std::integral_constant<int, 4>{}, first_tag, st);
    phase(std::integral_constant<int, 5>{}, first_tag, st);
    phase(std::integral_constant<int, 6>{}, first_tag, st);
    phase(std::integral_constant<int, 7>{}, first_tag, st);
    phase(std::integral_constant<int, 8>{}, first_tag, st);
  };
  step(std::true_type{}, 0);
  for (int st = 1; st < nsteps; ++st) step(std::false_type{}, st);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // every fragment read and every (redundant) DMA retired: LDS becomes the epilogue's parking area

  const int mw0 = (t * P.H + h0 + 2 * wave) * P.W + w0, nw = nt * BN;
  gemm_add_bias<2, NJ>(P.g, acc, lane, mw0, nw);
  gemm_epilogue_f32<2, NJ>(P.g, acc, smem, wave, lane, mw0, nw, P.W);
}

template <int NJ> constexpr int lds_bytes() {
  constexpr int main_loop = 2 * PATCH + 3 * (2 * 32 * NJ * PXB);
  constexpr int epi = 8 * 32 * (NJ * 32 * 4 + 16);
  return main_loop > epi ? main_loop : epi;
}
bool g_attr[3] = {};

template <int NJ>
int launch_nj(const HaloSplitP& P, int slot, void* stream) {
  constexpr int lds = lds_bytes<NJ>();
  static_assert(lds <= 160 * 1024, "LDS");
  if (!g_attr[slot]) {
    if (hipFuncSetAttribute((const void*)conv_halo_split_kernel<NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return V3A_ERR_LAUNCH;
    g_attr[slot] = true;
  }
  const long ntiles = (long)P.T * P.tilesH * P.tilesW * P.nN;
  hipLaunchKernelGGL(conv_halo_split_kernel<NJ>, dim3((unsigned)ntiles), dim3(512), lds, (hipStream_t)stream, P);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

}  // namespace

// output channels per workgroup the packing of `w_halo` must use for a layer with Cout channels (0 = no halo form)
extern "C" int v3a_conv_split_halo_bn(int Cout) { return Cout % 128 == 0 ? 128 : (Cout % 64 == 0 ? 64 : (Cout % 32 == 0 ? 32 : 0)); }

// workgroups the halo-split kernel would launch for this layer; 0 = the layer is not of its form
extern "C" long v3a_conv_split_halo_tiles(const v3a_conv_split_args* s) {
  if (!s || !s->c.w_halo) return 0;
  const v3a_conv_args* a = &s->c;
  const int bn = v3a_conv_split_halo_bn(a->Cout);
  if (!bn || a->halo_kT != 1 || a->T != a->oT) return 0;
  if (a->sT != 1 || a->sH != 1 || a->sW != 1 || a->pT != 0 || a->pH != 1 || a->pW != 1 || a->replicate || a->ups2) return 0;
  if (a->Cin % CK || a->oH % TH || a->oW % TW || a->oH != a->H || a->oW != a->W || a->out_row_group > 0) return 0;
  if ((size_t)a->H * a->W * a->Cin * 2 >= (1u << 31)) return 0;
  return (long)a->oT * (a->oH / TH) * (a->oW / TW) * (a->Cout / bn);
}

int v3a_conv_split_halo_launch(const v3a_conv_split_args* s, void* stream) {
  const v3a_conv_args* a = &s->c;
  if (v3a_conv_split_halo_tiles(s) <= 0) return V3A_ERR_SHAPE;
  HaloSplitP P = {};
  P.g = conv_gemm_params(a);
  P.g.C_lo = (char*)s->y_lo; P.g.res_lo = (const char*)s->residual_lo; P.g.res2_lo = (const char*)s->residual2_lo;
  P.xh = (const char*)a->x; P.xl = (const char*)s->x_lo; P.w = (const char*)a->w_halo;
  P.T = a->oT; P.H = a->oH; P.W = a->oW; P.Cin = a->Cin;
  const int bn = v3a_conv_split_halo_bn(a->Cout);
  P.tilesH = a->oH / TH; P.tilesW = a->oW / TW; P.nN = a->Cout / bn;
  return bn == 128 ? launch_nj<4>(P, 0, stream) : (bn == 64 ? launch_nj<2>(P, 1, stream) : launch_nj<1>(P, 2, stream));
}
