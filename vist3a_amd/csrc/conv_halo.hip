// Halo-tile 3x3 (x kT) convolution for gfx950: the Wan VAE decoder's wide-image layers (SURVEY.md rows V2, V4, V6, V7;
// /root/reference/utils/wan_utils.py:96-147 WanCausalConv3d, :202-330 WanResample's upsample + Conv2d, :333-425 WanResidualBlock).
//
// The implicit-GEMM form (gemm_nt_kernel<.., CONV = true>) gathers every K slab of the im2col matrix from global memory: a 3x3x3
// layer reads its input 27 times through L2 and, with 128-pixel tiles, the whole 0.5 MB weight tensor once per tile - the
// 96-channel 512^2 layers ran at 0.24 of the MFMA peak with the matrix pipe 23 % busy (profiles/r3).  Here the INPUT PATCH of an
// output tile is staged ONCE in LDS and the 9 spatial taps are taken as shifted views of it:
//
//   * one workgroup (8 waves) = 16 x 32 output pixels of one frame x 96 output channels; wave w owns image rows 2w, 2w + 1:
//     a 64 x 96 wave tile = 2 x 3 accumulators of v_mfma_f32_32x32x16_bf16 (the ping-pong GEMM's wave shape);
//   * K is walked as steps (dt, chunk of CK = 48 input channels) x 9 spatial taps.  A step's patch = the 18 x 34 halo pixels x 48
//     channels (57 KB), double buffered: the next step's patch arrives by 16-byte LDS-DMA, spread over the current step's taps.
//     Causal-in-time layers simply skip the steps whose input frame lies before the clip (zero frames);
//   * per tap only the 96 x 48 weight slab (9 KB, pre-packed on the host in exactly its LDS image) streams in, through a 3-deep
//     ring, two taps ahead; counted s_waitcnt vmcnt + one raw s_barrier per tap (18 MFMAs per wave);
//   * a pixel is 96 B = 6 chunks of 16 B, so consecutive pixels would hit the same banks 2-way on ds_read_b128.  Chunk c of
//     pixel (or weight row) p is therefore stored at position (c + 3 ((p >> 3) & 1)) mod 6: the sixteen lanes of a read group
//     then cover all sixteen 16-byte bank slots for every tap shift.  The LDS-DMA writes lane-linear, so the rotation is applied
//     to the per-lane SOURCE address (patch) / by the host packer (weights), and mirrored on the fragment reads;
//   * fused nearest-exact 2x upsample of the input (WanResample) = the patch gather reads pixel (h >> 1, w >> 1);
//   * epilogue = the GEMM's (gemm_epilogue.h): bias, activation, residuals, coalesced 16-byte stores.
//
// LDS traffic per MFMA flop is 5.5x below the implicit-GEMM tile's; results differ from it only in fp32 summation order.
#include "gemm_epilogue.h"
#include <type_traits>

namespace {

constexpr int CK = 48;                 // input channels per step
constexpr int PXB = CK * 2;            // bytes per pixel / weight row in LDS (96: six 16-byte chunks, rotated - see above)
constexpr int TH = 16, TW = 32;        // output tile
constexpr int HH = TH + 2, HW = TW + 2;
constexpr int NPIX = HH * HW;          // 612 halo pixels
constexpr int BN = 96;                 // output channels per workgroup
constexpr int SLAB = BN * PXB;         // 9216: one tap's weights for one step
constexpr int NPIECE = (NPIX * 6 + 63) / 64;   // 58 one-KiB DMA pieces per patch
constexpr int PATCH = NPIECE * 1024;   // 59392: the 612 x 96 B image rounded up to whole pieces (the last piece's tail lanes write zeros)
constexpr int NBI = SLAB / 1024;       // 9 DMA instructions per weight slab
constexpr int PPT = 9;                 // patch pieces issued per tap (taps 0 .. 6)
constexpr int LDS_RING = 2 * PATCH;    // weight ring behind the two patches
constexpr int LDS_TOTAL = 2 * PATCH + 3 * SLAB;
static_assert(NPIECE <= 7 * PPT, "the next patch must be issued within taps 0..6");
static_assert(LDS_TOTAL <= 160 * 1024, "LDS");

struct HaloP {
  GemmP g;               // epilogue descriptor (C, bias, res, ... ; M / N / ldc as for the implicit GEMM)
  const char* x;         // input, channels-last [T][cH][cW][Cin] bf16
  const char* w;         // packed weights [Cout/96][kT][Cin/48][9][96][48] bf16 (rotated chunks)
  int T, H, W;           // OUTPUT extent (= input extent, or 2x the stored input's H, W with ups)
  int cH, cW, Cin;       // stored input extent
  int kT, ups;
  int tilesH, tilesW, nN;
  int inhalo;            // 1: the stored input carries ONE explicit halo row above and below every frame (cH = its rows incl. the two halo rows): an
                         //    H-strip of a spatially sharded image, whose neighbours' boundary rows (or zeros at the image border) were put there
};

// instructions wave w issues in a tap of a step: B slab pieces i < NBI, patch pieces behind them
__host__ __device__ constexpr int issued(int w, int tap, bool hasnext, bool bissue) {
  int n = 0;
  for (int i = w; i < NBI + PPT; i += 8) {
    if (i < NBI) n += bissue ? 1 : 0;
    else if (hasnext && tap < 7 && tap * PPT + (i - NBI) < NPIECE) n += 1;
  }
  return n;
}
__host__ __device__ constexpr int min_issued(int tap, bool hasnext, bool bissue) {
  int m = 1 << 20;
  for (int w = 0; w < 8; ++w) { const int n = issued(w, tap, hasnext, bissue); m = n < m ? n : m; }
  return m;
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(512, 2) void conv_halo_kernel(const HaloP P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;

  // ---- tile decode: consecutive ids = the N tiles of one spatial tile, then neighbours along w, h, t (halo rows shared in L2) ----
  const int nsp = P.T * P.tilesH * P.tilesW;
  int id = xcd_remap(blockIdx.x, nsp * P.nN);
  const int nt = id % P.nN;
  id /= P.nN;
  const int tw = id % P.tilesW, th = (id / P.tilesW) % P.tilesH, t = id / (P.tilesW * P.tilesH);
  const int h0 = th * TH, w0 = tw * TW;

  // ---- steps: dt from the first input frame inside the clip (causal: frame t + dt - (kT - 1)), all channel chunks ----
  const int nchunk = P.Cin / CK;
  const int dt0 = max(0, (P.kT - 1) - t);
  const int nsteps = (P.kT - dt0) * nchunk;
  const size_t frame_bytes = (size_t)P.cH * P.cW * P.Cin * 2;
  const char* wp = P.w + ((size_t)(nt * P.kT + dt0) * nchunk) * 9 * SLAB;   // this tile's slab stream is contiguous from here

  // ---- per-lane sources of this wave's patch pieces (relative to the step's frame / channel origin) ----
  // slot A: piece tap * 9 + (wave - 1) for waves 1..7; slot B: piece tap * 9 + 7 + wave for waves 0, 1
  int offA[7], offB[7];
  unsigned okA = 0, okB = 0;
  auto piece_src = [&](int g, int& off, bool& ok) {
    const int q = g * 64 + lane;
    const int px = q / 6, s = q - px * 6;
    const int ph = px / HW, pw = px - ph * HW;
    const int hh = h0 - 1 + ph, ww = w0 - 1 + pw;
    ok = g < NPIECE && px < NPIX && (P.inhalo || (hh >= 0 && hh < P.H)) && ww >= 0 && ww < P.W;
    // (arithmetic shift: row -1 of the 2x upsampled strip is stored row -1, i.e. halo row 0)
    const int hs = (P.ups ? hh >> 1 : hh) + P.inhalo, ws = P.ups ? ww >> 1 : ww;
    int c = s - 3 * ((px >> 3) & 1);     // LDS position s of pixel px holds logical chunk c
    c += c < 0 ? 6 : 0;
    off = ok ? ((hs * P.cW + ws) * P.Cin + c * 8) * 2 : 0;
  };
#pragma unroll
  for (int tp = 0; tp < 7; ++tp) {
    bool ok;
    piece_src(tp * PPT + wave - 1, offA[tp], ok);
    okA |= (ok && wave >= 1 ? 1u : 0u) << tp;
    piece_src(tp * PPT + 7 + wave, offB[tp], ok);
    okB |= (ok && wave < 2 ? 1u : 0u) << tp;
  }
  // patch pieces of tap-phase tp for step `st` -> patch buffer (st & 1)
  auto issue_patch = [&](auto tp_tag, int st) {
    constexpr int tp = decltype(tp_tag)::value;
    const int dt = dt0 + st / nchunk, ch = st - (st / nchunk) * nchunk;
    const char* org = P.x + (size_t)(t + dt - (P.kT - 1)) * frame_bytes + ch * (CK * 2);
    char* dst = smem + (st & 1) * PATCH;
    if (wave >= 1 && tp * PPT + wave - 1 < NPIECE)
      glds16((okA >> tp) & 1 ? org + offA[tp] : (const char*)&g_zero16, dst + (tp * PPT + wave - 1) * 1024);
    if (wave < 2 && tp * PPT + 7 + wave < NPIECE)
      glds16((okB >> tp) & 1 ? org + offB[tp] : (const char*)&g_zero16, dst + (tp * PPT + 7 + wave) * 1024);
  };
  // weight slab `sl` of the tile's stream -> ring slot `slot`: a linear copy (the host packed the LDS image)
  const char* wlane = wp + lane * 16;
  auto issue_b = [&](int sl, int slot) {
    const char* src = wlane + (size_t)sl * SLAB;
    char* dst = smem + LDS_RING + slot * SLAB;
    glds16(src + wave * 1024, dst + wave * 1024);
    if (wave == 0) glds16(src + 8 * 1024, dst + 8 * 1024);
  };

  // ---- fragment addressing ----
  // B: row n = j * 32 + l31, logical chunk 2 ks + hi at position (chunk + 3 ((n >> 3) & 1)) mod 6
  int boff[3];
  {
    const int rot = 3 * ((l31 >> 3) & 1);
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) { int c = 2 * ks + hi + rot; c -= c >= 6 ? 6 : 0; boff[ks] = l31 * PXB + c * 16; }
  }
  // A: halo pixel (2 wave + i + dh) * 34 + l31 + dw
  const int pbase = (2 * wave) * HW + l31;

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: patch of step 0, slabs 0 and 1 ----
  issue_patch(std::integral_constant<int, 0>{}, 0); issue_patch(std::integral_constant<int, 1>{}, 0);
  issue_patch(std::integral_constant<int, 2>{}, 0); issue_patch(std::integral_constant<int, 3>{}, 0);
  issue_patch(std::integral_constant<int, 4>{}, 0); issue_patch(std::integral_constant<int, 5>{}, 0);
  issue_patch(std::integral_constant<int, 6>{}, 0);
  issue_b(0, 0);
  issue_b(1, 1);

  // one tap of step `st`: wait for its slab (and, at tap 0, the patch), barrier, refill two taps ahead, 18 MFMAs
  auto phase = [&](auto tap_tag, auto next_tag, auto first_tag, int st) {
    constexpr int tap = decltype(tap_tag)::value;
    constexpr bool HASNEXT = decltype(next_tag)::value, FIRST = decltype(first_tag)::value;
    // what may stay in flight: only what this wave issued in the PREVIOUS phase (the minimum over waves: conservative for the rest)
    if constexpr (FIRST && tap == 0) wait_vm<0>();
    else {
      constexpr int ptap = (tap + 8) % 9;
      // the previous phase belongs to this step (tap > 0) or to the previous step, which had a next step (this one)
      constexpr bool phasnext = tap > 0 ? HASNEXT : true;
      constexpr bool pbissue = tap > 0 ? (HASNEXT || ptap + 2 < 9) : true;
      wait_vm<min_issued(ptap, phasnext, pbissue)>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every fragment read of the previous phase returned: its slot may be refilled
    __builtin_amdgcn_s_barrier();
    if (HASNEXT || tap + 2 < 9) issue_b(st * 9 + tap + 2, (tap + 2) % 3);
    if constexpr (HASNEXT && tap < 7) issue_patch(tap_tag, st + 1);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int dh = tap / 3, dw = tap % 3;
    const char* pb = smem + (st & 1) * PATCH;
    const char* sb = smem + LDS_RING + (tap % 3) * SLAB;
    int aaddr[2], arot[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int px = pbase + (i + dh) * HW + dw;
      aaddr[i] = px * PXB;
      arot[i] = 3 * ((px >> 3) & 1) + hi;
    }
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      bf16x8 a[2], b[3];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int c = 2 * ks + arot[i];
        c -= c >= 6 ? 6 : 0;
        a[i] = *(const bf16x8*)(pb + aaddr[i] + c * 16);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) b[j] = *(const bf16x8*)(sb + j * 32 * PXB + boff[ks]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
  };
  using TT = std::true_type;
  using FF = std::false_type;
  auto step = [&](auto next_tag, auto first_tag, int st) {
    phase(std::integral_constant<int, 0>{}, next_tag, first_tag, st);
    phase(std::integral_constant<int, 1>{}, next_tag, first_tag, st);
    phase(std::integral_constant<int, 2>{}, next_tag, first_tag, st);
    phase(std::integral_constant<int, 3>{}, next_tag, first_tag, st);
    phase(std::integral_constant<int, 4>{}, next_tag, first_tag, st);
    phase(std::integral_constant<int, 5>{}, next_tag, first_tag, st);
    phase(std::integral_constant<int, 6>{}, next_tag, first_tag, st);
    phase(std::integral_constant<int, 7>{}, next_tag, first_tag, st);
    phase(std::integral_constant<int, 8>{}, next_tag, first_tag, st);
  };
  if (nsteps == 1) {
    step(FF{}, TT{}, 0);
  } else {
    step(TT{}, TT{}, 0);
    int st = 1;
    for (; st + 1 < nsteps; ++st) step(TT{}, FF{}, st);
    step(FF{}, FF{}, st);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // every fragment read retired: the patches become the epilogue's parking area

  EpiAux aux;
  aux.gstride = P.W;
  const int mw0 = (t * P.H + h0 + 2 * wave) * P.W + w0, nw = nt * BN;
  gemm_add_bias<2, 3>(P.g, acc, lane, mw0, nw);
  gemm_epilogue<2, 3, 3, true>(P.g, acc, smem, wave, lane, mw0, nw, aux);
}

bool g_attr_set = false;

// An H-strip with explicit halo rows (a spatially sharded image, wan/vae.py decode_cl_sharded): the stored input has one more row above and
// below than the output needs centre rows for, and the convolution is VALID in H.  Spelled through the ordinary geometry fields:
//   plain:  pH = 0,  H = oH + 2;        fused 2x upsample:  pH = -1,  2 (H - 2) = oH   (taps of output row j read upsampled rows j + 1 + dh)
inline bool explicit_h_halo(const v3a_conv_args* a) {
  return a->ups2 ? (a->pH == -1 && 2 * (a->H - 2) == a->oH) : (a->pH == 0 && a->H == a->oH + 2);
}
// the layer has this kernel's form (conv_bf16 falls back to the implicit GEMM otherwise)
inline bool halo_form(const v3a_conv_args* a) {
  const int kT = a->halo_kT;
  if (kT != 1 && kT != 3) return false;
  if (a->sT != 1 || a->sH != 1 || a->sW != 1 || a->pW != 1 || a->pT != kT - 1 || a->replicate) return false;
  if (a->Cin % CK || a->Cout % BN || a->oH % TH || a->oW % TW) return false;
  const int eW = a->ups2 ? 2 * a->W : a->W;
  if (a->oW != eW || a->oT != a->T) return false;
  if (!explicit_h_halo(a)) {
    const int eH = a->ups2 ? 2 * a->H : a->H;
    if (a->pH != 1 || a->oH != eH) return false;
  }
  return a->out_row_group <= 0 && !(a->flags & V3A_GEMM_SCALE_PER_BATCH);
}

}  // namespace

// Eligibility + launch; returns V3A_ERR_SHAPE when the layer is not of this kernel's form (the caller falls back to the implicit GEMM).
int v3a_conv_halo_launch(const v3a_conv_args* a, void* stream) {
  if (!a->w_halo) return V3A_ERR_SHAPE;
  const int kT = a->halo_kT;
  if (!halo_form(a)) return V3A_ERR_SHAPE;
  const int inhalo = explicit_h_halo(a) ? 1 : 0;
  if ((size_t)a->H * a->W * a->Cin * 2 >= (1u << 31)) return V3A_ERR_SHAPE;   // 32-bit per-lane offsets inside a frame
  HaloP P = {};
  P.g = conv_gemm_params(a);
  P.x = (const char*)a->x; P.w = (const char*)a->w_halo;
  P.T = a->oT; P.H = a->oH; P.W = a->oW;
  P.cH = a->H; P.cW = a->W; P.Cin = a->Cin;
  P.kT = kT; P.ups = a->ups2 ? 1 : 0; P.inhalo = inhalo;
  P.tilesH = a->oH / TH; P.tilesW = a->oW / TW; P.nN = a->Cout / BN;
  if (!g_attr_set) {
    if (hipFuncSetAttribute((const void*)conv_halo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess)
      return V3A_ERR_LAUNCH;
    g_attr_set = true;
  }
  const long ntiles = (long)P.T * P.tilesH * P.tilesW * P.nN;
  hipLaunchKernelGGL(conv_halo_kernel, dim3((unsigned)ntiles), dim3(512), LDS_TOTAL, (hipStream_t)stream, P);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

// number of tiles the halo kernel would launch for this layer (0 = not eligible): the dispatcher uses it only for layers that fill the chip
extern "C" long v3a_conv_halo_tiles(const v3a_conv_args* a) {
  if (!a || !a->w_halo || !halo_form(a)) return 0;
  return (long)a->oT * (a->oH / TH) * (a->oW / TW) * (a->Cout / BN);
}
