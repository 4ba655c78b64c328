/*
 * libvist3a_hip.so — C ABI of the MI355X (gfx950) kernels behind the VIST3A text->3DGS inference path.
 *
 * The reference (gohyojun15/VIST3A) has no FFI layer: its "operator API" for this path is the PyTorch
 * nn.Module surface.  Every entry point below replaces the ATen/cuDNN/SDPA call(s) the
 * reference reaches at the cited file:line; the Python host in vist3a_amd/ calls them through ctypes with
 * raw device pointers.  Conventions:
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless noted;
 *   - `stream` is a hipStream_t passed as void*; calls are stream-ordered, re-entrant, allocate nothing;
 *   - return 0 on success, negative V3A_ERR_* otherwise; never throws;
 *   - bf16 tensors are raw uint16 storage (round-to-nearest-even, same as torch.bfloat16).
 */
#ifndef VIST3A_HIP_H
#define VIST3A_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define V3A_OK 0
#define V3A_ERR_ARG (-1)
#define V3A_ERR_SHAPE (-2)
#define V3A_ERR_LAUNCH (-3)
#define V3A_ERR_WORKSPACE (-4)

int v3a_abi_version(void);            /* bumps whenever a signature changes (currently 21) */
const char* v3a_build_info(void);     /* "gfx950 <date> <compiler>" */

/* ------------------------------------------------------------------------------------------------
 * GEMM  C[M,N] = epilogue( A[M,K] . B[N,K]^T )      bf16 inputs, fp32 MFMA accumulation
 * replaces torch.nn.Linear / F.linear under bf16 autocast:
 *   DiT   : diffusers==0.33.1 WanAttnProcessor2_0 to_q/to_k/to_v/to_out, FeedForward net.0.proj/net.2,
 *           condition_embedder, proj_out (call site /root/reference/inference_t23d.py:94-103)
 *   recon : vggt/layers/attention.py:51,77 (qkv, proj), vggt/layers/mlp.py:33-38 (fc1, fc2)
 * Epilogue order (each stage optional):  v = acc + bias ; v = bf16(v) ; v = act(v) ; v = bf16(v) ;
 *   v = v * scale ; [v = bf16(v)] ; v = v + residual ; store as bf16 or f32.
 * These rounding points are the ones CUDA autocast produces in the reference.
 * ---------------------------------------------------------------------------------------------- */
enum {
  V3A_ACT_NONE = 0,
  V3A_ACT_GELU_TANH = 1, /* FeedForward(activation_fn="gelu-approximate") */
  V3A_ACT_GELU_ERF = 2,  /* vggt/layers/mlp.py nn.GELU */
  V3A_ACT_SILU = 3,
  V3A_ACT_RELU = 4
};
enum {
  V3A_GEMM_BIAS_ROW = 1 << 0,     /* bias indexed by output row (used for the V^T = Wv.X^T form) */
  V3A_GEMM_SCALE_PER_BATCH = 1 << 1, /* scale[(row / rows_per_batch) * scale_stride + col] (AdaLN gate) else scale[col] (LayerScale) */
  V3A_GEMM_ROUND_AFTER_SCALE = 1 << 2, /* round v*scale to bf16 before adding the residual */
  V3A_GEMM_RES_F32 = 1 << 3,      /* residual is float32 (aggregator residual stream) else bf16 */
  V3A_GEMM_OUT_F32 = 1 << 4,      /* store float32 else bf16 */
  V3A_GEMM_NO_ROUND_ACC = 1 << 5, /* reserved (not implemented: accumulators are parked as bf16) */
  V3A_GEMM_RELU_OUT = 1 << 6      /* ReLU after the residual adds (DPT ResidualConvUnit chains) */
};
typedef struct {
  const void* A;        /* bf16 [M, lda] */
  const void* B;        /* bf16 [N, ldb] (nn.Linear weight layout) */
  void* C;              /* bf16 or f32 [M, ldc] */
  const float* bias;    /* f32 [N] (or [M] with BIAS_ROW); may be NULL */
  const void* residual; /* bf16/f32 [M, ldr]; may be NULL; may alias C */
  const float* scale;   /* f32; may be NULL */
  int M, N, K;          /* K % 64 == 0, lda/ldb % 8 == 0, ldc % 8 == 0 */
  int lda, ldb, ldc, ldr;
  int rows_per_batch, scale_stride;
  int act;              /* V3A_ACT_* */
  int flags;            /* V3A_GEMM_* */
  int tile;             /* -1 = auto, else index into the tile table (bench/tuning; all tiles give bit-identical results): 0-5 lockstep, 6-8 ping-pong,
                         * 9-11 small, 12 128x96, 13 transposed tail, 14 one wave per SIMD (hand-scheduled K loop), 15-21 deep-ring forms of the
                         * small / medium tiles (shard-size launches); v3a_gemm_tile_name(i) names them */
  const void* residual2; /* optional second residual, bf16 [M, ldr2] */
  int ldr2;
  int res_row_mod;      /* > 0: `residual` row index is (row % res_row_mod): broadcast table (positional embedding) */
  int out_row_group, out_row_skip, out_row_off; /* group > 0: output row = row + (row / group) * skip + off */
  int split_k;          /* > 1 (v3a_gemm_bf16_nt only): K is cut into split_k equal slices (K % (64 * split_k) == 0) computed side by side
                         * into bf16 partials, summed in slice order by a second launch that applies the epilogue - for problems with
                         * few output tiles and a long K (a sequence-parallel shard's FFN2).  Deterministic; not bit-identical to 0 / 1. */
  void* workspace;      /* split_k > 1: v3a_gemm_split_workspace_bytes(M, N, split_k) bytes */
  int batch;            /* > 1 (v3a_gemm_bf16_nt, not with split_k): `batch` equally shaped problems side by side in ONE launch; problem z
                         * reads A + z * a_batch_stride, B + z * b_batch_stride, writes C + z * c_batch_stride, adds residual +
                         * z * res_batch_stride (strides in ELEMENTS of the respective tensor; 0 = shared).  bias / scale / residual2 are
                         * shared.  Used for per-head and per-prompt operands (the cached-context cross-attention of the DiT: one B
                         * operand per CFG batch item) where separate launches would each fill a fraction of the chip. */
  long a_batch_stride, b_batch_stride, c_batch_stride, res_batch_stride;
  float* row_sumsq;     /* optional by-product (v3a_gemm_bf16_nt; N % 32 == 0, act NONE, no scale, no BIAS_ROW, no split_k): f32 [M][N / 32],
                         * entry [m][n / 32] = sum over the 32-column block of bf16(acc + bias)^2 - the statistics of an RMS norm that the
                         * CONSUMER applies (v3a_xattn_probs_args.q_row_sumsq), so the normalisation pass over the output disappears.
                         * Independent of the tile shape (bit-identical for every tile); with `batch` the rows of problem z are
                         * offset by z * M. */
  void* C_t;            /* optional TRANSPOSED TAIL (v3a_gemm_bf16_nt): output columns n >= t_col0 are written to C_t[(n - t_col0) * ldct + m] (bf16)
                         * instead of C; columns below t_col0 go to C as usual (ldc may be t_col0).  For the fused q | k | v projection of a DiT
                         * block (diffusers WanAttnProcessor2_0 to_q / to_k / to_v, call site /root/reference/inference_t23d.py:94-103): B =
                         * (Wq | Wk | Wv) stacked, q | k land row-major in C and V^T - what the flash kernel reads - in C_t, as 768 tiles = three
                         * full rounds of ONE launch.  t_col0 % 192 == 0, ldct % 8 == 0, M % 8 == 0; bias only (no act / scale / residual / row
                         * scatter / split_k / batch / row_sumsq).  Bit-identical to the two separate GEMMs. */
  int ldct, t_col0;
} v3a_gemm_args;
int v3a_gemm_bf16_nt(const v3a_gemm_args* args, void* stream);
size_t v3a_gemm_split_workspace_bytes(int M, int N, int split_k);
int v3a_gemm_num_tiles(void);
int v3a_gemm_pick_tile(int M, int N);   /* the tile index tile=-1 resolves to (profiling / roofline bookkeeping) */
int v3a_gemm_pick_tile_act(int M, int N, int act);   /* ... for a launch with activation `act` (ties between mirrored tiles depend on it) */
/* ... the tile a launch REALLY runs on: `mult` = batch / split_k problems side by side, `has_tail` != 0 when C_t is set (the
 * transposed-tail tile whatever M, N).  An explicit `tile` >= 0 that contradicts C_t is rejected with V3A_ERR_ARG by the launch. */
int v3a_gemm_pick_tile_ex(int M, int N, int act, int mult, int has_tail);
const char* v3a_gemm_tile_name(int tile);

/* e4m3 (OCP fp8) form of v3a_gemm_bf16_nt for BASELINE config #4 (Wan-14B: "MFMA bf16/fp8 GEMMs for the attention/FFN contractions"):
 *   C[m][n] = epilogue( a_scale[m] * b_scale[n] * sum_k A8[m][k] * B8[n][k] )        (fp32 accumulation, scales multiplied first)
 * A8 [M][lda], B8 [N][ldb] are e4m3 BYTES (K % 128 == 0, lda/ldb % 16 == 0) as written by v3a_quantize_fp8_rows (per-row dynamic
 * scales: activations per token, weights per output channel).  Everything after the dequantisation - bias, activation, gate scale,
 * residuals, output type, row scatter, flags - is the bf16 GEMM's epilogue, field for field (g.*).  The main loop is the ping-pong
 * loop of v3a_gemm_bf16_nt with v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales) in place of the bf16 MFMA: same bytes per
 * LDS row, twice the K per phase.  g.tile: -1 = heuristic, or 0 .. v3a_gemm_fp8_num_tiles()-1.
 * v3a_quantize_fp8_rows: scale[r] = max(amax_r, 1e-12) / 448 (IEEE), y[r][c] = e4m3(clamp(x[r][c] * (1 / scale[r]), +-448)), RNE.
 * x bf16 [rows][ldx] (cols % 8 == 0), y bytes [rows][ldy], scale fp32 [rows]. */
typedef struct {
  v3a_gemm_args g;          /* A, B: e4m3 bytes; lda, ldb, K in elements (= bytes) */
  const float* a_scale;     /* [M] */
  const float* b_scale;     /* [N] */
} v3a_gemm_fp8_args;
int v3a_gemm_fp8_nt(const v3a_gemm_fp8_args* args, void* stream);
int v3a_gemm_fp8_num_tiles(void);
int v3a_gemm_fp8_pick_tile(int M, int N);
const char* v3a_gemm_fp8_tile_name(int tile);
int v3a_quantize_fp8_rows(const void* x, void* y, float* scale, long rows, int cols, int ldx, int ldy, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Convolution (1-D/2-D/3-D, stride, zero or replicate padding, causal-in-time, optional fused nearest-exact 2x
 * spatial upsample of the input) as an implicit GEMM on the same MFMA main loop and epilogue as v3a_gemm_bf16_nt.
 * replaces nn.Conv3d / nn.Conv2d at
 *   /root/reference/utils/wan_utils.py:96-147 (WanCausalConv3d), :202-330 (WanResample upsample + Conv2d, time_conv)
 *   /root/reference/models/stitching_layer_builder.py:32-42 (stitching Conv3d, padding_mode="replicate")
 *   /root/reference/third_party_model/anysplat/src/model/encoder/vggt/heads/dpt_head.py (DPT convs)
 * Activations are CHANNELS-LAST: x[T][H][W][Cin] bf16, y[oT*oH*oW][ldy] (first Cout columns written).
 * Weights are pre-packed  w[Cout][Kpad],  k = ((dt*kH + dh)*kW + dw)*Cin + c, zero padded to Kpad % 64 == 0.
 * ktab[Kpad/8] describes each 8-channel K chunk:  c | dw<<16 | dh<<20 | dt<<24 | 1<<31 (0 for padding chunks).
 * Tap (dt,dh,dw) of output (t,h,w) reads input (t*sT+dt-pT, h*sH+dh-pH, w*sW+dw-pW); out-of-range taps read zero
 * (or the clamped coordinate when `replicate`).  With ups2 the (H,W) coordinates address the 2x upsampled image.
 * Epilogue = the GEMM epilogue (bias per Cout, act, scale per Cout, residual [M][ldr], bf16/f32 out).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const void* x; const void* w; const int* ktab; void* y;
  const float* bias; const void* residual; const float* scale;
  int T, H, W, Cin;
  int oT, oH, oW, Cout, Kpad;
  int sT, sH, sW;
  int pT, pH, pW;
  int ups2, replicate;
  int ldy, ldr;
  int act, flags, tile;
  const void* residual2; int ldr2; int res_row_mod;
  int out_row_group, out_row_skip, out_row_off;   /* as in v3a_gemm_args */
  /* Optional second packing of the same weights for the HALO-TILE kernel (csrc/conv_halo.hip): 3x3 spatial taps, kT = halo_kT in
   * {1, 3} (causal in time: pT = kT - 1), stride 1, zero padding 1, Cin % 48 == 0, Cout % 96 == 0, oH % 16 == 0, oW % 32 == 0.
   * The input patch of a 16 x 32-pixel output tile is staged once in LDS and the nine spatial taps are read as shifted views of it
   * (the Wan VAE decoder's wide-image layers, /root/reference/utils/wan_utils.py:96-147, 202-330, 333-425).
   * Layout: w_halo[Cout/96][kT][Cin/48][9 = dh*3+dw][96 rows n][48 k] bf16, where the six 16-byte chunks of a row are stored
   * rotated: logical chunk c of row n sits at position (c + 3*((n >> 3) & 1)) % 6 (the kernel's LDS bank layout; the slab is
   * copied to LDS verbatim).  NULL = implicit-GEMM path only.  v3a_conv_bf16 takes the halo kernel when w_halo is set, the layer
   * has this form, tile < 0 and v3a_conv_halo_tiles(args) >= 512; tile == -2 forces it, tile == -3 forbids it.
   * H-STRIPS (a spatially sharded image: wan/vae.py decode_cl_sharded): the stored input may carry ONE explicit halo row above and below
   * every frame and be convolved VALID in H - pH = 0 with H = oH + 2, or with ups2 pH = -1 with 2 (H - 2) = oH (output row j reads rows
   * j + 1 + dh of the 2x upsampled haloed strip).  Both kernels take it (negative pH is plain arithmetic in the implicit GEMM). */
  const void* w_halo; int halo_kT;
} v3a_conv_args;
int v3a_conv_bf16(const v3a_conv_args* args, void* stream);
long v3a_conv_halo_tiles(const v3a_conv_args* args);   /* workgroups the halo kernel would launch; 0 = layer not of its form */

/* ------------------------------------------------------------------------------------------------
 * fp32-EQUIVALENT convolution on the bf16 matrix pipe ("split bf16"), for the layers the reference runs with autocast OFF:
 *   /root/reference/models/anysplat_stitched.py:335 (`with torch.amp.autocast("cuda", enabled=False)` around camera / depth / Gaussian heads)
 *   /root/reference/third_party_model/anysplat/src/model/encoder/vggt/heads/dpt_head.py:185-309 (DPT trunk + output convs, fp32 weights:
 *   utils/utils_for_thirdparty.py:53-69), .../encoder/heads/vggt_dpt_gs_head.py:122-176.
 * An fp32 tensor t travels as an unevaluated PAIR of bf16 planes of its shape, t = hi + lo with hi = bf16(t), lo = bf16(t - hi)
 * (|t - hi - lo| <= 2^-17 |t|).  With x = xh + xl and w = wh + wl the convolution is the sum of three bf16 products with fp32
 * accumulation,  xl.wh + xh.wl + xh.wh  (the dropped xl.wl term is <= 2^-16 relative), laid out ALONG K of one implicit GEMM:
 *   w[Cout][Kpad]: three consecutive copies of the tap-major K range of v3a_conv_args, holding (wh | wl | wh);
 *   ktab: the chunk table of the three ranges, bit 28 set where the chunk reads the LO plane of x (first range), clear for the hi plane.
 * Everything that follows the accumulation stays in fp32: v = act(acc + bias) (act NONE or RELU) + residual + residual2 ; [RELU_OUT] ;
 * stored as f32 (V3A_GEMM_OUT_F32) or split into the (y, y_lo) planes.  residual: f32 [.., ldr] with V3A_GEMM_RES_F32 (a table, see
 * res_row_mod) else a pair (c.residual, residual_lo); residual2: a pair.  c.scale must be NULL.  Geometry fields (T..pW, ups2, replicate,
 * ldy, out_row_*) as in v3a_conv_args; both planes of a pair share strides.
 * HALO-TILE form (csrc/conv_halo_split.hip) for 3x3 layers with stride 1, zero padding 1, kT = 1, Cin % 16 == 0, Cout % 32 == 0,
 * oH % 16 == 0, oW % 32 == 0: the (hi, lo) input patch of a 16 x 32-pixel output tile is staged once in LDS, the nine taps are shifted
 * views of it and every fragment feeds all three products.  c.w_halo = the second packing
 *   [Cout/BN][Cin/16][9 = dh*3+dw][2 = hi, lo][BN rows n][16 k] bf16,  BN = v3a_conv_split_halo_bn(Cout)  (128, 64 or 32),
 * where the two 16-byte chunks of a row are swapped when (n >> 3) & 1 (the kernel's LDS bank layout; slabs are copied verbatim), and
 * c.halo_kT = 1.  Taken when w_halo is set, the layer has this form, tile < 0 and v3a_conv_split_halo_tiles(args) >= 128; tile == -2
 * forces it, tile == -3 forbids it, tile >= 0 selects that implicit-GEMM tile.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  v3a_conv_args c;            /* x, y, residual, residual2 = the HI planes */
  const void* x_lo; void* y_lo; const void* residual_lo; const void* residual2_lo;
} v3a_conv_split_args;
int v3a_conv_split(const v3a_conv_split_args* args, void* stream);
int v3a_conv_split_halo_bn(int Cout);                               /* output channels per workgroup of the halo packing; 0 = none */
long v3a_conv_split_halo_tiles(const v3a_conv_split_args* args);    /* workgroups the halo form would launch; 0 = layer not of its form */
/* Row / pixel passes between the split convolutions (csrc/pair.hip), fp32 arithmetic, pair in / pair out:
 *   v3a_split_f32        x f32 [n] (n % 8 == 0) -> (hi, lo)
 *   v3a_layernorm_pair   F.layer_norm of f32 rows x[row(m)][0..d) (row(m) = m + (m / in_row_group) * in_row_skip + in_row_off when
 *                        in_row_group > 0), affine weight / bias f32 [d] or NULL -> pair rows [M][ldy]; d % 8 == 0, d <= 2048
 *                        (dpt_head.py:213-216 `self.norm` on the tapped tokens)
 *   v3a_bilinear_cl_pair v3a_bilinear_cl on pairs: x [T][h][w][C] -> y [T][H][W][C], + optional pair `add` [T][H][W][C], + optional f32
 *                        `table` [H*W][C] broadcast over T (dpt_head.py:291-309,460-466, vggt_dpt_gs_head.py:166) */
int v3a_split_f32(const float* x, void* hi, void* lo, long n, void* stream);
int v3a_layernorm_pair(const float* x, void* y_hi, void* y_lo, const float* weight, const float* bias, int M, int d, int ldx, int ldy,
                       float eps, int in_row_group, int in_row_skip, int in_row_off, void* stream);
int v3a_bilinear_cl_pair(const void* x_hi, const void* x_lo, void* y_hi, void* y_lo, const void* add_hi, const void* add_lo,
                         const float* table, int T, int h, int w, int H, int W, int C, int align_corners, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Flash attention forward (non-causal, no mask, no dropout), bf16 in/out, fp32 softmax.
 * replaces F.scaled_dot_product_attention at
 *   diffusers==0.33.1 WanAttnProcessor2_0 (DiT self/cross attention; call site inference_t23d.py:94-103)
 *   /root/reference/third_party_model/anysplat/src/model/encoder/vggt/layers/attention.py:64-69
 * Q,K: [B][N][H*D] with row stride ld* ; V is passed TRANSPOSED: vt[(h*D+d)*ldvt + b*vt_batch_stride + key],
 * readable and finite up to the next multiple of 64 keys past each batch's Nk.  D in {64,128}.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const void* q; const void* k; const void* vt; void* o;   /* bf16 */
  long q_batch_stride, k_batch_stride, vt_batch_stride, o_batch_stride; /* elements */
  int ldq, ldk, ldvt, ldo;                                  /* elements */
  int B, H, Nq, Nk, D;
  float scale;                                              /* softmax scale, normally D^-0.5 */
  int kv_period, kv_valid;  /* kv_period > 0: key k participates only if (k % kv_period) < kv_valid (per-frame row padding);
                               not together with rel_bias / key_bias (V3A_ERR_ARG) */
  const float* rel_bias;    /* optional (D = 64 only): additive bias by relative position, fp32 [H][rel_bias_stride], entry
                             * (key - query + rel_bias_center) is added to scale*q.k — T5/UMT5 relative attention bias */
  int rel_bias_stride, rel_bias_center;
  const float* key_bias;    /* optional (D = 128 only): additive per-key bias, fp32 [B][key_bias_stride >= Nk], added to scale*q.k.
                             * Used to merge the identical zero-padding keys of a prompt into one key carrying log(count). */
  int key_bias_stride;
  int key_bias_first;       /* keys < key_bias_first have zero bias (their tiles skip the bias loads); 0 = any key may carry bias */
  int kv_seg;               /* optional (D = 128 only): keys of a batch item are split into segments of kv_seg keys (multiple of 64, divides Nk),
                             * segment s of K at k + s * k_seg_stride, of V^T at vt + s * vt_seg_stride (elements): the all-gathered per-rank
                             * slabs of sequence-parallel attention are read in place.  Inside a segment: ldk / k_batch_stride / ldvt /
                             * vt_batch_stride as usual */
  long k_seg_stride, vt_seg_stride;
  int kv_split;             /* optional (D = 128 only): > 1 divides the key tiles of every query block among kv_split workgroups and
                             * merges their partial softmaxes in a second launch (fixed order: deterministic, but not bit-identical to
                             * kv_split <= 1).  For launches too small to occupy the chip: a sequence-parallel shard has N / P query rows
                             * but walks all N keys per workgroup. */
  void* workspace;          /* kv_split > 1: v3a_attention_split_workspace_bytes(B, H, Nq, D, kv_split) bytes */
} v3a_attn_args;
int v3a_attention_fwd_bf16(const v3a_attn_args* args, void* stream);
size_t v3a_attention_split_workspace_bytes(int B, int H, int Nq, int D, int kv_split);

/* ------------------------------------------------------------------------------------------------
 * Cross-attention PROBABILITIES over a short per-prompt key set (csrc/xattn_probs.hip): the first half of the cached-context form
 *     attn2(x) = softmax(q K^T) V Wo^T + bo = sum_h P_h (V_h Wo_h^T) + bo
 * of diffusers==0.33.1 WanTransformerBlock.attn2 (call site /root/reference/inference_t23d.py:94-103).  K, V depend on the prompt only:
 * the host caches (V_h Wo_h^T) per prompt and v3a_gemm_bf16_nt (batch = CFG items, K = H * Lkp) finishes the layer.
 *   p[b][m][h * Lkp + j] = bf16( softmax_j( scale * q[b][m][h*D:(h+1)*D] . k[b][j][h*D:(h+1)*D] + key_bias[b][j] ) ),  j < Nk;  0 for
 *   Nk <= j < Lkp.  D = 128, Nk <= 128, Lkp % 16 == 0, Nk <= Lkp <= 128, ldp >= H * Lkp.  key_bias (optional, f32 [B][key_bias_stride])
 *   is added for keys >= key_bias_first (the merged zero-padding key carries log(count), v3a_attn_args.key_bias).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const void* q; const void* k; void* p;      /* bf16 */
  const float* key_bias;
  long q_batch_stride, k_batch_stride, p_batch_stride;   /* elements */
  int ldq, ldk, ldp;
  int B, H, Nq, Nk, D, Lkp;
  int key_bias_stride, key_bias_first;
  float scale;
  const float* q_row_sumsq;   /* optional: q is NOT yet RMS-normalised; f32 [B * Nq rows of q][q_sumsq_parts] partial sums of squares of each
                               * full q row (v3a_gemm_args.row_sumsq of the projection that produced q: parts = H * D / 32).  The kernel
                               * multiplies the scores of query m by rsqrt(sum(parts) / (H * D) + q_eps) - the per-row factor of
                               * diffusers' RMSNorm "across heads"; its per-column weight must be folded into k by the caller. */
  int q_sumsq_parts;
  float q_eps;
} v3a_xattn_probs_args;
int v3a_xattn_probs_bf16(const v3a_xattn_probs_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * fp8 (OCP e4m3) flash attention forward on the block-scaled MFMA (K = 64, twice the bf16 rate): the self-attention launch of
 * BASELINE config #4 (Wan-14B; model choice at /root/reference/utils/argument.py:399, /root/reference/inference_t23d.py:73).
 * The reference computes this attention in bf16 (F.scaled_dot_product_attention under autocast): this entry is an opt-in
 * precision mode whose rounding points are pinned by the oracle's e4m3 emulation (oracle/wan_dit.py attention_fp8_emulated).
 * q, k: e4m3 bytes [B][N][H*128] (row stride ld* BYTES); vt: e4m3 bytes [H*128][B*vt_batch_stride + key], readable up to the next
 * multiple of 64 keys; o: bf16.  Stored values are x / *_scale; D must be 128.
 * v3a_quantize_fp8: y[r][c] = e4m3(x[r][c] / scale), RNE, clamped to +-448; x bf16 [rows][ldx], y bytes [rows][ldy], cols % 16 == 0.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const void* q; const void* k; const void* vt; void* o;
  long q_batch_stride, k_batch_stride, vt_batch_stride, o_batch_stride;   /* elements of the respective tensor */
  int ldq, ldk, ldvt, ldo;
  int B, H, Nq, Nk, D;
  float scale;                      /* softmax scale, normally D^-0.5 */
  float q_scale, k_scale, v_scale;  /* per-tensor quantisation scales */
  int kv_seg;               /* as in v3a_attn_args: K / V^T read in place from per-rank slabs; k_seg_stride / vt_seg_stride in BYTES */
  long k_seg_stride, vt_seg_stride;
  int kv_split;             /* as in v3a_attn_args (workspace: v3a_attention_split_workspace_bytes(B, H, Nq, 128, kv_split)) */
  void* workspace;
} v3a_attn_fp8_args;
int v3a_attention_fwd_fp8(const v3a_attn_fp8_args* args, void* stream);
int v3a_quantize_fp8(const void* x, void* y, long rows, int cols, int ldx, int ldy, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm in fp32 (+ optional affine) (+ optional AdaLN modulation  y = LN(x)*(1+scale[b])+shift[b]).
 * replaces diffusers==0.33.1 FP32LayerNorm in WanTransformerBlock (norm1/norm2/norm3) and norm_out,
 * and nn.LayerNorm at /root/reference/third_party_model/anysplat/src/model/encoder/vggt/layers/block.py:41,47.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const void* x; void* y;             /* [M, ld*]; bf16 unless *_is_f32 */
  const float* weight; const float* bias;   /* [d] or NULL */
  const float* scale; const float* shift;   /* [nbatch, mod_stride] or NULL (both or neither) */
  int M, d, ldx, ldy;
  int rows_per_batch, mod_stride;
  float eps;
  int x_is_f32, y_is_f32;
  int in_row_group, in_row_skip, in_row_off;    /* group > 0: logical row m reads x row m + (m/group)*skip + off */
  int out_row_group, out_row_skip, out_row_off; /* same for the output (gather/scatter of per-frame token blocks) */
  int rms;                                      /* 1: RMS norm, no mean subtraction (T5LayerNorm of the UMT5 text encoder) */
  float* y_fp8_scale;                           /* non-NULL: y is e4m3 BYTES [rows][ldy] = v3a_quantize_fp8_rows of the bf16 result, its
                                                   per-row scales written here (fp8 GEMM mode: saves the separate quantisation pass) */
} v3a_layernorm_args;
int v3a_layernorm(const v3a_layernorm_args* args, void* stream);

/* RMSNorm across the full width + optional RoPE (complex multiply of adjacent pairs, per head).
 * replaces diffusers==0.33.1 RMSNorm (attn.norm_q / norm_k, "rms_norm_across_heads") followed by
 * WanAttnProcessor2_0.apply_rotary_emb.  rope = float [tokens_per_batch][head_dim/2][2] = (cos, sin). */
typedef struct {
  const void* x; void* y;     /* bf16 [M, ld*]; y may alias x */
  const float* weight;        /* [d] */
  const float* rope;          /* or NULL */
  int M, d, ldx, ldy, head_dim, tokens_per_batch;
  float eps;
  const float* weight2;       /* non-NULL: a SECOND column block x[:, d:2d] -> y[:, d:2d] normalised with weight2 in the same launch
                               * (q and k of the fused q|k projection: one launch instead of two) */
  float y_fp8_scale;          /* > 0: y is e4m3 BYTES [M, ldy] (ldy in bytes) holding e4m3(bf16(result) / y_fp8_scale), the operand format of
                               * v3a_attention_fwd_fp8 (= this call followed by v3a_quantize_fp8, bit for bit, in one pass); y != x */
} v3a_rmsnorm_rope_args;
int v3a_rmsnorm_rope(const v3a_rmsnorm_rope_args* args, void* stream);

/* Channel norm over channels-last pixels (+ optional SiLU): mode 0 RMSNorm(eps); mode 1 = F.normalize(x,dim=C)
 * * sqrt(C) * gamma (+ bias), i.e. WanRMS_norm (/root/reference/utils/wan_utils.py:150-184) fused with the SiLU that
 * follows it in WanResidualBlock (:370-372,:399-400) and WanDecoder3d (:873-874). bf16 in / bf16 out. */
typedef struct {
  const void* x; void* y;
  const float* weight; const float* bias;   /* [d]; bias may be NULL */
  long M; int d, ldx, ldy;
  float eps; int mode; int act;             /* act: V3A_ACT_NONE or V3A_ACT_SILU */
} v3a_rownorm_args;
int v3a_rownorm_act(const v3a_rownorm_args* args, void* stream);

/* P[M,N] (bf16) = softmax(scale * S[M,N]) (f32), row-wise.  The VAE mid-block's single 384-wide head
 * (/root/reference/utils/wan_utils.py:428-475) runs as GEMM -> this -> GEMM. */
int v3a_softmax_rows(const float* s, void* p, int M, int N, int lds, int ldp, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Stitched-reconstruction glue.  Channels-last (CL) tensors are [T][H][W][C].
 * ---------------------------------------------------------------------------------------------- */
/* per-head LayerNorm(64, affine) on q and k + 2-D RoPE, in place on the fused [q|k] buffer qk[M][ld] (q cols 0..C-1,
 * k cols C..2C-1).  vggt/layers/attention.py:56-61, rope.py:154-188.  cos_sin = f32 [maxpos][16][2]; token row r has
 * p = r % rows_per_frame; p < n_special or p >= n_valid -> position 0, else patch p-n_special -> (y,x)+1 on a wp-wide grid. */
int v3a_qknorm_rope2d(void* qk, int M, int ld, int C, const float* q_w, const float* q_b, const float* k_w, const float* k_b,
                      const float* cos_sin, int rows_per_frame, int n_special, int n_valid, int wp, float eps, void* stream);
/* latent z[C][Tl][H][W] f32 -> CL bf16 [4(Tl-1)+1][H][W][C], align_corners=True lerp in T (models/stitched_model.py:92-107) */
int v3a_latent_upsample_t_cl(const float* z, void* y, int C, int Tl, int H, int W, void* stream);
/* bilinear resize of a CL bf16 stack (+ bf16 addend [T*H*W][C]) (+ f32 table [H*W][C] broadcast over T) (+ ReLU)
 * vggt/heads/dpt_head.py:460-466 (align_corners=True), inference_t23d.py:118-123 (False) */
int v3a_bilinear_cl(const void* x, void* y, const void* add, const float* table, int T, int h, int w, int H, int W, int C,
                    int align_corners, int relu, int out_f32, void* stream);
/* depth head activation + unprojection: raw[M][ld] f32 (col0 log-depth, col1 log-conf) -> depth[M]=exp, conf[M]=1+exp,
 * pts[M][3] world points.  cam[S][16] = {fx,fy,cx,cy, R^T (9, row-major), -R^T t (3)}.  head_act.py:61-112, geometry.py:10-58 */
int v3a_depth_unproject(const float* raw, int ld, const float* cam, float* depth, float* conf, float* pts, int S, int H, int W,
                        void* stream);
/* voxel fusion: see csrc/voxel.hip.  feat[M][ldf] f32 holds nfeat feature columns (0..nfeat-1) and the raw confidence at
 * conf_col.  Outputs sized for the worst case U = M: keys_out[M][3] i32 (lexicographically sorted unique voxel coords),
 * inverse_out[M] i32, counts_out[M] i32, voxel_pts[M][3], voxel_feat[M][ldo]; *num_voxels = U (device int).
 * *status (device int) != 0 if a coordinate fell outside [-2^20, 2^20) voxels.  anysplat.py:298-335
 * The call SYNCHRONISES `stream` once (the per-axis coordinate range is read back to size the sort key): not capturable in a hipGraph. */
long v3a_voxelize_workspace_bytes(long M);
int v3a_voxelize_fuse(const float* pts, const float* feat, int ldf, int nfeat, int conf_col, long M, float voxel_size,
                      void* workspace, long workspace_bytes, int* keys_out, int* inverse_out, int* counts_out,
                      float* voxel_pts, float* voxel_feat, int ldo, int* num_voxels, int* status, void* stream);
/* confidence-quantile mask + row compaction (the voxelize = False branch): threshold = torch.quantile(conf[M], q) ("linear"),
 * rows with conf > threshold are copied in row-major order to out_pts[.][3] / out_feat[.][ldo]; *count_out (device int) = rows kept,
 * *threshold_out (device float) = the quantile.  M <= 2^24.  /root/reference/models/anysplat_stitched.py:381-387, 441-446 */
long v3a_conf_compact_workspace_bytes(long M);
int v3a_conf_quantile_compact(const float* conf, float q, const float* pts, const float* feat, int ldf, int nfeat, long M,
                              void* workspace, long workspace_bytes, float* threshold_out, float* out_pts, float* out_feat, int ldo,
                              int* count_out, void* stream);
/* UnifiedGaussianAdapter + opacity map: feats[U][ldf] = {density logit, 3 scale, 4 quat xyzw, 3*(deg+1)^2 SH}
 * common/gaussian_adapter.py:114-147, common/gaussians.py:33-44, anysplat.py:225-238 (opacity_exponent = 2^x) */
int v3a_gaussian_adapter(const float* pts, const float* feats, int ldf, long U, int sh_degree, float opacity_exponent,
                         const float* sh_mask, float* means, float* cov, float* sh, float* opac, float* scales, float* rot,
                         void* stream);

/* One denoise step's glue between two DiT forwards, as ONE launch: unpatchify of the DiT output tokens, classifier-free guidance,
 * flow-prediction -> x0, UniPC corrector + predictor, and the patchified bf16 input tokens of the next forward.  Replaces the ~25
 * tensor ops of diffusers 0.33.1 `WanPipeline.__call__` (loop body) + `UniPCMultistepScheduler.step` that the reference runs per step
 * (/root/reference/inference_t23d.py:94-103; restated as tensor ops in vist3a_amd/wan/{pipeline,scheduler}.py) with the same
 * rounding points, bit for bit.  The scalar coefficients are the host-side fp32 values `UniPCMultistepScheduler.plan_step` returns. */
typedef struct {
  const void* dit_out;        /* [batch * N][4 C] bf16: DiT output tokens, column (ph * 2 + pw) * C + c; item 0 = conditional, 1 = unconditional */
  void* tok;                  /* [batch * N][4 C] bf16: next forward's input tokens, column c * 4 + ph * 2 + pw (may be NULL) */
  const float* sample;        /* [C][T][H][W] fp32 current latents */
  const float* last_sample;   /* corrected sample of the previous step (corr_order > 0) */
  const float* m_prev1;       /* x0 prediction of the previous step (model_outputs[-1]) */
  const float* m_prev2;       /* ... of the step before (order-2 corrector) */
  float* m_out;               /* this step's x0 prediction */
  float* sample_corrected;    /* corrector output (= sample when corr_order == 0): next step's last_sample */
  float* prev;                /* predictor output: next step's latents */
  int C, T, H, W, batch, guided;
  float guidance, sigma;
  int corr_order;             /* 0 = no corrector (first step), 1, 2 */
  float cc1, cc2, cc3, c_rho_last, c_rho0, c_inv_rk;
  int pred_order;             /* 1, 2 */
  float pc1, pc2, pc3, p_rho0, p_inv_rk;
} v3a_unipc_step_args;
int v3a_unipc_cfg_step(const v3a_unipc_step_args* args, void* stream);
/* fp32 camera-head primitives (vggt/heads/camera_head.py:87-170): y[M<=32][N] = act(x.W^T + b) * gamma + residual */
int v3a_linear_f32(const float* x, const float* w, const float* bias, float* y, const float* residual, const float* gamma,
                   int M, int N, int K, int ldx, int ldy, int ldr, int act, void* stream);
int v3a_attention_small_f32(const float* qkv, float* out, int S, int H, int hd, float scale, void* stream);

/* Skinny GEMM: X has M <= 128 rows, W [N][K] is streamed once (K % 512 == 0).  C[m][n] = act(X.W^T + bias[n]) (+ residual), or the
 * transposed store C[n][m] (bias still indexed by n) for the V^T = Wv.X^T form.  Same rounding points as v3a_gemm_bf16_nt.
 * Replaces nn.Linear on a prompt's tokens: transformers UMT5 q/k/v/o/wi_0/wi_1/wo (the text encoder behind
 * diffusers WanPipeline.encode_prompt; /root/reference/utils/wan_utils.py:25-60). */
typedef struct {
  const void* X; const void* W; void* C;   /* bf16 X, W; C bf16 or fp32 (V3A_GEMM_OUT_F32) */
  const float* bias;                        /* [N] or NULL */
  const void* residual;                     /* C-shaped, bf16 or fp32 (V3A_GEMM_RES_F32), or NULL */
  int M, N, K, ldx, ldw, ldc, ldr;
  int act, flags, transposed_out;
  void* workspace; long workspace_bytes;    /* v3a_gemm_skinny_workspace_bytes(M, N, K) */
} v3a_gemm_skinny_args;
long v3a_gemm_skinny_workspace_bytes(int M, int N, int K);
int v3a_gemm_skinny_bf16(const v3a_gemm_skinny_args* a, void* stream);

/* ---- 3D-Gaussian rasteriser: gsplat==1.4.0 `rasterization(..., render_mode="RGB+D", packed=False,
 * near_plane=1e-10, radius_clip=0.1, covars=..., rasterize_mode="classic")` as driven by
 * third_party_model/anysplat/src/model/decoder/decoder_splatting_cuda.py:96-125.  The reference passes one camera per call;
 * here a call takes C cameras (gsplat's own leading C dimension): per camera the arithmetic is identical, the batch is what
 * fills 256 CUs.  Per-camera arrays are camera-major ([C,U,...], [C,H,W,...]).  All pointers are device pointers unless noted.  Stage 1 = fully_fused_projection + spherical_harmonics (colours only for radii > 0, clamp_min(c + 0.5, 0)). */
typedef struct {
  const float* means;    /* [U,3] world */
  const float* covars;   /* [U,3,3] world covariance, row-major; the upper triangle is used */
  const float* sh;       /* harmonics */
  int sh_layout;         /* 0: [U,K,3] (gsplat argument layout), 1: [U,3,K] (Gaussians.harmonics layout, types.py) */
  int sh_k;              /* K coefficients stored per channel (25) */
  int sh_degree;         /* 0..4 evaluated */
  const float* viewmat;  /* [C,4,4] world->camera, row-major */
  const float* campos;   /* [C,3] camera centre in world = inverse(viewmat)[:3,3] */
  const float* K;        /* [C,3,3] pixel intrinsics */
  long U;
  int C;
  int width, height;
  float near_plane, far_plane, radius_clip, eps2d;
  int* radii;            /* out [C,U]: 3-sigma pixel radius, 0 = culled */
  float* means2d;        /* out [C,U,2] */
  float* depths;         /* out [C,U] camera z */
  float* conics;         /* out [C,U,3] inverse 2D covariance (a,b,c) */
  float* colors;         /* out [C,U,4] rgb + depth */
} v3a_gs_project_args;
int v3a_gs_project(const v3a_gs_project_args* a, void* stream);

/* Stage 2 = isect_tiles + radix sort by (tile, depth bits) + isect_offset_encode + rasterize_to_pixels (16x16 tiles).
 * Synchronises the stream once (4-byte read-back of the intersection count that sizes the sort, like gsplat's .item()).
 * Returns V3A_ERR_WORKSPACE (-4) when *n_isect > max_isect: call again with a workspace sized for *n_isect. */
typedef struct {
  const int* radii; const float* means2d; const float* depths; const float* conics; const float* colors;
  const float* opacities;   /* [U] (shared by the cameras) */
  const float* background;  /* [3] RGB or NULL (the depth channel's background is 0) */
  long U;
  int C;
  int width, height;
  int clamp_rgb;            /* 1: clamp RGB to [0,1] (decoder_splatting_cuda.py:117) */
  float* out_color;         /* [C,H,W,3] */
  float* out_depth;         /* [C,H,W]  sum(vis_i * z_i) */
  float* out_alpha;         /* [C,H,W]  1 - T */
  void* workspace; long workspace_bytes; long max_isect;
  long* n_isect;            /* HOST pointer: intersections found */
  unsigned int* tile_offsets_out;  /* optional [C*ntiles+1] */
  unsigned int* flatten_ids_out;   /* optional [max_isect]: entry ids c*U+g in composite order */
} v3a_gs_rasterize_args;
long v3a_gs_rasterize_workspace_bytes(long U, int C, int width, int height, long max_isect);
int v3a_gs_rasterize(const v3a_gs_rasterize_args* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif
