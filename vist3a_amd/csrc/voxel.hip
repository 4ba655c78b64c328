// Voxelisation with confidence-softmax fusion:
//   /root/reference/third_party_model/anysplat/src/model/encoder/anysplat.py:298-335 (voxelizaton_with_fusion)
//   = (pts/voxel).round().int()  ->  torch.unique(dim=0, return_inverse, return_counts)  ->  torch_scatter
//     scatter_max / scatter_add softmax over the per-point confidence  ->  weighted sums of xyz and features.
// Integer part is bit-exact by construction: IEEE divide + round-half-even, a 63-bit lexicographic key
// ((kx+2^20)<<42 | (ky+2^20)<<21 | (kz+2^20)) sorted with a stable LSD radix sort (rocPRIM device primitive, the one
// library call in this file), run-length boundaries by an exclusive scan.  Because the sort is stable, every voxel's
// points appear in ascending original index: the float sums have ONE defined order (the reference's CUDA atomics
// have none).  The fusion pass gives a 16-lane group to a voxel (see fuse_kernel).
#include "common.h"
#include "../../include/vist3a_hip.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace {

constexpr int KB = 1 << 20;  // coordinate bias: keys must lie in [-2^20, 2^20)

struct KeyP { const float* pts; float vs; long M; unsigned long long* key; unsigned int* idx; int* bad; };
__global__ __launch_bounds__(256) void make_keys_kernel(const KeyP p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.M) return;
  unsigned long long k = 0;
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const float q = rintf(__fdiv_rn(p.pts[i * 3 + e], p.vs));  // torch: (pts / voxel_size).round()  (half to even)
    int v = (int)q;                                            // .int(): truncation of an integral float
    if (!(q >= (float)-KB && q < (float)KB)) { atomicOr(p.bad, 1); v = q < 0 ? -KB : KB - 1; }
    k = (k << 21) | (unsigned long long)(unsigned)(v + KB);
  }
  p.key[i] = k;
  p.idx[i] = (unsigned)i;
}

struct HeadP { const unsigned long long* key; unsigned int* head; long M; };
__global__ __launch_bounds__(256) void heads_kernel(const HeadP p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.M) return;
  p.head[i] = (i == 0 || p.key[i] != p.key[i - 1]) ? 1u : 0u;
}

struct SegP {
  const unsigned long long* key; const unsigned int* idx; const unsigned int* head; const unsigned int* vid;  // vid = inclusive scan of head
  long M; int* keys_out; int* inverse; unsigned int* start; int* U;
};
__global__ __launch_bounds__(256) void segments_kernel(const SegP p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.M) return;
  const unsigned u = p.vid[i] - 1;
  p.inverse[p.idx[i]] = (int)u;
  if (p.head[i]) {
    p.start[u] = (unsigned)i;
    const unsigned long long k = p.key[i];
    p.keys_out[u * 3 + 0] = (int)((k >> 42) & 0x1fffff) - KB;
    p.keys_out[u * 3 + 1] = (int)((k >> 21) & 0x1fffff) - KB;
    p.keys_out[u * 3 + 2] = (int)(k & 0x1fffff) - KB;
  }
  if (i == p.M - 1) { *p.U = (int)(u + 1); p.start[u + 1] = (unsigned)p.M; }
}

struct FuseP {
  const float* pts; const float* feat; int ldf, nfeat, conf_col;
  const unsigned int* idx; const unsigned int* start; const int* U;
  float* vpts; float* vfeat; int ldo; int* counts;
};
// One 16-LANE GROUP per voxel (four voxels per wave, grid-stride): real scenes put 1-3 points in a voxel, so a whole wave per
// voxel idles 3/4 of its lanes and serialises four dependent round trips (start -> idx -> confidence -> row) per voxel.
// Per voxel: (1) the group gathers the confidences of its points 16 at a time and reduces the maximum with 4 shuffles;
// (2) ONE pass over the points in sorted (= ascending original index) order: e_j = exp(c_j - max) is computed 16 points at a
// time (lane i holds point i), then for each point its row index and e_j are broadcast inside the group and the row is read as
// NK independent 64-byte pieces (lane sl takes channels sl + 16 k): NK loads in flight per point, accumulated with one FMA each.
// The normaliser is applied once at the end: sum_j f_j e_j / (sum_j e_j + 1e-6)  ==  sum_j f_j softmax_j  up to one rounding.
// Every sum runs in a fixed order (points ascending, then a fixed shuffle tree for the denominator): results do not depend on
// scheduling.  Replaces the three-serial-passes-per-wave kernel that ran at 1 % of HBM bandwidth (11.4 ms per scene).
template <int NK>
__global__ __launch_bounds__(256) void fuse_kernel(const FuseP p) {
  const int lane = threadIdx.x & 63, sl = lane & 15, gbase = lane & 48;
  const int U = *p.U;
  const int ngroups = gridDim.x * 16;
  const size_t ldf = (size_t)p.ldf;
  for (int u = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4); u < U; u += ngroups) {
    const unsigned s = p.start[u], e = p.start[u + 1];
    if (sl == 0) p.counts[u] = (int)(e - s);
    float mx = -INFINITY;
    for (unsigned j0 = s; j0 < e; j0 += 16) {
      const unsigned j = j0 + sl;
      if (j < e) mx = fmaxf(mx, p.feat[(size_t)p.idx[j] * ldf + p.conf_col]);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
    float acc[NK], ap = 0.f, dpart = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) acc[k] = 0.f;
    for (unsigned j0 = s; j0 < e; j0 += 16) {
      const unsigned j = j0 + sl;
      unsigned myr = 0;
      float mye = 0.f;
      if (j < e) {
        myr = p.idx[j];
        mye = expf(p.feat[(size_t)myr * ldf + p.conf_col] - mx);
      }
      dpart += mye;
      const int cnt = (int)min(16u, e - j0);
      // four points per trip: their rows are requested together (4 x NK loads in flight instead of NK - a crowded voxel, e.g. the
      // collapsed cloud of seeded random weights with ~76 points per voxel, is a serial chain of row fetches otherwise: 4.3 ms per scene),
      // the FMAs then run in the same ascending point order as before
      for (int i = 0; i < cnt; i += 4) {
        float v[4][NK], pv[4], w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int src = gbase + min(i + q, 15);
          const size_t r = (size_t)__shfl(myr, src, 64);
          w[q] = __shfl(mye, src, 64);
          const float* row = p.feat + r * ldf;
#pragma unroll
          for (int k = 0; k < NK; ++k) v[q][k] = (i + q < cnt && sl + 16 * k < p.nfeat) ? row[sl + 16 * k] : 0.f;
          pv[q] = (i + q < cnt && sl < 3) ? p.pts[r * 3 + sl] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (i + q < cnt) {
#pragma unroll
            for (int k = 0; k < NK; ++k) acc[k] = fmaf(v[q][k], w[q], acc[k]);
            ap = fmaf(pv[q], w[q], ap);
          }
        }
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) dpart += __shfl_xor(dpart, o, 16);
    const float inv = 1.0f / (dpart + 1e-6f);
#pragma unroll
    for (int k = 0; k < NK; ++k)
      if (sl + 16 * k < p.nfeat) p.vfeat[(size_t)u * p.ldo + sl + 16 * k] = acc[k] * inv;
    if (sl < 3) p.vpts[(size_t)u * 3 + sl] = ap * inv;
  }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
  size_t key_in, key_out, idx_in, idx_out, head, vid, start, tmp, tmp_bytes, total;
};
Layout layout(long M) {
  Layout l{};
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += align256(b); return o; };
  l.key_in = take(M * 8); l.key_out = take(M * 8);
  l.idx_in = take(M * 4); l.idx_out = take(M * 4);
  l.head = take(M * 4); l.vid = take(M * 4); l.start = take((M + 1) * 4);
  size_t t1 = 0, t2 = 0;
  (void)rocprim::radix_sort_pairs(nullptr, t1, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned int*)nullptr,
                            (unsigned int*)nullptr, (size_t)M, 0, 63);
  (void)rocprim::inclusive_scan(nullptr, t2, (unsigned int*)nullptr, (unsigned int*)nullptr, (size_t)M, rocprim::plus<unsigned int>());
  l.tmp_bytes = t1 > t2 ? t1 : t2;
  l.tmp = take(l.tmp_bytes);
  l.total = off;
  return l;
}


// ---------------------------------------------------------------------------------------------------------------------
// Confidence-quantile mask + stream compaction (SURVEY.md §8a R13', the voxelize = False branch):
//   /root/reference/models/anysplat_stitched.py:381-387  conf_valid = torch.quantile(depth_conf.flatten(0,1), conf_threshold);
//                                                          mask = depth_conf > conf_valid
//   /root/reference/models/anysplat_stitched.py:441-446  feats = anchor_feats.permute(0,2,3,1)[mask];  pts = pts_all[mask]
// torch.quantile (aten/native/Sorting.cpp quantile_compute, "linear"): sort ascending; rank = q * (n - 1) in the INPUT dtype
// (fp32); lo = floor(rank), hi = ceil(rank), w = rank - lo; result = lerp(sorted[lo], sorted[hi], w) with ATen's lerp
// (w < 0.5 ? a + w (b - a) : b - (b - a)(1 - w)).  Restated with a rocPRIM radix sort of the keys; boolean-mask indexing keeps
// row-major order, i.e. an exclusive scan of the flags gives every kept row its output position (index-exact).
struct QuantP { const float* sorted; long n; float q; float* out; };
__global__ void quantile_pick_kernel(const QuantP p) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float rank = p.q * (float)(p.n - 1);
  const float lo = floorf(rank), hi = ceilf(rank);
  const float w = rank - lo;
  const float a = p.sorted[(long)lo], b = p.sorted[(long)hi];
  *p.out = w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.0f - w);
}
struct FlagP { const float* conf; const float* thr; long M; unsigned int* flag; };
__global__ __launch_bounds__(256) void conf_flags_kernel(const FlagP p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < p.M) p.flag[i] = p.conf[i] > *p.thr ? 1u : 0u;
}
struct CompactP {
  const float* pts; const float* feat; int ldf, nfeat; const unsigned int* flag; const unsigned int* pos; long M;
  float* opts; float* ofeat; int ldo; int* count;
};
// one 16-lane group per input row: kept rows are copied as 64-byte pieces to their scanned position
__global__ __launch_bounds__(256) void compact_rows_kernel(const CompactP p) {
  const int sl = threadIdx.x & 15;
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
  if (i >= p.M) return;
  if (i == p.M - 1 && sl == 0) *p.count = (int)(p.pos[i] + p.flag[i]);
  if (!p.flag[i]) return;
  const size_t o = p.pos[i];
  for (int c = sl; c < p.nfeat; c += 16) p.ofeat[o * p.ldo + c] = p.feat[(size_t)i * p.ldf + c];
  if (sl < 3) p.opts[o * 3 + sl] = p.pts[(size_t)i * 3 + sl];
}

struct CLayout { size_t sorted, flag, pos, thr, tmp, tmp_bytes, total; };
CLayout clayout(long M) {
  CLayout l{};
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += align256(b); return o; };
  l.sorted = take(M * 4); l.flag = take(M * 4); l.pos = take(M * 4); l.thr = take(4);
  size_t t1 = 0, t2 = 0;
  (void)rocprim::radix_sort_keys(nullptr, t1, (float*)nullptr, (float*)nullptr, (size_t)M, 0, 32);
  (void)rocprim::exclusive_scan(nullptr, t2, (unsigned int*)nullptr, (unsigned int*)nullptr, 0u, (size_t)M, rocprim::plus<unsigned int>());
  l.tmp_bytes = t1 > t2 ? t1 : t2;
  l.tmp = take(l.tmp_bytes);
  l.total = off;
  return l;
}
}  // namespace

extern "C" long v3a_voxelize_workspace_bytes(long M) {
  if (M <= 0) return 0;
  return (long)layout(M).total;
}

extern "C" int v3a_voxelize_fuse(const float* pts, const float* feat, int ldf, int nfeat, int conf_col, long M, float voxel_size,
                                 void* workspace, long workspace_bytes, int* keys_out, int* inverse_out, int* counts_out,
                                 float* voxel_pts, float* voxel_feat, int ldo, int* num_voxels, int* status, void* stream_) {
  if (!pts || !feat || !workspace || !keys_out || !inverse_out || !counts_out || !voxel_pts || !voxel_feat || !num_voxels || !status)
    return V3A_ERR_ARG;
  if (M <= 0 || M > 0x7fffffffL || nfeat <= 0 || nfeat > 128 || conf_col < 0 || conf_col >= ldf || nfeat > ldf || ldo < nfeat || !(voxel_size > 0.f))
    return V3A_ERR_SHAPE;
  const Layout l = layout(M);
  if ((size_t)workspace_bytes < l.total) return V3A_ERR_SHAPE;
  hipStream_t stream = (hipStream_t)stream_;
  char* ws = (char*)workspace;
  auto* key_in = (unsigned long long*)(ws + l.key_in);
  auto* key_out = (unsigned long long*)(ws + l.key_out);
  auto* idx_in = (unsigned int*)(ws + l.idx_in);
  auto* idx_out = (unsigned int*)(ws + l.idx_out);
  auto* head = (unsigned int*)(ws + l.head);
  auto* vid = (unsigned int*)(ws + l.vid);
  auto* start = (unsigned int*)(ws + l.start);
  const unsigned nb = (unsigned)((M + 255) / 256);
  if (hipMemsetAsync(status, 0, sizeof(int), stream) != hipSuccess) return V3A_ERR_LAUNCH;
  hipLaunchKernelGGL(make_keys_kernel, dim3(nb), dim3(256), 0, stream, KeyP{pts, voxel_size, M, key_in, idx_in, status});
  size_t tb = l.tmp_bytes;
  if (rocprim::radix_sort_pairs(ws + l.tmp, tb, key_in, key_out, idx_in, idx_out, (size_t)M, 0, 63, stream) != hipSuccess)
    return V3A_ERR_LAUNCH;
  hipLaunchKernelGGL(heads_kernel, dim3(nb), dim3(256), 0, stream, HeadP{key_out, head, M});
  tb = l.tmp_bytes;
  if (rocprim::inclusive_scan(ws + l.tmp, tb, head, vid, (size_t)M, rocprim::plus<unsigned int>(), stream) != hipSuccess)
    return V3A_ERR_LAUNCH;
  hipLaunchKernelGGL(segments_kernel, dim3(nb), dim3(256), 0, stream,
                     SegP{key_out, idx_out, head, vid, M, keys_out, inverse_out, start, num_voxels});
  const FuseP fp{pts, feat, ldf, nfeat, conf_col, idx_out, start, num_voxels, voxel_pts, voxel_feat, ldo, counts_out};
  if (nfeat <= 32) hipLaunchKernelGGL(fuse_kernel<2>, dim3(2048), dim3(256), 0, stream, fp);
  else if (nfeat <= 96) hipLaunchKernelGGL(fuse_kernel<6>, dim3(2048), dim3(256), 0, stream, fp);
  else hipLaunchKernelGGL(fuse_kernel<8>, dim3(2048), dim3(256), 0, stream, fp);
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}

extern "C" long v3a_conf_compact_workspace_bytes(long M) { return M > 0 ? (long)clayout(M).total : 0; }

extern "C" int v3a_conf_quantile_compact(const float* conf, float q, const float* pts, const float* feat, int ldf, int nfeat, long M,
                                         void* workspace, long workspace_bytes, float* threshold_out, float* out_pts, float* out_feat,
                                         int ldo, int* count_out, void* stream_) {
  if (!conf || !pts || !feat || !workspace || !threshold_out || !out_pts || !out_feat || !count_out) return V3A_ERR_ARG;
  if (M <= 0 || M > (1L << 24) || nfeat <= 0 || nfeat > ldf || ldo < nfeat || !(q >= 0.f && q <= 1.f)) return V3A_ERR_SHAPE;  // fp32 rank is exact up to 2^24
  const CLayout l = clayout(M);
  if ((size_t)workspace_bytes < l.total) return V3A_ERR_SHAPE;
  hipStream_t stream = (hipStream_t)stream_;
  char* ws = (char*)workspace;
  auto* sorted = (float*)(ws + l.sorted);
  auto* flag = (unsigned int*)(ws + l.flag);
  auto* pos = (unsigned int*)(ws + l.pos);
  size_t tb = l.tmp_bytes;
  if (rocprim::radix_sort_keys(ws + l.tmp, tb, conf, sorted, (size_t)M, 0, 32, stream) != hipSuccess) return V3A_ERR_LAUNCH;
  hipLaunchKernelGGL(quantile_pick_kernel, dim3(1), dim3(64), 0, stream, QuantP{sorted, M, q, threshold_out});
  const unsigned nb = (unsigned)((M + 255) / 256);
  hipLaunchKernelGGL(conf_flags_kernel, dim3(nb), dim3(256), 0, stream, FlagP{conf, threshold_out, M, flag});
  tb = l.tmp_bytes;
  if (rocprim::exclusive_scan(ws + l.tmp, tb, flag, pos, 0u, (size_t)M, rocprim::plus<unsigned int>(), stream) != hipSuccess)
    return V3A_ERR_LAUNCH;
  hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)((M * 16 + 255) / 256)), dim3(256), 0, stream,
                     CompactP{pts, feat, ldf, nfeat, flag, pos, M, out_pts, out_feat, ldo, count_out});
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}
