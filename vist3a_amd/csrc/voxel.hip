// Voxelisation with confidence-softmax fusion:
//   /root/reference/third_party_model/anysplat/src/model/encoder/anysplat.py:298-335 (voxelizaton_with_fusion)
//   = (pts/voxel).round().int()  ->  torch.unique(dim=0, return_inverse, return_counts)  ->  torch_scatter
//     scatter_max / scatter_add softmax over the per-point confidence  ->  weighted sums of xyz and features.
// Integer part is bit-exact by construction: IEEE divide + round-half-even, a 63-bit lexicographic key
// ((kx+2^20)<<42 | (ky+2^20)<<21 | (kz+2^20)) sorted with a stable LSD radix sort (rocPRIM device primitive, the one
// library call in this file), run-length boundaries by an exclusive scan.  Because the sort is stable, every voxel's
// points appear in ascending original index: the float sums have ONE defined order (the reference's CUDA atomics
// have none).  The fusion pass gives one wave to a voxel at a time: lanes span the feature channels, so every point
// row (<= 88 floats) is one coalesced read.
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
__global__ __launch_bounds__(256) void fuse_kernel(const FuseP p) {
  const int lane = threadIdx.x & 63;
  const int U = *p.U;
  const int nw = gridDim.x * 4;
  for (int u = blockIdx.x * 4 + (threadIdx.x >> 6); u < U; u += nw) {
    const unsigned s = p.start[u], e = p.start[u + 1];
    if (lane == 0) p.counts[u] = (int)(e - s);
    // softmax over the confidences of the voxel's points (scatter_max, exp, scatter_add, +1e-6)
    float mx = -INFINITY;
    for (unsigned j = s; j < e; ++j) mx = fmaxf(mx, p.feat[(size_t)p.idx[j] * p.ldf + p.conf_col]);
    float den = 0.f;
    for (unsigned j = s; j < e; ++j) den += expf(p.feat[(size_t)p.idx[j] * p.ldf + p.conf_col] - mx);
    den += 1e-6f;
    float a0 = 0.f, a1 = 0.f, ap = 0.f;
    for (unsigned j = s; j < e; ++j) {
      const size_t r = p.idx[j];
      const float w = expf(p.feat[r * p.ldf + p.conf_col] - mx) / den;
      if (lane < p.nfeat) a0 += p.feat[r * p.ldf + lane] * w;
      if (lane + 64 < p.nfeat) a1 += p.feat[r * p.ldf + lane + 64] * w;
      if (lane < 3) ap += p.pts[r * 3 + lane] * w;
    }
    if (lane < p.nfeat) p.vfeat[(size_t)u * p.ldo + lane] = a0;
    if (lane + 64 < p.nfeat) p.vfeat[(size_t)u * p.ldo + lane + 64] = a1;
    if (lane < 3) p.vpts[(size_t)u * 3 + lane] = ap;
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
  hipLaunchKernelGGL(fuse_kernel, dim3(2048), dim3(256), 0, stream,
                     FuseP{pts, feat, ldf, nfeat, conf_col, idx_out, start, num_voxels, voxel_pts, voxel_feat, ldo, counts_out});
  return hipGetLastError() == hipSuccess ? V3A_OK : V3A_ERR_LAUNCH;
}
