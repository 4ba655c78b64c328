// Voxelisation with confidence-softmax fusion:
//   /root/reference/third_party_model/anysplat/src/model/encoder/anysplat.py:298-335 (voxelizaton_with_fusion)
//   = (pts/voxel).round().int()  ->  torch.unique(dim=0, return_inverse, return_counts)  ->  torch_scatter
//     scatter_max / scatter_add softmax over the per-point confidence  ->  weighted sums of xyz and features.
// Integer part is bit-exact by construction: IEEE divide + round-half-even, a lexicographic (x, y, z) key sorted with a stable
// LSD radix sort (rocPRIM device primitive, the one library call in this file), run-length boundaries by an exclusive scan.
// The key is COMPACT: a first pass takes the per-axis minimum and maximum voxel coordinate, the host reads the six integers
// back (one 24-byte copy + stream synchronise - the caller reads the voxel count back right after this call anyway) and the
// key packs (kx - min_x, ky - min_y, kz - min_z) into exactly bits(x) + bits(y) + bits(z) bits: a 13-view scene spans a few
// thousand voxels per axis (~36 key bits), and the radix sort walks only those instead of all 63 of the fixed (k + 2^20) << {42, 21, 0}
// layout it replaces - same order, same outputs, ~40 % fewer sort passes.  Because the sort is stable, every voxel's
// points appear in ascending original index: the float sums have ONE defined order (the reference's CUDA atomics
// have none).  The fusion pass is a segmented reduction over fixed chunks of the sorted points (see fuse_chunk_kernel).
#include "common.h"
#include "../../include/vist3a_hip.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace {

constexpr int KB = 1 << 20;  // coordinate bias: keys must lie in [-2^20, 2^20)

// voxel coordinate of one component: torch's (pts / voxel_size).round().int() - IEEE divide, round half to even, truncation of an
// integral float; out-of-range values are clamped and flagged
__device__ __forceinline__ int voxel_coord(float x, float vs, int* bad) {
  const float q = rintf(__fdiv_rn(x, vs));
  int v = (int)q;
  if (!(q >= (float)-KB && q < (float)KB)) { atomicOr(bad, 1); v = q < 0 ? -KB : KB - 1; }
  return v;
}

// pass 1: per-axis range of the voxel coordinates, mm = {min x, min y, min z, max x, max y, max z}.  A fixed grid of workgroups walks the points
// with a grid stride (coalesced 12-byte rows), reduces in registers, across the wave, across the workgroup's four waves through LDS, and issues SIX atomics
// per WORKGROUP: 6 k atomics per call on six addresses instead of one set per wave (245 k of them serialised in L2: 2.8 ms on 2.6 M points).
constexpr int RANGE_BLOCKS = 1024;
struct RangeP { const float* pts; float vs; long M; int* mm; int* bad; };
__global__ __launch_bounds__(256) void coord_range_kernel(const RangeP p) {
  __shared__ int red[4][6];
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.M; i += (long)gridDim.x * 256) {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const int v = voxel_coord(p.pts[i * 3 + e], p.vs, p.bad);
      lo[e] = min(lo[e], v); hi[e] = max(hi[e], v);
    }
  }
#pragma unroll
  for (int e = 0; e < 3; ++e) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo[e] = min(lo[e], __shfl_xor(lo[e], o, 64)); hi[e] = max(hi[e], __shfl_xor(hi[e], o, 64)); }
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int e = 0; e < 3; ++e) { red[wave][e] = lo[e]; red[wave][3 + e] = hi[e]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int e = threadIdx.x;
    int v = red[0][e];
    for (int w = 1; w < 4; ++w) v = e < 3 ? min(v, red[w][e]) : max(v, red[w][e]);
    if (e < 3) atomicMin(p.mm + e, v); else atomicMax(p.mm + e, v);
  }
}

// pass 2: compact lexicographic key ((kx - min_x) << (by + bz)) | ((ky - min_y) << bz) | (kz - min_z)
struct KeyP { const float* pts; float vs; long M; unsigned long long* key; unsigned int* idx; int* bad; int mn[3]; int by, bz; };
__global__ __launch_bounds__(256) void make_keys_kernel(const KeyP p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.M) return;
  const unsigned long long x = (unsigned)(voxel_coord(p.pts[i * 3 + 0], p.vs, p.bad) - p.mn[0]);
  const unsigned long long y = (unsigned)(voxel_coord(p.pts[i * 3 + 1], p.vs, p.bad) - p.mn[1]);
  const unsigned long long z = (unsigned)(voxel_coord(p.pts[i * 3 + 2], p.vs, p.bad) - p.mn[2]);
  p.key[i] = (x << (p.by + p.bz)) | (y << p.bz) | z;
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
  long M; int* keys_out; int* inverse; unsigned int* start; int* U; int mn[3]; int by, bz;
};
__global__ __launch_bounds__(256) void segments_kernel(const SegP p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.M) return;
  const unsigned u = p.vid[i] - 1;
  p.inverse[p.idx[i]] = (int)u;
  if (p.head[i]) {
    p.start[u] = (unsigned)i;
    const unsigned long long k = p.key[i];
    p.keys_out[u * 3 + 0] = (int)(k >> (p.by + p.bz)) + p.mn[0];
    p.keys_out[u * 3 + 1] = (int)((k >> p.bz) & ((1ull << p.by) - 1)) + p.mn[1];
    p.keys_out[u * 3 + 2] = (int)(k & ((1ull << p.bz) - 1)) + p.mn[2];
  }
  if (i == p.M - 1) { *p.U = (int)(u + 1); p.start[u + 1] = (unsigned)p.M; }
}

constexpr int CH = 64;      // sorted points per fusion chunk

__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {   // exact and order-independent: the result is deterministic
  // branch on the SIGN BIT, not on v >= 0: -0.0f (0x80000000 = INT_MIN as a signed int) would never replace the -inf initial value
  if (__float_as_int(v) >= 0) atomicMax((int*)addr, __float_as_int(v));
  else atomicMin((unsigned int*)addr, __float_as_uint(v));
}

// per sorted point: its confidence (gathered once, then read coalesced) and the running maximum of its voxel
struct CMaxP { const float* feat; int ldf, conf_col; const unsigned int* idx; const unsigned int* vid; long M; float* csorted; float* vmax; };
__global__ __launch_bounds__(256) void conf_max_kernel(const CMaxP p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.M) return;
  const float c = p.feat[(size_t)p.idx[i] * p.ldf + p.conf_col];
  p.csorted[i] = c;
  atomic_max_f32(p.vmax + (p.vid[i] - 1), c);
}

struct FuseP {
  const float* pts; const float* feat; int ldf, nfeat;
  const unsigned int* idx; const unsigned int* vid; const float* csorted; const float* vmax; long M;
  float* vpts; float* vfeat; int ldo; float* spill;
};
// Confidence-softmax fusion as a SEGMENTED REDUCTION OVER FIXED CHUNKS of the sorted point list: a 16-lane group owns CH = 64 consecutive
// sorted points whatever voxels they belong to, so the work per group does not depend on the cloud - a trained checkpoint puts 1-3 points
// in a voxel, seeded random weights can put thousands in one (the round-4 kernel, one group per VOXEL, ran such a cloud at 7 % of the HBM
// rate: the largest voxel was a serial chain on 16 lanes).  Per chunk: e_j = exp(c_j - max of the voxel) for 16 points at a time, the rows
// of four points requested together (NK 64-byte pieces each: lane sl takes channels sl + 16 k), FMAs in ascending point order.  When the
// voxel id changes the finished voxel is flushed: normalised and written if it lies inside the chunk (bit-identical to a serial sum over
// its points), else left as a PARTIAL (lead = the chunk begins inside the voxel, tail = it ends inside it) that fuse_combine_kernel sums
// in chunk order.  Every sum has one defined order: results do not depend on scheduling.
template <int NK>
__global__ __launch_bounds__(256) void fuse_chunk_kernel(const FuseP p) {
  const int lane = threadIdx.x & 63, sl = lane & 15, gbase = lane & 48;
  const long c = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
  const long j_begin = c * CH;
  if (j_begin >= p.M) return;
  const long j_end = min(j_begin + CH, p.M);
  constexpr int PS = (NK + 2) * 16;      // floats per partial: NK feature slots, one xyz slot, one denominator slot (x 16 lanes)
  float* lead = p.spill + (size_t)c * 2 * PS;
  float* tail = lead + PS;
  const size_t ldf = (size_t)p.ldf;
  unsigned cur = p.vid[j_begin];
  bool inside = j_begin == 0 || p.vid[j_begin - 1] != cur;    // the current voxel began in this chunk
  float acc[NK], ap = 0.f, den = 0.f;
#pragma unroll
  for (int k = 0; k < NK; ++k) acc[k] = 0.f;
  auto flush = [&](bool ends_here) {     // voxel `cur` is done for this chunk
    if (inside && ends_here) {
      const float inv = 1.0f / (den + 1e-6f);
      const size_t u = cur - 1;
#pragma unroll
      for (int k = 0; k < NK; ++k)
        if (sl + 16 * k < p.nfeat) p.vfeat[u * p.ldo + sl + 16 * k] = acc[k] * inv;
      if (sl < 3) p.vpts[u * 3 + sl] = ap * inv;
    } else {
      float* dst = inside ? tail : lead;
#pragma unroll
      for (int k = 0; k < NK; ++k) dst[k * 16 + sl] = acc[k];
      dst[NK * 16 + sl] = ap;
      dst[(NK + 1) * 16 + sl] = den;
    }
  };
  for (long j0 = j_begin; j0 < j_end; j0 += 16) {
    const long j = j0 + sl;
    unsigned myr = 0, myv = 0;
    float mye = 0.f;
    if (j < j_end) {
      myr = p.idx[j];
      myv = p.vid[j];
      mye = expf(p.csorted[j] - p.vmax[myv - 1]);
    }
    const int cnt = (int)min(16L, j_end - j0);
    for (int i = 0; i < cnt; i += 4) {
      float v[4][NK], pv[4], w[4];
      unsigned vv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int src = gbase + min(i + q, 15);
        const size_t r = (size_t)__shfl(myr, src, 64);
        w[q] = __shfl(mye, src, 64);
        vv[q] = __shfl(myv, src, 64);
        const float* row = p.feat + r * ldf;
#pragma unroll
        for (int k = 0; k < NK; ++k) v[q][k] = (i + q < cnt && sl + 16 * k < p.nfeat) ? row[sl + 16 * k] : 0.f;
        pv[q] = (i + q < cnt && sl < 3) ? p.pts[r * 3 + sl] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (i + q < cnt) {
          if (vv[q] != cur) {
            flush(true);
            cur = vv[q]; inside = true; ap = 0.f; den = 0.f;
#pragma unroll
            for (int k = 0; k < NK; ++k) acc[k] = 0.f;
          }
#pragma unroll
          for (int k = 0; k < NK; ++k) acc[k] = fmaf(v[q][k], w[q], acc[k]);
          ap = fmaf(pv[q], w[q], ap);
          den += w[q];
        }
      }
    }
  }
  flush(j_end == p.M || p.vid[j_end] != cur);
}

// voxels that span several chunks: tail partial of the chunk they begin in + lead partials of the following ones, in chunk order;
// and the point count of every voxel
struct CombP { const unsigned int* start; const int* U; const float* spill; float* vpts; float* vfeat; int ldo, nfeat; int* counts; };
template <int NK>
__global__ __launch_bounds__(256) void fuse_combine_kernel(const CombP p) {
  const int sl = threadIdx.x & 15;
  const int U = *p.U;
  constexpr int PS = (NK + 2) * 16;
  for (long u = ((long)blockIdx.x * 256 + threadIdx.x) >> 4; u < U; u += (long)gridDim.x * 16) {
    const unsigned s = p.start[u], e = p.start[u + 1];
    if (sl == 0) p.counts[u] = (int)(e - s);
    const unsigned c0 = s / CH, c1 = (e - 1) / CH;
    if (c0 == c1) continue;      // written by fuse_chunk_kernel
    float acc[NK], ap = 0.f, den = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) acc[k] = 0.f;
    for (unsigned c = c0; c <= c1; ++c) {
      const float* src = p.spill + (size_t)c * 2 * PS + (c == c0 ? PS : 0);
#pragma unroll
      for (int k = 0; k < NK; ++k) acc[k] += src[k * 16 + sl];
      ap += src[NK * 16 + sl];
      den += src[(NK + 1) * 16 + sl];
    }
    const float inv = 1.0f / (den + 1e-6f);
#pragma unroll
    for (int k = 0; k < NK; ++k)
      if (sl + 16 * k < p.nfeat) p.vfeat[(size_t)u * p.ldo + sl + 16 * k] = acc[k] * inv;
    if (sl < 3) p.vpts[(size_t)u * 3 + sl] = ap * inv;
  }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
  size_t key_in, key_out, idx_in, idx_out, head, vid, start, spill, tmp, tmp_bytes, total;
};
Layout layout(long M) {
  Layout l{};
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += align256(b); return o; };
  l.key_in = take(M * 8); l.key_out = take(M * 8);
  l.idx_in = take(M * 4); l.idx_out = take(M * 4);
  l.head = take(M * 4); l.vid = take(M * 4); l.start = take((M + 1) * 4);
  l.spill = take(((size_t)(M + CH - 1) / CH) * 2 * (8 + 2) * 16 * 4);   // (lead, tail) partials of every fusion chunk, NK <= 8
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
  // per-axis coordinate range -> host (the one synchronisation of this call; `head` is free until the sort has run)
  int* mm = (int*)head;
  const int mm_init[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000};
  if (hipMemcpyAsync(mm, mm_init, sizeof(mm_init), hipMemcpyHostToDevice, stream) != hipSuccess) return V3A_ERR_LAUNCH;
  hipLaunchKernelGGL(coord_range_kernel, dim3(nb < (unsigned)RANGE_BLOCKS ? nb : (unsigned)RANGE_BLOCKS), dim3(256), 0, stream, RangeP{pts, voxel_size, M, mm, status});
  int h[6];
  if (hipMemcpyAsync(h, mm, sizeof(h), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
    return V3A_ERR_LAUNCH;
  auto bits = [](long span) { int b = 0; while ((1L << b) <= span) ++b; return b; };   // bits that hold 0 .. span
  const int bx = bits((long)h[3] - h[0]), by = bits((long)h[4] - h[1]), bz = bits((long)h[5] - h[2]);   // <= 21 each (coordinates are clamped to +-2^20)
  const int key_bits = bx + by + bz > 0 ? bx + by + bz : 1;
  KeyP kp{pts, voxel_size, M, key_in, idx_in, status, {h[0], h[1], h[2]}, by, bz};
  hipLaunchKernelGGL(make_keys_kernel, dim3(nb), dim3(256), 0, stream, kp);
  size_t tb = l.tmp_bytes;
  if (rocprim::radix_sort_pairs(ws + l.tmp, tb, key_in, key_out, idx_in, idx_out, (size_t)M, 0, (unsigned)key_bits, stream) != hipSuccess)
    return V3A_ERR_LAUNCH;
  hipLaunchKernelGGL(heads_kernel, dim3(nb), dim3(256), 0, stream, HeadP{key_out, head, M});
  tb = l.tmp_bytes;
  if (rocprim::inclusive_scan(ws + l.tmp, tb, head, vid, (size_t)M, rocprim::plus<unsigned int>(), stream) != hipSuccess)
    return V3A_ERR_LAUNCH;
  hipLaunchKernelGGL(segments_kernel, dim3(nb), dim3(256), 0, stream,
                     SegP{key_out, idx_out, head, vid, M, keys_out, inverse_out, start, num_voxels, {h[0], h[1], h[2]}, by, bz});
  // after the sort key_in (8 M bytes) is free: the sorted confidences and the per-voxel maxima live there
  float* csorted = (float*)key_in;
  float* vmax = csorted + M;
  if (hipMemsetD32Async((hipDeviceptr_t)vmax, 0xff800000u, (size_t)M, stream) != hipSuccess) return V3A_ERR_LAUNCH;   // -inf
  hipLaunchKernelGGL(conf_max_kernel, dim3(nb), dim3(256), 0, stream, CMaxP{feat, ldf, conf_col, idx_out, vid, M, csorted, vmax});
  float* spill = (float*)(ws + l.spill);
  const FuseP fp{pts, feat, ldf, nfeat, idx_out, vid, csorted, vmax, M, voxel_pts, voxel_feat, ldo, spill};
  const CombP cp{start, num_voxels, spill, voxel_pts, voxel_feat, ldo, nfeat, counts_out};
  const unsigned nchunk_blocks = (unsigned)(((M + CH - 1) / CH + 15) / 16);
  if (nfeat <= 32) {
    hipLaunchKernelGGL(fuse_chunk_kernel<2>, dim3(nchunk_blocks), dim3(256), 0, stream, fp);
    hipLaunchKernelGGL(fuse_combine_kernel<2>, dim3(2048), dim3(256), 0, stream, cp);
  } else if (nfeat <= 96) {
    hipLaunchKernelGGL(fuse_chunk_kernel<6>, dim3(nchunk_blocks), dim3(256), 0, stream, fp);
    hipLaunchKernelGGL(fuse_combine_kernel<6>, dim3(2048), dim3(256), 0, stream, cp);
  } else {
    hipLaunchKernelGGL(fuse_chunk_kernel<8>, dim3(nchunk_blocks), dim3(256), 0, stream, fp);
    hipLaunchKernelGGL(fuse_combine_kernel<8>, dim3(2048), dim3(256), 0, stream, cp);
  }
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
