// K1 gather-sum (GraphConv.sum_neigh) and K3 gather-max (GraphPool) over the
// degree-bucketed CSR of a collated batch, plus their scatter backwards.
//
// Mapping to the hardware (MI355X, wave = 64 lanes):
//   * a "lane slot" = one 16-byte (V=4) or 4-byte (V=1) column chunk of one
//     atom row; slots are laid out row-major, so the 64 lanes of a wave read /
//     write contiguous 1 KiB (V=4) pieces of the output and whole neighbour
//     rows of the input -- every global access is coalesced at row granularity.
//   * one workgroup works inside ONE degree block (the block index -> degree
//     lookup is scalar, from a table in the kernarg segment), so the neighbour
//     loop has a wave-uniform trip count and the row pointer is implicit:
//     no row_ptr array is read at all.
//   * each lane keeps up to 4 neighbour rows in flight (unrolled loop) and each
//     thread walks UNROLL slots, so a CU has >= 32 KiB of loads outstanding.
// Bound: HBM (indices + neighbour rows in, one row out).  Algorithmic bytes per
// launch: E*(4F+4) + N*4F  (SURVEY.md 8d).
#include "common.h"

namespace gcmi {

constexpr int kBlock = 256;
constexpr int kUnroll = 4;  // slots per thread

struct TileTable {
  int32_t tile_start[GCMI_MAX_DEG + 2];  // first workgroup of every degree block
};

static int make_tiles(const gcmi_graph* g, int lpr, bool skip_deg0, TileTable* tt) {
  int64_t tiles = 0;
  const int64_t per_tile = (int64_t)kBlock * kUnroll;
  for (int d = 0; d <= GCMI_MAX_DEG + 1; ++d) {
    tt->tile_start[d] = (int32_t)tiles;
    if (d <= g->max_deg && !(skip_deg0 && d == 0)) {
      int64_t slots = (int64_t)(g->deg_start[d + 1] - g->deg_start[d]) * lpr;
      tiles += (slots + per_tile - 1) / per_tile;
    }
  }
  return (int)tiles;
}

__device__ __forceinline__ int block_degree(const TileTable& tt, int b) {
  int d = 0;
#pragma unroll
  for (int k = 1; k <= GCMI_MAX_DEG; ++k) d += (b >= tt.tile_start[k]) ? 1 : 0;
  return d;  // tile_start is non-decreasing; empty blocks share a start and are skipped
}

// picks table entry d with scalar selects (d is wave-uniform)
__device__ __forceinline__ int pick(const int32_t* a, int d) {
  int v = a[0];
#pragma unroll
  for (int k = 1; k <= GCMI_MAX_DEG + 1; ++k) v = (d == k) ? a[k] : v;
  return v;
}

template <int V>
__device__ __forceinline__ typename Vec<V>::T ldv(const float* p) {
  return *reinterpret_cast<const typename Vec<V>::T*>(p);
}
template <int V>
__device__ __forceinline__ void stv(float* p, typename Vec<V>::T v) {
  *reinterpret_cast<typename Vec<V>::T*>(p) = v;
}

template <int V>
__device__ __forceinline__ void ld_arr(const float* p, float (&v)[V]) {
  if constexpr (V == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = *p;
  }
}
template <int V>
__device__ __forceinline__ void st_arr(float* p, const float (&v)[V]) {
  if constexpr (V == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    *p = v[0];
  }
}

// ------------------------------------------------------------------ gather-sum
template <int V, bool ACC>
__global__ void __launch_bounds__(kBlock)
gather_sum_kernel(DegTable t, TileTable tt, const int32_t* __restrict__ col,
                  const float* __restrict__ x, int64_t ldx, int lpr, float* __restrict__ s,
                  int64_t lds) {
  using VT = typename Vec<V>::T;
  const int b = blockIdx.x;
  const int d = block_degree(tt, b);
  const int row0 = pick(t.deg_start, d);
  const int n_d = pick(t.deg_start, d + 1) - row0;
  const int64_t e0 = pick(t.edge_start, d);
  const int64_t slots = (int64_t)n_d * lpr;
  const int64_t first = (int64_t)(b - pick(tt.tile_start, d)) * (kBlock * kUnroll) + threadIdx.x;
#pragma unroll
  for (int u = 0; u < kUnroll; ++u) {
    const int64_t e = first + (int64_t)u * kBlock;
    if (e >= slots) break;
    const int r = (int)(e / lpr);
    const int c = (int)(e - (int64_t)r * lpr) * V;
    const int32_t* nb = col + e0 + (int64_t)r * d;
    VT acc = vzero(VT{});
    int j = 0;
    for (; j + 4 <= d; j += 4) {
      const int k0 = nb[j], k1 = nb[j + 1], k2 = nb[j + 2], k3 = nb[j + 3];
      const VT v0 = ldv<V>(x + (int64_t)k0 * ldx + c);
      const VT v1 = ldv<V>(x + (int64_t)k1 * ldx + c);
      const VT v2 = ldv<V>(x + (int64_t)k2 * ldx + c);
      const VT v3 = ldv<V>(x + (int64_t)k3 * ldx + c);
      acc = vadd(vadd(vadd(vadd(acc, v0), v1), v2), v3);
    }
    for (; j < d; ++j) acc = vadd(acc, ldv<V>(x + (int64_t)nb[j] * ldx + c));
    float* dst = s + (int64_t)(row0 + r) * lds + c;
    if (ACC) acc = vadd(acc, ldv<V>(dst));
    stv<V>(dst, acc);
  }
}

// dx[col[..], :] += ds[i, :]  (general adjacency; float atomics)
template <int V>
__global__ void __launch_bounds__(kBlock)
scatter_add_kernel(DegTable t, TileTable tt, const int32_t* __restrict__ col,
                   const float* __restrict__ ds, int64_t ldds, int lpr, float* __restrict__ dx,
                   int64_t lddx) {
  const int b = blockIdx.x;
  const int d = block_degree(tt, b);
  const int row0 = pick(t.deg_start, d);
  const int n_d = pick(t.deg_start, d + 1) - row0;
  const int64_t e0 = pick(t.edge_start, d);
  const int64_t slots = (int64_t)n_d * lpr;
  const int64_t first = (int64_t)(b - pick(tt.tile_start, d)) * (kBlock * kUnroll) + threadIdx.x;
#pragma unroll
  for (int u = 0; u < kUnroll; ++u) {
    const int64_t e = first + (int64_t)u * kBlock;
    if (e >= slots) break;
    const int r = (int)(e / lpr);
    const int c = (int)(e - (int64_t)r * lpr) * V;
    const int32_t* nb = col + e0 + (int64_t)r * d;
    float v[V];
    ld_arr<V>(ds + (int64_t)(row0 + r) * ldds + c, v);
    for (int j = 0; j < d; ++j) {
      float* dst = dx + (int64_t)nb[j] * lddx + c;
#pragma unroll
      for (int q = 0; q < V; ++q) atomicAdd(dst + q, v[q]);
    }
  }
}

// ------------------------------------------------------------------ gather-max
template <int V, bool BN>
__global__ void __launch_bounds__(kBlock)
gather_max_kernel(DegTable t, TileTable tt, const int32_t* __restrict__ col,
                  const float* __restrict__ x, int64_t ldx, int lpr, int n_feat,
                  const float* __restrict__ scale, const float* __restrict__ shift,
                  float* __restrict__ out, int64_t ldo, uint8_t* __restrict__ arg) {
  const int b = blockIdx.x;
  const int d = block_degree(tt, b);
  const int row0 = pick(t.deg_start, d);
  const int n_d = pick(t.deg_start, d + 1) - row0;
  const int64_t e0 = pick(t.edge_start, d);
  const int64_t slots = (int64_t)n_d * lpr;
  const int64_t first = (int64_t)(b - pick(tt.tile_start, d)) * (kBlock * kUnroll) + threadIdx.x;
#pragma unroll
  for (int u = 0; u < kUnroll; ++u) {
    const int64_t e = first + (int64_t)u * kBlock;
    if (e >= slots) break;
    const int r = (int)(e / lpr);
    const int c = (int)(e - (int64_t)r * lpr) * V;
    const int i = row0 + r;
    const int32_t* nb = col + e0 + (int64_t)r * d;
    float sc[V], sh[V], best[V];
    uint8_t ba[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      sc[q] = BN ? scale[c + q] : 1.f;
      sh[q] = BN ? shift[c + q] : 0.f;
    }
    {
      float v[V];
      ld_arr<V>(x + (int64_t)i * ldx + c, v);
#pragma unroll
      for (int q = 0; q < V; ++q) {
        best[q] = BN ? fmaf(v[q], sc[q], sh[q]) : v[q];
        ba[q] = 0;
      }
    }
    int j = 0;
    for (; j + 2 <= d; j += 2) {
      float v0[V], v1[V];
      const int k0 = nb[j], k1 = nb[j + 1];
      ld_arr<V>(x + (int64_t)k0 * ldx + c, v0);
      ld_arr<V>(x + (int64_t)k1 * ldx + c, v1);
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const float a0 = BN ? fmaf(v0[q], sc[q], sh[q]) : v0[q];
        const float a1 = BN ? fmaf(v1[q], sc[q], sh[q]) : v1[q];
        if (a0 > best[q]) { best[q] = a0; ba[q] = (uint8_t)(j + 1); }
        if (a1 > best[q]) { best[q] = a1; ba[q] = (uint8_t)(j + 2); }
      }
    }
    for (; j < d; ++j) {
      float v0[V];
      ld_arr<V>(x + (int64_t)nb[j] * ldx + c, v0);
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const float a0 = BN ? fmaf(v0[q], sc[q], sh[q]) : v0[q];
        if (a0 > best[q]) { best[q] = a0; ba[q] = (uint8_t)(j + 1); }
      }
    }
    st_arr<V>(out + (int64_t)i * ldo + c, best);
    if (arg) {
      uint8_t* ap = arg + (int64_t)i * n_feat + c;
      if constexpr (V == 4) {
        *reinterpret_cast<uchar4*>(ap) = make_uchar4(ba[0], ba[1], ba[2], ba[3]);
      } else {
        ap[0] = ba[0];
      }
    }
  }
}

template <int V>
__global__ void __launch_bounds__(kBlock)
gather_max_bwd_kernel(DegTable t, TileTable tt, const int32_t* __restrict__ col,
                      const float* __restrict__ dout, int64_t lddo, int lpr, int n_feat,
                      const uint8_t* __restrict__ arg, float* __restrict__ dx, int64_t lddx) {
  const int b = blockIdx.x;
  const int d = block_degree(tt, b);
  const int row0 = pick(t.deg_start, d);
  const int n_d = pick(t.deg_start, d + 1) - row0;
  const int64_t e0 = pick(t.edge_start, d);
  const int64_t slots = (int64_t)n_d * lpr;
  const int64_t first = (int64_t)(b - pick(tt.tile_start, d)) * (kBlock * kUnroll) + threadIdx.x;
#pragma unroll
  for (int u = 0; u < kUnroll; ++u) {
    const int64_t e = first + (int64_t)u * kBlock;
    if (e >= slots) break;
    const int r = (int)(e / lpr);
    const int c = (int)(e - (int64_t)r * lpr) * V;
    const int i = row0 + r;
    const int32_t* nb = col + e0 + (int64_t)r * d;
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const float gq = dout[(int64_t)i * lddo + c + q];
      const int a = arg[(int64_t)i * n_feat + c + q];
      const int target = (a == 0) ? i : nb[a - 1];
      atomicAdd(dx + (int64_t)target * lddx + c + q, gq);
    }
  }
}

// Transposed (gather) form of the GraphPool backward for a symmetric adjacency:
//   dx[k,f] = dout[k,f]*[arg[k,f]==0] + sum_j dout[i_j,f]*[arg[i_j,f] == rev_pos(k,j)+1]
// with i_j the j-th neighbour of k.  Every dx row is produced by exactly one lane
// group: no atomics, no pre-zeroing, bitwise reproducible.  Traffic per edge:
// 16 B of dout + 4 B of arg + 5 B of index/slot per 16-byte column chunk.
template <int V>
__global__ void __launch_bounds__(kBlock)
gather_max_bwd_gather_kernel(DegTable t, TileTable tt, const int32_t* __restrict__ col,
                             const uint8_t* __restrict__ rev, const float* __restrict__ dout,
                             int64_t lddo, int lpr, int n_feat, const uint8_t* __restrict__ arg,
                             float* __restrict__ dx, int64_t lddx) {
  const int b = blockIdx.x;
  const int d = block_degree(tt, b);
  const int row0 = pick(t.deg_start, d);
  const int n_d = pick(t.deg_start, d + 1) - row0;
  const int64_t e0 = pick(t.edge_start, d);
  const int64_t slots = (int64_t)n_d * lpr;
  const int64_t first = (int64_t)(b - pick(tt.tile_start, d)) * (kBlock * kUnroll) + threadIdx.x;
#pragma unroll
  for (int u = 0; u < kUnroll; ++u) {
    const int64_t e = first + (int64_t)u * kBlock;
    if (e >= slots) break;
    const int r = (int)(e / lpr);
    const int c = (int)(e - (int64_t)r * lpr) * V;
    const int k = row0 + r;
    const int32_t* nb = col + e0 + (int64_t)r * d;
    const uint8_t* rp = rev + e0 + (int64_t)r * d;
    float acc[V], gv[V];
    uint8_t av[V];
    ld_arr<V>(dout + (int64_t)k * lddo + c, gv);
    if constexpr (V == 4) {
      const uchar4 a4 = *reinterpret_cast<const uchar4*>(arg + (int64_t)k * n_feat + c);
      av[0] = a4.x; av[1] = a4.y; av[2] = a4.z; av[3] = a4.w;
    } else {
      av[0] = arg[(int64_t)k * n_feat + c];
    }
#pragma unroll
    for (int q = 0; q < V; ++q) acc[q] = av[q] == 0 ? gv[q] : 0.f;
    for (int j = 0; j < d; ++j) {
      const int i = nb[j];
      const uint8_t want = (uint8_t)(rp[j] + 1);
      ld_arr<V>(dout + (int64_t)i * lddo + c, gv);
      if constexpr (V == 4) {
        const uchar4 a4 = *reinterpret_cast<const uchar4*>(arg + (int64_t)i * n_feat + c);
        av[0] = a4.x; av[1] = a4.y; av[2] = a4.z; av[3] = a4.w;
      } else {
        av[0] = arg[(int64_t)i * n_feat + c];
      }
#pragma unroll
      for (int q = 0; q < V; ++q) acc[q] += av[q] == want ? gv[q] : 0.f;
    }
    st_arr<V>(dx + (int64_t)k * lddx + c, acc);
  }
}

// rev_pos[(k, j)] = slot of k inside the neighbour list of i = col[(k, j)]; the n-th
// slot of k that points at i is matched with the n-th slot of i that points at k, so
// multi-bonds pair up one to one.  One thread per edge slot.
__global__ void __launch_bounds__(kBlock)
rev_pos_kernel(DegTable t, int n_edges, const int32_t* __restrict__ col,
               uint8_t* __restrict__ rev, int32_t* __restrict__ flag) {
  for (int e = blockIdx.x * kBlock + threadIdx.x; e < n_edges; e += gridDim.x * kBlock) {
    int d = 0;
#pragma unroll
    for (int q = 1; q <= GCMI_MAX_DEG; ++q) d += (q <= t.max_deg && e >= t.edge_start[q]) ? 1 : 0;
    int es = 0, ds = 0;
#pragma unroll
    for (int q = 0; q <= GCMI_MAX_DEG; ++q) {
      if (q == d) { es = t.edge_start[q]; ds = t.deg_start[q]; }
    }
    const int r = (e - es) / d;
    const int j = (e - es) - r * d;
    const int k = ds + r;
    const int i = col[e];
    int nth = 0;
    for (int q = 0; q < j; ++q) nth += (col[es + r * d + q] == i) ? 1 : 0;
    const int di = degree_of_row(t, i);
    int esi = 0, dsi = 0;
#pragma unroll
    for (int q = 0; q <= GCMI_MAX_DEG; ++q) {
      if (q == di) { esi = t.edge_start[q]; dsi = t.deg_start[q]; }
    }
    const int32_t* li = col + esi + (int64_t)(i - dsi) * di;
    int found = -1;
    for (int p = 0; p < di; ++p) {
      if (li[p] == k) {
        if (nth == 0) { found = p; break; }
        --nth;
      }
    }
    if (found < 0) {
      *flag = 1;
      found = 255;
    }
    rev[e] = (uint8_t)found;
  }
}

}  // namespace gcmi

using namespace gcmi;

extern "C" {

int gcmi_gather_sum_fwd(const gcmi_graph* g, const float* d_x, int64_t ldx, int32_t n_feat,
                        float* d_s, int64_t lds, int32_t accumulate, void* stream) {
  int rc = check_graph(g, true);
  if (rc) return rc;
  GCMI_CHECK_ARG(n_feat > 0 && ldx >= n_feat && lds >= n_feat, "gather_sum: bad n_feat/ld");
  GCMI_CHECK_ARG(g->n_atoms == 0 || (d_x && d_s), "gather_sum: NULL buffer");
  if (g->n_atoms == 0) return GCMI_OK;
  hipStream_t st = (hipStream_t)stream;
  const int V = (vec_width(d_x, ldx, n_feat) == 4 && vec_width(d_s, lds, n_feat) == 4) ? 4 : 1;
  if (V == 4 && win_has_width(n_feat) && win_usable(g, n_feat, false)) {
    TimedScope ts(GCMI_K_GATHER_SUM, st);
    return win_gather_sum(g, d_x, ldx, n_feat, d_s, lds, st, accumulate != 0);
  }
  const int lpr = n_feat / V;
  TileTable tt;
  // accumulate: degree-0 rows receive nothing, so their tiles are skipped
  const int tiles = make_tiles(g, lpr, accumulate != 0, &tt);
  if (tiles == 0) return GCMI_OK;
  DegTable t = make_deg_table(g);
  TimedScope ts(GCMI_K_GATHER_SUM, st);
#define LAUNCH_GS(VV, AA)                                                                     \
  hipLaunchKernelGGL((gather_sum_kernel<VV, AA>), dim3(tiles), dim3(kBlock), 0, st, t, tt,     \
                     g->d_col_idx, d_x, ldx, lpr, d_s, lds)
  if (V == 4) {
    if (accumulate) LAUNCH_GS(4, true); else LAUNCH_GS(4, false);
  } else {
    if (accumulate) LAUNCH_GS(1, true); else LAUNCH_GS(1, false);
  }
#undef LAUNCH_GS
  GCMI_CHECK_LAUNCH("gather_sum_fwd");
  return GCMI_OK;
}

int gcmi_scatter_add(const gcmi_graph* g, const float* d_ds, int64_t ldds, int32_t n_feat,
                     float* d_dx, int64_t lddx, void* stream) {
  int rc = check_graph(g, true);
  if (rc) return rc;
  GCMI_CHECK_ARG(n_feat > 0 && ldds >= n_feat && lddx >= n_feat, "scatter_add: bad n_feat/ld");
  GCMI_CHECK_ARG(g->n_atoms == 0 || (d_ds && d_dx), "scatter_add: NULL buffer");
  hipStream_t st = (hipStream_t)stream;
  const int V = (vec_width(d_ds, ldds, n_feat) == 4) ? 4 : 1;
  const int lpr = n_feat / V;
  TileTable tt;
  const int tiles = make_tiles(g, lpr, true, &tt);
  if (tiles == 0) return GCMI_OK;
  DegTable t = make_deg_table(g);
  if (V == 4)
    hipLaunchKernelGGL(scatter_add_kernel<4>, dim3(tiles), dim3(kBlock), 0, st, t, tt,
                       g->d_col_idx, d_ds, ldds, lpr, d_dx, lddx);
  else
    hipLaunchKernelGGL(scatter_add_kernel<1>, dim3(tiles), dim3(kBlock), 0, st, t, tt,
                       g->d_col_idx, d_ds, ldds, lpr, d_dx, lddx);
  GCMI_CHECK_LAUNCH("scatter_add");
  return GCMI_OK;
}

int gcmi_gather_max_fwd(const gcmi_graph* g, const float* d_x, int64_t ldx, int32_t n_feat,
                        const float* d_scale, const float* d_shift, float* d_out, int64_t ldo,
                        uint8_t* d_arg, void* stream) {
  int rc = check_graph(g, true);
  if (rc) return rc;
  GCMI_CHECK_ARG(n_feat > 0 && ldx >= n_feat && ldo >= n_feat, "gather_max: bad n_feat/ld");
  GCMI_CHECK_ARG(g->n_atoms == 0 || (d_x && d_out), "gather_max: NULL buffer");
  GCMI_CHECK_ARG((d_scale == nullptr) == (d_shift == nullptr), "gather_max: scale/shift must come together");
  if (g->n_atoms == 0) return GCMI_OK;
  hipStream_t st = (hipStream_t)stream;
  const int V = (vec_width(d_x, ldx, n_feat) == 4 && vec_width(d_out, ldo, n_feat) == 4 &&
                 (d_arg == nullptr || (reinterpret_cast<uintptr_t>(d_arg) & 3u) == 0))
                    ? 4
                    : 1;
  if (V == 4 && win_has_width(n_feat) && win_usable(g, n_feat, false) &&
      (d_scale == nullptr || (aligned16(d_scale) && aligned16(d_shift)))) {
    TimedScope ts(GCMI_K_GATHER_MAX, st);
    return win_gather_max(g, d_x, ldx, n_feat, d_scale, d_shift, d_out, ldo, d_arg, st);
  }
  const int lpr = n_feat / V;
  TileTable tt;
  const int tiles = make_tiles(g, lpr, false, &tt);
  DegTable t = make_deg_table(g);
  const bool bn = d_scale != nullptr;
  TimedScope ts(GCMI_K_GATHER_MAX, st);
#define LAUNCH_GM(VV, BB)                                                                       \
  hipLaunchKernelGGL((gather_max_kernel<VV, BB>), dim3(tiles), dim3(kBlock), 0, st, t, tt,      \
                     g->d_col_idx, d_x, ldx, lpr, n_feat, d_scale, d_shift, d_out, ldo, d_arg)
  if (V == 4) {
    if (bn) LAUNCH_GM(4, true); else LAUNCH_GM(4, false);
  } else {
    if (bn) LAUNCH_GM(1, true); else LAUNCH_GM(1, false);
  }
#undef LAUNCH_GM
  GCMI_CHECK_LAUNCH("gather_max_fwd");
  return GCMI_OK;
}

int gcmi_build_rev_pos(const gcmi_graph* g, uint8_t* d_rev_pos, int32_t* d_flag, void* stream) {
  int rc = check_graph(g, true);
  if (rc) return rc;
  GCMI_CHECK_ARG(d_flag != nullptr, "build_rev_pos: d_flag is NULL");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(d_flag, 0, sizeof(int32_t), st) != hipSuccess) {
    set_error("build_rev_pos: memset failed");
    return GCMI_ERR_LAUNCH;
  }
  if (g->n_edges == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_rev_pos != nullptr, "build_rev_pos: NULL output");
  DegTable t = make_deg_table(g);
  hipLaunchKernelGGL(rev_pos_kernel, dim3(grid_for(g->n_edges, kBlock)), dim3(kBlock), 0, st, t,
                     g->n_edges, g->d_col_idx, d_rev_pos, d_flag);
  GCMI_CHECK_LAUNCH("build_rev_pos");
  return GCMI_OK;
}

int gcmi_gather_max_bwd(const gcmi_graph* g, const float* d_dout, int64_t lddo, int32_t n_feat,
                        const uint8_t* d_arg, float* d_dx, int64_t lddx, void* stream) {
  int rc = check_graph(g, true);
  if (rc) return rc;
  GCMI_CHECK_ARG(n_feat > 0 && lddo >= n_feat && lddx >= n_feat, "gather_max_bwd: bad n_feat/ld");
  GCMI_CHECK_ARG(g->n_atoms == 0 || (d_dout && d_arg && d_dx), "gather_max_bwd: NULL buffer");
  if (g->n_atoms == 0) return GCMI_OK;
  hipStream_t st = (hipStream_t)stream;
  const int V = (n_feat % 4 == 0 && vec_width(d_dout, lddo, n_feat) == 4 &&
                 vec_width(d_dx, lddx, n_feat) == 4 && (reinterpret_cast<uintptr_t>(d_arg) & 3u) == 0)
                    ? 4
                    : 1;
  const int lpr = n_feat / V;
  TileTable tt;
  const int tiles = make_tiles(g, lpr, false, &tt);
  DegTable t = make_deg_table(g);
  if (V == 4 && (g->d_rev_pos != nullptr || g->n_edges == 0) && win_has_width(n_feat) &&
      win_usable(g, n_feat, true)) {
    TimedScope ts(GCMI_K_GATHER_MAX_BWD, st);
    return win_gather_max_bwd(g, d_dout, lddo, n_feat, d_arg, d_dx, lddx, st);
  }
  if (g->d_rev_pos != nullptr || g->n_edges == 0) {
    TimedScope ts(GCMI_K_GATHER_MAX_BWD, st);
    if (V == 4)
      hipLaunchKernelGGL(gather_max_bwd_gather_kernel<4>, dim3(tiles), dim3(kBlock), 0, st, t, tt,
                         g->d_col_idx, g->d_rev_pos, d_dout, lddo, lpr, n_feat, d_arg, d_dx, lddx);
    else
      hipLaunchKernelGGL(gather_max_bwd_gather_kernel<1>, dim3(tiles), dim3(kBlock), 0, st, t, tt,
                         g->d_col_idx, g->d_rev_pos, d_dout, lddo, lpr, n_feat, d_arg, d_dx, lddx);
    GCMI_CHECK_LAUNCH("gather_max_bwd (gather form)");
    return GCMI_OK;
  }
  if (V == 4)
    hipLaunchKernelGGL(gather_max_bwd_kernel<4>, dim3(tiles), dim3(kBlock), 0, st, t, tt,
                       g->d_col_idx, d_dout, lddo, lpr, n_feat, d_arg, d_dx, lddx);
  else
    hipLaunchKernelGGL(gather_max_bwd_kernel<1>, dim3(tiles), dim3(kBlock), 0, st, t, tt,
                       g->d_col_idx, d_dout, lddo, lpr, n_feat, d_arg, d_dx, lddx);
  GCMI_CHECK_LAUNCH("gather_max_bwd");
  return GCMI_OK;
}

}  // extern "C"
