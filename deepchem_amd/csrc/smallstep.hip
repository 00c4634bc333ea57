// The small-batch engine: one optimizer step of GraphConvModel as 8 launches (reference gradient
// semantics) or 12 (full), for batches whose activations live in L2 (the reference's default batch of
// 100 molecules is ~1 850 atoms; MolNet's Tox21 preset uses 64).
//
// Why a second execution model.  The large-batch kernels of this library (gather_lds.hip, gemm_split.hip,
// bn.hip, readout.hip) are streaming kernels: 128-row tiles, persistent workgroups, one pass per operand.
// At 2 000 atoms they occupy ~15 of 256 CUs and the step is ~45 launches of a few microseconds each,
// bounded by launch cadence.  Here the unit of work is a 16-row tile of ONE degree block (so the per-degree
// weights of GraphConv, models/torch_models/layers.py:6199-6229, are workgroup-uniform): one workgroup of
// four waves per tile, each wave owning 16 output columns per pass on v_mfma_f32_16x16x4_f32 (fp32 operands,
// fp32 accumulate: the reference's arithmetic), operands staged through LDS, everything else of a layer --
// neighbour gather, bias, ReLU, BatchNorm statistics, folded BatchNorm, pooling, readout, task head, loss,
// their backwards -- fused around it.  A kernel boundary is needed only where the model has a batch-wide
// dependency (BatchNorm statistics; neighbour rows written by other tiles):
//
//   forward  conv0 | pool0 | conv1 | pool1+dense | readout+head+loss(+d head input)
//   backward dense (dgrad tiles + weight-gradient slabs + head weight gradient) | pool1 (+BatchNorm 1 grads)
//            [full mode: conv1 | dP0 | pool0 | conv0]
//   adam     (+ loss, running statistics, counters, zeroing for the next step)
//
// Reference functions replaced: GraphConv.forward + sum_neigh (layers.py:6167-6246), GraphPool.forward
// (:6319-6367), GraphGather.forward (:6450-6479), nn.BatchNorm1d / nn.Linear of _GraphConvTorchModel.forward
// (graphconvmodel.py:188-249), SoftmaxCrossEntropy / L2Loss through _StandardLoss (losses.py:251-259, :85-94;
// torch_model.py:1275-1294), torch.optim.Adam (optimizers.py:231-241), the step of fit_generator
// (torch_model.py:435-443), over MANY batches per call: the host loop here replaces the Python loop.
#include <math.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace gcmi {

typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int kTileRows = 16;
constexpr int kSlabRows = 32;
constexpr int kSBlock = 256;
constexpr int kMaxL = GCMI_MAX_CONV_LAYERS;
constexpr int kAhead = 8;            // conv stacks computed ahead per group (reference gradient mode)
constexpr int kSlots = 2 * kAhead;   // two sets of them

// A collated batch as the kernels see it (by value in the kernarg segment).
struct SmallGraph {
  int32_t n_atoms, n_mols, max_deg, n_tiles;
  int32_t diag;  // GCMI_SMALL_DIAG: parts of the kernels switched off for timing experiments (results are wrong then)
  int32_t bf16;  // activations the step writes (GraphConv outputs, pooled rows, dense output) are stored as bf16
  int32_t deg_start[GCMI_MAX_DEG + 2];
  int32_t edge_start[GCMI_MAX_DEG + 2];
  int32_t tile_start[GCMI_MAX_DEG + 2];  // 16-row tiles per degree block, prefix
  const int32_t* col_idx;
  const int32_t* membership;
  const int32_t* mol_runs;
  const uint8_t* rev_pos;
};

struct Tile {
  int d, row0, nrows, e0;  // degree, first row, rows (<= 16), first edge of row0
};

__device__ __forceinline__ Tile tile_of(const SmallGraph& g, int t) {
  int d = 0;
#pragma unroll
  for (int k = 1; k <= GCMI_MAX_DEG; ++k) d += (k <= g.max_deg && t >= g.tile_start[k]) ? 1 : 0;
  Tile tl;
  tl.d = d;
  tl.row0 = g.deg_start[d] + (t - g.tile_start[d]) * kTileRows;
  const int left = g.deg_start[d + 1] - tl.row0;
  tl.nrows = left < kTileRows ? left : kTileRows;
  tl.e0 = g.edge_start[d] + (tl.row0 - g.deg_start[d]) * d;
  return tl;
}

__device__ __forceinline__ f4v mfma16(float a, float b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// LDS pitch (floats) of a K-column operand tile read as MFMA A fragments (lane = row + 16*kq reads
// [row][4*ks + kq]): conflict-free when pitch/4 is odd.
__host__ __device__ inline int pitch_a(int k4) { return ((k4 / 4) & 1) ? k4 : k4 + 4; }
// pitch for tiles read "transposed" (lane = col + 16*kq reads [4*ks + kq][col0 + col]): pitch % 64 == 16
__host__ __device__ inline int pitch_t(int n) { return ((n + 63) / 64) * 64 + 16; }

// BatchNorm coefficients of one column from the accumulated sums (training) or the running statistics (eval)
struct BnCol {
  float mean, invstd, scale, shift;
};
__device__ __forceinline__ BnCol bn_col_train(const double* __restrict__ acc, int width, int c, double inv_n,
                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                              float eps) {
  // sums in fp64 (E[x^2] - mean^2 cancels), everything after the subtraction in fp32: no fp64 divide or root
  const double mean = acc[c] * inv_n;
  const float var = fmaxf((float)(acc[width + c] * inv_n - mean * mean), 0.f);
  BnCol b;
  b.mean = (float)mean;
  b.invstd = 1.0f / sqrtf(var + eps);
  b.scale = gamma[c] * b.invstd;
  b.shift = beta[c] - b.mean * b.scale;
  return b;
}
__device__ __forceinline__ BnCol bn_col_eval(const float* __restrict__ rm, const float* __restrict__ rv, int c,
                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                             float eps) {
  BnCol b;
  b.mean = rm[c];
  b.invstd = 1.0f / sqrtf(rv[c] + eps);
  b.scale = gamma[c] * b.invstd;
  b.shift = beta[c] - b.mean * b.scale;
  return b;
}

struct BnArgs {          // how a kernel obtains the folded BatchNorm of its input
  int mode;              // 0 none, 1 batch statistics from acc, 2 running statistics
  const double* acc;     // [sum(width) | sum of squares(width)]
  const float* rm;
  const float* rv;
  const float* gamma;
  const float* beta;
  float eps;
  int n_rows;
  double inv_n;          // 1 / n_rows
};

__device__ __forceinline__ BnCol bn_col(const BnArgs& a, int width, int c) {
  if (a.mode == 1) return bn_col_train(a.acc, width, c, a.inv_n, a.gamma, a.beta, a.eps);
  if (a.mode == 2) return bn_col_eval(a.rm, a.rv, c, a.gamma, a.beta, a.eps);
  BnCol b;
  b.mean = 0.f;
  b.invstd = 1.f;
  b.scale = 1.f;
  b.shift = 0.f;
  return b;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---- activation storage.  gcmi_model_desc.storage = 1 ("bf16 storage"): every matrix the step WRITES
// and reads back -- GraphConv outputs, pooled rows, the dense output -- is kept as bfloat16 (round to nearest even),
// half the bytes per row; all arithmetic stays fp32 (operands widen on load, products and sums accumulate in fp32,
// BatchNorm sums in fp64, parameters / gradients / optimizer state fp32).  The pointers keep their float type in the
// signatures; with `bf` set the same buffer is addressed as 16-bit elements.
__device__ __forceinline__ float bf2f(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t f2bf(float f) {
  const uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;  // NaN stays NaN
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float round_act(int bf, float v) { return bf ? bf2f(f2bf(v)) : v; }
// four consecutive columns 4q..4q+3 of row `row` (leading dimension ld, in elements)
__device__ __forceinline__ float4 ldA(int bf, const float* p, int64_t row, int64_t ld, int q) {
  if (!bf) return ld4(p + row * ld + 4 * q);
  const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p) + row * ld + 4 * q);
  return make_float4(bf2f(w.x & 0xffffu), bf2f(w.x >> 16), bf2f(w.y & 0xffffu), bf2f(w.y >> 16));
}
__device__ __forceinline__ void stA(int bf, float* p, int64_t row, int64_t ld, int q, float4 v) {
  if (!bf) {
    st4(p + row * ld + 4 * q, v);
    return;
  }
  uint2 w;
  w.x = f2bf(v.x) | (f2bf(v.y) << 16);
  w.y = f2bf(v.z) | (f2bf(v.w) << 16);
  *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p) + row * ld + 4 * q) = w;
}
__device__ __forceinline__ void stA1(int bf, float* p, int64_t idx, float v) {  // v already rounded by round_act
  if (!bf)
    p[idx] = v;
  else
    reinterpret_cast<uint16_t*>(p)[idx] = (uint16_t)(__float_as_uint(v) >> 16);
}

// ------------------------------------------------------------------------------------------------ pooled row chunk
// One row of GraphPool over the folded BatchNorm of gc: max over {self} U neighbours, first maximum wins
// (self, then neighbours in table order: layers.py:6353-6361; torch.max(dim) tie rule).  arg: 0 = self, j+1.
// Folded BatchNorm of the four columns of quad q, in registers: every thread derives the coefficients of ITS
// columns from the accumulated sums itself (the loads go out together with the thread's first gather loads; a
// fold through LDS would put a barrier and one more memory round trip in front of them).
struct BnQuad {
  float4 mean, invstd, scale, shift;
};
__device__ __forceinline__ BnQuad bn_quad(const BnArgs& bn, int W, int q) {
  BnQuad o;
  float* m = reinterpret_cast<float*>(&o.mean);
  float* iv = reinterpret_cast<float*>(&o.invstd);
  float* sc = reinterpret_cast<float*>(&o.scale);
  float* sh = reinterpret_cast<float*>(&o.shift);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const BnCol b = bn_col(bn, W, 4 * q + e);
    m[e] = b.mean;
    iv[e] = b.invstd;
    sc[e] = b.scale;
    sh[e] = b.shift;
  }
  return o;
}

// A dependent global load costs about a microsecond here (the operands were written by the previous launch on
// other XCDs, so they come from the Infinity Cache / HBM, not the local L2): gathers therefore issue their index
// loads together, then their row loads together, four neighbours per round, instead of one neighbour per trip.
__device__ __forceinline__ void pool_chunk(const SmallGraph& g, int row, int d, int e, const float* __restrict__ gc,
                                           int W, int q, const float4& sc, const float4& sh, float4& best,
                                           uint32_t& arg) {
  float4 v = ldA(g.bf16, gc, row, W, q);
  const int32_t* nb = g.col_idx + e;
  uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  bool first = true;
  for (int j0 = 0; j0 < d || first; j0 += 4) {
    int id[4];
    float4 u[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) id[t] = j0 + t < d ? nb[j0 + t] : row;
#pragma unroll
    for (int t = 0; t < 4; ++t) u[t] = ldA(g.bf16, gc, id[t], W, q);
    if (first) {
      best = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
      first = false;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (j0 + t < d) {
        const float y0 = fmaf(u[t].x, sc.x, sh.x), y1 = fmaf(u[t].y, sc.y, sh.y), y2 = fmaf(u[t].z, sc.z, sh.z),
                    y3 = fmaf(u[t].w, sc.w, sh.w);
        const uint32_t j = j0 + t + 1;
        if (y0 > best.x) { best.x = y0; a0 = j; }
        if (y1 > best.y) { best.y = y1; a1 = j; }
        if (y2 > best.z) { best.z = y2; a2 = j; }
        if (y3 > best.w) { best.w = y3; a3 = j; }
      }
    }
  }
  arg = a0 | (a1 << 8) | (a2 << 16) | (a3 << 24);
}

// sum of the d neighbour rows (quad q), four rows in flight per round
__device__ __forceinline__ float4 gather_sum_quad(int bf, const float* __restrict__ x, int64_t ld, const int32_t* nb,
                                                  int d, int self_row, int q) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = 0; j0 < d; j0 += 4) {
    int id[4];
    float4 u[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) id[t] = j0 + t < d ? nb[j0 + t] : self_row;
#pragma unroll
    for (int t = 0; t < 4; ++t) u[t] = ldA(bf, x, id[t], ld, q);
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (j0 + t < d) {
        s.x += u[t].x;
        s.y += u[t].y;
        s.z += u[t].z;
        s.w += u[t].w;
      }
  }
  return s;
}

__device__ __forceinline__ int degree_of(const SmallGraph& g, int row) {
  int d = 0;
#pragma unroll
  for (int k = 1; k <= GCMI_MAX_DEG; ++k) d += (k <= g.max_deg && row >= g.deg_start[k]) ? 1 : 0;
  return d;
}

// Own pooled row and the sum of the neighbours' pooled rows (quad q) for an atom of degree d <= 4 whose neighbours
// have degree <= 4 (every atom of an organic molecule but a handful): three rounds of loads -- own neighbour ids |
// neighbour rows + their neighbour ids | second-neighbour rows -- each issued as one batch.  Returns false when a
// neighbour has more than four neighbours (the caller then takes the general path).
__device__ __forceinline__ bool pool_two_hop(const SmallGraph& g, int row, int d, int e, const float* __restrict__ gc,
                                             int W, int q, const float4& sc, const float4& sh, float4& self,
                                             uint32_t& arg, float4& nsum) {
  const int32_t* nb = g.col_idx + e;
  int id[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) id[t] = t < d ? nb[t] : row;
  const float4 own = ldA(g.bf16, gc, row, W, q);
  int dn[4];
  const int32_t* nn[4];
  bool ok = true;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    dn[t] = t < d ? degree_of(g, id[t]) : 0;
    nn[t] = g.col_idx + g.edge_start[dn[t]] + (id[t] - g.deg_start[dn[t]]) * dn[t];
    ok = ok && dn[t] <= 4;
  }
  if (!ok) return false;
  float4 r1[4];
  int id2[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    r1[t] = ldA(g.bf16, gc, id[t], W, q);
#pragma unroll
    for (int k = 0; k < 4; ++k) id2[t][k] = (t < d && k < dn[t]) ? nn[t][k] : row;
  }
  float4 r2[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) r2[t][k] = ldA(g.bf16, gc, id2[t][k], W, q);
  auto bnq = [&](const float4& v) {
    return make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
  };
  self = bnq(own);
  uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  nsum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < d) {
      const float4 y = bnq(r1[t]);
      if (y.x > self.x) { self.x = y.x; a0 = t + 1; }
      if (y.y > self.y) { self.y = y.y; a1 = t + 1; }
      if (y.z > self.z) { self.z = y.z; a2 = t + 1; }
      if (y.w > self.w) { self.w = y.w; a3 = t + 1; }
      float4 p = y;  // pooled row of neighbour t: its own row, then its neighbours
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k < dn[t]) {
          const float4 z = bnq(r2[t][k]);
          p.x = fmaxf(p.x, z.x);
          p.y = fmaxf(p.y, z.y);
          p.z = fmaxf(p.z, z.z);
          p.w = fmaxf(p.w, z.w);
        }
      }
      if (g.bf16) p = make_float4(round_act(1, p.x), round_act(1, p.y), round_act(1, p.z), round_act(1, p.w));
      nsum.x += p.x;
      nsum.y += p.y;
      nsum.z += p.z;
      nsum.w += p.w;
    }
  }
  if (g.bf16) self = make_float4(round_act(1, self.x), round_act(1, self.y), round_act(1, self.z), round_act(1, self.w));
  arg = a0 | (a1 << 8) | (a2 << 16) | (a3 << 24);
  return true;
}

// ------------------------------------------------------------------------------------------------ conv forward
// out[rows of degree d] = relu(S . W_rel[d] + X . W_self[d] + b_rel[d] + b_self[d]),  S = sum of neighbour rows
// (degree 0: X . W_self[0] + b_self[0]); optional column sums of out and out^2 into acc (fp64 atomics).
// W_list order: rel_1, self_1, ..., rel_10, self_10, self_0 (layers.py:6189-6224).
// POOL_IN: the layer input X is the GraphPool of the previous layer's output, computed HERE from gc_prev with
// its folded BatchNorm -- own rows (written to pool_out / arg_out when given: the backward of "full" mode reads
// them) and, recomputed, the rows of the neighbours -- so the pooled matrix needs no launch of its own.
// The weight fragments of a wave (its 16 columns per pass) are fetched a chunk of k-steps ahead into registers --
// the whole K for the default widths -- and the bias with them: they are in flight while the operand tile is
// gathered, so the product starts without a memory round trip of its own.
constexpr int kBChunk = 8;
template <int NT>
struct ConvChunk {
  static constexpr int value = NT == 1 ? 20 : (NT == 2 ? 10 : 5);
};

template <int NT, bool POOL_IN>
__global__ void __launch_bounds__(kSBlock)
small_conv_fwd_kernel(SmallGraph g, const float* __restrict__ x, int ldx, int K, const float* __restrict__ Wl,
                      const float* __restrict__ bl, float* __restrict__ out, double* __restrict__ acc, BnArgs bn_in,
                      float* __restrict__ pool_out, uint8_t* __restrict__ arg_out) {
  if (g.diag & 1024) return;
  extern __shared__ float smem[];
  const int t = blockIdx.x;
  if (t >= g.n_tiles) return;
  Tile tl = tile_of(g, t);
  if (g.diag & 16) tl.d = 0;
  const int W = 64 * NT;
  const int K4 = (K + 3) & ~3;
  const int KP = pitch_a(K4);
  float* sS = smem;                   // [16][KP] neighbour sums
  float* sX = smem + kTileRows * KP;  // [16][KP] own rows
  const int q4 = K4 / 4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lr = lane & 15, kq = lane >> 4;
  const int64_t blk = (int64_t)K * W;
  const float* Wself = Wl + (tl.d == 0 ? (int64_t)(2 * g.max_deg) * blk : (int64_t)(2 * (tl.d - 1) + 1) * blk);
  const float* Wrel = tl.d == 0 ? nullptr : Wl + (int64_t)(2 * (tl.d - 1)) * blk;
  constexpr int CH = ConvChunk<NT>::value;
  float bx[CH][NT], bs[CH][NT], bias[NT];
  auto load_b = [&](int ks0) {
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      int k = 4 * (ks0 + u) + kq;
      k = k < K ? k : K - 1;  // beyond K the A operand is zero (and beyond the loop nothing is used)
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int col = 16 * (wave + 4 * i) + lr;
        bx[u][i] = Wself[(int64_t)k * W + col];
        bs[u][i] = Wrel ? Wrel[(int64_t)k * W + col] : 0.f;
      }
    }
  };
  load_b(0);
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int col = 16 * (wave + 4 * i) + lr;
    bias[i] = tl.d == 0 ? bl[(int64_t)(2 * g.max_deg) * W + col]
                        : bl[(int64_t)(2 * (tl.d - 1)) * W + col] + bl[(int64_t)(2 * (tl.d - 1) + 1) * W + col];
  }
  for (int idx = threadIdx.x; idx < kTileRows * q4; idx += kSBlock) {
    const int r = idx / q4, q = idx - r * q4;
    float4 self = make_float4(0.f, 0.f, 0.f, 0.f), s = self;
    if (r < tl.nrows) {
      const int row = tl.row0 + r;
      const int32_t* nb = g.col_idx + tl.e0 + r * tl.d;
      if (POOL_IN) {
        uint32_t a;
        const BnQuad bq = bn_quad(bn_in, K, q);
        const int e = tl.e0 + r * tl.d;
        if (!(tl.d <= 4 && pool_two_hop(g, row, tl.d, e, x, K, q, bq.scale, bq.shift, self, a, s))) {
          pool_chunk(g, row, tl.d, e, x, K, q, bq.scale, bq.shift, self, a);
          if (g.bf16) self = make_float4(round_act(1, self.x), round_act(1, self.y), round_act(1, self.z), round_act(1, self.w));
          s = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int j = 0; j < tl.d; ++j) {
            const int nr = nb[j];
            const int dn = degree_of(g, nr);
            float4 v;
            uint32_t an;
            pool_chunk(g, nr, dn, g.edge_start[dn] + (nr - g.deg_start[dn]) * dn, x, K, q, bq.scale, bq.shift, v, an);
            if (g.bf16) v = make_float4(round_act(1, v.x), round_act(1, v.y), round_act(1, v.z), round_act(1, v.w));
            s.x += v.x;
            s.y += v.y;
            s.z += v.z;
            s.w += v.w;
          }
        }
        if (pool_out) stA(g.bf16, pool_out, row, K, q, self);
        if (arg_out) *reinterpret_cast<uint32_t*>(arg_out + (int64_t)row * K + 4 * q) = a;
      } else {
        self = ld4(x + (int64_t)row * ldx + 4 * q);  // the atom features arrive as fp32 rows (one-hot: exact anyway)
        s = gather_sum_quad(0, x, ldx, nb, tl.d, row, q);
        if (4 * q + 3 >= K) {  // columns beyond K (alignment padding of the input) never count
          float* sf = reinterpret_cast<float*>(&self);
          float* ss = reinterpret_cast<float*>(&s);
          for (int c = 0; c < 4; ++c)
            if (4 * q + c >= K) sf[c] = ss[c] = 0.f;
        }
      }
    }
    st4(sX + r * KP + 4 * q, self);
    st4(sS + r * KP + 4 * q, s);
  }
  __syncthreads();
  f4v c[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) c[i] = (f4v){0.f, 0.f, 0.f, 0.f};
  for (int ks0 = 0; ks0 < q4 && !(g.diag & 8); ks0 += CH) {
    float cx[CH][NT], cs[CH][NT];
#pragma unroll
    for (int u = 0; u < CH; ++u)
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        cx[u][i] = bx[u][i];
        cs[u][i] = bs[u][i];
      }
    if (ks0 + CH < q4) load_b(ks0 + CH);
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      if (ks0 + u < q4) {
        const int k = 4 * (ks0 + u) + kq;
        const float ax = sX[lr * KP + k];
        const float as = sS[lr * KP + k];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          c[i] = mfma16(ax, cx[u][i], c[i]);
          if (Wrel) c[i] = mfma16(as, cs[u][i], c[i]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int col = 16 * (wave + 4 * i) + lr;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * kq + j;
      if (r < tl.nrows) {
        const float v = round_act(g.bf16, fmaxf(c[i][j] + bias[i], 0.f));  // statistics describe what is stored
        stA1(g.bf16, out, (int64_t)(tl.row0 + r) * W + col, v);
        s1 += v;
        s2 += v * v;
      }
    }
    if (acc && !(g.diag & 1)) {
      s1 += __shfl_xor(s1, 16);
      s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (kq == 0) {
        atomicAdd(acc + col, (double)s1);
        atomicAdd(acc + W + col, (double)s2);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ pool + dense forward
// The last GraphPool fused with the atom-level nn.Linear(K -> D) + ReLU (graphconvmodel.py:222-223): the pooled
// tile goes to LDS (and to HBM with its arg-max, for the backward) and straight into the product.
// Wd is (D, K) row-major (nn.Linear).  Column sums of the output into acc (training).
template <int NT>
__global__ void __launch_bounds__(kSBlock)
small_pool_dense_fwd_kernel(SmallGraph g, const float* __restrict__ gc, int K, BnArgs bn, float* __restrict__ pool,
                            uint8_t* __restrict__ arg, const float* __restrict__ Wd, const float* __restrict__ bd,
                            float* __restrict__ dense, double* __restrict__ acc) {
  if (g.diag & 1024) return;
  extern __shared__ float smem[];
  const int t = blockIdx.x;
  if (t >= g.n_tiles) return;
  Tile tl = tile_of(g, t);
  if (g.diag & 16) tl.d = 0;
  const int D = 64 * NT;
  const int KP = pitch_a(K);
  float* sP = smem;  // [16][KP]
  const int q4 = K / 4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lr = lane & 15, kq = lane >> 4;
  // k order of the product: step (s, j) takes k = 16 s + 4 kq + j from lane group kq (any order sums the same
  // terms; A and B use the same one), so a lane's four B values of a step group are ONE float4 of its row of Wd --
  // a wave load touches 16 rows x 64 contiguous bytes instead of 16 rows x 16 bytes four times over
  const int n_s = K / 16;  // K is a multiple of 64
  constexpr int SG = NT <= 2 ? 4 : 2;  // step groups fetched ahead
  float4 bw[SG][NT];
  auto load_w = [&](int s0) {
#pragma unroll
    for (int u = 0; u < SG; ++u) {
      const int sg = s0 + u < n_s ? s0 + u : n_s - 1;
#pragma unroll
      for (int i = 0; i < NT; ++i) bw[u][i] = ld4(Wd + (int64_t)(16 * (wave + 4 * i) + lr) * K + 16 * sg + 4 * kq);
    }
  };
  load_w(0);  // in flight while the pooled tile is gathered
  float biasd[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) biasd[i] = bd[16 * (wave + 4 * i) + lr];
  for (int idx = threadIdx.x; idx < kTileRows * q4; idx += kSBlock) {
    const int r = idx / q4, q = idx - r * q4;
    float4 best = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < tl.nrows) {
      uint32_t a;
      const BnQuad bq = bn_quad(bn, K, q);
      pool_chunk(g, tl.row0 + r, tl.d, tl.e0 + r * tl.d, gc, K, q, bq.scale, bq.shift, best, a);
      stA(g.bf16, pool, tl.row0 + r, K, q, best);
      if (g.bf16) best = ldA(1, pool, tl.row0 + r, K, q);  // the product sees the stored (rounded) rows, like the backward will
      if (arg) *reinterpret_cast<uint32_t*>(arg + (int64_t)(tl.row0 + r) * K + 4 * q) = a;
    }
    st4(sP + r * KP + 4 * q, best);
  }
  __syncthreads();
  f4v c[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) c[i] = (f4v){0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < n_s && !(g.diag & 8); s0 += SG) {
    float4 cw[SG][NT];
#pragma unroll
    for (int u = 0; u < SG; ++u)
#pragma unroll
      for (int i = 0; i < NT; ++i) cw[u][i] = bw[u][i];
    if (s0 + SG < n_s) load_w(s0 + SG);
#pragma unroll
    for (int u = 0; u < SG; ++u) {
      if (s0 + u < n_s) {
        const float4 a4 = ld4(sP + lr * KP + 16 * (s0 + u) + 4 * kq);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          c[i] = mfma16(a4.x, cw[u][i].x, c[i]);
          c[i] = mfma16(a4.y, cw[u][i].y, c[i]);
          c[i] = mfma16(a4.z, cw[u][i].z, c[i]);
          c[i] = mfma16(a4.w, cw[u][i].w, c[i]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int col = 16 * (wave + 4 * i) + lr;
    const float bias = biasd[i];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * kq + j;
      if (r < tl.nrows) {
        const float v = round_act(g.bf16, fmaxf(c[i][j] + bias, 0.f));
        stA1(g.bf16, dense, (int64_t)(tl.row0 + r) * D + col, v);
        s1 += v;
        s2 += v * v;
      }
    }
    if (acc && !(g.diag & 1)) {
      s1 += __shfl_xor(s1, 16);
      s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (kq == 0) {
        atomicAdd(acc + col, (double)s1);
        atomicAdd(acc + D + col, (double)s2);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ readout + head + loss
// One wave per molecule: GraphGather = [segment sum | segment max] of the folded BatchNorm of the dense output,
// tanh (layers.py:6450-6479, graphconvmodel.py:174-175); task head (nn.Linear(2F, T*C), (T*C, 2F) row-major);
// softmax (classification predictions); in training the loss term of every (molecule, task), d loss / d logits,
// the gradient w.r.t. the fingerprint through tanh, and the column sums the BatchNorm backward of the dense layer
// needs -- sum_r dy and sum_r dy * xhat -- formed per molecule from what the wave already holds:
//   dy[r] = gs + [r == arg] gm   =>   sum_r dy = n gs + gm,   sum_r dy xhat = gs sum_r xhat + gm xhat[arg].
struct ReadoutArgs {
  const float* dense;   // N x F
  BnArgs bn;
  const float* Wh;      // TC x 2F
  const float* bh;      // TC
  float* fp;            // B x 2F   (output: embedding)
  float* logits;        // B x TC
  float* probs;         // B x TC or NULL
  // training only (labels == NULL: prediction)
  const float* labels;  // B x T x C (classification one-hot) or B x T
  const float* weights; // B x T or NULL
  int n_rows;           // molecules that count in the loss
  float inv_count;      // 1 / (n_rows * T)
  float* dlogits;       // B x TC
  float* g2;            // B x 2F: gradient w.r.t. the gather output (before tanh)
  int32_t* argrow;      // B x F: row of the maximum (-1: empty molecule)
  double* loss_acc;     // 1
  double* bsum;         // [sum dy (F) | sum dy*xhat (F)]
  int F, T, C, mode;    // mode 0 classification, 1 regression
  int wh_in_lds;        // the head matrix fits into LDS beside the per-wave scratch: staged once per workgroup
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// One workgroup per molecule.  Lanes along the feature axis in float4 pieces: LPR = F / 4 lanes cover one row,
// 64 / LPR rows per wave instruction; each of the four waves takes a quarter of the molecule's rows (<= 11
// contiguous runs, enumerated up front so that eight wave-loads of rows are in flight together), the quarters
// meet in LDS.  Head, loss and the gradient of the fingerprint are spread over all 256 threads, and every piece
// of the head matrix a thread will need is requested at the top of the kernel: the kernel's memory round trips
// are (1) parameters + run bounds, (2) rows.
template <int LPR>  // F = 4 * LPR in {64, 128, 256}
__global__ void __launch_bounds__(kSBlock)
small_readout_kernel(SmallGraph g, ReadoutArgs a) {
  if (g.diag & 1024) return;
  extern __shared__ float smem[];
  constexpr int F = 4 * LPR, RPW = 64 / LPR, F2 = 2 * F;
  constexpr int NH4 = F2 / 32;       // float4 pieces of a head row per thread (8 column segments)
  constexpr int SEG = F2 / 8;        // columns per segment
  constexpr int ND = 6;              // head rows per thread fetched ahead for the fingerprint gradient
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m = blockIdx.x;
  const int TC = a.T * a.C, TCp = (TC + 3) & ~3;
  float* sfp = smem;                  // [2F] fingerprint
  float* slog = sfp + F2;             // [TCp] logits
  float* sdl = slog + TCp;            // [TCp] d loss / d logits
  float* sgr = sdl + TCp;             // [2F] gradient w.r.t. the gather output
  float* comb = sgr + F2;             // [4 waves][5][F]
  float* part = comb + 20 * F;        // [8][TCp] head partial sums
  float* gpart = part + 8 * TCp;      // [4][2F] gradient partial sums
  const bool train = a.labels != nullptr;
  // ---- requests that do not depend on the rows
  const int tcl = tid & 31, fseg = tid >> 5;
  float4 wh[NH4];
#pragma unroll
  for (int j = 0; j < NH4; ++j)
    wh[j] = tcl < TC ? ld4(a.Wh + (int64_t)tcl * F2 + fseg * SEG + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int f4 = tid & 63, tcg = tid >> 6;
  const int TQ = (TC + 3) / 4, tc_lo = tcg * TQ, tc_hi = min(TC, tc_lo + TQ);
  float4 wd[ND];
  if (train) {
#pragma unroll
    for (int j = 0; j < ND; ++j)
      wd[j] = (tc_lo + j < tc_hi && 4 * f4 < F2) ? ld4(a.Wh + (int64_t)(tc_lo + j) * F2 + 4 * f4)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float bias0 = tid < TC ? a.bh[tid] : 0.f;
  float y_pre[2] = {0.f, 0.f}, w_pre = 1.f;
  const bool pre = train && tid < a.T && (a.mode == 1 || a.C <= 2);
  if (pre) {
    if (a.mode == 0) {
      y_pre[0] = a.labels[((int64_t)m * a.T + tid) * a.C];
      if (a.C > 1) y_pre[1] = a.labels[((int64_t)m * a.T + tid) * a.C + 1];
    } else {
      y_pre[0] = a.labels[(int64_t)m * a.T + tid];
    }
    if (a.weights) w_pre = a.weights[(int64_t)m * a.T + tid];
  }
  const int fq = lane % LPR, grp = lane / LPR;  // this lane's 4 features, its row slot
  BnCol bn[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bn[e] = bn_col(a.bn, F, 4 * fq + e);
  BnCol bnf = bn[0];
  if (tid < F) bnf = bn_col(a.bn, F, tid);
  const int n_b = 2 * (g.max_deg + 1);
  const int rb = lane < n_b ? g.mol_runs[(int64_t)m * n_b + lane] : 0;
  int n_m = 0;
  for (int dd = 0; dd <= g.max_deg; ++dd) n_m += __shfl(rb, 2 * dd + 1) - __shfl(rb, 2 * dd);
  // ---- rows: wave w takes list positions [w Q, (w + 1) Q)
  float s[4] = {0.f, 0.f, 0.f, 0.f}, raw[4] = {0.f, 0.f, 0.f, 0.f}, rawarg[4] = {0.f, 0.f, 0.f, 0.f};
  float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int arow[4] = {-1, -1, -1, -1};
  const int Q = (n_m + 3) / 4;
  const int p_lo = wave * Q, p_hi = min(n_m, p_lo + Q);
  for (int base = p_lo; base < p_hi && !(g.diag & 32); base += 64) {
    int myrow = -1, pos = 0;
    for (int dd = 0; dd <= g.max_deg; ++dd) {
      const int b0 = __shfl(rb, 2 * dd), len = __shfl(rb, 2 * dd + 1) - b0;
      const int j = base + lane - pos;
      if (j >= 0 && j < len) myrow = b0 + j;
      pos += len;
    }
    const int cnt = min(64, p_hi - base);
    for (int p = 0; p < cnt; p += 8 * RPW) {  // eight wave-loads of rows in flight
      float4 v4[8];
      int rr[8];
      bool ok[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int pp = p + u * RPW + grp;
        rr[u] = __shfl(myrow, pp & 63);
        ok[u] = pp < cnt;
        v4[u] = ok[u] ? ldA(g.bf16, a.dense, rr[u], F, fq) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (!ok[u]) continue;
        const float* v = reinterpret_cast<const float*>(&v4[u]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float y = fmaf(v[e], bn[e].scale, bn[e].shift);
          raw[e] += v[e];
          s[e] += y;
          if (y > mx[e]) {
            mx[e] = y;
            arow[e] = rr[u];
            rawarg[e] = v[e];
          }
        }
      }
    }
  }
  // row slots of a wave: sums add; the maximum keeps the lowest row index on ties (= the first row of the walk)
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s[e] += __shfl_xor(s[e], o);
      raw[e] += __shfl_xor(raw[e], o);
      const float omx = __shfl_xor(mx[e], o);
      const int orow = __shfl_xor(arow[e], o);
      const float oraw = __shfl_xor(rawarg[e], o);
      if (omx > mx[e] || (omx == mx[e] && orow >= 0 && (arow[e] < 0 || orow < arow[e]))) {
        mx[e] = omx;
        arow[e] = orow;
        rawarg[e] = oraw;
      }
    }
  }
  if (grp == 0) {
    float* cw = comb + wave * 5 * F;
    st4(cw + 4 * fq, make_float4(s[0], s[1], s[2], s[3]));
    st4(cw + F + 4 * fq, make_float4(raw[0], raw[1], raw[2], raw[3]));
    st4(cw + 2 * F + 4 * fq, make_float4(mx[0], mx[1], mx[2], mx[3]));
    st4(cw + 3 * F + 4 * fq, make_float4(rawarg[0], rawarg[1], rawarg[2], rawarg[3]));
    *reinterpret_cast<int4*>(cw + 4 * F + 4 * fq) = make_int4(arow[0], arow[1], arow[2], arow[3]);
  }
  __syncthreads();
  // ---- the four quarters meet: thread f owns feature f from here on
  float f_raw = 0.f, f_rawarg = 0.f;
  if (tid < F) {
    float fs = 0.f, fmx = -INFINITY;
    int frow = -1;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* cw = comb + w * 5 * F;
      fs += cw[tid];
      f_raw += cw[F + tid];
      const float omx = cw[2 * F + tid];
      const int orow = reinterpret_cast<const int*>(cw + 4 * F)[tid];
      if (omx > fmx || (omx == fmx && orow >= 0 && (frow < 0 || orow < frow))) {
        fmx = omx;
        frow = orow;
        f_rawarg = cw[3 * F + tid];
      }
    }
    const float ts = tanhf(fs), tm = tanhf(fmx);
    sfp[tid] = ts;
    sfp[F + tid] = tm;
    a.fp[(int64_t)m * F2 + tid] = ts;
    a.fp[(int64_t)m * F2 + F + tid] = tm;
    if (train) a.argrow[(int64_t)m * F + tid] = frow;
  }
  __syncthreads();
  if (g.diag & 64) return;
  // ---- task head: thread = (output tc mod 32, one of 8 column segments)
  for (int tc0 = 0; tc0 < TC; tc0 += 32) {
    const int tc = tc0 + tcl;
    float p = 0.f;
    if (tc < TC) {
#pragma unroll
      for (int j = 0; j < NH4; ++j) {
        const float4 w4 = tc0 == 0 ? wh[j] : ld4(a.Wh + (int64_t)tc * F2 + fseg * SEG + 4 * j);
        const float4 x4 = ld4(sfp + fseg * SEG + 4 * j);
        p = fmaf(x4.x, w4.x, p);
        p = fmaf(x4.y, w4.y, p);
        p = fmaf(x4.z, w4.z, p);
        p = fmaf(x4.w, w4.w, p);
      }
      part[fseg * TCp + tc] = p;
    }
  }
  __syncthreads();
  for (int tc = tid; tc < TC; tc += kSBlock) {
    float v = tc == tid && tid < TC ? bias0 : a.bh[tc];
#pragma unroll
    for (int sgm = 0; sgm < 8; ++sgm) v += part[sgm * TCp + tc];
    slog[tc] = v;
    a.logits[(int64_t)m * TC + tc] = v;
  }
  __syncthreads();
  float lsum = 0.f;
  for (int t = tid; t < a.T; t += kSBlock) {
    const bool use_pre = pre && t == tid;
    if (a.mode == 0) {
      const float* x = slog + t * a.C;
      float mxl = -INFINITY;
      for (int c = 0; c < a.C; ++c) mxl = fmaxf(mxl, x[c]);
      float se = 0.f;
      for (int c = 0; c < a.C; ++c) se += expf(x[c] - mxl);
      const float lse = logf(se);
      const float w = use_pre ? w_pre : ((train && a.weights) ? a.weights[(int64_t)m * a.T + t] : 1.f);
      float ysum = 0.f, l = 0.f;
      if (train)
        for (int c = 0; c < a.C; ++c) ysum += use_pre ? y_pre[c] : a.labels[((int64_t)m * a.T + t) * a.C + c];
      for (int c = 0; c < a.C; ++c) {
        const float logp = x[c] - mxl - lse;
        const float p = expf(logp);
        if (a.probs) a.probs[((int64_t)m * a.T + t) * a.C + c] = p;
        if (train) {
          const float y = use_pre ? y_pre[c] : a.labels[((int64_t)m * a.T + t) * a.C + c];
          l -= y * logp;
          sdl[t * a.C + c] = m < a.n_rows ? w * (p * ysum - y) * a.inv_count : 0.f;
        }
      }
      if (train && m < a.n_rows) lsum += w * l;
    } else if (train) {
      const float w = use_pre ? w_pre : (a.weights ? a.weights[(int64_t)m * a.T + t] : 1.f);
      const float dlt = slog[t] - (use_pre ? y_pre[0] : a.labels[(int64_t)m * a.T + t]);
      sdl[t] = m < a.n_rows ? 2.f * dlt * w * a.inv_count : 0.f;
      if (m < a.n_rows) lsum += w * dlt * dlt;
    }
  }
  if (!train) return;
  lsum = wave_sum(lsum);
  if (lane == 0 && lsum != 0.f && !(g.diag & 2)) atomicAdd(a.loss_acc, (double)lsum);
  __syncthreads();
  for (int tc = tid; tc < TC; tc += kSBlock) a.dlogits[(int64_t)m * TC + tc] = sdl[tc];
  // ---- d fingerprint = dlogits . Wh: thread = (4 consecutive inputs, a quarter of the outputs)
  for (int f0 = 0; f0 < F2; f0 += 256) {
    const int f = f0 + 4 * f4;
    if (f < F2) {
      float4 gacc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int tc = tc_lo; tc < tc_hi; ++tc) {
        const float d = sdl[tc];
        const int j = tc - tc_lo;
        float4 w4;
        if (f0 == 0 && j < ND) {
          w4 = wd[0];
#pragma unroll
          for (int jj = 1; jj < ND; ++jj)
            if (j == jj) w4 = wd[jj];
        } else {
          w4 = ld4(a.Wh + (int64_t)tc * F2 + f);
        }
        gacc.x = fmaf(d, w4.x, gacc.x);
        gacc.y = fmaf(d, w4.y, gacc.y);
        gacc.z = fmaf(d, w4.z, gacc.z);
        gacc.w = fmaf(d, w4.w, gacc.w);
      }
      st4(gpart + tcg * F2 + f, gacc);
    }
  }
  __syncthreads();
  for (int f = tid; f < F2; f += kSBlock) {
    const float y = sfp[f];
    const float gsum = (gpart[f] + gpart[F2 + f] + gpart[2 * F2 + f] + gpart[3 * F2 + f]) * (1.f - y * y);
    a.g2[(int64_t)m * F2 + f] = gsum;
    sgr[f] = gsum;
  }
  __syncthreads();
  if (tid < F && a.bsum && n_m > 0 && !(g.diag & 2)) {
    const float gs = sgr[tid], gm = sgr[F + tid];
    const float sum_xhat = bnf.invstd * (f_raw - (float)n_m * bnf.mean);
    const float xhat_arg = (f_rawarg - bnf.mean) * bnf.invstd;
    atomicAdd(a.bsum + tid, (double)((float)n_m * gs + gm));
    atomicAdd(a.bsum + F + tid, (double)(gs * sum_xhat + gm * xhat_arg));
  }
}

// ------------------------------------------------------------------------------------------------ dense backward
// One launch, three kinds of workgroups:
//   [0, n_tiles)            16-row tiles: dx of the dense pre-activation (readout backward recomputed from the
//                           per-molecule gradient, BatchNorm backward, ReLU mask) -> dP = dx . Wd  (N x K)
//   [n_tiles, +n_slabs)     64-row slabs: dWd += dx^T . P, dbd += column sums of dx (float atomics, one per
//                           output element and slab); slab 0 also writes the BatchNorm gradient
//   [.., +head blocks)      dWh = dlogits^T . fp, dbh = column sums of dlogits (plain stores)
struct DenseBwdArgs {
  const float* dense;      // N x F (post-ReLU dense output = BatchNorm input)
  const float* pool;       // N x K
  const float* g2;         // B x 2F
  const int32_t* argrow;   // B x F
  BnArgs bn;               // mode 1 (batch statistics) or 0
  const double* bsum;      // [sum dy | sum dy xhat]
  const float* Wd;         // F x K
  float* dpool;            // N x K
  float* dWd;              // F x K   (accumulated)
  float* dbd;              // F       (accumulated)
  float* dgamma;           // F       (written)
  float* dbeta;            // F
  const float* dlogits;    // B x TC
  const float* fp;         // B x 2F
  float* dWh;              // TC x 2F (written)
  float* dbh;              // TC
  int F, K, TC, n_slabs, n_head_blocks;
};

// BatchNorm backward as dx = A dy + B x + C per column:
//   xhat = (x - mean) invstd;  dx = gamma invstd (dy - S1/N - xhat S2/N),  S1 = sum dy, S2 = sum dy xhat
//   => A = gamma invstd,  B = -A invstd S2/N,  C = -A S1/N - B mean.   Without BatchNorm: A = 1, B = C = 0.
// (s1, s2: fp64 sums for the dense layer, the float gradient entries dbeta / dgamma for a GraphConv layer.)
struct BwdQuad {
  float4 A, B, C;
};
template <typename T>
__device__ __forceinline__ BwdQuad bn_bwd_quad(const BnArgs& bn, const T* s1, const T* s2, int F, int q) {
  BwdQuad o;
  float* pA = reinterpret_cast<float*>(&o.A);
  float* pB = reinterpret_cast<float*>(&o.B);
  float* pC = reinterpret_cast<float*>(&o.C);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * q + e;
    if (bn.mode == 0) {
      pA[e] = 1.f;
      pB[e] = 0.f;
      pC[e] = 0.f;
      continue;
    }
    const BnCol b = bn_col(bn, F, c);
    const float A = bn.gamma[c] * b.invstd;
    const float B = -A * b.invstd * (float)((double)s2[c] * bn.inv_n);
    const float C = -A * (float)((double)s1[c] * bn.inv_n) - B * b.mean;
    pA[e] = A;
    pB[e] = B;
    pC[e] = C;
  }
  return o;
}

// dx of `rows` consecutive rows starting at row0 into LDS tile sDx[rows][pitch].  F / 4 divides the block size, so
// a thread keeps ONE column quad over all its rows: coefficients once, then its rows four at a time -- the four
// membership loads together, then the sixteen loads that depend on them together.
__device__ __forceinline__ void dense_dx_to_lds(const SmallGraph& g, const DenseBwdArgs& a, int row0, int nrows,
                                                int tile_rows, float* sDx, int pitch) {
  const int F = a.F, q4 = F / 4;
  const int q = threadIdx.x % q4, r_first = threadIdx.x / q4, r_step = kSBlock / q4;
  const BwdQuad cq = bn_bwd_quad(a.bn, a.bsum, a.bsum + F, F, q);
  const float4 cA = cq.A, cB = cq.B, cC = cq.C;
  for (int r0 = r_first; r0 < tile_rows; r0 += 4 * r_step) {
    int m[4];
    float4 x[4], gs[4], gm[4];
    int4 ar[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = r0 + t * r_step;
      m[t] = r < nrows ? g.membership[row0 + r] : 0;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = r0 + t * r_step;
      const int row = r < nrows ? row0 + r : row0;
      x[t] = ldA(g.bf16, a.dense, row, F, q);
      gs[t] = ld4(a.g2 + (int64_t)m[t] * 2 * F + 4 * q);
      gm[t] = ld4(a.g2 + (int64_t)m[t] * 2 * F + F + 4 * q);
      ar[t] = *reinterpret_cast<const int4*>(a.argrow + (int64_t)m[t] * F + 4 * q);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = r0 + t * r_step;
      if (r >= tile_rows) continue;
      float4 dx = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < nrows) {
        const int row = row0 + r;
        const float dy0 = gs[t].x + (ar[t].x == row ? gm[t].x : 0.f), dy1 = gs[t].y + (ar[t].y == row ? gm[t].y : 0.f),
                    dy2 = gs[t].z + (ar[t].z == row ? gm[t].z : 0.f), dy3 = gs[t].w + (ar[t].w == row ? gm[t].w : 0.f);
        // dx = A dy + B x + C  (BatchNorm backward folded per column), masked by the ReLU in front
        dx.x = x[t].x > 0.f ? fmaf(cA.x, dy0, fmaf(cB.x, x[t].x, cC.x)) : 0.f;
        dx.y = x[t].y > 0.f ? fmaf(cA.y, dy1, fmaf(cB.y, x[t].y, cC.y)) : 0.f;
        dx.z = x[t].z > 0.f ? fmaf(cA.z, dy2, fmaf(cB.z, x[t].z, cC.z)) : 0.f;
        dx.w = x[t].w > 0.f ? fmaf(cA.w, dy3, fmaf(cB.w, x[t].w, cC.w)) : 0.f;
      }
      st4(sDx + r * pitch + 4 * q, dx);
    }
  }
}

template <int NKT>  // K = 64 * NKT columns of dP per row
__global__ void __launch_bounds__(kSBlock)
small_dense_bwd_kernel(SmallGraph g, DenseBwdArgs a) {
  if (g.diag & 1024) return;
  extern __shared__ float smem[];
  const int F = a.F, K = a.K;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lr = lane & 15, kq = lane >> 4;
  int b = blockIdx.x;
  if (b < g.n_tiles) {
    if (g.diag & 512) return;
    const Tile tl = tile_of(g, b);
    const int FP = pitch_a(F);
    float* sDx = smem;  // [16][FP]
    float bw[kBChunk][NKT];
    auto load_w = [&](int ks0) {
#pragma unroll
      for (int u = 0; u < kBChunk; ++u)
#pragma unroll
        for (int i = 0; i < NKT; ++i) bw[u][i] = a.Wd[(int64_t)(4 * (ks0 + u) + kq) * K + 16 * (wave + 4 * i) + lr];
    };
    load_w(0);  // in flight while dx is formed
    dense_dx_to_lds(g, a, tl.row0, tl.nrows, kTileRows, sDx, FP);
    __syncthreads();
    f4v c[NKT];
#pragma unroll
    for (int i = 0; i < NKT; ++i) c[i] = (f4v){0.f, 0.f, 0.f, 0.f};
    for (int ks0 = 0; ks0 < F / 4 && !(g.diag & 8); ks0 += kBChunk) {
      float cw[kBChunk][NKT];
#pragma unroll
      for (int u = 0; u < kBChunk; ++u)
#pragma unroll
        for (int i = 0; i < NKT; ++i) cw[u][i] = bw[u][i];
      if (ks0 + kBChunk < F / 4) load_w(ks0 + kBChunk);
#pragma unroll
      for (int u = 0; u < kBChunk; ++u) {
        const float av = sDx[lr * FP + 4 * (ks0 + u) + kq];  // F / 4 is a multiple of the chunk
#pragma unroll
        for (int i = 0; i < NKT; ++i) c[i] = mfma16(av, cw[u][i], c[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < NKT; ++i) {
      const int col = 16 * (wave + 4 * i) + lr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * kq + j;
        if (r < tl.nrows) a.dpool[(int64_t)(tl.row0 + r) * K + col] = c[i][j];
      }
    }
    return;
  }
  b -= g.n_tiles;
  if (b < a.n_slabs) {
    if (g.diag & 128) return;
    // dWd[o][k] += sum_r dx[r][o] P[r][k] over the slab's rows
    const int row0 = b * kSlabRows;
    const int nrows = min(kSlabRows, g.n_atoms - row0);
    const int FT = pitch_t(F), KT = pitch_t(K);
    float* sDx = smem;                    // [slab][FT]
    float* sP = sDx + kSlabRows * FT;     // [slab][KT]
    if (b == 0 && a.bn.mode != 0)         // the BatchNorm gradient is the pair of sums itself
      for (int c = threadIdx.x; c < F; c += kSBlock) {
        a.dgamma[c] = (float)a.bsum[F + c];
        a.dbeta[c] = (float)a.bsum[c];
      }
    dense_dx_to_lds(g, a, row0, nrows, kSlabRows, sDx, FT);
#pragma unroll 4
    for (int idx = threadIdx.x; idx < kSlabRows * (K / 4); idx += kSBlock) {
      const int r = idx / (K / 4), q = idx - r * (K / 4);
      st4(sP + r * KT + 4 * q, r < nrows ? ldA(g.bf16, a.pool, row0 + r, K, q) : make_float4(0.f, 0.f, 0.f, 0.f));
    }
    __syncthreads();
    const int n_ot = F / 16, n_kt = K / 16;
    for (int ot = wave; ot < n_ot && !(g.diag & 8); ot += 4) {
      for (int kt0 = 0; kt0 < n_kt; kt0 += 4) {
        f4v c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = (f4v){0.f, 0.f, 0.f, 0.f};
        for (int rs = 0; rs < kSlabRows / 4; ++rs) {
          const int r = 4 * rs + kq;
          const float av = sDx[r * FT + 16 * ot + lr];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (kt0 + i < n_kt) c[i] = mfma16(av, sP[r * KT + 16 * (kt0 + i) + lr], c[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (kt0 + i >= n_kt) continue;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int o = 16 * ot + 4 * kq + j, k = 16 * (kt0 + i) + lr;
            if (!(g.diag & 4)) atomicAdd(a.dWd + (int64_t)o * K + k, c[i][j]);
          }
        }
      }
    }
    for (int o = threadIdx.x; o < F; o += kSBlock) {
      float s = 0.f;
      for (int r = 0; r < nrows; ++r) s += sDx[r * FT + o];
      atomicAdd(a.dbd + o, s);
    }
    return;
  }
  b -= a.n_slabs;
  // head weight gradient: one workgroup per output row tc; the molecules are split over the four waves, a lane
  // owns 4 consecutive input columns (coalesced float4 rows of fp), partial sums meet in LDS
  if (g.diag & 256) return;
  const int tc = b;
  const int F2 = 2 * F;
  float* red = smem;  // [4][F2]
  const int mg = threadIdx.x >> 6, ln = threadIdx.x & 63;
  float dsum = 0.f;
  for (int f0 = 0; f0 < F2; f0 += 256) {
    const int f = f0 + 4 * ln;
    float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < F2) {
#pragma unroll 8
      for (int m = mg; m < g.n_mols; m += 4) {
        const float d = a.dlogits[(int64_t)m * a.TC + tc];
        const float4 x4 = ld4(a.fp + (int64_t)m * F2 + f);
        acc4.x = fmaf(d, x4.x, acc4.x);
        acc4.y = fmaf(d, x4.y, acc4.y);
        acc4.z = fmaf(d, x4.z, acc4.z);
        acc4.w = fmaf(d, x4.w, acc4.w);
        if (f0 == 0 && ln == 0) dsum += d;
      }
      st4(red + mg * F2 + f, acc4);
    }
  }
  if (ln == 0) red[4 * F2 + mg] = dsum;
  __syncthreads();
  for (int f = threadIdx.x; f < F2; f += kSBlock)
    a.dWh[(int64_t)tc * F2 + f] = red[f] + red[F2 + f] + red[2 * F2 + f] + red[3 * F2 + f];
  if (threadIdx.x == 0) a.dbh[tc] = red[4 * F2] + red[4 * F2 + 1] + red[4 * F2 + 2] + red[4 * F2 + 3];
}

// ------------------------------------------------------------------------------------------------ pool backward
// dA[k] = [arg[k] == self] dP[k] + sum_j [arg[nb_j] == slot of k in nb_j's list + 1] dP[nb_j]   (a gather: every
// bond is listed from both ends, rev_pos names the slot), i.e. the gradient w.r.t. the BatchNorm OUTPUT of the
// layer; its column sums with and without xhat are the BatchNorm gradient (dbeta, dgamma; float atomics).
__global__ void __launch_bounds__(kSBlock)
small_pool_bwd_kernel(SmallGraph g, const float* __restrict__ dpool, const uint8_t* __restrict__ arg,
                      const float* __restrict__ gc, int W, BnArgs bn, float* __restrict__ dA,
                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
  if (g.diag & 1024) return;
  extern __shared__ float smem[];
  const int t = blockIdx.x;
  if (t >= g.n_tiles) return;
  const Tile tl = tile_of(g, t);
  const int q4 = W / 4;
  float* red = smem;  // [16][2W] partial sums per row of the tile
  for (int idx = threadIdx.x; idx < kTileRows * q4; idx += kSBlock) {
    const int r = idx / q4, q = idx - r * q4;
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f), dx = d;
    if (r < tl.nrows) {
      const int row = tl.row0 + r;
      const uint32_t a = *reinterpret_cast<const uint32_t*>(arg + (int64_t)row * W + 4 * q);
      const float4 v = ld4(dpool + (int64_t)row * W + 4 * q);
      d.x = (a & 0xff) == 0 ? v.x : 0.f;
      d.y = ((a >> 8) & 0xff) == 0 ? v.y : 0.f;
      d.z = ((a >> 16) & 0xff) == 0 ? v.z : 0.f;
      d.w = (a >> 24) == 0 ? v.w : 0.f;
      const int e = tl.e0 + r * tl.d;
      for (int j0 = 0; j0 < tl.d; j0 += 4) {  // four neighbours per round: ids together, then their rows together
        int nb[4];
        uint32_t want[4], an[4];
        float4 vn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool in = j0 + t < tl.d;
          nb[t] = in ? g.col_idx[e + j0 + t] : row;
          want[t] = in ? (uint32_t)g.rev_pos[e + j0 + t] + 1u : 0xffffu;  // 0xffff matches no arg byte
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          an[t] = *reinterpret_cast<const uint32_t*>(arg + (int64_t)nb[t] * W + 4 * q);
          vn[t] = ld4(dpool + (int64_t)nb[t] * W + 4 * q);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          d.x += (an[t] & 0xff) == want[t] ? vn[t].x : 0.f;
          d.y += ((an[t] >> 8) & 0xff) == want[t] ? vn[t].y : 0.f;
          d.z += ((an[t] >> 16) & 0xff) == want[t] ? vn[t].z : 0.f;
          d.w += (an[t] >> 24) == want[t] ? vn[t].w : 0.f;
        }
      }
      if (dA) st4(dA + (int64_t)row * W + 4 * q, d);
      const float4 x = ldA(g.bf16, gc, row, W, q);
      const BnQuad bq = bn_quad(bn, W, q);
      const float4 mu = bq.mean, iv = bq.invstd;
      dx.x = d.x * (x.x - mu.x) * iv.x;
      dx.y = d.y * (x.y - mu.y) * iv.y;
      dx.z = d.z * (x.z - mu.z) * iv.z;
      dx.w = d.w * (x.w - mu.w) * iv.w;
    }
    st4(red + r * 2 * W + 4 * q, d);
    st4(red + r * 2 * W + W + 4 * q, dx);
  }
  __syncthreads();
  if (bn.mode == 0 || dgamma == nullptr) return;
  for (int c = threadIdx.x; c < 2 * W; c += kSBlock) {
    float s = 0.f;
    for (int r = 0; r < tl.nrows; ++r) s += red[r * 2 * W + c];
    if (c < W)
      atomicAdd(dbeta + c, s);
    else
      atomicAdd(dgamma + c - W, s);
  }
}

// ------------------------------------------------------------------------------------------------ conv backward (full mode)
// Gradient through BatchNorm + ReLU of a GraphConv layer, then
//   tiles:  dS = dg . W_rel[d]^T (N x K), dXs = dg . W_self[d]^T (N x K)       (layer > 0 only)
//   slabs:  dW_rel[d] += S^T dg, dW_self[d] += X^T dg, db_rel[d] += colsum dg, db_self[d] += colsum dg
// with S recomputed by the neighbour gather.  dA = gradient w.r.t. the BatchNorm output (from pool backward),
// dgamma/dbeta = its column sums (already complete: previous launch).
struct ConvBwdArgs {
  const float* dA;       // N x W
  const float* gc;       // N x W post-ReLU conv output (BatchNorm input)
  BnArgs bn;
  const float* dgamma;   // sum dA xhat
  const float* dbeta;    // sum dA
  const float* x;        // N x ldx layer input
  int x_bf;              // the layer input is a stored activation (bf16 storage mode), not the fp32 atom features
  int ldx, K, W;
  const float* Wl;       // 21 x K x W
  float* dWl;            // accumulated
  float* dbl;            // 21 x W accumulated
  float* dS;             // N x K4 (layer > 0) or NULL
  float* dXs;            // N x K4
  int n_slabs;
  int32_t slab_start[GCMI_MAX_DEG + 2];  // 64-row slabs per degree block, prefix
};

__device__ __forceinline__ void conv_dg_to_lds(int bf, const ConvBwdArgs& a, int row0, int nrows, int tile_rows,
                                               float* sG, int pitch) {
  const int W = a.W, q4 = W / 4;
  for (int idx = threadIdx.x; idx < tile_rows * q4; idx += kSBlock) {
    const int r = idx / q4, q = idx - r * q4;
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nrows) {
      const BwdQuad cq = bn_bwd_quad(a.bn, a.dbeta, a.dgamma, W, q);
      const float4 cA = cq.A, cB = cq.B, cC = cq.C;
      const float4 dy = ld4(a.dA + (int64_t)(row0 + r) * W + 4 * q);
      const float4 x = ldA(bf, a.gc, row0 + r, W, q);
      dg.x = x.x > 0.f ? fmaf(cA.x, dy.x, fmaf(cB.x, x.x, cC.x)) : 0.f;
      dg.y = x.y > 0.f ? fmaf(cA.y, dy.y, fmaf(cB.y, x.y, cC.y)) : 0.f;
      dg.z = x.z > 0.f ? fmaf(cA.z, dy.z, fmaf(cB.z, x.z, cC.z)) : 0.f;
      dg.w = x.w > 0.f ? fmaf(cA.w, dy.w, fmaf(cB.w, x.w, cC.w)) : 0.f;
    }
    st4(sG + r * pitch + 4 * q, dg);
  }
}

__global__ void __launch_bounds__(kSBlock)
small_conv_bwd_kernel(SmallGraph g, ConvBwdArgs a) {
  extern __shared__ float smem[];
  const int W = a.W, K = a.K, K4 = (K + 3) & ~3;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lr = lane & 15, kq = lane >> 4;
  const int64_t blk = (int64_t)K * W;
  int b = blockIdx.x;
  const int n_dgrad = a.dS ? g.n_tiles : 0;
  if (b < n_dgrad) {
    const Tile tl = tile_of(g, b);
    const int WP = pitch_a(W);
    float* sG = smem;  // [16][WP]
    conv_dg_to_lds(g.bf16, a, tl.row0, tl.nrows, kTileRows, sG, WP);
    __syncthreads();
    const float* Wself = a.Wl + (tl.d == 0 ? (int64_t)(2 * g.max_deg) * blk : (int64_t)(2 * (tl.d - 1) + 1) * blk);
    const float* Wrel = tl.d == 0 ? nullptr : a.Wl + (int64_t)(2 * (tl.d - 1)) * blk;
    // out[r][k] = sum_c dg[r][c] W[k][c]: B fragment = W[k = col][c = 4 ks + kq]
    for (int kt = wave; kt < K4 / 16 + ((K4 % 16) ? 1 : 0); kt += 4) {
      const int k = 16 * kt + lr;
      const int kc = k < K ? k : K - 1;
      f4v cs = (f4v){0.f, 0.f, 0.f, 0.f}, cx = cs;
      for (int ks = 0; ks < W / 4; ++ks) {
        const int c = 4 * ks + kq;
        const float av = sG[lr * WP + c];
        cx = mfma16(av, Wself[(int64_t)kc * W + c], cx);
        if (Wrel) cs = mfma16(av, Wrel[(int64_t)kc * W + c], cs);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * kq + j;
        if (r < tl.nrows && k < K4) {
          a.dXs[(int64_t)(tl.row0 + r) * K4 + k] = k < K ? cx[j] : 0.f;
          a.dS[(int64_t)(tl.row0 + r) * K4 + k] = k < K ? cs[j] : 0.f;
        }
      }
    }
    return;
  }
  b -= n_dgrad;
  if (b >= a.n_slabs) return;
  // weight-gradient slab: 64 rows of one degree
  int d = 0;
#pragma unroll
  for (int k = 1; k <= GCMI_MAX_DEG; ++k) d += (k <= g.max_deg && b >= a.slab_start[k]) ? 1 : 0;
  const int row0 = g.deg_start[d] + (b - a.slab_start[d]) * kSlabRows;
  const int nrows = min(kSlabRows, g.deg_start[d + 1] - row0);
  const int e0 = g.edge_start[d] + (row0 - g.deg_start[d]) * d;
  const int WT = pitch_t(W), KT = pitch_t(K4);
  float* sG = smem;                     // [slab][WT]
  float* sX = sG + kSlabRows * WT;      // [slab][KT] own rows
  float* sS = sX + kSlabRows * KT;      // [slab][KT] neighbour sums
  conv_dg_to_lds(g.bf16, a, row0, nrows, kSlabRows, sG, WT);
  const int q4 = K4 / 4;
  for (int idx = threadIdx.x; idx < kSlabRows * q4; idx += kSBlock) {
    const int r = idx / q4, q = idx - r * q4;
    float4 self = make_float4(0.f, 0.f, 0.f, 0.f), s = self;
    if (r < nrows) {
      self = ldA(a.x_bf, a.x, row0 + r, a.ldx, q);
      s = gather_sum_quad(a.x_bf, a.x, a.ldx, g.col_idx + e0 + r * d, d, row0 + r, q);
    }
    st4(sX + r * KT + 4 * q, self);
    st4(sS + r * KT + 4 * q, s);
  }
  __syncthreads();
  float* dWself = a.dWl + (d == 0 ? (int64_t)(2 * g.max_deg) * blk : (int64_t)(2 * (d - 1) + 1) * blk);
  float* dWrel = d == 0 ? nullptr : a.dWl + (int64_t)(2 * (d - 1)) * blk;
  // dW[k][c] = sum_r X[r][k] dg[r][c]: A fragment = X^T (m = k), B fragment = dg (n = c)
  const int n_kt = (K4 + 15) / 16, n_ct = W / 16;
  for (int kt = wave; kt < n_kt; kt += 4) {
    const int k = 16 * kt + lr;
    for (int ct0 = 0; ct0 < n_ct; ct0 += 4) {
      f4v cx[4], cs[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) cx[i] = cs[i] = (f4v){0.f, 0.f, 0.f, 0.f};
      for (int rs = 0; rs < kSlabRows / 4; ++rs) {
        const int r = 4 * rs + kq;
        const float ax = k < K4 ? sX[r * KT + k] : 0.f;
        const float as = k < K4 ? sS[r * KT + k] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (ct0 + i >= n_ct) continue;
          const float bv = sG[r * WT + 16 * (ct0 + i) + lr];
          cx[i] = mfma16(ax, bv, cx[i]);
          if (dWrel) cs[i] = mfma16(as, bv, cs[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (ct0 + i >= n_ct) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kk = 16 * kt + 4 * kq + j, c = 16 * (ct0 + i) + lr;
          if (kk < K) {
            atomicAdd(dWself + (int64_t)kk * W + c, cx[i][j]);
            if (dWrel) atomicAdd(dWrel + (int64_t)kk * W + c, cs[i][j]);
          }
        }
      }
    }
  }
  for (int c = threadIdx.x; c < W; c += kSBlock) {
    float s = 0.f;
    for (int r = 0; r < nrows; ++r) s += sG[r * WT + c];
    if (d == 0) {
      atomicAdd(a.dbl + (int64_t)(2 * g.max_deg) * W + c, s);
    } else {
      atomicAdd(a.dbl + (int64_t)(2 * (d - 1)) * W + c, s);
      atomicAdd(a.dbl + (int64_t)(2 * (d - 1) + 1) * W + c, s);
    }
  }
}

// dP[k] = dXs[k] + sum_j dS[nb_j(k)]: the transposed neighbour sum over a symmetric adjacency is a gather
__global__ void __launch_bounds__(kSBlock)
small_gather_add_kernel(SmallGraph g, const float* __restrict__ dXs, const float* __restrict__ dS, int W,
                        float* __restrict__ dP) {
  const int t = blockIdx.x;
  if (t >= g.n_tiles) return;
  const Tile tl = tile_of(g, t);
  const int q4 = W / 4;
  for (int idx = threadIdx.x; idx < tl.nrows * q4; idx += kSBlock) {
    const int r = idx / q4, q = idx - r * q4;
    const int row = tl.row0 + r;
    float4 s = ld4(dXs + (int64_t)row * W + 4 * q);
    const float4 n = gather_sum_quad(0, dS, W, g.col_idx + tl.e0 + r * tl.d, tl.d, row, q);
    s.x += n.x;
    s.y += n.y;
    s.z += n.z;
    s.w += n.w;
    st4(dP + (int64_t)row * W + 4 * q, s);
  }
}

// ------------------------------------------------------------------------------------------------ adam + bookkeeping
struct StepEnd {
  float* p;
  float* grad;
  float* m;
  float* v;
  int64_t lo, hi;          // trained range of the flat arenas
  float one_minus_b1, b2, one_minus_b2, step_size, inv_bc2_sqrt, eps;
  double* loss_acc;
  float* loss_out;         // this step's loss
  float inv_count;
  int n_bn;                // BatchNorm layers with batch statistics to fold into the running ones
  int n_rows;
  float momentum;
  double* acc[kMaxL + 1];  // [sum | sumsq] per layer
  int width[kMaxL + 1];
  float* rm[kMaxL + 1];
  float* rv[kMaxL + 1];
  int64_t* tracked[kMaxL + 1];
  double* zero_from;       // accumulator regions to clear for the next step that uses them
  int64_t zero_doubles;
  double* zero2_from;
  int64_t zero2_doubles;
};

__global__ void __launch_bounds__(kSBlock)
small_step_end_kernel(StepEnd s) {
  // torch.optim.Adam (optimizers.py:231-241): exp_avg.lerp_, exp_avg_sq.mul_.addcmul_, addcdiv_
  const int64_t n4 = (s.hi - s.lo) / 4;
  for (int64_t i = (int64_t)blockIdx.x * kSBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kSBlock) {
    const int64_t o = s.lo + 4 * i;
    float4 g = ld4(s.grad + o), m = ld4(s.m + o), v = ld4(s.v + o), p = ld4(s.p + o);
    float* gf = reinterpret_cast<float*>(&g);
    float* mf = reinterpret_cast<float*>(&m);
    float* vf = reinterpret_cast<float*>(&v);
    float* pf = reinterpret_cast<float*>(&p);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      mf[c] = mf[c] + (gf[c] - mf[c]) * s.one_minus_b1;
      vf[c] = vf[c] * s.b2 + gf[c] * gf[c] * s.one_minus_b2;
      const float denom = sqrtf(vf[c]) * s.inv_bc2_sqrt + s.eps;
      pf[c] -= s.step_size * (mf[c] / denom);
    }
    st4(s.m + o, m);
    st4(s.v + o, v);
    st4(s.p + o, p);
    st4(s.grad + o, make_float4(0.f, 0.f, 0.f, 0.f));  // the next step accumulates into a clean arena
  }
  if (blockIdx.x != 0) return;
  if (threadIdx.x == 0 && s.loss_out) *s.loss_out = (float)(*s.loss_acc * (double)s.inv_count);
  // running statistics: running = (1 - momentum) running + momentum batch; unbiased variance (nn.BatchNorm1d)
  for (int l = 0; l < s.n_bn; ++l) {
    const int W = s.width[l];
    for (int c = threadIdx.x; c < W; c += kSBlock) {
      const double inv_n = 1.0 / (double)s.n_rows;
      const double mean = s.acc[l][c] * inv_n;
      double var = s.acc[l][W + c] * inv_n - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const double unbiased = s.n_rows > 1 ? var * (double)s.n_rows / (double)(s.n_rows - 1) : var;
      s.rm[l][c] = (1.f - s.momentum) * s.rm[l][c] + s.momentum * (float)mean;
      s.rv[l][c] = (1.f - s.momentum) * s.rv[l][c] + s.momentum * (float)unbiased;
    }
    if (threadIdx.x == 0 && s.tracked[l]) *s.tracked[l] += 1;
  }
  __syncthreads();
  for (int64_t i = threadIdx.x; i < s.zero_doubles; i += kSBlock) s.zero_from[i] = 0.0;
  for (int64_t i = threadIdx.x; i < s.zero2_doubles; i += kSBlock) s.zero2_from[i] = 0.0;
}

}  // namespace gcmi

// ================================================================================================ host side
namespace gcmi {

static inline int64_t up4s(int64_t n) { return (n + 3) / 4 * 4; }

struct SmallWs {  // offsets in floats into the workspace
  // GraphConv outputs and their BatchNorm sums exist in kSlots copies: in reference gradient mode the conv stack of
  // a later step depends on nothing an earlier step trains, so the stacks of the next kAhead steps run on a second
  // stream while the current group of steps finishes on the caller's (two sets of kAhead slots)
  int64_t gc[kSlots][kMaxL], pool[kMaxL], arg[kMaxL], dA[kMaxL];
  int64_t dense, fp, logits, dlogits, g2, argrow, dpool, dS, dXs;
  int64_t acc0;                   // start of the fp64 accumulator region
  int64_t acc[kSlots][kMaxL];     // doubles, relative to acc0: [sum | sumsq] per GraphConv BatchNorm and slot
  int64_t acc_dense, bsum, loss;  // doubles, relative to acc0: dense-layer sums, its backward sums, the loss
  int64_t par_begin[kSlots], par_doubles, shared_begin, shared_doubles;
  int64_t acc_doubles;
  int64_t total;
};

static SmallWs small_carve(const gcmi_model_desc* m, int64_t N, int64_t B) {
  SmallWs w;
  memset(&w, 0, sizeof(w));
  int64_t off = 0;
  auto take = [&](int64_t n) {
    int64_t o = off;
    off += up4s(n);
    return o;
  };
  const int L = m->n_layers;
  int64_t wmax = 0;
  for (int l = 0; l < L; ++l) {
    const int64_t W = m->conv_width[l];
    for (int p = 0; p < kSlots; ++p) w.gc[p][l] = take(N * W);
    w.pool[l] = take(N * W);
    w.arg[l] = take((N * W + 3) / 4);
    w.dA[l] = take(N * W);
    if (W > wmax) wmax = W;
  }
  const int64_t F = m->dense_width, TC = (int64_t)m->n_tasks * m->n_classes;
  w.dense = take(N * F);
  w.fp = take(B * 2 * F);
  w.logits = take(B * TC);
  w.dlogits = take(B * TC);
  w.g2 = take(B * 2 * F);
  w.argrow = take(B * F);
  w.dpool = take(N * wmax);
  w.dS = take(N * wmax);
  w.dXs = take(N * wmax);
  off = (off + 3) / 4 * 4;
  w.acc0 = off;
  int64_t d = 0;
  for (int p = 0; p < kSlots; ++p) {
    w.par_begin[p] = d;
    for (int l = 0; l < L; ++l) {
      w.acc[p][l] = d;
      d += 2 * m->conv_width[l];
    }
  }
  w.par_doubles = w.par_begin[1] - w.par_begin[0];
  w.shared_begin = d;
  w.acc_dense = d;
  d += 2 * F;
  w.bsum = d;
  d += 2 * F;
  w.loss = d;
  d += 2;
  w.shared_doubles = d - w.shared_begin;
  w.acc_doubles = d;
  off += 2 * d;
  w.total = off;
  return w;
}

static int small_check(const gcmi_model_desc* m) {
  GCMI_CHECK_ARG(m != nullptr, "small: model desc is NULL");
  GCMI_CHECK_ARG(m->n_layers >= 1 && m->n_layers <= kMaxL, "small: n_layers %d outside [1,%d]", m->n_layers, kMaxL);
  GCMI_CHECK_ARG(m->max_deg >= 0 && m->max_deg <= GCMI_MAX_DEG, "small: bad max_deg");
  GCMI_CHECK_ARG(m->n_feat_in > 0 && m->n_tasks > 0 && m->n_classes > 0, "small: bad widths");
  GCMI_CHECK_ARG(m->mode == 0 || (m->mode == 1 && m->n_classes == 1), "small: bad mode / n_classes");
  GCMI_CHECK_ARG(m->storage >= 0 && m->storage <= 2,
                 "small: storage must be 0 (fp32), 1 (bf16 activations) or 2 (and bf16 gradient streams: same as 1 here)");
  for (int l = 0; l < m->n_layers; ++l)
    if (m->conv_width[l] <= 0 || m->conv_width[l] % 64 || m->conv_width[l] > 256) {
      set_error("small: GraphConv width %d is not a multiple of 64 in [64, 256]", m->conv_width[l]);
      return GCMI_ERR_UNSUPPORTED;
    }
  if (m->dense_width != 64 && m->dense_width != 128 && m->dense_width != 256) {
    set_error("small: dense width %d is not 64, 128 or 256", m->dense_width);
    return GCMI_ERR_UNSUPPORTED;
  }
  return GCMI_OK;
}

static int make_small_graph(const gcmi_graph* g, bool need_rev, SmallGraph* out) {
  int rc = check_graph(g, true);
  if (rc) return rc;
  GCMI_CHECK_ARG(g->n_mols > 1, "graph_gather requires batches larger than 1");
  GCMI_CHECK_ARG(g->d_membership && g->d_mol_runs, "small: the graph lacks membership / mol_runs");
  if (need_rev && g->n_edges > 0 && !g->d_rev_pos) {
    set_error("small: the training step needs d_rev_pos (every bond listed from both ends)");
    return GCMI_ERR_UNSUPPORTED;
  }
  SmallGraph s;
  memset(&s, 0, sizeof(s));
  s.n_atoms = g->n_atoms;
  s.n_mols = g->n_mols;
  s.max_deg = g->max_deg;
  int t = 0;
  for (int d = 0; d < GCMI_MAX_DEG + 2; ++d) {
    const int dd = d < g->max_deg + 1 ? d : g->max_deg + 1;
    s.deg_start[d] = g->deg_start[dd];
    s.edge_start[d] = g->edge_start[dd];
    s.tile_start[d] = t;
    if (d <= g->max_deg) t += (g->deg_start[d + 1] - g->deg_start[d] + kTileRows - 1) / kTileRows;
  }
  s.n_tiles = t;
  // the diagnostic switches exist only in a build made for them (tools/small_diag.sh: -DGCMI_SMALL_DIAG_BUILD): in
  // the shipped library a stray environment variable cannot switch parts of the training kernels off
#ifdef GCMI_SMALL_DIAG_BUILD
  static const int diag = getenv("GCMI_SMALL_DIAG") ? atoi(getenv("GCMI_SMALL_DIAG")) : 0;
  if (diag) {
    static bool warned = false;
    if (!warned) fprintf(stderr, "libgcmi: GCMI_SMALL_DIAG=%d -- parts of the small-batch kernels are OFF, results are wrong\n", diag);
    warned = true;
  }
  s.diag = diag;
#else
  s.diag = 0;
#endif
  s.col_idx = g->d_col_idx;
  s.membership = g->d_membership;
  s.mol_runs = g->d_mol_runs;
  s.rev_pos = g->d_rev_pos;
  *out = s;
  return GCMI_OK;
}

template <typename Kern>
static int ensure_lds(Kern kern, size_t bytes) {
  if (bytes > 64 * 1024) {
    if (bytes > 160 * 1024) {
      set_error("small: a tile needs %zu bytes of LDS (> 160 KB)", bytes);
      return GCMI_ERR_UNSUPPORTED;
    }
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes) != hipSuccess) {
      set_error("small: hipFuncSetAttribute(max dynamic LDS = %zu) failed", bytes);
      return GCMI_ERR_LAUNCH;
    }
  }
  return GCMI_OK;
}

#define SRUN(call)          \
  do {                      \
    int rc__ = (call);      \
    if (rc__) return rc__;  \
  } while (0)

struct SmallCtx {
  const gcmi_model_desc* m;
  const float* params;
  const gcmi_model_io* io;
  SmallWs w;
  float* ws;
  double* accs;
  hipStream_t st;
  bool training;
  int parity;
};

static inline float* gc_of(const SmallCtx& c, int l) { return c.ws + c.w.gc[c.parity][l]; }
static inline double* acc_of(const SmallCtx& c, int l) {
  return c.accs + (l < c.m->n_layers ? c.w.acc[c.parity][l] : c.w.acc_dense);
}

static BnArgs bn_args(const SmallCtx& c, int layer, int n_rows) {
  BnArgs b;
  memset(&b, 0, sizeof(b));
  const gcmi_model_desc* m = c.m;
  if (!m->batch_norm) return b;
  b.mode = c.training ? 1 : 2;
  b.acc = acc_of(c, layer);
  b.rm = c.io->d_bn_running_mean[layer];
  b.rv = c.io->d_bn_running_var[layer];
  b.gamma = c.params + m->off_bn_gamma[layer];
  b.beta = c.params + m->off_bn_beta[layer];
  b.eps = m->bn_eps;
  b.n_rows = n_rows;
  b.inv_n = n_rows > 0 ? 1.0 / (double)n_rows : 0.0;
  return b;
}

static int launch_conv_fwd(const SmallGraph& g, const float* x, int ldx, int K, const float* Wl, const float* bl,
                           int W, float* out, double* acc, bool pool_in, const BnArgs& bn_in, float* pool_out,
                           uint8_t* arg_out, hipStream_t st) {
  const int K4 = (K + 3) & ~3;
  const size_t lds = sizeof(float) * (2 * kTileRows * pitch_a(K4) + 2 * K4);
  const dim3 grid(g.n_tiles), block(kSBlock);
#define CF(NT, PI)                                                                                              \
  hipLaunchKernelGGL((small_conv_fwd_kernel<NT, PI>), grid, block, lds, st, g, x, ldx, K, Wl, bl, out, acc, bn_in, \
                     pool_out, arg_out)
  if (pool_in) {
    switch (W / 64) {
      case 1: CF(1, true); break;
      case 2: CF(2, true); break;
      case 3: CF(3, true); break;
      default: CF(4, true); break;
    }
  } else {
    switch (W / 64) {
      case 1: CF(1, false); break;
      case 2: CF(2, false); break;
      case 3: CF(3, false); break;
      default: CF(4, false); break;
    }
  }
#undef CF
  GCMI_CHECK_LAUNCH("small_conv_fwd");
  return GCMI_OK;
}

// the GraphConv stack (with the pools between the layers) on `st`
static int small_conv_stack(const SmallCtx& c, const SmallGraph& g, const float* x, int64_t ldx, hipStream_t st) {
  const gcmi_model_desc* m = c.m;
  const int L = m->n_layers;
  const int N = g.n_atoms;
  int K = m->n_feat_in;
  if (g.n_tiles == 0) return GCMI_OK;
  GCMI_CHECK_ARG(ldx % 4 == 0 && ldx >= ((K + 3) & ~3) && aligned16(x),
                 "small: atom features must be 16-byte aligned rows (ld %lld for %d columns)", (long long)ldx, K);
  for (int l = 0; l < L; ++l) {
    const int W = m->conv_width[l];
    double* acc = (m->batch_norm && c.training) ? acc_of(c, l) : nullptr;
    if (l == 0) {
      BnArgs none;
      memset(&none, 0, sizeof(none));
      SRUN(launch_conv_fwd(g, x, (int)ldx, K, c.params + m->off_conv_w[l], c.params + m->off_conv_b[l], W, gc_of(c, l),
                           acc, false, none, nullptr, nullptr, st));
    } else {
      // the GraphPool between the layers is computed by the consumer's gather; its rows and arg-max are written
      // only when a backward pass will read them ("full" gradient mode)
      const bool keep = c.training && m->grad_mode == 1;
      SRUN(launch_conv_fwd(g, gc_of(c, l - 1), K, K, c.params + m->off_conv_w[l], c.params + m->off_conv_b[l], W,
                           gc_of(c, l), acc, true, bn_args(c, l - 1, N), keep ? c.ws + c.w.pool[l - 1] : nullptr,
                           keep ? reinterpret_cast<uint8_t*>(c.ws + c.w.arg[l - 1]) : nullptr, st));
    }
    K = W;
  }
  return GCMI_OK;
}

// last pool + dense layer; leaves the dense output in the workspace
static int small_dense_stage(const SmallCtx& c, const SmallGraph& g) {
  const gcmi_model_desc* m = c.m;
  const int L = m->n_layers;
  const int N = g.n_atoms;
  if (g.n_tiles == 0) return GCMI_OK;
  const int Wl = m->conv_width[L - 1], F = m->dense_width;
  double* accD = (m->batch_norm && c.training) ? acc_of(c, L) : nullptr;
  const size_t lds = sizeof(float) * (2 * Wl + kTileRows * pitch_a(Wl));
  uint8_t* arg = c.training ? reinterpret_cast<uint8_t*>(c.ws + c.w.arg[L - 1]) : nullptr;
  const float* Wd = c.params + m->off_dense_w;
  const float* bd = c.params + m->off_dense_b;
  const dim3 grid(g.n_tiles), block(kSBlock);
#define PD(NT)                                                                                                      \
  hipLaunchKernelGGL(small_pool_dense_fwd_kernel<NT>, grid, block, lds, c.st, g, gc_of(c, L - 1), Wl,                \
                     bn_args(c, L - 1, N), c.ws + c.w.pool[L - 1], arg, Wd, bd, c.ws + c.w.dense, accD)
  switch (F / 64) {
    case 1: PD(1); break;
    case 2: PD(2); break;
    case 3: PD(3); break;
    default: PD(4); break;
  }
#undef PD
  GCMI_CHECK_LAUNCH("small_pool_dense_fwd");
  return GCMI_OK;
}

static int small_forward_body(const SmallCtx& c, const SmallGraph& g, const float* x, int64_t ldx) {
  SRUN(small_conv_stack(c, g, x, ldx, c.st));
  return small_dense_stage(c, g);
}

static int launch_readout(const SmallCtx& c, const SmallGraph& g, ReadoutArgs& a) {
  const gcmi_model_desc* m = c.m;
  const int F = m->dense_width, TC = m->n_tasks * m->n_classes;
  a.dense = c.ws + c.w.dense;
  a.bn = bn_args(c, m->n_layers, g.n_atoms);
  if (g.n_atoms == 0) a.bn.mode = 0;  // no rows: every molecule is empty, nothing is normalised
  a.Wh = c.params + m->off_head_w;
  a.bh = c.params + m->off_head_b;
  a.F = F;
  a.T = m->n_tasks;
  a.C = m->n_classes;
  a.mode = m->mode;
  const int TCp = (TC + 3) & ~3;
  // fingerprint [2F], logits + dlogits [2 TCp], gradient [2F], wave quarters [20 F], head partials [8 TCp],
  // gradient partials [8 F]
  const size_t lds = sizeof(float) * (size_t)(4 * F + 2 * TCp + 20 * F + 8 * TCp + 8 * F);
  a.wh_in_lds = 0;
  const dim3 grid(g.n_mols), block(kSBlock);
  switch (F) {
    case 64:
      SRUN(ensure_lds(small_readout_kernel<16>, lds));
      hipLaunchKernelGGL(small_readout_kernel<16>, grid, block, lds, c.st, g, a);
      break;
    case 128:
      SRUN(ensure_lds(small_readout_kernel<32>, lds));
      hipLaunchKernelGGL(small_readout_kernel<32>, grid, block, lds, c.st, g, a);
      break;
    default:
      SRUN(ensure_lds(small_readout_kernel<64>, lds));
      hipLaunchKernelGGL(small_readout_kernel<64>, grid, block, lds, c.st, g, a);
      break;
  }
  GCMI_CHECK_LAUNCH("small_readout");
  return GCMI_OK;
}

static int small_backward(const SmallCtx& c, const SmallGraph& g, const gcmi_small_batch* b, float* grads) {
  const gcmi_model_desc* m = c.m;
  const int L = m->n_layers, N = g.n_atoms;
  const int F = m->dense_width, TC = m->n_tasks * m->n_classes, Kd = m->conv_width[L - 1];
  const bool full = m->grad_mode == 1;
  const bool need_dpool = full || m->batch_norm;
  DenseBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.dense = c.ws + c.w.dense;
  a.pool = c.ws + c.w.pool[L - 1];
  a.g2 = c.ws + c.w.g2;
  a.argrow = reinterpret_cast<const int32_t*>(c.ws + c.w.argrow);
  a.bn = bn_args(c, L, N);
  a.bsum = c.accs + c.w.bsum;
  a.Wd = c.params + m->off_dense_w;
  a.dpool = need_dpool ? c.ws + c.w.dpool : nullptr;
  a.dWd = grads + m->off_dense_w;
  a.dbd = grads + m->off_dense_b;
  a.dgamma = m->batch_norm ? grads + m->off_bn_gamma[L] : nullptr;
  a.dbeta = m->batch_norm ? grads + m->off_bn_beta[L] : nullptr;
  a.dlogits = c.ws + c.w.dlogits;
  a.fp = c.ws + c.w.fp;
  a.dWh = grads + m->off_head_w;
  a.dbh = grads + m->off_head_b;
  a.F = F;
  a.K = Kd;
  a.TC = TC;
  a.n_slabs = (N + kSlabRows - 1) / kSlabRows;
  a.n_head_blocks = TC;
  SmallGraph gd = g;
  if (!need_dpool) gd.n_tiles = 0;  // nothing in front of the dense layer trains: no dgrad tiles
  {
    const size_t lds_tile = sizeof(float) * (3 * F + kTileRows * pitch_a(F));
    const size_t lds_slab = sizeof(float) * (3 * F + kSlabRows * (pitch_t(F) + pitch_t(Kd)));
    const size_t lds_head = sizeof(float) * (4 * 2 * F + 4);
    const size_t lds = std::max(lds_tile, std::max(lds_slab, lds_head));
    const dim3 grid(gd.n_tiles + a.n_slabs + a.n_head_blocks), block(kSBlock);
    switch (Kd / 64) {
      case 1:
        SRUN(ensure_lds(small_dense_bwd_kernel<1>, lds));
        hipLaunchKernelGGL(small_dense_bwd_kernel<1>, grid, block, lds, c.st, gd, a);
        break;
      case 2:
        SRUN(ensure_lds(small_dense_bwd_kernel<2>, lds));
        hipLaunchKernelGGL(small_dense_bwd_kernel<2>, grid, block, lds, c.st, gd, a);
        break;
      case 3:
        SRUN(ensure_lds(small_dense_bwd_kernel<3>, lds));
        hipLaunchKernelGGL(small_dense_bwd_kernel<3>, grid, block, lds, c.st, gd, a);
        break;
      default:
        SRUN(ensure_lds(small_dense_bwd_kernel<4>, lds));
        hipLaunchKernelGGL(small_dense_bwd_kernel<4>, grid, block, lds, c.st, gd, a);
        break;
    }
    GCMI_CHECK_LAUNCH("small_dense_bwd");
  }
  if (!need_dpool || g.n_tiles == 0) return GCMI_OK;
  const float* dpool = c.ws + c.w.dpool;
  for (int l = L - 1; l >= 0; --l) {
    const int W = m->conv_width[l];
    const int K = l == 0 ? m->n_feat_in : m->conv_width[l - 1];
    float* dA = full ? c.ws + c.w.dA[l] : nullptr;
    hipLaunchKernelGGL(small_pool_bwd_kernel, dim3(g.n_tiles), dim3(kSBlock), sizeof(float) * (2 * W + kTileRows * 2 * W),
                       c.st, g, dpool, reinterpret_cast<const uint8_t*>(c.ws + c.w.arg[l]), gc_of(c, l), W,
                       bn_args(c, l, N), dA, m->batch_norm ? grads + m->off_bn_gamma[l] : nullptr,
                       m->batch_norm ? grads + m->off_bn_beta[l] : nullptr);
    GCMI_CHECK_LAUNCH("small_pool_bwd");
    if (!full) break;  // reference semantics: nothing in front of a GraphConv output trains
    ConvBwdArgs cb;
    memset(&cb, 0, sizeof(cb));
    cb.dA = dA;
    cb.gc = gc_of(c, l);
    cb.bn = bn_args(c, l, N);
    cb.dgamma = m->batch_norm ? grads + m->off_bn_gamma[l] : nullptr;
    cb.dbeta = m->batch_norm ? grads + m->off_bn_beta[l] : nullptr;
    cb.x = l == 0 ? b->d_atom_features : c.ws + c.w.pool[l - 1];
    cb.ldx = l == 0 ? (int)b->ld_features : m->conv_width[l - 1];
    cb.x_bf = (l > 0 && g.bf16) ? 1 : 0;
    cb.K = K;
    cb.W = W;
    cb.Wl = c.params + m->off_conv_w[l];
    cb.dWl = grads + m->off_conv_w[l];
    cb.dbl = grads + m->off_conv_b[l];
    cb.dS = l > 0 ? c.ws + c.w.dS : nullptr;
    cb.dXs = l > 0 ? c.ws + c.w.dXs : nullptr;
    int s = 0;
    for (int d = 0; d < GCMI_MAX_DEG + 2; ++d) {
      cb.slab_start[d] = s;
      if (d <= g.max_deg) s += (g.deg_start[d + 1] - g.deg_start[d] + kSlabRows - 1) / kSlabRows;
    }
    cb.n_slabs = s;
    const int K4 = (K + 3) & ~3;
    const size_t lds_tile = sizeof(float) * (3 * W + kTileRows * pitch_a(W));
    const size_t lds_slab = sizeof(float) * (3 * W + kSlabRows * (pitch_t(W) + 2 * pitch_t(K4)));
    const size_t lds = lds_tile > lds_slab ? lds_tile : lds_slab;
    SRUN(ensure_lds(small_conv_bwd_kernel, lds));
    hipLaunchKernelGGL(small_conv_bwd_kernel, dim3((l > 0 ? g.n_tiles : 0) + cb.n_slabs), dim3(kSBlock), lds, c.st, g,
                       cb);
    GCMI_CHECK_LAUNCH("small_conv_bwd");
    if (l == 0) break;
    hipLaunchKernelGGL(small_gather_add_kernel, dim3(g.n_tiles), dim3(kSBlock), 0, c.st, g, c.ws + c.w.dXs,
                       c.ws + c.w.dS, K4, c.ws + c.w.dpool);
    GCMI_CHECK_LAUNCH("small_gather_add");
    dpool = c.ws + c.w.dpool;
  }
  return GCMI_OK;
}

// the second stream of gcmi_small_fit and its events: one set per DEVICE (created on the device that is current at
// the first call for it, and only ever used there); `busy` serialises the calls that use it -- two host threads
// enqueueing gcmi_small_fit at once would otherwise share ev_start / ev_conv / ev_free and corrupt each other's order
struct SideStream {
  hipStream_t st;
  hipEvent_t ev_start, ev_conv[2], ev_free[2];
  std::mutex busy;
};
static SideStream* side_stream() {
  constexpr int kMaxDev = 64;
  static std::mutex mu;
  static SideStream* per_dev[kMaxDev] = {nullptr};
  static bool failed[kMaxDev] = {false};
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;  // the one-stream path
  std::lock_guard<std::mutex> lock(mu);
  if (per_dev[dev] || failed[dev]) return per_dev[dev];
  SideStream*& s = per_dev[dev];
  SideStream* n = new SideStream();
  bool ok = hipStreamCreateWithFlags(&n->st, hipStreamNonBlocking) == hipSuccess;
  hipEvent_t* evs[5] = {&n->ev_start, &n->ev_conv[0], &n->ev_conv[1], &n->ev_free[0], &n->ev_free[1]};
  for (int i = 0; ok && i < 5; ++i) ok = hipEventCreateWithFlags(evs[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    failed[dev] = true;  // no overlap then: the one-stream path is always valid
    delete n;
    return nullptr;
  }
  s = n;
  return s;
}

}  // namespace gcmi

using namespace gcmi;

extern "C" {

int64_t gcmi_small_workspace_floats(const gcmi_model_desc* m, int64_t max_atoms, int64_t max_mols) {
  if (small_check(m) != GCMI_OK || max_atoms < 0 || max_mols < 0) return -1;
  return small_carve(m, max_atoms, max_mols).total;
}

int gcmi_small_fit(const gcmi_model_desc* m, float* d_params, float* d_grads, float* d_adam_m, float* d_adam_v,
                   const gcmi_model_io* io, const gcmi_small_batch* batches, int64_t n_batches, int64_t ws_atoms,
                   int64_t ws_mols, float lr, float beta1, float beta2, float eps, int64_t first_step,
                   float* d_losses, int64_t* grad_lo, int64_t* grad_hi, void* stream) {
  return gcmi_small_fit_dp(m, d_params, d_grads, d_adam_m, d_adam_v, io, batches, n_batches, ws_atoms, ws_mols, lr, beta1,
                           beta2, eps, first_step, d_losses, grad_lo, grad_hi, nullptr, nullptr, stream);
}

int gcmi_small_fit_dp(const gcmi_model_desc* m, float* d_params, float* d_grads, float* d_adam_m, float* d_adam_v,
                      const gcmi_model_io* io, const gcmi_small_batch* batches, int64_t n_batches, int64_t ws_atoms,
                      int64_t ws_mols, float lr, float beta1, float beta2, float eps, int64_t first_step,
                      float* d_losses, int64_t* grad_lo, int64_t* grad_hi, gcmi_grad_sync_fn sync, void* sync_ctx,
                      void* stream) {
  SRUN(small_check(m));
  GCMI_CHECK_ARG(d_params && d_grads && d_adam_m && d_adam_v && io && io->d_workspace && (batches || n_batches == 0),
                 "small_fit: NULL buffer");
  GCMI_CHECK_ARG(n_batches >= 0 && first_step >= 1, "small_fit: bad n_batches / first_step");
  const int L = m->n_layers;
  const bool full = m->grad_mode == 1;
  const int64_t lo = full ? 0 : (m->batch_norm ? m->off_bn_gamma[L - 1] : m->off_dense_w);
  const int64_t hi = m->n_params;
  GCMI_CHECK_ARG(lo % 4 == 0 && hi % 4 == 0 && lo < hi, "small_fit: parameter blocks must be 16-byte aligned");
  if (grad_lo) *grad_lo = lo;
  if (grad_hi) *grad_hi = hi;
  if (n_batches == 0) return GCMI_OK;
  SmallCtx c;
  c.m = m;
  c.params = d_params;
  c.io = io;
  c.w = small_carve(m, ws_atoms, ws_mols);
  c.ws = io->d_workspace;
  c.accs = reinterpret_cast<double*>(c.ws + c.w.acc0);
  c.st = (hipStream_t)stream;
  c.training = true;
  c.parity = 0;
  if (m->batch_norm)
    for (int l = 0; l <= L; ++l)
      GCMI_CHECK_ARG(io->d_bn_running_mean[l] && io->d_bn_running_var[l], "small_fit: NULL running statistics");
  // clean accumulators and a clean gradient range at entry; every step leaves them clean for the next
  if (hipMemsetAsync(c.accs, 0, sizeof(double) * (size_t)c.w.acc_doubles, c.st) != hipSuccess ||
      hipMemsetAsync(d_grads + lo, 0, sizeof(float) * (size_t)(hi - lo), c.st) != hipSuccess) {
    set_error("small_fit: memset failed");
    return GCMI_ERR_LAUNCH;
  }
  // Reference gradient mode trains nothing the GraphConv stack reads (frozen weights, frozen BatchNorm 0..L-2), so
  // the stack of step i+1 runs on a second stream while step i's dense layer, readout, backward and Adam run on the
  // caller's; the two parities of the conv outputs / statistics keep them apart.  (GCMI_SMALL_OVERLAP=0: one stream.)
  static const bool overlap_env = !(getenv("GCMI_SMALL_OVERLAP") && atoi(getenv("GCMI_SMALL_OVERLAP")) == 0);
  SideStream* side = (!full && overlap_env && n_batches > 1) ? side_stream() : nullptr;
  std::unique_lock<std::mutex> side_lock;
  if (side) side_lock = std::unique_lock<std::mutex>(side->busy);  // held while this call enqueues
  std::vector<SmallGraph> graphs((size_t)n_batches);
  for (int64_t i = 0; i < n_batches; ++i) {
    const gcmi_small_batch* b = batches + i;
    GCMI_CHECK_ARG(b->graph.n_atoms <= ws_atoms && b->graph.n_mols <= ws_mols,
                   "small_fit: batch %lld (%d atoms, %d molecules) exceeds the workspace (%lld, %lld)", (long long)i,
                   b->graph.n_atoms, b->graph.n_mols, (long long)ws_atoms, (long long)ws_mols);
    GCMI_CHECK_ARG(b->d_labels && b->n_rows > 0 && b->n_rows <= b->graph.n_mols, "small_fit: batch %lld: bad labels / n_rows",
                   (long long)i);
    GCMI_CHECK_ARG(b->graph.n_atoms == 0 || b->d_atom_features, "small_fit: NULL atom features");
    GCMI_CHECK_ARG(b->graph.max_deg == m->max_deg, "small_fit: graph max_deg %d != model max_deg %d", b->graph.max_deg,
                   m->max_deg);
    SRUN(make_small_graph(&b->graph, true, &graphs[(size_t)i]));
    graphs[(size_t)i].bf16 = m->storage >= 1 ? 1 : 0;  // (2: gradient streams too -- no such streams here, the batch lives in L2)
  }
  auto slot_of = [](int64_t i) { return (int)(((i / kAhead) & 1) * kAhead + i % kAhead); };
  const int64_t n_groups = (n_batches + kAhead - 1) / kAhead;
  auto conv_group = [&](int64_t k) -> int {  // the conv stacks of group k on the side stream
    const int set = (int)(k & 1);
    if (k >= 2 && hipStreamWaitEvent(side->st, side->ev_free[set], 0) != hipSuccess) return GCMI_ERR_LAUNCH;
    for (int64_t i = k * kAhead; i < std::min(n_batches, (k + 1) * kAhead); ++i) {
      SmallCtx cc = c;
      cc.parity = slot_of(i);
      SRUN(small_conv_stack(cc, graphs[(size_t)i], batches[i].d_atom_features, batches[i].ld_features, side->st));
    }
    if (hipEventRecord(side->ev_conv[set], side->st) != hipSuccess) return GCMI_ERR_LAUNCH;
    return GCMI_OK;
  };
  if (side) {
    // the side stream starts behind everything queued on the caller's stream so far (copies, the memsets above)
    if (hipEventRecord(side->ev_start, c.st) != hipSuccess || hipStreamWaitEvent(side->st, side->ev_start, 0) != hipSuccess) {
      set_error("small_fit: event record / wait failed");
      return GCMI_ERR_LAUNCH;
    }
    SRUN(conv_group(0));
  }
  for (int64_t i = 0; i < n_batches; ++i) {
    const gcmi_small_batch* b = batches + i;
    const SmallGraph& g = graphs[(size_t)i];
    c.parity = side ? slot_of(i) : 0;
    if (side) {
      if (i % kAhead == 0) {
        const int64_t k = i / kAhead;
        if (k + 1 < n_groups) SRUN(conv_group(k + 1));
        if (hipStreamWaitEvent(c.st, side->ev_conv[k & 1], 0) != hipSuccess) {
          set_error("small_fit: stream wait failed");
          return GCMI_ERR_LAUNCH;
        }
      }
      SRUN(small_dense_stage(c, g));
    } else {
      SRUN(small_forward_body(c, g, b->d_atom_features, b->ld_features));
    }
    ReadoutArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.fp = c.ws + c.w.fp;
    ra.logits = c.ws + c.w.logits;
    ra.probs = nullptr;
    ra.labels = b->d_labels;
    ra.weights = b->d_weights;
    ra.n_rows = (int)b->n_rows;
    ra.inv_count = 1.f / (float)(b->n_rows * m->n_tasks);
    ra.dlogits = c.ws + c.w.dlogits;
    ra.g2 = c.ws + c.w.g2;
    ra.argrow = reinterpret_cast<int32_t*>(c.ws + c.w.argrow);
    ra.loss_acc = c.accs + c.w.loss;
    ra.bsum = m->batch_norm ? c.accs + c.w.bsum : nullptr;
    SRUN(launch_readout(c, g, ra));
    SRUN(small_backward(c, g, b, d_grads));
    // data parallel: the ranks' gradients of the trained range are summed here, in order on the step's stream,
    // between the backward launches above and the Adam launch below (the caller's callback enqueues the all-reduce)
    if (sync != nullptr && sync(sync_ctx, d_grads + lo, hi - lo, stream) != 0) {
      set_error("small_fit: the gradient all-reduce callback failed at step %lld", (long long)(first_step + i));
      return GCMI_ERR_LAUNCH;
    }
    // Adam over the trained range, loss, running statistics, counters, zeroing
    StepEnd se;
    memset(&se, 0, sizeof(se));
    const int64_t step = first_step + i;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    se.p = d_params;
    se.grad = d_grads;
    se.m = d_adam_m;
    se.v = d_adam_v;
    se.lo = lo;
    se.hi = hi;
    se.one_minus_b1 = 1.f - beta1;
    se.b2 = beta2;
    se.one_minus_b2 = 1.f - beta2;
    se.step_size = (float)((double)lr / bc1);
    se.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    se.eps = eps;
    se.loss_acc = c.accs + c.w.loss;
    se.loss_out = d_losses ? d_losses + i : nullptr;
    se.inv_count = ra.inv_count;
    se.n_bn = (m->batch_norm && g.n_atoms > 0) ? L + 1 : 0;
    se.n_rows = g.n_atoms;
    se.momentum = m->bn_momentum;
    for (int l = 0; l <= L; ++l) {
      se.acc[l] = acc_of(c, l);
      se.width[l] = l < L ? m->conv_width[l] : m->dense_width;
      se.rm[l] = io->d_bn_running_mean[l];
      se.rv[l] = io->d_bn_running_var[l];
      se.tracked[l] = io->d_bn_batches_tracked[l];
    }
    se.zero_from = c.accs + c.w.par_begin[c.parity];
    se.zero_doubles = c.w.par_doubles;
    se.zero2_from = c.accs + c.w.shared_begin;
    se.zero2_doubles = c.w.shared_doubles;
    const int64_t n4 = (hi - lo) / 4;
    int blocks = (int)((n4 + kSBlock - 1) / kSBlock);
    if (blocks < 1) blocks = 1;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(small_step_end_kernel, dim3(blocks), dim3(kSBlock), 0, c.st, se);
    GCMI_CHECK_LAUNCH("small_step_end");
    if (side && ((i + 1) % kAhead == 0 || i + 1 == n_batches) &&
        hipEventRecord(side->ev_free[(i / kAhead) & 1], c.st) != hipSuccess) {
      set_error("small_fit: event record failed");
      return GCMI_ERR_LAUNCH;
    }
  }
  return GCMI_OK;
}

int gcmi_small_predict(const gcmi_model_desc* m, const float* d_params, const gcmi_model_io* io,
                       const gcmi_small_batch* batches, int64_t n_batches, int64_t ws_atoms, int64_t ws_mols,
                       void* stream) {
  SRUN(small_check(m));
  GCMI_CHECK_ARG(d_params && io && io->d_workspace && (batches || n_batches == 0), "small_predict: NULL buffer");
  SmallCtx c;
  c.m = m;
  c.params = d_params;
  c.io = io;
  c.w = small_carve(m, ws_atoms, ws_mols);
  c.ws = io->d_workspace;
  c.accs = reinterpret_cast<double*>(c.ws + c.w.acc0);
  c.st = (hipStream_t)stream;
  c.training = false;
  c.parity = 0;
  if (m->batch_norm)
    for (int l = 0; l <= m->n_layers; ++l)
      GCMI_CHECK_ARG(io->d_bn_running_mean[l] && io->d_bn_running_var[l], "small_predict: NULL running statistics");
  for (int64_t i = 0; i < n_batches; ++i) {
    const gcmi_small_batch* b = batches + i;
    GCMI_CHECK_ARG(b->graph.n_atoms <= ws_atoms && b->graph.n_mols <= ws_mols,
                   "small_predict: batch %lld exceeds the workspace", (long long)i);
    GCMI_CHECK_ARG(b->d_logits && b->d_fingerprint, "small_predict: NULL output");
    GCMI_CHECK_ARG(b->graph.n_atoms == 0 || b->d_atom_features, "small_predict: NULL atom features");
    GCMI_CHECK_ARG(b->graph.max_deg == m->max_deg, "small_predict: graph max_deg != model max_deg");
    SmallGraph g;
    SRUN(make_small_graph(&b->graph, false, &g));
    g.bf16 = m->storage >= 1 ? 1 : 0;
    SRUN(small_forward_body(c, g, b->d_atom_features, b->ld_features));
    ReadoutArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.fp = b->d_fingerprint;
    ra.logits = b->d_logits;
    ra.probs = m->mode == 0 ? b->d_probs : nullptr;
    SRUN(launch_readout(c, g, ra));
  }
  return GCMI_OK;
}

}  // extern "C"
