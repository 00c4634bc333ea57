// K2: BatchNorm1d over atom rows (training statistics, folded scale/shift,
// backward).  nn.BatchNorm1d(eps=1e-3, momentum=0.99, affine, running stats)
// at models/torch_models/graphconvmodel.py:150-165.
//
// One streaming pass per reduction: every thread owns one 16-byte column chunk
// and a strided set of rows, accumulates sum and sum of squares in fp64
// registers (so E[x^2]-E[x]^2 has no cancellation problem at fp32 output
// precision), the row-lanes of a workgroup are combined through LDS, and each
// workgroup issues one fp64 atomic per column into a 2F-double scratch.  A
// one-workgroup finalize turns that into mean / invstd / folded scale+shift
// and updates the running statistics.  Consumers (gather-max, readout) apply
// scale/shift on the fly, so the normalised tensor is never written to HBM.
// Bound: HBM, N*4F bytes read per pass.
#include "common.h"
#include "split_bf16.h"

namespace gcmi {

constexpr int kBBlock = 256;
constexpr int kRowsPerBlock = 512;     // smallest share of rows a workgroup takes
constexpr int kResidentBlocks = 2048;  // 256 CUs x 8 workgroups of 256 threads
constexpr int kReplicas = kBnReplicas;  // accumulator replicas: same-address fp64 atomics serialise
// scratch layout (doubles): [0, 2F) backward coefficient vectors (3F floats), then kReplicas blocks
// of 2F partial sums.  The kernels that consume the partial sums zero them again, so a scratch
// that starts clean stays clean (the whole-model path zeroes it once per pass).

// The gradient w.r.t. the readout input, recomputed instead of read (GraphGather backward fused
// into the BatchNorm backward that consumes it): with g2[m] = [dsum | dmax] of molecule m (tanh
// derivative already applied) and arg[m,f] = row of the first maximum,
//   dy[r, f] = g2[mol(r)][f] + (arg[mol(r)][f] == r) * g2[mol(r)][F + f].
struct ReadoutGrad {
  const int32_t* membership;  // N
  const float* g2;            // n_mols x ldg2 (>= 2F)
  int64_t ldg2;
  const int32_t* arg;         // n_mols x F
  // optional: what the column sums need to come from per-molecule data (readout_bn_sums_kernel)
  const float* rawsum = nullptr;  // n_mols x 2F: [row sums | arg-max row's value] of the BatchNorm input
  const int32_t* runs = nullptr;  // n_mols x n_deg x 2 row runs
  int32_t n_mols = 0, n_deg = 0;
};

template <int V>
__device__ __forceinline__ void readout_dy(const ReadoutGrad& rg, int64_t r, int c, int n_feat, float (&dy)[V]) {
  const int m = rg.membership[r];
  const float* grow = rg.g2 + (int64_t)m * rg.ldg2;
  const int32_t* arow = rg.arg + (int64_t)m * n_feat;
  if constexpr (V == 4) {
    const float4 s4 = *reinterpret_cast<const float4*>(grow + c);
    const float4 m4 = *reinterpret_cast<const float4*>(grow + n_feat + c);
    const int4 a4 = *reinterpret_cast<const int4*>(arow + c);
    dy[0] = s4.x + (a4.x == r ? m4.x : 0.f);
    dy[1] = s4.y + (a4.y == r ? m4.y : 0.f);
    dy[2] = s4.z + (a4.z == r ? m4.z : 0.f);
    dy[3] = s4.w + (a4.w == r ? m4.w : 0.f);
  } else {
    dy[0] = grow[c] + (arow[c] == r ? grow[n_feat + c] : 0.f);
  }
}

// sums[0:F] += sum_r a[r,:],  sums[F:2F] += sum_r a[r,:]*b[r,:]
// MODE 0: b = a (sum of squares).  MODE 1: b = (x - mean)*invstd (x given), a = dy.
// MODE 2: as MODE 1 with dy recomputed from the readout gradient (rg) instead of read from a.
// XH (V == 4, MODE 1): x is stored as bf16 (gcmi_model_desc.storage == 1; ldx counts elements).
// AH (with XH): a (the incoming gradient) is stored as bf16 too (storage == 2).
template <int V, int MODE, bool XH = false, bool AH = false>
__global__ void __launch_bounds__(kBBlock)
col_sums_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ x, int64_t ldx,
                const float* __restrict__ mean, const float* __restrict__ invstd, int64_t n_rows,
                int64_t rows_per_block, int n_feat, int lpr, int lx, double* __restrict__ sums, ReadoutGrad rg,
                int rev, const float* __restrict__ only_if_gamma = nullptr,
                const float* __restrict__ only_if_beta = nullptr) {
  // (bn_bwd_pool_impl) needed only where the pooled sums are ill-conditioned: otherwise gone before the first load
  if (only_if_gamma != nullptr && !bn_pool_ill_conditioned(only_if_gamma, only_if_beta, n_feat)) return;
  __shared__ double red[2 * kBBlock * 4];
  const int ry = kBBlock / lx;  // row lanes
  const int ty = threadIdx.x / lx;
  const int tx = threadIdx.x - ty * lx;
  const int64_t r_begin = (int64_t)(rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * rows_per_block;
  const int64_t r_end = (r_begin + rows_per_block < n_rows) ? r_begin + rows_per_block : n_rows;
  // four rows per round, the next round's loads issued before this round is added up: one row per round
  // leaves a single 16-byte load in flight per lane and the loop runs at memory latency
  constexpr int R = 4;
  struct Round {
    float av[R][V], xv[R][V];
  };
  auto load_round = [&](int64_t r0, int c, Round& t) {
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int64_t r = r0 + (int64_t)u * ry;
      const int64_t rc = r < r_end ? r : r_end - 1;  // clamped: the load is unconditional, the add is not
      if constexpr (MODE == 2) readout_dy<V>(rg, rc, c, n_feat, t.av[u]);
      if constexpr (V == 4) {
        if (MODE != 2) {
          float4 t4;
          if constexpr (AH) t4 = widen4(*reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(a) + rc * lda + c));
          else t4 = *reinterpret_cast<const float4*>(a + rc * lda + c);
          t.av[u][0] = t4.x; t.av[u][1] = t4.y; t.av[u][2] = t4.z; t.av[u][3] = t4.w;
        }
        if (MODE >= 1) {
          float4 u4;
          if constexpr (XH) u4 = widen4(*reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(x) + rc * ldx + c));
          else u4 = *reinterpret_cast<const float4*>(x + rc * ldx + c);
          t.xv[u][0] = u4.x; t.xv[u][1] = u4.y; t.xv[u][2] = u4.z; t.xv[u][3] = u4.w;
        }
      } else {
        if (MODE != 2) t.av[u][0] = a[rc * lda + c];
        if (MODE >= 1) t.xv[u][0] = x[rc * ldx + c];
      }
    }
  };
  for (int cc0 = 0; cc0 < lpr; cc0 += lx) {  // uniform trip count: barriers inside
    const int cc = cc0 + tx;
    const bool active = cc < lpr && ty < ry && r_begin < r_end;
    const int c = (cc < lpr ? cc : 0) * V;
    double s1[V], s2[V];
    float mu[V], is[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      s1[q] = 0.0;
      s2[q] = 0.0;
      mu[q] = MODE >= 1 ? mean[c + q] : 0.f;
      is[q] = MODE >= 1 ? invstd[c + q] : 0.f;
    }
    if (active) {
      // MODE 2 chains three dependent loads per row (membership -> gradient row -> arg-max row): holding a
      // second round of those in flight costs more registers than it hides latency (291 us against 203)
      constexpr bool PREFETCH = MODE != 2;
      Round cur, nxt;
      const int64_t step = (int64_t)R * ry;
      if (PREFETCH) load_round(r_begin + ty, c, cur);
      for (int64_t r0 = r_begin + ty; r0 < r_end; r0 += step) {
        if (!PREFETCH) load_round(r0, c, cur);
        if (PREFETCH && r0 + step < r_end) load_round(r0 + step, c, nxt);
#pragma unroll
        for (int u = 0; u < R; ++u) {
          if (r0 + (int64_t)u * ry >= r_end) continue;
#pragma unroll
          for (int q = 0; q < V; ++q) {
            s1[q] += (double)cur.av[u][q];
            if (MODE == 0)
              s2[q] += (double)cur.av[u][q] * (double)cur.av[u][q];
            else
              s2[q] += (double)cur.av[u][q] * (double)((cur.xv[u][q] - mu[q]) * is[q]);
          }
        }
        if (PREFETCH) cur = nxt;
      }
    }
    // combine the row lanes of this column chunk
#pragma unroll
    for (int q = 0; q < V; ++q) {
      red[(threadIdx.x * V + q) * 2 + 0] = s1[q];
      red[(threadIdx.x * V + q) * 2 + 1] = s2[q];
    }
    __syncthreads();
    if (ty == 0 && cc < lpr) {
#pragma unroll
      for (int q = 0; q < V; ++q) {
        double t1 = 0.0, t2 = 0.0;
        for (int y = 0; y < ry; ++y) {
          t1 += red[((y * lx + tx) * V + q) * 2 + 0];
          t2 += red[((y * lx + tx) * V + q) * 2 + 1];
        }
        double* rep = sums + (size_t)2 * n_feat * (1 + (blockIdx.x % kReplicas));
        atomicAdd(rep + c + q, t1);
        atomicAdd(rep + n_feat + c + q, t2);
      }
    }
    __syncthreads();
  }
}

// Column c of the reduced sums: adds the replicas up and leaves them zero.  32 lanes per column, one replica each
// (kReplicas == 32), combined by shuffles in a fixed order (a serial walk is 64 loads per thread issued one behind the
// other: 6 us per launch).
__device__ __forceinline__ void take_sums_lanes(double* __restrict__ sums, int n_feat, int c, int r, bool ok, double& t1,
                                                double& t2) {
  static_assert(kReplicas == 32, "one lane per replica");
  t1 = 0.0;
  t2 = 0.0;
  if (ok) {
    double* rep = sums + (size_t)2 * n_feat * (1 + r);
    t1 = rep[c];
    t2 = rep[n_feat + c];
    rep[c] = 0.0;
    rep[n_feat + c] = 0.0;
  }
#pragma unroll
  for (int o = kReplicas / 2; o > 0; o >>= 1) {
    t1 += __shfl_xor(t1, o, kReplicas);
    t2 += __shfl_xor(t2, o, kReplicas);
  }
}

__global__ void bn_finalize_kernel(double* __restrict__ sums, int64_t n_rows, int n_feat,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float eps, float momentum, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ mean,
                                   float* __restrict__ invstd, float* __restrict__ scale,
                                   float* __restrict__ shift, int64_t* __restrict__ batches_tracked) {
  // (nn.BatchNorm1d.num_batches_tracked of this layer: the step's counter launch, folded in)
  if (batches_tracked != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *batches_tracked += 1;
  const int cpb = blockDim.x / kReplicas;  // columns per workgroup: 32 lanes each
  const int rl = threadIdx.x % kReplicas;
  for (int c0 = blockIdx.x * cpb; c0 < n_feat; c0 += gridDim.x * cpb) {
    const int c = c0 + threadIdx.x / kReplicas;
    const double n = (double)n_rows;
    double t1, t2;
    take_sums_lanes(sums, n_feat, c, rl, c < n_feat, t1, t2);
    if (c >= n_feat || rl != 0) continue;
    const double m = t1 / n;
    double var = t2 / n - m * m;  // biased
    if (var < 0.0) var = 0.0;
    const float mf = (float)m;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f;
    const float b = beta ? beta[c] : 0.f;
    const float sc = g * is;
    if (mean) mean[c] = mf;
    if (invstd) invstd[c] = is;
    scale[c] = sc;
    shift[c] = b - mf * sc;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mf;
    if (running_var) {
      const double unbiased = n_rows > 1 ? var * n / (n - 1.0) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
}

__global__ void bn_fold_eval_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ rm, const float* __restrict__ rv,
                                    float eps, int n_feat, float* __restrict__ scale,
                                    float* __restrict__ shift) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n_feat; c += gridDim.x * blockDim.x) {
    const float is = 1.f / sqrtf(rv[c] + eps);
    const float sc = (gamma ? gamma[c] : 1.f) * is;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
  }
}

template <int V>
__global__ void __launch_bounds__(kBBlock)
bn_apply_kernel(const float* __restrict__ x, int64_t ldx, int64_t slots, int lpr,
                const float* __restrict__ scale, const float* __restrict__ shift,
                float* __restrict__ y, int64_t ldy) {
  for (int64_t e = (int64_t)blockIdx.x * kBBlock + threadIdx.x; e < slots;
       e += (int64_t)gridDim.x * kBBlock) {
    const int64_t r = e / lpr;
    const int c = (int)(e - r * lpr) * V;
    if constexpr (V == 4) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
      const float4 s = *reinterpret_cast<const float4*>(scale + c);
      const float4 h = *reinterpret_cast<const float4*>(shift + c);
      *reinterpret_cast<float4*>(y + r * ldy + c) =
          make_float4(fmaf(v.x, s.x, h.x), fmaf(v.y, s.y, h.y), fmaf(v.z, s.z, h.z), fmaf(v.w, s.w, h.w));
    } else {
      y[r * ldy + c] = fmaf(x[r * ldx + c], scale[c], shift[c]);
    }
  }
}

// BatchNorm backward, elementwise part.  With dbeta = sum dy and dgamma = sum dy*xhat
// already reduced, dx = gamma*invstd*(dy - dbeta/n - xhat*dgamma/n) is affine in (dy, x):
//   dx = A[c]*dy + B[c]*x + C[c],   A = gamma*invstd,  B = -A*invstd*dgamma/n,
//   C = -A*dbeta/n - B*mean
// The three coefficient vectors are produced once (bn_bwd_params_kernel); the streaming
// kernel keeps them in registers: each thread owns one 16-byte column chunk and walks rows.
// RELU: x is a ReLU output and dx is wanted w.r.t. the ReLU input: dx *= (x > 0).
__global__ void bn_bwd_params_kernel(double* __restrict__ sums, int64_t n_rows, int n_feat,
                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                     const float* __restrict__ invstd, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, float* __restrict__ coef, double* __restrict__ loss_acc,
                                     int loss_rep, float loss_inv_count, float* __restrict__ loss) {
  // (the loss of the step: sum of the accumulator replicas the head kernel added into, left clean -- loss.hip's
  // loss_finalize_kernel, folded into the launch that follows the head kernel anyway)
  if (loss_acc != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < loss_rep; ++i) {
      t += loss_acc[i];
      loss_acc[i] = 0.0;
    }
    *loss = (float)(t * (double)loss_inv_count);
  }
  const double inv_n = 1.0 / (double)n_rows;
  const int cpb = blockDim.x / kReplicas;  // columns per workgroup: 32 lanes each
  const int rl = threadIdx.x % kReplicas;
  for (int c0 = blockIdx.x * cpb; c0 < n_feat; c0 += gridDim.x * cpb) {
    const int c = c0 + threadIdx.x / kReplicas;
    double db, dg;
    take_sums_lanes(sums, n_feat, c, rl, c < n_feat, db, dg);
    if (c >= n_feat || rl != 0) continue;
    if (dbeta) dbeta[c] = (float)db;
    if (dgamma) dgamma[c] = (float)dg;
    const double is = (double)invstd[c];
    const double A = (double)(gamma ? gamma[c] : 1.f) * is;
    const double B = -A * is * dg * inv_n;
    const double C = -A * db * inv_n - B * (double)mean[c];
    coef[c] = (float)A;
    coef[n_feat + c] = (float)B;
    coef[2 * n_feat + c] = (float)C;
  }
}

constexpr int kDxRows = 256;  // smallest share of rows a workgroup takes

template <int V, bool RELU, bool RD>
__global__ void __launch_bounds__(kBBlock)
bn_bwd_dx_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x,
                 int64_t ldx, int64_t n_rows, int64_t rows_per_block, int n_feat, int lpr, int lx,
                 const float* __restrict__ coef, float* __restrict__ dx, int64_t lddx, ReadoutGrad rg, int rev) {
  const int ry = kBBlock / lx;
  const int ty = threadIdx.x / lx;
  const int tx = threadIdx.x - ty * lx;
  if (ty >= ry) return;
  const int64_t r_begin = (int64_t)(rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * rows_per_block;
  const int64_t r_end = (r_begin + rows_per_block < n_rows) ? r_begin + rows_per_block : n_rows;
  for (int cc = tx; cc < lpr; cc += lx) {
    const int c = cc * V;
    float A[V], B[V], C[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      A[q] = coef[c + q];
      B[q] = coef[n_feat + c + q];
      C[q] = coef[2 * n_feat + c + q];
    }
    constexpr int R = 4;  // rows per round, loads issued together
    for (int64_t r0 = r_begin + ty; r0 < r_end; r0 += (int64_t)R * ry) {
      float xv[R][V], gv[R][V];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int64_t r = r0 + (int64_t)u * ry;
        const int64_t rc = r < r_end ? r : r_end - 1;
        if constexpr (RD) readout_dy<V>(rg, rc, c, n_feat, gv[u]);
        if constexpr (V == 4) {
          const float4 a4 = *reinterpret_cast<const float4*>(x + rc * ldx + c);
          xv[u][0] = a4.x; xv[u][1] = a4.y; xv[u][2] = a4.z; xv[u][3] = a4.w;
          if (!RD) {
            const float4 b4 = *reinterpret_cast<const float4*>(dy + rc * lddy + c);
            gv[u][0] = b4.x; gv[u][1] = b4.y; gv[u][2] = b4.z; gv[u][3] = b4.w;
          }
        } else {
          xv[u][0] = x[rc * ldx + c];
          if (!RD) gv[u][0] = dy[rc * lddy + c];
        }
      }
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int64_t r = r0 + (int64_t)u * ry;
        if (r >= r_end) continue;
        float o[V];
#pragma unroll
        for (int q = 0; q < V; ++q) {
          const float v = fmaf(A[q], gv[u][q], fmaf(B[q], xv[u][q], C[q]));
          o[q] = (RELU && !(xv[u][q] > 0.f)) ? 0.f : v;
        }
        if constexpr (V == 4) {
          *reinterpret_cast<float4*>(dx + r * lddx + c) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
          dx[r * lddx + c] = o[0];
        }
      }
    }
  }
}

static int launch_col_sums(int mode, const float* a, int64_t lda, const float* x, int64_t ldx,
                           const float* mean, const float* invstd, int64_t n_rows, int n_feat,
                           double* sums, bool acc_clean, hipStream_t st, const ReadoutGrad* rgp = nullptr,
                           const float* only_if_gamma = nullptr, const float* only_if_beta = nullptr, int x_bf16 = 0) {
  if (!acc_clean &&
      hipMemsetAsync(sums, 0, sizeof(double) * 2 * n_feat * (1 + kReplicas), st) != hipSuccess) {
    set_error("bn: memset failed");
    return GCMI_ERR_LAUNCH;
  }
  if (n_rows == 0) return GCMI_OK;
  ReadoutGrad rg{nullptr, nullptr, 0, nullptr};
  int V;
  if (rgp) {
    rg = *rgp;
    mode = 2;
    V = (vec_width(x, ldx, n_feat) == 4 && aligned16(rg.g2) && rg.ldg2 % 4 == 0 && aligned16(rg.arg)) ? 4 : 1;
  } else if (x_bf16) {
    const bool a_ok = x_bf16 == 2 ? ((reinterpret_cast<uintptr_t>(a) & 7u) == 0 && lda % 4 == 0 && n_feat % 4 == 0)
                                  : vec_width(a, lda, n_feat) == 4;
    if (mode != 1 || !a_ok || (reinterpret_cast<uintptr_t>(x) & 7u) || ldx % 4) {
      set_error("bn col_sums: bf16 rows need mode 1 and 8-byte addressable rows");
      return GCMI_ERR_UNSUPPORTED;
    }
    V = 4;
  } else {
    V = vec_width(a, lda, n_feat);
    if (mode == 1 && vec_width(x, ldx, n_feat) != 4) V = 1;
  }
  const int lpr = n_feat / V;
  const int lx = lpr < kBBlock ? lpr : kBBlock;
  // One resident wave of equal workgroups (256 CUs x 8): a fixed 512 rows per workgroup leaves a short second
  // wave of workgroups behind the first and the launch takes two workgroup lifetimes.
  const int64_t round_rows = 4 * (kBBlock / lx);
  int64_t rpb = (n_rows + kResidentBlocks - 1) / kResidentBlocks;
  if (rpb < kRowsPerBlock) rpb = kRowsPerBlock;
  rpb = (rpb + round_rows - 1) / round_rows * round_rows;
  const int blocks = (int)((n_rows + rpb - 1) / rpb);
  const int rev = next_sweep_direction();
#define LAUNCH_CS(VV, MM)                                                                     \
  hipLaunchKernelGGL((col_sums_kernel<VV, MM>), dim3(blocks), dim3(kBBlock), 0, st, a, lda, x, \
                     ldx, mean, invstd, n_rows, rpb, n_feat, lpr, lx, sums, rg, rev, only_if_gamma, only_if_beta)
  if (x_bf16 == 2) {
    hipLaunchKernelGGL((col_sums_kernel<4, 1, true, true>), dim3(blocks), dim3(kBBlock), 0, st, a, lda, x, ldx, mean, invstd,
                       n_rows, rpb, n_feat, lpr, lx, sums, rg, rev, only_if_gamma, only_if_beta);
  } else if (x_bf16) {
    hipLaunchKernelGGL((col_sums_kernel<4, 1, true>), dim3(blocks), dim3(kBBlock), 0, st, a, lda, x, ldx, mean, invstd,
                       n_rows, rpb, n_feat, lpr, lx, sums, rg, rev, only_if_gamma, only_if_beta);
  } else if (V == 4) {
    if (mode == 0) LAUNCH_CS(4, 0); else if (mode == 1) LAUNCH_CS(4, 1); else LAUNCH_CS(4, 2);
  } else {
    if (mode == 0) LAUNCH_CS(1, 0); else if (mode == 1) LAUNCH_CS(1, 1); else LAUNCH_CS(1, 2);
  }
#undef LAUNCH_CS
  GCMI_CHECK_LAUNCH("bn col_sums");
  return GCMI_OK;
}

int bn_stats_impl(const float* d_x, int64_t ldx, int64_t n_rows, int32_t n_feat,
                  const float* d_gamma, const float* d_beta, float eps, float momentum,
                  float* d_running_mean, float* d_running_var, float* d_mean, float* d_invstd,
                  float* d_scale, float* d_shift, double* d_acc, bool acc_clean, void* stream,
                  int64_t* d_batches_tracked) {
  GCMI_CHECK_ARG(n_feat > 0 && n_rows > 0 && ldx >= n_feat, "bn_stats: bad shape (n_rows=%lld)",
                 (long long)n_rows);
  GCMI_CHECK_ARG(d_x && d_scale && d_shift && d_acc, "bn_stats: NULL buffer");
  hipStream_t st = (hipStream_t)stream;
  TimedScope ts(GCMI_K_BATCHNORM, st);
  int rc = launch_col_sums(0, d_x, ldx, nullptr, 0, nullptr, nullptr, n_rows, n_feat, d_acc, acc_clean, st);
  if (rc) return rc;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((n_feat + 7) / 8), dim3(256), 0, st, d_acc, n_rows,
                     n_feat, d_gamma, d_beta, eps, momentum, d_running_mean, d_running_var, d_mean,
                     d_invstd, d_scale, d_shift, d_batches_tracked);
  GCMI_CHECK_LAUNCH("bn_finalize");
  return GCMI_OK;
}

int bn_finalize_impl(int64_t n_rows, int32_t n_feat, const float* d_gamma, const float* d_beta, float eps,
                     float momentum, float* d_running_mean, float* d_running_var, float* d_mean, float* d_invstd,
                     float* d_scale, float* d_shift, double* d_acc, void* stream, int64_t* d_batches_tracked) {
  GCMI_CHECK_ARG(n_feat > 0 && n_rows > 0 && d_scale && d_shift && d_acc, "bn_finalize: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  TimedScope ts(GCMI_K_BATCHNORM, st);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((n_feat + 7) / 8), dim3(256), 0, st, d_acc, n_rows,
                     n_feat, d_gamma, d_beta, eps, momentum, d_running_mean, d_running_var, d_mean,
                     d_invstd, d_scale, d_shift, d_batches_tracked);
  GCMI_CHECK_LAUNCH("bn_finalize");
  return GCMI_OK;
}

int bn_bwd_impl(const float* d_dy, int64_t lddy, const float* d_x, int64_t ldx, int64_t n_rows,
                int32_t n_feat, const float* d_gamma, const float* d_mean,
                const float* d_invstd, float* d_dgamma, float* d_dbeta, float* d_dx,
                int64_t lddx, int32_t relu_mask, double* d_acc, bool acc_clean, void* stream);

}  // namespace gcmi

using namespace gcmi;

extern "C" {

int gcmi_bn_stats(const float* d_x, int64_t ldx, int64_t n_rows, int32_t n_feat,
                  const float* d_gamma, const float* d_beta, float eps, float momentum,
                  float* d_running_mean, float* d_running_var, float* d_mean, float* d_invstd,
                  float* d_scale, float* d_shift, double* d_acc, void* stream) {
  return bn_stats_impl(d_x, ldx, n_rows, n_feat, d_gamma, d_beta, eps, momentum, d_running_mean,
                       d_running_var, d_mean, d_invstd, d_scale, d_shift, d_acc, false, stream);
}

int gcmi_bn_fold_eval(const float* d_gamma, const float* d_beta, const float* d_running_mean,
                      const float* d_running_var, float eps, int32_t n_feat, float* d_scale,
                      float* d_shift, void* stream) {
  GCMI_CHECK_ARG(n_feat > 0 && d_running_mean && d_running_var && d_scale && d_shift,
                 "bn_fold_eval: bad arguments");
  hipLaunchKernelGGL(bn_fold_eval_kernel, dim3((n_feat + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, d_gamma, d_beta, d_running_mean, d_running_var, eps, n_feat,
                     d_scale, d_shift);
  GCMI_CHECK_LAUNCH("bn_fold_eval");
  return GCMI_OK;
}

int gcmi_bn_apply(const float* d_x, int64_t ldx, int64_t n_rows, int32_t n_feat,
                  const float* d_scale, const float* d_shift, float* d_y, int64_t ldy,
                  void* stream) {
  GCMI_CHECK_ARG(n_feat > 0 && n_rows >= 0 && ldx >= n_feat && ldy >= n_feat, "bn_apply: bad shape");
  if (n_rows == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_x && d_y && d_scale && d_shift, "bn_apply: NULL buffer");
  const int V = (vec_width(d_x, ldx, n_feat) == 4 && vec_width(d_y, ldy, n_feat) == 4 &&
                 aligned16(d_scale) && aligned16(d_shift))
                    ? 4
                    : 1;
  const int lpr = n_feat / V;
  const int64_t slots = n_rows * lpr;
  hipStream_t st = (hipStream_t)stream;
  if (V == 4)
    hipLaunchKernelGGL(bn_apply_kernel<4>, dim3(grid_for(slots, kBBlock)), dim3(kBBlock), 0, st, d_x,
                       ldx, slots, lpr, d_scale, d_shift, d_y, ldy);
  else
    hipLaunchKernelGGL(bn_apply_kernel<1>, dim3(grid_for(slots, kBBlock)), dim3(kBBlock), 0, st, d_x,
                       ldx, slots, lpr, d_scale, d_shift, d_y, ldy);
  GCMI_CHECK_LAUNCH("bn_apply");
  return GCMI_OK;
}

int gcmi_bn_bwd(const float* d_dy, int64_t lddy, const float* d_x, int64_t ldx, int64_t n_rows,
                int32_t n_feat, const float* d_gamma, const float* d_mean,
                const float* d_invstd, float* d_dgamma, float* d_dbeta, float* d_dx,
                int64_t lddx, int32_t relu_mask, double* d_acc, void* stream) {
  return bn_bwd_impl(d_dy, lddy, d_x, ldx, n_rows, n_feat, d_gamma, d_mean, d_invstd, d_dgamma, d_dbeta,
                     d_dx, lddx, relu_mask, d_acc, false, stream);
}

}  // extern "C"

namespace gcmi {

static int bn_bwd_any(const ReadoutGrad* rgp, const float* d_dy, int64_t lddy, const float* d_x, int64_t ldx,
                      int64_t n_rows, int32_t n_feat, const float* d_gamma, const float* d_mean,
                      const float* d_invstd, float* d_dgamma, float* d_dbeta, float* d_dx, int64_t lddx,
                      int32_t relu_mask, double* d_acc, bool acc_clean, void* stream);

int bn_bwd_impl(const float* d_dy, int64_t lddy, const float* d_x, int64_t ldx, int64_t n_rows,
                int32_t n_feat, const float* d_gamma, const float* d_mean,
                const float* d_invstd, float* d_dgamma, float* d_dbeta, float* d_dx,
                int64_t lddx, int32_t relu_mask, double* d_acc, bool acc_clean, void* stream) {
  GCMI_CHECK_ARG(lddy >= n_feat && d_dy, "bn_bwd: bad dy");
  return bn_bwd_any(nullptr, d_dy, lddy, d_x, ldx, n_rows, n_feat, d_gamma, d_mean, d_invstd, d_dgamma, d_dbeta,
                    d_dx, lddx, relu_mask, d_acc, acc_clean, stream);
}

// BatchNorm backward whose incoming gradient is the GraphGather backward, recomputed on the fly
// Column sums of the BatchNorm backward behind a GraphGather, from per-molecule data only.  The gradient of a row is
// dy[r,f] = gs[mol][f] + (arg[mol][f] == r) gm[mol][f], so
//   sum_r dy       = sum_mol n_mol gs + gm                       (gm only where the molecule has an arg-max row)
//   sum_r dy xhat  = sum_mol gs (sum_{r in mol} xhat) + gm xhat[arg]
// with sum_{r in mol} xhat = invstd (sum_{r in mol} x - n_mol mean): the readout's forward leaves the per-molecule
// row sums and the arg-max row's value behind.  B x F work instead of a pass over N x F.
__global__ void __launch_bounds__(256)
readout_bn_sums_kernel(int n_mols, int n_feat, int n_deg, const int32_t* __restrict__ runs,
                       const float* __restrict__ g2, int64_t ldg2, const int32_t* __restrict__ arg,
                       const float* __restrict__ rawsum, const float* __restrict__ x, int64_t ldx,
                       const float* __restrict__ mean, const float* __restrict__ invstd, int chunks,
                       double* __restrict__ sums) {
  // a wave owns one 64-column chunk and walks molecules; the launcher makes the wave count a multiple of `chunks`
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * blockDim.x) >> 6;
  const int groups = n_waves / chunks;  // waves past groups * chunks have no chunk of their own
  if (groups == 0 || wave >= groups * chunks) return;
  const int chunk = wave % chunks;
  const int f = chunk * 64 + lane;
  const bool ok = f < n_feat;
  const int fc = ok ? f : 0;
  const double mu = (double)mean[fc], is = (double)invstd[fc];
  double t1 = 0.0, t2 = 0.0;
  for (int m = wave / chunks; m < n_mols; m += groups) {
    // atoms of the molecule: the first n_deg lanes each take one run
    int len = 0;
    if (lane < n_deg) {
      const int2 r = *reinterpret_cast<const int2*>(runs + ((int64_t)m * n_deg + lane) * 2);
      len = r.y - r.x;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) len += __shfl_xor(len, o, 64);  // n_deg <= 16
    const int n = __shfl(len, 0, 64);
    const double gs = (double)g2[(int64_t)m * ldg2 + fc];
    const int a = arg[(int64_t)m * n_feat + fc];
    const float gmf = g2[(int64_t)m * ldg2 + n_feat + fc];
    const float rs = rawsum[(int64_t)m * 2 * n_feat + fc];
    const float xaf = rawsum[(int64_t)m * 2 * n_feat + n_feat + fc];
    const double gm = a >= 0 ? (double)gmf : 0.0;
    const double xa = a >= 0 ? ((double)xaf - mu) * is : 0.0;
    const double xs = ((double)rs - (double)n * mu) * is;
    t1 += (double)n * gs + gm;
    t2 += gs * xs + gm * xa;
  }
  if (ok) {
    double* rep = sums + (size_t)2 * n_feat * (1 + (wave / chunks) % kReplicas);
    atomicAdd(rep + f, t1);
    atomicAdd(rep + n_feat + f, t2);
  }
}

int bn_bwd_readout_impl(const int32_t* d_membership, const float* d_g2, int64_t ldg2, const int32_t* d_arg,
                        const float* d_x, int64_t ldx, int64_t n_rows, int32_t n_feat, const float* d_gamma,
                        const float* d_mean, const float* d_invstd, float* d_dgamma, float* d_dbeta, float* d_dx,
                        int64_t lddx, int32_t relu_mask, double* d_acc, bool acc_clean, void* stream,
                        const float* d_rawsum, const int32_t* d_mol_runs, int32_t n_mols, int32_t n_deg) {
  GCMI_CHECK_ARG(d_membership && d_g2 && d_arg && ldg2 >= 2 * (int64_t)n_feat, "bn_bwd_readout: bad readout gradient");
  ReadoutGrad rg{d_membership, d_g2, ldg2, d_arg};
  rg.rawsum = d_rawsum;
  rg.runs = d_mol_runs;
  rg.n_mols = n_mols;
  rg.n_deg = n_deg;
  return bn_bwd_any(&rg, nullptr, 0, d_x, ldx, n_rows, n_feat, d_gamma, d_mean, d_invstd, d_dgamma, d_dbeta, d_dx,
                    lddx, relu_mask, d_acc, acc_clean, stream);
}

// dbeta = sum dy, dgamma = sum dy * xhat and the coefficient vectors, from the pooled sums (psums) or, where those
// are ill-conditioned, from the direct sums (sums); both accumulators are left clean
__global__ void bn_bwd_params_pool_kernel(double* __restrict__ psums, double* __restrict__ sums, int64_t n_rows,
                                          int n_feat, const float* __restrict__ gamma, const float* __restrict__ beta,
                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                          float* __restrict__ coef) {
  const bool direct = bn_pool_ill_conditioned(gamma, beta, n_feat);
  const double inv_n = 1.0 / (double)n_rows;
  // 32 lanes per column, one accumulator replica each (kReplicas == 32), combined by shuffles: a serial walk over the
  // replicas of both accumulators is 128 dependent loads per column
  const int c = blockIdx.x * (blockDim.x / kReplicas) + threadIdx.x / kReplicas;
  const int r = threadIdx.x % kReplicas;
  const bool ok = c < n_feat;
  double p1 = 0.0, p2 = 0.0, d1 = 0.0, d2 = 0.0;
  if (ok) {
    double* rp = psums + (size_t)2 * n_feat * (1 + r);
    double* rd = sums + (size_t)2 * n_feat * (1 + r);
    p1 = rp[c]; p2 = rp[n_feat + c]; d1 = rd[c]; d2 = rd[n_feat + c];
    rp[c] = 0.0; rp[n_feat + c] = 0.0; rd[c] = 0.0; rd[n_feat + c] = 0.0;
  }
#pragma unroll
  for (int o = kReplicas / 2; o > 0; o >>= 1) {
    p1 += __shfl_xor(p1, o, kReplicas); p2 += __shfl_xor(p2, o, kReplicas);
    d1 += __shfl_xor(d1, o, kReplicas); d2 += __shfl_xor(d2, o, kReplicas);
  }
  if (ok && r == 0) {
    const double gm = (double)(gamma ? gamma[c] : 1.f), bt = (double)(beta ? beta[c] : 0.f);
    const double db = direct ? d1 : p1;
    const double dg = direct ? d2 : (p2 - bt * p1) / gm;
    if (dbeta) dbeta[c] = (float)db;
    if (dgamma) dgamma[c] = (float)dg;
    const double is = (double)invstd[c];
    const double A = gm * is;
    const double B = -A * is * dg * inv_n;
    const double C = -A * db * inv_n - B * (double)mean[c];
    coef[c] = (float)A;
    coef[n_feat + c] = (float)B;
    coef[2 * n_feat + c] = (float)C;
  }
}

int bn_bwd_pool_impl(const float* d_dy, int64_t lddy, const float* d_x, int64_t ldx, int64_t n_rows, int32_t n_feat,
                     const float* d_gamma, const float* d_beta, const float* d_mean, const float* d_invstd,
                     float* d_dgamma, float* d_dbeta, double* d_psums, double* d_acc, void* stream, int32_t x_bf16) {
  GCMI_CHECK_ARG(n_feat > 0 && n_rows > 0 && d_mean && d_invstd && d_acc && d_psums && d_x && d_gamma && d_beta,
                 "bn_bwd_pool: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  TimedScope ts(GCMI_K_BATCHNORM, st);
  if (d_dy != nullptr) {
    const int rc = launch_col_sums(1, d_dy, lddy, d_x, ldx, d_mean, d_invstd, n_rows, n_feat, d_acc, true, st, nullptr,
                                   d_gamma, d_beta, x_bf16);
    if (rc) return rc;
  }
  static_assert(kReplicas == 32, "bn_bwd_params_pool_kernel: one lane per replica");
  hipLaunchKernelGGL(bn_bwd_params_pool_kernel, dim3((n_feat + 7) / 8), dim3(256), 0, st, d_psums, d_acc, n_rows, n_feat, d_gamma,
                     d_beta, d_mean, d_invstd, d_dgamma, d_dbeta, reinterpret_cast<float*>(d_acc));
  GCMI_CHECK_LAUNCH("bn_bwd_params_pool");
  return GCMI_OK;
}

int bn_bwd_params_impl(int64_t n_rows, int32_t n_feat, const float* d_gamma, const float* d_mean,
                       const float* d_invstd, float* d_dgamma, float* d_dbeta, double* d_acc, void* stream,
                       double* d_loss_acc, int loss_rep, float loss_inv_count, float* d_loss) {
  GCMI_CHECK_ARG(n_feat > 0 && n_rows > 0 && d_mean && d_invstd && d_acc, "bn_bwd_params: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  TimedScope ts(GCMI_K_BATCHNORM, st);
  hipLaunchKernelGGL(bn_bwd_params_kernel, dim3((n_feat + 7) / 8), dim3(256), 0, st, d_acc, n_rows, n_feat,
                     d_gamma, d_mean, d_invstd, d_dgamma, d_dbeta, reinterpret_cast<float*>(d_acc), d_loss_acc, loss_rep,
                     loss_inv_count, d_loss);
  GCMI_CHECK_LAUNCH("bn_bwd_params");
  return GCMI_OK;
}

static int bn_bwd_any(const ReadoutGrad* rgp, const float* d_dy, int64_t lddy, const float* d_x, int64_t ldx,
                      int64_t n_rows, int32_t n_feat, const float* d_gamma, const float* d_mean,
                      const float* d_invstd, float* d_dgamma, float* d_dbeta, float* d_dx, int64_t lddx,
                      int32_t relu_mask, double* d_acc, bool acc_clean, void* stream) {
  GCMI_CHECK_ARG(n_feat > 0 && n_rows > 0 && ldx >= n_feat, "bn_bwd: bad shape");
  GCMI_CHECK_ARG(d_x && d_mean && d_invstd && d_acc, "bn_bwd: NULL buffer");
  GCMI_CHECK_ARG(d_dx == nullptr || lddx >= n_feat, "bn_bwd: bad lddx");
  hipStream_t st = (hipStream_t)stream;
  TimedScope ts(GCMI_K_BATCHNORM, st);
  int rc = GCMI_OK;
  if (rgp && rgp->rawsum && rgp->runs && rgp->n_mols > 0) {
    if (!acc_clean &&
        hipMemsetAsync(d_acc, 0, sizeof(double) * 2 * n_feat * (1 + kReplicas), st) != hipSuccess) {
      set_error("bn: memset failed");
      return GCMI_ERR_LAUNCH;
    }
    const int chunks = (n_feat + 63) / 64;
    int waves = 4096 / chunks * chunks;  // resident waves of 256 CUs x 16, a multiple of the chunk count
    const int64_t jobs = (int64_t)rgp->n_mols * chunks;
    if (jobs < waves) waves = (int)((jobs + chunks - 1) / chunks * chunks);
    const int blocks = (waves + 3) / 4;  // waves beyond `waves` (at most 3) simply walk from a later molecule
    hipLaunchKernelGGL(readout_bn_sums_kernel, dim3(blocks), dim3(256), 0, st, rgp->n_mols, n_feat, rgp->n_deg,
                       rgp->runs, rgp->g2, rgp->ldg2, rgp->arg, rgp->rawsum, d_x, ldx, d_mean, d_invstd, chunks,
                       d_acc);
    GCMI_CHECK_LAUNCH("readout_bn_sums");
  } else {
    rc = launch_col_sums(1, d_dy, lddy, d_x, ldx, d_mean, d_invstd, n_rows, n_feat, d_acc, acc_clean, st, rgp);
  }
  if (rc) return rc;
  // coefficient vectors (3F floats) live in the first 2F doubles of the scratch
  float* coef = reinterpret_cast<float*>(d_acc);
  hipLaunchKernelGGL(bn_bwd_params_kernel, dim3((n_feat + 7) / 8), dim3(256), 0, st, d_acc, n_rows,
                     n_feat, d_gamma, d_mean, d_invstd, d_dgamma, d_dbeta, coef, nullptr, 0, 0.f, nullptr);
  GCMI_CHECK_LAUNCH("bn_bwd_params");
  if (d_dx) {
    ReadoutGrad rg{nullptr, nullptr, 0, nullptr};
    if (rgp) rg = *rgp;
    const bool src_ok = rgp ? (aligned16(rg.g2) && rg.ldg2 % 4 == 0 && aligned16(rg.arg))
                            : vec_width(d_dy, lddy, n_feat) == 4;
    const int V = (vec_width(d_dx, lddx, n_feat) == 4 && src_ok && vec_width(d_x, ldx, n_feat) == 4) ? 4 : 1;
    const int lpr = n_feat / V;
    const int lx = lpr < kBBlock ? lpr : kBBlock;
    const int64_t round_rows = 4 * (kBBlock / lx);
    int64_t rpb = (n_rows + kResidentBlocks - 1) / kResidentBlocks;  // one resident wave of equal workgroups
    if (rpb < kDxRows) rpb = kDxRows;
    rpb = (rpb + round_rows - 1) / round_rows * round_rows;
    const int blocks = (int)((n_rows + rpb - 1) / rpb);
    const int rev = next_sweep_direction();
#define LAUNCH_DX(VV, RR, DD)                                                                     \
  hipLaunchKernelGGL((bn_bwd_dx_kernel<VV, RR, DD>), dim3(blocks), dim3(kBBlock), 0, st, d_dy, lddy, \
                     d_x, ldx, n_rows, rpb, n_feat, lpr, lx, coef, d_dx, lddx, rg, rev)
#define LAUNCH_DX_R(VV, RR) \
  do { if (rgp) LAUNCH_DX(VV, RR, true); else LAUNCH_DX(VV, RR, false); } while (0)
    if (V == 4) {
      if (relu_mask) LAUNCH_DX_R(4, true); else LAUNCH_DX_R(4, false);
    } else {
      if (relu_mask) LAUNCH_DX_R(1, true); else LAUNCH_DX_R(1, false);
    }
#undef LAUNCH_DX_R
#undef LAUNCH_DX
    GCMI_CHECK_LAUNCH("bn_bwd_dx");
  }
  return GCMI_OK;
}

}  // namespace gcmi
