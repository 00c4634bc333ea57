// Weave ("pair-feature") convolution kernels: the graph parts of WeaveLayer
// (models/torch_models/layers.py:4327-4429) and WeaveGather (:4547-4648).
//
// A Weave batch is: atoms A[N x Fa] (molecule-major, atom_split ascending), ordered pairs
// Pf[P x Fp] listed source atom by source atom (pair_split ascending -> a CSR pair_ptr[N+1]) and
// atom_to_pair[P x 2] = (source, destination).  The dense products with more than a handful of
// rows go through gcmi_seg_gemm; what lives here are the index-driven parts, each fused so that no
// pair-sized intermediate is written that the reference's op-by-op graph materialises:
//
//   weave_pair_to_atom   relu(affine(Pf.W_PA))  summed over the pairs of every source atom
//                        (the P x H activation is never written; K = Fp = 14 runs on the VALUs)
//   weave_pair_features  Z[p] = [ relu(U[i]+V[j]+b) + relu(U[j]+V[i]+b) | relu(affine(Pf[p].W_PP)) ]
//                        with U = A.W_AP[:Fa], V = A.W_AP[Fa:] computed per ATOM: the reference's
//                        two P x 2Fa gathered matmuls become two N x Fa ones plus row gathers
//   weave_gather         Gaussian-histogram expansion (11 bins) + per-molecule sum in one pass
//                        (the N x 11F expansion is never written)
// BatchNorm layers of the Weave path run in eval mode in the reference (layers.py:4361 etc.):
// they are affine maps folded into the weights by gcmi_fold_affine.
// Bound: HBM (one pass over the pair features / atom rows); lanes run along the feature columns.
#include <math.h>

#include "common.h"

namespace gcmi {

constexpr int kWvBlock = 256;
constexpr int kMaxFp = 32;  // pair input features held in registers per lane

// W' = W * diag(scale), b' = b*scale + shift  (W is K x n row-major, or n x K when trans)
__global__ void fold_affine_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                   const float* __restrict__ scale, const float* __restrict__ shift, int K,
                                   int n, int trans, float* __restrict__ w_out, float* __restrict__ b_out) {
  const int total = K * n;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total + n; e += gridDim.x * blockDim.x) {
    if (e < total) {
      const int col = trans ? e / K : e % n;
      w_out[e] = w[e] * (scale ? scale[col] : 1.f);
    } else {
      const int c = e - total;
      const float bv = b ? b[c] : 0.f;
      b_out[c] = bv * (scale ? scale[c] : 1.f) + (shift ? shift[c] : 0.f);
    }
  }
}

// out[a, h] = sum over pairs p of atom a of relu(Pf[p,:] . W[:,h] + b[h]); one wave per atom at a
// time, lane = output column h (H <= 64 per pass), W column in registers.
template <int FP>
__global__ void __launch_bounds__(kWvBlock)
pair_to_atom_kernel(const float* __restrict__ pf, int64_t ldp, int fp, const int32_t* __restrict__ pair_ptr,
                    int n_atoms, const float* __restrict__ w, const float* __restrict__ b, int H,
                    float* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * kWvBlock + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * kWvBlock) >> 6;
  for (int h0 = 0; h0 < H; h0 += 64) {
    const int h = h0 + lane;
    float wc[FP];
#pragma unroll
    for (int k = 0; k < FP; ++k) wc[k] = (h < H && k < fp) ? w[(int64_t)k * H + h] : 0.f;
    const float bh = (h < H && b) ? b[h] : 0.f;
    for (int a = wave; a < n_atoms; a += n_waves) {
      const int p0 = pair_ptr[a], p1 = pair_ptr[a + 1];
      float acc = 0.f;
      for (int p = p0; p < p1; ++p) {
        const float* row = pf + (int64_t)p * ldp;  // same address in every lane: one broadcast load
        float v = bh;
#pragma unroll
        for (int k = 0; k < FP; ++k)
          if (k < fp) v = fmaf(row[k], wc[k], v);
        acc += v > 0.f ? v : 0.f;
      }
      if (h < H) out[(int64_t)a * ldo + h] = acc;
    }
  }
}

// Z[p, 0:H]   = relu(U[i]+V[j]+b_ap) + relu(U[j]+V[i]+b_ap)
// Z[p, H:H+H2] = relu(Pf[p,:].W_pp + b_pp)
template <int FP>
__global__ void __launch_bounds__(kWvBlock)
pair_features_kernel(const float* __restrict__ u, const float* __restrict__ v, int64_t lduv, int H,
                     const float* __restrict__ b_ap, const float* __restrict__ pf, int64_t ldp, int fp,
                     const float* __restrict__ w_pp, const float* __restrict__ b_pp, int H2,
                     const int32_t* __restrict__ atom_to_pair, int64_t n_pairs, float* __restrict__ z,
                     int64_t ldz) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * kWvBlock + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * kWvBlock) >> 6;
  const int W = H + H2;
  for (int c0 = 0; c0 < W; c0 += 64) {
    const int c = c0 + lane;
    const bool is_ap = c < H;
    const int h2 = c - H;
    float wc[FP];
#pragma unroll
    for (int k = 0; k < FP; ++k) wc[k] = (!is_ap && c < W && k < fp) ? w_pp[(int64_t)k * H2 + h2] : 0.f;
    const float bias = c >= W ? 0.f : (is_ap ? (b_ap ? b_ap[c] : 0.f) : (b_pp ? b_pp[h2] : 0.f));
    for (int64_t p = wave; p < n_pairs; p += n_waves) {
      float o;
      if (is_ap) {
        const int i = atom_to_pair[2 * p], j = atom_to_pair[2 * p + 1];
        const float ij = u[(int64_t)i * lduv + c] + v[(int64_t)j * lduv + c] + bias;
        const float ji = u[(int64_t)j * lduv + c] + v[(int64_t)i * lduv + c] + bias;
        o = (ij > 0.f ? ij : 0.f) + (ji > 0.f ? ji : 0.f);
      } else {
        const float* row = pf + p * ldp;
        float t = bias;
#pragma unroll
        for (int k = 0; k < FP; ++k)
          if (k < fp) t = fmaf(row[k], wc[k], t);
        o = t > 0.f ? t : 0.f;
      }
      if (c < W) z[p * ldz + c] = o;
    }
  }
}

// WeaveGather.gaussian_histogram (layers.py:4600-4648): 11 unit-height Gaussians
// exp(-(x-mu)^2 / (2 sigma^2)), normalised over the bins; output column f*11 + bin.
__constant__ float kGaussMu[11] = {-1.645f, -1.080f, -0.739f, -0.468f, -0.228f, 0.f,
                                   0.228f, 0.468f, 0.739f, 1.080f, 1.645f};
__constant__ float kGaussSigma[11] = {0.283f, 0.170f, 0.134f, 0.118f, 0.114f, 0.114f,
                                      0.114f, 0.118f, 0.134f, 0.170f, 0.283f};

// one wave per molecule at a time, lane = feature column; the 11 bin sums stay in registers
__global__ void __launch_bounds__(kWvBlock)
weave_gather_kernel(const float* __restrict__ x, int64_t ldx, int n_feat, const int32_t* __restrict__ mol_ptr,
                    int n_mols, int expand, float* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * kWvBlock + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * kWvBlock) >> 6;
  for (int m = wave; m < n_mols; m += n_waves) {
    const int a0 = mol_ptr[m], a1 = mol_ptr[m + 1];
    for (int f0 = 0; f0 < n_feat; f0 += 64) {
      const int f = f0 + lane;
      float acc[11];
#pragma unroll
      for (int k = 0; k < 11; ++k) acc[k] = 0.f;
      if (f < n_feat) {
        for (int a = a0; a < a1; ++a) {
          const float xv = x[(int64_t)a * ldx + f];
          if (expand) {
            float g[11], tot = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
              // Normal(mu, sigma).log_prob(x).exp() / (its value at mu), as the reference computes it
              const float var = kGaussSigma[k] * kGaussSigma[k];
              const float logs = logf(kGaussSigma[k]);
              const float d = xv - kGaussMu[k];
              const float lp = -(d * d) / (2.f * var) - logs - 0.91893853320467274178f;
              const float lp0 = -logs - 0.91893853320467274178f;
              g[k] = expf(lp) / expf(lp0);
              tot += g[k];
            }
#pragma unroll
            for (int k = 0; k < 11; ++k) acc[k] += g[k] / tot;
          } else {
            acc[0] += xv;
          }
        }
        if (expand) {
#pragma unroll
          for (int k = 0; k < 11; ++k) out[(int64_t)m * ldo + (int64_t)f * 11 + k] = acc[k];
        } else {
          out[(int64_t)m * ldo + f] = acc[0];
        }
      }
    }
  }
}

// y = tanh(x) elementwise over a row-major matrix (the final weave convolution's activation)
__global__ void tanh_kernel(float* __restrict__ x, int64_t ldx, int64_t n_rows, int n_feat) {
  const int64_t total = n_rows * n_feat;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / n_feat;
    const int c = (int)(e - r * n_feat);
    x[r * ldx + c] = tanhf(x[r * ldx + c]);
  }
}

}  // namespace gcmi

using namespace gcmi;

extern "C" {

int gcmi_fold_affine(const float* d_w, const float* d_b, const float* d_scale, const float* d_shift, int32_t k,
                     int32_t n, int32_t trans_w, float* d_w_out, float* d_b_out, void* stream) {
  GCMI_CHECK_ARG(k > 0 && n > 0 && d_w && d_w_out && d_b_out, "fold_affine: bad arguments");
  hipLaunchKernelGGL(fold_affine_kernel, dim3(grid_for((int64_t)k * n + n, 256)), dim3(256), 0, (hipStream_t)stream,
                     d_w, d_b, d_scale, d_shift, k, n, trans_w, d_w_out, d_b_out);
  GCMI_CHECK_LAUNCH("fold_affine");
  return GCMI_OK;
}

int gcmi_weave_pair_to_atom(const float* d_pair_feat, int64_t ldp, int32_t n_pair_feat, const int32_t* d_pair_ptr,
                            int32_t n_atoms, const float* d_w, const float* d_b, int32_t n_hidden, float* d_out,
                            int64_t ldo, void* stream) {
  GCMI_CHECK_ARG(n_atoms >= 0 && n_pair_feat > 0 && n_pair_feat <= kMaxFp && n_hidden > 0 && ldp >= n_pair_feat &&
                     ldo >= n_hidden,
                 "weave_pair_to_atom: bad shape (pair features <= %d)", kMaxFp);
  if (n_atoms == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_pair_ptr && d_w && d_out, "weave_pair_to_atom: NULL buffer");
  const int grid = grid_for((int64_t)n_atoms * 64, kWvBlock);
  hipStream_t st = (hipStream_t)stream;
  if (n_pair_feat <= 16)
    hipLaunchKernelGGL(pair_to_atom_kernel<16>, dim3(grid), dim3(kWvBlock), 0, st, d_pair_feat, ldp, n_pair_feat,
                       d_pair_ptr, n_atoms, d_w, d_b, n_hidden, d_out, ldo);
  else
    hipLaunchKernelGGL(pair_to_atom_kernel<kMaxFp>, dim3(grid), dim3(kWvBlock), 0, st, d_pair_feat, ldp,
                       n_pair_feat, d_pair_ptr, n_atoms, d_w, d_b, n_hidden, d_out, ldo);
  GCMI_CHECK_LAUNCH("weave_pair_to_atom");
  return GCMI_OK;
}

int gcmi_weave_pair_features(const float* d_u, const float* d_v, int64_t lduv, int32_t n_hidden_ap,
                             const float* d_b_ap, const float* d_pair_feat, int64_t ldp, int32_t n_pair_feat,
                             const float* d_w_pp, const float* d_b_pp, int32_t n_hidden_pp,
                             const int32_t* d_atom_to_pair, int64_t n_pairs, float* d_z, int64_t ldz,
                             void* stream) {
  GCMI_CHECK_ARG(n_pairs >= 0 && n_hidden_ap > 0 && n_hidden_pp >= 0 && n_pair_feat > 0 && n_pair_feat <= kMaxFp &&
                     lduv >= n_hidden_ap && ldp >= n_pair_feat && ldz >= n_hidden_ap + n_hidden_pp,
                 "weave_pair_features: bad shape");
  if (n_pairs == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_u && d_v && d_pair_feat && d_atom_to_pair && d_z && (n_hidden_pp == 0 || d_w_pp),
                 "weave_pair_features: NULL buffer");
  const int grid = grid_for(n_pairs * 64, kWvBlock);
  hipStream_t st = (hipStream_t)stream;
  if (n_pair_feat <= 16)
    hipLaunchKernelGGL(pair_features_kernel<16>, dim3(grid), dim3(kWvBlock), 0, st, d_u, d_v, lduv, n_hidden_ap,
                       d_b_ap, d_pair_feat, ldp, n_pair_feat, d_w_pp, d_b_pp, n_hidden_pp, d_atom_to_pair, n_pairs,
                       d_z, ldz);
  else
    hipLaunchKernelGGL(pair_features_kernel<kMaxFp>, dim3(grid), dim3(kWvBlock), 0, st, d_u, d_v, lduv, n_hidden_ap,
                       d_b_ap, d_pair_feat, ldp, n_pair_feat, d_w_pp, d_b_pp, n_hidden_pp, d_atom_to_pair, n_pairs,
                       d_z, ldz);
  GCMI_CHECK_LAUNCH("weave_pair_features");
  return GCMI_OK;
}

int gcmi_weave_gather(const float* d_x, int64_t ldx, int32_t n_feat, const int32_t* d_mol_ptr, int32_t n_mols,
                      int32_t gaussian_expand, float* d_out, int64_t ldo, void* stream) {
  GCMI_CHECK_ARG(n_mols >= 0 && n_feat > 0 && ldx >= n_feat && ldo >= (gaussian_expand ? 11 : 1) * (int64_t)n_feat,
                 "weave_gather: bad shape");
  if (n_mols == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_x && d_mol_ptr && d_out, "weave_gather: NULL buffer");
  hipLaunchKernelGGL(weave_gather_kernel, dim3(grid_for((int64_t)n_mols * 64, kWvBlock)), dim3(kWvBlock), 0,
                     (hipStream_t)stream, d_x, ldx, n_feat, d_mol_ptr, n_mols, gaussian_expand, d_out, ldo);
  GCMI_CHECK_LAUNCH("weave_gather");
  return GCMI_OK;
}

int gcmi_tanh_(float* d_x, int64_t ldx, int64_t n_rows, int32_t n_feat, void* stream) {
  GCMI_CHECK_ARG(n_rows >= 0 && n_feat > 0 && ldx >= n_feat, "tanh: bad shape");
  if (n_rows == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_x, "tanh: NULL buffer");
  hipLaunchKernelGGL(tanh_kernel, dim3(grid_for(n_rows * n_feat, 256)), dim3(256), 0, (hipStream_t)stream, d_x, ldx,
                     n_rows, n_feat);
  GCMI_CHECK_LAUNCH("tanh");
  return GCMI_OK;
}

}  // extern "C"
