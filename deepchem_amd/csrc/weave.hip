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
//                        (the P x H activation is never written; K = Fp = 14 runs on the VALUs,
//                        pair rows staged in LDS, one float atomic per column and atom)
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

// out[a, h] += relu(Pf[p,:] . W[:,h] + b[h]) for every pair p of source atom a (out pre-zeroed).
// A workgroup takes kP2aPairs consecutive pairs: their feature rows and source atoms are staged in
// LDS with coalesced loads; every wave then walks a quarter of them with lane = output column, the
// pair row read by LDS broadcast, and flushes its running sum with one float atomic per column
// whenever the source atom changes (pairs are sorted by source atom: ~one flush per atom).
constexpr int kP2aPairs = 256;

template <int FP>
__global__ void __launch_bounds__(kWvBlock)
pair_to_atom_kernel(const float* __restrict__ pf, int64_t ldp, int fp, const int32_t* __restrict__ pair_src,
                    int64_t n_pairs, const float* __restrict__ w, const float* __restrict__ b, int H,
                    float* __restrict__ out, int64_t ldo) {
  __shared__ float feat[kP2aPairs][FP];
  __shared__ int src[kP2aPairs];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t p0 = (int64_t)blockIdx.x * kP2aPairs;
  const int np = (int)((n_pairs - p0 < kP2aPairs) ? n_pairs - p0 : kP2aPairs);
  for (int e = tid; e < np * FP; e += kWvBlock) {  // rows are contiguous when ldp == fp
    const int r = e / FP, k = e - r * FP;
    feat[r][k] = k < fp ? pf[(p0 + r) * ldp + k] : 0.f;
  }
  for (int r = tid; r < np; r += kWvBlock) src[r] = pair_src[p0 + r];
  __syncthreads();
  const int q0 = wave * (kP2aPairs / 4);
  const int q1 = (q0 + kP2aPairs / 4 < np) ? q0 + kP2aPairs / 4 : np;
  for (int h0 = 0; h0 < H; h0 += 64) {
    const int h = h0 + lane;
    const bool h_ok = h < H;
    float wc[FP];
#pragma unroll
    for (int k = 0; k < FP; ++k) wc[k] = (h_ok && k < fp) ? w[(int64_t)k * H + h] : 0.f;
    const float bh = (h_ok && b) ? b[h] : 0.f;
    float acc = 0.f;
    int cur = q0 < q1 ? src[q0] : -1;
    for (int q = q0; q < q1; ++q) {
      const int a = src[q];
      if (a != cur) {  // wave-uniform
        if (h_ok) atomicAdd(out + (int64_t)cur * ldo + h, acc);
        acc = 0.f;
        cur = a;
      }
      float v = bh;
#pragma unroll
      for (int k = 0; k < FP; ++k) v = fmaf(feat[q][k], wc[k], v);  // k >= fp: wc = 0, feat unread garbage * 0
      acc += v > 0.f ? v : 0.f;
    }
    if (cur >= 0 && h_ok) atomicAdd(out + (int64_t)cur * ldo + h, acc);
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
  constexpr int R = 4;  // pairs per round: their loads are issued together
  // ---- atom -> pair columns [0, H): four gathered rows per pair
  for (int c0 = 0; c0 < H; c0 += 64) {
    const int c = c0 + lane;
    const bool ok = c < H;
    const int cu = ok ? c : 0;
    const float bias = (ok && b_ap) ? b_ap[c] : 0.f;
    for (int64_t p0 = wave * R; p0 < n_pairs; p0 += n_waves * R) {
      float uij[R], vij[R], uji[R], vji[R];
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const int64_t p = p0 + q < n_pairs ? p0 + q : p0;
        const int i = atom_to_pair[2 * p], j = atom_to_pair[2 * p + 1];
        uij[q] = u[(int64_t)i * lduv + cu]; vij[q] = v[(int64_t)j * lduv + cu];
        uji[q] = u[(int64_t)j * lduv + cu]; vji[q] = v[(int64_t)i * lduv + cu];
      }
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const float ij = uij[q] + vij[q] + bias, ji = uji[q] + vji[q] + bias;
        if (ok && p0 + q < n_pairs) z[(p0 + q) * ldz + c] = (ij > 0.f ? ij : 0.f) + (ji > 0.f ? ji : 0.f);
      }
    }
  }
  // ---- pair -> pair columns [H, H + H2): K = fp product on the VALUs, the pair row by broadcast loads
  for (int c0 = 0; c0 < H2; c0 += 64) {
    const int h2 = c0 + lane;
    const bool ok = h2 < H2;
    float wc[FP];
#pragma unroll
    for (int k = 0; k < FP; ++k) wc[k] = (ok && k < fp) ? w_pp[(int64_t)k * H2 + h2] : 0.f;
    const float bias = (ok && b_pp) ? b_pp[h2] : 0.f;
    for (int64_t p0 = wave * R; p0 < n_pairs; p0 += n_waves * R) {
      float t[R];
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const int64_t p = p0 + q < n_pairs ? p0 + q : p0;
        const float* row = pf + p * ldp;
        t[q] = bias;
#pragma unroll
        for (int k = 0; k < FP; ++k)
          if (k < fp) t[q] = fmaf(row[k], wc[k], t[q]);
      }
#pragma unroll
      for (int q = 0; q < R; ++q)
        if (ok && p0 + q < n_pairs) z[(p0 + q) * ldz + H + h2] = t[q] > 0.f ? t[q] : 0.f;
    }
  }
}

// The same map with a lane per OUTPUT QUAD instead of a lane per column: a workgroup is `slots` pair
// slots x Q = (H + H2) / 4 lanes, every lane owns four fixed columns (their biases and, for pair->pair
// columns, their weight columns live in registers), gathers the four atom rows of its pair with 16-byte
// loads and writes its quad with one 16-byte store -- whole 400-byte output rows per pair slot instead of
// two half rows from 50 of 64 lanes, and a quarter of the load instructions.  Needs 16-byte addressable
// U, V and Z rows and (H + H2) % 4 == 0; anything else takes the per-column kernel above.
template <int FP>
__global__ void __launch_bounds__(kWvBlock)
pair_features_quad_kernel(const float* __restrict__ u, const float* __restrict__ v, int64_t lduv, int H,
                          const float* __restrict__ b_ap, const float* __restrict__ pf, int64_t ldp, int fp,
                          const float* __restrict__ w_pp, const float* __restrict__ b_pp, int H2,
                          const int32_t* __restrict__ atom_to_pair, int64_t n_pairs, float* __restrict__ z,
                          int64_t ldz, int Q, int slots) {
  const int slot = threadIdx.x / Q;
  if (slot >= slots) return;
  const int q = threadIdx.x - slot * Q;
  const int c0 = 4 * q;
  const bool pure_ap = c0 + 3 < H, pure_pp = c0 >= H;
  float bias[4], w[4][FP];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = c0 + e;
    if (c < H) {
      bias[e] = b_ap ? b_ap[c] : 0.f;
#pragma unroll
      for (int k = 0; k < FP; ++k) w[e][k] = 0.f;
    } else {
      const int h2 = c - H;
      bias[e] = b_pp ? b_pp[h2] : 0.f;
#pragma unroll
      for (int k = 0; k < FP; ++k) w[e][k] = k < fp ? w_pp[(int64_t)k * H2 + h2] : 0.f;
    }
  }
  constexpr int R = 2;  // pairs per round and slot: their loads are issued together
  const int64_t stride = (int64_t)gridDim.x * slots;
  for (int64_t p0 = (int64_t)blockIdx.x * slots + slot; p0 < n_pairs; p0 += stride * R) {
    float uij[R][4], vij[R][4], uji[R][4], vji[R][4], pv[R][FP];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t pr = p0 + r * stride;
      const int64_t p = pr < n_pairs ? pr : p0;
      const int i = atom_to_pair[2 * p], j = atom_to_pair[2 * p + 1];
      if (pure_ap) {
        const float4 a = *reinterpret_cast<const float4*>(u + (int64_t)i * lduv + c0);
        const float4 b = *reinterpret_cast<const float4*>(v + (int64_t)j * lduv + c0);
        const float4 c = *reinterpret_cast<const float4*>(u + (int64_t)j * lduv + c0);
        const float4 d = *reinterpret_cast<const float4*>(v + (int64_t)i * lduv + c0);
        uij[r][0] = a.x; uij[r][1] = a.y; uij[r][2] = a.z; uij[r][3] = a.w;
        vij[r][0] = b.x; vij[r][1] = b.y; vij[r][2] = b.z; vij[r][3] = b.w;
        uji[r][0] = c.x; uji[r][1] = c.y; uji[r][2] = c.z; uji[r][3] = c.w;
        vji[r][0] = d.x; vji[r][1] = d.y; vji[r][2] = d.z; vji[r][3] = d.w;
      } else if (!pure_pp) {  // the quad that straddles the two column groups
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = c0 + e < H ? c0 + e : 0;
          uij[r][e] = u[(int64_t)i * lduv + c]; vij[r][e] = v[(int64_t)j * lduv + c];
          uji[r][e] = u[(int64_t)j * lduv + c]; vji[r][e] = v[(int64_t)i * lduv + c];
        }
      }
      if (!pure_ap) {
        const float* row = pf + p * ldp;
#pragma unroll
        for (int k = 0; k < FP; ++k) pv[r][k] = k < fp ? row[k] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t pr = p0 + r * stride;
      if (pr >= n_pairs) continue;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (c0 + e < H) {
          const float ij = uij[r][e] + vij[r][e] + bias[e], ji = uji[r][e] + vji[r][e] + bias[e];
          o[e] = (ij > 0.f ? ij : 0.f) + (ji > 0.f ? ji : 0.f);
        } else {
          float t = bias[e];
#pragma unroll
          for (int k = 0; k < FP; ++k)
            if (k < fp) t = fmaf(pv[r][k], w[e][k], t);
          o[e] = t > 0.f ? t : 0.f;
        }
      }
      *reinterpret_cast<float4*>(z + pr * ldz + c0) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// WeaveGather.gaussian_histogram (layers.py:4600-4648): 11 unit-height Gaussians
// exp(-(x-mu)^2 / (2 sigma^2)), normalised over the bins; output column f*11 + bin.
__constant__ float kGaussMu[11] = {-1.645f, -1.080f, -0.739f, -0.468f, -0.228f, 0.f,
                                   0.228f, 0.468f, 0.739f, 1.080f, 1.645f};
__constant__ float kGaussSigma[11] = {0.283f, 0.170f, 0.134f, 0.118f, 0.114f, 0.114f,
                                      0.114f, 0.118f, 0.134f, 0.170f, 0.283f};

// one wave per (molecule, 64-column chunk), lane = feature column; the 11 bin sums stay in registers.
// Normal(mu, sigma).log_prob(x).exp() / (its value at mu) == exp(-(x-mu)^2 / (2 sigma^2)): the
// normalisation constants of the reference cancel (difference ~1 ulp).
__global__ void __launch_bounds__(kWvBlock)
weave_gather_kernel(const float* __restrict__ x, int64_t ldx, int n_feat, const int32_t* __restrict__ mol_ptr,
                    int n_mols, int expand, float* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * kWvBlock + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * kWvBlock) >> 6;
  const int chunks = (n_feat + 63) / 64;
  float inv2var[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) inv2var[k] = 1.f / (2.f * kGaussSigma[k] * kGaussSigma[k]);
  for (int64_t job = wave; job < (int64_t)n_mols * chunks; job += n_waves) {
    const int m = (int)(job / chunks);
    const int f = (int)(job - (int64_t)m * chunks) * 64 + lane;
    const int a0 = mol_ptr[m], a1 = mol_ptr[m + 1];
    float acc[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) acc[k] = 0.f;
    if (f < n_feat) {
      for (int a = a0; a < a1; a += 2) {
        const bool two = a + 1 < a1;
        const float x0 = x[(int64_t)a * ldx + f];
        const float x1 = x[(int64_t)(two ? a + 1 : a) * ldx + f];
        if (expand) {
          float g0[11], g1[11], t0 = 0.f, t1 = 0.f;
#pragma unroll
          for (int k = 0; k < 11; ++k) {
            const float d0 = x0 - kGaussMu[k], d1 = x1 - kGaussMu[k];
            g0[k] = expf(-(d0 * d0) * inv2var[k]);
            g1[k] = expf(-(d1 * d1) * inv2var[k]);
            t0 += g0[k];
            t1 += g1[k];
          }
#pragma unroll
          for (int k = 0; k < 11; ++k) {
            acc[k] += g0[k] / t0;
            if (two) acc[k] += g1[k] / t1;
          }
        } else {
          acc[0] += x0;
          if (two) acc[0] += x1;
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

int gcmi_weave_pair_to_atom(const float* d_pair_feat, int64_t ldp, int32_t n_pair_feat, const int32_t* d_pair_src,
                            int64_t n_pairs, int32_t n_atoms, const float* d_w, const float* d_b, int32_t n_hidden,
                            float* d_out, int64_t ldo, void* stream) {
  GCMI_CHECK_ARG(n_atoms >= 0 && n_pairs >= 0 && n_pair_feat > 0 && n_pair_feat <= kMaxFp && n_hidden > 0 &&
                     ldp >= n_pair_feat && ldo >= n_hidden,
                 "weave_pair_to_atom: bad shape (pair features <= %d)", kMaxFp);
  if (n_atoms == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_w && d_out && (n_pairs == 0 || (d_pair_src && d_pair_feat)), "weave_pair_to_atom: NULL buffer");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemset2DAsync(d_out, sizeof(float) * (size_t)ldo, 0, sizeof(float) * (size_t)n_hidden, (size_t)n_atoms, st) !=
      hipSuccess) {
    (void)hipGetLastError();
    set_error("weave_pair_to_atom: memset failed");
    return GCMI_ERR_LAUNCH;
  }
  if (n_pairs == 0) return GCMI_OK;
  const int64_t blocks = (n_pairs + kP2aPairs - 1) / kP2aPairs;
  GCMI_CHECK_ARG(blocks < (1LL << 31), "weave_pair_to_atom: too many pairs");
  if (n_pair_feat <= 16)
    hipLaunchKernelGGL(pair_to_atom_kernel<16>, dim3((unsigned)blocks), dim3(kWvBlock), 0, st, d_pair_feat, ldp,
                       n_pair_feat, d_pair_src, n_pairs, d_w, d_b, n_hidden, d_out, ldo);
  else
    hipLaunchKernelGGL(pair_to_atom_kernel<kMaxFp>, dim3((unsigned)blocks), dim3(kWvBlock), 0, st, d_pair_feat, ldp,
                       n_pair_feat, d_pair_src, n_pairs, d_w, d_b, n_hidden, d_out, ldo);
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
  hipStream_t st = (hipStream_t)stream;
  const int HT = n_hidden_ap + n_hidden_pp;
  if (HT % 4 == 0 && HT / 4 <= kWvBlock && n_pair_feat <= 16 && lduv % 4 == 0 && ldz % 4 == 0 && aligned16(d_u) &&
      aligned16(d_v) && aligned16(d_z)) {
    const int Q = HT / 4, slots = kWvBlock / Q;
    const int64_t want = (n_pairs + 2 * slots - 1) / (2 * slots);
    const int grid_q = (int)(want < 2048 ? want : 2048);  // 256 CUs x 8 resident workgroups
    hipLaunchKernelGGL(pair_features_quad_kernel<16>, dim3(grid_q), dim3(kWvBlock), 0, st, d_u, d_v, lduv, n_hidden_ap,
                       d_b_ap, d_pair_feat, ldp, n_pair_feat, d_w_pp, d_b_pp, n_hidden_pp, d_atom_to_pair, n_pairs,
                       d_z, ldz, Q, slots);
    GCMI_CHECK_LAUNCH("weave_pair_features");
    return GCMI_OK;
  }
  const int grid = grid_for(n_pairs * 64, kWvBlock);
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
  hipLaunchKernelGGL(weave_gather_kernel, dim3(grid_for((int64_t)n_mols * ((n_feat + 63) / 64) * 64, kWvBlock)),
                     dim3(kWvBlock), 0,
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
