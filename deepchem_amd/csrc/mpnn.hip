// Message-passing sub-layers: the index-driven / elementwise parts of EdgeNetwork
// (models/torch_models/layers.py:4006-4088), GatedRecurrentUnit (:2884-2913) and SetGather
// (set2set, :2976-3138); the dense products go through gcmi_seg_gemm.
//
// EdgeNetwork maps every pair's feature vector to a d x d matrix A_p = reshape(Pf[p].W + b) and
// multiplies it with the hidden state of the pair's second atom: a P x d x d tensor (40 KB per
// pair at d = 100) in the reference.  Re-associated here:
//     (A_p . h_j)[r] = sum_k Pf[p,k] * G[j][k*d + r] + G[j][K*d + r],
//     G[j][k*d + r]  = sum_c W[k][r*d + c] * h_j[c]      (k < K),   G[j][K*d + r] = sum_c b[r*d+c] h_j[c]
// G is ONE GEMM per message-passing step over the ATOMS (N x d by d x (K+1)d, weights read in place:
// W viewed as a (K*d) x d matrix is exactly the transposed-layout operand), and the per-pair work
// drops from K*d*d + d*d to (K+1)*d multiply-adds with no pair-sized intermediate.
// Bound: L2/HBM gathers of G rows; lanes run along r.
#include <math.h>

#include "common.h"

namespace gcmi {

constexpr int kMpBlock = 256;
constexpr int kMaxPairFeat = 32;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// out[i, r] = sum over pairs p of destination i of (sum_k pf[p,k] G[src[p]][k*d + r] + G[src[p]][K*d + r])
__global__ void __launch_bounds__(kMpBlock)
edge_network_kernel(const float* __restrict__ g, int64_t ldg, int d, int K, const float* __restrict__ pf, int64_t ldp,
                    const int32_t* __restrict__ dst_ptr, const int32_t* __restrict__ src, int n_dst,
                    float* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * kMpBlock + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * kMpBlock) >> 6;
  for (int i = wave; i < n_dst; i += n_waves) {
    const int p0 = dst_ptr[i], p1 = dst_ptr[i + 1];
    for (int r0 = 0; r0 < d; r0 += 64) {
      const int r = r0 + lane;
      float acc = 0.f;
      if (r < d) {
        // two pairs per round, their K+1 gathered row pieces requested together
        for (int p = p0; p < p1; p += 2) {
          const bool two = p + 1 < p1;
          const float* g0 = g + (int64_t)src[p] * ldg;
          const float* g1 = g + (int64_t)src[two ? p + 1 : p] * ldg;
          const float* f0 = pf + (int64_t)p * ldp;  // broadcast loads
          const float* f1 = pf + (int64_t)(two ? p + 1 : p) * ldp;
          float m0 = g0[(int64_t)K * d + r], m1 = g1[(int64_t)K * d + r];
#pragma unroll 7
          for (int k = 0; k < K; ++k) {
            m0 = fmaf(f0[k], g0[(int64_t)k * d + r], m0);
            m1 = fmaf(f1[k], g1[(int64_t)k * d + r], m1);
          }
          acc += m0;
          if (two) acc += m1;
        }
        out[(int64_t)i * ldo + r] = acc;
      }
    }
  }
}

// The same message with the products taken in the other order:
//   m[i] = sum_p A(pf_p) h[src_p],  A(pf) = sum_k pf_k W_k + B
//        = sum_k W_k (sum_p pf[p,k] h[src_p]) + B (sum_p h[src_p])
// so the pair-level work is T[i, k*d + c] = sum_p pf[p,k] h[src_p, c] (k < K) and T[i, K*d + c] = sum_p h[src_p, c]:
// one 4*d-byte row of h per pair instead of the (K+1)*4*d-byte row of G, (K+1)*d multiply-adds per pair as
// before, and the weights meet the data in ONE atom-level product T . [W_0 | ... | W_{K-1} | B]^T afterwards.
// One wave per destination atom, a lane owns columns lane and lane + 64, the (K+1) x 2 partial sums stay in
// registers; the pair's feature row is loaded by the first K lanes and broadcast with readlane.
template <int KMAX, int NC>
__global__ void __launch_bounds__(kMpBlock)
edge_moments_kernel(const float* __restrict__ h, int64_t ldh, int d, int K, const float* __restrict__ pf, int64_t ldp,
                    const int32_t* __restrict__ dst_ptr, const int32_t* __restrict__ src, int n_dst,
                    float* __restrict__ t, int64_t ldt) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * kMpBlock + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * kMpBlock) >> 6;
  for (int i = wave; i < n_dst; i += n_waves) {
    const int p0 = dst_ptr[i], p1 = dst_ptr[i + 1];
    float acc[KMAX + 1][NC];
#pragma unroll
    for (int k = 0; k <= KMAX; ++k)
#pragma unroll
      for (int q = 0; q < NC; ++q) acc[k][q] = 0.f;
    for (int p = p0; p < p1; p += 2) {  // two pairs per round: their rows are requested together
      const bool two = p + 1 < p1;
      const int pb = two ? p + 1 : p;
      const float* ha = h + (int64_t)src[p] * ldh;
      const float* hb = h + (int64_t)src[pb] * ldh;
      float va[NC], vb[NC];
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        const int c = lane + 64 * q;
        va[q] = c < d ? ha[c] : 0.f;
        vb[q] = (two && c < d) ? hb[c] : 0.f;
      }
      const float fa = lane < K ? pf[(int64_t)p * ldp + lane] : 0.f;
      const float fb = (two && lane < K) ? pf[(int64_t)pb * ldp + lane] : 0.f;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
          const float wa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fa), k));
          const float wb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fb), k));
#pragma unroll
          for (int q = 0; q < NC; ++q) acc[k][q] = fmaf(wa, va[q], fmaf(wb, vb[q], acc[k][q]));
        }
      }
#pragma unroll
      for (int q = 0; q < NC; ++q) acc[KMAX][q] += va[q] + vb[q];
    }
    float* row = t + (int64_t)i * ldt;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < K) {
#pragma unroll
        for (int q = 0; q < NC; ++q) {
          const int c = lane + 64 * q;
          if (c < d) row[(int64_t)k * d + c] = acc[k][q];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      const int c = lane + 64 * q;
      if (c < d) row[(int64_t)K * d + c] = acc[KMAX][q];
    }
  }
}

// The same moments with the molecule as the unit of work.  In the MPNN batch every pair joins two atoms of ONE
// molecule (all n x n ordered pairs, graph_models.py:1197-1247), so the n state rows of a molecule are gathered n times
// over: a workgroup takes one molecule, stages its state rows in LDS once (16-byte loads), and its waves -- one
// destination atom at a time -- read the states from there; the pair-feature rows of a destination atom are
// contiguous and come 64 / KP pairs per coalesced load, the source indices with them.  A pair whose source lies
// outside the molecule (the kernel does not assume there is none) and molecules too large for the LDS budget read the
// state row from global memory as edge_moments_kernel does.  (edge_moments_kernel: a dependent src -> row gather per
// pair round, 540 MB of L2 reads per launch at 4 096 molecules; 225 us.)
constexpr int kMolLdsFloats = 12 * 1024;  // 48 KB of state rows per workgroup

template <int KP, int NC>
__global__ void __launch_bounds__(kMpBlock)
edge_moments_mol_kernel(const float* __restrict__ h, int64_t ldh, int d, int K, const float* __restrict__ pf, int64_t ldp,
                        const int32_t* __restrict__ dst_ptr, const int32_t* __restrict__ src,
                        const int32_t* __restrict__ mol_ptr, int n_mols, float* __restrict__ t, int64_t ldt,
                        int lds_floats) {
  // (sized by the host for the largest molecule when it knows it: a fixed 48 KB leaves three workgroups per CU)
  extern __shared__ __attribute__((aligned(16))) float hs[];
  constexpr int PPR = 64 / KP;  // pairs per round
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int dq = d / 4;
  for (int m = blockIdx.x; m < n_mols; m += gridDim.x) {
    const int a0 = mol_ptr[m], a1 = mol_ptr[m + 1];
    const int n = a1 - a0;
    const bool staged = n * d <= lds_floats && (d & 3) == 0 && (ldh & 3) == 0;
    __syncthreads();  // (the previous molecule's rows are no longer read)
    if (staged) {
      for (int e = threadIdx.x; e < n * dq; e += kMpBlock) {
        const int r = e / dq, q = e - r * dq;
        *reinterpret_cast<float4*>(hs + r * d + 4 * q) = *reinterpret_cast<const float4*>(h + (int64_t)(a0 + r) * ldh + 4 * q);
      }
    }
    __syncthreads();
    for (int i = a0 + wave; i < a1; i += kMpBlock / 64) {
      const int p0 = dst_ptr[i], p1 = dst_ptr[i + 1];
      float acc[KP + 1][NC];
#pragma unroll
      for (int k = 0; k <= KP; ++k)
#pragma unroll
        for (int q = 0; q < NC; ++q) acc[k][q] = 0.f;
      for (int p = p0; p < p1; p += PPR) {
        // lane l: feature l % KP of pair p + l / KP; lanes 0 .. PPR - 1 also the pairs' source atoms
        const int pl = p + lane / KP, kl = lane % KP;
        const float f = (pl < p1 && kl < K) ? pf[(int64_t)pl * ldp + kl] : 0.f;
        const int sl = (lane < PPR && p + lane < p1) ? src[p + lane] : -1;
        const int np = p1 - p < PPR ? p1 - p : PPR;
        for (int u = 0; u < np; ++u) {
          const int sj = __builtin_amdgcn_readlane(sl, u);
          const bool in_lds = staged && sj >= a0 && sj < a1;
          float v[NC];
#pragma unroll
          for (int q = 0; q < NC; ++q) {
            const int c = lane + 64 * q;
            v[q] = c < d ? (in_lds ? hs[(sj - a0) * d + c] : h[(int64_t)sj * ldh + c]) : 0.f;
          }
#pragma unroll
          for (int k = 0; k < KP; ++k) {
            if (k < K) {
              const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, f), u * KP + k));
#pragma unroll
              for (int q = 0; q < NC; ++q) acc[k][q] = fmaf(w, v[q], acc[k][q]);
            }
          }
#pragma unroll
          for (int q = 0; q < NC; ++q) acc[KP][q] += v[q];
        }
      }
      float* row = t + (int64_t)i * ldt;
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        if (k < K) {
#pragma unroll
          for (int q = 0; q < NC; ++q) {
            const int c = lane + 64 * q;
            if (c < d) row[(int64_t)k * d + c] = acc[k][q];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        const int c = lane + 64 * q;
        if (c < d) row[(int64_t)K * d + c] = acc[KP][q];
      }
    }
  }
}

// z <- sigmoid(z), r <- sigmoid(r), hr = h * r
__global__ void gru_gates_kernel(float* __restrict__ z, float* __restrict__ r, const float* __restrict__ h,
                                 float* __restrict__ hr, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float zz = sigmoidf_(z[e]), rr = sigmoidf_(r[e]);
    z[e] = zz;
    r[e] = rr;
    hr[e] = h[e] * rr;
  }
}

// out = (1 - z) * tanh(hpre) + z * x
__global__ void gru_out_kernel(const float* __restrict__ z, const float* __restrict__ hpre,
                               const float* __restrict__ x, float* __restrict__ out, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    out[e] = (1.f - z[e]) * tanhf(hpre[e]) + z[e] * x[e];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// set2set attention of one step: per molecule m, e_a = <x_a, h_m>, a = softmax over its atoms,
// q_star[m] = [h_m | sum_a a x_a].  One wave per molecule at a time, lanes along the features.
__global__ void __launch_bounds__(kMpBlock)
set2set_attend_kernel(const float* __restrict__ x, int64_t ldx, int n_feat, const int32_t* __restrict__ mol_ptr,
                      int n_mols, const float* __restrict__ h, int64_t ldh, float* __restrict__ qstar,
                      int64_t ldq) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * kMpBlock + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * kMpBlock) >> 6;
  constexpr int kPer = 8;  // features per lane: n_feat <= 512
  for (int m = wave; m < n_mols; m += n_waves) {
    const int a0 = mol_ptr[m], a1 = mol_ptr[m + 1];
    float hv[kPer], racc[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const int f = lane + 64 * q;
      hv[q] = f < n_feat ? h[(int64_t)m * ldh + f] : 0.f;
      racc[q] = 0.f;
    }
    float emax = -INFINITY;
    for (int a = a0; a < a1; ++a) {
      float part = 0.f;
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int f = lane + 64 * q;
        if (f < n_feat) part = fmaf(x[(int64_t)a * ldx + f], hv[q], part);
      }
      emax = fmaxf(emax, wave_sum(part));
    }
    float denom = 0.f;
    for (int a = a0; a < a1; ++a) {
      float xv[kPer], part = 0.f;
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int f = lane + 64 * q;
        xv[q] = f < n_feat ? x[(int64_t)a * ldx + f] : 0.f;
        part = fmaf(xv[q], hv[q], part);
      }
      const float w = expf(wave_sum(part) - emax);
      denom += w;
#pragma unroll
      for (int q = 0; q < kPer; ++q) racc[q] = fmaf(w, xv[q], racc[q]);
    }
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const int f = lane + 64 * q;
      if (f < n_feat) {
        qstar[(int64_t)m * ldq + f] = hv[q];
        qstar[(int64_t)m * ldq + n_feat + f] = a1 > a0 ? racc[q] / denom : 0.f;
      }
    }
  }
}

// z = [i | f | o | g] pre-activations (B x 4H); c <- sig(f) c + sig(i) tanh(g); h = sig(o) tanh(c)
__global__ void lstm_cell_kernel(const float* __restrict__ z, int64_t ldz, int n_hidden, int64_t n_rows,
                                 float* __restrict__ c, int64_t ldc, float* __restrict__ h, int64_t ldh) {
  const int64_t total = n_rows * n_hidden;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / n_hidden;
    const int j = (int)(e - r * n_hidden);
    const float* zr = z + r * ldz;
    const float i = sigmoidf_(zr[j]), f = sigmoidf_(zr[n_hidden + j]), o = sigmoidf_(zr[2 * n_hidden + j]);
    const float cn = f * c[r * ldc + j] + i * tanhf(zr[3 * n_hidden + j]);
    c[r * ldc + j] = cn;
    h[r * ldh + j] = o * tanhf(cn);
  }
}

// ---------------------------------------------------------------- backward of the elementwise pieces
// GRU gates: z = sig(zp), r = sig(rp), hr = h r.  Given dz (w.r.t. the z OUTPUT, from gru_out) and dhr:
//   dzp = dz z (1 - z);  drp = dhr h r (1 - r);  dh = dhr r
__global__ void gru_gates_bwd_kernel(const float* __restrict__ z, const float* __restrict__ r,
                                     const float* __restrict__ h, const float* __restrict__ dz,
                                     const float* __restrict__ dhr, float* __restrict__ dzp, float* __restrict__ drp,
                                     float* __restrict__ dh, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float zz = z[e], rr = r[e], g = dhr[e];
    dzp[e] = dz[e] * zz * (1.f - zz);
    drp[e] = g * h[e] * rr * (1.f - rr);
    dh[e] = g * rr;
  }
}

// out = (1 - z) tanh(hpre) + z x:  dz = dout (x - t);  dhpre = dout (1 - z)(1 - t^2);  dx = dout z
__global__ void gru_out_bwd_kernel(const float* __restrict__ z, const float* __restrict__ hpre,
                                   const float* __restrict__ x, const float* __restrict__ dout,
                                   float* __restrict__ dz, float* __restrict__ dhpre, float* __restrict__ dx,
                                   int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float t = tanhf(hpre[e]), g = dout[e], zz = z[e];
    dz[e] = g * (x[e] - t);
    dhpre[e] = g * (1.f - zz) * (1.f - t * t);
    dx[e] = g * zz;
  }
}

// LSTM cell backward.  Forward: i, f, o = sig(z[0:3H]), gg = tanh(z[3H:4H]); c' = f c + i gg; h' = o tanh(c').
// Given dh' and dc' (gradient arriving at c' from the next step): dz (B x 4H) and dc (w.r.t. the incoming c).
__global__ void lstm_cell_bwd_kernel(const float* __restrict__ z, int64_t ldz, int n_hidden, int64_t n_rows,
                                     const float* __restrict__ c_prev, const float* __restrict__ dh,
                                     const float* __restrict__ dc_next, float* __restrict__ dz,
                                     float* __restrict__ dc_prev) {
  const int64_t total = n_rows * n_hidden;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / n_hidden;
    const int j = (int)(e - r * n_hidden);
    const float* zr = z + r * ldz;
    const float i = sigmoidf_(zr[j]), f = sigmoidf_(zr[n_hidden + j]), o = sigmoidf_(zr[2 * n_hidden + j]);
    const float gg = tanhf(zr[3 * n_hidden + j]);
    const float cp = c_prev[e];
    const float tc = tanhf(f * cp + i * gg);
    const float gh = dh[e];
    const float dc = (dc_next ? dc_next[e] : 0.f) + gh * o * (1.f - tc * tc);
    float* dzr = dz + r * 4 * n_hidden;
    dzr[j] = dc * gg * i * (1.f - i);
    dzr[n_hidden + j] = dc * cp * f * (1.f - f);
    dzr[2 * n_hidden + j] = gh * tc * o * (1.f - o);
    dzr[3 * n_hidden + j] = dc * i * (1.f - gg * gg);
    dc_prev[e] = dc * f;
  }
}

// set2set attention backward for one step.  Forward per molecule: e_a = <x_a, h>, w = softmax(e), r = sum_a w_a x_a,
// q = [h | r].  Given dq = [dqh | dr]:  dw_a = <dr, x_a>;  de_a = w_a (dw_a - sum_b w_b dw_b);
//   dx_a (+)= w_a dr + de_a h;   dh = dqh + sum_a de_a x_a.   One wave per molecule; the softmax is recomputed.
__global__ void __launch_bounds__(kMpBlock)
set2set_attend_bwd_kernel(const float* __restrict__ x, int64_t ldx, int n_feat, const int32_t* __restrict__ mol_ptr,
                          int n_mols, const float* __restrict__ h, int64_t ldh, const float* __restrict__ dq,
                          int64_t lddq, float* __restrict__ dx, int64_t lddx, int accumulate, float* __restrict__ dh,
                          int64_t lddh) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * kMpBlock + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * kMpBlock) >> 6;
  constexpr int kPer = 8;
  for (int m = wave; m < n_mols; m += n_waves) {
    const int a0 = mol_ptr[m], a1 = mol_ptr[m + 1];
    float hv[kPer], drv[kPer], dhacc[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const int f = lane + 64 * q;
      hv[q] = f < n_feat ? h[(int64_t)m * ldh + f] : 0.f;
      drv[q] = f < n_feat ? dq[(int64_t)m * lddq + n_feat + f] : 0.f;
      dhacc[q] = f < n_feat ? dq[(int64_t)m * lddq + f] : 0.f;
    }
    float emax = -INFINITY;
    for (int a = a0; a < a1; ++a) {
      float part = 0.f;
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int f = lane + 64 * q;
        if (f < n_feat) part = fmaf(x[(int64_t)a * ldx + f], hv[q], part);
      }
      emax = fmaxf(emax, wave_sum(part));
    }
    float denom = 0.f, wdw = 0.f;  // sum_b exp(e_b - max), sum_b exp(e_b - max) dw_b
    for (int a = a0; a < a1; ++a) {
      float pe = 0.f, pw = 0.f;
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int f = lane + 64 * q;
        const float xv = f < n_feat ? x[(int64_t)a * ldx + f] : 0.f;
        pe = fmaf(xv, hv[q], pe);
        pw = fmaf(xv, drv[q], pw);
      }
      const float w = expf(wave_sum(pe) - emax);
      denom += w;
      wdw += w * wave_sum(pw);
    }
    const float inv = a1 > a0 ? 1.f / denom : 0.f;
    const float mean_dw = wdw * inv;
    for (int a = a0; a < a1; ++a) {
      float xv[kPer], pe = 0.f, pw = 0.f;
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int f = lane + 64 * q;
        xv[q] = f < n_feat ? x[(int64_t)a * ldx + f] : 0.f;
        pe = fmaf(xv[q], hv[q], pe);
        pw = fmaf(xv[q], drv[q], pw);
      }
      const float w = expf(wave_sum(pe) - emax) * inv;
      const float de = w * (wave_sum(pw) - mean_dw);
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int f = lane + 64 * q;
        if (f < n_feat) {
          const float g = fmaf(w, drv[q], de * hv[q]);
          float* p = dx + (int64_t)a * lddx + f;
          *p = accumulate ? *p + g : g;
          dhacc[q] = fmaf(de, xv[q], dhacc[q]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const int f = lane + 64 * q;
      if (f < n_feat) dh[(int64_t)m * lddh + f] = dhacc[q];
    }
  }
}

}  // namespace gcmi

using namespace gcmi;

extern "C" {

int gcmi_edge_network_sum(const float* d_g, int64_t ldg, int32_t n_hidden, int32_t n_pair_feat,
                          const float* d_pair_feat, int64_t ldp, const int32_t* d_dst_ptr, const int32_t* d_src,
                          int32_t n_dst, float* d_out, int64_t ldo, void* stream) {
  GCMI_CHECK_ARG(n_hidden > 0 && n_pair_feat > 0 && n_pair_feat <= kMaxPairFeat && n_dst >= 0 &&
                     ldg >= (int64_t)(n_pair_feat + 1) * n_hidden && ldp >= n_pair_feat && ldo >= n_hidden,
                 "edge_network_sum: bad shape");
  if (n_dst == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_g && d_pair_feat && d_dst_ptr && d_src && d_out, "edge_network_sum: NULL buffer");
  hipLaunchKernelGGL(edge_network_kernel, dim3(grid_for((int64_t)n_dst * 64, kMpBlock)), dim3(kMpBlock), 0,
                     (hipStream_t)stream, d_g, ldg, n_hidden, n_pair_feat, d_pair_feat, ldp, d_dst_ptr, d_src, n_dst,
                     d_out, ldo);
  GCMI_CHECK_LAUNCH("edge_network_sum");
  return GCMI_OK;
}

int gcmi_edge_network_moments(const float* d_h, int64_t ldh, int32_t n_hidden, int32_t n_pair_feat,
                              const float* d_pair_feat, int64_t ldp, const int32_t* d_dst_ptr, const int32_t* d_src,
                              int32_t n_dst, float* d_t, int64_t ldt, void* stream) {
  GCMI_CHECK_ARG(n_hidden > 0 && n_hidden <= 128 && n_pair_feat > 0 && n_pair_feat <= 16 && n_dst >= 0 &&
                     ldh >= n_hidden && ldp >= n_pair_feat && ldt >= (int64_t)(n_pair_feat + 1) * n_hidden,
                 "edge_network_moments: bad shape (n_hidden <= 128, n_pair_feat <= 16)");
  if (n_dst == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_h && d_pair_feat && d_dst_ptr && d_src && d_t, "edge_network_moments: NULL buffer");
  const dim3 grid(grid_for((int64_t)n_dst * 64, kMpBlock));
  hipStream_t st = (hipStream_t)stream;
  if (n_hidden <= 64)
    hipLaunchKernelGGL((edge_moments_kernel<16, 1>), grid, dim3(kMpBlock), 0, st, d_h, ldh, n_hidden, n_pair_feat,
                       d_pair_feat, ldp, d_dst_ptr, d_src, n_dst, d_t, ldt);
  else
    hipLaunchKernelGGL((edge_moments_kernel<16, 2>), grid, dim3(kMpBlock), 0, st, d_h, ldh, n_hidden, n_pair_feat,
                       d_pair_feat, ldp, d_dst_ptr, d_src, n_dst, d_t, ldt);
  GCMI_CHECK_LAUNCH("edge_network_moments");
  return GCMI_OK;
}

int gcmi_edge_network_moments_mol(const float* d_h, int64_t ldh, int32_t n_hidden, int32_t n_pair_feat,
                                  const float* d_pair_feat, int64_t ldp, const int32_t* d_dst_ptr, const int32_t* d_src,
                                  int32_t n_dst, const int32_t* d_mol_ptr, int32_t n_mols, int32_t max_mol_atoms,
                                  float* d_t, int64_t ldt, void* stream) {
  // Measured (QM9-like molecules, 18 atoms, all pairs; us per launch, this kernel / the per-atom kernel): 18 k atoms
  // 93 / 52, 73 k atoms 229 / 227, 147 k atoms 404 / 488 -- both are bound by the moment arithmetic and the 3.6 KB row
  // every atom writes; the staged rows only pay once the per-atom gathers outgrow L2.  GCMI_EDGE_MOMENTS_MOL=1 / 0
  // forces it on / off.
  static const int mode = getenv("GCMI_EDGE_MOMENTS_MOL") ? atoi(getenv("GCMI_EDGE_MOMENTS_MOL")) : -1;
  const bool on = mode == 1 || (mode != 0 && n_dst >= 96 * 1024);
  if (!on || d_mol_ptr == nullptr || n_mols <= 0)
    return gcmi_edge_network_moments(d_h, ldh, n_hidden, n_pair_feat, d_pair_feat, ldp, d_dst_ptr, d_src, n_dst, d_t, ldt, stream);
  GCMI_CHECK_ARG(n_hidden > 0 && n_hidden <= 128 && n_pair_feat > 0 && n_pair_feat <= 16 && n_dst >= 0 &&
                     ldh >= n_hidden && ldp >= n_pair_feat && ldt >= (int64_t)(n_pair_feat + 1) * n_hidden,
                 "edge_network_moments_mol: bad shape (n_hidden <= 128, n_pair_feat <= 16)");
  if (n_dst == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_h && d_pair_feat && d_dst_ptr && d_src && d_t, "edge_network_moments_mol: NULL buffer");
  GCMI_CHECK_ARG(aligned16(d_h) || (n_hidden & 3) || (ldh & 3), "edge_network_moments_mol: state rows must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)std::min<int64_t>(n_mols, 256 * 32));
  // LDS for the state rows of one molecule: the largest one if the caller knows it (max_mol_atoms > 0), at most 48 KB
  int lds_floats = kMolLdsFloats;
  if (max_mol_atoms > 0) lds_floats = (int)std::min<int64_t>(kMolLdsFloats, ((int64_t)max_mol_atoms * n_hidden + 63) / 64 * 64);
#define LAUNCH_EMM(KP, NC)                                                                                                 \
  hipLaunchKernelGGL((edge_moments_mol_kernel<KP, NC>), grid, dim3(kMpBlock), sizeof(float) * (size_t)lds_floats, st, d_h, \
                     ldh, n_hidden, n_pair_feat, d_pair_feat, ldp, d_dst_ptr, d_src, d_mol_ptr, n_mols, d_t, ldt, lds_floats)
  if (n_pair_feat <= 8) {
    if (n_hidden <= 64) LAUNCH_EMM(8, 1); else LAUNCH_EMM(8, 2);
  } else {
    if (n_hidden <= 64) LAUNCH_EMM(16, 1); else LAUNCH_EMM(16, 2);
  }
#undef LAUNCH_EMM
  GCMI_CHECK_LAUNCH("edge_network_moments_mol");
  return GCMI_OK;
}

int gcmi_gru_gates(float* d_z, float* d_r, const float* d_h, float* d_hr, int64_t n, void* stream) {
  GCMI_CHECK_ARG(n >= 0, "gru_gates: bad size");
  if (n == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_z && d_r && d_h && d_hr, "gru_gates: NULL buffer");
  hipLaunchKernelGGL(gru_gates_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, d_z, d_r, d_h, d_hr, n);
  GCMI_CHECK_LAUNCH("gru_gates");
  return GCMI_OK;
}

int gcmi_gru_out(const float* d_z, const float* d_hpre, const float* d_x, float* d_out, int64_t n, void* stream) {
  GCMI_CHECK_ARG(n >= 0, "gru_out: bad size");
  if (n == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_z && d_hpre && d_x && d_out, "gru_out: NULL buffer");
  hipLaunchKernelGGL(gru_out_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, d_z, d_hpre, d_x, d_out, n);
  GCMI_CHECK_LAUNCH("gru_out");
  return GCMI_OK;
}

int gcmi_set2set_attend(const float* d_x, int64_t ldx, int32_t n_feat, const int32_t* d_mol_ptr, int32_t n_mols,
                        const float* d_h, int64_t ldh, float* d_qstar, int64_t ldq, void* stream) {
  GCMI_CHECK_ARG(n_feat > 0 && n_feat <= 512 && n_mols >= 0 && ldx >= n_feat && ldh >= n_feat && ldq >= 2 * n_feat,
                 "set2set_attend: bad shape (n_feat <= 512)");
  if (n_mols == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_x && d_mol_ptr && d_h && d_qstar, "set2set_attend: NULL buffer");
  hipLaunchKernelGGL(set2set_attend_kernel, dim3(grid_for((int64_t)n_mols * 64, kMpBlock)), dim3(kMpBlock), 0,
                     (hipStream_t)stream, d_x, ldx, n_feat, d_mol_ptr, n_mols, d_h, ldh, d_qstar, ldq);
  GCMI_CHECK_LAUNCH("set2set_attend");
  return GCMI_OK;
}

int gcmi_lstm_cell(const float* d_z, int64_t ldz, int32_t n_hidden, int64_t n_rows, float* d_c, int64_t ldc,
                   float* d_h, int64_t ldh, void* stream) {
  GCMI_CHECK_ARG(n_hidden > 0 && n_rows >= 0 && ldz >= 4 * n_hidden && ldc >= n_hidden && ldh >= n_hidden,
                 "lstm_cell: bad shape");
  if (n_rows == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_z && d_c && d_h, "lstm_cell: NULL buffer");
  hipLaunchKernelGGL(lstm_cell_kernel, dim3(grid_for(n_rows * n_hidden, 256)), dim3(256), 0, (hipStream_t)stream, d_z,
                     ldz, n_hidden, n_rows, d_c, ldc, d_h, ldh);
  GCMI_CHECK_LAUNCH("lstm_cell");
  return GCMI_OK;
}

int gcmi_gru_gates_bwd(const float* d_z, const float* d_r, const float* d_h, const float* d_dz, const float* d_dhr,
                       float* d_dzp, float* d_drp, float* d_dh, int64_t n, void* stream) {
  GCMI_CHECK_ARG(n >= 0, "gru_gates_bwd: bad size");
  if (n == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_z && d_r && d_h && d_dz && d_dhr && d_dzp && d_drp && d_dh, "gru_gates_bwd: NULL buffer");
  hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, d_z, d_r, d_h,
                     d_dz, d_dhr, d_dzp, d_drp, d_dh, n);
  GCMI_CHECK_LAUNCH("gru_gates_bwd");
  return GCMI_OK;
}

int gcmi_gru_out_bwd(const float* d_z, const float* d_hpre, const float* d_x, const float* d_dout, float* d_dz,
                     float* d_dhpre, float* d_dx, int64_t n, void* stream) {
  GCMI_CHECK_ARG(n >= 0, "gru_out_bwd: bad size");
  if (n == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_z && d_hpre && d_x && d_dout && d_dz && d_dhpre && d_dx, "gru_out_bwd: NULL buffer");
  hipLaunchKernelGGL(gru_out_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, d_z, d_hpre, d_x,
                     d_dout, d_dz, d_dhpre, d_dx, n);
  GCMI_CHECK_LAUNCH("gru_out_bwd");
  return GCMI_OK;
}

int gcmi_lstm_cell_bwd(const float* d_z, int64_t ldz, int32_t n_hidden, int64_t n_rows, const float* d_c_prev,
                       const float* d_dh, const float* d_dc_next, float* d_dz, float* d_dc_prev, void* stream) {
  GCMI_CHECK_ARG(n_hidden > 0 && n_rows >= 0 && ldz >= 4 * n_hidden, "lstm_cell_bwd: bad shape");
  if (n_rows == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_z && d_c_prev && d_dh && d_dz && d_dc_prev, "lstm_cell_bwd: NULL buffer");
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(grid_for(n_rows * n_hidden, 256)), dim3(256), 0, (hipStream_t)stream,
                     d_z, ldz, n_hidden, n_rows, d_c_prev, d_dh, d_dc_next, d_dz, d_dc_prev);
  GCMI_CHECK_LAUNCH("lstm_cell_bwd");
  return GCMI_OK;
}

int gcmi_set2set_attend_bwd(const float* d_x, int64_t ldx, int32_t n_feat, const int32_t* d_mol_ptr, int32_t n_mols,
                            const float* d_h, int64_t ldh, const float* d_dq, int64_t lddq, float* d_dx, int64_t lddx,
                            int32_t accumulate, float* d_dh, int64_t lddh, void* stream) {
  GCMI_CHECK_ARG(n_feat > 0 && n_feat <= 512 && n_mols >= 0 && ldx >= n_feat && ldh >= n_feat && lddq >= 2 * n_feat &&
                     lddx >= n_feat && lddh >= n_feat,
                 "set2set_attend_bwd: bad shape (n_feat <= 512)");
  if (n_mols == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_x && d_mol_ptr && d_h && d_dq && d_dx && d_dh, "set2set_attend_bwd: NULL buffer");
  hipLaunchKernelGGL(set2set_attend_bwd_kernel, dim3(grid_for((int64_t)n_mols * 64, kMpBlock)), dim3(kMpBlock), 0,
                     (hipStream_t)stream, d_x, ldx, n_feat, d_mol_ptr, n_mols, d_h, ldh, d_dq, lddq, d_dx, lddx,
                     accumulate, d_dh, lddh);
  GCMI_CHECK_LAUNCH("set2set_attend_bwd");
  return GCMI_OK;
}

}  // extern "C"
