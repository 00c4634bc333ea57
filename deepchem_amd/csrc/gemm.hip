// Row-segmented GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact
// fp32, an fmaf chain in k order -- MI355X_MICROARCH "FP32-input MFMA").
//
//   gcmi_seg_gemm        out[rows_s] = act(a1[rows_s].w1[s] + a2[rows_s].w2[s] + bias[s])
//   gcmi_seg_gemm_wgrad  dw[s] += a[rows_s]^T . g[rows_s],  dbias[s] += colsum g[rows_s]
//
// Why "segmented": a collated batch is sorted by atom degree, GraphConv has one
// weight pair per degree (layers.py:6202-6226), so every 64-row tile lies
// inside ONE segment and sees ONE weight block -- the 21 small matmuls of the
// reference become a single launch.  The dense layer and the task heads are
// the one-segment case.
//
// Forward tile: 64 rows x 64 columns per 256-thread workgroup, 2x2 waves, each
// wave one 32x32 accumulator (16 VGPRs).  K is walked in chunks of 32 staged
// through LDS (A chunk padded to 33 floats/row, B chunk to 65: conflict-free
// ds_read_b32 for the 32x32x2 operand maps A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]).
// wgrad: no LDS at all -- the operand maps of a^T.g are lane-contiguous along
// the feature dimension, so both operands are loaded straight from global
// memory (one dword per lane per row pair) and every wave keeps <= 8
// accumulators (k <= 256 per pass).
#include <atomic>

#include "common.h"

namespace gcmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMaxSeg = 16;
constexpr int kGBlock = 256;
constexpr int BM = 64;
constexpr int BNT = 64;
constexpr int KC = 32;
constexpr int BM2 = 128;

struct SegTable {
  int32_t n_seg;
  int32_t seg_begin[kMaxSeg];
  int32_t seg_end[kMaxSeg];
  int32_t tile_start[kMaxSeg + 1];
  int64_t w1_off[kMaxSeg];  // < 0: term absent
  int64_t w2_off[kMaxSeg];
  int64_t bias_off[kMaxSeg];
};

__device__ __forceinline__ int seg_of_tile(const SegTable& st, int b) {
  int s = 0;
#pragma unroll
  for (int k = 1; k < kMaxSeg; ++k) s += (k < st.n_seg && b >= st.tile_start[k]) ? 1 : 0;
  return s;
}

template <typename T>
__device__ __forceinline__ T pick_seg(const T* a, int s) {
  T v = a[0];
#pragma unroll
  for (int k = 1; k < kMaxSeg; ++k) v = (s == k) ? a[k] : v;
  return v;
}

// stage one K-chunk of A (64 rows x KC) and of W (KC x 64 cols) into LDS
template <bool VEC4>
__device__ __forceinline__ void stage_a(float (*As)[KC + 1], const float* __restrict__ a,
                                        int64_t lda, int row0, int rows_valid, int k0, int K) {
  const int tid = threadIdx.x;
  if constexpr (VEC4) {
    const int kq = (tid & 7) * 4;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int r = (tid >> 3) + pass * 32;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < rows_valid && k0 + kq < K) {
        // lda % 4 == 0 and lda >= K: the 16 bytes stay inside the row even when K % 4 != 0;
        // columns >= K (row padding) are forced to zero
        v = *reinterpret_cast<const float4*>(a + (int64_t)(row0 + r) * lda + k0 + kq);
        if (k0 + kq + 1 >= K) v.y = 0.f;
        if (k0 + kq + 2 >= K) v.z = 0.f;
        if (k0 + kq + 3 >= K) v.w = 0.f;
      }
      As[r][kq + 0] = v.x;
      As[r][kq + 1] = v.y;
      As[r][kq + 2] = v.z;
      As[r][kq + 3] = v.w;
    }
  } else {
    const int kk = tid & 31;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int r = (tid >> 5) + pass * 8;
      float v = 0.f;
      if (r < rows_valid && k0 + kk < K) v = a[(int64_t)(row0 + r) * lda + k0 + kk];
      As[r][kk] = v;
    }
  }
}

template <bool TRANS>
__device__ __forceinline__ void stage_b(float (*Bs)[BNT + 1], const float* __restrict__ w, int K,
                                        int n_out, int col0, int k0) {
  const int tid = threadIdx.x;
  if constexpr (!TRANS) {  // w is K x n_out
    const int j = tid & 63;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int kk = (tid >> 6) + pass * 4;
      float v = 0.f;
      if (k0 + kk < K && col0 + j < n_out) v = w[(int64_t)(k0 + kk) * n_out + col0 + j];
      Bs[kk][j] = v;
    }
  } else {  // w is n_out x K
    const int kk = tid & 31;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int j = (tid >> 5) + pass * 8;
      float v = 0.f;
      if (k0 + kk < K && col0 + j < n_out) v = w[(int64_t)(col0 + j) * K + k0 + kk];
      Bs[kk][j] = v;
    }
  }
}

template <bool TRANS, bool VEC4>
__global__ void __launch_bounds__(kGBlock)
seg_gemm_kernel(SegTable st, const float* __restrict__ a1, int64_t lda1, int k1,
                const float* __restrict__ w1, const float* __restrict__ a2, int64_t lda2, int k2,
                const float* __restrict__ w2, const float* __restrict__ bias, int n_out, int act,
                float* __restrict__ out, int64_t ldo) {
  __shared__ float As[BM][KC + 1];
  __shared__ float Bs[KC][BNT + 1];
  const int b = blockIdx.x;
  const int s = seg_of_tile(st, b);
  const int row0 = pick_seg(st.seg_begin, s) + (b - pick_seg(st.tile_start, s)) * BM;
  const int seg_end = pick_seg(st.seg_end, s);
  const int rows_valid = (seg_end - row0 < BM) ? seg_end - row0 : BM;
  const int col0 = blockIdx.y * BNT;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int wr = wave >> 1, wc = wave & 1;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float* a = p == 0 ? a1 : a2;
    const int64_t lda = p == 0 ? lda1 : lda2;
    const int K = p == 0 ? k1 : k2;
    const int64_t woff = p == 0 ? pick_seg(st.w1_off, s) : pick_seg(st.w2_off, s);
    const float* wbase = p == 0 ? w1 : w2;
    if (a == nullptr || wbase == nullptr || woff < 0) continue;  // block-uniform
    const float* w = wbase + woff;
    for (int k0 = 0; k0 < K; k0 += KC) {
      stage_a<VEC4>(As, a, lda, row0, rows_valid, k0, K);
      stage_b<TRANS>(Bs, w, K, n_out, col0, k0);
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < KC; kk += 2) {
        const float av = As[wr * 32 + (lane & 31)][kk + (lane >> 5)];
        const float bv = Bs[kk + (lane >> 5)][wc * 32 + (lane & 31)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
      }
      __syncthreads();
    }
  }
  const int64_t boff = pick_seg(st.bias_off, s);
  const int col = col0 + wc * 32 + (lane & 31);
  if (col < n_out) {
    const float bv = (bias != nullptr && boff >= 0) ? bias[boff + col] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int r = wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      if (r < rows_valid) {
        float v = acc[reg] + bv;
        if (act == 1) v = v > 0.f ? v : 0.f;
        if (act == 2) v += out[(int64_t)(row0 + r) * ldo + col];
        out[(int64_t)(row0 + r) * ldo + col] = v;
      }
    }
  }
}

// ------------------------------------------------------------------ forward, pipelined
// Same contract as seg_gemm_kernel for 16-byte addressable rows, restructured for the matrix
// pipe: 128 rows x (NT*32) columns per workgroup (each wave 32 rows x all columns: one A
// fragment feeds NT MFMAs), and the NEXT K-chunk is fetched from global memory into registers
// while the MFMAs of the current chunk run, so HBM/L2 latency overlaps the matrix work of the
// same workgroup instead of relying on other workgroups to cover it.

template <bool TRANS, int NT, int BMT, bool FRAG, int KCT>
__global__ void __launch_bounds__(kGBlock)
seg_gemm2_kernel(SegTable st, const float* __restrict__ a1, int64_t lda1, int k1,
                 const float* __restrict__ w1, const float* __restrict__ a2, int64_t lda2, int k2,
                 const float* __restrict__ w2, const float* __restrict__ bias, int n_out, int act,
                 float* __restrict__ out, int64_t ldo, int diag) {
  constexpr int BN2 = NT * 32;
  constexpr int BPASS = (KCT * BN2) / kGBlock;  // B elements per thread per chunk
  constexpr int LPRW = KCT / 4;          // lanes per A row (16 bytes each)
  constexpr int RPP = kGBlock / LPRW;    // A rows per pass
  constexpr int APASS = BMT / RPP;   // A float4 per thread per chunk
  constexpr int WR = BMT / 32;      // wave rows: BMT=128 -> 4 waves x 32 rows; BMT=64 -> 2 x 2 waves
  constexpr int WC = 4 / WR;        // waves side by side along the columns
  constexpr int NTW = NT / WC > 0 ? NT / WC : 1;  // 32-column tiles per wave
  __shared__ float As[BMT][KCT + 1];
  __shared__ float Bs[KCT][BN2 + 1];
  const int b = blockIdx.x;
  const int s = seg_of_tile(st, b);
  const int row0 = pick_seg(st.seg_begin, s) + (b - pick_seg(st.tile_start, s)) * BMT;
  const int seg_end = pick_seg(st.seg_end, s);
  const int rows_valid = (seg_end - row0 < BMT) ? seg_end - row0 : BMT;
  const int col0 = blockIdx.y * BN2;
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int wrow = wave % WR, wcol = wave / WR;
  f32x16 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const int64_t woff1 = pick_seg(st.w1_off, s), woff2 = pick_seg(st.w2_off, s);
  const bool on1 = a1 != nullptr && w1 != nullptr && woff1 >= 0;
  const bool on2 = a2 != nullptr && w2 != nullptr && woff2 >= 0;
  const int n1 = on1 ? (k1 + KCT - 1) / KCT : 0;
  const int n2 = on2 ? (k2 + KCT - 1) / KCT : 0;
  const int nchunks = n1 + n2;

  // Prefetch registers.  Loads are UNCONDITIONAL from clamped (always valid) addresses and
  // nothing consumes them until sstore(): a guarded load ("cond ? load : 0") makes hipcc branch
  // around it and wait vmcnt(0) on the spot, which serialises the prefetch.  Out-of-range rows,
  // columns >= K and columns >= n_out are zeroed when the registers go to LDS.
  float4 ra[APASS];
  float rb[BPASS];
  int pend_k0 = 0, pend_K = 0;
  auto gload = [&](int c) {
    const bool first = c < n1;
    const float* a = first ? a1 : a2;
    const int64_t lda = first ? lda1 : lda2;
    const int K = first ? k1 : k2;
    const float* w = first ? w1 + woff1 : w2 + woff2;
    const int k0 = (first ? c : c - n1) * KCT;
    pend_k0 = k0;
    pend_K = K;
    const int kcol = k0 + (tid % LPRW) * 4;
    const int kc = kcol < K ? kcol : 0;
#pragma unroll
    for (int pass = 0; pass < APASS; ++pass) {
      const int r = tid / LPRW + pass * RPP;
      const int rc = r < rows_valid ? r : rows_valid - 1;
      ra[pass] = *reinterpret_cast<const float4*>(a + (int64_t)(row0 + rc) * lda + kc);
    }
    if constexpr (!TRANS) {  // w is K x n_out
      const int j = col0 + tid % BN2;
      const int jc = j < n_out ? j : n_out - 1;
#pragma unroll
      for (int pass = 0; pass < BPASS; ++pass) {
        const int kk = k0 + tid / BN2 + pass * (kGBlock / BN2);
        const int kkc = kk < K ? kk : K - 1;
        rb[pass] = w[(int64_t)kkc * n_out + jc];
      }
    } else {  // w is n_out x K
      const int kk = k0 + (tid % KCT);
      const int kkc = kk < K ? kk : K - 1;
#pragma unroll
      for (int pass = 0; pass < BPASS; ++pass) {
        const int j = col0 + tid / KCT + pass * (kGBlock / KCT);
        const int jc = j < n_out ? j : n_out - 1;
        rb[pass] = w[(int64_t)jc * K + kkc];
      }
    }
  };
  auto sstore = [&]() {
    const int kq = (tid % LPRW) * 4;
    const int tail = pend_K - (pend_k0 + kq);  // columns of this float4 inside the matrix
#pragma unroll
    for (int pass = 0; pass < APASS; ++pass) {
      const int r = tid / LPRW + pass * RPP;
      const bool ok = r < rows_valid;
      As[r][kq + 0] = (ok && tail > 0) ? ra[pass].x : 0.f;
      As[r][kq + 1] = (ok && tail > 1) ? ra[pass].y : 0.f;
      As[r][kq + 2] = (ok && tail > 2) ? ra[pass].z : 0.f;
      As[r][kq + 3] = (ok && tail > 3) ? ra[pass].w : 0.f;
    }
    if constexpr (!TRANS) {
      const int j = tid % BN2;
      const bool jok = col0 + j < n_out;
#pragma unroll
      for (int pass = 0; pass < BPASS; ++pass) {
        const int kk = tid / BN2 + pass * (kGBlock / BN2);
        Bs[kk][j] = (jok && pend_k0 + kk < pend_K) ? rb[pass] : 0.f;
      }
    } else {
      const int kk = tid % KCT;
      const bool kok = pend_k0 + kk < pend_K;
#pragma unroll
      for (int pass = 0; pass < BPASS; ++pass) {
        const int j = tid / KCT + pass * (kGBlock / KCT);
        Bs[kk][j] = (kok && col0 + j < n_out) ? rb[pass] : 0.f;
      }
    }
  };

  if (nchunks > 0) {
    gload(0);
    sstore();
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks && !(diag & 2)) gload(c + 1);  // in flight while the matrix pipe works on chunk c
      if (!(diag & 1)) {
        if constexpr (FRAG) {
          // fragments of half a chunk first, then the MFMAs back to back: the LDS latency is
          // paid once per 8 k-steps instead of once per k-step
#pragma unroll
          for (int h = 0; h < KCT / 16; ++h) {
            float af[8], bf[NTW][8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int kk = h * 16 + q * 2;
              af[q] = As[wrow * 32 + (lane & 31)][kk + half];
#pragma unroll
              for (int t = 0; t < NTW; ++t) bf[t][q] = Bs[kk + half][(wcol * NTW + t) * 32 + (lane & 31)];
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMAs
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
              for (int t = 0; t < NTW; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[t][q], af[q], acc[t], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int kk = 0; kk < KCT; kk += 2) {
            const float av = As[wrow * 32 + (lane & 31)][kk + half];
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
              const float bv = Bs[kk + half][(wcol * NTW + t) * 32 + (lane & 31)];
              // operands swapped on purpose: the accumulator holds out^T -- lane = atom row, 4
              // consecutive registers = 4 consecutive output columns -> 16-byte stores per lane
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[t], 0, 0, 0);
            }
          }
        }
      }
      __syncthreads();
      if (c + 1 < nchunks && !(diag & 2)) {
        sstore();
        __syncthreads();
      }
    }
  }
  const int64_t boff = pick_seg(st.bias_off, s);
  const bool has_bias = bias != nullptr && boff >= 0;
  const int r = wrow * 32 + (lane & 31);
  float bv[NTW][4][4];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = col0 + (wcol * NTW + t) * 32 + 8 * rg + 4 * half + q;
        bv[t][rg][q] = has_bias ? bias[boff + (c < n_out ? c : n_out - 1)] : 0.f;
      }
  if (r < rows_valid) {
    float* orow = out + (int64_t)(row0 + r) * ldo;
    const bool vec_out = (ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int c = col0 + (wcol * NTW + t) * 32 + 8 * rg + 4 * half;  // columns c .. c+3 = registers 4rg .. 4rg+3
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float x = acc[t][4 * rg + q] + bv[t][rg][q];
          if (act == 1) x = x > 0.f ? x : 0.f;
          v[q] = x;
        }
        if (vec_out && c + 3 < n_out) {
          if (act == 2) {
            const float4 o = *reinterpret_cast<const float4*>(orow + c);
            v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
          }
          *reinterpret_cast<float4*>(orow + c) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (c + q < n_out) orow[c + q] = act == 2 ? v[q] + orow[c + q] : v[q];
        }
      }
    }
  }
}

// ------------------------------------------------------------------ wgrad
template <int KT, bool TRANS>
__global__ void __launch_bounds__(kGBlock)
wgrad_kernel(SlabTable st, const float* __restrict__ a, int64_t lda, int k, int kt0,
             const float* __restrict__ g, int64_t ldg, int n, int ntw, float* __restrict__ dw,
             float* __restrict__ dbias) {
  const int b = blockIdx.x;
  int s = 0;
#pragma unroll
  for (int q = 1; q < kMaxSeg; ++q) s += (q < st.n_seg && b >= st.slab_start[q]) ? 1 : 0;
  const int seg_end = pick_seg(st.seg_end, s);
  const int slab0 = pick_seg(st.seg_begin, s) + (b - pick_seg(st.slab_start, s)) * st.slab_rows;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int nt = blockIdx.y * ntw + (wave % ntw);  // n-tile of this wave
  const int rp = wave / ntw;                       // row part of this wave
  const int parts = 4 / ntw;
  const int part_rows = st.slab_rows / parts;      // slab_rows is a multiple of 8
  const int r_begin = slab0 + rp * part_rows;
  int r_end = r_begin + part_rows;
  if (r_end > seg_end) r_end = seg_end;
  const int ncol = nt * 32 + (lane & 31);
  const bool n_ok = ncol < n;
  const int half = lane >> 5;
  f32x16 acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float bsum = 0.f;
  bool k_ok[KT];
  int kcol[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    kcol[t] = (kt0 + t) * 32 + (lane & 31);
    k_ok[t] = kcol[t] < k;
  }
  // Row pairs are loaded U at a time into one register set while the MFMAs of the previous
  // set run (software double buffering): with 2-4 waves per SIMD that keeps enough bytes in
  // flight to cover HBM latency.
  constexpr int U = KT <= 4 ? 8 : 4;
  float av[2][U][KT], bv[2][U];
  auto load_set = [&](int buf, int r) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rr = r + 2 * u + half;
      const bool ok = rr < r_end;
      bv[buf][u] = (ok && n_ok) ? g[(int64_t)rr * ldg + ncol] : 0.f;
#pragma unroll
      for (int t = 0; t < KT; ++t)
        av[buf][u][t] = (ok && k_ok[t]) ? a[(int64_t)rr * lda + kcol[t]] : 0.f;
    }
  };
  auto mma_set = [&](int buf) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bsum += bv[buf][u];
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        // TRANS (nn.Linear layout, dW stored n x k): swap the operands so that the accumulator
        // holds dW^T -- its lane index is then the k-feature, i.e. the CONTIGUOUS index of the
        // destination, and every atomic wave-instruction covers two 128-byte runs instead of 64
        // different rows (which the memory-side atomic units serve an order of magnitude slower)
        if constexpr (TRANS)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[buf][u], av[buf][u][t], acc[t], 0, 0, 0);
        else
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u][t], bv[buf][u], acc[t], 0, 0, 0);
      }
    }
  };
  if (r_begin < r_end) {
    load_set(0, r_begin);
    int r = r_begin;
    while (true) {
      const int r1 = r + 2 * U;
      if (r1 < r_end) load_set(1, r1);
      mma_set(0);
      if (r1 >= r_end) break;
      const int r2 = r1 + 2 * U;
      if (r2 < r_end) load_set(0, r2);
      mma_set(1);
      if (r2 >= r_end) break;
      r = r2;
    }
  }
  // combine the row parts of this workgroup through LDS, so that ONE wave per n-tile
  // issues the atomics (fewer, less contended adds into the small dW block)
  if (parts > 1) {
    extern __shared__ float red[];
    float* racc = red + (size_t)(wave % ntw) * KT * 16 * 64;
    float* rb = red + (size_t)ntw * KT * 16 * 64 + (wave % ntw) * 64;
    for (int p = 1; p < parts; ++p) {
      if (rp == p) {
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) racc[(t * 16 + reg) * 64 + lane] = acc[t][reg];
        rb[lane] = bsum;
      }
      __syncthreads();
      if (rp == 0) {
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) acc[t][reg] += racc[(t * 16 + reg) * 64 + lane];
        bsum += rb[lane];
      }
      __syncthreads();
    }
    if (rp != 0) return;
  }
  if (slab0 >= seg_end) return;
  const int64_t woff = pick_seg(st.dw_off, s);
  if (woff >= 0) {
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int rix = (reg & 3) + 8 * (reg >> 2) + 4 * half;  // accumulator row of this register
        if constexpr (TRANS) {
          const int nc = nt * 32 + rix;  // rows of dW^T = output columns n
          if (nc < n && k_ok[t]) atomicAdd(dw + woff + (int64_t)nc * k + kcol[t], acc[t][reg]);
        } else {
          const int kf = (kt0 + t) * 32 + rix;
          if (kf < k && n_ok) atomicAdd(dw + woff + (int64_t)kf * n + ncol, acc[t][reg]);
        }
      }
    }
  }
  if (dbias != nullptr && kt0 == 0) {
    const int64_t boff = pick_seg(st.db_off, s);
    bsum += __shfl_xor(bsum, 32);
    if (half == 0 && n_ok && boff >= 0) atomicAdd(dbias + boff + ncol, bsum);
  }
}

template <int V>
__global__ void __launch_bounds__(256)
relu_bwd_kernel(float* __restrict__ g, int64_t ldg, const float* __restrict__ y, int64_t ldy,
                int64_t slots, int lpr) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < slots; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / lpr;
    const int c = (int)(e - r * lpr) * V;
    if constexpr (V == 4) {
      float4 gv = *reinterpret_cast<float4*>(g + r * ldg + c);
      const float4 yv = *reinterpret_cast<const float4*>(y + r * ldy + c);
      gv.x = yv.x > 0.f ? gv.x : 0.f;
      gv.y = yv.y > 0.f ? gv.y : 0.f;
      gv.z = yv.z > 0.f ? gv.z : 0.f;
      gv.w = yv.w > 0.f ? gv.w : 0.f;
      *reinterpret_cast<float4*>(g + r * ldg + c) = gv;
    } else {
      g[r * ldg + c] = y[r * ldy + c] > 0.f ? g[r * ldg + c] : 0.f;
    }
  }
}

// GCMI_OPT_GEMM_EXACT: 1 = every product on the exact-fp32 MFMA (a k-ordered fmaf chain: the same
// summation order as the reference's CPU matmuls, so discrete decisions downstream -- arg-max of
// the pools, ReLU boundaries -- fall the same way and whole training trajectories track the
// reference to ~1e-5); 0 (default) = the faster split-bf16 kernels, equally accurate per product
// but with their own rounding.
static std::atomic<int> g_gemm_exact{getenv("GCMI_GEMM_EXACT") && atoi(getenv("GCMI_GEMM_EXACT")) == 1 ? 1 : 0};
bool gemm_exact_mode() { return g_gemm_exact.load(std::memory_order_relaxed) != 0; }
// GCMI_OPT_FUSED_BN_STATS (env GCMI_GEMM_STATS=0 starts it off)
static std::atomic<int> g_fused_bn_stats{getenv("GCMI_GEMM_STATS") && atoi(getenv("GCMI_GEMM_STATS")) == 0 ? 0 : 1};

}  // namespace gcmi

using namespace gcmi;

extern "C" {

int gcmi_set_option(int32_t option, int32_t value) {
  GCMI_CHECK_ARG(option == GCMI_OPT_GEMM_EXACT || option == GCMI_OPT_FUSED_BN_STATS || option == GCMI_OPT_FUSED_BWD ||
                     option == GCMI_OPT_READOUT_PIPELINED,
                 "set_option: unknown option %d", option);
  if (option == GCMI_OPT_GEMM_EXACT) g_gemm_exact.store(value != 0 ? 1 : 0, std::memory_order_relaxed);
  else if (option == GCMI_OPT_FUSED_BN_STATS) g_fused_bn_stats.store(value != 0 ? 1 : 0, std::memory_order_relaxed);
  else if (option == GCMI_OPT_READOUT_PIPELINED) set_readout_pipelined(value);
  else set_fused_bwd(value);
  return GCMI_OK;
}

int gcmi_get_option(int32_t option, int32_t* value) {
  GCMI_CHECK_ARG((option == GCMI_OPT_GEMM_EXACT || option == GCMI_OPT_FUSED_BN_STATS || option == GCMI_OPT_FUSED_BWD ||
                  option == GCMI_OPT_FUSED_BWD_LAUNCHES || option == GCMI_OPT_READOUT_PIPELINED) && value,
                 "get_option: unknown option %d", option);
  if (option == GCMI_OPT_GEMM_EXACT) *value = g_gemm_exact.load(std::memory_order_relaxed);
  else if (option == GCMI_OPT_FUSED_BN_STATS) *value = g_fused_bn_stats.load(std::memory_order_relaxed);
  else if (option == GCMI_OPT_FUSED_BWD) *value = get_fused_bwd();
  else if (option == GCMI_OPT_READOUT_PIPELINED) *value = get_readout_pipelined();
  else *value = fused_bwd_launches();
  return GCMI_OK;
}

int gcmi_seg_gemm(int32_t n_seg, const int32_t* seg_begin, const int32_t* seg_end,
                  const float* d_a1, int64_t lda1, int32_t k1, const float* d_w1,
                  const int64_t* w1_off, const float* d_a2, int64_t lda2, int32_t k2,
                  const float* d_w2, const int64_t* w2_off, const float* d_bias,
                  const int64_t* bias_off, int32_t n_out, int32_t trans_w, int32_t act,
                  float* d_out, int64_t ldo, void* stream) {
  GCMI_CHECK_ARG(n_seg >= 1 && n_seg <= kMaxSeg, "seg_gemm: n_seg %d outside [1,%d]", n_seg, kMaxSeg);
  GCMI_CHECK_ARG(seg_begin && seg_end, "seg_gemm: NULL segment table");
  GCMI_CHECK_ARG(n_out > 0 && ldo >= n_out && d_out, "seg_gemm: bad output");
  GCMI_CHECK_ARG((d_a1 && d_w1 && w1_off && k1 > 0 && lda1 >= k1) || (d_a1 == nullptr),
                 "seg_gemm: bad first operand");
  GCMI_CHECK_ARG((d_a2 && d_w2 && w2_off && k2 > 0 && lda2 >= k2) || (d_a2 == nullptr),
                 "seg_gemm: bad second operand");
  GCMI_CHECK_ARG(d_a1 || d_a2, "seg_gemm: no operand");
  GCMI_CHECK_ARG(d_bias == nullptr || bias_off != nullptr, "seg_gemm: bias without offsets");
  GCMI_CHECK_ARG(act >= 0 && act <= 2, "seg_gemm: act must be 0 (none), 1 (ReLU) or 2 (out += result)");
  hipStream_t sm = (hipStream_t)stream;
  const bool vec4 = (d_a1 == nullptr || (aligned16(d_a1) && lda1 % 4 == 0)) &&
                    (d_a2 == nullptr || (aligned16(d_a2) && lda2 % 4 == 0));
  static const bool use_v1 = getenv("GCMI_GEMM_V1") != nullptr;  // A/B switches for tools/kbench.py
  static const int bm_env = getenv("GCMI_GEMM_BM") ? atoi(getenv("GCMI_GEMM_BM")) : 128;
  static const int frag_env = getenv("GCMI_GEMM_FRAG") ? atoi(getenv("GCMI_GEMM_FRAG")) : 1;
  static const int diag = getenv("GCMI_GEMM_DIAG") ? atoi(getenv("GCMI_GEMM_DIAG")) : 0;
  const bool v2 = vec4 && !use_v1;
  const int nt = n_out <= 32 ? 1 : (n_out <= 64 ? 2 : 4);
  const int bm = v2 ? ((bm_env == 64 && nt >= 2) ? 64 : BM2) : BM;
  SegTable st;
  memset(&st, 0, sizeof(st));
  st.n_seg = n_seg;
  int64_t tiles = 0;
  for (int s = 0; s < kMaxSeg; ++s) {
    st.tile_start[s] = (int32_t)tiles;
    if (s < n_seg) {
      GCMI_CHECK_ARG(seg_end[s] >= seg_begin[s] && seg_begin[s] >= 0, "seg_gemm: bad segment %d", s);
      st.seg_begin[s] = seg_begin[s];
      st.seg_end[s] = seg_end[s];
      st.w1_off[s] = (d_a1 && w1_off) ? w1_off[s] : -1;
      st.w2_off[s] = (d_a2 && w2_off) ? w2_off[s] : -1;
      st.bias_off[s] = (d_bias && bias_off) ? bias_off[s] : -1;
      tiles += (seg_end[s] - seg_begin[s] + bm - 1) / bm;
    } else {
      st.w1_off[s] = st.w2_off[s] = st.bias_off[s] = -1;
    }
  }
  st.tile_start[kMaxSeg] = (int32_t)tiles;
  if (tiles == 0) return GCMI_OK;
  TimedScope ts(GCMI_K_SEG_GEMM, sm);
  // GCMI_GEMM_V3=1: fp32-accurate product on the bf16 matrix cores (gemm_split.hip).  Measured on
  // MI355X it is 20-28 % faster than the exact-fp32 MFMA kernels below on cache-warm operands
  // (tools/kbench.py) and equal to them inside the model, where both are bound by their operand
  // streams from HBM; the exact-fp32 chain stays the default.
  static const bool v3 = getenv("GCMI_GEMM_V3") && atoi(getenv("GCMI_GEMM_V3")) == 1;
  const bool exact = gemm_exact_mode();
  // GCMI_GEMM_V4=0 disables the LDS-staged split-bf16 kernel (default path)
  static const bool v4 = !(getenv("GCMI_GEMM_V4") && atoi(getenv("GCMI_GEMM_V4")) == 0);
  if (!exact && n_seg == 1 && d_a2 == nullptr && trans_w && seg_begin[0] == 0 && w1_off[0] >= 0) {
    // the task head with more than 32 outputs (head_bwd.hip)
    const int rc = head_fwd_wide(d_a1, lda1, seg_end[0], k1, d_w1 + w1_off[0],
                                 (d_bias && bias_off && bias_off[0] >= 0) ? d_bias + bias_off[0] : nullptr, n_out, act, d_out,
                                 ldo, sm);
    if (rc != GCMI_ERR_UNSUPPORTED) return rc;
  }
  if (v4 && !v3 && !exact) {
    const int rc = launch_seg_gemm4(n_seg, seg_begin, seg_end, d_a1, lda1, k1, d_w1, w1_off, d_a2, lda2, k2,
                                    d_w2, w2_off, d_bias, bias_off, n_out, trans_w, act, d_out, ldo, sm);
    if (rc != GCMI_ERR_UNSUPPORTED) return rc;
  }
  if (v3 && vec4 && !exact) {
    const int rc = launch_seg_gemm3(n_seg, seg_begin, seg_end, d_a1, lda1, k1, d_w1, w1_off, d_a2, lda2, k2,
                                    d_w2, w2_off, d_bias, bias_off, n_out, trans_w, act, d_out, ldo, sm);
    if (rc != GCMI_ERR_UNSUPPORTED) return rc;
  }
  if (v2) {
    dim3 grid((unsigned)tiles, (unsigned)((n_out + nt * 32 - 1) / (nt * 32)));
    static const int kc_env = getenv("GCMI_GEMM_KC") ? atoi(getenv("GCMI_GEMM_KC")) : 32;
#define LAUNCH_SG2(TT, NN, BB, FF)                                                                \
  do {                                                                                            \
    if (kc_env == 64)                                                                             \
      hipLaunchKernelGGL((seg_gemm2_kernel<TT, NN, BB, FF, 64>), grid, dim3(kGBlock), 0, sm, st, d_a1, \
                         lda1, k1, d_w1, d_a2, lda2, k2, d_w2, d_bias, n_out, act, d_out, ldo, diag); \
    else                                                                                          \
      hipLaunchKernelGGL((seg_gemm2_kernel<TT, NN, BB, FF, 32>), grid, dim3(kGBlock), 0, sm, st, d_a1, \
                         lda1, k1, d_w1, d_a2, lda2, k2, d_w2, d_bias, n_out, act, d_out, ldo, diag); \
  } while (0)
#define LAUNCH_SG2_T(TT)                                                                     \
  do {                                                                                       \
    if (nt == 1) { if (frag_env) LAUNCH_SG2(TT, 1, 128, true); else LAUNCH_SG2(TT, 1, 128, false); }         \
    else if (nt == 2 && bm == 128) { if (frag_env) LAUNCH_SG2(TT, 2, 128, true); else LAUNCH_SG2(TT, 2, 128, false); } \
    else if (nt == 2) { if (frag_env) LAUNCH_SG2(TT, 2, 64, true); else LAUNCH_SG2(TT, 2, 64, false); }    \
    else if (bm == 128) { if (frag_env) LAUNCH_SG2(TT, 4, 128, true); else LAUNCH_SG2(TT, 4, 128, false); } \
    else { if (frag_env) LAUNCH_SG2(TT, 4, 64, true); else LAUNCH_SG2(TT, 4, 64, false); }               \
  } while (0)
    if (trans_w) LAUNCH_SG2_T(true); else LAUNCH_SG2_T(false);
#undef LAUNCH_SG2_T
#undef LAUNCH_SG2
  } else {
    dim3 grid((unsigned)tiles, (unsigned)((n_out + BNT - 1) / BNT));
#define LAUNCH_SG(TT, VV)                                                                       \
  hipLaunchKernelGGL((seg_gemm_kernel<TT, VV>), grid, dim3(kGBlock), 0, sm, st, d_a1, lda1, k1,  \
                     d_w1, d_a2, lda2, k2, d_w2, d_bias, n_out, act, d_out, ldo)
    if (trans_w) {
      if (vec4) LAUNCH_SG(true, true); else LAUNCH_SG(true, false);
    } else {
      if (vec4) LAUNCH_SG(false, true); else LAUNCH_SG(false, false);
    }
#undef LAUNCH_SG
  }
  GCMI_CHECK_LAUNCH("seg_gemm");
  return GCMI_OK;
}

int gcmi_seg_gemm_wgrad(int32_t n_seg, const int32_t* seg_begin, const int32_t* seg_end,
                        const float* d_a, int64_t lda, int32_t k, const float* d_g, int64_t ldg,
                        int32_t n, float* d_dw, const int64_t* dw_off, float* d_dbias,
                        const int64_t* dbias_off, int32_t trans_w, void* stream) {
  GCMI_CHECK_ARG(n_seg >= 1 && n_seg <= kMaxSeg, "wgrad: n_seg %d outside [1,%d]", n_seg, kMaxSeg);
  GCMI_CHECK_ARG(seg_begin && seg_end && dw_off, "wgrad: NULL segment table");
  GCMI_CHECK_ARG(k > 0 && n > 0 && lda >= k && ldg >= n, "wgrad: bad shape");
  GCMI_CHECK_ARG(d_a && d_g && d_dw, "wgrad: NULL buffer");
  GCMI_CHECK_ARG(d_dbias == nullptr || dbias_off != nullptr, "wgrad: dbias without offsets");
  SlabTable st;
  memset(&st, 0, sizeof(st));
  st.n_seg = n_seg;
  int64_t total_rows = 0;
  for (int s = 0; s < n_seg; ++s) {
    GCMI_CHECK_ARG(seg_end[s] >= seg_begin[s] && seg_begin[s] >= 0, "wgrad: bad segment %d", s);
    total_rows += seg_end[s] - seg_begin[s];
  }
  if (total_rows == 0) return GCMI_OK;
  // slab size: about eight workgroups per CU (the row loop is latency-bound: it wants many
  // waves), but never so small that the atomic epilogue (k x n adds per workgroup into one
  // small block) outweighs the row loop
  // ... and sized so that ALL slabs are resident at once: the split-bf16 kernel holds 4 / 4 / 3 / 2 workgroups per
  // CU for K <= 32 / 64 / 96 / 128 (registers); 1 880 slabs of 640 rows ran as 1.8 (2.4) rounds of workgroups.
  const int kt = (k + 31) / 32;
  const int64_t resident = 256 * (kt <= 2 ? 4 : (kt == 3 ? 3 : 2)) - n_seg;  // every segment may add a partial slab
  // (more than four k-tiles: the split kernel runs ceil(kt / 4) chunks per slab, so slabs are that much longer)
  static const bool wg3_on = !(getenv("GCMI_WGRAD_V3") && atoi(getenv("GCMI_WGRAD_V3")) == 0);
  const int64_t chunks = (kt > 4 && wg3_on && !gemm_exact_mode()) ? (kt + 3) / 4 : 1;
  int64_t slab = (total_rows * chunks + resident - 1) / resident;
  slab = ((slab + 63) / 64) * 64;
  if (slab < 256) slab = 256;
  if (slab > 4096) slab = 4096;
  {
    // Few rows (the task head's weight gradient at a per-GPU batch of a few thousand molecules): 256-row slabs leave
    // most CUs idle.  Smaller slabs, down to a floor, until the launch has a workgroup per CU; below the floor the
    // k x n atomic adds every workgroup ends with outweigh its row loop.
    static const int floor_env = getenv("GCMI_WGRAD_MIN_SLAB") ? atoi(getenv("GCMI_WGRAD_MIN_SLAB")) : 128;
    const int64_t floor_rows = floor_env >= 64 && floor_env % 64 == 0 ? floor_env : 256;
    const int nt_all = (n + 31) / 32;
    const int64_t col_groups = nt_all >= 3 ? (nt_all + 3) / 4 : 1;
    while (slab > floor_rows && ((total_rows + slab - 1) / slab + n_seg - 1) * col_groups * chunks < 256) slab -= 64;
  }
  st.slab_rows = (int32_t)slab;
  int64_t slabs = 0;
  for (int s = 0; s < kMaxSeg; ++s) {
    st.slab_start[s] = (int32_t)slabs;
    if (s < n_seg) {
      st.seg_begin[s] = seg_begin[s];
      st.seg_end[s] = seg_end[s];
      st.dw_off[s] = dw_off[s];
      st.db_off[s] = (d_dbias && dbias_off) ? dbias_off[s] : -1;
      slabs += (seg_end[s] - seg_begin[s] + slab - 1) / slab;
    } else {
      st.dw_off[s] = st.db_off[s] = -1;
    }
  }
  st.slab_start[kMaxSeg] = (int32_t)slabs;
  hipStream_t sm = (hipStream_t)stream;
  const int NT = (n + 31) / 32;
  const int ntw = NT >= 3 ? 4 : NT;  // n-tiles per workgroup: 1, 2 or 4
  const int KT_total = (k + 31) / 32;
  dim3 grid((unsigned)slabs, (unsigned)((NT + ntw - 1) / ntw));
  TimedScope ts(GCMI_K_WGRAD, sm);
  // GCMI_WGRAD_V3=0 keeps the exact-fp32 MFMA kernel below; default: the split-bf16 form
  // (gemm_split.hip: fp32-accurate, 6 bf16 MFMAs per 16 rows instead of 8 fp32 ones)
  static const bool wg3 = !(getenv("GCMI_WGRAD_V3") && atoi(getenv("GCMI_WGRAD_V3")) == 0);
  if (wg3 && !gemm_exact_mode()) {
    const int rc = launch_wgrad3(st, (int)slabs, d_a, lda, k, d_g, ldg, n, d_dw, d_dbias, trans_w, sm);
    if (rc != GCMI_ERR_UNSUPPORTED) return rc;
  }
  for (int kt0 = 0; kt0 < KT_total; kt0 += 8) {
    const int KT = KT_total - kt0 < 8 ? KT_total - kt0 : 8;
#define LAUNCH_WG(KK)                                                                            \
  do {                                                                                           \
    const size_t lds = ntw == 4 ? 0 : (size_t)ntw * (KK * 16 * 64 + 64) * sizeof(float);         \
    if (trans_w)                                                                                 \
      hipLaunchKernelGGL((wgrad_kernel<KK, true>), grid, dim3(kGBlock), lds, sm, st, d_a, lda,   \
                         k, kt0, d_g, ldg, n, ntw, d_dw, d_dbias);                               \
    else                                                                                         \
      hipLaunchKernelGGL((wgrad_kernel<KK, false>), grid, dim3(kGBlock), lds, sm, st, d_a, lda,  \
                         k, kt0, d_g, ldg, n, ntw, d_dw, d_dbias);                               \
  } while (0)
    switch (KT) {
      case 1: LAUNCH_WG(1); break;
      case 2: LAUNCH_WG(2); break;
      case 3: LAUNCH_WG(3); break;
      case 4: LAUNCH_WG(4); break;
      case 5: LAUNCH_WG(5); break;
      case 6: LAUNCH_WG(6); break;
      case 7: LAUNCH_WG(7); break;
      default: LAUNCH_WG(8); break;
    }
#undef LAUNCH_WG
    GCMI_CHECK_LAUNCH("seg_gemm_wgrad");
  }
  return GCMI_OK;
}

int64_t gcmi_task_head_scratch_floats(void) { return kHeadImgFloats; }

int gcmi_task_head_forward(const float* d_fingerprint, int64_t ld, int64_t n_rows, int32_t k, const float* d_w,
                           const float* d_bias, int32_t n_out, float* d_img_scratch, float* d_out, int64_t ldo,
                           void* stream) {
  GCMI_CHECK_ARG(d_fingerprint && d_w && d_out && n_rows >= 0 && k > 0 && n_out > 0 && ld >= k && ldo >= n_out,
                 "task_head_forward: bad arguments");
  if (n_rows == 0) return GCMI_OK;
  hipStream_t st = (hipStream_t)stream;
  if (d_img_scratch != nullptr && !gemm_exact_mode()) {
    int rc = head_prep(d_w, n_out, d_img_scratch, st);
    if (rc == GCMI_OK) rc = head_fwd_wide(d_fingerprint, ld, n_rows, k, d_w, d_bias, n_out, 0, d_out, ldo, st, d_img_scratch);
    if (rc != GCMI_ERR_UNSUPPORTED) return rc;
  }
  const int32_t zero32 = 0, nr = (int32_t)n_rows;
  const int64_t zero64 = 0;
  return gcmi_seg_gemm(1, &zero32, &nr, d_fingerprint, ld, k, d_w, &zero64, nullptr, 0, 0, nullptr, nullptr, d_bias,
                       d_bias ? &zero64 : nullptr, n_out, 1, 0, d_out, ldo, stream);
}

int gcmi_relu_bwd(float* d_g, int64_t ldg, const float* d_y, int64_t ldy, int64_t n_rows,
                  int32_t n_feat, void* stream) {
  GCMI_CHECK_ARG(n_feat > 0 && n_rows >= 0 && ldg >= n_feat && ldy >= n_feat, "relu_bwd: bad shape");
  if (n_rows == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_g && d_y, "relu_bwd: NULL buffer");
  const int V = (vec_width(d_g, ldg, n_feat) == 4 && vec_width(d_y, ldy, n_feat) == 4) ? 4 : 1;
  const int lpr = n_feat / V;
  const int64_t slots = n_rows * lpr;
  hipStream_t sm = (hipStream_t)stream;
  if (V == 4)
    hipLaunchKernelGGL(relu_bwd_kernel<4>, dim3(grid_for(slots, 256)), dim3(256), 0, sm, d_g, ldg,
                       d_y, ldy, slots, lpr);
  else
    hipLaunchKernelGGL(relu_bwd_kernel<1>, dim3(grid_for(slots, 256)), dim3(256), 0, sm, d_g, ldg,
                       d_y, ldy, slots, lpr);
  GCMI_CHECK_LAUNCH("relu_bwd");
  return GCMI_OK;
}

}  // extern "C"

namespace gcmi {

int seg_gemm_stats(int32_t n_seg, const int32_t* seg_begin, const int32_t* seg_end, const float* d_a1, int64_t lda1,
                   int32_t k1, const float* d_w1, const int64_t* w1_off, const float* d_a2, int64_t lda2, int32_t k2,
                   const float* d_w2, const int64_t* w2_off, const float* d_bias, const int64_t* bias_off,
                   int32_t n_out, int32_t trans_w, int32_t act, float* d_out, int64_t ldo, double* d_stats,
                   bool* fused, void* stream, float* d_wimg_scratch) {
  *fused = false;
  const bool allow = g_fused_bn_stats.load(std::memory_order_relaxed) != 0;  // GCMI_OPT_FUSED_BN_STATS
  static const bool v3 = getenv("GCMI_GEMM_V3") && atoi(getenv("GCMI_GEMM_V3")) == 1;
  static const bool v4 = !(getenv("GCMI_GEMM_V4") && atoi(getenv("GCMI_GEMM_V4")) == 0);
  const bool shapes_ok = n_seg >= 1 && n_seg <= kMaxSeg && seg_begin && seg_end && n_out > 0 && ldo >= n_out && d_out &&
                         (d_a1 || d_a2) && (d_a1 == nullptr || (d_w1 && w1_off && k1 > 0 && lda1 >= k1)) &&
                         (d_a2 == nullptr || (d_w2 && w2_off && k2 > 0 && lda2 >= k2)) &&
                         (d_bias == nullptr || bias_off != nullptr) && (act == 0 || act == 1);
  if (shapes_ok && v4 && !v3 && !gemm_exact_mode()) {
    // the persistent form (fwd_fused.hip) for the shapes it covers
    hipStream_t sm = (hipStream_t)stream;
    TimedScope ts(GCMI_K_SEG_GEMM, sm);
    double* stats = (allow && d_stats) ? d_stats : nullptr;
    const int rc = fwd_fused_gemm(n_seg, seg_begin, seg_end, d_a1, lda1, k1, d_w1, w1_off, d_a2, lda2, k2, d_w2, w2_off,
                                  d_bias, bias_off, n_out, trans_w, act, d_out, ldo, stats, sm, d_wimg_scratch);
    if (rc != GCMI_ERR_UNSUPPORTED) {
      *fused = rc == GCMI_OK && stats != nullptr;
      return rc;
    }
  }
  if (allow && d_stats && shapes_ok && v4 && !v3 && !gemm_exact_mode()) {
    hipStream_t sm = (hipStream_t)stream;
    TimedScope ts(GCMI_K_SEG_GEMM, sm);
    const int rc = launch_seg_gemm4(n_seg, seg_begin, seg_end, d_a1, lda1, k1, d_w1, w1_off, d_a2, lda2, k2, d_w2,
                                    w2_off, d_bias, bias_off, n_out, trans_w, act, d_out, ldo, sm, d_stats);
    if (rc != GCMI_ERR_UNSUPPORTED) {
      *fused = rc == GCMI_OK;
      return rc;
    }
  }
  return gcmi_seg_gemm(n_seg, seg_begin, seg_end, d_a1, lda1, k1, d_w1, w1_off, d_a2, lda2, k2, d_w2, w2_off, d_bias,
                       bias_off, n_out, trans_w, act, d_out, ldo, stream);
}

}  // namespace gcmi


// ------------------------------------------------------------------ diagnostics
// Peak rate of v_mfma_f32_32x32x2_f32 from registers (no memory traffic): the ceiling the
// GEMM kernels are priced against on THIS device (tools/kbench.py --only mfma_peak).
namespace gcmi {
__global__ void __launch_bounds__(256) mfma_peak_kernel(int iters, float* out) {
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float a = (float)(threadIdx.x & 7) * 0.25f, b = (float)(threadIdx.x & 3) * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[t][i];
  if (s == 12345.678f) out[0] = s;  // keep the chain alive
}
}  // namespace gcmi

extern "C" int gcmi_diag_mfma_peak(int32_t blocks, int32_t iters, float* d_out, void* stream) {
  hipLaunchKernelGGL(gcmi::mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, d_out);
  GCMI_CHECK_LAUNCH("mfma_peak");
  return GCMI_OK;
}
