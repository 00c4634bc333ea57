// Row-segmented GEMM with fp32 accuracy on the bf16 matrix cores.
//
// The exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, gemm.hip) peaks at ~157 TFLOP/s, which for the
// shapes of this model (K = 64..152, 64..128 output columns) is the same time as streaming the
// operands from HBM: the kernel would have to sit at both ceilings at once.  Here every fp32
// operand is split into three bf16 pieces by round-to-nearest (x = x1 + x2 + x3 + r, |r| <= 2^-24 |x|,
// residuals signed and unbiased) and the product is assembled from the six piece products that
// matter,
//     a.b ~= a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1        (dropped terms <= 2^-23 |a||b|,
//                                                            the size of an fp32 rounding)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 6 bf16 MFMAs do the work of 8 fp32-MFMA
// k-steps in 3/8 of the matrix-pipe time, so the kernel is bound by HBM alone.
//
// Structure: one workgroup = 128 rows of one segment x NT*32 output columns.
//   * the segment's weight block(s) are split once per workgroup into LDS, stored [piece][column][k]
//     (k contiguous, 16-byte padded rows): a lane's B-operand for a k-step is one ds_read_b128;
//   * A is NOT staged: each lane reads the 8 consecutive floats of its own row that form its
//     MFMA fragment (2 x 16 bytes) straight from global memory, one k-step ahead, and splits them
//     in registers;
//   * operands are swapped in the MFMA so the accumulator holds out^T: lane = atom row, four
//     consecutive registers = four consecutive output columns -> 16-byte stores.
#include "common.h"
#include "split_bf16.h"

namespace gcmi {

constexpr int kS3Block = 256;
constexpr int kS3Rows = 128;
constexpr int kS3MaxSeg = 16;

struct SegTable3 {
  int32_t n_seg;
  int32_t seg_begin[kS3MaxSeg];
  int32_t seg_end[kS3MaxSeg];
  int32_t tile_start[kS3MaxSeg + 1];
  int64_t w1_off[kS3MaxSeg];  // < 0: term absent
  int64_t w2_off[kS3MaxSeg];
  int64_t bias_off[kS3MaxSeg];
};

template <typename T>
__device__ __forceinline__ T pick3(const T* a, int s) {
  T v = a[0];
#pragma unroll
  for (int k = 1; k < kS3MaxSeg; ++k) v = (s == k) ? a[k] : v;
  return v;
}

// WVEC: the weight blocks are 16-byte addressable along their contiguous dimension.
template <bool TRANS, int NT, bool WVEC>
__global__ void __launch_bounds__(kS3Block) __attribute__((amdgpu_waves_per_eu(NT <= 2 ? 3 : 2)))
seg_gemm3_kernel(SegTable3 st, int n_tiles, const float* __restrict__ a1, int64_t lda1, int k1, int k1p,
                 const float* __restrict__ w1, const float* __restrict__ a2, int64_t lda2, int k2, int k2p,
                 const float* __restrict__ w2, const float* __restrict__ bias, int n_out, int act,
                 float* __restrict__ out, int64_t ldo) {
  constexpr int NB = NT * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned short ws[];  // [3][NB][kp] bf16
  const int col0 = blockIdx.y * NB;
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int kp = k1p + k2p + 8;  // LDS row length (bf16 elements); + 8: rows start 16 bytes apart mod 128
  const int ktot = k1p + k2p;

  // this workgroup's contiguous range of 128-row tiles (tiles are ordered by segment, so the
  // weights in LDS change at most a few times per workgroup)
  const int t_begin = (int)((int64_t)blockIdx.x * n_tiles / gridDim.x);
  const int t_end = (int)((int64_t)(blockIdx.x + 1) * n_tiles / gridDim.x);
  int cur_seg = -1;
  bool on1 = false, on2 = false;
  int64_t boff = -1;
  int seg_first_tile = 0, seg_next_tile = 0, seg_begin = 0, seg_end = 0;  // of the current segment

  for (int tile = t_begin; tile < t_end; ++tile) {
    if (cur_seg < 0 || tile >= seg_next_tile) {
      int s = 0;
#pragma unroll
      for (int k = 1; k < kS3MaxSeg; ++k) s += (k < st.n_seg && tile >= st.tile_start[k]) ? 1 : 0;
      // ---- weights of segment s -> three bf16 images in LDS (zero outside the matrix)
      if (cur_seg >= 0) __syncthreads();  // everyone is done reading the previous images
      cur_seg = s;
      seg_first_tile = pick3(st.tile_start, s);
      seg_begin = pick3(st.seg_begin, s);
      seg_end = pick3(st.seg_end, s);
      seg_next_tile = seg_first_tile + (seg_end - seg_begin + kS3Rows - 1) / kS3Rows;
      const int64_t woff1 = pick3(st.w1_off, s), woff2 = pick3(st.w2_off, s);
      on1 = a1 != nullptr && w1 != nullptr && woff1 >= 0;
      on2 = a2 != nullptr && w2 != nullptr && woff2 >= 0;
      boff = pick3(st.bias_off, s);
      if constexpr (WVEC) {
        // four consecutive elements of the contiguous dimension per thread, 8 loads in flight
        const int n4 = ktot * NB / 4;
#pragma unroll 4
        for (int e = tid; e < n4; e += kS3Block) {
          int kk, j;
          if (TRANS) {  // w is n_out x K: the quad runs along k
            const int q = ktot / 4;
            j = e / q;
            kk = (e - j * q) * 4;
          } else {  // w is K x n_out: the quad runs along the columns
            const int q = NB / 4;
            kk = e / q;
            j = (e - kk * q) * 4;
          }
          const bool first = kk < k1p;
          const int kr = first ? kk : kk - k1p;
          const int K = first ? k1 : k2;
          const bool on = first ? on1 : on2;
          const float* w = first ? w1 + (on ? woff1 : 0) : w2 + (on ? woff2 : 0);
          float v[4] = {0.f, 0.f, 0.f, 0.f};
          if (TRANS) {
            if (on && col0 + j < n_out && kr < K) {  // K % 4 == 0: the quad is inside the row
              const float4 t4 = *reinterpret_cast<const float4*>(w + (int64_t)(col0 + j) * K + kr);
              v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              unsigned p1, p2, p3;
              split3(v[i], p1, p2, p3);
              ws[(0 * NB + j) * kp + kk + i] = (unsigned short)(p1 >> 16);
              ws[(1 * NB + j) * kp + kk + i] = (unsigned short)(p2 >> 16);
              ws[(2 * NB + j) * kp + kk + i] = (unsigned short)(p3 >> 16);
            }
          } else {
            if (on && kr < K && col0 + j < n_out) {  // n_out % 4 == 0: the quad is inside the row
              const float4 t4 = *reinterpret_cast<const float4*>(w + (int64_t)kr * n_out + col0 + j);
              v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              unsigned p1, p2, p3;
              split3(v[i], p1, p2, p3);
              ws[(0 * NB + j + i) * kp + kk] = (unsigned short)(p1 >> 16);
              ws[(1 * NB + j + i) * kp + kk] = (unsigned short)(p2 >> 16);
              ws[(2 * NB + j + i) * kp + kk] = (unsigned short)(p3 >> 16);
            }
          }
        }
      } else {
#pragma unroll 2
        for (int e = tid; e < ktot * NB; e += kS3Block) {
          int kk, j;
          if (TRANS) {
            j = e / ktot;
            kk = e - j * ktot;
          } else {
            kk = e / NB;
            j = e - kk * NB;
          }
          const bool first = kk < k1p;
          const int kr = first ? kk : kk - k1p;
          const int K = first ? k1 : k2;
          const bool on = first ? on1 : on2;
          float v = 0.f;
          if (on && kr < K && col0 + j < n_out) {
            const float* w = first ? w1 + woff1 : w2 + woff2;
            v = TRANS ? w[(int64_t)(col0 + j) * K + kr] : w[(int64_t)kr * n_out + col0 + j];
          }
          unsigned p1, p2, p3;
          split3(v, p1, p2, p3);
          ws[(0 * NB + j) * kp + kk] = (unsigned short)(p1 >> 16);
          ws[(1 * NB + j) * kp + kk] = (unsigned short)(p2 >> 16);
          ws[(2 * NB + j) * kp + kk] = (unsigned short)(p3 >> 16);
        }
      }
      // the segment's bias row (zeros when absent) behind the weight images
      float* bias_lds = reinterpret_cast<float*>(ws + (size_t)3 * NB * kp);
      for (int j = tid; j < NB; j += kS3Block)
        bias_lds[j] = (bias != nullptr && boff >= 0 && col0 + j < n_out) ? bias[boff + col0 + j] : 0.f;
      __syncthreads();
    }

    const int row0 = seg_begin + (tile - seg_first_tile) * kS3Rows;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const int r = row0 + wave * 32 + (lane & 31);
    const bool row_ok = r < seg_end;
    const int64_t rc = row_ok ? r : (seg_end - 1);  // clamped: loads stay in bounds, values are zeroed
    const int steps1 = on1 ? k1p / 16 : 0;
    const int steps2 = on2 ? k2p / 16 : 0;
    const int steps = steps1 + steps2;

    // One k-step (16 columns of one operand) = this lane's 8 consecutive floats.  The loads are
    // UNCONDITIONAL from clamped, always valid addresses (a guarded load makes hipcc branch around
    // it and wait vmcnt(0) on the spot, which would serialise the prefetch); what lies outside
    // the matrix is zeroed by selects when the step is consumed.
    auto fetch = [&](int step, float4& lo, float4& hi) {
      const int sc = step < steps ? step : 0;  // an odd step count is padded with a masked step
      const bool first = sc < steps1;
      const float* a = first ? a1 : a2;
      const int64_t lda = first ? lda1 : lda2;
      const int kk = (first ? sc : sc - steps1) * 16 + 8 * half;
      const float* row = a + rc * lda;
      lo = *reinterpret_cast<const float4*>(row + (kk + 4 <= lda ? kk : 0));
      hi = *reinterpret_cast<const float4*>(row + (kk + 8 <= lda ? kk + 4 : 0));
    };
    auto consume = [&](int step, const float4& lo, const float4& hi) {
      if (step >= steps) return;  // uniform
      const bool first = step < steps1;
      const int K = first ? k1 : k2;
      const int kk = (first ? step : step - steps1) * 16 + 8 * half;
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (row_ok && kk + i < K) ? v[i] : 0.f;
      const Frag3 fa = split_frag(v);
      // LDS column of this step: the second operand's rows follow the first's, whether or not
      // the first operand is present for this segment
      const int kl = (first ? step * 16 : k1p + (step - steps1) * 16) + 8 * half;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t > 0) __builtin_amdgcn_sched_barrier(0);
        const int j = t * 32 + (lane & 31);
        const u32x4 b1 = *reinterpret_cast<const u32x4*>(&ws[(0 * NB + j) * kp + kl]);
        const u32x4 b2 = *reinterpret_cast<const u32x4*>(&ws[(1 * NB + j) * kp + kl]);
        const u32x4 b3 = *reinterpret_cast<const u32x4*>(&ws[(2 * NB + j) * kp + kl]);
        // small terms first
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(b3), as_bf16x8(fa.p[0]), acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(b1), as_bf16x8(fa.p[2]), acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(b2), as_bf16x8(fa.p[1]), acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(b2), as_bf16x8(fa.p[0]), acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(b1), as_bf16x8(fa.p[1]), acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(b1), as_bf16x8(fa.p[0]), acc[t], 0, 0, 0);
      }
    };

    // k-steps in pairs, the next pair's loads in flight while this pair runs on the matrix pipe
    float4 l0, h0, l1, h1, m0, g0, m1, g1;
    fetch(0, l0, h0);
    fetch(1, l1, h1);
#pragma unroll 1
    for (int step = 0; step < steps; step += 2) {
      fetch(step + 2, m0, g0);
      fetch(step + 3, m1, g1);
      __builtin_amdgcn_sched_barrier(0);  // keep the two steps apart: interleaving them doubles
      consume(step, l0, h0);              // the live fragment registers and spills
      __builtin_amdgcn_sched_barrier(0);
      consume(step + 1, l1, h1);
      __builtin_amdgcn_sched_barrier(0);
      l0 = m0; h0 = g0; l1 = m1; h1 = g1;
    }

    // ---- epilogue: bias, activation / accumulate, 16-byte stores (lane = row).  n_out % 4 == 0 and
    // 16-byte addressable output rows are preconditions of this kernel (checked by the launcher).
    if (row_ok) {
      float* orow = out + (int64_t)r * ldo + col0;
      const float4* bias4 = reinterpret_cast<const float4*>(ws + (size_t)3 * NB * kp);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int cl = t * 32 + 8 * rg + 4 * half;  // columns cl .. cl+3 = registers 4rg .. 4rg+3
          if (col0 + cl < n_out) {
            const float4 bq = bias4[cl >> 2];
            float4 v = make_float4(acc[t][4 * rg] + bq.x, acc[t][4 * rg + 1] + bq.y, acc[t][4 * rg + 2] + bq.z,
                                   acc[t][4 * rg + 3] + bq.w);
            if (act == 1) {
              v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
              v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
            }
            if (act == 2) {
              const float4 o = *reinterpret_cast<const float4*>(orow + cl);
              v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *reinterpret_cast<float4*>(orow + cl) = v;
          }
        }
      }
    }
  }
}

// Host side: returns GCMI_ERR_UNSUPPORTED when this kernel does not cover the shape (the caller
// falls back to the exact-fp32 MFMA kernels of gemm.hip).
int launch_seg_gemm3(int32_t n_seg, const int32_t* seg_begin, const int32_t* seg_end, const float* d_a1,
                     int64_t lda1, int32_t k1, const float* d_w1, const int64_t* w1_off, const float* d_a2,
                     int64_t lda2, int32_t k2, const float* d_w2, const int64_t* w2_off, const float* d_bias,
                     const int64_t* bias_off, int32_t n_out, int32_t trans_w, int32_t act, float* d_out,
                     int64_t ldo, hipStream_t sm) {
  if (n_seg > kS3MaxSeg) return GCMI_ERR_UNSUPPORTED;
  if ((d_a1 && (!aligned16(d_a1) || lda1 % 4)) || (d_a2 && (!aligned16(d_a2) || lda2 % 4)))
    return GCMI_ERR_UNSUPPORTED;
  if (n_out % 4 || ldo % 4 || !aligned16(d_out)) return GCMI_ERR_UNSUPPORTED;  // 16-byte stores only
  const int k1p = d_a1 ? (k1 + 15) / 16 * 16 : 0;
  const int k2p = d_a2 ? (k2 + 15) / 16 * 16 : 0;
  const int kp = k1p + k2p + 8;
  // widest column group whose three bf16 images fit LDS twice per CU
  int nt = n_out <= 32 ? 1 : (n_out <= 64 ? 2 : 4);
  while (nt > 1 && (size_t)3 * nt * 32 * kp * 2 > 72 * 1024) nt /= 2;
  const size_t shmem = (size_t)3 * nt * 32 * kp * 2 + (size_t)nt * 32 * 4;  // images + bias row
  if (shmem > 150 * 1024) return GCMI_ERR_UNSUPPORTED;
  SegTable3 st;
  memset(&st, 0, sizeof(st));
  st.n_seg = n_seg;
  int64_t tiles = 0;
  for (int s = 0; s < kS3MaxSeg; ++s) {
    st.tile_start[s] = (int32_t)tiles;
    if (s < n_seg) {
      st.seg_begin[s] = seg_begin[s];
      st.seg_end[s] = seg_end[s];
      st.w1_off[s] = (d_a1 && w1_off) ? w1_off[s] : -1;
      st.w2_off[s] = (d_a2 && w2_off) ? w2_off[s] : -1;
      st.bias_off[s] = (d_bias && bias_off) ? bias_off[s] : -1;
      tiles += (seg_end[s] - seg_begin[s] + kS3Rows - 1) / kS3Rows;
    } else {
      st.w1_off[s] = st.w2_off[s] = st.bias_off[s] = -1;
    }
  }
  st.tile_start[kS3MaxSeg] = (int32_t)tiles;
  if (tiles == 0) return GCMI_OK;
  // weights 16-byte addressable along their contiguous dimension?
  bool wvec = trans_w ? ((d_a1 == nullptr || k1 % 4 == 0) && (d_a2 == nullptr || k2 % 4 == 0)) : (n_out % 4 == 0);
  for (int s = 0; s < n_seg && wvec; ++s) {
    if (st.w1_off[s] >= 0 && !aligned16(d_w1 + st.w1_off[s])) wvec = false;
    if (st.w2_off[s] >= 0 && !aligned16(d_w2 + st.w2_off[s])) wvec = false;
  }
  // persistent workgroups over contiguous tile ranges: the weight images are rebuilt only when a
  // range crosses a segment boundary
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(160 * 1024) / (shmem + 256)));
  const int col_groups = (n_out + nt * 32 - 1) / (nt * 32);
  const int gx = (int)std::min<int64_t>(tiles, std::max(1, 256 * per_cu / col_groups));
  dim3 grid((unsigned)gx, (unsigned)col_groups);
  const int n_tiles = (int)tiles;
#define LAUNCH_S3(TT, NN)                                                                              \
  do {                                                                                                 \
    if (wvec) LAUNCH_S3W(TT, NN, true); else LAUNCH_S3W(TT, NN, false);                                \
  } while (0)
#define LAUNCH_S3W(TT, NN, WW)                                                                         \
  do {                                                                                                 \
    auto kern = seg_gemm3_kernel<TT, NN, WW>;                                                            \
    static bool attr_done = false;                                                                     \
    if (!attr_done) {                                                                                  \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { \
        (void)hipGetLastError();                                                                       \
        return GCMI_ERR_UNSUPPORTED;                                                                   \
      }                                                                                                \
      attr_done = true;                                                                                \
    }                                                                                                  \
    hipLaunchKernelGGL(kern, grid, dim3(kS3Block), shmem, sm, st, n_tiles, d_a1, lda1, k1, k1p, d_w1, d_a2, lda2, k2,  \
                       k2p, d_w2, d_bias, n_out, act, d_out, ldo);                                     \
  } while (0)
  if (trans_w) {
    if (nt == 1) LAUNCH_S3(true, 1); else if (nt == 2) LAUNCH_S3(true, 2); else LAUNCH_S3(true, 4);
  } else {
    if (nt == 1) LAUNCH_S3(false, 1); else if (nt == 2) LAUNCH_S3(false, 2); else LAUNCH_S3(false, 4);
  }
#undef LAUNCH_S3
#undef LAUNCH_S3W
  GCMI_CHECK_LAUNCH("seg_gemm3");
  return GCMI_OK;
}


// ------------------------------------------------------------------ forward / dgrad, LDS-staged
// Same product as seg_gemm3_kernel, different operand path: A K-chunks of 32 columns are fetched
// with coalesced 16-byte loads (8 lanes = one 128-byte row piece) into registers one chunk ahead
// and staged through LDS as fp32; the matching weight chunk is split into its three bf16 images
// when it is written to LDS.  Inside the model the operands stream from HBM, where this access
// shape moves ~1.5x the bytes per second of the per-lane fragment loads of seg_gemm3_kernel.
constexpr int kS4KC = 32;          // K chunk
constexpr int kS4AStride = kS4KC + 4;   // floats per staged A row: 16-byte aligned, 2-way b128 reads
constexpr int kS4WStride = kS4KC + 8;   // bf16 per staged W column: conflict-free b128 reads

// AVEC: the A rows are 16-byte addressable (else four scalar loads per quad: same values, same
// arithmetic, so the result does not depend on how the caller laid its rows out)
template <bool TRANS, int NT, bool AVEC>
__global__ void __launch_bounds__(kS3Block) __attribute__((amdgpu_waves_per_eu(NT <= 2 ? 3 : 2)))
seg_gemm4_kernel(SegTable3 st, const float* __restrict__ a1, int64_t lda1, int k1, const float* __restrict__ w1,
                 const float* __restrict__ a2, int64_t lda2, int k2, const float* __restrict__ w2,
                 const float* __restrict__ bias, int n_out, int act, float* __restrict__ out, int64_t ldo,
                 double* __restrict__ stats, int rev) {
  constexpr int NB = NT * 32;
  constexpr int APASS = 4;                       // 128 rows / (256 threads / 8 lanes per row piece)
  constexpr int BPASS = (kS4KC * NB) / kS3Block;  // weight elements per thread per chunk
  // one raw LDS block: the staged operands during the K loop, the output tile of the statistics epilogue after it
  constexpr int HC = NB < 64 ? NB : 64;           // columns summed per pass of the statistics epilogue
  constexpr int TP = HC + 4;                      // tile pitch in floats: 16-byte rows, column reads hit 64 different banks
  constexpr size_t kOperandBytes = sizeof(float) * kS3Rows * kS4AStride + sizeof(unsigned short) * 3 * NB * kS4WStride;
  constexpr size_t kTileBytes = sizeof(float) * 4 * 32 * TP;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[kOperandBytes > kTileBytes ? kOperandBytes : kTileBytes];
  float (*As)[kS4AStride] = reinterpret_cast<float (*)[kS4AStride]>(lds_raw);
  unsigned short (*Ws)[NB][kS4WStride] =
      reinterpret_cast<unsigned short (*)[NB][kS4WStride]>(lds_raw + sizeof(float) * kS3Rows * kS4AStride);
  __shared__ __attribute__((aligned(16))) float bias_lds[NB];
  __shared__ double col_part[4][2][NB];           // per-wave column sums / sums of squares
  const int b = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;  // last-written rows first on alternate launches
  int s = 0;
#pragma unroll
  for (int k = 1; k < kS3MaxSeg; ++k) s += (k < st.n_seg && b >= st.tile_start[k]) ? 1 : 0;
  const int row0 = pick3(st.seg_begin, s) + (b - pick3(st.tile_start, s)) * kS3Rows;
  const int seg_end = pick3(st.seg_end, s);
  const int rows_valid = (seg_end - row0 < kS3Rows) ? seg_end - row0 : kS3Rows;
  const int col0 = blockIdx.y * NB;
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int64_t woff1 = pick3(st.w1_off, s), woff2 = pick3(st.w2_off, s);
  const int64_t boff = pick3(st.bias_off, s);
  const bool on1 = a1 != nullptr && w1 != nullptr && woff1 >= 0;
  const bool on2 = a2 != nullptr && w2 != nullptr && woff2 >= 0;
  const int n1 = on1 ? (k1 + kS4KC - 1) / kS4KC : 0;
  const int n2 = on2 ? (k2 + kS4KC - 1) / kS4KC : 0;
  const int nchunks = n1 + n2;
  for (int j = tid; j < NB; j += kS3Block)
    bias_lds[j] = (bias != nullptr && boff >= 0 && col0 + j < n_out) ? bias[boff + col0 + j] : 0.f;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // prefetch registers: unconditional loads from clamped addresses, masked when stored to LDS
  float4 ra[APASS];
  float rb[BPASS];
  int pend_k0 = 0, pend_K = 0;
  auto gload = [&](int c) {
    const bool first = c < n1;
    const float* a = first ? a1 : a2;
    const int64_t lda = first ? lda1 : lda2;
    const int K = first ? k1 : k2;
    const float* w = first ? w1 + woff1 : w2 + woff2;
    const int k0 = (first ? c : c - n1) * kS4KC;
    pend_k0 = k0;
    pend_K = K;
    const int kcol = k0 + (tid & 7) * 4;
    const int kc = kcol + 4 <= lda ? kcol : 0;
#pragma unroll
    for (int pass = 0; pass < APASS; ++pass) {
      const int r = (tid >> 3) + pass * 32;
      const int rc = r < rows_valid ? r : rows_valid - 1;
      const float* p = a + (int64_t)(row0 + rc) * lda;
      if constexpr (AVEC) {
        ra[pass] = *reinterpret_cast<const float4*>(p + kc);
      } else {  // clamped scalar loads; columns >= K are zeroed when the quad goes to LDS
        ra[pass].x = p[kcol + 0 < K ? kcol + 0 : 0];
        ra[pass].y = p[kcol + 1 < K ? kcol + 1 : 0];
        ra[pass].z = p[kcol + 2 < K ? kcol + 2 : 0];
        ra[pass].w = p[kcol + 3 < K ? kcol + 3 : 0];
      }
    }
    if constexpr (!TRANS) {  // w is K x n_out
      const int j = col0 + tid % NB;
      const int jc = j < n_out ? j : n_out - 1;
#pragma unroll
      for (int pass = 0; pass < BPASS; ++pass) {
        const int kk = k0 + tid / NB + pass * (kS3Block / NB);
        rb[pass] = w[(int64_t)(kk < K ? kk : K - 1) * n_out + jc];
      }
    } else {  // w is n_out x K
      const int kk = k0 + (tid % kS4KC);
      const int kkc = kk < K ? kk : K - 1;
#pragma unroll
      for (int pass = 0; pass < BPASS; ++pass) {
        const int j = col0 + tid / kS4KC + pass * (kS3Block / kS4KC);
        rb[pass] = w[(int64_t)(j < n_out ? j : n_out - 1) * K + kkc];
      }
    }
  };
  auto sstore = [&]() {
    const int kq = (tid & 7) * 4;
    const int tail = pend_K - (pend_k0 + kq);
#pragma unroll
    for (int pass = 0; pass < APASS; ++pass) {
      const int r = (tid >> 3) + pass * 32;
      const bool ok = r < rows_valid;
      float4 v = ra[pass];
      v.x = (ok && tail > 0) ? v.x : 0.f;
      v.y = (ok && tail > 1) ? v.y : 0.f;
      v.z = (ok && tail > 2) ? v.z : 0.f;
      v.w = (ok && tail > 3) ? v.w : 0.f;
      *reinterpret_cast<float4*>(&As[r][kq]) = v;
    }
#pragma unroll
    for (int pass = 0; pass < BPASS; ++pass) {
      int kk, j;
      if constexpr (!TRANS) {
        j = tid % NB;
        kk = tid / NB + pass * (kS3Block / NB);
      } else {
        kk = tid % kS4KC;
        j = tid / kS4KC + pass * (kS3Block / kS4KC);
      }
      const float v = (pend_k0 + kk < pend_K && col0 + j < n_out) ? rb[pass] : 0.f;
      unsigned p1, p2, p3;
      split3(v, p1, p2, p3);
      Ws[0][j][kk] = (unsigned short)(p1 >> 16);
      Ws[1][j][kk] = (unsigned short)(p2 >> 16);
      Ws[2][j][kk] = (unsigned short)(p3 >> 16);
    }
  };

  if (nchunks > 0) {
    gload(0);
    sstore();
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks) gload(c + 1);  // in flight while the matrix pipe works on chunk c
#pragma unroll
      for (int step = 0; step < kS4KC / 16; ++step) {
        const int kk0 = step * 16 + 8 * half;
        const float* arow = &As[wave * 32 + (lane & 31)][kk0];
        const float4 lo = *reinterpret_cast<const float4*>(arow);
        const float4 hi = *reinterpret_cast<const float4*>(arow + 4);
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const Frag3 fa = split_frag(v);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int j = t * 32 + (lane & 31);
          const u32x4 b1 = *reinterpret_cast<const u32x4*>(&Ws[0][j][kk0]);
          const u32x4 b2 = *reinterpret_cast<const u32x4*>(&Ws[1][j][kk0]);
          const u32x4 b3 = *reinterpret_cast<const u32x4*>(&Ws[2][j][kk0]);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(b3), as_bf16x8(fa.p[0]), acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(b1), as_bf16x8(fa.p[2]), acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(b2), as_bf16x8(fa.p[1]), acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(b2), as_bf16x8(fa.p[0]), acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(b1), as_bf16x8(fa.p[1]), acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(b1), as_bf16x8(fa.p[0]), acc[t], 0, 0, 0);
        }
      }
      __syncthreads();
      if (c + 1 < nchunks) {
        sstore();
        __syncthreads();
      }
    }
  } else {
    __syncthreads();
  }

  // ---- epilogue (n_out % 4 == 0, 16-byte addressable output rows: checked by the launcher)
  // The accumulator holds out^T (lane = row, registers = columns): stored straight from registers, one instruction
  // writes 32-byte pieces of 32 different rows, i.e. four times the write transactions the bytes need (the address
  // unit stalls on them: TA_ADDR_STALLED_BY_TC 20-100x that of the other streaming kernels).  Every wave therefore
  // lays its 32 x HC tile out in the LDS the K loop has finished with (pitch HC + 4: 16-byte rows, conflict-free
  // column reads) and writes whole rows: a lane group of HC/4 lanes per row, 16 bytes per lane.  On the way, for the
  // training forward, a lane adds one column over the 32 rows in fp64 (BatchNorm statistics of the values just
  // written, so that the layer output is not read again for them); the four waves' partials meet in LDS and one
  // fp64 atomic per column and workgroup goes to the replicated accumulators bn_finalize_kernel reads.
  typedef float f32x4e __attribute__((ext_vector_type(4)));
  const int r = wave * 32 + (lane & 31);
  {
    float* T = reinterpret_cast<float*>(lds_raw) + wave * 32 * TP;
    const int rl = lane & 31;
    const bool row_ok = r < rows_valid;
    const f32x4e* bias4 = reinterpret_cast<const f32x4e*>(bias_lds);
    constexpr int QPR = HC / 4;        // lanes per output row
    constexpr int RPI = 64 / QPR;      // rows per store instruction
    const int srow = lane / QPR, sq = lane - srow * QPR;
#pragma unroll
    for (int h = 0; h < (NT + 1) / 2; ++h) {
#pragma unroll
      for (int tt = 0; tt < (NT >= 2 ? 2 : 1); ++tt) {
        const int t = 2 * h + tt;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int cl = t * 32 + 8 * rg + 4 * half;
          const f32x4e bq = bias4[cl >> 2];
          f32x4e v = {acc[t][4 * rg] + bq.x, acc[t][4 * rg + 1] + bq.y, acc[t][4 * rg + 2] + bq.z,
                      acc[t][4 * rg + 3] + bq.w};
          if (act == 1) {
            v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
            v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
          }
          if (!row_ok || col0 + cl >= n_out) v = f32x4e{0.f, 0.f, 0.f, 0.f};  // n_out % 4 == 0: whole quads
          *reinterpret_cast<f32x4e*>(T + rl * TP + (cl - 64 * h)) = v;
        }
      }
      // the tile is this wave's own: no workgroup barrier between its writes and its reads
      if (stats != nullptr && lane < HC) {  // `stats` is uniform
        double s1 = 0.0, s2 = 0.0;  // fp64 like the stand-alone column-sum kernel: the statistics then differ
#pragma unroll 8               // from it by summation order only (~1e-16), not by fp32 rounding (~1e-7)
        for (int rr = 0; rr < 32; ++rr) {
          const double x = (double)T[rr * TP + lane];
          s1 += x;
          s2 += x * x;
        }
        col_part[wave][0][64 * h + lane] = s1;
        col_part[wave][1][64 * h + lane] = s2;
      }
      // whole rows out: RPI rows per instruction
      const int cg = col0 + 64 * h + 4 * sq;
      constexpr int NI = 32 / RPI;           // store instructions per wave and pass
      constexpr int GI = NI < 4 ? NI : 4;    // ... taken four at a time (old pieces requested together)
#pragma unroll
      for (int g0 = 0; g0 < NI; g0 += GI) {
        f32x4e oldv[GI];
        if (act == 2) {
#pragma unroll
          for (int i = 0; i < GI; ++i) {
            const int rowl = (g0 + i) * RPI + srow;
            const bool ok = wave * 32 + rowl < rows_valid && cg < n_out;
            const int64_t rr = row0 + (wave * 32 + rowl < rows_valid ? wave * 32 + rowl : 0);
            oldv[i] = *reinterpret_cast<const f32x4e*>(out + rr * ldo + (ok ? cg : col0));
          }
        }
#pragma unroll
        for (int i = 0; i < GI; ++i) {
          const int rowl = (g0 + i) * RPI + srow;
          if (wave * 32 + rowl < rows_valid && cg < n_out) {
            f32x4e v = *reinterpret_cast<const f32x4e*>(T + rowl * TP + 4 * sq);
            if (act == 2) v += oldv[i];
            *reinterpret_cast<f32x4e*>(out + (int64_t)(row0 + wave * 32 + rowl) * ldo + cg) = v;
          }
        }
      }
    }
  }
  if (stats != nullptr) {  // uniform
    __syncthreads();
    for (int e = tid; e < 2 * NB; e += kS3Block) {
      const int which = e / NB, c = e - which * NB;
      if (col0 + c < n_out) {
        const double tot = col_part[0][which][c] + col_part[1][which][c] + col_part[2][which][c] + col_part[3][which][c];
        atomicAdd(stats + (size_t)2 * n_out * (1 + (blockIdx.x % kBnReplicas)) + (size_t)which * n_out + col0 + c, tot);
      }
    }
  }
}

int launch_seg_gemm4(int32_t n_seg, const int32_t* seg_begin, const int32_t* seg_end, const float* d_a1,
                     int64_t lda1, int32_t k1, const float* d_w1, const int64_t* w1_off, const float* d_a2,
                     int64_t lda2, int32_t k2, const float* d_w2, const int64_t* w2_off, const float* d_bias,
                     const int64_t* bias_off, int32_t n_out, int32_t trans_w, int32_t act, float* d_out,
                     int64_t ldo, hipStream_t sm, double* d_stats) {
  if (n_seg > kS3MaxSeg) return GCMI_ERR_UNSUPPORTED;
  if (d_stats && act == 2) return GCMI_ERR_UNSUPPORTED;
  const bool avec = !((d_a1 && (!aligned16(d_a1) || lda1 % 4)) || (d_a2 && (!aligned16(d_a2) || lda2 % 4)));
  if (n_out % 4 || ldo % 4 || !aligned16(d_out)) return GCMI_ERR_UNSUPPORTED;
  int nt = n_out <= 32 ? 1 : (n_out <= 64 ? 2 : 4);
  SegTable3 st;
  memset(&st, 0, sizeof(st));
  st.n_seg = n_seg;
  int64_t tiles = 0;
  for (int s = 0; s < kS3MaxSeg; ++s) {
    st.tile_start[s] = (int32_t)tiles;
    if (s < n_seg) {
      st.seg_begin[s] = seg_begin[s];
      st.seg_end[s] = seg_end[s];
      st.w1_off[s] = (d_a1 && w1_off) ? w1_off[s] : -1;
      st.w2_off[s] = (d_a2 && w2_off) ? w2_off[s] : -1;
      st.bias_off[s] = (d_bias && bias_off) ? bias_off[s] : -1;
      tiles += (seg_end[s] - seg_begin[s] + kS3Rows - 1) / kS3Rows;
    } else {
      st.w1_off[s] = st.w2_off[s] = st.bias_off[s] = -1;
    }
  }
  st.tile_start[kS3MaxSeg] = (int32_t)tiles;
  if (tiles == 0) return GCMI_OK;
  // Few row tiles (a per-molecule product: the task heads at a per-GPU batch of a few thousand molecules) leave most
  // of the 256 CUs idle at 128 rows x 128 columns per workgroup: narrower column groups then, until the launch has a
  // workgroup per CU.  The operand tile is read (from L2) and split once more per extra group; every output element
  // is the same sum in the same order whatever the group width.
  while (nt > 1 && tiles * ((n_out + nt * 32 - 1) / (nt * 32)) < 256) nt >>= 1;
  dim3 grid((unsigned)tiles, (unsigned)((n_out + nt * 32 - 1) / (nt * 32)));
  const int rev = next_sweep_direction();
#define LAUNCH_S4(TT, NN)                                                                                      \
  do {                                                                                                         \
    if (avec)                                                                                                  \
      hipLaunchKernelGGL((seg_gemm4_kernel<TT, NN, true>), grid, dim3(kS3Block), 0, sm, st, d_a1, lda1, k1, d_w1,   \
                         d_a2, lda2, k2, d_w2, d_bias, n_out, act, d_out, ldo, d_stats, rev);                  \
    else                                                                                                       \
      hipLaunchKernelGGL((seg_gemm4_kernel<TT, NN, false>), grid, dim3(kS3Block), 0, sm, st, d_a1, lda1, k1, d_w1,  \
                         d_a2, lda2, k2, d_w2, d_bias, n_out, act, d_out, ldo, d_stats, rev);                  \
  } while (0)
  if (trans_w) {
    if (nt == 1) LAUNCH_S4(true, 1); else if (nt == 2) LAUNCH_S4(true, 2); else LAUNCH_S4(true, 4);
  } else {
    if (nt == 1) LAUNCH_S4(false, 1); else if (nt == 2) LAUNCH_S4(false, 2); else LAUNCH_S4(false, 4);
  }
#undef LAUNCH_S4
  GCMI_CHECK_LAUNCH("seg_gemm4");
  return GCMI_OK;
}

// ------------------------------------------------------------------ weight gradient, split-bf16
// dW[s] += a[rows_s]^T . g[rows_s] with the same exact three-way operand split: the contraction
// runs over ROWS, so a lane's MFMA fragment is 8 consecutive rows of one feature column (8 dword
// loads, coalesced across the 32 columns of the tile), split and packed in registers.  Structure as
// wgrad_kernel (gemm.hip): a workgroup = one row slab x `ntw` 32-column tiles of g, every wave one
// g tile x all KT a tiles, row parts combined through LDS, one float atomic per dW element.
template <int KT, bool TRANS>
__global__ void __launch_bounds__(kS3Block)
wgrad3_kernel(SlabTable st, const float* __restrict__ a, int64_t lda, int k, const float* __restrict__ g,
              int64_t ldg, int n, int ntw, float* __restrict__ dw, float* __restrict__ dbias, int rev) {
  const int b = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;  // last-written rows first on alternate launches
  int s = 0;
#pragma unroll
  for (int q = 1; q < kMaxSegW; ++q) s += (q < st.n_seg && b >= st.slab_start[q]) ? 1 : 0;
  const int seg_end = pick3(st.seg_end, s);
  const int slab0 = pick3(st.seg_begin, s) + (b - pick3(st.slab_start, s)) * st.slab_rows;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5;
  const int nt = blockIdx.y * ntw + (wave % ntw);
  const int rp = wave / ntw;
  const int parts = 4 / ntw;
  const int part_rows = st.slab_rows / parts;  // a multiple of 16
  const int r_begin = slab0 + rp * part_rows;
  int r_end = r_begin + part_rows;
  if (r_end > seg_end) r_end = seg_end;
  const int ncol = nt * 32 + (lane & 31);
  const bool n_ok = ncol < n;
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
    kcol[t] = blockIdx.z * (KT * 32) + t * 32 + (lane & 31);  // (grid.z: chunks of KT tiles when k > 128 columns)
    k_ok[t] = kcol[t] < k;
  }
  // 16 rows per step: this lane's rows r + 8*half .. + 7; clamped unconditional loads, zeroed by selects
  float av[2][KT][8], gv[2][8];
  auto load_set = [&](int buf, int r) {
    const int r8 = r + 8 * half;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int rr = r8 + u;
      const int rc = rr < r_end ? rr : r_end - 1;
      gv[buf][u] = g[(int64_t)rc * ldg + (n_ok ? ncol : 0)];
#pragma unroll
      for (int t = 0; t < KT; ++t) av[buf][t][u] = a[(int64_t)rc * lda + (k_ok[t] ? kcol[t] : 0)];
    }
  };
  auto mma_set = [&](int buf, int r) {
    const int r8 = r + 8 * half;
    float gq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      gq[u] = (n_ok && r8 + u < r_end) ? gv[buf][u] : 0.f;
      bsum += gq[u];
    }
    const Frag3 fg = split_frag(gq);
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      float aq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) aq[u] = (k_ok[t] && r8 + u < r_end) ? av[buf][t][u] : 0.f;
      const Frag3 fa = split_frag(aq);
      // TRANS (nn.Linear layout, dW stored n x k): operands swapped so the accumulator holds dW^T
      // and its lane index is the contiguous index of the destination (see wgrad_kernel)
      const Frag3& L = TRANS ? fg : fa;
      const Frag3& R = TRANS ? fa : fg;
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[2]), as_bf16x8(R.p[0]), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[0]), as_bf16x8(R.p[2]), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[1]), as_bf16x8(R.p[1]), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[1]), as_bf16x8(R.p[0]), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[0]), as_bf16x8(R.p[1]), acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[0]), as_bf16x8(R.p[0]), acc[t], 0, 0, 0);
    }
  };
  if (r_begin < r_end) {
    load_set(0, r_begin);
    int r = r_begin;
    while (true) {
      const int r1 = r + 16;
      if (r1 < r_end) load_set(1, r1);
      mma_set(0, r);
      if (r1 >= r_end) break;
      const int r2 = r1 + 16;
      if (r2 < r_end) load_set(0, r2);
      mma_set(1, r1);
      if (r2 >= r_end) break;
      r = r2;
    }
  }
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
  const int64_t woff = pick3(st.dw_off, s);
  if (woff >= 0) {
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int rix = (reg & 3) + 8 * (reg >> 2) + 4 * half;
        if constexpr (TRANS) {
          const int nc = nt * 32 + rix;
          if (nc < n && k_ok[t]) atomicAdd(dw + woff + (int64_t)nc * k + kcol[t], acc[t][reg]);
        } else {
          const int kf = blockIdx.z * (KT * 32) + t * 32 + rix;
          if (kf < k && n_ok) atomicAdd(dw + woff + (int64_t)kf * n + ncol, acc[t][reg]);
        }
      }
    }
  }
  if (dbias != nullptr && blockIdx.z == 0) {
    const int64_t boff = pick3(st.db_off, s);
    bsum += __shfl_xor(bsum, 32);
    if (half == 0 && n_ok && boff >= 0) atomicAdd(dbias + boff + ncol, bsum);
  }
}

int launch_wgrad3(const SlabTable& st, int slabs, const float* d_a, int64_t lda, int k, const float* d_g, int64_t ldg,
                  int n, float* d_dw, float* d_dbias, int trans_w, hipStream_t sm) {
  const int KT_all = (k + 31) / 32;
  if (st.slab_rows % 64) return GCMI_ERR_UNSUPPORTED;
  // wider K (EdgeNetwork's 900 moment columns): chunks of four tiles in grid.z, every chunk re-reading g.  (They used
  // to fall to the fp32 kernel's k-passes: four launches of 229 us each for 73 000 rows x 900 x 100.)
  const int KT = KT_all > 4 ? 4 : KT_all;
  const int chunks = (KT_all + KT - 1) / KT;
  const int NT = (n + 31) / 32;
  const int ntw = NT >= 3 ? 4 : NT;
  dim3 grid((unsigned)slabs, (unsigned)((NT + ntw - 1) / ntw), (unsigned)chunks);
  const int rev = next_sweep_direction();
#define LAUNCH_W3(KK)                                                                               \
  do {                                                                                              \
    const size_t lds = ntw == 4 ? 0 : (size_t)ntw * (KK * 16 * 64 + 64) * sizeof(float);            \
    if (trans_w)                                                                                    \
      hipLaunchKernelGGL((wgrad3_kernel<KK, true>), grid, dim3(kS3Block), lds, sm, st, d_a, lda, k, d_g, ldg, n, \
                         ntw, d_dw, d_dbias, rev);                                                  \
    else                                                                                            \
      hipLaunchKernelGGL((wgrad3_kernel<KK, false>), grid, dim3(kS3Block), lds, sm, st, d_a, lda, k, d_g, ldg, n, \
                         ntw, d_dw, d_dbias, rev);                                                  \
  } while (0)
  switch (KT) {
    case 1: LAUNCH_W3(1); break;
    case 2: LAUNCH_W3(2); break;
    case 3: LAUNCH_W3(3); break;
    default: LAUNCH_W3(4); break;
  }
#undef LAUNCH_W3
  GCMI_CHECK_LAUNCH("seg_gemm_wgrad3");
  return GCMI_OK;
}

}  // namespace gcmi
