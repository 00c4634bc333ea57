// Everything the backward does per MOLECULE in one kernel: loss and d loss / d logits (loss.hip: SoftmaxCrossEntropy /
// L2Loss through _StandardLoss, models/losses.py:251-259, :85-94, torch_model.py:1275-1294), the task head's weight
// and bias gradient, its input gradient, the tanh derivative of GraphGather's activation (layers.py:6472-6479), and
// the column sums the BatchNorm backward of the dense layer takes from per-molecule data (bn.hip,
// readout_bn_sums_kernel).  Separately these were six launches over B x 256 floats (loss 27 us, head dW 41, head dX 28,
// tanh' 33, BatchNorm sums 54, at 65 536 molecules) that each re-read the fingerprint or the gradient rows.
//
// A workgroup of 256 threads = the 256 fingerprint columns walks a contiguous range of molecules, 32 at a time:
// phase 1 computes d logits of the 32 molecules (one (molecule, task) per thread and round) into LDS, phase 2 lets
// thread k read fingerprint[b][k] once and form, from its column of the head matrix held in registers, the input
// gradient, its share of the weight gradient (24..32 register accumulators, one atomic per element at the end) and
// its share of the BatchNorm sums (fp64).  Plain fp32 FMA chains: no matrix cores, no operand splitting.
#include <math.h>

#include "common.h"
#include "split_bf16.h"

namespace gcmi {

constexpr int kHB = 256;   // threads = fingerprint columns (2 x dense width)
constexpr int kHM = 32;    // molecules per round
constexpr int kHT = 32;    // padded tasks x classes

struct HeadArgs {
  int32_t kind, n_tasks, n_classes, tc;
  int64_t n_rows, n_mols;
  float inv_count;
  const float* logits;
  const float* labels;
  const float* weights;
  const float* fp;
  int64_t ldfp;
  const float* w;        // tc x 256 (nn.Linear)
  float* dw;
  float* db;
  float* g2;             // n_mols x ldg2: gradient w.r.t. GraphGather's pre-activation [dsum | dmax]
  int64_t ldg2;
  double* loss_acc;
  // BatchNorm sums (sums == nullptr: skipped)
  const int32_t* runs;
  int32_t n_deg;
  const int32_t* arg;
  const float* rawsum;
  const float* mean;
  const float* invstd;
  double* sums;
};

__global__ void __launch_bounds__(kHB) head_bwd_kernel(HeadArgs a) {
  __shared__ __attribute__((aligned(16))) float dl_s[kHM][kHT];
  __shared__ int n_s[kHM];
  __shared__ double red[2][kHB];
  const int k = threadIdx.x;
  const int D = kHB / 2;
  const int f = k & (D - 1), part = k >> 7;
  const int TC = a.tc;
  float wk[kHT], dwacc[kHT];
#pragma unroll
  for (int t = 0; t < kHT; ++t) {
    wk[t] = t < TC ? a.w[(int64_t)t * kHB + k] : 0.f;
    dwacc[t] = 0.f;
  }
  float dbacc = 0.f;
  double t1 = 0.0, t2 = 0.0, loss_local = 0.0;
  const double mu = a.sums ? (double)a.mean[f] : 0.0, is = a.sums ? (double)a.invstd[f] : 0.0;
  for (int i = k; i < kHM * kHT; i += kHB) (&dl_s[0][0])[i] = 0.f;  // the padding columns stay zero
  const int64_t b0 = (int64_t)blockIdx.x * a.n_mols / gridDim.x;
  const int64_t b1 = (int64_t)(blockIdx.x + 1) * a.n_mols / gridDim.x;
  __syncthreads();
  for (int64_t c0 = b0; c0 < b1; c0 += kHM) {
    const int nm = (int)((b1 - c0) < kHM ? (b1 - c0) : kHM);
    // ---- phase 1: d logits of this round's molecules (rows beyond n_rows: padding molecules, no loss)
    for (int it = k; it < kHM * a.n_tasks; it += kHB) {
      const int m = it / a.n_tasks, t = it - m * a.n_tasks;
      const int64_t b = c0 + m;
      const bool live = m < nm && b < a.n_rows;
      const int64_t item = b * a.n_tasks + t;
      const float w = (live && a.weights) ? a.weights[item] : 1.f;
      if (a.kind == 0) {
        const int C = a.n_classes;
        if (live) {
          const float* x = a.logits + item * C;
          const float* y = a.labels + item * C;
          float mx = -INFINITY;
          for (int c = 0; c < C; ++c) mx = fmaxf(mx, x[c]);
          float se = 0.f, ysum = 0.f;
          for (int c = 0; c < C; ++c) {
            se += expf(x[c] - mx);
            ysum += y[c];
          }
          const float lse = logf(se);
          float l = 0.f;
          for (int c = 0; c < C; ++c) {
            const float logp = x[c] - mx - lse;
            const float p = expf(logp);
            l -= y[c] * logp;
            dl_s[m][t * C + c] = w * (p * ysum - y[c]) * a.inv_count;
          }
          loss_local += (double)(w * l);
        } else {
          for (int c = 0; c < C; ++c) dl_s[m][t * C + c] = 0.f;
        }
      } else {
        float d = 0.f;
        if (live) {
          const float dlt = a.logits[item] - a.labels[item];
          loss_local += (double)(w * dlt * dlt);
          d = 2.f * dlt * w * a.inv_count;
        }
        dl_s[m][t] = d;
      }
    }
    if (a.sums != nullptr && k < kHM) {
      int n = 0;
      if (k < nm) {
        const int32_t* r = a.runs + ((c0 + k) * a.n_deg) * 2;
        for (int d = 0; d < a.n_deg; ++d) n += r[2 * d + 1] - r[2 * d];
      }
      n_s[k] = n;
    }
    __syncthreads();
    // ---- phase 2: this thread's fingerprint column of every molecule of the round, eight molecules' loads in flight
    // at a time (one dependent load per molecule would run the loop at memory latency)
    constexpr int U = 8;
    for (int m0 = 0; m0 < nm; m0 += U) {
      float fv[U], rsv[U];
      int av[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t b = c0 + (m0 + u < nm ? m0 + u : nm - 1);
        fv[u] = a.fp[b * a.ldfp + k];
        if (a.sums != nullptr) {
          rsv[u] = a.rawsum[b * 2 * D + k];               // k = part * D + f: [row sums | arg-max row's value]
          av[u] = part == 1 ? a.arg[b * D + f] : 0;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int m = m0 + u;
        if (m >= nm) break;
        const int64_t b = c0 + m;
        float dl[kHT];
#pragma unroll
        for (int q = 0; q < kHT / 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(&dl_s[m][4 * q]);
          dl[4 * q] = v.x; dl[4 * q + 1] = v.y; dl[4 * q + 2] = v.z; dl[4 * q + 3] = v.w;
        }
        float dfp = 0.f;
#pragma unroll
        for (int t = 0; t < kHT; ++t) {
          dfp = fmaf(dl[t], wk[t], dfp);
          dwacc[t] = fmaf(dl[t], fv[u], dwacc[t]);
        }
        const float g = dfp * (1.f - fv[u] * fv[u]);
        a.g2[b * a.ldg2 + k] = g;
        if (k < kHT) dbacc += dl_s[m][k];
        if (a.sums != nullptr) {
          if (part == 0) {
            const double n = (double)n_s[m];
            const double xs = ((double)rsv[u] - n * mu) * is;
            t1 += n * (double)g;
            t2 += (double)g * xs;
          } else if (av[u] >= 0) {
            const double xa = ((double)rsv[u] - mu) * is;
            t1 += (double)g;
            t2 += (double)g * xa;
          }
        }
      }
    }
    __syncthreads();
  }
  // ---- this workgroup's shares out
#pragma unroll
  for (int t = 0; t < kHT; ++t)
    if (t < TC) atomicAdd(a.dw + (int64_t)t * kHB + k, dwacc[t]);
  if (k < TC && a.db != nullptr) atomicAdd(a.db + k, dbacc);
  red[0][k] = loss_local;
  red[1][k] = 0.0;
  __syncthreads();
  if (k == 0) {
    double t = 0.0;
    for (int i = 0; i < kHB; ++i) t += red[0][i];
    atomicAdd(a.loss_acc, t);
  }
  if (a.sums != nullptr) {
    __syncthreads();
    red[0][k] = t1;
    red[1][k] = t2;
    __syncthreads();
    if (k < D) {
      double* rep = a.sums + (size_t)2 * D * (1 + (blockIdx.x % kBnReplicas));
      atomicAdd(rep + k, red[0][k] + red[0][k + D]);
      atomicAdd(rep + D + k, red[1][k] + red[1][k + D]);
    }
  }
}

// ---------------------------------------------------------------- more than 32 task outputs (PCBA: 128 tasks x 2)
// With up to 256 outputs the two head products are real matrix products (8 192 x 256 x 256 at PCBA's per-GPU batch)
// and go to the matrix cores in the arithmetic of every other product of the library (three-way bf16 split, six
// products per term, fp32 accumulation).  Two kernels:
//   head_bwd_wide_kernel   one workgroup per 32 molecules: loss and d logits (split once, pieces in LDS; fp32 copy to
//                          global for the weight gradient), d fingerprint = d logits . W on the matrix cores with W's
//                          fragments split on the way in from L2, then from the accumulators the tanh derivative, the
//                          rows of g2 and the dense BatchNorm's backward sums, as head_bwd_kernel above;
//   head_wgrad_wide_kernel dW = d logits^T . fingerprint: 64 x 64 blocks of dW times slabs of molecules, both
//                          operands read column-wise (a lane = a column: coalesced rows), split in registers.
// They replace loss_kernel + wgrad_kernel + seg_gemm4_kernel + readout_grad_prep + readout_bn_sums (126 us at 8 192
// molecules x 256 outputs).
constexpr int kWTC = 256;        // most task outputs
constexpr int kWP = kWTC + 8;    // LDS pitch of a row of d logits pieces (bf16): 528 bytes, 16-byte reads conflict-free
// the head matrix as fragment images (head_prep_kernel below): [tile][k-step][piece][lane] 16-byte entries
constexpr int kImgTiles = kHB / 32, kImgKs = kWTC / 16;
constexpr int kImgEntries = kImgTiles * kImgKs * 3 * 64;  // u32x4 entries of one image (393 KB)
static_assert(kHeadImgFloats == 2 * kImgEntries * 4, "common.h: workspace floats of the two head images");

__device__ __forceinline__ f32x16 six_products(const u32x4 (&r)[3], const Frag3& c, f32x16 acc) {
  // rows fragment r (lane = row), columns fragment c (lane = column); small terms first
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(r[2]), as_bf16x8(c.p[0]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(r[0]), as_bf16x8(c.p[2]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(r[1]), as_bf16x8(c.p[1]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(r[1]), as_bf16x8(c.p[0]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(r[0]), as_bf16x8(c.p[1]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(r[0]), as_bf16x8(c.p[0]), acc, 0, 0, 0);
  return acc;
}

#ifdef GCMI_HEAD_DIAG_BUILD  // diagnostic build only (tools/head_diag.sh): phase clocks of workgroup 0, thread 0
__device__ unsigned long long g_head_clk[3][8];
#define HD_BEGIN() unsigned long long hd_t0 = 0, hd_t1 = 0; const bool hd_on = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0; \
  if (hd_on) hd_t0 = __builtin_amdgcn_s_memtime()
#define HD_T(kern, k) do { if (hd_on) { hd_t1 = __builtin_amdgcn_s_memtime(); g_head_clk[kern][k] = hd_t1 - hd_t0; hd_t0 = hd_t1; } } while (0)
#else
#define HD_BEGIN() do { } while (0)
#define HD_T(kern, k) do { } while (0)
#endif

constexpr int kWT = 512;         // threads of the wide kernels: eight waves, one 32-column tile of the fingerprint each

// (the wide kernel spreads its loss over kLossRep accumulators, common.h: same-address fp64 atomics serialise at ~14 ns
// each, and 256 workgroups on one address -- plus 256 on each entry of the bias gradient -- were 7 us of tail)

__global__ void __launch_bounds__(kWT) head_bwd_wide_kernel(HeadArgs a, float* __restrict__ dl_out,
                                                            const u32x4* __restrict__ img) {
  // img: the backward image of head_prep_kernel ([tile of 32 fingerprint columns][k-step over the outputs][piece][lane])
  extern __shared__ __attribute__((aligned(16))) unsigned char head_lds[];
  typedef unsigned short (*DlpT)[kHM][kWP];
  DlpT dlp = reinterpret_cast<DlpT>(head_lds);                                   // [3][32][264] pieces of d logits
  float* fp_s = reinterpret_cast<float*>(head_lds + sizeof(unsigned short) * 3 * kHM * kWP);   // [32][256] fingerprint rows
  float* rs_s = fp_s + kHM * kHB;                                                // [32][256] [row sums | arg-max values]
  int* arg_s = reinterpret_cast<int*>(rs_s + kHM * kHB);                         // [32][128]
  __shared__ int n_s[kHM];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  constexpr int D = kHB / 2;
  const int TC = a.tc;
  const int TCP = (TC + 15) & ~15;
  const int nks = TCP / 16;
  const int64_t c0 = (int64_t)blockIdx.x * kHM;
  const int nm = (int)((a.n_mols - c0) < kHM ? (a.n_mols - c0) : kHM);
  HD_BEGIN();
  // atoms per molecule (the row sums' BatchNorm terms): one (molecule, degree) run per thread, added up in LDS
  if (tid < kHM) n_s[tid] = 0;
  __syncthreads();
  if (a.sums != nullptr) {
    for (int i = tid; i < nm * a.n_deg; i += kWT) {
      const int2 r = *reinterpret_cast<const int2*>(a.runs + (c0 * a.n_deg + i) * 2);
      atomicAdd(&n_s[i / a.n_deg], r.y - r.x);
    }
  }
  // the padding columns of the pieces
  for (int i = tid; i < kHM * (TCP - TC); i += kWT) {
    const int m = i / (TCP - TC), c = TC + i - m * (TCP - TC);
    dlp[0][m][c] = dlp[1][m][c] = dlp[2][m][c] = 0;
  }
  // ---- phase 1: d logits of the 32 molecules (rows beyond n_rows: padding molecules, no loss; the expressions of
  // head_bwd_kernel).  A wave takes four molecules, a lane the tasks lane, lane + 64, ...: no division, and the inputs
  // of eight (molecule, task) items are in flight before the first is worked on.  (The first version mapped items to
  // threads by division and wrote every d logit with its own split and stores: 35 000 cycles of VALU work.)
  double loss_local = 0.0;
  const int C = a.n_classes;
  // this wave's fragments of W from the prepared image (lane = fingerprint column k, eight consecutive outputs per
  // k-step, already split): half of them issued behind phase 1's input loads and ahead of its arithmetic, the other
  // half behind phase 1
  const int k = wave * 32 + l31, f = k & (D - 1);
  const int part = wave >> 2;  // 0: the sum half of the fingerprint, 1: the max half
  u32x4 wr[kWTC / 16][3];
  const u32x4* wsrc = img + ((size_t)wave * kImgKs * 3) * 64 + lane;
  auto load_w = [&](const int ks0, const int ks1) {
#pragma unroll
    for (int ks = 0; ks < kWTC / 16; ++ks) {
      if (ks >= ks0 && ks < ks1 && ks < nks) {
#pragma unroll
        for (int p = 0; p < 3; ++p) wr[ks][p] = wsrc[(ks * 3 + p) * 64];
      }
    }
  };
  // the rows phase 3 needs of the 32 molecules -- fingerprint, BatchNorm inputs, arg-max -- as 16-byte loads into LDS
  // (a lane = a column reading them itself is 48 dword loads per lane)
  auto stage_rows = [&]() {
    const int mb = nm - 1;
#pragma unroll
    for (int p = 0; p < kHM * (kHB / 4) / kWT; ++p) {
      const int slot = tid + p * kWT;
      const int r = slot >> 6, q = slot & 63;
      const int64_t b = c0 + (r < nm ? r : mb);
      *reinterpret_cast<float4*>(fp_s + r * kHB + 4 * q) = *reinterpret_cast<const float4*>(a.fp + b * a.ldfp + 4 * q);
      if (a.sums != nullptr)
        *reinterpret_cast<float4*>(rs_s + r * kHB + 4 * q) = *reinterpret_cast<const float4*>(a.rawsum + b * kHB + 4 * q);
    }
    if (a.sums != nullptr) {
#pragma unroll
      for (int p = 0; p < kHM * (D / 4) / kWT; ++p) {
        const int slot = tid + p * kWT;
        const int r = slot >> 5, q = slot & 31;
        const int64_t b = c0 + (r < nm ? r : mb);
        *reinterpret_cast<int4*>(arg_s + r * D + 4 * q) = *reinterpret_cast<const int4*>(a.arg + b * D + 4 * q);
      }
    }
  };
  if (a.kind == 1 || C == 2) {
    const int E = a.kind == 0 ? 2 : 1;  // outputs per task
    auto round = [&](const int t0, const bool first_round) {
      float x0[8], x1[8], y0[8], y1[8], wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int m = wave * 4 + (u & 3), t = t0 + 64 * (u >> 2);
        const int64_t b = c0 + m;
        const bool live = t < a.n_tasks && m < nm && b < a.n_rows;
        const int64_t item = live ? b * a.n_tasks + t : 0;
        wv[u] = a.weights ? a.weights[item] : 1.f;
        if (a.kind == 0) {
          const float2 xv = *reinterpret_cast<const float2*>(a.logits + item * 2);
          const float2 yv = *reinterpret_cast<const float2*>(a.labels + item * 2);
          x0[u] = xv.x; x1[u] = xv.y; y0[u] = yv.x; y1[u] = yv.y;
        } else {
          x0[u] = a.logits[item]; y0[u] = a.labels[item];
          x1[u] = y1[u] = 0.f;
        }
      }
      if (first_round) load_w(0, kWTC / 32);  // the first half of the fragments; the second half follows phase 1 (registers)
      if (first_round) { HD_T(0, 0); }
      if (first_round) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        HD_T(0, 1);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int m = wave * 4 + (u & 3), t = t0 + 64 * (u >> 2);
        if (t >= a.n_tasks) continue;
        const bool live = m < nm && c0 + m < a.n_rows;
        const float w = live ? wv[u] : 1.f;
        float d0 = 0.f, d1 = 0.f;
        if (a.kind == 0) {
          if (live) {
            // two-class softmax cross entropy with one exponential and one logarithm: with d = x_other - x_max <= 0,
            // log p_max = -log(1 + e^d), log p_other = d - log(1 + e^d), p_max = 1 / (1 + e^d), p_other = e^d / (1 + e^d)
            // (the general loop below, five transcendental calls per item, was 12 us of VALU work at 2 M items)
            const bool first = x0[u] >= x1[u];
            const float dd = first ? x1[u] - x0[u] : x0[u] - x1[u];
            const float e = expf(dd);
            const float se = 1.f + e;
            const float lse = logf(se);
            const float inv = 1.f / se;
            const float lp_max = -lse, lp_oth = dd - lse;
            const float p_max = inv, p_oth = e * inv;
            const float lp0 = first ? lp_max : lp_oth, lp1 = first ? lp_oth : lp_max;
            const float p0 = first ? p_max : p_oth, p1 = first ? p_oth : p_max;
            const float ysum = y0[u] + y1[u];
            const float l = -(y0[u] * lp0) - y1[u] * lp1;
            d0 = w * (p0 * ysum - y0[u]) * a.inv_count;
            d1 = w * (p1 * ysum - y1[u]) * a.inv_count;
            loss_local += (double)(w * l);
          }
        } else if (live) {
          const float dlt = x0[u] - y0[u];
          loss_local += (double)(w * dlt * dlt);
          d0 = 2.f * dlt * w * a.inv_count;
        }
        unsigned p1, p2, p3;
        split3_pair(d0, d1, p1, p2, p3);  // low half = d0
        if (E == 2) {
          *reinterpret_cast<unsigned*>(&dlp[0][m][2 * t]) = p1;
          *reinterpret_cast<unsigned*>(&dlp[1][m][2 * t]) = p2;
          *reinterpret_cast<unsigned*>(&dlp[2][m][2 * t]) = p3;
          if (m < nm) *reinterpret_cast<float2*>(dl_out + (c0 + m) * TC + 2 * t) = make_float2(d0, d1);
        } else {
          dlp[0][m][t] = (unsigned short)p1;
          dlp[1][m][t] = (unsigned short)p2;
          dlp[2][m][t] = (unsigned short)p3;
          if (m < nm) dl_out[(c0 + m) * TC + t] = d0;
        }
      }
    };
    round(lane, true);
    HD_T(0, 2);
    for (int t0 = lane + 128; t0 < a.n_tasks; t0 += 128) round(t0, false);
  } else {
    auto put = [&](int m, int col, float v, bool in_batch) {
      unsigned p1, p2, p3;
      split3(v, p1, p2, p3);
      dlp[0][m][col] = (unsigned short)(p1 >> 16);
      dlp[1][m][col] = (unsigned short)(p2 >> 16);
      dlp[2][m][col] = (unsigned short)(p3 >> 16);
      if (in_batch) dl_out[(c0 + m) * TC + col] = v;
    };
    for (int it = tid; it < kHM * a.n_tasks; it += kWT) {
      const int m = it / a.n_tasks, t = it - m * a.n_tasks;
      const int64_t b = c0 + m;
      const bool live = m < nm && b < a.n_rows;
      const int64_t item = b * a.n_tasks + t;
      const float w = (live && a.weights) ? a.weights[item] : 1.f;
      if (live) {
        const float* x = a.logits + item * C;
        const float* y = a.labels + item * C;
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, x[c]);
        float se = 0.f, ysum = 0.f;
        for (int c = 0; c < C; ++c) {
          se += expf(x[c] - mx);
          ysum += y[c];
        }
        const float lse = logf(se);
        float l = 0.f;
        for (int c = 0; c < C; ++c) {
          const float logp = x[c] - mx - lse;
          const float p = expf(logp);
          l -= y[c] * logp;
          put(m, t * C + c, w * (p * ysum - y[c]) * a.inv_count, true);
        }
        loss_local += (double)(w * l);
      } else {
        for (int c = 0; c < C; ++c) put(m, t * C + c, 0.f, m < nm);
      }
    }
  }
  HD_T(0, 3);
  if (!(a.kind == 1 || C == 2)) load_w(0, kWTC / 32);
  load_w(kWTC / 32, kWTC / 16);
  // ---- what phase 3 reads per molecule
  stage_rows();
  __syncthreads();
  HD_T(0, 4);
  // ---- phase 2: d fingerprint[32 x 256] = d logits[32 x TC] . W[TC x 256], this wave's 32 columns
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < kWTC / 16; ++ks) {
    if (ks < nks) {
      u32x4 r[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) r[p] = *reinterpret_cast<const u32x4*>(&dlp[p][l31][ks * 16 + 8 * half]);
      Frag3 fw;
#pragma unroll
      for (int p = 0; p < 3; ++p) fw.p[p] = wr[ks][p];
      acc = six_products(r, fw, acc);
    }
  }
  HD_T(0, 5);
  // ---- phase 3: from the accumulators (lane = column, registers = molecules) the tanh derivative, g2 and the sums
  {
    const double mu = a.sums ? (double)a.mean[f] : 0.0, is = a.sums ? (double)a.invstd[f] : 0.0;
    double t1 = 0.0, t2 = 0.0;
    float* grow = a.g2 + c0 * a.ldg2 + k;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m < nm) {
        const float fvr = fp_s[m * kHB + k];
        const float g = acc[r] * (1.f - fvr * fvr);
        grow[m * a.ldg2] = g;
        if (a.sums != nullptr) {
          const float rsvr = rs_s[m * kHB + k];            // [row sums | arg-max row's value]
          if (part == 0) {
            const double n = (double)n_s[m];
            const double xs = ((double)rsvr - n * mu) * is;
            t1 += n * (double)g;
            t2 += (double)g * xs;
          } else if (arg_s[m * D + f] >= 0) {
            const double xa = ((double)rsvr - mu) * is;
            t1 += (double)g;
            t2 += (double)g * xa;
          }
        }
      }
    }
    if (a.sums != nullptr) {
      t1 += __shfl_xor(t1, 32);
      t2 += __shfl_xor(t2, 32);
      if (half == 0) {
        double* rep = a.sums + (size_t)2 * D * (1 + (blockIdx.x % kBnReplicas));
        atomicAdd(rep + f, t1);
        atomicAdd(rep + D + f, t2);
      }
    }
  }
  HD_T(0, 6);
  // ---- this wave's loss into one of the replicas (the bias gradient is the weight-gradient kernel's: it reads d logits)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) loss_local += __shfl_xor(loss_local, o);
  if (lane == 0 && loss_local != 0.0) atomicAdd(a.loss_acc + ((blockIdx.x * (kWT / 64) + wave) % kLossRep), loss_local);
  HD_T(0, 7);
}

// dW[tc][k] += sum over the slab's molecules of dl[b][tc] * fp[b][k], db[tc] += sum of dl[b][tc]; grid (slabs, tc blocks
// of 64, 4 column blocks).  Both operands are read column-wise (lane = a column, eight consecutive molecules per k-step:
// every load instruction is two whole 128-byte rows) and split in registers; three workgroups share a CU, so one's
// loads overlap another's products.  Addresses are a pointer per operand and constant multiples of the row pitch: with
// 64-bit index arithmetic per load the loop was bound by the address VALU work (12 700 cycles per 32 molecules).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))
head_wgrad_wide_kernel(const float* __restrict__ dl, int TC, const float* __restrict__ fp, int ldfp, int64_t n_mols,
                       int rows_per_slab, float* __restrict__ dw, float* __restrict__ db) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int tc = blockIdx.y * 64 + (wave & 1) * 32 + l31;
  const int tcc = tc < TC ? tc : TC - 1;
  const int k = blockIdx.z * 64 + (wave >> 1) * 32 + l31;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_slab;
  const int64_t r1 = (r0 + rows_per_slab) < n_mols ? (r0 + rows_per_slab) : n_mols;
  if ((blockIdx.y * 64 + (wave & 1) * 32) >= TC) return;  // (a whole wave: no barrier below)
  const bool bias_wave = db != nullptr && blockIdx.z == 0 && (wave >> 1) == 0;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float dbacc = 0.f;
  HD_BEGIN();
  const float* pa = dl + (r0 + 8 * half) * TC + tcc;
  const float* pb = fp + (r0 + 8 * half) * ldfp + k;
  const float keep = tc < TC ? 1.f : 0.f;
  int64_t b0 = r0;
  // two k-steps (32 molecules) per round; the next round's 32 loads per lane are in flight during this round's products
  float ca[2][8], cb[2][8], na[2][8], nb[2][8];
  auto load_round = [&](float (&va)[2][8], float (&vb)[2][8]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        va[s][i] = pa[(16 * s + i) * TC];
        vb[s][i] = pb[(16 * s + i) * ldfp];
      }
    pa += 32 * TC;
    pb += 32 * ldfp;
  };
  if (b0 + 32 <= r1) load_round(ca, cb);
  for (; b0 + 32 <= r1; b0 += 32) {
    if (b0 + 64 <= r1) load_round(na, nb);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < 8; ++i) ca[s][i] *= keep;
      if (bias_wave) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dbacc += ca[s][i];
      }
      const Frag3 fa = split_frag(ca[s]);
      const Frag3 fb = split_frag(cb[s]);
      const u32x4 r[3] = {fa.p[0], fa.p[1], fa.p[2]};
      acc = six_products(r, fb, acc);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ca[s][i] = na[s][i];
        cb[s][i] = nb[s][i];
      }
  }
  if (b0 < r1) {  // the last slab's ragged end: clamped addresses, zeroed values
    float va[2][8], vb[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t b = b0 + 16 * s + 8 * half + i;
        const int64_t bc = b < r1 ? b : r1 - 1;
        const float x = dl[bc * TC + tcc], y = fp[bc * ldfp + k];
        va[s][i] = b < r1 ? x * keep : 0.f;
        vb[s][i] = b < r1 ? y : 0.f;
      }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (bias_wave) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dbacc += va[s][i];
      }
      const Frag3 fa = split_frag(va[s]);
      const Frag3 fb = split_frag(vb[s]);
      const u32x4 r[3] = {fa.p[0], fa.p[1], fa.p[2]};
      acc = six_products(r, fb, acc);
    }
  }
  HD_T(1, 0);
  const int row0 = blockIdx.y * 64 + (wave & 1) * 32 + 4 * half;
  float* dcol = dw + k;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + (r & 3) + 8 * (r >> 2);
    if (row < TC && acc[r] != 0.f) atomicAdd(dcol + row * kHB, acc[r]);
  }
  if (bias_wave) {
    dbacc += __shfl_xor(dbacc, 32);
    if (half == 0 && tc < TC && dbacc != 0.f) atomicAdd(db + tc, dbacc);
  }
  HD_T(1, 1);
}

// ---------------------------------------------------------------- forward head with more than 32 outputs
// out[rows x TC] = fingerprint[rows x 256] . W^T + b (graphconvmodel.py:177-179): one workgroup per 32 rows -- 256
// workgroups at PCBA's 8 192 molecules per GPU, one per CU -- eight waves, a 32-column tile of the output each.  The 32
// rows are split once into their three bf16 pieces in LDS.  The weights (TC x 256, nn.Linear) pass through LDS as well,
// one k-step (16 contraction columns) at a time through two buffers: read 16 bytes per lane along the rows of W (every
// thread's pieces of all chunks in flight from the start), split once per workgroup, and picked up by the waves as
// 16-byte fragments.  (Fragments straight from global -- a lane = an output
// column = a row of W, 32 bytes per k-step -- are 64 different lines per load instruction: 22 000 cycles of a 35 000
// cycle kernel went into that.)
constexpr int kFWC = 16;          // contraction columns per weight chunk: one k-step
constexpr int kFWP = kFWC + 8;    // LDS pitch of a weight row's pieces (bf16): 48 bytes, 16-byte reads conflict-free

__global__ void __launch_bounds__(kWT) head_fwd_wide_kernel(const float* __restrict__ in, int64_t ldin, int64_t n_rows,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            int TC, float* __restrict__ out, int64_t ldo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char head_lds[];
  typedef unsigned short (*ApT)[kHM][kWP];
  typedef unsigned short (*WpT)[3][kWTC][kFWP];
  ApT ap = reinterpret_cast<ApT>(head_lds);                                             // [3][32][264]
  WpT wp = reinterpret_cast<WpT>(head_lds + sizeof(unsigned short) * 3 * kHM * kWP);   // [2][3][256][24]: two chunks
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int64_t row0 = (int64_t)blockIdx.x * kHM;
  const int valid = (int)((n_rows - row0) < kHM ? (n_rows - row0) : kHM);
  const int n = wave * 32 + l31;                 // this lane's output column
  const bool tile = wave * 32 < TC;              // (wave-uniform)
  HD_BEGIN();
  const float bv = (bias != nullptr && n < TC) ? bias[n] : 0.f;
  // a chunk of W: 256 rows x 16 columns = 1 024 float4, two per thread; thread -> (row, 16-byte piece), four threads per
  // 64-byte half line (the other half is the next chunk's: an L1 hit)
  constexpr int WPT = kWTC * (kFWC / 4) / kWT;
  constexpr int NCH = kHB / kFWC;
  float4 wreg[NCH][WPT];
  bool wok[WPT];
#pragma unroll
  for (int p = 0; p < WPT; ++p) wok[p] = ((tid + p * kWT) >> 2) < TC;
  auto load_chunks = [&]() {  // this thread's pieces of ALL chunks: one exposure of the L2 latency
#pragma unroll
    for (int p = 0; p < WPT; ++p) {
      const int slot = tid + p * kWT;
      const int r = slot >> 2, q = slot & 3;
      const float* src = w + (r < TC ? r : TC - 1) * kHB + 4 * q;
#pragma unroll
      for (int c = 0; c < NCH; ++c) wreg[c][p] = *reinterpret_cast<const float4*>(src + c * kFWC);
    }
  };
  auto store_chunk = [&](int c) {
#pragma unroll
    for (int p = 0; p < WPT; ++p) {
      const int slot = tid + p * kWT;
      const int r = slot >> 2, q = slot & 3;
      float4 v = wreg[c][p];
      if (!wok[p]) v = make_float4(0.f, 0.f, 0.f, 0.f);
      unsigned a1, a2, a3, b1, b2, b3;
      split3_pair(v.x, v.y, a1, a2, a3);
      split3_pair(v.z, v.w, b1, b2, b3);
      *reinterpret_cast<uint2*>(&wp[c & 1][0][r][4 * q]) = make_uint2(a1, b1);
      *reinterpret_cast<uint2*>(&wp[c & 1][1][r][4 * q]) = make_uint2(a2, b2);
      *reinterpret_cast<uint2*>(&wp[c & 1][2][r][4 * q]) = make_uint2(a3, b3);
    }
  };
  // the rows first (their loads return first), the weights behind them
  float4 ain[kHM * (kHB / 4) / kWT];
#pragma unroll
  for (int p = 0; p < kHM * (kHB / 4) / kWT; ++p) {
    const int slot = tid + p * kWT;
    const int r = slot >> 6, q = slot & 63;
    ain[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < valid) ain[p] = *reinterpret_cast<const float4*>(in + (row0 + r) * ldin + 4 * q);
  }
  load_chunks();
#pragma unroll
  for (int p = 0; p < kHM * (kHB / 4) / kWT; ++p) {
    const int slot = tid + p * kWT;
    const int r = slot >> 6, q = slot & 63;
    unsigned a1, a2, a3, b1, b2, b3;
    split3_pair(ain[p].x, ain[p].y, a1, a2, a3);
    split3_pair(ain[p].z, ain[p].w, b1, b2, b3);
    *reinterpret_cast<uint2*>(&ap[0][r][4 * q]) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(&ap[1][r][4 * q]) = make_uint2(a2, b2);
    *reinterpret_cast<uint2*>(&ap[2][r][4 * q]) = make_uint2(a3, b3);
  }
  store_chunk(0);
  __syncthreads();
  HD_T(2, 0);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // one barrier per k-step: while a wave multiplies chunk c, the others split chunk c + 1 into the other buffer
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c + 1 < NCH) store_chunk(c + 1);
    if (tile) {
      u32x4 r[3];
      Frag3 fw;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        r[p] = *reinterpret_cast<const u32x4*>(&ap[p][l31][c * kFWC + 8 * half]);
        fw.p[p] = *reinterpret_cast<const u32x4*>(&wp[c & 1][p][n][8 * half]);
      }
      acc = six_products(r, fw, acc);
    }
    __syncthreads();
  }
  HD_T(2, 1);
  if (tile && n < TC) {
    float* ocol = out + row0 * ldo + n;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m < valid) ocol[m * ldo] = acc[r] + bv;
    }
  }
  HD_T(2, 2);
}

// ---------------------------------------------------------------- the head matrix, prepared once per step
// What the phase clocks of the kernels above say: splitting W in every workgroup is most of the forward,
// and reading W[tc][k] for a lane = a column k (the backward's operand) is 128 dword loads per lane at ~22 cycles of
// the CU's memory pipe each.  head_prep_kernel splits W ONCE into the two fragment orders -- [tile][k-step][piece][lane]
// 16-byte entries, for the forward (a lane = an output, eight consecutive fingerprint columns) and for the backward (a
// lane = a fingerprint column, eight consecutive outputs) -- so that a wave fetches its operand as 3 x 16 whole 1 KiB
// lines and multiplies without any VALU work on it.

__global__ void __launch_bounds__(256) head_prep_kernel(const float* __restrict__ w, int TC, u32x4* __restrict__ img) {
  const int id = blockIdx.x * 256 + threadIdx.x;  // (image, tile, k-step, lane)
  const int which = id / (kImgTiles * kImgKs * 64);
  const int rem = id - which * (kImgTiles * kImgKs * 64);
  const int tile = rem / (kImgKs * 64), ks = (rem / 64) % kImgKs, lane = rem & 63;
  const int half = lane >> 5, l31 = lane & 31;
  float v[8];
  if (which == 0) {  // forward: lane = output n, contraction along the fingerprint
    const int n = tile * 32 + l31;
    const float* src = w + (int64_t)(n < TC ? n : TC - 1) * kHB + ks * 16 + 8 * half;
    const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
    const float keep = n < TC ? 1.f : 0.f;
    v[0] = a.x * keep; v[1] = a.y * keep; v[2] = a.z * keep; v[3] = a.w * keep;
    v[4] = b.x * keep; v[5] = b.y * keep; v[6] = b.z * keep; v[7] = b.w * keep;
  } else {           // backward: lane = fingerprint column k, contraction along the outputs
    const int k = tile * 32 + l31;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int tc = ks * 16 + 8 * half + i;
      const float x = w[(int64_t)(tc < TC ? tc : TC - 1) * kHB + k];
      v[i] = tc < TC ? x : 0.f;
    }
  }
  const Frag3 f = split_frag(v);
  u32x4* dst = img + (size_t)which * kImgEntries + ((size_t)(tile * kImgKs + ks) * 3) * 64 + lane;
  dst[0] = f.p[0];
  dst[64] = f.p[1];
  dst[128] = f.p[2];
}

// The forward over the prepared image: a workgroup = 32 rows, eight waves = the eight 32-column tiles of the output;
// rows split once into LDS pieces; the wave's 48 weight fragments straight from the image, half of them in flight
// while the other half is multiplied.
__global__ void __launch_bounds__(kWT) head_fwd_img_kernel(const float* __restrict__ in, int64_t ldin, int64_t n_rows,
                                                           const u32x4* __restrict__ img, const float* __restrict__ bias,
                                                           int TC, float* __restrict__ out, int64_t ldo) {
  __shared__ __attribute__((aligned(16))) unsigned short ap[3][kHM][kWP];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int64_t row0 = (int64_t)blockIdx.x * kHM;
  const int valid = (int)((n_rows - row0) < kHM ? (n_rows - row0) : kHM);
  const int n = wave * 32 + l31;
  const bool tile = wave * 32 < TC;  // (wave-uniform)
  HD_BEGIN();
  float4 ain[kHM * (kHB / 4) / kWT];
#pragma unroll
  for (int p = 0; p < kHM * (kHB / 4) / kWT; ++p) {
    const int slot = tid + p * kWT;
    const int r = slot >> 6, q = slot & 63;
    ain[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < valid) ain[p] = *reinterpret_cast<const float4*>(in + (row0 + r) * ldin + 4 * q);
  }
  const float bv = (bias != nullptr && n < TC) ? bias[n] : 0.f;
  constexpr int NKS = kHB / 16, HK = NKS / 2;
  const u32x4* wsrc = img + ((size_t)wave * kImgKs * 3) * 64 + lane;
  u32x4 wa[HK][3], wb[HK][3];
  if (tile) {
#pragma unroll
    for (int ks = 0; ks < HK; ++ks)
#pragma unroll
      for (int p = 0; p < 3; ++p) wa[ks][p] = wsrc[(ks * 3 + p) * 64];
  }
#pragma unroll
  for (int p = 0; p < kHM * (kHB / 4) / kWT; ++p) {
    const int slot = tid + p * kWT;
    const int r = slot >> 6, q = slot & 63;
    unsigned a1, a2, a3, b1, b2, b3;
    split3_pair(ain[p].x, ain[p].y, a1, a2, a3);
    split3_pair(ain[p].z, ain[p].w, b1, b2, b3);
    *reinterpret_cast<uint2*>(&ap[0][r][4 * q]) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(&ap[1][r][4 * q]) = make_uint2(a2, b2);
    *reinterpret_cast<uint2*>(&ap[2][r][4 * q]) = make_uint2(a3, b3);
  }
  if (tile) {
#pragma unroll
    for (int ks = 0; ks < HK; ++ks)
#pragma unroll
      for (int p = 0; p < 3; ++p) wb[ks][p] = wsrc[((HK + ks) * 3 + p) * 64];
  }
  __syncthreads();
  HD_T(2, 0);
  if (!tile) return;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    u32x4 r[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) r[p] = *reinterpret_cast<const u32x4*>(&ap[p][l31][ks * 16 + 8 * half]);
    Frag3 fw;
#pragma unroll
    for (int p = 0; p < 3; ++p) fw.p[p] = ks < HK ? wa[ks < HK ? ks : 0][p] : wb[ks >= HK ? ks - HK : 0][p];
    acc = six_products(r, fw, acc);
  }
  HD_T(2, 1);
  if (n < TC) {
    float* ocol = out + row0 * ldo + n;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m < valid) ocol[m * ldo] = acc[r] + bv;
    }
  }
  HD_T(2, 2);
}

#ifdef GCMI_HEAD_DIAG_BUILD
static void head_diag_print(const char* what, int kern, hipStream_t st) {
  static int printed = 0;
  unsigned long long h[3][8];
  if (printed++ < 9 && hipStreamSynchronize(st) == hipSuccess &&
      hipMemcpyFromSymbol(h, HIP_SYMBOL(g_head_clk), sizeof(h)) == hipSuccess)
    fprintf(stderr, "head_diag %s: %llu %llu %llu %llu %llu %llu %llu %llu (s_memtime ticks per phase, workgroup 0)\n", what,
            h[kern][0], h[kern][1], h[kern][2], h[kern][3], h[kern][4], h[kern][5], h[kern][6], h[kern][7]);
}
#define HD_PRINT(what, kern, st) head_diag_print(what, kern, st)
#else
#define HD_PRINT(what, kern, st) do { } while (0)
#endif

// fewest outputs that take the matrix-core head kernels (GCMI_HEAD_WIDE_MIN; default 33: up to 32 outputs head_bwd_kernel
// keeps a thread's column of the head matrix in registers)
static int head_wide_min() {
  static const int v = getenv("GCMI_HEAD_WIDE_MIN") ? atoi(getenv("GCMI_HEAD_WIDE_MIN")) : kHT + 1;
  return v < 1 ? 1 : v;
}

static bool head_wide_enabled() {
  static const bool on = !(getenv("GCMI_HEAD_WIDE") && atoi(getenv("GCMI_HEAD_WIDE")) == 0);
  return on && !gemm_exact_mode();
}

// the two fragment images of the head matrix (kHeadImgFloats floats at d_img); GCMI_ERR_UNSUPPORTED: outside 33..256 outputs
int head_prep(const float* d_w, int32_t n_out, float* d_img, hipStream_t st) {
  if (!head_wide_enabled() || n_out < head_wide_min() || n_out > kWTC || d_img == nullptr || !aligned16(d_w) || !aligned16(d_img))
    return GCMI_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(head_prep_kernel, dim3(2 * kImgTiles * kImgKs * 64 / 256), dim3(256), 0, st, d_w, n_out,
                     reinterpret_cast<u32x4*>(d_img));
  GCMI_CHECK_LAUNCH("head_prep");
  return GCMI_OK;
}

// GCMI_ERR_UNSUPPORTED: not (one segment of 256-column rows times an nn.Linear weight with 33..256 outputs, no activation).
// d_img: the images head_prep made of THIS d_w (the forward one is read), or nullptr: the weights are split per workgroup
int head_fwd_wide(const float* d_in, int64_t ldin, int64_t n_rows, int32_t k, const float* d_w, const float* d_bias,
                  int32_t n_out, int32_t act, float* d_out, int64_t ldo, hipStream_t st, const float* d_img) {
  if (!head_wide_enabled() || k != kHB || n_out < head_wide_min() || n_out > kWTC || act != 0 || ldin % 4 != 0 || !aligned16(d_in) ||
      !aligned16(d_w) || n_rows <= 0)
    return GCMI_ERR_UNSUPPORTED;
  if (d_img != nullptr) {
    hipLaunchKernelGGL(head_fwd_img_kernel, dim3((unsigned)((n_rows + kHM - 1) / kHM)), dim3(kWT), 0, st, d_in, ldin, n_rows,
                       reinterpret_cast<const u32x4*>(d_img), d_bias, n_out, d_out, ldo);
    GCMI_CHECK_LAUNCH("head_fwd_img");
    HD_PRINT("fwd_img", 2, st);
    return GCMI_OK;
  }
  constexpr size_t shmem = sizeof(unsigned short) * 3 * (kHM * kWP + 2 * kWTC * kFWP);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(head_fwd_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)shmem) != hipSuccess) {
      (void)hipGetLastError();
      return GCMI_ERR_UNSUPPORTED;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(head_fwd_wide_kernel, dim3((unsigned)((n_rows + kHM - 1) / kHM)), dim3(kWT), shmem, st, d_in, ldin,
                     n_rows, d_w, d_bias, n_out, d_out, ldo);
  GCMI_CHECK_LAUNCH("head_fwd_wide");
  HD_PRINT("fwd_wide", 2, st);
  return GCMI_OK;
}

// GCMI_ERR_UNSUPPORTED: other widths than a 256-column fingerprint; more than 32 task outputs without d_dl_scratch
// (n_mols x outputs floats) and d_img (head_prep's images of d_w), or more than 256
int head_bwd_fused(int32_t kind, const float* d_logits, const float* d_labels, const float* d_weights, int64_t n_rows,
                   int32_t n_tasks, int32_t n_classes, int64_t n_mols, const float* d_fp, int64_t ldfp,
                   const float* d_w, float* d_dw, float* d_db, float* d_g2, int64_t ldg2, double* d_loss_acc,
                   const int32_t* d_runs, int32_t n_deg, const int32_t* d_arg, const float* d_rawsum,
                   const float* d_mean, const float* d_invstd, double* d_sums, int32_t dense_width, hipStream_t st,
                   float* d_dl_scratch, const float* d_img) {
  static const bool on = !(getenv("GCMI_FUSED_HEAD") && atoi(getenv("GCMI_FUSED_HEAD")) == 0);
  const int tc = n_tasks * (kind == 0 ? n_classes : 1);
  if (!on || !fused_bwd_enabled() || 2 * dense_width != kHB || tc < 1 || n_mols <= 0) return GCMI_ERR_UNSUPPORTED;
  const bool wide = tc > kHT || (tc >= head_wide_min() && d_dl_scratch != nullptr && d_img != nullptr && head_wide_enabled());
  if (wide && (tc > kWTC || d_dl_scratch == nullptr || d_img == nullptr || !head_wide_enabled() || ldfp % 4 != 0 ||
               !aligned16(d_fp)))
    return GCMI_ERR_UNSUPPORTED;
  if (d_sums != nullptr && (!d_runs || !d_arg || !d_rawsum || !d_mean || !d_invstd)) return GCMI_ERR_UNSUPPORTED;
  HeadArgs a;
  memset(&a, 0, sizeof(a));
  a.kind = kind; a.n_tasks = n_tasks; a.n_classes = kind == 0 ? n_classes : 1; a.tc = tc;
  a.n_rows = n_rows; a.n_mols = n_mols; a.inv_count = 1.f / (float)(n_rows * n_tasks);
  a.logits = d_logits; a.labels = d_labels; a.weights = d_weights; a.fp = d_fp; a.ldfp = ldfp;
  a.w = d_w; a.dw = d_dw; a.db = d_db; a.g2 = d_g2; a.ldg2 = ldg2; a.loss_acc = d_loss_acc;
  a.runs = d_runs; a.n_deg = n_deg; a.arg = d_arg; a.rawsum = d_rawsum; a.mean = d_mean; a.invstd = d_invstd;
  a.sums = d_sums;
  if (wide) {
    constexpr size_t shmem = sizeof(unsigned short) * 3 * kHM * kWP + sizeof(float) * (2 * kHM * kHB + kHM * (kHB / 2));
    static bool attr_done = false;
    if (!attr_done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(head_bwd_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)shmem) != hipSuccess) {
        (void)hipGetLastError();
        return GCMI_ERR_UNSUPPORTED;
      }
      attr_done = true;
    }
    hipLaunchKernelGGL(head_bwd_wide_kernel, dim3((unsigned)((n_mols + kHM - 1) / kHM)), dim3(kWT), shmem, st, a, d_dl_scratch,
                       reinterpret_cast<const u32x4*>(d_img) + kImgEntries);
    GCMI_CHECK_LAUNCH("head_bwd_wide");
    HD_PRINT("bwd_wide", 0, st);
    // slabs: two workgroups per CU over all blocks of dW (measured at 8 192 x 256: 1 024 workgroups 24.1 us, 512: 18.8, 256: 19.8 -- the atomics of more, shorter slabs against the latency of fewer, longer ones), never below 64 molecules
    static const int wg_env = getenv("GCMI_HEAD_WGRAD_WGS") ? atoi(getenv("GCMI_HEAD_WGRAD_WGS")) : 512;
    const int blocks = ((tc + 63) / 64) * 4;
    int64_t slabs = std::max<int64_t>(1, wg_env / blocks);
    slabs = std::min<int64_t>(slabs, (n_mols + 63) / 64);
    const int rps = (int)(((n_mols + slabs - 1) / slabs + 31) / 32 * 32);
    slabs = (n_mols + rps - 1) / rps;
    hipLaunchKernelGGL(head_wgrad_wide_kernel, dim3((unsigned)slabs, (unsigned)((tc + 63) / 64), 4), dim3(256), 0, st,
                       d_dl_scratch, tc, d_fp, (int)ldfp, n_mols, rps, d_dw, d_db);
    GCMI_CHECK_LAUNCH("head_wgrad_wide");
    HD_PRINT("wgrad_wide", 1, st);
    return GCMI_OK;
  }
  // three workgroups are resident per CU (~130 VGPRs): 3 x 256 CUs, every workgroup in the first wave.  Measured at
  // 65 536 molecules: 256 workgroups 149 us, 512: 97, 768: 90, 1 024: 107, 2 048: 111.
  const int grid = (int)std::min<int64_t>(768, (n_mols + kHM - 1) / kHM);
  hipLaunchKernelGGL(head_bwd_kernel, dim3(grid), dim3(kHB), 0, st, a);
  GCMI_CHECK_LAUNCH("head_bwd");
  return GCMI_OK;
}

}  // namespace gcmi
