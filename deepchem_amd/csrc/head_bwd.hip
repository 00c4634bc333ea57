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

// GCMI_ERR_UNSUPPORTED: other widths than a 256-column fingerprint, more than 32 task outputs
int head_bwd_fused(int32_t kind, const float* d_logits, const float* d_labels, const float* d_weights, int64_t n_rows,
                   int32_t n_tasks, int32_t n_classes, int64_t n_mols, const float* d_fp, int64_t ldfp,
                   const float* d_w, float* d_dw, float* d_db, float* d_g2, int64_t ldg2, double* d_loss_acc,
                   const int32_t* d_runs, int32_t n_deg, const int32_t* d_arg, const float* d_rawsum,
                   const float* d_mean, const float* d_invstd, double* d_sums, int32_t dense_width, hipStream_t st) {
  static const bool on = !(getenv("GCMI_FUSED_HEAD") && atoi(getenv("GCMI_FUSED_HEAD")) == 0);
  const int tc = n_tasks * (kind == 0 ? n_classes : 1);
  if (!on || !fused_bwd_enabled() || 2 * dense_width != kHB || tc > kHT || tc < 1 || n_mols <= 0) return GCMI_ERR_UNSUPPORTED;
  if (d_sums != nullptr && (!d_runs || !d_arg || !d_rawsum || !d_mean || !d_invstd)) return GCMI_ERR_UNSUPPORTED;
  HeadArgs a;
  memset(&a, 0, sizeof(a));
  a.kind = kind; a.n_tasks = n_tasks; a.n_classes = kind == 0 ? n_classes : 1; a.tc = tc;
  a.n_rows = n_rows; a.n_mols = n_mols; a.inv_count = 1.f / (float)(n_rows * n_tasks);
  a.logits = d_logits; a.labels = d_labels; a.weights = d_weights; a.fp = d_fp; a.ldfp = ldfp;
  a.w = d_w; a.dw = d_dw; a.db = d_db; a.g2 = d_g2; a.ldg2 = ldg2; a.loss_acc = d_loss_acc;
  a.runs = d_runs; a.n_deg = n_deg; a.arg = d_arg; a.rawsum = d_rawsum; a.mean = d_mean; a.invstd = d_invstd;
  a.sums = d_sums;
  // three workgroups are resident per CU (~130 VGPRs): 3 x 256 CUs, every workgroup in the first wave.  Measured at
  // 65 536 molecules: 256 workgroups 149 us, 512: 97, 768: 90, 1 024: 107, 2 048: 111.
  const int grid = (int)std::min<int64_t>(768, (n_mols + kHM - 1) / kHM);
  hipLaunchKernelGGL(head_bwd_kernel, dim3(grid), dim3(kHB), 0, st, a);
  GCMI_CHECK_LAUNCH("head_bwd");
  return GCMI_OK;
}

}  // namespace gcmi
