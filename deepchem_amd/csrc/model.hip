// Whole-model sequencing: _GraphConvTorchModel.forward
// (models/torch_models/graphconvmodel.py:188-249) and the loss + backward of one
// fit_generator step (models/torch_models/torch_model.py:436-442) as one C call each.
// Host code only enqueues the kernels of this library on the caller's stream; the few
// device functions here are parameter-layout helpers (bias packing, counters).
#include "common.h"
#include "split_bf16.h"

namespace gcmi {

constexpr int kMaxL = GCMI_MAX_CONV_LAYERS;

struct BiasLayers {  // the GraphConv layers' bias blocks: one launch packs (or unpacks) all of them (blockIdx.y = layer)
  const float* src[kMaxL];
  float* dst[kMaxL];
  int width[kMaxL];
};

__global__ void bias_pack_kernel(BiasLayers bl, int max_deg) {
  // b_list: (2*max_deg+1, width) in reference order; bsum[d] = b_rel_d + b_self_d, bsum[0] = b_self_0
  const int l = blockIdx.y;
  const float* __restrict__ b_list = pick_n(bl.src, l);
  float* __restrict__ bsum = pick_n(bl.dst, l);
  const int width = pick_n(bl.width, l);
  if (b_list == nullptr) return;
  const int n = (max_deg + 1) * width;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int d = i / width, c = i - d * width;
    bsum[i] = d == 0 ? b_list[(2 * max_deg) * width + c]
                     : b_list[(2 * (d - 1)) * width + c] + b_list[(2 * (d - 1) + 1) * width + c];
  }
}

__global__ void bias_unpack_kernel(BiasLayers bl, int max_deg) {
  const int l = blockIdx.y;
  const float* __restrict__ dbsum = pick_n(bl.src, l);
  float* __restrict__ db_list = pick_n(bl.dst, l);
  const int width = pick_n(bl.width, l);
  if (dbsum == nullptr) return;
  const int n = (2 * max_deg + 1) * width;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int k = i / width, c = i - k * width;
    const int d = k == 2 * max_deg ? 0 : k / 2 + 1;
    db_list[i] = dbsum[d * width + c];
  }
}

struct CounterPtrs {
  int64_t* p[kMaxL + 1];
  int n;
};
__global__ void bump_counters_kernel(CounterPtrs c) {
  const int i = threadIdx.x;
  if (i < c.n && c.p[i] != nullptr) *c.p[i] += 1;
}

static inline int64_t up4(int64_t n) { return (n + 3) / 4 * 4; }

// Workspace carve-up (floats).  Everything the backward needs from the forward, plus the
// backward's temporaries.  All blocks start 16-byte aligned.
struct Ws {
  int64_t S[kMaxL], gc[kMaxL], pool[kMaxL], arg[kMaxL], bsum[kMaxL], bnv[kMaxL + 1];
  int64_t ldS[kMaxL], ngather[kMaxL];
  int64_t dense, arg_r, rsum, dfp, tA, tB, tC, tD, tE, total;
  int64_t xb;    // storage == 1: bf16 copy of the atom features (ld = ldS[0])
  int64_t wimg;  // scratch of the forward products (split weight fragments in lane order, rebuilt by every launch)
  int64_t himg;  // more than 32 task outputs: the head matrix's two fragment images (head_bwd.hip: head_prep), forward -> backward
  // one region the backward zeroes with a single memset: [dlogits | dbsum per layer | lacc | acc]
  int64_t dlogits, dbsum[kMaxL], lacc, acc, acc2, z_end;
};

static inline int64_t up8(int64_t n) { return (n + 7) / 8 * 8; }

static Ws carve(const gcmi_model_desc* m, int64_t N, int64_t B, int64_t ld_features) {
  Ws w;
  memset(&w, 0, sizeof(w));
  int64_t off = 0;
  auto take = [&](int64_t n) {
    int64_t o = off;
    off += up4(n);
    return o;
  };
  // storage == 1: the matrices the step writes and reads back (S, gc, pool, dense, and a copy of the atom features)
  // are bf16: rows of ld ELEMENTS with ld a multiple of 8, half the floats of the workspace per element
  const bool h = m->storage >= 1;
  auto take_act = [&](int64_t rows, int64_t ld) { return take(h ? (rows * ld + 1) / 2 : rows * ld); };
  const int L = m->n_layers;
  int64_t wmax = m->dense_width, kmax = up4(m->n_feat_in);
  for (int l = 0; l < L; ++l) {
    const int64_t k = l == 0 ? m->n_feat_in : m->conv_width[l - 1];
    const int64_t ldx = l == 0 ? ld_features : m->conv_width[l - 1];
    // gather over the padded width when the rows are 16-byte addressable (pad columns are 0)
    w.ngather[l] = (l == 0 && ldx % 4 == 0 && ldx < k + 4) ? ldx : k;
    if (h && l == 0) w.ngather[l] = up4(k);
    w.ldS[l] = h ? up8(w.ngather[l]) : up4(w.ngather[l]);
    const int64_t wd = m->conv_width[l];
    w.S[l] = take_act(N, w.ldS[l]);
    if (h && l == 0) w.xb = take_act(N, w.ldS[0]);
    if (l == 0) w.wimg = take(kFwdHWimgFloats);
    w.gc[l] = take_act(N, wd);
    w.pool[l] = take_act(N, wd);
    w.arg[l] = take((N * wd + 3) / 4);
    w.bsum[l] = take((int64_t)(m->max_deg + 1) * wd);
    w.bnv[l] = take(4 * wd);
    if (wd > wmax) wmax = wd;
    if (w.ldS[l] > kmax) kmax = w.ldS[l];
  }
  const int64_t D = m->dense_width;
  const int64_t TC = (int64_t)m->n_tasks * m->n_classes;
  w.bnv[L] = take(4 * D);
  w.himg = (TC <= 256 && 2 * D == 256) ? take(kHeadImgFloats) : -1;  // (786 KB; used from head_wide_min() outputs on)
  w.dense = take_act(N, D);
  w.arg_r = take(B * D);
  w.rsum = take(2 * B * D);  // per-molecule [row sums | arg-max row value] of the dense output (BatchNorm backward)
  w.dfp = take(B * 2 * D);
  w.tA = take(N * wmax);
  w.tB = take(N * wmax);
  w.tC = take(N * wmax);
  w.tD = take(N * wmax);
  w.tE = take(N * kmax);
  w.dlogits = take(B * TC);
  for (int l = 0; l < L; ++l) w.dbsum[l] = take((int64_t)(m->max_deg + 1) * m->conv_width[l]);
  w.lacc = take(2 * kLossRep);
  w.acc = take(2 * GCMI_BN_ACC_DOUBLES(wmax));
  w.acc2 = take(2 * GCMI_BN_ACC_DOUBLES(wmax));  // pooled BatchNorm-backward sums of the block below (bwd_fused.hip)
  w.z_end = off;
  w.total = off;
  return w;
}

// b_rel_d + b_self_d of every GraphConv layer (the products add ONE bias row per degree): one launch for all layers
static int pack_biases(const gcmi_model_desc* m, const Ws& w, float* ws, const float* d_params, hipStream_t st) {
  BiasLayers bl;
  memset(&bl, 0, sizeof(bl));
  for (int l = 0; l < m->n_layers; ++l) {
    bl.src[l] = d_params + m->off_conv_b[l];
    bl.dst[l] = ws + w.bsum[l];
    bl.width[l] = m->conv_width[l];
  }
  hipLaunchKernelGGL(bias_pack_kernel, dim3(4, m->n_layers), dim3(256), 0, st, bl, m->max_deg);
  GCMI_CHECK_LAUNCH("bias_pack");
  return GCMI_OK;
}

// ... and the per-degree bias gradients back into the reference's (2 max_deg + 1) rows, for the layers whose block ran
static int unpack_bias_grads(const gcmi_model_desc* m, const BiasLayers& bl, int n_layers, hipStream_t st) {
  bool any = false;
  for (int l = 0; l < n_layers; ++l) any = any || bl.src[l] != nullptr;
  if (!any) return GCMI_OK;
  hipLaunchKernelGGL(bias_unpack_kernel, dim3(4, n_layers), dim3(256), 0, st, bl, m->max_deg);
  GCMI_CHECK_LAUNCH("bias_unpack");
  return GCMI_OK;
}

// The task head's forward product: with more than 32 outputs on the prepared images (head_bwd.hip), which stay in the
// workspace for the backward; otherwise (and in the exact product mode) the segmented product.
static int head_forward(const gcmi_model_desc* m, const Ws& w, float* ws, const float* d_params, const gcmi_model_io* io,
                        int64_t B, void* stream) {
  const int D = m->dense_width;
  const int TC = m->n_tasks * m->n_classes;
  const int32_t zero32 = 0, nB = (int32_t)B;
  const int64_t zero64 = 0;
  if (w.himg >= 0 && B > 0) {
    hipStream_t st = (hipStream_t)stream;
    int rc = head_prep(d_params + m->off_head_w, TC, ws + w.himg, st);
    if (rc == GCMI_OK)
      rc = head_fwd_wide(io->d_fingerprint, 2 * D, B, 2 * D, d_params + m->off_head_w, d_params + m->off_head_b, TC, 0,
                         io->d_logits, TC, st, ws + w.himg);
    if (rc != GCMI_ERR_UNSUPPORTED) return rc;
  }
  return gcmi_seg_gemm(1, &zero32, &nB, io->d_fingerprint, 2 * D, 2 * D, d_params + m->off_head_w, &zero64, nullptr, 0, 0,
                       nullptr, nullptr, d_params + m->off_head_b, &zero64, TC, 1, 0, io->d_logits, TC, stream);
}

static int check_desc(const gcmi_model_desc* m) {
  GCMI_CHECK_ARG(m != nullptr, "model desc is NULL");
  GCMI_CHECK_ARG(m->n_layers >= 1 && m->n_layers <= kMaxL, "n_layers %d outside [1,%d]", m->n_layers, kMaxL);
  GCMI_CHECK_ARG(m->max_deg >= 0 && m->max_deg <= GCMI_MAX_DEG, "bad max_deg");
  GCMI_CHECK_ARG(m->n_feat_in > 0 && m->dense_width > 0 && m->n_tasks > 0 && m->n_classes > 0, "bad widths");
  GCMI_CHECK_ARG(m->mode == 0 || m->mode == 1, "mode must be 0 (classification) or 1 (regression)");
  GCMI_CHECK_ARG(m->mode == 0 || m->n_classes == 1, "regression needs n_classes == 1");
  for (int l = 0; l < m->n_layers; ++l) GCMI_CHECK_ARG(m->conv_width[l] > 0, "bad conv width");
  if (m->storage != 0) {
    // bf16 activation storage in the streaming kernels (fwd_bf16.hip, bwd_fused.hip HB, gather_lds.hip *OpH): the
    // default shapes -- GraphConv widths 64 over 65..80 input columns, dense width 128, BatchNorm on
    // (2 = the gradient streams between the kernels are bf16 as well)
    bool ok = (m->storage == 1 || m->storage == 2) && m->batch_norm && m->dense_width == 128 && m->n_feat_in > 64 &&
              m->n_feat_in <= 80;
    for (int l = 0; l < m->n_layers; ++l) ok = ok && m->conv_width[l] == 64;
    if (!ok) {
      set_error("gcmi_model_*: bf16 activation storage covers graph_conv_layers of width 64 over 65..80 atom features, "
                "dense_layer_size 128 and batch_normalize=True (other shapes: gcmi_small_* or storage 0)");
      return GCMI_ERR_UNSUPPORTED;
    }
  }
  return GCMI_OK;
}

struct Segs {
  int32_t begin[GCMI_MAX_DEG + 1], end[GCMI_MAX_DEG + 1];
  int64_t w_rel[GCMI_MAX_DEG + 1], w_self[GCMI_MAX_DEG + 1], b_off[GCMI_MAX_DEG + 1];
  int n;
};

static Segs make_segs(const gcmi_graph* g, int64_t k, int64_t width) {
  Segs s;
  s.n = g->max_deg + 1;
  const int64_t blk = k * width;
  for (int d = 0; d <= g->max_deg; ++d) {
    s.begin[d] = g->deg_start[d];
    s.end[d] = g->deg_start[d + 1];
    s.w_rel[d] = d == 0 ? -1 : (int64_t)(2 * (d - 1)) * blk;
    s.w_self[d] = d == 0 ? (int64_t)(2 * g->max_deg) * blk : (int64_t)(2 * (d - 1) + 1) * blk;
    s.b_off[d] = (int64_t)d * width;
  }
  return s;
}

static int model_forward_h(const gcmi_model_desc* m, const gcmi_graph* g, const float* d_params, const gcmi_model_io* io,
                           int32_t training, void* stream);
static int model_loss_backward_h(const gcmi_model_desc* m, const gcmi_graph* g, const float* d_params, float* d_grads,
                                 const gcmi_model_io* io, const float* d_labels, const float* d_weights, int64_t n_rows,
                                 int64_t* grad_lo, int64_t* grad_hi, void* stream);

#define RUN(call)            \
  do {                       \
    int rc__ = (call);       \
    if (rc__) return rc__;   \
  } while (0)


// ------------------------------------------------------------------------------------------------------------------
// storage == 1: the same step with every matrix it writes and reads back kept as bf16 (fp32 arithmetic, fp64
// statistics, fp32 parameters / gradients / gradient streams).  One linear sequence of the kernels that have a bf16
// form; what they do not cover (other widths, no BatchNorm, graphs without window plans or reverse slots, the
// exact-fp32 product mode) is refused, never computed some other way.
static int require_h(const gcmi_model_desc* m, const gcmi_graph* g, const gcmi_model_io* io, bool backward) {
  if (gemm_exact_mode()) {
    set_error("bf16 activation storage: not available in the exact-fp32 product mode (set_gemm_mode('fast'))");
    return GCMI_ERR_UNSUPPORTED;
  }
  if (g->n_atoms > 0 && (!win_usable_h(g, 64) || !win_usable_h(g, 80) || !win_usable(g, (int)up4(m->n_feat_in), false))) {
    set_error("bf16 activation storage: the graph carries no usable molecule-window plan (collate with gcmi_collate_plans)");
    return GCMI_ERR_UNSUPPORTED;
  }
  if (g->n_atoms > 0 && (io->ld_features % 4 != 0 || io->ld_features < up4(m->n_feat_in) || !aligned16(io->d_atom_features))) {
    set_error("bf16 activation storage: atom feature rows must be 16-byte addressable and padded to %d columns",
              (int)up4(m->n_feat_in));
    return GCMI_ERR_UNSUPPORTED;
  }
  if (backward && g->n_atoms > 0) {
    if (!(g->d_rev_pos != nullptr || g->n_edges == 0)) {
      set_error("bf16 activation storage: the backward needs reverse slots (every bond listed from both ends)");
      return GCMI_ERR_UNSUPPORTED;
    }
    if (!fused_bwd_enabled()) {
      set_error("bf16 activation storage: the one-pass block kernels are switched off (GCMI_OPT_FUSED_BWD)");
      return GCMI_ERR_UNSUPPORTED;
    }
    if (m->storage == 2 && !win_usable_gh(g, 64)) {
      set_error("bf16 gradient streams: the graph carries no usable molecule-window plan");
      return GCMI_ERR_UNSUPPORTED;
    }
  }
  return GCMI_OK;
}

static int model_forward_h(const gcmi_model_desc* m, const gcmi_graph* g, const float* d_params, const gcmi_model_io* io,
                           int32_t training, void* stream) {
  RUN(require_h(m, g, io, false));
  hipStream_t st = (hipStream_t)stream;
  const int L = m->n_layers;
  const int64_t N = g->n_atoms, B = g->n_mols;
  const Ws w = carve(m, N, B, io->ld_features);
  float* ws = io->d_workspace;
  auto H = [&](int64_t off) { return reinterpret_cast<bf16_t*>(ws + off); };
  if (training && N > 0 && hipMemsetAsync(ws + w.acc, 0, sizeof(float) * (size_t)(w.z_end - w.acc), st) != hipSuccess) {
    set_error("model_forward: memset failed");
    return GCMI_ERR_LAUNCH;
  }
  const bf16_t* xin = nullptr;
  int64_t ldin = 0;
  bool sum_done = false;  // S[l] was written by the fused pass of block l - 1
  RUN(pack_biases(m, w, ws, d_params, st));
  for (int l = 0; l < L; ++l) {
    const int K = l == 0 ? m->n_feat_in : m->conv_width[l - 1];
    const int W = m->conv_width[l];
    const Segs sg = make_segs(g, K, W);
    float* bnv = ws + w.bnv[l];
    float* scale = bnv + 2 * W;
    float* shift = bnv + 3 * W;
    if (N > 0) {
      {
        TimedScope ts(GCMI_K_GATHER_SUM, st);
        if (l == 0) {  // fp32 atom features -> bf16 neighbour sums + a bf16 copy of the rows themselves
          RUN(win_gather_sum_fh(g, io->d_atom_features, io->ld_features, (int)w.ngather[0], H(w.S[0]), H(w.xb), w.ldS[0], st));
          xin = H(w.xb);
          ldin = w.ldS[0];
        } else if (!sum_done) {
          RUN(win_gather_sum_h(g, xin, ldin, K, H(w.S[l]), w.ldS[l], st));
        }
      }
      {
        TimedScope ts(GCMI_K_SEG_GEMM, st);
        const int rc = fwd_h_gemm(sg.n, sg.begin, sg.end, H(w.S[l]), w.ldS[l], K, d_params + m->off_conv_w[l], sg.w_rel, xin,
                                  ldin, K, d_params + m->off_conv_w[l], sg.w_self, ws + w.bsum[l], sg.b_off, W, 0, 1,
                                  H(w.gc[l]), W, training ? reinterpret_cast<double*>(ws + w.acc) : nullptr, ws + w.wimg,
                                  st);
        if (rc == GCMI_ERR_UNSUPPORTED) set_error("bf16 activation storage: GraphConv %d has no bf16 product kernel", l);
        RUN(rc);
      }
      if (training) {
        RUN(bn_finalize_impl(N, W, d_params + m->off_bn_gamma[l], d_params + m->off_bn_beta[l], m->bn_eps, m->bn_momentum,
                             io->d_bn_running_mean[l], io->d_bn_running_var[l], bnv, bnv + W, scale, shift,
                             reinterpret_cast<double*>(ws + w.acc), stream, io->d_bn_batches_tracked[l]));
      } else {
        RUN(gcmi_bn_fold_eval(d_params + m->off_bn_gamma[l], d_params + m->off_bn_beta[l], io->d_bn_running_mean[l],
                              io->d_bn_running_var[l], m->bn_eps, W, scale, shift, stream));
      }
      {
        TimedScope ts(GCMI_K_GATHER_MAX, st);
        // GraphPool of this block, and where another GraphConv follows its neighbour sums in the same window pass
        sum_done = l + 1 < L && m->conv_width[l + 1] == W && win_max_sum_usable_h(g, W);
        if (sum_done)
          RUN(win_gather_max_sum_h(g, H(w.gc[l]), W, W, scale, shift, H(w.pool[l]), W,
                                   training ? reinterpret_cast<uint8_t*>(ws + w.arg[l]) : nullptr, H(w.S[l + 1]),
                                   w.ldS[l + 1], st));
        else
          RUN(win_gather_max_h(g, H(w.gc[l]), W, W, scale, shift, H(w.pool[l]), W,
                               training ? reinterpret_cast<uint8_t*>(ws + w.arg[l]) : nullptr, st));
      }
    }
    xin = H(w.pool[l]);
    ldin = W;
  }
  const int Wl = m->conv_width[L - 1];
  const int D = m->dense_width;
  const int32_t zero32 = 0;
  const int64_t zero64 = 0;
  float* bnvD = ws + w.bnv[L];
  float* scale = bnvD + 2 * D;
  float* shift = bnvD + 3 * D;
  if (N > 0) {
    const int32_t nN = (int32_t)N;
    {
      TimedScope ts(GCMI_K_SEG_GEMM, st);
      const int rc = fwd_h_gemm(1, &zero32, &nN, xin, ldin, Wl, d_params + m->off_dense_w, &zero64, nullptr, 0, 0, nullptr,
                                nullptr, d_params + m->off_dense_b, &zero64, D, 1, 1, H(w.dense), D,
                                training ? reinterpret_cast<double*>(ws + w.acc) : nullptr, ws + w.wimg, st);
      if (rc == GCMI_ERR_UNSUPPORTED) set_error("bf16 activation storage: the dense layer has no bf16 product kernel");
      RUN(rc);
    }
    if (training) {
      RUN(bn_finalize_impl(N, D, d_params + m->off_bn_gamma[L], d_params + m->off_bn_beta[L], m->bn_eps, m->bn_momentum,
                           io->d_bn_running_mean[L], io->d_bn_running_var[L], bnvD, bnvD + D, scale, shift,
                           reinterpret_cast<double*>(ws + w.acc), stream, io->d_bn_batches_tracked[L]));
    } else {
      RUN(gcmi_bn_fold_eval(d_params + m->off_bn_gamma[L], d_params + m->off_bn_beta[L], io->d_bn_running_mean[L],
                            io->d_bn_running_var[L], m->bn_eps, D, scale, shift, stream));
    }
  }
  RUN(readout_fwd_impl(g, reinterpret_cast<const float*>(H(w.dense)), D, D, N > 0 ? scale : nullptr,
                       N > 0 ? shift : nullptr, 1, io->d_fingerprint, 2 * D, reinterpret_cast<int32_t*>(ws + w.arg_r),
                       training ? ws + w.rsum : nullptr, stream, 1));
  const int TC = m->n_tasks * m->n_classes;
  const int32_t nB = (int32_t)B;
  RUN(head_forward(m, w, ws, d_params, io, B, stream));
  if (m->mode == 0 && io->d_probs) RUN(gcmi_softmax(io->d_logits, B * m->n_tasks, m->n_classes, io->d_probs, stream));
  if (training && N == 0) {  // (with atoms, every layer's statistics launch bumps its own counter)
    CounterPtrs c;
    c.n = L + 1;
    bool any = false;
    for (int i = 0; i <= kMaxL; ++i) {
      c.p[i] = i <= L ? io->d_bn_batches_tracked[i] : nullptr;
      any = any || c.p[i] != nullptr;
    }
    if (any) {
      hipLaunchKernelGGL(bump_counters_kernel, dim3(1), dim3(64), 0, st, c);
      GCMI_CHECK_LAUNCH("bump_counters");
    }
  }
  return GCMI_OK;
}

static int model_loss_backward_h(const gcmi_model_desc* m, const gcmi_graph* g, const float* d_params, float* d_grads,
                                 const gcmi_model_io* io, const float* d_labels, const float* d_weights, int64_t n_rows,
                                 int64_t* grad_lo, int64_t* grad_hi, void* stream) {
  RUN(require_h(m, g, io, true));
  hipStream_t st = (hipStream_t)stream;
  const int L = m->n_layers;
  const int64_t N = g->n_atoms, B = g->n_mols;
  const Ws w = carve(m, N, B, io->ld_features);
  float* ws = io->d_workspace;
  auto H = [&](int64_t off) { return reinterpret_cast<bf16_t*>(ws + off); };
  auto HF = [&](int64_t off) { return reinterpret_cast<const float*>(ws + off); };  // a bf16 matrix behind a float* parameter
  const int D = m->dense_width;
  const int TC = m->n_tasks * m->n_classes;
  const bool full = m->grad_mode == 1;
  const int64_t lo = full ? 0 : m->off_bn_gamma[L - 1];
  const int64_t hi = m->n_params;
  if (grad_lo) *grad_lo = lo;
  if (grad_hi) *grad_hi = hi;
  if (hipMemsetAsync(d_grads + lo, 0, sizeof(float) * (size_t)(hi - lo), st) != hipSuccess ||
      hipMemsetAsync(ws + w.dlogits, 0, sizeof(float) * (size_t)(w.z_end - w.dlogits), st) != hipSuccess) {
    set_error("model_loss_backward: memset failed");
    return GCMI_ERR_LAUNCH;
  }
  const int32_t zero32 = 0;
  const int64_t zero64 = 0;
  const int32_t nB = (int32_t)B;
  // ---- per-molecule part (fp32 throughout: the fingerprint and everything behind it are per-molecule rows)
  const float* bnvL = ws + w.bnv[L];
  bool head_sums = false;
  {
    const int rc = head_bwd_fused(m->mode == 0 ? 0 : 1, io->d_logits, d_labels, d_weights, n_rows, m->n_tasks, m->n_classes,
                                  B, io->d_fingerprint, 2 * D, d_params + m->off_head_w, d_grads + m->off_head_w,
                                  d_grads + m->off_head_b, ws + w.dfp, 2 * D, reinterpret_cast<double*>(ws + w.lacc),
                                  g->d_mol_runs, g->max_deg + 1, reinterpret_cast<const int32_t*>(ws + w.arg_r), ws + w.rsum,
                                  bnvL, bnvL + D, (N > 0 && g->d_mol_runs) ? reinterpret_cast<double*>(ws + w.acc) : nullptr,
                                  D, st, ws + w.dlogits, w.himg >= 0 ? ws + w.himg : nullptr);
    if (rc == GCMI_OK) {
      head_sums = N > 0 && g->d_mol_runs != nullptr;
      // (with head_sums the loss is finalised by the BatchNorm parameter launch that follows)
      if (!head_sums)
        RUN(loss_finalize_impl(reinterpret_cast<double*>(ws + w.lacc), 1.f / (float)(n_rows * m->n_tasks), io->d_loss, stream, kLossRep));
    } else if (rc != GCMI_ERR_UNSUPPORTED) {
      return rc;
    } else {
      RUN(loss_impl(m->mode == 0 ? 0 : 1, io->d_logits, d_labels, d_weights, n_rows, m->n_tasks, m->n_classes, io->d_loss,
                    ws + w.dlogits, nullptr, reinterpret_cast<double*>(ws + w.lacc), true, stream));
      RUN(gcmi_seg_gemm_wgrad(1, &zero32, &nB, io->d_fingerprint, 2 * D, 2 * D, ws + w.dlogits, TC, TC,
                              d_grads + m->off_head_w, &zero64, d_grads + m->off_head_b, &zero64, 1, stream));
      RUN(gcmi_seg_gemm(1, &zero32, &nB, ws + w.dlogits, TC, TC, d_params + m->off_head_w, &zero64, nullptr, 0, 0, nullptr,
                        nullptr, nullptr, nullptr, 2 * D, 0, 0, ws + w.dfp, 2 * D, stream));
      RUN(readout_grad_prep(ws + w.dfp, 2 * D, io->d_fingerprint, 2 * D, B, D, st));
    }
  }
  if (N == 0) return GCMI_OK;
  const int Wl = m->conv_width[L - 1];
  float* dpool = ws + w.tC;
  const float* coef = ws + w.acc;
  // storage == 2: dpool, dy, dS and dXs are bf16 rows (in the same fp32-sized workspace blocks, ld in elements)
  const bool gb = m->storage == 2;
  auto HG = [](float* p) { return reinterpret_cast<bf16_t*>(p); };
  // ---- dense block: BatchNorm sums from per-molecule data, then one pass (dense and pool rows arrive as bf16)
  if (head_sums) {
    RUN(bn_bwd_params_impl(N, D, d_params + m->off_bn_gamma[L], bnvL, bnvL + D, d_grads + m->off_bn_gamma[L],
                           d_grads + m->off_bn_beta[L], reinterpret_cast<double*>(ws + w.acc), stream,
                           reinterpret_cast<double*>(ws + w.lacc), kLossRep, 1.f / (float)(n_rows * m->n_tasks), io->d_loss));
  } else {
    // (the per-molecule sums kernel reads rawsum, never the atom rows: the bf16 matrix is only passed through)
    RUN(bn_bwd_readout_impl(g->d_membership, ws + w.dfp, 2 * D, reinterpret_cast<const int32_t*>(ws + w.arg_r),
                            HF(w.dense), D, N, D, d_params + m->off_bn_gamma[L], bnvL, bnvL + D,
                            d_grads + m->off_bn_gamma[L], d_grads + m->off_bn_beta[L], nullptr, D, 1,
                            reinterpret_cast<double*>(ws + w.acc), true, stream, ws + w.rsum, g->d_mol_runs, g->n_mols,
                            g->max_deg + 1));
  }
  {
    TimedScope ts(GCMI_K_FUSED_BWD, st);
    const int rc = fused_dense_bwd(N, g->d_membership, ws + w.dfp, 2 * D, reinterpret_cast<const int32_t*>(ws + w.arg_r),
                                   HF(w.dense), D, coef, D, HF(w.pool[L - 1]), Wl, Wl, d_params + m->off_dense_w,
                                   d_grads + m->off_dense_w, d_grads + m->off_dense_b, dpool, Wl,
                                   reinterpret_cast<double*>(ws + w.acc2), st, gb ? 2 : 1);
    if (rc == GCMI_ERR_UNSUPPORTED) set_error("bf16 activation storage: the dense block has no one-pass backward");
    RUN(rc);
  }
  // ---- GraphConv / BatchNorm / GraphPool blocks, last to first (gradient streams fp32: the window kernels as they are)
  bool dy_ready = false;
  BiasLayers ub;
  memset(&ub, 0, sizeof(ub));
  for (int l = L - 1; l >= 0; --l) {
    const int W = m->conv_width[l];
    const int K = l == 0 ? m->n_feat_in : m->conv_width[l - 1];
    float* dy = ws + w.tD;
    const Segs sg = make_segs(g, K, W);
    const bf16_t* xin = l == 0 ? H(w.xb) : H(w.pool[l - 1]);
    const int64_t ldx = l == 0 ? w.ldS[0] : m->conv_width[l - 1];
    float* dS = ws + w.tE;
    float* dX = ws + w.tC;
    const float* bnv = ws + w.bnv[l];
    // the block above left sum dP and sum dP * P: this BatchNorm's backward needs no pass over dy (bn_bwd_pool_impl);
    // dy itself only when the GraphConv below trains, or where the pooled sums are ill-conditioned
    if (dy_ready) {
      // (left by win_gather_sumacc_max_bwd below, one iteration ago)
    } else if (gb) {
      TimedScope ts(GCMI_K_GATHER_MAX_BWD, st);
      RUN(win_gather_max_bwd_h(g, HG(dpool), W, W, reinterpret_cast<const uint8_t*>(ws + w.arg[l]), HG(dy), W,
                               full ? nullptr : d_params + m->off_bn_gamma[l], full ? nullptr : d_params + m->off_bn_beta[l],
                               st));
    } else if (full) {
      RUN(gcmi_gather_max_bwd(g, dpool, W, W, reinterpret_cast<const uint8_t*>(ws + w.arg[l]), dy, W, stream));
    } else {
      TimedScope ts(GCMI_K_GATHER_MAX_BWD, st);
      RUN(win_gather_max_bwd_if_ill(g, dpool, W, W, reinterpret_cast<const uint8_t*>(ws + w.arg[l]), dy, W,
                                    d_params + m->off_bn_gamma[l], d_params + m->off_bn_beta[l], st));
    }
    RUN(bn_bwd_pool_impl(dy, W, HF(w.gc[l]), W, N, W, d_params + m->off_bn_gamma[l], d_params + m->off_bn_beta[l], bnv, bnv + W,
                         d_grads + m->off_bn_gamma[l], d_grads + m->off_bn_beta[l], reinterpret_cast<double*>(ws + w.acc2),
                         reinterpret_cast<double*>(ws + w.acc), stream, gb ? 2 : 1));
    dy_ready = false;
    if (!full) break;  // reference semantics: nothing in front of a GraphConv output trains
    {
      TimedScope ts(GCMI_K_FUSED_BWD, st);
      const int rc = fused_conv_bwd(sg.n, sg.begin, sg.end, sg.w_rel, sg.w_self, sg.b_off, dy, W, HF(w.gc[l]), W, coef, W,
                                    HF(w.S[l]), w.ldS[l], reinterpret_cast<const float*>(xin), ldx, K,
                                    d_params + m->off_conv_w[l], d_grads + m->off_conv_w[l], ws + w.dbsum[l],
                                    l > 0 ? dS : nullptr, K, l > 0 ? dX : nullptr, K,
                                    l > 0 ? reinterpret_cast<double*>(ws + w.acc2) : nullptr, st, gb ? 2 : 1);
      if (rc == GCMI_ERR_UNSUPPORTED) set_error("bf16 activation storage: GraphConv %d has no one-pass backward", l);
      RUN(rc);
    }
    ub.src[l] = ws + w.dbsum[l];  // (unpacked into the reference's bias rows in one launch after the loop)
    ub.dst[l] = d_grads + m->off_conv_b[l];
    ub.width[l] = W;
    if (l == 0) break;  // the atom features need no gradient
    // dX holds the self part; the neighbour part is added onto it, and where the window kernels can hold a third tile
    // the GraphPool backward of the block below runs in the same pass
    if (gb) {
      if (!win_two_stage_usable_h(g, K)) {
        set_error("bf16 gradient streams: no LDS for the two-stage window pass");
        return GCMI_ERR_UNSUPPORTED;
      }
      TimedScope ts(GCMI_K_GATHER_MAX_BWD, st);
      RUN(win_gather_sumacc_max_bwd_h(g, HG(dS), K, K, HG(dX), K, reinterpret_cast<const uint8_t*>(ws + w.arg[l - 1]),
                                      HG(ws + w.tD), K, st));
      dy_ready = true;
    } else if (win_two_stage_usable(g, K) && aligned16(dS) && aligned16(dX) && aligned16(ws + w.tD)) {
      TimedScope ts(GCMI_K_GATHER_MAX_BWD, st);
      RUN(win_gather_sumacc_max_bwd(g, dS, K, K, dX, K, reinterpret_cast<const uint8_t*>(ws + w.arg[l - 1]), ws + w.tD, K, st));
      dy_ready = true;
    } else {
      RUN(gcmi_gather_sum_fwd(g, dS, K, K, dX, K, 1, stream));
    }
    dpool = dX;
  }
  RUN(unpack_bias_grads(m, ub, L, st));
  return GCMI_OK;
}

}  // namespace gcmi

using namespace gcmi;

extern "C" {

int64_t gcmi_model_workspace_floats(const gcmi_model_desc* m, int64_t n_atoms, int64_t n_mols) {
  if (check_desc(m) != GCMI_OK || n_atoms < 0 || n_mols < 0) return -1;
  // ld of the features is not known here: assume the padded width (worst case)
  return carve(m, n_atoms, n_mols, up4(m->n_feat_in)).total;
}

int gcmi_model_forward(const gcmi_model_desc* m, const gcmi_graph* g, const float* d_params,
                       const gcmi_model_io* io, int32_t training, void* stream) {
  RUN(check_desc(m));
  RUN(check_graph(g, true));
  GCMI_CHECK_ARG(io && d_params && io->d_workspace && io->d_logits && io->d_fingerprint,
                 "model_forward: NULL buffer");
  GCMI_CHECK_ARG(g->max_deg == m->max_deg, "graph max_deg %d != model max_deg %d", g->max_deg, m->max_deg);
  GCMI_CHECK_ARG(g->n_mols > 1, "graph_gather requires batches larger than 1");
  GCMI_CHECK_ARG(g->n_atoms == 0 || io->d_atom_features, "model_forward: NULL atom features");
  GCMI_CHECK_ARG(io->ld_features >= m->n_feat_in, "ld_features < n_feat_in");
  if (m->storage >= 1) return model_forward_h(m, g, d_params, io, training, stream);
  hipStream_t st = (hipStream_t)stream;
  const int L = m->n_layers;
  const int64_t N = g->n_atoms, B = g->n_mols;
  const Ws w = carve(m, N, B, io->ld_features);
  float* ws = io->d_workspace;
  const float* x = io->d_atom_features;
  int64_t ldx = io->ld_features;
  if (training && m->batch_norm && N > 0 &&
      hipMemsetAsync(ws + w.acc, 0, sizeof(float) * (size_t)(w.z_end - w.acc), st) != hipSuccess) {
    set_error("model_forward: memset failed");
    return GCMI_ERR_LAUNCH;
  }
  bool stats_fused = false;
  RUN(pack_biases(m, w, ws, d_params, st));
  for (int l = 0; l < L; ++l) {
    const int K = l == 0 ? m->n_feat_in : m->conv_width[l - 1];
    const int W = m->conv_width[l];
    const Segs sg = make_segs(g, K, W);
    stats_fused = false;
    if (N > 0) {
      RUN(gcmi_gather_sum_fwd(g, x, ldx, (int32_t)w.ngather[l], ws + w.S[l], w.ldS[l], 0, stream));
      // training with BatchNorm: the product's epilogue also adds the column sums of its output into the
      // BatchNorm accumulators (clean: zeroed above, self-cleaning afterwards), so the layer output is not read again
      RUN(seg_gemm_stats(sg.n, sg.begin, sg.end, ws + w.S[l], w.ldS[l], K, d_params + m->off_conv_w[l],
                         sg.w_rel, x, ldx, K, d_params + m->off_conv_w[l], sg.w_self, ws + w.bsum[l],
                         sg.b_off, W, 0, 1, ws + w.gc[l], W,
                         (m->batch_norm && training) ? reinterpret_cast<double*>(ws + w.acc) : nullptr,
                         &stats_fused, stream, ws + w.wimg));
    }
    float* scale = nullptr;
    float* shift = nullptr;
    if (m->batch_norm && N > 0) {
      float* bnv = ws + w.bnv[l];
      scale = bnv + 2 * W;
      shift = bnv + 3 * W;
      if (training && stats_fused) {
        RUN(bn_finalize_impl(N, W, d_params + m->off_bn_gamma[l], d_params + m->off_bn_beta[l], m->bn_eps,
                             m->bn_momentum, io->d_bn_running_mean[l], io->d_bn_running_var[l], bnv, bnv + W,
                             scale, shift, reinterpret_cast<double*>(ws + w.acc), stream, io->d_bn_batches_tracked[l]));
      } else if (training) {
        RUN(bn_stats_impl(ws + w.gc[l], W, N, W, d_params + m->off_bn_gamma[l],
                          d_params + m->off_bn_beta[l], m->bn_eps, m->bn_momentum,
                          io->d_bn_running_mean[l], io->d_bn_running_var[l], bnv, bnv + W, scale, shift,
                          reinterpret_cast<double*>(ws + w.acc), true, stream, io->d_bn_batches_tracked[l]));
      } else {
        RUN(gcmi_bn_fold_eval(d_params + m->off_bn_gamma[l], d_params + m->off_bn_beta[l],
                              io->d_bn_running_mean[l], io->d_bn_running_var[l], m->bn_eps, W, scale,
                              shift, stream));
      }
    }
    if (N > 0)
      RUN(gcmi_gather_max_fwd(g, ws + w.gc[l], W, W, scale, shift, ws + w.pool[l], W,
                              training ? reinterpret_cast<uint8_t*>(ws + w.arg[l]) : nullptr, stream));
    x = ws + w.pool[l];
    ldx = W;
  }
  const int Wl = m->conv_width[L - 1];
  const int D = m->dense_width;
  const int32_t zero32 = 0;
  const int64_t zero64 = 0;
  if (N > 0) {
    const int32_t nN = (int32_t)N;
    RUN(seg_gemm_stats(1, &zero32, &nN, x, ldx, Wl, d_params + m->off_dense_w, &zero64, nullptr, 0, 0,
                       nullptr, nullptr, d_params + m->off_dense_b, &zero64, D, 1, 1, ws + w.dense, D,
                       (m->batch_norm && training) ? reinterpret_cast<double*>(ws + w.acc) : nullptr,
                       &stats_fused, stream));
  }
  float* scale = nullptr;
  float* shift = nullptr;
  if (m->batch_norm && N > 0) {
    float* bnv = ws + w.bnv[L];
    scale = bnv + 2 * D;
    shift = bnv + 3 * D;
    if (training && stats_fused) {
      RUN(bn_finalize_impl(N, D, d_params + m->off_bn_gamma[L], d_params + m->off_bn_beta[L], m->bn_eps,
                           m->bn_momentum, io->d_bn_running_mean[L], io->d_bn_running_var[L], bnv, bnv + D, scale,
                           shift, reinterpret_cast<double*>(ws + w.acc), stream, io->d_bn_batches_tracked[L]));
    } else if (training) {
      RUN(bn_stats_impl(ws + w.dense, D, N, D, d_params + m->off_bn_gamma[L], d_params + m->off_bn_beta[L],
                        m->bn_eps, m->bn_momentum, io->d_bn_running_mean[L], io->d_bn_running_var[L], bnv,
                        bnv + D, scale, shift, reinterpret_cast<double*>(ws + w.acc), true, stream,
                        io->d_bn_batches_tracked[L]));
    } else {
      RUN(gcmi_bn_fold_eval(d_params + m->off_bn_gamma[L], d_params + m->off_bn_beta[L],
                            io->d_bn_running_mean[L], io->d_bn_running_var[L], m->bn_eps, D, scale, shift,
                            stream));
    }
  }
  RUN(readout_fwd_impl(g, ws + w.dense, D, D, scale, shift, 1, io->d_fingerprint, 2 * D,
                       reinterpret_cast<int32_t*>(ws + w.arg_r), (training && m->batch_norm) ? ws + w.rsum : nullptr,
                       stream));
  const int TC = m->n_tasks * m->n_classes;
  const int32_t nB = (int32_t)B;
  RUN(head_forward(m, w, ws, d_params, io, B, stream));
  if (m->mode == 0 && io->d_probs)
    RUN(gcmi_softmax(io->d_logits, B * m->n_tasks, m->n_classes, io->d_probs, stream));
  if (training && m->batch_norm && N == 0) {  // (with atoms, every layer's statistics launch bumps its own counter)
    CounterPtrs c;
    c.n = L + 1;
    bool any = false;
    for (int i = 0; i <= kMaxL; ++i) {
      c.p[i] = i <= L ? io->d_bn_batches_tracked[i] : nullptr;
      any = any || c.p[i] != nullptr;
    }
    if (any) {
      hipLaunchKernelGGL(bump_counters_kernel, dim3(1), dim3(64), 0, st, c);
      GCMI_CHECK_LAUNCH("bump_counters");
    }
  }
  return GCMI_OK;
}

int gcmi_model_loss_backward(const gcmi_model_desc* m, const gcmi_graph* g, const float* d_params,
                             float* d_grads, const gcmi_model_io* io, const float* d_labels,
                             const float* d_weights, int64_t n_rows, int64_t* grad_lo,
                             int64_t* grad_hi, void* stream) {
  RUN(check_desc(m));
  RUN(check_graph(g, true));
  GCMI_CHECK_ARG(io && d_params && d_grads && io->d_workspace && io->d_logits && io->d_fingerprint &&
                     io->d_loss && d_labels,
                 "model_loss_backward: NULL buffer");
  GCMI_CHECK_ARG(n_rows > 0 && n_rows <= g->n_mols, "n_rows %lld outside (0, n_mols=%d]", (long long)n_rows,
                 g->n_mols);
  if (m->storage >= 1)
    return model_loss_backward_h(m, g, d_params, d_grads, io, d_labels, d_weights, n_rows, grad_lo, grad_hi, stream);
  hipStream_t st = (hipStream_t)stream;
  const int L = m->n_layers;
  const int64_t N = g->n_atoms, B = g->n_mols;
  const Ws w = carve(m, N, B, io->ld_features);
  float* ws = io->d_workspace;
  const int D = m->dense_width;
  const int TC = m->n_tasks * m->n_classes;
  const bool full = m->grad_mode == 1;
  const bool sym = g->d_rev_pos != nullptr || g->n_edges == 0;
  const int64_t lo = full ? 0 : (m->batch_norm ? m->off_bn_gamma[L - 1] : m->off_dense_w);
  const int64_t hi = m->n_params;
  if (grad_lo) *grad_lo = lo;
  if (grad_hi) *grad_hi = hi;
  auto zero = [&](void* p, size_t bytes) -> int {
    if (bytes && hipMemsetAsync(p, 0, bytes, st) != hipSuccess) {
      set_error("model_loss_backward: memset failed");
      return GCMI_ERR_LAUNCH;
    }
    return GCMI_OK;
  };
  RUN(zero(d_grads + lo, sizeof(float) * (size_t)(hi - lo)));
  // dlogits (rows beyond n_rows carry no gradient), the bias-gradient sums and every accumulator
  RUN(zero(ws + w.dlogits, sizeof(float) * (size_t)(w.z_end - w.dlogits)));
  const int32_t zero32 = 0;
  const int64_t zero64 = 0;
  const int32_t nB = (int32_t)B, nN = (int32_t)N;
  // ---- per-molecule part in one kernel where the shapes allow (head_bwd.hip): loss, d logits, head gradients, the
  // gradient w.r.t. GraphGather's pre-activation (tanh derivative applied), and -- when the one-pass dense block
  // follows -- the BatchNorm backward sums of the dense layer
  const bool dense_fused_next = m->batch_norm && N > 0 && fused_bwd_enabled() && D == 128 &&
                                m->conv_width[L - 1] > 32 && m->conv_width[L - 1] <= 64;
  bool head_done = false, head_sums = false;
  if (m->batch_norm) {  // (without BatchNorm the readout backward applies the tanh derivative itself)
    const float* bnvL = ws + w.bnv[L];
    const int rc = head_bwd_fused(m->mode == 0 ? 0 : 1, io->d_logits, d_labels, d_weights, n_rows, m->n_tasks,
                                  m->n_classes, B, io->d_fingerprint, 2 * D, d_params + m->off_head_w,
                                  d_grads + m->off_head_w, d_grads + m->off_head_b, ws + w.dfp, 2 * D,
                                  reinterpret_cast<double*>(ws + w.lacc), g->d_mol_runs, g->max_deg + 1,
                                  reinterpret_cast<const int32_t*>(ws + w.arg_r), ws + w.rsum, bnvL, bnvL + D,
                                  (dense_fused_next && g->d_mol_runs) ? reinterpret_cast<double*>(ws + w.acc) : nullptr, D,
                                  st, ws + w.dlogits, w.himg >= 0 ? ws + w.himg : nullptr);
    if (rc == GCMI_OK) {
      head_done = true;
      head_sums = dense_fused_next && g->d_mol_runs != nullptr;
      // (with head_sums the loss is finalised by the BatchNorm parameter launch that follows)
      if (!head_sums)
        RUN(loss_finalize_impl(reinterpret_cast<double*>(ws + w.lacc), 1.f / (float)(n_rows * m->n_tasks), io->d_loss,
                               stream, kLossRep));
    } else if (rc != GCMI_ERR_UNSUPPORTED) {
      return rc;
    }
  }
  if (!head_done) {
    // ---- loss on the first n_rows molecules
    RUN(loss_impl(m->mode == 0 ? 0 : 1, io->d_logits, d_labels, d_weights, n_rows, m->n_tasks,
                  m->n_classes, io->d_loss, ws + w.dlogits, nullptr,
                  reinterpret_cast<double*>(ws + w.lacc), true, stream));
    // ---- head
    RUN(gcmi_seg_gemm_wgrad(1, &zero32, &nB, io->d_fingerprint, 2 * D, 2 * D, ws + w.dlogits, TC, TC,
                            d_grads + m->off_head_w, &zero64, d_grads + m->off_head_b, &zero64, 1, stream));
    RUN(gcmi_seg_gemm(1, &zero32, &nB, ws + w.dlogits, TC, TC, d_params + m->off_head_w, &zero64, nullptr, 0,
                      0, nullptr, nullptr, nullptr, nullptr, 2 * D, 0, 0, ws + w.dfp, 2 * D, stream));
  }
  if (N == 0) return GCMI_OK;
  // ---- readout (+ folded BatchNorm of the dense layer, + its ReLU mask)
  float* dyD = ws + w.tA;   // grad w.r.t. the (normalised) readout input
  float* dxD = ws + w.tB;   // grad w.r.t. the dense pre-activation
  const int Wl = m->conv_width[L - 1];
  float* dpool = ws + w.tC;  // grad w.r.t. the output of the last GraphPool
  const float* coef = ws + w.acc;  // [A | B | C] of the BatchNorm backward just computed (bn.hip: head of its scratch)
  bool dense_done = false;
  bool have_psums = false;  // acc2 holds sum dP, sum dP * P for the BatchNorm below the block just processed
  static const bool psums_env = !(getenv("GCMI_FUSED_PSUMS") && atoi(getenv("GCMI_FUSED_PSUMS")) == 0);
  static const bool two_stage_env = !(getenv("GCMI_TWO_STAGE_GATHER") && atoi(getenv("GCMI_TWO_STAGE_GATHER")) == 0);
  bool dy_ready = false;  // the gather of the block above already left this block's dy (two-stage window pass)
  if (m->batch_norm) {
    // GraphGather backward is recomputed inside the BatchNorm backward from the per-molecule
    // gradient (tanh derivative applied in place): the N x D gradient is never written or re-read
    if (!head_done) RUN(readout_grad_prep(ws + w.dfp, 2 * D, io->d_fingerprint, 2 * D, B, D, st));
    const float* bnv = ws + w.bnv[L];
    const bool try_fused = fused_bwd_enabled() && D == 128 && Wl > 32 && Wl <= 64;
    if (head_sums && !try_fused)  // (cannot happen: head_sums is computed from the same conditions)
      RUN(loss_finalize_impl(reinterpret_cast<double*>(ws + w.lacc), 1.f / (float)(n_rows * m->n_tasks), io->d_loss,
                             stream, kLossRep));
    if (head_sums && try_fused) {
      // the sums are in place (head_bwd.hip): dgamma, dbeta and the coefficient vectors
      RUN(bn_bwd_params_impl(N, D, d_params + m->off_bn_gamma[L], bnv, bnv + D, d_grads + m->off_bn_gamma[L],
                             d_grads + m->off_bn_beta[L], reinterpret_cast<double*>(ws + w.acc), stream,
                             reinterpret_cast<double*>(ws + w.lacc), kLossRep, 1.f / (float)(n_rows * m->n_tasks),
                             io->d_loss));
    } else {
      RUN(bn_bwd_readout_impl(g->d_membership, ws + w.dfp, 2 * D, reinterpret_cast<const int32_t*>(ws + w.arg_r),
                              ws + w.dense, D, N, D, d_params + m->off_bn_gamma[L], bnv, bnv + D,
                              d_grads + m->off_bn_gamma[L], d_grads + m->off_bn_beta[L], try_fused ? nullptr : dxD, D, 1,
                              reinterpret_cast<double*>(ws + w.acc), true, stream, ws + w.rsum, g->d_mol_runs,
                              g->n_mols, g->max_deg + 1));
    }
    if (try_fused) {
      // one pass: dxD formed per 64-row tile in LDS, dW_dense += dxD^T pool, db += colsum, dpool = dxD W_dense
      TimedScope ts(GCMI_K_FUSED_BWD, st);
      const int rc = fused_dense_bwd(N, g->d_membership, ws + w.dfp, 2 * D,
                                     reinterpret_cast<const int32_t*>(ws + w.arg_r), ws + w.dense, D, coef, D,
                                     ws + w.pool[L - 1], Wl, Wl, d_params + m->off_dense_w, d_grads + m->off_dense_w,
                                     d_grads + m->off_dense_b, dpool, Wl,
                                     psums_env ? reinterpret_cast<double*>(ws + w.acc2) : nullptr, st);
      if (rc == GCMI_OK) {
        dense_done = true;
        have_psums = psums_env;
      }
      else if (rc != GCMI_ERR_UNSUPPORTED) return rc;
      else  // not covered after all (misaligned buffers): the separate pass, with its sums once more
        RUN(bn_bwd_readout_impl(g->d_membership, ws + w.dfp, 2 * D, reinterpret_cast<const int32_t*>(ws + w.arg_r),
                                ws + w.dense, D, N, D, d_params + m->off_bn_gamma[L], bnv, bnv + D,
                                d_grads + m->off_bn_gamma[L], d_grads + m->off_bn_beta[L], dxD, D, 1,
                                reinterpret_cast<double*>(ws + w.acc), true, stream, ws + w.rsum, g->d_mol_runs,
                                g->n_mols, g->max_deg + 1));
    }
  } else {
    RUN(gcmi_readout_bwd(g, ws + w.dfp, 2 * D, io->d_fingerprint, 2 * D, D, 1,
                         reinterpret_cast<const int32_t*>(ws + w.arg_r), dyD, D, stream));
    RUN(gcmi_relu_bwd(dyD, D, ws + w.dense, D, N, D, stream));
    dxD = dyD;
  }
  // ---- dense layer
  if (!dense_done) {
    RUN(gcmi_seg_gemm_wgrad(1, &zero32, &nN, ws + w.pool[L - 1], Wl, Wl, dxD, D, D, d_grads + m->off_dense_w,
                            &zero64, d_grads + m->off_dense_b, &zero64, 1, stream));
    RUN(gcmi_seg_gemm(1, &zero32, &nN, dxD, D, D, d_params + m->off_dense_w, &zero64, nullptr, 0, 0, nullptr,
                      nullptr, nullptr, nullptr, Wl, 0, 0, dpool, Wl, stream));
  }
  // ---- GraphConv / BatchNorm / GraphPool blocks, last to first
  BiasLayers ub;
  memset(&ub, 0, sizeof(ub));
  for (int l = L - 1; l >= 0; --l) {
    if (!full && !m->batch_norm) break;  // nothing trainable in front of the dense layer
    const int W = m->conv_width[l];
    const int K = l == 0 ? m->n_feat_in : m->conv_width[l - 1];
    float* dy = ws + w.tD;  // grad w.r.t. the (normalised) pool input
    float* dgc = ws + w.tA;  // grad w.r.t. the GraphConv pre-activation
    const Segs sg = make_segs(g, K, W);
    const float* xin = l == 0 ? io->d_atom_features : ws + w.pool[l - 1];
    const int64_t ldx = l == 0 ? io->ld_features : m->conv_width[l - 1];
    float* dS = ws + w.tE;
    float* dX = ws + w.tC;
    // the fused pass covers the default widths in split-bf16 mode; it wants 16-byte rows of every operand
    const bool try_fused = full && fused_bwd_enabled() && W == 64 && ldx % 4 == 0 && aligned16(xin) &&
                           ((l == 0 && K > 32 && K <= 96) || (l > 0 && K > 32 && K <= 64));
    // GraphPool backward; when nothing needs the elementwise BatchNorm pass afterwards, the window kernel also
    // takes the column sums of the BatchNorm backward from the rows it stores (dy is then not read again for them)
    static const bool stats_env = getenv("GCMI_STATS_IN_GATHER") && atoi(getenv("GCMI_STATS_IN_GATHER")) != 0;
    const bool stats_in_gather = stats_env && m->batch_norm && sym && fused_bwd_enabled() && (try_fused || !full) &&
                                 win_stats_usable(g, W) && aligned16(dpool) && aligned16(dy) && aligned16(ws + w.gc[l]);
    if (!sym) RUN(zero(dy, sizeof(float) * (size_t)(N * W)));
    const bool pool_sums = have_psums && m->batch_norm && sym && (try_fused || !full) && win_usable(g, W, true) &&
                           win_has_width(W);
    have_psums = false;
    if (pool_sums) {
      // The block above left sum dP and sum dP * P: this BatchNorm's backward needs no pass over dy (bn.hip,
      // bn_bwd_pool_impl).  dy itself is needed when the GraphConv below trains; otherwise only by the
      // ill-conditioned fallback, and the kernel returns at once unless that applies.
      const float* bnv = ws + w.bnv[l];
      if (dy_ready) {
        // (left by win_gather_sumacc_max_bwd below, one iteration ago)
      } else if (full) {
        RUN(gcmi_gather_max_bwd(g, dpool, W, W, reinterpret_cast<const uint8_t*>(ws + w.arg[l]), dy, W, stream));
      } else {
        TimedScope ts(GCMI_K_GATHER_MAX_BWD, st);
        RUN(win_gather_max_bwd_if_ill(g, dpool, W, W, reinterpret_cast<const uint8_t*>(ws + w.arg[l]), dy, W,
                                      d_params + m->off_bn_gamma[l], d_params + m->off_bn_beta[l], st));
      }
      RUN(bn_bwd_pool_impl(dy, W, ws + w.gc[l], W, N, W, d_params + m->off_bn_gamma[l], d_params + m->off_bn_beta[l], bnv,
                           bnv + W, d_grads + m->off_bn_gamma[l], d_grads + m->off_bn_beta[l],
                           reinterpret_cast<double*>(ws + w.acc2), reinterpret_cast<double*>(ws + w.acc), stream));
    } else if (stats_in_gather && !dy_ready) {
      const float* bnv = ws + w.bnv[l];
      {
        TimedScope ts(GCMI_K_GATHER_MAX_BWD, st);
        RUN(win_gather_max_bwd_stats(g, dpool, W, W, reinterpret_cast<const uint8_t*>(ws + w.arg[l]), dy, W,
                                     ws + w.gc[l], W, bnv, bnv + W, reinterpret_cast<double*>(ws + w.acc), st));
      }
      RUN(bn_bwd_params_impl(N, W, d_params + m->off_bn_gamma[l], bnv, bnv + W, d_grads + m->off_bn_gamma[l],
                             d_grads + m->off_bn_beta[l], reinterpret_cast<double*>(ws + w.acc), stream));
    } else {
      if (!dy_ready)
        RUN(gcmi_gather_max_bwd(g, dpool, W, W, reinterpret_cast<const uint8_t*>(ws + w.arg[l]), dy, W, stream));
      if (m->batch_norm) {
        const float* bnv = ws + w.bnv[l];
        RUN(bn_bwd_impl(dy, W, ws + w.gc[l], W, N, W, d_params + m->off_bn_gamma[l], bnv, bnv + W,
                        d_grads + m->off_bn_gamma[l], d_grads + m->off_bn_beta[l], (full && !try_fused) ? dgc : nullptr,
                        W, 1, reinterpret_cast<double*>(ws + w.acc), true, stream));
      } else if (full && !try_fused) {
        RUN(gcmi_relu_bwd(dy, W, ws + w.gc[l], W, N, W, stream));
        dgc = dy;
      }
    }
    dy_ready = false;
    if (!full) break;  // reference semantics: nothing in front of a GraphConv output trains
    bool fused_done = false;
    if (try_fused) {
      // one pass over the rows: dgc formed per tile in LDS; dW_rel, dW_self, dbsum; and for l > 0 dS = dgc W_rel^T
      // and the self part of dX
      TimedScope ts(GCMI_K_FUSED_BWD, st);
      const int rc = fused_conv_bwd(sg.n, sg.begin, sg.end, sg.w_rel, sg.w_self, sg.b_off, dy, W, ws + w.gc[l], W,
                                    m->batch_norm ? coef : nullptr, W, ws + w.S[l], w.ldS[l], xin, ldx, K,
                                    d_params + m->off_conv_w[l], d_grads + m->off_conv_w[l], ws + w.dbsum[l],
                                    l > 0 ? dS : nullptr, K, l > 0 ? dX : nullptr, K,
                                    (psums_env && l > 0 && sym && m->batch_norm) ? reinterpret_cast<double*>(ws + w.acc2)
                                                                                 : nullptr, st);
      if (rc == GCMI_OK) {
        fused_done = true;
        have_psums = psums_env && l > 0 && sym && m->batch_norm;
      }
      else if (rc != GCMI_ERR_UNSUPPORTED) return rc;
      else if (m->batch_norm) {  // not covered after all (misaligned buffers): the separate pass, sums once more
        const float* bnv = ws + w.bnv[l];
        RUN(bn_bwd_impl(dy, W, ws + w.gc[l], W, N, W, d_params + m->off_bn_gamma[l], bnv, bnv + W,
                        d_grads + m->off_bn_gamma[l], d_grads + m->off_bn_beta[l], dgc, W, 1,
                        reinterpret_cast<double*>(ws + w.acc), true, stream));
      } else {
        RUN(gcmi_relu_bwd(dy, W, ws + w.gc[l], W, N, W, stream));
        dgc = dy;
      }
    }
    if (!fused_done) {
      RUN(gcmi_seg_gemm_wgrad(sg.n, sg.begin, sg.end, ws + w.S[l], w.ldS[l], K, dgc, W, W,
                              d_grads + m->off_conv_w[l], sg.w_rel, nullptr, nullptr, 0, stream));
      RUN(gcmi_seg_gemm_wgrad(sg.n, sg.begin, sg.end, xin, ldx, K, dgc, W, W, d_grads + m->off_conv_w[l],
                              sg.w_self, ws + w.dbsum[l], sg.b_off, 0, stream));
    }
    ub.src[l] = ws + w.dbsum[l];  // (unpacked into the reference's bias rows in one launch after the loop)
    ub.dst[l] = d_grads + m->off_conv_b[l];
    ub.width[l] = W;
    if (l == 0) break;  // the atom features need no gradient
    // dS = dgc . W_rel^T ; dX = dgc . W_self^T + (transposed gather of dS)
    if (fused_done) {
      // dX holds the self part: the neighbour part is added onto it (bonds listed from both ends: the scatter
      // of dS is a gather)
      // ... and when the window kernels can hold a third tile, the GraphPool backward of the block below in the
      // same pass: dX = dXs + gather(dS) is consumed in LDS and never exists in HBM
      const bool two_stage = sym && two_stage_env && fused_bwd_enabled() && win_two_stage_usable(g, K) &&
                             aligned16(dS) && aligned16(dX) && aligned16(ws + w.tD);
      if (two_stage) {
        TimedScope ts(GCMI_K_GATHER_MAX_BWD, st);
        RUN(win_gather_sumacc_max_bwd(g, dS, K, K, dX, K, reinterpret_cast<const uint8_t*>(ws + w.arg[l - 1]),
                                      ws + w.tD, K, st));
        dy_ready = true;
      } else if (sym) RUN(gcmi_gather_sum_fwd(g, dS, K, K, dX, K, 1, stream));
      else RUN(gcmi_scatter_add(g, dS, K, K, dX, K, stream));
    } else {
      RUN(gcmi_seg_gemm(sg.n, sg.begin, sg.end, dgc, W, W, d_params + m->off_conv_w[l], sg.w_rel, nullptr, 0,
                        0, nullptr, nullptr, nullptr, nullptr, K, 1, 0, dS, K, stream));
      // bonds listed from both ends: the scatter of dS is a gather (LDS-window kernel), and the
      // self term accumulates onto it in the GEMM epilogue
      if (sym) RUN(gcmi_gather_sum_fwd(g, dS, K, K, dX, K, 0, stream));
      RUN(gcmi_seg_gemm(sg.n, sg.begin, sg.end, dgc, W, W, d_params + m->off_conv_w[l], sg.w_self, nullptr, 0,
                        0, nullptr, nullptr, nullptr, nullptr, K, 1, sym ? 2 : 0, dX, K, stream));
      if (!sym) RUN(gcmi_scatter_add(g, dS, K, K, dX, K, stream));
    }
    dpool = dX;
  }
  RUN(unpack_bias_grads(m, ub, L, st));
  return GCMI_OK;
}

}  // extern "C"
