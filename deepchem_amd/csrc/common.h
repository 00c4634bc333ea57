// Shared host/device helpers for libgcmi.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/gcmi.h"

namespace gcmi {

void set_error(const char* fmt, ...);
// 0 / 1 alternately: whether the next row-streaming launch walks its rows backwards (core.cpp)
int next_sweep_direction();
int next_sweep_direction_windows();  // the same for the window gathers (GCMI_SWEEP=1 leaves them forwards)

#define GCMI_CHECK_ARG(cond, ...)          \
  do {                                     \
    if (!(cond)) {                         \
      ::gcmi::set_error(__VA_ARGS__);      \
      return GCMI_ERR_ARG;                 \
    }                                      \
  } while (0)

#define GCMI_CHECK_LAUNCH(what)                                                        \
  do {                                                                                 \
    hipError_t e__ = hipGetLastError();                                                \
    if (e__ != hipSuccess) {                                                           \
      ::gcmi::set_error("%s: %s", what, hipGetErrorString(e__));                       \
      return GCMI_ERR_LAUNCH;                                                          \
    }                                                                                  \
  } while (0)

// Per-kernel-family timing (bench.py roofline): events recorded on the launch stream.
void timing_begin(int kernel_id, hipStream_t s);
void timing_end(int kernel_id, hipStream_t s);

struct TimedScope {
  int id;
  hipStream_t s;
  TimedScope(int id_, hipStream_t s_) : id(id_), s(s_) { timing_begin(id, s); }
  ~TimedScope() { timing_end(id, s); }
};

// The degree blocks of a collated batch, passed BY VALUE to kernels (lives in
// SGPRs / the kernarg segment: no memory traffic for row -> degree lookups).
struct DegTable {
  int32_t max_deg;
  int32_t deg_start[GCMI_MAX_DEG + 2];
  int32_t edge_start[GCMI_MAX_DEG + 2];
};

inline DegTable make_deg_table(const gcmi_graph* g) {
  DegTable t;
  t.max_deg = g->max_deg;
  for (int d = 0; d < GCMI_MAX_DEG + 2; ++d) {
    t.deg_start[d] = g->deg_start[d < g->max_deg + 1 ? d : g->max_deg + 1];
    t.edge_start[d] = g->edge_start[d < g->max_deg + 1 ? d : g->max_deg + 1];
  }
  return t;
}

int check_graph(const gcmi_graph* g, bool need_cols);

// LDS-window forms of the gather kernels (gather_lds.hip)
bool win_usable(const gcmi_graph* g, int n_feat, bool aux);
bool win_has_width(int n_feat);
int win_gather_sum(const gcmi_graph* g, const float* d_x, int64_t ldx, int n_feat, float* d_s,
                   int64_t lds, hipStream_t st, bool accumulate = false);
int win_gather_max(const gcmi_graph* g, const float* d_x, int64_t ldx, int n_feat, const float* d_scale,
                   const float* d_shift, float* d_out, int64_t ldo, uint8_t* d_arg, hipStream_t st);
int win_gather_max_bwd(const gcmi_graph* g, const float* d_dout, int64_t lddo, int n_feat,
                       const uint8_t* d_arg, float* d_dx, int64_t lddx, hipStream_t st);
// SumOp<true> followed by the GraphPool backward of the block below in one window pass, dX in LDS only
bool win_two_stage_usable(const gcmi_graph* g, int n_feat);
int win_gather_sumacc_max_bwd(const gcmi_graph* g, const float* d_ds, int64_t ldds, int n_feat, float* d_dxs,
                              int64_t lddxs, const uint8_t* d_arg, float* d_dy, int64_t lddy, hipStream_t st);
// the same, also adding the column sums of the BatchNorm backward (sum dx, sum dx*xhat; bn.hip scratch layout)
bool win_stats_usable(const gcmi_graph* g, int n_feat);
int win_gather_max_bwd_stats(const gcmi_graph* g, const float* d_dout, int64_t lddo, int n_feat, const uint8_t* d_arg,
                             float* d_dx, int64_t lddx, const float* d_x, int64_t ldx, const float* d_mean,
                             const float* d_invstd, double* d_sums, hipStream_t st);
// BatchNorm backward from the sums sum dP, sum dP * P over the rows of the block ABOVE (bwd_fused.hip: psums), with
// P = max over neighbours of the BatchNorm output y = gamma * xhat + beta: sum dy = sum dP and
// sum dy * xhat = (sum dP * P - beta * sum dP) / gamma, no pass over dy.  Where that division is ill-conditioned
// (|beta| > 64 |gamma| in some column) the direct column sums are taken instead: the kernels that are only needed
// for them (column sums; in reference mode also the GraphPool backward) are launched every time and return at once
// unless the test below says so.  d_dy may be NULL when the caller never produces it (then the direct sums are too).
int bn_bwd_pool_impl(const float* d_dy, int64_t lddy, const float* d_x, int64_t ldx, int64_t n_rows, int32_t n_feat,
                     const float* d_gamma, const float* d_beta, const float* d_mean, const float* d_invstd,
                     float* d_dgamma, float* d_dbeta, double* d_psums, double* d_acc, void* stream, int32_t x_bf16 = 0);
int win_gather_max_bwd_if_ill(const gcmi_graph* g, const float* d_dout, int64_t lddo, int n_feat, const uint8_t* d_arg,
                              float* d_dx, int64_t lddx, const float* d_gamma, const float* d_beta, hipStream_t st);
// the part of the BatchNorm backward after its column sums (dgamma, dbeta, coefficient vectors at the head of d_acc)
// (d_loss_acc: also *d_loss = inv_count * sum of the loss_rep accumulator replicas, cleared -- loss_finalize_impl folded in)
int bn_bwd_params_impl(int64_t n_rows, int32_t n_feat, const float* d_gamma, const float* d_mean,
                       const float* d_invstd, float* d_dgamma, float* d_dbeta, double* d_acc, void* stream,
                       double* d_loss_acc = nullptr, int loss_rep = 0, float loss_inv_count = 0.f, float* d_loss = nullptr);

// ---- bf16 activation storage (gcmi_model_desc.storage == 1): raw 16-bit patterns, leading dimensions in elements
bool win_usable_h(const gcmi_graph* g, int n_feat);
int win_gather_sum_fh(const gcmi_graph* g, const float* d_x, int64_t ldx, int n_feat, unsigned short* d_s,
                      unsigned short* d_xcopy, int64_t ldo, hipStream_t st);
int win_gather_sum_h(const gcmi_graph* g, const unsigned short* d_x, int64_t ldx, int n_feat, unsigned short* d_s,
                     int64_t lds, hipStream_t st);
int win_gather_max_h(const gcmi_graph* g, const unsigned short* d_x, int64_t ldx, int n_feat, const float* d_scale,
                     const float* d_shift, unsigned short* d_out, int64_t ldo, uint8_t* d_arg, hipStream_t st);
// gradient streams in bf16 (storage == 2)
bool win_usable_gh(const gcmi_graph* g, int n_feat);
int win_gather_max_bwd_h(const gcmi_graph* g, const unsigned short* d_dout, int64_t lddo, int n_feat, const uint8_t* d_arg,
                         unsigned short* d_dx, int64_t lddx, const float* only_if_gamma, const float* only_if_beta,
                         hipStream_t st);
bool win_two_stage_usable_h(const gcmi_graph* g, int n_feat);
int win_gather_sumacc_max_bwd_h(const gcmi_graph* g, const unsigned short* d_ds, int64_t ldds, int n_feat,
                                unsigned short* d_dxs, int64_t lddxs, const uint8_t* d_arg, unsigned short* d_dy,
                                int64_t lddy, hipStream_t st);
bool win_max_sum_usable_h(const gcmi_graph* g, int n_feat);
int win_gather_max_sum_h(const gcmi_graph* g, const unsigned short* d_x, int64_t ldx, int n_feat, const float* d_scale,
                         const float* d_shift, unsigned short* d_out, int64_t ldo, uint8_t* d_arg, unsigned short* d_s,
                         int64_t lds, hipStream_t st);
// fwd_bf16.hip: forward product over bf16 operands, bf16 output, BatchNorm sums of the rounded output
int fwd_h_gemm(int32_t n_seg, const int32_t* seg_begin, const int32_t* seg_end, const unsigned short* d_a1, int64_t lda1,
               int32_t k1, const float* d_w1, const int64_t* w1_off, const unsigned short* d_a2, int64_t lda2, int32_t k2,
               const float* d_w2, const int64_t* w2_off, const float* d_bias, const int64_t* bias_off, int32_t n_out,
               int32_t trans_w, int32_t act, unsigned short* d_out, int64_t ldo, double* d_stats, float* d_wimg_scratch,
               hipStream_t sm);
int fwd_weight_images(int32_t n_seg, const int64_t* w1_off, const int64_t* w2_off, const float* d_w1, const float* d_w2,
                      int32_t k_in, int32_t ko, int32_t n_ops, int32_t n_out, int32_t trans_w, float* d_scratch,
                      hipStream_t sm);
constexpr int64_t kFwdHWimgFloats = 16 * 20 * 3 * 256;  // scratch of fwd_h_gemm: split weight fragments of <= 16 segments

bool gemm_exact_mode();  // gcmi_set_option(GCMI_OPT_GEMM_EXACT)

// row slabs of the weight-gradient kernels (gemm.hip, gemm_split.hip)
constexpr int kMaxSegW = 16;
struct SlabTable {
  int32_t n_seg;
  int32_t slab_rows;
  int32_t seg_begin[kMaxSegW];
  int32_t seg_end[kMaxSegW];
  int32_t slab_start[kMaxSegW + 1];
  int64_t dw_off[kMaxSegW];
  int64_t db_off[kMaxSegW];
};
int launch_wgrad3(const SlabTable& st, int slabs, const float* d_a, int64_t lda, int k, const float* d_g, int64_t ldg,
                  int n, float* d_dw, float* d_dbias, int trans_w, hipStream_t sm);

// gemm_split.hip: the segmented GEMM on the bf16 matrix cores with exactly split fp32 operands;
// GCMI_ERR_UNSUPPORTED = shape not covered (fall back to gemm.hip)
int launch_seg_gemm3(int32_t n_seg, const int32_t* seg_begin, const int32_t* seg_end, const float* d_a1,
                     int64_t lda1, int32_t k1, const float* d_w1, const int64_t* w1_off, const float* d_a2,
                     int64_t lda2, int32_t k2, const float* d_w2, const int64_t* w2_off, const float* d_bias,
                     const int64_t* bias_off, int32_t n_out, int32_t trans_w, int32_t act, float* d_out,
                     int64_t ldo, hipStream_t sm);

int launch_seg_gemm4(int32_t n_seg, const int32_t* seg_begin, const int32_t* seg_end, const float* d_a1,
                     int64_t lda1, int32_t k1, const float* d_w1, const int64_t* w1_off, const float* d_a2,
                     int64_t lda2, int32_t k2, const float* d_w2, const int64_t* w2_off, const float* d_bias,
                     const int64_t* bias_off, int32_t n_out, int32_t trans_w, int32_t act, float* d_out,
                     int64_t ldo, hipStream_t sm, double* d_stats = nullptr);

// bwd_fused.hip: BatchNorm-backward + weight gradients + input gradients of one block in one pass over the rows;
// GCMI_ERR_UNSUPPORTED = switched off / exact mode / shape not covered (the caller runs the separate kernels)
void set_readout_pipelined(int on);
int get_readout_pipelined();
void set_fused_bwd(int on);
int get_fused_bwd();
bool fused_bwd_enabled();
int fused_bwd_launches();  // launches of the one-pass kernel so far (tests)
int fused_conv_bwd(int32_t n_seg, const int32_t* seg_begin, const int32_t* seg_end, const int64_t* w_rel,
                   const int64_t* w_self, const int64_t* b_off, const float* d_dy, int64_t lddy, const float* d_gc,
                   int64_t ldgc, const float* d_coef, int32_t width, const float* d_s, int64_t lds, const float* d_x,
                   int64_t ldx, int32_t k_in, const float* d_w, float* d_dw, float* d_dbsum, float* d_ds_out,
                   int64_t ldds, float* d_dxs_out, int64_t lddxs, double* d_psums, hipStream_t sm, int32_t act_bf16 = 0);
int fused_dense_bwd(int64_t n_rows, const int32_t* d_membership, const float* d_g2, int64_t ldg2,
                    const int32_t* d_arg, const float* d_dense, int64_t ldd, const float* d_coef, int32_t width,
                    const float* d_p, int64_t ldp, int32_t k_in, const float* d_w, float* d_dw, float* d_db,
                    float* d_dp, int64_t lddp, double* d_psums, hipStream_t sm, int32_t act_bf16 = 0);

// fwd_fused.hip: the forward product of a block as persistent workgroups with resident weight images (default widths,
// split-bf16 mode); GCMI_ERR_UNSUPPORTED = shape not covered
int fwd_fused_gemm(int32_t n_seg, const int32_t* seg_begin, const int32_t* seg_end, const float* d_a1, int64_t lda1,
                   int32_t k1, const float* d_w1, const int64_t* w1_off, const float* d_a2, int64_t lda2, int32_t k2,
                   const float* d_w2, const int64_t* w2_off, const float* d_bias, const int64_t* bias_off,
                   int32_t n_out, int32_t trans_w, int32_t act, float* d_out, int64_t ldo, double* d_stats,
                   hipStream_t sm, float* d_wimg_scratch = nullptr);

// head_bwd.hip: loss + d logits + task-head gradients + tanh' of the readout + the dense BatchNorm's backward sums in
// one kernel over the molecules (<= 32 outputs), or two on the matrix cores (33..256 outputs, d_dl_scratch = n_mols x
// outputs floats); GCMI_ERR_UNSUPPORTED = shape not covered (256-column fingerprint)
int head_bwd_fused(int32_t kind, const float* d_logits, const float* d_labels, const float* d_weights, int64_t n_rows,
                   int32_t n_tasks, int32_t n_classes, int64_t n_mols, const float* d_fp, int64_t ldfp,
                   const float* d_w, float* d_dw, float* d_db, float* d_g2, int64_t ldg2, double* d_loss_acc,
                   const int32_t* d_runs, int32_t n_deg, const int32_t* d_arg, const float* d_rawsum,
                   const float* d_mean, const float* d_invstd, double* d_sums, int32_t dense_width, hipStream_t st,
                   float* d_dl_scratch = nullptr, const float* d_img = nullptr);
// ... and the forward head with 33..256 outputs (one segment, 256-column rows, nn.Linear weight, no activation); d_img:
// the fragment images head_prep made of d_w (kHeadImgFloats floats: forward order, then backward order), or nullptr
constexpr int kHeadImgFloats = 2 * 8 * 16 * 3 * 64 * 4;
int head_prep(const float* d_w, int32_t n_out, float* d_img, hipStream_t st);
int head_fwd_wide(const float* d_in, int64_t ldin, int64_t n_rows, int32_t k, const float* d_w, const float* d_bias,
                  int32_t n_out, int32_t act, float* d_out, int64_t ldo, hipStream_t st, const float* d_img = nullptr);
// replicas of the loss accumulator (doubles) that head_bwd_fused adds into; loss_finalize_impl sums and clears them
constexpr int kLossRep = 16;
int loss_finalize_impl(double* d_acc, float inv_count, float* d_loss, void* stream, int n_rep = 1);

// accumulator replicas of the BatchNorm column sums (same-address fp64 atomics serialise); scratch layout in
// doubles: [0, 2F) backward coefficient vectors, then kBnReplicas blocks of [sum(F) | sum of squares(F)]
constexpr int kBnReplicas = 32;
// gcmi_seg_gemm with the column sums of the output (after bias and activation) added into d_stats in that layout
// by the product's own epilogue; *fused = false (and nothing added) when the kernel in charge cannot do it
int seg_gemm_stats(int32_t n_seg, const int32_t* seg_begin, const int32_t* seg_end, const float* d_a1, int64_t lda1,
                   int32_t k1, const float* d_w1, const int64_t* w1_off, const float* d_a2, int64_t lda2, int32_t k2,
                   const float* d_w2, const int64_t* w2_off, const float* d_bias, const int64_t* bias_off,
                   int32_t n_out, int32_t trans_w, int32_t act, float* d_out, int64_t ldo, double* d_stats,
                   bool* fused, void* stream, float* d_wimg_scratch = nullptr);
// gcmi_readout_fwd that also leaves the per-molecule sums of the rows before the folded BatchNorm in d_rawsum
// GraphGather forward over the LDS molecule windows (gather_lds.hip: ReadoutOp); GCMI_ERR_UNSUPPORTED: not applicable
int win_readout(const gcmi_graph* g, const float* d_x, int64_t ldx, int n_feat, const float* d_scale, const float* d_shift,
                int act, float* d_out, int64_t ldo, int32_t* d_arg, float* d_rawsum, int x_bf16, hipStream_t st);
int readout_fwd_impl(const gcmi_graph* g, const float* d_x, int64_t ldx, int32_t n_feat, const float* d_scale,
                     const float* d_shift, int32_t act, float* d_out, int64_t ldo, int32_t* d_arg, float* d_rawsum,
                     void* stream, int32_t x_bf16 = 0);
// the part of bn_stats_impl after the column sums
int bn_finalize_impl(int64_t n_rows, int32_t n_feat, const float* d_gamma, const float* d_beta, float eps,
                     float momentum, float* d_running_mean, float* d_running_var, float* d_mean, float* d_invstd,
                     float* d_scale, float* d_shift, double* d_acc, void* stream, int64_t* d_batches_tracked = nullptr);

// BatchNorm / loss with a caller-guaranteed clean accumulator scratch (bn.hip, loss.hip): the
// whole-model path zeroes its scratch once per pass instead of once per call.
int bn_stats_impl(const float* d_x, int64_t ldx, int64_t n_rows, int32_t n_feat, const float* d_gamma,
                  const float* d_beta, float eps, float momentum, float* d_running_mean,
                  float* d_running_var, float* d_mean, float* d_invstd, float* d_scale, float* d_shift,
                  double* d_acc, bool acc_clean, void* stream, int64_t* d_batches_tracked = nullptr);
int bn_bwd_impl(const float* d_dy, int64_t lddy, const float* d_x, int64_t ldx, int64_t n_rows,
                int32_t n_feat, const float* d_gamma, const float* d_mean, const float* d_invstd,
                float* d_dgamma, float* d_dbeta, float* d_dx, int64_t lddx, int32_t relu_mask,
                double* d_acc, bool acc_clean, void* stream);
int bn_bwd_readout_impl(const int32_t* d_membership, const float* d_g2, int64_t ldg2, const int32_t* d_arg,
                        const float* d_x, int64_t ldx, int64_t n_rows, int32_t n_feat, const float* d_gamma,
                        const float* d_mean, const float* d_invstd, float* d_dgamma, float* d_dbeta, float* d_dx,
                        int64_t lddx, int32_t relu_mask, double* d_acc, bool acc_clean, void* stream,
                        const float* d_rawsum = nullptr, const int32_t* d_mol_runs = nullptr, int32_t n_mols = 0,
                        int32_t n_deg = 0);
int readout_grad_prep(float* d_g, int64_t ldg, const float* d_out, int64_t ldo, int64_t n_mols, int n_feat,
                      hipStream_t st);
int loss_impl(int32_t kind, const float* d_logits, const float* d_labels, const float* d_weights,
              int64_t n_rows, int32_t n_tasks, int32_t n_classes, float* d_loss, float* d_dlogits,
              float* d_probs, double* d_acc, bool acc_clean, void* stream);

// degree of batch row i: the number of block starts (d >= 1) that are <= i.
__device__ __forceinline__ int degree_of_row(const DegTable& t, int i) {
  int d = 0;
#pragma unroll
  for (int k = 1; k <= GCMI_MAX_DEG; ++k) d += (k <= t.max_deg && i >= t.deg_start[k]) ? 1 : 0;
  return d;
}

// wave-uniform: some column has |beta| > 64 |gamma| (or a NaN): (y - beta) / gamma does not recover xhat well there
__device__ __forceinline__ bool bn_pool_ill_conditioned(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int n_feat) {
  bool bad = false;
  for (int c = threadIdx.x & 63; c < n_feat; c += 64) {
    const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    bad = bad || !(fabsf(bt) <= 64.f * fabsf(gm));
  }
  return __ballot(bad) != 0ull;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// vector width usable for a row-major matrix access
inline int vec_width(const void* p, int64_t ld, int n_feat) {
  return (aligned16(p) && (ld % 4 == 0) && (n_feat % 4 == 0)) ? 4 : 1;
}

constexpr int kMaxBlocks = 256 * 8 * 4;  // grid cap for grid-stride kernels: 256 CUs x 8 x 4

inline int grid_for(int64_t work_items, int block) {
  int64_t b = (work_items + block - 1) / block;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return static_cast<int>(b);
}

template <int V>
struct Vec;
template <>
struct Vec<1> {
  using T = float;
};
template <>
struct Vec<4> {
  using T = float4;
};

__device__ __forceinline__ float vzero(float) { return 0.f; }
__device__ __forceinline__ float4 vzero(float4) { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
__device__ __forceinline__ float4 vadd(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

}  // namespace gcmi
