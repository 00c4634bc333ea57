/*
 * gcmi.h -- C ABI of libgcmi.so: the MI355X (gfx950) graph-convolution hot path.
 *
 * The reference (DeepChem, pure Python) has no FFI for this path; its boundary
 * is a Python class contract.  These entry points are what the
 * reference's layers would bind if they had one -- one per reference function
 * on the hot path -- and what deepchem_amd's host-side mirror calls through
 * ctypes.  Reference citations are relative to /root/reference/deepchem/.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / STL types.
 *   - every pointer named d_* is DEVICE memory, borrowed for the call; all
 *     other pointers are host memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *     work is enqueued on it; nothing synchronises, allocates or frees, so
 *     every entry point can be captured into a hipGraph.
 *   - return 0 on success, a negative gcmi_status otherwise; the message of
 *     the last failure on the calling thread: gcmi_last_error().
 *   - feature matrices are row-major float32 with an explicit leading
 *     dimension (ld, in floats).  When ld % 4 == 0 and the base is 16-byte
 *     aligned the kernels move 16 B per lane; otherwise 4 B per lane.
 *   - thread-compatible: no global mutable state except the per-thread error
 *     string.
 */
#ifndef GCMI_H
#define GCMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCMI_VERSION 100 /* 0.1.0 */
#define GCMI_MAX_DEG 10  /* degrees 0..10: ConvMol default, feat/mol_graphs.py:48 */

typedef enum gcmi_status {
  GCMI_OK = 0,
  GCMI_ERR_ARG = -1,    /* bad argument (null pointer, negative size, shape mismatch) */
  GCMI_ERR_LAUNCH = -2, /* HIP launch / runtime error */
  GCMI_ERR_UNSUPPORTED = -3
} gcmi_status;

/*
 * One collated batch of molecules in the layout ConvMol.agglomerate_mols
 * produces (feat/mol_graphs.py:256-349): atoms sorted by (degree, molecule),
 * so the atoms of degree d are the contiguous rows
 * [deg_start[d], deg_start[d+1]) and each of them has exactly d neighbours.
 * The per-degree (n_d, d) int32 tables deg_adj_lists[1..max_deg] are stored
 * back to back in ONE array d_col_idx; the table of degree d starts at
 * edge_start[d].  That is a CSR whose row pointer is implicit:
 *   row_ptr(i) = edge_start[d] + (i - deg_start[d]) * d.
 */
#define GCMI_WIN_META_INTS 24
#define GCMI_WIN_SLOT_BITS 12
#define GCMI_WIN_MAX_SLOTS 4095

typedef struct gcmi_graph {
  int32_t n_atoms;                      /* N                                        */
  int32_t n_edges;                      /* E = sum_d d * n_d  (directed)            */
  int32_t n_mols;                       /* B: rows GraphGather emits (batch_size)   */
  int32_t max_deg;                      /* <= GCMI_MAX_DEG                          */
  int32_t deg_start[GCMI_MAX_DEG + 2];  /* [max_deg+1] = N                          */
  int32_t edge_start[GCMI_MAX_DEG + 2]; /* [max_deg+1] = E                          */
  const int32_t* d_col_idx;             /* E neighbour rows                         */
  const int32_t* d_membership;          /* N: molecule of every atom, ascending
                                           inside every degree block               */
  const int32_t* d_mol_runs;            /* n_mols*(max_deg+1)*2: per molecule and
                                           degree the row range [begin,end) of its
                                           atoms; built by gcmi_build_mol_runs     */
  const uint8_t* d_rev_pos;             /* E (may be NULL): for the edge slot (k, j)
                                           with i = col_idx[row_ptr(k)+j], the slot of
                                           k in i's own neighbour list; exists when
                                           every bond is listed from both ends; built
                                           by gcmi_build_rev_pos or gcmi_collate    */
  /* Molecule windows (optional; produced by gcmi_collate_plans): consecutive molecules grouped
   * so that a window holds <= win_cap atoms (a larger molecule gets a window of its own; those
   * oversized windows come last in d_win_meta).
   * Because every degree block is sorted by molecule, the atoms of a window are <= max_deg+1
   * CONTIGUOUS row ranges, one per degree block.  A workgroup streams those ranges into LDS
   * once ("slots", numbered degree block by degree block) and serves every neighbour read of
   * the window from LDS: all neighbours of an atom belong to its own molecule, hence to its
   * window.
   *   d_win_meta[w][GCMI_WIN_META_INTS]:
   *     [0..10]  row_of_slot_base[d] = first row of the window in degree block d - slot_start[d]
   *              (global row of slot s of degree d = base[d] + s)
   *     [11..21] slot_start[1..11]  (slot_start[0] = 0, slot_start[11] = atoms in the window)
   *     [22]     first entry of the window in d_win_edges (a multiple of 8)
   *     [23]     edge entries of the window (unpadded)
   *   d_win_edges: the neighbour lists window by window, inside a window in (degree, row, j)
   *     order, every window padded to a multiple of 8 entries; entry = LDS slot of the
   *     neighbour | rev_pos << 12 (rev_pos = 15: no partner).                                */
  int32_t n_win;                        /* all windows: ordinary ones first, then     */
  int32_t n_win_big;                    /* the oversized ones (one molecule > win_cap) */
  int32_t win_alloc;                    /* atoms of the largest ordinary window       */
  int32_t win_ecap;                     /* its padded edge entries (largest)          */
  int32_t win_alloc_big;                /* the same for the oversized windows         */
  int32_t win_ecap_big;
  int32_t win_reserved[2];              /* [0]: molecules with at least one atom among the collated ones (filled by
                                           gcmi_collate / gcmi_collate_plans with the windows; 0 = unknown): when it
                                           equals n_mols every molecule lies in some window; [1]: 0 */
  const int32_t* d_win_meta;            /* n_win * GCMI_WIN_META_INTS                */
  const uint16_t* d_win_edges;          /* <= E + 8 * n_win                          */
} gcmi_graph;

int gcmi_version(void);
/* Process-wide options.  GCMI_OPT_GEMM_EXACT: 1 = all matrix products on the exact-fp32 MFMA (the
 * reference's summation order: training trajectories track the reference to ~1e-5); 0 (default) =
 * the split-bf16 kernels (fp32-accurate per product, ~1.25x faster, own rounding).  Also settable
 * by the environment variable GCMI_GEMM_EXACT=1 before the library is loaded.                   */
enum {
  GCMI_OPT_GEMM_EXACT = 1,
  /* 1 (default): the training forward of the whole-model entry points takes the BatchNorm column sums from the
   * epilogue of the producing product; 0: separate column-sum launches (same statistics up to summation order) */
  GCMI_OPT_FUSED_BN_STATS = 2,
  /* 1 (default): the whole-model backward forms the gradient of a block's pre-activation in LDS and computes the
   * weight gradients and the input gradients from it in one pass over the rows (bwd_fused.hip; split-bf16 mode,
   * default widths); 0: separate BatchNorm-backward, weight-gradient and input-gradient launches (same arithmetic
   * up to summation order).  Environment: GCMI_FUSED_BWD=0 before the library is loaded.                        */
  GCMI_OPT_FUSED_BWD = 3,
  /* read-only (gcmi_get_option): launches of the one-pass backward kernel in this process so far */
  GCMI_OPT_FUSED_BWD_LAUNCHES = 4,
  /* 1 (default): gcmi_readout_fwd and the whole-model forward walk a molecule's rows as one sequence of four-row
   * rounds with two rounds in flight and the run bounds held in registers (readout.hip); 0: run by run, one round
   * at a time.  Same rows in the same order: the results are bit-identical.  Environment: GCMI_READOUT_PRE=0
   * before the library is loaded.                                                                              */
  GCMI_OPT_READOUT_PIPELINED = 5
};
int gcmi_set_option(int32_t option, int32_t value);
int gcmi_get_option(int32_t option, int32_t* value);
const char* gcmi_last_error(void);

/* ---------------------------------------------------------------- host side
 * gcmi_collate: ConvMol.agglomerate_mols (feat/mol_graphs.py:256-349) over a
 * packed molecule set, plus the flattening the kernels want.  Host only.
 *   in : atom_features [A x n_feat] (molecule-major), atom_ptr [n_set+1],
 *        adj_ptr [A+1], adj_idx [nnz] (molecule-local ids), sel [n_sel]
 *        molecules to collate, in order (repeats allowed = pad_batch tiling,
 *        data/datasets.py:204-216).
 *   out: out_features [N x out_ld] (rows in batch order, columns >= n_feat
 *        zero), out_membership [N], out_col_idx [E], out_mol_runs
 *        [n_sel*(max_deg+1)*2], *graph (counts and offsets filled, device
 *        pointers left NULL).  Capacities are checked.
 */
int gcmi_collate_sizes(const int64_t* atom_ptr, const int64_t* adj_ptr, const int64_t* sel,
                       int64_t n_sel, int64_t* out_n_atoms, int64_t* out_n_edges);
int gcmi_collate(const float* atom_features, int64_t n_feat, const int64_t* atom_ptr,
                 const int64_t* adj_ptr, const int32_t* adj_idx, const int64_t* sel,
                 int64_t n_sel, int32_t max_deg, float* out_features, int64_t out_ld,
                 int64_t cap_atoms, int32_t* out_membership, int32_t* out_col_idx,
                 int64_t cap_edges, int32_t* out_mol_runs, gcmi_graph* graph);
/* The same plus the plans of the LDS-window kernels and of the gather-form backwards, all
 * optional (NULL = skip):
 *   out_rev_pos   [E]  uint8; *out_symmetric = 0 when some bond is not listed from both ends
 *                      (the table is then unusable);
 *   win_cap > 0:   out_win_meta [<= n_sel*GCMI_WIN_META_INTS], out_win_edges
 *                  [<= E + 8*n_sel] uint16; graph->n_win, n_win_big, win_alloc*, win_ecap* are filled
 *                  (n_win = 0 when a molecule exceeds GCMI_WIN_MAX_SLOTS atoms).               */
int gcmi_collate_plans(const float* atom_features, int64_t n_feat, const int64_t* atom_ptr,
                       const int64_t* adj_ptr, const int32_t* adj_idx, const int64_t* sel,
                       int64_t n_sel, int32_t max_deg, float* out_features, int64_t out_ld,
                       int64_t cap_atoms, int32_t* out_membership, int32_t* out_col_idx,
                       int64_t cap_edges, int32_t* out_mol_runs, uint8_t* out_rev_pos,
                       int32_t* out_symmetric, int32_t win_cap, int32_t* out_win_meta,
                       uint16_t* out_win_edges, gcmi_graph* graph);

/* ---------------------------------------------------------------- collation on the device
 * The same batches (ConvMol.agglomerate_mols, feat/mol_graphs.py:256-349; the arena gcmi_collate_plans
 * writes, byte for byte) built by the GPU from a molecule set that stays in HBM, so that an epoch of
 * shuffled batches (DiskDataset.iterbatches, data/datasets.py:1518-1623) moves a few MB of row bases per
 * batch over PCIe instead of the whole batch.
 *
 * gcmi_molset_tables (host, once per set): out_mol_hist [n_mols x 11] atoms per degree, out_rank [A] an
 *   atom's rank among the atoms of its degree in its molecule, out_rev [nnz] the reverse slot of every
 *   neighbour entry (15 = none; *out_symmetric = 0 then).  Fails on a degree above max_deg or a
 *   neighbour id outside its molecule, like gcmi_collate_plans.
 * gcmi_collate_plan (host, per batch): the serial part -- row bases of every molecule in every degree
 *   block, windows -- into `staging` (int32 words, >= gcmi_collate_plan_words(n_sel); only the first
 *   out_offsets[6] words are used and need copying).  out_offsets[8]: word offsets of
 *   base, mol_win, atom_off, atom0 (int64), win_meta (what graph->d_win_meta must point at on the
 *   device), window descriptors; [6] words used; [7] uint16 entries of the window edge array.
 *   *graph is filled as by gcmi_collate_plans.
 * gcmi_collate_rows (device, per batch): one thread per atom writes features (n_feat floats per atom of
 *   the resident set; 2 for 8-byte atom codes), membership, col_idx, rev_pos (NULL = skip), window
 *   entries; a second launch writes mol_runs (NULL = skip).  d_staging is the device copy of the plan.
 *   d_src_atom (NULL, or n_atoms int64 of scratch): with it the per-atom threads only note their source
 *   atom and a further launch copies the feature rows with consecutive lanes on consecutive columns
 *   (wide float rows); without it every thread copies its own row (8-byte atom codes).
 * gcmi_collate_rows_host: the same routine run by host loops over host buffers, for tests without a GPU. */
#define GCMI_COLLATE_WIN_DESC_INTS 36
int gcmi_molset_tables(const int64_t* atom_ptr, const int64_t* adj_ptr, const int32_t* adj_idx, int64_t n_mols,
                       int32_t max_deg, int32_t* out_mol_hist, int32_t* out_rank, uint8_t* out_rev,
                       int32_t* out_symmetric, int32_t n_threads);
int64_t gcmi_collate_plan_words(int64_t n_sel);
int gcmi_collate_plan(const int32_t* mol_hist, const int64_t* atom_ptr, const int64_t* sel, int64_t n_sel,
                      int32_t max_deg, int32_t win_cap, int32_t* staging, int64_t staging_words,
                      int64_t* out_offsets, gcmi_graph* graph);
int gcmi_collate_rows(const void* d_features, int64_t n_feat, const int64_t* d_adj_ptr, const int32_t* d_adj_idx,
                      const int32_t* d_rank, const uint8_t* d_rev, const int32_t* d_staging,
                      const int64_t* offsets, const gcmi_graph* plan, float* d_out_features, int64_t out_ld,
                      int32_t* d_membership, int32_t* d_col_idx, int32_t* d_mol_runs, uint8_t* d_rev_pos,
                      uint16_t* d_win_edges, int64_t* d_src_atom, void* stream);
int gcmi_collate_rows_host(const void* features, int64_t n_feat, const int64_t* adj_ptr, const int32_t* adj_idx,
                           const int32_t* rank, const uint8_t* rev, const int32_t* staging, const int64_t* offsets,
                           const gcmi_graph* plan, float* out_features, int64_t out_ld, int32_t* membership,
                           int32_t* col_idx, int32_t* mol_runs, uint8_t* rev_pos, uint16_t* win_edges);

/* ---------------------------------------------------------------- graph plan
 * d_mol_runs from d_membership (device).  d_flag (1 int, device) is set to 1
 * when membership is NOT ascending inside a degree block or out of range.   */
int gcmi_build_mol_runs(const gcmi_graph* g, int32_t* d_mol_runs, int32_t* d_flag, void* stream);
/* d_rev_pos from d_col_idx (device).  d_flag is set to 1 when some edge has no
 * partner (the adjacency is not symmetric): the table is then unusable and the
 * scatter (atomic) backward kernels must be used.                              */
int gcmi_build_rev_pos(const gcmi_graph* g, uint8_t* d_rev_pos, int32_t* d_flag, void* stream);

/* ---------------------------------------------------------------- K1 gather-sum
 * GraphConv.sum_neigh (models/torch_models/layers.py:6236-6246):
 *   s[i,:] = sum_{j<deg(i)} x[col_idx[row_ptr(i)+j], :]   (degree-0 rows: 0)
 * Backward of the gather (dx[k,:] += ds[i,:] for every edge i->k) as
 * gcmi_scatter_add (float atomics; any adjacency).  For a symmetric adjacency
 * (every bond listed from both ends: ConvMol) the same result is
 * gcmi_gather_sum_fwd applied to ds -- deterministic, no atomics.             */
int gcmi_gather_sum_fwd(const gcmi_graph* g, const float* d_x, int64_t ldx, int32_t n_feat,
                        float* d_s, int64_t lds, int32_t accumulate /* s += instead of s = */,
                        void* stream);
int gcmi_scatter_add(const gcmi_graph* g, const float* d_ds, int64_t ldds, int32_t n_feat,
                     float* d_dx, int64_t lddx, void* stream);

/* ---------------------------------------------------------------- K3 gather-max
 * GraphPool.forward (layers.py:6319-6367): out[i,:] = max over {x[i], x[nbrs]}
 * with candidates ordered self, nbr_0, nbr_1, ... and the FIRST maximum winning
 * (torch.max(dim) tie rule).  d_arg[i,f] (uint8, ld = n_feat) records the
 * winner: 0 = self, j+1 = neighbour j.  If d_scale/d_shift are non-NULL the
 * candidates are x*scale[f]+shift[f] (the preceding BatchNorm1d folded in).
 * Backward: dx[winner row, f] += dout[i,f].  With g->d_rev_pos the transposed
 * (gather) form runs -- every dx row is written once, no atomics, deterministic;
 * without it float atomics into a dx the caller pre-zeroed.                    */
int gcmi_gather_max_fwd(const gcmi_graph* g, const float* d_x, int64_t ldx, int32_t n_feat,
                        const float* d_scale, const float* d_shift, float* d_out, int64_t ldo,
                        uint8_t* d_arg, void* stream);
int gcmi_gather_max_bwd(const gcmi_graph* g, const float* d_dout, int64_t lddo, int32_t n_feat,
                        const uint8_t* d_arg, float* d_dx, int64_t lddx, void* stream);

/* ---------------------------------------------------------------- K4 readout
 * GraphGather.forward (layers.py:6450-6479) = unsorted_segment_sum
 * (utils/pytorch_utils.py:20-74) ++ unsorted_segment_max (:473-528) (+tanh):
 *   out[b, 0:F]  = act(sum_{i in mol b} x[i,:]),  out[b, F:2F] = act(max ...)
 * n_mols rows always; an empty molecule gives (0, -inf) -> tanh -> (0, -1).
 * Optional folded BatchNorm (scale/shift) as above.  act: 0 none, 1 tanh.
 * d_arg [n_mols x F] int32: row of the first maximum (-1 for empty).
 * Backward: dx[i,:] = dsum[b,:] + (i == arg[b,:]) * dmax[b,:], with the tanh
 * derivative taken from the saved output.                                     */
int gcmi_readout_fwd(const gcmi_graph* g, const float* d_x, int64_t ldx, int32_t n_feat,
                     const float* d_scale, const float* d_shift, int32_t act, float* d_out,
                     int64_t ldo, int32_t* d_arg, void* stream);
int gcmi_readout_bwd(const gcmi_graph* g, const float* d_dout, int64_t lddo, const float* d_out,
                     int64_t ldo, int32_t n_feat, int32_t act, const int32_t* d_arg,
                     float* d_dx, int64_t lddx, void* stream);

/* ---------------------------------------------------------------- K2 batch norm
 * nn.BatchNorm1d(F, eps=1e-3, momentum=0.99) over the atom rows
 * (models/torch_models/graphconvmodel.py:150-165, applied :215-216, :224-225).
 * gcmi_bn_stats: training statistics of x[n_rows x F] -> d_mean, d_invstd,
 *   folded d_scale = gamma*invstd, d_shift = beta - mean*scale; running stats
 *   updated with torch semantics (running = (1-m)*running + m*batch; unbiased
 *   variance into running_var).  d_acc: GCMI_BN_ACC_DOUBLES(F) doubles of scratch
 *   (column sums are accumulated in 32 replicas to keep fp64 atomics uncontended).
 * gcmi_bn_fold_eval: scale/shift from the running statistics (eval mode).
 * gcmi_bn_apply: y = x*scale + shift (only when the consumer cannot fold it).
 * gcmi_bn_bwd: given dy, x and the saved mean/invstd: dgamma, dbeta and
 *   (if d_dx) dx = gamma*invstd*(dy - dbeta/n - xhat*dgamma/n).  relu_mask != 0:
 *   x is the output of a ReLU and dx is wanted w.r.t. its input: dx *= (x > 0).  */
#define GCMI_BN_ACC_DOUBLES(F) (66 * (F))
int gcmi_bn_stats(const float* d_x, int64_t ldx, int64_t n_rows, int32_t n_feat,
                  const float* d_gamma, const float* d_beta, float eps, float momentum,
                  float* d_running_mean, float* d_running_var, float* d_mean, float* d_invstd,
                  float* d_scale, float* d_shift, double* d_acc, void* stream);
int gcmi_bn_fold_eval(const float* d_gamma, const float* d_beta, const float* d_running_mean,
                      const float* d_running_var, float eps, int32_t n_feat, float* d_scale,
                      float* d_shift, void* stream);
int gcmi_bn_apply(const float* d_x, int64_t ldx, int64_t n_rows, int32_t n_feat,
                  const float* d_scale, const float* d_shift, float* d_y, int64_t ldy,
                  void* stream);
int gcmi_bn_bwd(const float* d_dy, int64_t lddy, const float* d_x, int64_t ldx, int64_t n_rows,
                int32_t n_feat, const float* d_gamma, const float* d_mean,
                const float* d_invstd, float* d_dgamma, float* d_dbeta, float* d_dx,
                int64_t lddx, int32_t relu_mask, double* d_acc, void* stream);

/* ---------------------------------------------------------------- GEMMs (MFMA, exact fp32)
 * Row-segmented affine map on v_mfma_f32_32x32x2_f32:
 *   for every segment s (rows [seg_begin[s], seg_end[s])), s < n_seg <= 16:
 *     out[rows,:] = act( a1[rows,:k1] . W1_s + a2[rows,:k2] . W2_s + bias_s )
 * W1_s is the k1 x n_out row-major block at d_w1 + w1_off[s] (floats); a
 * negative offset drops that term for the segment.  a2/w2 may be NULL.
 * bias_s = d_bias + bias_off[s] (n_out floats; negative offset: none).
 * act: 0 none, 1 relu, 2 accumulate (out += the result; no activation).  trans_w != 0: the blocks are stored n_out x k
 * (nn.Linear layout, or the transposed product of a backward pass).
 * seg_begin, seg_end, *_off are HOST arrays of n_seg entries.
 *
 * Uses: GraphConv (layers.py:6202-6231: per degree S_d.W_rel + X_d.W_self + both
 * biases, relu; 11 segments, degree 0 without the S term), the atom-level
 * dense layer (graphconvmodel.py:222-223), the task heads
 * (graphconvmodel.py:230-247), and every data-gradient product of the
 * backward pass (dA = dOut . W^T is the same map with trans_w flipped).       */
int gcmi_seg_gemm(int32_t n_seg, const int32_t* seg_begin, const int32_t* seg_end,
                  const float* d_a1, int64_t lda1, int32_t k1, const float* d_w1,
                  const int64_t* w1_off, const float* d_a2, int64_t lda2, int32_t k2,
                  const float* d_w2, const int64_t* w2_off, const float* d_bias,
                  const int64_t* bias_off, int32_t n_out, int32_t trans_w, int32_t act,
                  float* d_out, int64_t ldo, void* stream);
/* dW_s += a[rows_s,:k]^T . g[rows_s,:n]  (block at d_dw + dw_off[s], k x n, or
 * n x k when trans_w) and dbias_s += column sums of g[rows_s] (d_dbias may be
 * NULL).  Accumulates with float atomics into caller-zeroed buffers.          */
int gcmi_seg_gemm_wgrad(int32_t n_seg, const int32_t* seg_begin, const int32_t* seg_end,
                        const float* d_a, int64_t lda, int32_t k, const float* d_g, int64_t ldg,
                        int32_t n, float* d_dw, const int64_t* dw_off, float* d_dbias,
                        const int64_t* dbias_off, int32_t trans_w, void* stream);
/* The task head's forward product as the whole-model path runs it (graphconvmodel.py:177-179, :230-236:
 * logits = fingerprint . W^T + b with nn.Linear's (n_out x 256) weight).  With 33..256 outputs -- PCBA: 128 tasks x 2 --
 * the head matrix is first split once into fragment images in d_img_scratch (gcmi_task_head_scratch_floats() floats; the
 * model keeps them in its workspace for the backward) and the product runs from those; other shapes, or
 * d_img_scratch == NULL, are gcmi_seg_gemm with trans_w = 1.                                                    */
int64_t gcmi_task_head_scratch_floats(void);
int gcmi_task_head_forward(const float* d_fingerprint, int64_t ld, int64_t n_rows, int32_t k, const float* d_w,
                           const float* d_bias, int32_t n_out, float* d_img_scratch, float* d_out, int64_t ldo,
                           void* stream);
/* elementwise g *= (y > 0): ReLU derivative, in place on g. */
int gcmi_relu_bwd(float* d_g, int64_t ldg, const float* d_y, int64_t ldy, int64_t n_rows,
                  int32_t n_feat, void* stream);

/* ---------------------------------------------------------------- Weave (pair-feature convolution)
 * The index-driven parts of WeaveLayer.forward (models/torch_models/layers.py:4327-4429) and
 * WeaveGather.forward (:4547-4648); the dense products go through gcmi_seg_gemm.  All BatchNorm
 * layers of this path run in eval mode in the reference (layers.py:4361, :4369, :4390, ...):
 * gcmi_fold_affine folds them into the neighbouring weights,
 *     W' = W * diag(scale),  b' = b * scale + shift      (W: k x n row-major, n x k if trans_w)
 * with scale/shift from gcmi_bn_fold_eval.
 *
 * gcmi_weave_pair_to_atom (layers.py:4366-4387): PA = relu(Pf . W + b) summed over the pairs of
 *   every source atom; d_pair_src [n_pairs] int32 = pair_split (ascending: pairs are listed source
 *   by source).  out [n_atoms x n_hidden] (zeroed here, float atomics, one per column and atom).
 * gcmi_weave_pair_features (layers.py:4397-4424): per ordered pair p = (i, j)
 *     z[p, 0:Hap]       = relu(U[i] + V[j] + b_ap) + relu(U[j] + V[i] + b_ap)
 *     z[p, Hap:Hap+Hpp] = relu(Pf[p] . W_pp + b_pp)
 *   where U = A . W_AP[:Fa], V = A . W_AP[Fa:] (n_atoms x Hap, computed per atom by the caller):
 *   the reference's gathered P x 2Fa matmuls, re-associated.  d_atom_to_pair [n_pairs x 2] int32.
 * gcmi_weave_gather (layers.py:4566-4648): per molecule (atoms [mol_ptr[m], mol_ptr[m+1])) the sum
 *   of the atom rows, after the 11-bin Gaussian-histogram expansion when gaussian_expand != 0
 *   (out [n_mols x 11*n_feat], column f*11 + bin) or of the raw rows (out [n_mols x n_feat]).
 * gcmi_tanh_: in-place tanh (final_conv_activation_fn of the Weave model).                      */
int gcmi_fold_affine(const float* d_w, const float* d_b, const float* d_scale, const float* d_shift, int32_t k,
                     int32_t n, int32_t trans_w, float* d_w_out, float* d_b_out, void* stream);
int gcmi_weave_pair_to_atom(const float* d_pair_feat, int64_t ldp, int32_t n_pair_feat, const int32_t* d_pair_src,
                            int64_t n_pairs, int32_t n_atoms, const float* d_w, const float* d_b, int32_t n_hidden,
                            float* d_out, int64_t ldo, void* stream);
int gcmi_weave_pair_features(const float* d_u, const float* d_v, int64_t lduv, int32_t n_hidden_ap,
                             const float* d_b_ap, const float* d_pair_feat, int64_t ldp, int32_t n_pair_feat,
                             const float* d_w_pp, const float* d_b_pp, int32_t n_hidden_pp,
                             const int32_t* d_atom_to_pair, int64_t n_pairs, float* d_z, int64_t ldz,
                             void* stream);
int gcmi_weave_gather(const float* d_x, int64_t ldx, int32_t n_feat, const int32_t* d_mol_ptr, int32_t n_mols,
                      int32_t gaussian_expand, float* d_out, int64_t ldo, void* stream);
int gcmi_tanh_(float* d_x, int64_t ldx, int64_t n_rows, int32_t n_feat, void* stream);

/* ---------------------------------------------------------------- message passing (MPNN sub-layers)
 * gcmi_edge_network_sum: EdgeNetwork.forward (models/torch_models/layers.py:4060-4088) after the
 *   re-association  (reshape(Pf[p].W + b) . h_j)[r] = sum_k Pf[p,k] G[j][k*d+r] + G[j][K*d+r]  with
 *   G = h . [W_0^T | ... | W_{K-1}^T | B^T]  (N x (K+1)d, one gcmi_seg_gemm over the atoms, W read in
 *   place as a (K*d) x d matrix in transposed layout):  out[i] = sum over the pairs p whose first
 *   atom is i (pairs sorted by it; d_dst_ptr [n_dst+1] is that CSR) of the message from d_src[p].
 * gcmi_gru_gates / gcmi_gru_out: the elementwise parts of GatedRecurrentUnit.forward (:2903-2913):
 *   z <- sigmoid(z), r <- sigmoid(r), hr = h*r;   out = (1-z) tanh(hpre) + z x   (contiguous, n floats).
 * gcmi_set2set_attend: one attention step of SetGather.forward (:3047-3064): per molecule
 *   e_a = <x_a, h_m>, a = softmax over its atoms, q_star[m] = [h_m | sum_a a_a x_a].
 * gcmi_lstm_cell: SetGather._LSTMStep (:3066-3092) after z = q_star.U + b: gate order i, f, o, g.  */
int gcmi_edge_network_sum(const float* d_g, int64_t ldg, int32_t n_hidden, int32_t n_pair_feat,
                          const float* d_pair_feat, int64_t ldp, const int32_t* d_dst_ptr, const int32_t* d_src,
                          int32_t n_dst, float* d_out, int64_t ldo, void* stream);
/* gcmi_edge_network_moments: the same message with the weights applied last,
 *   T[i, k*d + c] = sum_p pf[p,k] h[src_p, c] (k < K),  T[i, K*d + c] = sum_p h[src_p, c]   (d <= 128, K <= 16),
 *   message = T . [W_0 | ... | W_{K-1} | B]^T by one gcmi_seg_gemm: a 4d-byte gather per pair instead of 4(K+1)d. */
int gcmi_edge_network_moments(const float* d_h, int64_t ldh, int32_t n_hidden, int32_t n_pair_feat,
                              const float* d_pair_feat, int64_t ldp, const int32_t* d_dst_ptr, const int32_t* d_src,
                              int32_t n_dst, float* d_t, int64_t ldt, void* stream);
/* ... with the molecules of the batch given (d_mol_ptr [n_mols + 1]: atoms of molecule m are the rows
 *   [d_mol_ptr[m], d_mol_ptr[m+1]) -- the CSR of default_generator's atom_split, graph_models.py:1197-1247): a
 *   molecule's state rows are staged once in LDS and serve all its pairs.  Same T; pairs that leave their molecule are
 *   still right (read from memory).  max_mol_atoms: atoms of the largest molecule if the caller knows it (sizes the
 *   LDS: more workgroups per CU), else 0.  d_mol_ptr == NULL: gcmi_edge_network_moments.                          */
int gcmi_edge_network_moments_mol(const float* d_h, int64_t ldh, int32_t n_hidden, int32_t n_pair_feat,
                                  const float* d_pair_feat, int64_t ldp, const int32_t* d_dst_ptr, const int32_t* d_src,
                                  int32_t n_dst, const int32_t* d_mol_ptr, int32_t n_mols, int32_t max_mol_atoms,
                                  float* d_t, int64_t ldt, void* stream);
/* Backward of the same pieces (the training step of MPNNModel, models/graph_models.py:1045-1247; forward
 * formulas models/layers.py:3755-3887).  gru_gates_bwd: dzp = dz z(1-z), drp = dhr h r(1-r), dh = dhr r;
 * gru_out_bwd: dz = dout (x - tanh hpre), dhpre = dout (1-z)(1-tanh^2), dx = dout z; lstm_cell_bwd: dz (rows x 4H,
 * gate order i, f, o, g) and dc of the incoming cell state from dh', dc' (NULL = 0); set2set_attend_bwd: dx (written,
 * or added to when accumulate != 0) and dh from dq = [dh | dr], softmax recomputed.                          */
int gcmi_gru_gates_bwd(const float* d_z, const float* d_r, const float* d_h, const float* d_dz, const float* d_dhr,
                       float* d_dzp, float* d_drp, float* d_dh, int64_t n, void* stream);
int gcmi_gru_out_bwd(const float* d_z, const float* d_hpre, const float* d_x, const float* d_dout, float* d_dz,
                     float* d_dhpre, float* d_dx, int64_t n, void* stream);
int gcmi_lstm_cell_bwd(const float* d_z, int64_t ldz, int32_t n_hidden, int64_t n_rows, const float* d_c_prev,
                       const float* d_dh, const float* d_dc_next, float* d_dz, float* d_dc_prev, void* stream);
int gcmi_set2set_attend_bwd(const float* d_x, int64_t ldx, int32_t n_feat, const int32_t* d_mol_ptr, int32_t n_mols,
                            const float* d_h, int64_t ldh, const float* d_dq, int64_t lddq, float* d_dx, int64_t lddx,
                            int32_t accumulate, float* d_dh, int64_t lddh, void* stream);
int gcmi_gru_gates(float* d_z, float* d_r, const float* d_h, float* d_hr, int64_t n, void* stream);
int gcmi_gru_out(const float* d_z, const float* d_hpre, const float* d_x, float* d_out, int64_t n, void* stream);
int gcmi_set2set_attend(const float* d_x, int64_t ldx, int32_t n_feat, const int32_t* d_mol_ptr, int32_t n_mols,
                        const float* d_h, int64_t ldh, float* d_qstar, int64_t ldq, void* stream);
int gcmi_lstm_cell(const float* d_z, int64_t ldz, int32_t n_hidden, int64_t n_rows, float* d_c, int64_t ldc,
                   float* d_h, int64_t ldh, void* stream);

/* ---------------------------------------------------------------- atom codes
 * The 75-column rows of atom_features (feat/graph_features.py:282-391) are five one-hot blocks, two small integers
 * and a flag.  Code row (8 bytes): symbol column, degree, implicit-valence column, formal charge (int8), radical
 * electrons, hybridisation column, aromatic, total-H column.  Collation and the host-to-device copy move codes
 * (gcmi_collate_plans with n_feat = 2 "floats"); this writes the float rows on the device:
 * out[r, 0:75] as the reference lays them out, out[r, 75:ldo] = 0.                                            */
int gcmi_expand_atom_codes(const uint8_t* d_codes, int64_t ldc_bytes, int64_t n_atoms, float* d_out, int64_t ldo,
                           void* stream);

/* ---------------------------------------------------------------- SMILES featurization (host)
 * What the reference computes in Python per rdkit Mol, for SMILES input:
 *   atom_features feat/graph_features.py:282-391 (75 columns), bond_features :394-459 (6), pair_features +
 *   find_distance :532-695 (14, all pairs), ConvMolFeaturizer._featurize :845-914, WeaveFeaturizer._featurize
 *   :1037-1078; a molecule that cannot be read is skipped like MolecularFeaturizer.featurize does
 *   (feat/base_classes.py:254-330).  Atoms keep their order of appearance in the SMILES.
 * Two passes so that the caller owns all memory:
 *   gcmi_smiles_sizes      n_atoms[i] / n_bonds[i] of every molecule; n_atoms[i] = -1 when it cannot be read.
 *   gcmi_smiles_featurize  atom_off/bond_off (n+1 prefix sums of those sizes, 0 for rejected molecules) and,
 *                          when pair_features is given, pair_off (prefix sums of n_atoms^2).  Every output may
 *                          be NULL:  atom_features [A][75], adj_degree [A], adj_idx [2B] (neighbours LOCAL to
 *                          the molecule, per atom in bond order: together with adj_degree the CSR the collation
 *                          entry points read), bond_atoms [B][2], bond_features [B][6], pair_features
 *                          [sum n^2][14] (row a1*n+a2), atom_props [A][8] = atomic number, degree, implicit H,
 *                          explicit H, charge, radical electrons, hybridisation (0 unspecified, 1 S, 2 SP,
 *                          3 SP2, 4 SP3, 5 SP3D, 6 SP3D2), aromatic.
 *   gcmi_smiles_check      NULL when the SMILES can be read, else a static reason string.
 * n_threads worker threads; results do not depend on it.                                                   */
#define GCMI_ATOM_FEATURES 75
#define GCMI_BOND_FEATURES 6
#define GCMI_PAIR_FEATURES 14
int gcmi_smiles_sizes(const char* const* smiles, int64_t n, int32_t* n_atoms, int32_t* n_bonds, int n_threads);
int gcmi_smiles_featurize(const char* const* smiles, int64_t n, const int64_t* atom_off, const int64_t* bond_off,
                          const int64_t* pair_off, float* atom_features, int32_t* adj_degree, int32_t* adj_idx,
                          int32_t* bond_atoms, float* bond_features, float* pair_features, int32_t* atom_props,
                          int n_threads);
const char* gcmi_smiles_check(const char* smiles);

/* ---------------------------------------------------------------- loss
 * SoftmaxCrossEntropy (models/losses.py:251-259) / L2Loss (:85-94) through
 * _StandardLoss (models/torch_models/torch_model.py:1275-1294):
 *   loss = mean over (n_rows, n_tasks) of  w[b,t] * l[b,t]
 *   kind 0: l = -sum_c y[b,t,c] * log_softmax(logits[b,t,:])[c]
 *   kind 1: l = (out[b,t] - y[b,t])^2
 * Writes *d_loss (float, device) and d_dlogits (same shape as logits).
 * d_probs (may be NULL): softmax(logits) (the model's 'prediction' output,
 * graphconvmodel.py:235).  d_acc: 1 double of scratch.                        */
int gcmi_loss_fwd_bwd(int32_t kind, const float* d_logits, const float* d_labels,
                      const float* d_weights, int64_t n_rows, int32_t n_tasks,
                      int32_t n_classes, float* d_loss, float* d_dlogits, float* d_probs,
                      double* d_acc, void* stream);
int gcmi_softmax(const float* d_logits, int64_t n_rows_tasks, int32_t n_classes, float* d_probs,
                 void* stream);

/* ---------------------------------------------------------------- optimizer
 * torch.optim.Adam(lr, betas, eps, weight_decay=0) (models/optimizers.py:231-241)
 * on one flat range.  step = 1-based step count after this update.            */
int gcmi_adam_step(float* d_param, const float* d_grad, float* d_m, float* d_v, int64_t n,
                   float lr, float beta1, float beta2, float eps, int64_t step, void* stream);

/* ---------------------------------------------------------------- whole-model sequencing
 * _GraphConvTorchModel.forward (models/torch_models/graphconvmodel.py:188-249) and the
 * loss + backward of one fit_generator step (models/torch_models/torch_model.py:436-442)
 * as ONE call each: the library enqueues every kernel back to back on `stream`, so a
 * training step costs three C calls (forward, loss_backward, gcmi_adam_step on the flat
 * gradient range) instead of ~150 Python-driven launches.  Nothing here allocates: the
 * caller owns the parameter arena, the gradient arena and the workspace.
 *
 * Parameters live in ONE flat fp32 arena in nn.Module.parameters() order; the offsets
 * below (in floats) locate each block.  GraphConv layer l: 2*max_deg+1 weight blocks
 * (K_l x width_l each) in the reference order rel_1, self_1, ..., self_0, then the
 * 2*max_deg+1 bias vectors.  nn.Linear blocks are (out, in).  The gradient arena has
 * the same layout.
 */
#define GCMI_MAX_CONV_LAYERS 4
typedef struct gcmi_model_desc {
  int32_t n_layers;                      /* GraphConv layers, <= GCMI_MAX_CONV_LAYERS        */
  int32_t max_deg;                       /* 10                                               */
  int32_t n_feat_in;                     /* atom feature width (75)                          */
  int32_t conv_width[GCMI_MAX_CONV_LAYERS];
  int32_t dense_width;                   /* 128                                              */
  int32_t n_tasks;
  int32_t n_classes;                     /* classification: >= 2; regression: 1              */
  int32_t mode;                          /* 0 classification, 1 regression                   */
  int32_t batch_norm;                    /* 0 / 1                                            */
  int32_t grad_mode;                     /* 0 reference (autograd cut at GraphConv), 1 full  */
  float bn_eps, bn_momentum;
  int64_t off_conv_w[GCMI_MAX_CONV_LAYERS], off_conv_b[GCMI_MAX_CONV_LAYERS];
  int64_t off_bn_gamma[GCMI_MAX_CONV_LAYERS + 1], off_bn_beta[GCMI_MAX_CONV_LAYERS + 1];
  int64_t off_dense_w, off_dense_b, off_head_w, off_head_b;
  int64_t n_params;
  int32_t storage;                       /* what a step writes and reads back: 0 fp32; 1 the activations as bfloat16 (a
                                            copy of the atom features, neighbour sums, GraphConv outputs, pooled rows,
                                            the dense output: one rounding per stored element); 2 also the gradient
                                            STREAMS between kernels (dpool, dy, dS, dXs).  fp32 arithmetic and
                                            accumulation, fp64 BatchNorm sums of the rounded values, fp32 parameters,
                                            parameter gradients and Adam state in every mode.  gcmi_small_*: 1 and 2 are
                                            the same (its gradients never leave L2).  gcmi_model_*: the default shapes
                                            only (widths 64 over 65..80 features, dense 128, BatchNorm on), else
                                            GCMI_ERR_UNSUPPORTED                                                   */
  int32_t reserved_;
} gcmi_model_desc;

typedef struct gcmi_model_io {
  const float* d_atom_features;          /* N x ld_features                                  */
  int64_t ld_features;
  float* d_workspace;                    /* gcmi_model_workspace_floats() floats             */
  float* d_bn_running_mean[GCMI_MAX_CONV_LAYERS + 1];
  float* d_bn_running_var[GCMI_MAX_CONV_LAYERS + 1];
  int64_t* d_bn_batches_tracked[GCMI_MAX_CONV_LAYERS + 1]; /* may be NULL                    */
  /* outputs, all g->n_mols rows (TrimGraphOutput is a view the caller takes):               */
  float* d_logits;                       /* n_mols x (n_tasks*n_classes)  (regression: n_tasks) */
  float* d_probs;                        /* classification softmax; may be NULL              */
  float* d_fingerprint;                  /* n_mols x 2*dense_width                           */
  float* d_loss;                         /* 1 float (loss_backward)                          */
} gcmi_model_io;

/* floats of workspace needed for a batch of n_atoms / n_mols (forward + backward). */
int64_t gcmi_model_workspace_floats(const gcmi_model_desc* m, int64_t n_atoms, int64_t n_mols);
/* training != 0: BatchNorm uses batch statistics and updates the running ones. The
 * graph needs d_mol_runs; the backward additionally uses d_rev_pos when present.            */
int gcmi_model_forward(const gcmi_model_desc* m, const gcmi_graph* g, const float* d_params,
                       const gcmi_model_io* io, int32_t training, void* stream);
/* After a training-mode gcmi_model_forward on the same workspace: loss over the first
 * n_rows molecules (SoftmaxCrossEntropy / L2Loss through _StandardLoss) -> io->d_loss, then
 * the backward pass; gradients are WRITTEN (not accumulated) into d_grads for every
 * parameter the gradient mode trains; [*grad_lo, *grad_hi) returns that range (floats).     */
int gcmi_model_loss_backward(const gcmi_model_desc* m, const gcmi_graph* g, const float* d_params,
                             float* d_grads, const gcmi_model_io* io, const float* d_labels,
                             const float* d_weights, int64_t n_rows, int64_t* grad_lo,
                             int64_t* grad_hi, void* stream);

/* ---------------------------------------------------------------- small-batch engine
 * The same model step for batches whose activations live in L2 (the reference's default batch of 100
 * molecules, MolNet's 64: graphconvmodel.py:292, molnet/preset_hyper_parameters.py:49-56), over MANY batches
 * per call: the loop of TorchModel.fit_generator (torch_model.py:423-445) -- forward, loss, backward, Adam,
 * running statistics -- runs inside the library, 8 launches per step in reference gradient mode and 12 in
 * full mode, on 16-row degree tiles (csrc/smallstep.hip).  Same arithmetic contract as gcmi_model_*: fp32
 * operands and accumulation on v_mfma_f32_16x16x4_f32, BatchNorm statistics in fp64.
 *
 * gcmi_small_batch: one collated batch; every pointer is device memory.  graph needs d_col_idx, d_membership,
 *   d_mol_runs, and for gcmi_small_fit d_rev_pos (GCMI_ERR_UNSUPPORTED without: use gcmi_model_*).  Atom
 *   feature rows must be 16-byte aligned (ld_features % 4 == 0, >= n_feat_in rounded up to 4; pad columns 0).
 * gcmi_small_fit: for i in [0, n_batches): one optimizer step on batches[i] (labels (B, T, C) one-hot for
 *   classification / (B, T) for regression; weights (B, T) or NULL; loss over the first n_rows molecules,
 *   _StandardLoss, torch_model.py:1275-1294); Adam step number first_step + i (1-based; torch.optim.Adam as
 *   optimizers.py:231-241 configures it) on the trained range [*grad_lo, *grad_hi) of the flat arenas;
 *   d_losses[i] = that step's loss; BatchNorm running statistics and counters updated per step.
 *   d_grads: scratch arena of n_params floats (left zero on the trained range).  The workspace
 *   (io->d_workspace) holds gcmi_small_workspace_floats(m, ws_atoms, ws_mols) floats; every batch must fit.
 * gcmi_small_predict: eval-mode forward of every batch into its d_logits (B x T*C), d_probs (classification;
 *   may be NULL) and d_fingerprint (B x 2*dense_width).
 * Widths: GraphConv and dense widths multiples of 64 up to 256 (else GCMI_ERR_UNSUPPORTED).              */
typedef struct gcmi_small_batch {
  gcmi_graph graph;
  const float* d_atom_features;
  int64_t ld_features;
  const float* d_labels;
  const float* d_weights;
  int64_t n_rows;
  float* d_logits;
  float* d_probs;
  float* d_fingerprint;
} gcmi_small_batch;
int64_t gcmi_small_workspace_floats(const gcmi_model_desc* m, int64_t max_atoms, int64_t max_mols);
int gcmi_small_fit(const gcmi_model_desc* m, float* d_params, float* d_grads, float* d_adam_m, float* d_adam_v,
                   const gcmi_model_io* io, const gcmi_small_batch* batches, int64_t n_batches, int64_t ws_atoms,
                   int64_t ws_mols, float lr, float beta1, float beta2, float eps, int64_t first_step,
                   float* d_losses, int64_t* grad_lo, int64_t* grad_hi, void* stream);
/* Data-parallel form (molecules sharded by rank, ONE flat-bucket all-reduce per step; the reference
 * shards disk shards by rank, data/pytorch_datasets.py:104-113, and leaves the gradient exchange to torch DDP):
 * exactly gcmi_small_fit, with `sync` called once per optimizer step between the backward launches and the Adam
 * launch of that step, on the calling thread:
 *     sync(ctx, d_grads + *grad_lo, *grad_hi - *grad_lo, stream)
 * It must ENQUEUE, in order on `stream`, a sum all-reduce over the ranks of that range (and whatever scaling the
 * caller's mean needs) and return 0; a non-zero return aborts the call with GCMI_ERR_LAUNCH.  The library makes no
 * RCCL call itself: the host side owns the communicator (torch.distributed's "nccl" backend = RCCL over xGMI), and
 * the callback is the only thing it has to provide.  sync == NULL: gcmi_small_fit.  BatchNorm statistics stay per
 * rank.  In reference gradient mode the GraphConv stacks of the next steps still run ahead on the second stream:
 * nothing they read is trained, so nothing they read is exchanged.                                         */
typedef int (*gcmi_grad_sync_fn)(void* ctx, float* d_grad_range, int64_t n_floats, void* stream);
int gcmi_small_fit_dp(const gcmi_model_desc* m, float* d_params, float* d_grads, float* d_adam_m, float* d_adam_v,
                      const gcmi_model_io* io, const gcmi_small_batch* batches, int64_t n_batches, int64_t ws_atoms,
                      int64_t ws_mols, float lr, float beta1, float beta2, float eps, int64_t first_step,
                      float* d_losses, int64_t* grad_lo, int64_t* grad_hi, gcmi_grad_sync_fn sync, void* sync_ctx,
                      void* stream);
int gcmi_small_predict(const gcmi_model_desc* m, const float* d_params, const gcmi_model_io* io,
                       const gcmi_small_batch* batches, int64_t n_batches, int64_t ws_atoms, int64_t ws_mols,
                       void* stream);

/* Host side of the small-batch engine: every batch of a chunk of an epoch collated by ONE call into ONE arena
 * (one H2D copy), worker threads taking whole batches; per batch the output is what gcmi_collate_plans writes
 * (ConvMol.agglomerate_mols, feat/mol_graphs.py:256-349; no LDS windows).
 *   sel [batch_ptr[n_batches]] molecule indices, batch b = sel[batch_ptr[b] : batch_ptr[b+1]] (repeats = pad_batch
 *   tiling); mols_out > 0: every batch's graph says mols_out molecules, the ones beyond its selection empty
 *   (GraphGather always emits batch_size rows, layers.py:6469-6479).
 * gcmi_collate_batches_layout -> arena size in 4-byte words (negative: error); out_parts [n_batches x 5] word
 *   offsets of features, membership, col_idx, mol_runs, rev_pos; out_counts [n_batches x 3] atoms, edges, first
 *   feature row.  The feature rows of all batches form ONE contiguous [rows x ld] array at the start of the arena
 *   (every batch padded to an even row count), so 8-byte atom codes (ld = 2) expand in one launch.
 * gcmi_collate_batches fills the arena and graphs[n_batches] (device pointers NULL); out_symmetric[b] = 0 when a
 *   bond of batch b is listed from one end only.
 * gcmi_small_bind turns that into the gcmi_small_batch array once the arena is on the device.               */
int64_t gcmi_collate_batches_layout(const int64_t* atom_ptr, const int64_t* adj_ptr, const int64_t* sel,
                                    const int64_t* batch_ptr, int64_t n_batches, int64_t ld, int32_t max_deg,
                                    int64_t mols_out, int64_t* out_parts, int64_t* out_counts);
int gcmi_collate_batches(const float* atom_features, int64_t n_feat, const int64_t* atom_ptr, const int64_t* adj_ptr,
                         const int32_t* adj_idx, const int64_t* sel, const int64_t* batch_ptr, int64_t n_batches,
                         int32_t max_deg, int64_t ld, int64_t mols_out, float* arena, const int64_t* parts,
                         const int64_t* counts, gcmi_graph* graphs, int32_t* out_symmetric, int32_t n_threads);
int gcmi_small_bind(gcmi_small_batch* out, const gcmi_graph* graphs, const int64_t* parts, const int64_t* counts,
                    int64_t n_batches, const float* d_arena, const float* d_features, int64_t feature_ld,
                    int64_t mols_out, const int64_t* n_rows, const float* d_labels, int64_t label_stride,
                    const float* d_weights, int64_t weight_stride, float* d_logits, float* d_probs,
                    int64_t logit_stride, float* d_fingerprint, int64_t fp_stride);

/* ---------------------------------------------------------------- measurement
 * Optional per-kernel timing with hipEvents recorded on `stream` around the
 * launches of one kernel family (bench.py roofline).  id: see GCMI_K_*.       */
enum {
  GCMI_K_GATHER_SUM = 0,
  GCMI_K_GATHER_MAX = 1,
  GCMI_K_READOUT = 2,
  GCMI_K_SEG_GEMM = 3,
  GCMI_K_WGRAD = 4,
  GCMI_K_GATHER_MAX_BWD = 5,
  GCMI_K_BATCHNORM = 6, /* column sums, finalize, backward (statistics and dx) */
  GCMI_K_FUSED_BWD = 7, /* the one-pass backward of a GraphConv / dense block (bwd_fused.hip) */
  GCMI_K_COUNT = 8
};
int gcmi_timing_enable(int32_t kernel_id, int32_t on);
/* Diagnostic: `blocks` workgroups of 4 waves each issue iters*32 register-only
 * v_mfma_f32_32x32x2_f32 per wave (4096 flops each): the fp32 MFMA ceiling of the device.  */
int gcmi_diag_mfma_peak(int32_t blocks, int32_t iters, float* d_out, void* stream);
/* Synchronises the recorded events; returns launches and total milliseconds. */
int gcmi_timing_read(int32_t kernel_id, int64_t* n_launches, double* total_ms, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* GCMI_H */
