// Forward product of a GraphConv / dense block in the shape of bwd_fused.hip:
//
//     out = relu([S | X] . [W_rel[d]; W_self[d]] + b[d])        GraphConv.forward (models/torch_models/layers.py:6204-6246)
//     out = relu(P . W^T + b)                                   nn.Linear + ReLU (graphconvmodel.py:222-223)
//
// plus, for the training forward, the column sums of out and out^2 for the BatchNorm that follows.  Same arithmetic as
// seg_gemm4_kernel (three-way bf16 split, six products per term, fp32 accumulation); what differs is the schedule.
// seg_gemm4_kernel launches one workgroup per 128-row tile, which stages its K chunks through LDS one barrier pair
// per chunk and splits the weight chunk again for every tile: ~3.4 TB/s on its operands.  Here persistent workgroups
// (one per CU, eight waves) keep the segment's weight images in LDS, prefetch the next tile's rows into registers
// while the current tile is multiplied, and every wave owns one 32 x 32 output tile over the whole contraction:
// two barriers per tile, no per-tile weight work, whole-row stores through LDS.
#include "common.h"
#include "split_bf16.h"

namespace gcmi {

constexpr int kWMaxSeg = 16;

struct FwdTable {
  int32_t n_seg;
  int32_t seg_begin[kWMaxSeg];
  int32_t seg_end[kWMaxSeg];
  int32_t tile_start[kWMaxSeg + 1];
  int64_t w_off[2][kWMaxSeg];  // weight block of operand o; < 0: term absent
  int64_t b_off[kWMaxSeg];     // bias row; < 0: none
};

struct FwdArgs {
  const float* in[2];
  int32_t ldin[2];
  int32_t k_in;          // columns of every operand (<= KO)
  const float* w[2];
  const float* bias;
  float* out;
  int32_t ldo;
  int32_t relu;
  double* stats;         // bn.hip scratch layout, or nullptr
  const u32x4* wimg;     // fwd_reg_kernel: the segments' weight fragments, split and in lane order (fwd_weight_images,
                         // csrc/fwd_bf16.hip)
};

// ROWS rows per tile, NOPS operands of KO (padded) columns each, NOUT output columns; TRANS: weights stored
// NOUT x k_in (nn.Linear) instead of k_in x NOUT
template <int ROWS, int NOPS, int KO, int NOUT, bool TRANS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2)))
fwd_fused_kernel(FwdTable st, int n_tiles, FwdArgs a, int rev) {
  constexpr int NT = 512;
  constexpr int NC = NOPS * KO;               // contraction length
  constexpr int AP = NC + 4;                  // pitch of an operand row in LDS (floats)
  constexpr int WP = NC + 8;                  // pitch of a weight-image row (bf16)
  constexpr int OP = NOUT + 8;                // pitch of an output row in LDS (floats)
  constexpr int RB = ROWS / 32, TW = NOUT / 32;
  static_assert(RB * TW == 8, "eight waves, one 32 x 32 output tile each");
  constexpr int IQ = KO / 4;                  // 16-byte pieces of an operand row
  constexpr int IPASS = ROWS * IQ / NT;       // per operand
  constexpr int OQ = NOUT / 4;
  constexpr int OPASS = ROWS * OQ / NT;
  static_assert(ROWS * IQ % NT == 0 && ROWS * OQ % NT == 0 && NT % OQ == 0, "tile loads and stores divide evenly");
  constexpr int NKS = NC / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* As = reinterpret_cast<float*>(lds_raw);                                       // [ROWS][AP]
  unsigned short* Wimg = reinterpret_cast<unsigned short*>(As + ROWS * AP);            // [3][NOUT][WP]
  float* Outs = reinterpret_cast<float*>(Wimg + (size_t)3 * NOUT * WP);                // [ROWS][OP]
  __shared__ int t_begin_s[kWMaxSeg], t_end_s[kWMaxSeg], t_tile_s[kWMaxSeg + 1];
  __shared__ long long t_w_s[2][kWMaxSeg], t_b_s[kWMaxSeg];
  __shared__ __attribute__((aligned(16))) float bias_s[NOUT];
  __shared__ double stat_s[2][NOUT];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int rb = wave % RB, tw = wave / RB;   // this wave's output tile: rows rb*32.., columns tw*32..

  if (tid <= kWMaxSeg) {
    t_tile_s[tid] = pick_n(st.tile_start, tid);
    if (tid < kWMaxSeg) {
      t_begin_s[tid] = pick_n(st.seg_begin, tid);
      t_end_s[tid] = pick_n(st.seg_end, tid);
      t_w_s[0][tid] = pick_n(st.w_off[0], tid);
      t_w_s[1][tid] = pick_n(st.w_off[1], tid);
      t_b_s[tid] = pick_n(st.b_off, tid);
    }
  }
  for (int c = tid; c < 2 * NOUT; c += NT) stat_s[c / NOUT][c % NOUT] = 0.0;
  const int n_seg = st.n_seg;
  __syncthreads();

  const int b = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const int t_begin = (int)((int64_t)b * n_tiles / gridDim.x);
  const int t_end = (int)((int64_t)(b + 1) * n_tiles / gridDim.x);
  const int my_tiles = t_end - t_begin;  // >= 1: the grid is never larger than the tile count
  auto tile_at = [&](int i) { return rev ? t_end - 1 - i : t_begin + i; };
  // Tile -> (segment, first row, rows) by a cursor that moves with the walk: a workgroup's tiles are consecutive, so the
  // segment changes now and then and a look-up is otherwise two scalar operations.  (The per-tile search it replaces --
  // a loop of LDS reads over the segment starts, each landing in a vector register -- measured ~1 200 cycles per
  // look-up in fwd_hd_kernel's phase clocks, csrc/fwd_bf16.hip.)
  struct Cursor { int seg, t0, t1, r0, r1; } cur;
  auto cur_load = [&]() {
    cur.t0 = __builtin_amdgcn_readfirstlane(t_tile_s[cur.seg]);
    cur.t1 = __builtin_amdgcn_readfirstlane(t_tile_s[cur.seg + 1]);
    cur.r0 = __builtin_amdgcn_readfirstlane(t_begin_s[cur.seg]);
    cur.r1 = __builtin_amdgcn_readfirstlane(t_end_s[cur.seg]);
  };
  {
    const int first = tile_at(0);
    int sg = 0;
    for (int k = 1; k < n_seg; ++k) sg += first >= t_tile_s[k] ? 1 : 0;
    cur.seg = __builtin_amdgcn_readfirstlane(sg);
    cur_load();
  }
  auto tile_info = [&](int tile, int& seg, int& row0, int& valid) {
    while (tile >= cur.t1) { ++cur.seg; cur_load(); }  // (uniform; empty segments are stepped over)
    while (tile < cur.t0) { --cur.seg; cur_load(); }
    seg = cur.seg;
    row0 = cur.r0 + (tile - cur.t0) * ROWS;
    const int left = cur.r1 - row0;
    valid = left < ROWS ? left : ROWS;
  };

  // ---- prefetch registers: the next tile's operand rows, 16 bytes per lane, unconditional from clamped addresses
  float4 pin[NOPS][IPASS];
  auto clampr = [](int r, int valid) { return r < valid ? r : valid - 1; };
  auto load_src = [&](int row0, int valid) {
#pragma unroll
    for (int o = 0; o < NOPS; ++o) {
#pragma unroll
      for (int p = 0; p < IPASS; ++p) {
        const int slot = tid + p * NT;
        const int r = slot / IQ, q = slot - r * IQ;
        const int ld = a.ldin[o];
        const int qc = 4 * q + 4 <= ld ? 4 * q : 0;
        pin[o][p] = *reinterpret_cast<const float4*>(a.in[o] + ((unsigned)(row0 + clampr(r, valid)) * (unsigned)ld + qc));
      }
    }
  };

  // ---- the previous tile's output leaves through LDS as whole rows (bias and ReLU already applied), in program
  // order before the next loads; the BatchNorm sums are taken on the way: fp32 partials per thread over eight tiles
  // (a thread keeps its column piece), then fp64
  float ps1[4] = {0.f, 0.f, 0.f, 0.f}, ps2[4] = {0.f, 0.f, 0.f, 0.f};
  auto flush_stats = [&]() {
    const int q = tid % OQ;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      atomicAdd(&stat_s[0][4 * q + i], (double)ps1[i]);
      atomicAdd(&stat_s[1][4 * q + i], (double)ps2[i]);
      ps1[i] = ps2[i] = 0.f;
    }
  };
  auto store_out = [&](int prow0, int pvalid) {
#pragma unroll
    for (int p = 0; p < OPASS; ++p) {
      const int slot = tid + p * NT;
      const int r = slot / OQ, q = slot - r * OQ;
      if (r < pvalid) {
        const float4 v = *reinterpret_cast<const float4*>(Outs + r * OP + 4 * q);
        *reinterpret_cast<float4*>(a.out + ((unsigned)(prow0 + r) * (unsigned)a.ldo + 4u * q)) = v;
        if (a.stats != nullptr) {
          ps1[0] += v.x; ps1[1] += v.y; ps1[2] += v.z; ps1[3] += v.w;
          ps2[0] = fmaf(v.x, v.x, ps2[0]); ps2[1] = fmaf(v.y, v.y, ps2[1]);
          ps2[2] = fmaf(v.z, v.z, ps2[2]); ps2[3] = fmaf(v.w, v.w, ps2[3]);
        }
      }
    }
  };

  int seg, row0, valid;
  tile_info(tile_at(0), seg, row0, valid);
  int nseg = seg, nrow0 = row0, nvalid = valid;
  if (my_tiles > 1) tile_info(tile_at(1), nseg, nrow0, nvalid);
  load_src(row0, valid);
  int cur_seg = -1;
  int prow0 = row0, pvalid = 0;

  for (int i = 0; i < my_tiles; ++i) {
    if (seg != cur_seg) {
      cur_seg = seg;
      // the segment's weight blocks, stacked along the contraction, as three bf16 images [piece][output column][c]
      // (everyone passed the barrier that ended the previous tile: nobody reads the old images any more)
      constexpr int NE = NC * NOUT;
      for (int e = tid; e < NE; e += NT) {
        int c, n;
        if constexpr (TRANS) {  // w is NOUT x k_in: consecutive threads along c
          n = e / NC;
          c = e - n * NC;
        } else {                // w is k_in x NOUT: consecutive threads along n
          c = e / NOUT;
          n = e - c * NOUT;
        }
        const int o = c / KO, ck = c - o * KO;
        const int64_t woff = t_w_s[o][seg];
        float v = 0.f;
        if (woff >= 0 && ck < a.k_in) {
          const float* w = o == 1 ? a.w[1] : a.w[0];
          v = TRANS ? w[woff + (int64_t)n * a.k_in + ck] : w[woff + (int64_t)ck * NOUT + n];
        }
        unsigned p1, p2, p3;
        split3(v, p1, p2, p3);
        unsigned short* dst = Wimg + (size_t)n * WP + c;
        dst[0] = (unsigned short)(p1 >> 16);
        dst[(size_t)NOUT * WP] = (unsigned short)(p2 >> 16);
        dst[(size_t)2 * NOUT * WP] = (unsigned short)(p3 >> 16);
      }
      const int64_t boff = t_b_s[seg];
      for (int n = tid; n < NOUT; n += NT) bias_s[n] = (a.bias != nullptr && boff >= 0) ? a.bias[boff + n] : 0.f;
    }

    // ---- phase (a): the previous tile's rows out, this tile's operand rows -> LDS
    if (i > 0) {
      store_out(prow0, pvalid);
      if (a.stats != nullptr && (i & 7) == 0) flush_stats();
    }
#pragma unroll
    for (int o = 0; o < NOPS; ++o) {
      const bool present = t_w_s[o][seg] >= 0;
#pragma unroll
      for (int p = 0; p < IPASS; ++p) {
        const int slot = tid + p * NT;
        const int r = slot / IQ, q = slot - r * IQ;
        const int tail = a.k_in - 4 * q;
        float4 v = pin[o][p];
        const bool ok = present && r < valid;
        v.x = (ok && tail > 0) ? v.x : 0.f;
        v.y = (ok && tail > 1) ? v.y : 0.f;
        v.z = (ok && tail > 2) ? v.z : 0.f;
        v.w = (ok && tail > 3) ? v.w : 0.f;
        *reinterpret_cast<float4*>(As + r * AP + o * KO + 4 * q) = v;
      }
    }
    __syncthreads();

    // ---- phase (b): the next tile's rows in flight, this wave's 32 x 32 output tile over the whole contraction
    int n2seg = nseg, n2row0 = nrow0, n2valid = nvalid;
    if (i + 2 < my_tiles) tile_info(tile_at(i + 2), n2seg, n2row0, n2valid);
    load_src(nrow0, nvalid);
    {
      f32x16 acc;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] = 0.f;
      float4 glo, ghi;
      u32x4 wv[2][3];
      auto read_a = [&](int ks) {
        const float* arow = As + (rb * 32 + l31) * AP + ks * 16 + 8 * half;
        glo = *reinterpret_cast<const float4*>(arow);
        ghi = *reinterpret_cast<const float4*>(arow + 4);
      };
      auto read_w = [&](int ks) {
        const unsigned short* wrow = Wimg + (size_t)(tw * 32 + l31) * WP + ks * 16 + 8 * half;
        wv[ks & 1][0] = *reinterpret_cast<const u32x4*>(wrow);
        wv[ks & 1][1] = *reinterpret_cast<const u32x4*>(wrow + (size_t)NOUT * WP);
        wv[ks & 1][2] = *reinterpret_cast<const u32x4*>(wrow + (size_t)2 * NOUT * WP);
      };
      read_a(0);
      read_w(0);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const float v[8] = {glo.x, glo.y, glo.z, glo.w, ghi.x, ghi.y, ghi.z, ghi.w};
        const Frag3 fa = split_frag(v);
        if (ks + 1 < NKS) {  // the next k-step's LDS reads, issued before this k-step's MFMAs
          read_w(ks + 1);
          read_a(ks + 1);
        }
        const u32x4 w1 = wv[ks & 1][0], w2 = wv[ks & 1][1], w3 = wv[ks & 1][2];
        // rows x output columns: lane = output column, registers = rows; small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fa.p[2]), as_bf16x8(w1), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fa.p[0]), as_bf16x8(w3), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fa.p[1]), as_bf16x8(w2), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fa.p[1]), as_bf16x8(w1), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fa.p[0]), as_bf16x8(w2), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fa.p[0]), as_bf16x8(w1), acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // bias, ReLU -> LDS [row][column]: 32 consecutive banks per half-wave
      const float bv = bias_s[tw * 32 + l31];
      float* orow = Outs + (rb * 32 + 4 * half) * OP + tw * 32 + l31;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        float v = acc[reg] + bv;
        if (a.relu) v = v > 0.f ? v : 0.f;
        orow[((reg & 3) + 8 * (reg >> 2)) * OP] = v;
      }
    }
    __syncthreads();
    prow0 = row0; pvalid = valid;
    seg = nseg; row0 = nrow0; valid = nvalid;
    nseg = n2seg; nrow0 = n2row0; nvalid = n2valid;
  }
  store_out(prow0, pvalid);
  if (a.stats != nullptr) {
    flush_stats();
    __syncthreads();
    for (int c = tid; c < 2 * NOUT; c += NT) {
      const int which = c / NOUT, col = c - which * NOUT;
      atomicAdd(a.stats + (size_t)2 * NOUT * (1 + (blockIdx.x % kBnReplicas)) + (size_t)which * NOUT + col,
                stat_s[which][col]);
    }
  }
}

// ---------------------------------------------------------------- weights in registers, two workgroups per CU
// The first GraphConv (two 76-column operands) does not fit the scheme above: its operand tile and weight images need
// more LDS than a CU has at 128 rows.  Here the weight images are not in LDS at all: a workgroup is four waves on a
// 64-row tile, a wave owns a 32 x 32 output tile over the whole contraction and keeps ITS weight fragments -- already
// split -- in registers (120 VGPRs at K = 160; reloaded from the L2-resident weights when the tile loop crosses into
// another degree's segment, no barrier for that).  LDS holds operand rows only (42 KB), so two workgroups share a CU
// and one's loads, stores and barriers overlap the other's products.  The output leaves straight from the
// accumulators (lane = column: a store instruction writes two whole 128-byte lines), the BatchNorm sums are per-lane
// scalars.  Same products in the same order as seg_gemm4_kernel / fwd_fused_kernel.
// Measured (same box): first GraphConv 294 -> 247 us.  For the two shapes fwd_fused_kernel serves, this form (with
// two operand buffers and one barrier per tile) measured 211 / 226 us against 215 / 216: no better, not instantiated.

// Where the lanes of rows beyond a ragged tile's end store (see products()): a word per thread of the largest grid.
// (One shared line would do for correctness, and serialises two million same-address stores in one L2 channel.)
__device__ float g_fwd_dump[512 * 256];

template <int NOPS, int KO, int NOUT, bool TRANS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
fwd_reg_kernel(FwdTable st, int n_tiles, FwdArgs a, int rev) {
  constexpr int NT = 256, ROWS = 64;
  constexpr int NC = NOPS * KO;               // contraction length
  constexpr int AP = NC + 4;                  // pitch of an operand row in LDS (floats)
  constexpr int NKS = NC / 16;
  constexpr int TW = NOUT / 32;               // 32-column tiles of the output
  constexpr int TPW = TW / 2;                 // ... per wave: waves = 2 row blocks x 2 column groups
  static_assert(TW % 2 == 0 && KO % 8 == 0 && NC % 16 == 0, "tile shapes");
  constexpr int IQ = KO / 4;                  // 16-byte pieces of an operand row
  constexpr int IPASS = ROWS * IQ / NT;       // per operand
  static_assert(ROWS * IQ % NT == 0, "tile loads divide evenly");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* As = reinterpret_cast<float*>(lds_raw);  // [ROWS][AP]
  __shared__ int t_begin_s[kWMaxSeg], t_end_s[kWMaxSeg], t_tile_s[kWMaxSeg + 1];
  __shared__ long long t_w_s[2][kWMaxSeg], t_b_s[kWMaxSeg];
  __shared__ double stat_s[2][NOUT];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int rb = wave & 1, twb = wave >> 1;   // rows rb*32.., column tiles twb, twb + 2, ...
  float* const my_dump = &g_fwd_dump[(blockIdx.x % 512) * 256 + tid];

  if (tid <= kWMaxSeg) {
    t_tile_s[tid] = pick_n(st.tile_start, tid);
    if (tid < kWMaxSeg) {
      t_begin_s[tid] = pick_n(st.seg_begin, tid);
      t_end_s[tid] = pick_n(st.seg_end, tid);
      t_w_s[0][tid] = pick_n(st.w_off[0], tid);
      t_w_s[1][tid] = pick_n(st.w_off[1], tid);
      t_b_s[tid] = pick_n(st.b_off, tid);
    }
  }
  for (int c = tid; c < 2 * NOUT; c += NT) stat_s[c / NOUT][c % NOUT] = 0.0;
  const int n_seg = st.n_seg;
  __syncthreads();

  const int b = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
  // (64-bit division runs on the vector unit: say that the results are uniform)
  const int t_begin = __builtin_amdgcn_readfirstlane((int)((int64_t)b * n_tiles / gridDim.x));
  const int t_end = __builtin_amdgcn_readfirstlane((int)((int64_t)(b + 1) * n_tiles / gridDim.x));
  const int my_tiles = t_end - t_begin;  // >= 1: the grid is never larger than the tile count
  auto tile_at = [&](int i) { return rev ? t_end - 1 - i : t_begin + i; };
  // Tile -> (segment, first row, rows) by a cursor that moves with the walk: a workgroup's tiles are consecutive, so the
  // segment changes now and then and a look-up is otherwise two scalar operations.  (The per-tile search it replaces --
  // a loop of LDS reads over the segment starts, each landing in a vector register -- measured ~1 200 cycles per
  // look-up in fwd_hd_kernel's phase clocks, csrc/fwd_bf16.hip.)
  struct Cursor { int seg, t0, t1, r0, r1; } cur;
  auto cur_load = [&]() {
    cur.t0 = __builtin_amdgcn_readfirstlane(t_tile_s[cur.seg]);
    cur.t1 = __builtin_amdgcn_readfirstlane(t_tile_s[cur.seg + 1]);
    cur.r0 = __builtin_amdgcn_readfirstlane(t_begin_s[cur.seg]);
    cur.r1 = __builtin_amdgcn_readfirstlane(t_end_s[cur.seg]);
  };
  {
    const int first = tile_at(0);
    int sg = 0;
    for (int k = 1; k < n_seg; ++k) sg += first >= t_tile_s[k] ? 1 : 0;
    cur.seg = __builtin_amdgcn_readfirstlane(sg);
    cur_load();
  }
  auto tile_info = [&](int tile, int& seg, int& row0, int& valid) {
    while (tile >= cur.t1) { ++cur.seg; cur_load(); }  // (uniform; empty segments are stepped over)
    while (tile < cur.t0) { --cur.seg; cur_load(); }
    seg = cur.seg;
    row0 = cur.r0 + (tile - cur.t0) * ROWS;
    const int left = cur.r1 - row0;
    valid = left < ROWS ? left : ROWS;
  };

  // ---- prefetch registers: the next tile's operand rows, 16 bytes per lane, unconditional from clamped addresses.
  // Addresses stay (uniform base, 32-bit byte offset) formed at the load: anything the compiler can hoist out of the
  // tile loop as a 64-bit per-lane value it does, and then spills it.
  constexpr bool kEven = NT % IQ == 0;        // every pass of a thread has the same 16-byte column
  constexpr int RSTEP = NT / IQ;
  float4 pin[NOPS][IPASS];
  auto slot_rq = [&](int p, int& r, int& q) {
    if constexpr (kEven) {
      r = tid / IQ + p * RSTEP;
      q = tid % IQ;
    } else {
      int slot = tid + p * NT;
      asm volatile("" : "+v"(slot));  // a multiply and a shift at each use, instead of ten hoisted (and spilled) values
      r = slot / IQ;
      q = slot - r * IQ;
    }
  };
  auto load_src = [&](int row0, int valid) {
#pragma unroll
    for (int o = 0; o < NOPS; ++o) {
#pragma unroll
      for (int p = 0; p < IPASS; ++p) {
        int r, q;
        slot_rq(p, r, q);
        const int ld = a.ldin[o];
        const int qc = 4 * q + 4 <= ld ? 4 * q : 0;
        const int rc = r < valid ? r : valid - 1;
        unsigned off = ((unsigned)(row0 + rc) * (unsigned)ld + (unsigned)qc) * 4u;
        asm volatile("" : "+v"(off));
        pin[o][p] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.in[o]) + off);
      }
    }
  };
  // the prefetched rows -> LDS; what lies outside the tile (ragged tail, columns beyond k_in, an operand the segment
  // does not have) is zeroed
  auto write_as = [&](int seg_, int valid_) {
    bool whole = valid_ == ROWS && a.k_in == KO;
#pragma unroll
    for (int o = 0; o < NOPS; ++o) whole = whole && t_w_s[o][seg_] >= 0;
#pragma unroll
    for (int o = 0; o < NOPS; ++o) {
      const bool present = t_w_s[o][seg_] >= 0;
#pragma unroll
      for (int p = 0; p < IPASS; ++p) {
        int r, q;
        slot_rq(p, r, q);
        float4 v = pin[o][p];
        if (!whole) {  // uniform
          const int tail = a.k_in - 4 * q;
          const bool ok = present && r < valid_;
          v.x = (ok && tail > 0) ? v.x : 0.f;
          v.y = (ok && tail > 1) ? v.y : 0.f;
          v.z = (ok && tail > 2) ? v.z : 0.f;
          v.w = (ok && tail > 3) ? v.w : 0.f;
        }
        *reinterpret_cast<float4*>(As + r * AP + o * KO + 4 * q) = v;
      }
    }
  };

  // ---- this wave's weight fragments of the segment, split, and its bias column(s)
  u32x4 wf[TPW][NKS][3];
  float bv[TPW];
  auto load_w = [&](int seg_) {
    // prepared images ([segment][32-column tile][k-step][piece][lane], fwd_weight_images): coalesced 1 KiB wave loads
    // and ONE wait.  (Read from the fp32 parameter block a fragment was eight strided dword loads per lane, which the
    // compiler, short of registers, issued one at a time, each behind a full wait: 80 dependent L2 round trips per
    // reload, 20-40 us that every workgroup paid at its start -- DESIGN 23.3.)
    const u32x4* base = a.wimg + (size_t)seg_ * TW * NKS * 3 * 64 + lane;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) wf[j][ks][pc] = base[(size_t)(((twb + 2 * j) * NKS + ks) * 3 + pc) * 64];
      const int64_t boff = t_b_s[seg_];
      bv[j] = (a.bias != nullptr && boff >= 0) ? a.bias[boff + (twb + 2 * j) * 32 + l31] : 0.f;
    }
  };

  // BatchNorm sums of this lane's output column(s): fp32 over up to eight tiles, then fp64 in LDS
  float ps1[TPW], ps2[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) ps1[j] = ps2[j] = 0.f;
  auto flush_stats = [&]() {
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      atomicAdd(&stat_s[0][(twb + 2 * j) * 32 + l31], (double)ps1[j]);
      atomicAdd(&stat_s[1][(twb + 2 * j) * 32 + l31], (double)ps2[j]);
      ps1[j] = ps2[j] = 0.f;
    }
  };

  // this wave's output tile(s): products, bias, ReLU, store, sums
  auto products = [&](int row0_, int valid_) {
    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[j][k] = 0.f;
    float4 glo, ghi;
    auto read_a = [&](int ks) {
      const float* arow = As + (rb * 32 + l31) * AP + ks * 16 + 8 * half;
      glo = *reinterpret_cast<const float4*>(arow);
      ghi = *reinterpret_cast<const float4*>(arow + 4);
    };
    read_a(0);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const float v[8] = {glo.x, glo.y, glo.z, glo.w, ghi.x, ghi.y, ghi.z, ghi.w};
      const Frag3 fa = split_frag(v);
      if (ks + 1 < NKS) read_a(ks + 1);  // the next k-step's LDS read, issued before this k-step's MFMAs
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const u32x4 w1 = wf[j][ks][0], w2 = wf[j][ks][1], w3 = wf[j][ks][2];
        // rows x output columns: lane = output column, registers = rows; small terms first
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fa.p[2]), as_bf16x8(w1), acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fa.p[0]), as_bf16x8(w3), acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fa.p[1]), as_bf16x8(w2), acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fa.p[1]), as_bf16x8(w1), acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fa.p[0]), as_bf16x8(w2), acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fa.p[0]), as_bf16x8(w1), acc[j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const int rl0 = rb * 32 + 4 * half;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      char* const obase = reinterpret_cast<char*>(a.out);
      const unsigned o0 = ((unsigned)(row0_ + rl0) * (unsigned)a.ldo + (unsigned)((twb + 2 * j) * 32 + l31)) * 4u;
      float s1 = 0.f, s2 = 0.f;
      // One path for whole and ragged tiles, sixteen stores either way: the rows beyond a ragged tile's end go to a
      // dump word.  (With the stores under branches, or in two alternative blocks, the compiler cannot count the stores
      // that follow the next tile's loads in the memory queue -- one in-order counter -- and makes write_as() wait for
      // all of them instead of s_waitcnt vmcnt(16 + ...): the stores' latency would be serialised with the next tile.)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r8 = (reg & 3) + 8 * (reg >> 2);
        const bool live = rl0 + r8 < valid_;
        float v = acc[j][reg] + bv[j];
        if (a.relu) v = v > 0.f ? v : 0.f;
        unsigned off = o0 + (unsigned)r8 * (unsigned)a.ldo * 4u;
        asm volatile("" : "+v"(off));  // formed at the store: sixteen hoisted offsets would not fit the registers
        float* dst = live ? reinterpret_cast<float*>(obase + off) : my_dump;
        *dst = v;
        v = live ? v : 0.f;
        s1 += v;
        s2 = fmaf(v, v, s2);
      }
      ps1[j] += s1;
      ps2[j] += s2;
    }
  };

  int seg, row0, valid;
  tile_info(tile_at(0), seg, row0, valid);
  int nseg = seg, nrow0 = row0, nvalid = valid;
  if (my_tiles > 1) tile_info(tile_at(1), nseg, nrow0, nvalid);
  load_src(row0, valid);
  int cur_seg = seg;
  load_w(seg);

  for (int i = 0; i < my_tiles; ++i) {
    write_as(seg, valid);
    __syncthreads();
    int n2seg = nseg, n2row0 = nrow0, n2valid = nvalid;
    if (i + 2 < my_tiles) tile_info(tile_at(i + 2), n2seg, n2row0, n2valid);
    load_src(nrow0, nvalid);  // ahead of this tile's stores in the memory queue
    if (seg != cur_seg) {  // uniform
      cur_seg = seg;
      load_w(seg);
    }
    products(row0, valid);
    if (a.stats != nullptr && (i & 7) == 7) flush_stats();
    __syncthreads();
    seg = nseg; row0 = nrow0; valid = nvalid;
    nseg = n2seg; nrow0 = n2row0; nvalid = n2valid;
  }
  if (a.stats != nullptr) {
    flush_stats();
    __syncthreads();
    for (int c = tid; c < 2 * NOUT; c += NT) {
      const int which = c / NOUT, col = c - which * NOUT;
      atomicAdd(a.stats + (size_t)2 * NOUT * (1 + (blockIdx.x % kBnReplicas)) + (size_t)which * NOUT + col,
                stat_s[which][col]);
    }
  }
}

static bool fwd_fused_on() {
  static const int env = getenv("GCMI_FUSED_FWD") ? atoi(getenv("GCMI_FUSED_FWD")) : 1;
  return env != 0;
}

template <int ROWS, int NOPS, int KO, int NOUT, bool TRANS>
static int launch_fwd(const FwdTable& st, int n_tiles, const FwdArgs& a, hipStream_t sm) {
  constexpr int NC = NOPS * KO;
  const size_t shmem = sizeof(float) * ROWS * (NC + 4) + sizeof(unsigned short) * (size_t)3 * NOUT * (NC + 8) +
                       sizeof(float) * ROWS * (NOUT + 8);
  auto kern = fwd_fused_kernel<ROWS, NOPS, KO, NOUT, TRANS>;
  static bool attr_done = false;  // per instantiation
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)shmem) != hipSuccess) {
      (void)hipGetLastError();
      return GCMI_ERR_UNSUPPORTED;
    }
    attr_done = true;
  }
  const int grid = std::min(n_tiles, 256);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), shmem, sm, st, n_tiles, a, next_sweep_direction());
  GCMI_CHECK_LAUNCH("fwd_fused");
  return GCMI_OK;
}

static bool fwd_reg_on() {
  static const int env = getenv("GCMI_FWD_REG") ? atoi(getenv("GCMI_FWD_REG")) : 1;
  return env != 0;
}

template <int NOPS, int KO, int NOUT, bool TRANS>
static int launch_fwd_reg(const FwdTable& st, int n_tiles, const FwdArgs& a, hipStream_t sm) {
  constexpr int NC = NOPS * KO;
  const size_t shmem = sizeof(float) * 64 * (NC + 4);
  auto kern = fwd_reg_kernel<NOPS, KO, NOUT, TRANS>;
  static bool attr_done = false;  // per instantiation
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)shmem) != hipSuccess) {
      (void)hipGetLastError();
      return GCMI_ERR_UNSUPPORTED;
    }
    attr_done = true;
  }
  const int grid = std::min(n_tiles, 512);  // two workgroups per CU (g_fwd_dump is sized for this)
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, sm, st, n_tiles, a, next_sweep_direction());
  GCMI_CHECK_LAUNCH("fwd_reg");
  return GCMI_OK;
}

// The shapes of the default model in split-bf16 mode: two operands of 65..80 columns -> 64 columns (the first
// GraphConv: fwd_reg_kernel), two 64-column operands -> 64 columns (GraphConv over pooled rows), one 64-column
// operand -> 128 columns in nn.Linear layout (the atom-level dense layer).  Anything else:
// GCMI_ERR_UNSUPPORTED, and the caller runs seg_gemm4_kernel.  *fused: the BatchNorm sums were added to d_stats.
int fwd_fused_gemm(int32_t n_seg, const int32_t* seg_begin, const int32_t* seg_end, const float* d_a1, int64_t lda1,
                   int32_t k1, const float* d_w1, const int64_t* w1_off, const float* d_a2, int64_t lda2, int32_t k2,
                   const float* d_w2, const int64_t* w2_off, const float* d_bias, const int64_t* bias_off,
                   int32_t n_out, int32_t trans_w, int32_t act, float* d_out, int64_t ldo, double* d_stats,
                   hipStream_t sm, float* d_wimg_scratch) {
  if (!fwd_fused_on() || !fused_bwd_enabled() || n_seg > kWMaxSeg || (act != 0 && act != 1)) return GCMI_ERR_UNSUPPORTED;
  const bool two = d_a1 != nullptr && d_a2 != nullptr;
  const bool conv = two && !trans_w && n_out == 64 && k1 == k2 && k1 > 32 && k1 <= 64;
  const bool conv80 = fwd_reg_on() && two && !trans_w && n_out == 64 && k1 == k2 && k1 > 64 && k1 <= 80;
  const bool dense = !two && d_a1 != nullptr && trans_w && n_out == 128 && k1 > 32 && k1 <= 64;
  if (!conv && !dense && !conv80) return GCMI_ERR_UNSUPPORTED;
  if (!aligned16(d_a1) || lda1 % 4 || (two && (!aligned16(d_a2) || lda2 % 4)) || !aligned16(d_out) || ldo % 4 ||
      (d_bias && !aligned16(d_bias)))
    return GCMI_ERR_UNSUPPORTED;
  int64_t rows = 0;
  for (int s = 0; s < n_seg; ++s) rows = std::max<int64_t>(rows, seg_end[s]);
  if (rows * std::max(std::max(lda1, two ? lda2 : 0), ldo) >= (int64_t)1 << 30) return GCMI_ERR_UNSUPPORTED;
  const int tile_rows = conv ? 128 : 64;
  FwdTable st;
  memset(&st, 0, sizeof(st));
  st.n_seg = n_seg;
  int64_t tiles = 0;
  for (int s = 0; s < kWMaxSeg; ++s) {
    st.tile_start[s] = (int32_t)tiles;
    st.w_off[0][s] = st.w_off[1][s] = st.b_off[s] = -1;
    if (s < n_seg) {
      st.seg_begin[s] = seg_begin[s];
      st.seg_end[s] = seg_end[s];
      st.w_off[0][s] = w1_off ? w1_off[s] : -1;
      st.w_off[1][s] = (two && w2_off) ? w2_off[s] : -1;
      st.b_off[s] = (d_bias && bias_off) ? bias_off[s] : -1;
      tiles += (seg_end[s] - seg_begin[s] + tile_rows - 1) / tile_rows;
    }
  }
  st.tile_start[kWMaxSeg] = (int32_t)tiles;
  if (tiles == 0) return GCMI_OK;
  FwdArgs a;
  memset(&a, 0, sizeof(a));
  a.in[0] = d_a1; a.ldin[0] = (int32_t)lda1; a.in[1] = d_a2; a.ldin[1] = (int32_t)lda2; a.k_in = k1;
  a.w[0] = d_w1; a.w[1] = d_w2; a.bias = d_bias; a.out = d_out; a.ldo = (int32_t)ldo; a.relu = act; a.stats = d_stats;
  if (conv80) {
    // (needs the caller's scratch for the weight images; without it the shape goes to seg_gemm4_kernel)
    if (d_wimg_scratch == nullptr ||
        fwd_weight_images(n_seg, w1_off, w2_off, d_w1, d_w2, k1, 80, 2, 64, 0, d_wimg_scratch, sm) != GCMI_OK)
      return GCMI_ERR_UNSUPPORTED;
    a.wimg = reinterpret_cast<const u32x4*>(d_wimg_scratch);
    return launch_fwd_reg<2, 80, 64, false>(st, (int)tiles, a, sm);
  }
  if (conv) return launch_fwd<128, 2, 64, 64, false>(st, (int)tiles, a, sm);
  return launch_fwd<64, 1, 64, 128, true>(st, (int)tiles, a, sm);
}

}  // namespace gcmi
