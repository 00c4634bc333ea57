// One pass over the rows for everything the backward of a GraphConv / dense block does with the gradient of its
// pre-activation (autograd of models/torch_models/layers.py:6204-6246 and graphconvmodel.py:222-229):
//
//     G   = relu'(x) * (A*dy + B*x + C)              BatchNorm backward, elementwise part (bn.hip), never written
//     dW_o += In_o^T . G   (o = rel, self | dense)   weight gradients, bias gradient = column sums of G
//     dIn_o = G . W_o^T                              gradients of the block's inputs
//
// The separate kernels (bn_bwd_dx, two wgrad launches, two dgrad launches) each stream N x 64..128 floats; together
// they move 896 floats per atom for a K = 64 GraphConv and 640 (+ per-molecule rows) for the dense layer.  Here a
// 64-row tile of G is formed once in LDS from the prefetched dy / x rows (for the dense layer dy itself is recomputed
// from the per-molecule GraphGather gradient, readout_dy of bn.hip) and both products read it from there:
// 384 / 256 floats per atom.
//
// Workgroup = 8 waves on one 64-row tile of ONE degree segment: waves 0-3 form dIn (32 rows x 32..64 columns each,
// weights as three bf16 images in LDS, built once per segment), waves 4-7 accumulate dW in registers over all tiles
// of the segment the workgroup walks (one (operand, 32-column group of G) each; the rows of In arrive as per-lane
// dword loads one tile ahead, as in wgrad3_kernel) and add them to the gradient with one float atomic per element at
// the end of the segment.  Arithmetic = the exact three-way bf16 split of gemm_split.hip (six products per term on
// v_mfma_f32_32x32x16_bf16, fp32 accumulation).  A block that needs no input gradient (the first GraphConv) runs
// the four weight-gradient waves only.
#include <atomic>
#include <type_traits>

#include "common.h"
#include "split_bf16.h"

namespace gcmi {

constexpr int kFRows = 64;
constexpr int kFMaxSeg = 16;

struct FusedTable {
  int32_t n_seg;
  int32_t seg_begin[kFMaxSeg];
  int32_t seg_end[kFMaxSeg];
  int32_t tile_start[kFMaxSeg + 1];
  int64_t w_off[2][kFMaxSeg];  // weight block of operand o = the same block of the gradient; < 0: term absent
  int64_t db_off[kFMaxSeg];    // bias-gradient row; < 0: none
};

struct FusedArgs {
  // sources of G
  // (leading dimensions and element offsets are 32-bit: every array is < 4 GB, checked by the launchers; an
  // address is then a scalar base plus one 32-bit register instead of a 64-bit register pair per load stream)
  const float* dy;       // incoming gradient rows (nullptr: recomputed from the readout gradient)
  int32_t lddy;
  const float* x;        // BatchNorm input = the block's ReLU output
  int32_t ldx;
  const float* coef;     // [A | B | C], 3 * NG floats; nullptr: G = relu'(x) * dy
  const int32_t* membership;
  const float* g2;       // n_mols x ldg2: [dsum | dmax]
  int32_t ldg2;
  const int32_t* arg;    // n_mols x NG
  // weight-gradient operands
  const float* in[2];
  int32_t ldin[2];
  int32_t k_in;
  const float* w;
  float* dw;
  float* db;
  // input gradients (DGRAD)
  float* dout[2];
  int32_t lddout[2];
  // DGRAD, optional: column sums for the BatchNorm backward of the block BELOW this one, whose output P is this
  // block's input.  With dP this block's input gradient, sum dP and sum dP * P give that BatchNorm's sum dy and
  // sum dy * y without a pass over dy (bn.hip: bn_bwd_params_pool_kernel).  For a GraphConv block
  // dP = dXs + gather(dS), and over a symmetric adjacency sum gather(dS) * P = sum dS * gather(P) = sum dS * S:
  // both tiles and both operands are in LDS here.  Scratch layout of bn.hip: [2F unused][replica][F | F] doubles.
  double* psums;
  // diagnostics (GCMI_FUSED_DIAG=1): 100 MHz ticks spent per phase by wave 0 and wave 4 (or 3) of workgroup 0
  unsigned long long* diag;
};

// HB: the block's forward activations (x = its ReLU output, In = its inputs) are stored as bf16
// (gcmi_model_desc.storage == 1): they arrive as 8-byte pieces of four elements -- half the prefetch registers -- and
// are widened on their way to LDS; a weight-gradient fragment of In then IS its own first bf16 piece (one v_perm_b32
// per pair instead of the three-way split).  The products of this form keep the terms above 2^-16: the result of every
// forward product was itself rounded to 2^-9 when it was stored, so the third pieces of G and W buy nothing here --
// dW += In^T G is In x the two leading pieces of G (2 MFMAs per fragment instead of 6), dIn = G W^T the three products
// g1 w1 + g1 w2 + g2 w1 (3 instead of 6): the matrix pipe, which two waves per SIMD share and which bounds these
// kernels together with the split's vector work, does half the work.  Accumulation stays fp32; gradients (dy, the
// outputs) stay fp32.
// GB (with HB; storage == 2): the gradient STREAMS are bf16 as well -- dy arrives as 8-byte pieces, the input gradients
// leave rounded once (the column sums for the BatchNorm below are taken from the rounded values: they describe what the
// next kernel reads).
template <int NG, int KT, int NOPS, bool TRANS, bool RD, bool DGRAD, bool HB = false, bool GB = false>
__global__ void __launch_bounds__(DGRAD ? 512 : 256) __attribute__((amdgpu_waves_per_eu(2)))
fused_bwd_kernel(FusedTable st, int n_tiles, FusedArgs a, int rev) {
  static_assert(!GB || HB, "bf16 gradient streams come with bf16 activations");
  constexpr int NT = DGRAD ? 512 : 256;
  constexpr int NJ = NG / 32;                 // 32-column groups of G
  constexpr int KP = KT * 32;                 // padded width of In
  constexpr int GP = NG + 4;                  // pitch of a G row in LDS (floats): 16-byte rows
  constexpr int IP = KP + 4;                  // pitch of an In row in LDS
  constexpr int WP = NG + 8;                  // pitch of a weight-image row (bf16): conflict-free b128 reads
  constexpr size_t kWBytes = !DGRAD ? 0 : sizeof(unsigned short) * (size_t)NOPS * 3 * KP * WP;
  constexpr int QPR = NG / 4;                 // 16-byte pieces of a G row
  constexpr int RPP = NT / QPR;               // rows per pass of the tile loads
  constexpr int GPASS = kFRows / RPP;         // passes
  constexpr int IQ = KP / 4;                  // 16-byte pieces of an In row
  constexpr int IPASS = kFRows * IQ / NT;     // passes of the In loads, per operand
  constexpr int STEPS = kFRows / 16;          // k-steps of the weight-gradient contraction
  // the weight-gradient waves interleave the split of the next fragment with the MFMAs of the current one; the
  // dense-layer form has no registers left for that (80 hold its prefetched rows) and only reads a fragment ahead
  constexpr bool INTERLEAVE = !RD;
  constexpr int OT = NOPS * KT;               // 32-column tiles of the input gradients
  constexpr int TPW = OT / 2;                 // ... per dgrad wave (two waves per 32-row block)
  constexpr int OP = OT * 32 + 8;             // pitch of a row of the input-gradient tile in LDS (floats)
  constexpr int OQ = OT * 8;                  // its 16-byte pieces
  constexpr int OPASS = DGRAD ? kFRows * OQ / NT : 1;
  static_assert(NOPS * NJ == 4, "four weight-gradient waves: one (operand, column group) each");
  static_assert(!DGRAD || (OT % 2 == 0), "input-gradient tiles split over two waves per row block");
  static_assert(kFRows * IQ % NT == 0, "In tile loads divide evenly");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* Gs = reinterpret_cast<float*>(lds_raw);
  float* Ins = Gs + kFRows * GP;                                   // [NOPS][64][IP]
  unsigned short* Wimg = reinterpret_cast<unsigned short*>(Ins + NOPS * kFRows * IP);
  float* Outs = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(Wimg) + kWBytes);  // [64][OP] (DGRAD)
  // the segment table, in LDS (indexing the by-value struct dynamically would go through scratch)
  __shared__ int t_begin_s[kFMaxSeg], t_end_s[kFMaxSeg], t_tile_s[kFMaxSeg + 1];
  __shared__ long long t_w_s[2][kFMaxSeg], t_db_s[kFMaxSeg];
  __shared__ __attribute__((aligned(16))) float coef_s[3 * NG];  // [A | B | C] of the BatchNorm backward
  __shared__ double stat_s[2][DGRAD ? KP : 1];                   // sum dP | sum dP * P per input column (psums)

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: role tests and table picks stay off the vector unit
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const bool is_dgrad = DGRAD && wave < 4;
  const int wq = wave & 3;
  // weight-gradient wave: operand wo, column group wj of G
  const int wo = wq / NJ, wj = wq % NJ;
  // input-gradient wave: row block rb, output tiles [cg * TPW, cg * TPW + TPW)
  const int rb = wq & 1, cg = wq >> 1;

  if (tid <= kFMaxSeg) {
    t_tile_s[tid] = pick_n(st.tile_start, tid);
    if (tid < kFMaxSeg) {
      t_begin_s[tid] = pick_n(st.seg_begin, tid);
      t_end_s[tid] = pick_n(st.seg_end, tid);
      t_w_s[0][tid] = pick_n(st.w_off[0], tid);
      t_w_s[1][tid] = pick_n(st.w_off[1], tid);
      t_db_s[tid] = pick_n(st.db_off, tid);
    }
  }
  const int n_seg = st.n_seg;

  const int b = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const int t_begin = (int)((int64_t)b * n_tiles / gridDim.x);
  const int t_end = (int)((int64_t)(b + 1) * n_tiles / gridDim.x);
  const int my_tiles = t_end - t_begin;  // >= 1: the grid is never larger than the tile count
  auto tile_at = [&](int i) { return rev ? t_end - 1 - i : t_begin + i; };

  // ---- per-thread constants of the G phase: this thread's 16-byte column piece
  const int gq = tid % QPR;
  const int gr = tid / QPR;
  for (int c = tid; c < 3 * NG; c += NT) coef_s[c] = a.coef != nullptr ? a.coef[c] : (c < NG ? 1.f : 0.f);
  if constexpr (DGRAD)
    for (int c = tid; c < 2 * KP; c += NT) stat_s[c / KP][c % KP] = 0.0;

  __syncthreads();

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
    row0 = cur.r0 + (tile - cur.t0) * kFRows;
    const int left = cur.r1 - row0;
    valid = left < kFRows ? left : kFRows;
  };

  // ---- prefetch registers (one tile ahead; the molecule index two tiles ahead): all loads are 16 bytes per lane,
  // unconditional, from clamped addresses; what lies outside the tile is zeroed when it goes to LDS
  using ActV = typename std::conditional<HB, uint2, float4>::type;  // four stored activations
  auto load_act = [](const float* base, unsigned elem) -> ActV {
    if constexpr (HB) return *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(base) + elem);
    else return *reinterpret_cast<const float4*>(base + elem);
  };
  auto act4 = [](const ActV& v) -> float4 {
    if constexpr (HB) return widen4(v);
    else return v;
  };
  using DyV = typename std::conditional<GB && !RD, uint2, float4>::type;  // four incoming gradients
  DyV pdy[GPASS];
  float4 pgm[RD ? GPASS : 1];
  ActV px[GPASS], pin[NOPS][IPASS];
  int4 parg[RD ? GPASS : 1];
  int mem1[RD ? GPASS : 1];
  auto clampr = [](int r, int valid) { return r < valid ? r : valid - 1; };
  auto load_mem = [&](int row0, int valid, int (&mem)[RD ? GPASS : 1]) {
    if constexpr (RD) {
#pragma unroll
      for (int p = 0; p < GPASS; ++p) mem[p] = a.membership[(unsigned)(row0 + clampr(gr + p * RPP, valid))];
    }
  };
  auto load_src = [&](int row0, int valid) {
#pragma unroll
    for (int p = 0; p < GPASS; ++p) {
      const unsigned r = (unsigned)(row0 + clampr(gr + p * RPP, valid));
      px[p] = load_act(a.x, r * (unsigned)a.ldx + 4u * gq);
      if constexpr (RD) {
        const unsigned m = (unsigned)mem1[p];
        pdy[p] = *reinterpret_cast<const float4*>(a.g2 + (m * (unsigned)a.ldg2 + 4u * gq));
        pgm[p] = *reinterpret_cast<const float4*>(a.g2 + (m * (unsigned)a.ldg2 + NG + 4u * gq));
        parg[p] = *reinterpret_cast<const int4*>(a.arg + (m * (unsigned)NG + 4u * gq));
      } else {
        if constexpr (GB) pdy[p] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(a.dy) + (r * (unsigned)a.lddy + 4u * gq));
        else pdy[p] = *reinterpret_cast<const float4*>(a.dy + (r * (unsigned)a.lddy + 4u * gq));
      }
    }
#pragma unroll
    for (int o = 0; o < NOPS; ++o) {
#pragma unroll
      for (int p = 0; p < IPASS; ++p) {
        const int slot = tid + p * NT;
        const int r = slot / IQ, q = slot - r * IQ;
        const int ld = a.ldin[o];
        const int qc = 4 * q + 4 <= ld ? 4 * q : 0;
        pin[o][p] = load_act(a.in[o], (unsigned)(row0 + clampr(r, valid)) * (unsigned)ld + qc);
      }
    }
  };

  // ---- weight-gradient state
  // one set of accumulator registers for both roles: a weight-gradient wave keeps dW in it across the tiles of a
  // segment, an input-gradient wave starts every tile from zero
  constexpr int NACC = (DGRAD && TPW > KT) ? TPW : KT;
  f32x16 accs[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) accs[t][i] = 0.f;
  float bsum = 0.f;
  auto flush_w = [&](int seg) {
    const int64_t woff = t_w_s[NOPS == 2 ? wo : 0][seg];
    if (woff >= 0) {
      // this lane's first element; its offset is made opaque so that the 32 addresses behind it are formed here
      // and not hoisted out of the tile loop into 64 registers
      int64_t woff_lane = woff + (TRANS ? (int64_t)(wj * 32 + 4 * half) * a.k_in + l31
                                        : (int64_t)(4 * half) * NG + wj * 32 + l31);
      asm volatile("" : "+v"(woff_lane));  // (an offset, not the pointer: the pointer keeps its address space)
      float* wp = a.dw + woff_lane;
#pragma unroll
      for (int t = 0; t < KT; ++t) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int r8 = (reg & 3) + 8 * (reg >> 2);  // + 4 * half: row of the accumulator tile
          if constexpr (TRANS) {  // dW stored NG x k_in: the accumulator holds (column of G) x (column of In)
            if (t * 32 + l31 < a.k_in) atomicAdd(wp + (int64_t)r8 * a.k_in + t * 32, accs[t][reg]);
          } else {                // dW stored k_in x NG
            if (t * 32 + r8 + 4 * half < a.k_in) atomicAdd(wp + (t * 32 + r8) * NG, accs[t][reg]);
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) accs[t][i] = 0.f;
    if (wo == 0) {
      const int64_t boff = t_db_s[seg];
      const float s = bsum + __shfl_xor(bsum, 32);
      if (half == 0 && a.db != nullptr && boff >= 0) atomicAdd(a.db + boff + wj * 32 + l31, s);
    }
    bsum = 0.f;
  };

  static_assert(!DGRAD || NT % IQ == 0, "a thread keeps its column piece over the passes");
  // The input-gradient tile leaves through LDS: the waves that computed it hold (column, 16 rows) per lane, which as
  // global stores would be 32 dword stores per lane -- and, issued behind the next tile's loads, every wave's wait
  // for those loads (one instruction stream for both roles: the wait cannot tell which role it is in) would also
  // wait for the store acknowledgements.  Written out by all threads at the start of the next tile instead: whole
  // rows, 16 bytes per lane, and in program order BEFORE the loads that follow.
  // (thread -> element mapping = that of the In tile loads: a thread reads here exactly the In elements it is about
  // to overwrite with the next tile's, so no barrier is needed between the two)
  // (the sums: fp32 partials per thread over a few tiles -- a thread keeps its column piece --, then fp64 in LDS:
  // per-element fp64 products and LDS atomics cost a microsecond per tile)
  float ps1[4] = {0.f, 0.f, 0.f, 0.f}, ps2[4] = {0.f, 0.f, 0.f, 0.f};
  auto flush_psums = [&]() {
    if constexpr (DGRAD) {
      const int q = tid % IQ;  // NT % IQ == 0: the same column piece in every pass
      if (4 * q < a.k_in) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          atomicAdd(&stat_s[0][4 * q + i], (double)ps1[i]);
          atomicAdd(&stat_s[1][4 * q + i], (double)ps2[i]);
          ps1[i] = ps2[i] = 0.f;
        }
      }
    }
  };
  auto store_out = [&](int prow0, int pvalid, int pseg) {
    if constexpr (DGRAD) {
      static_assert(OQ == NOPS * IQ, "one input-gradient row piece per In row piece");
      const float deg = (float)pseg;  // segment = degree block: a row of dS is gathered by `deg` neighbours
#pragma unroll
      for (int o = 0; o < NOPS; ++o) {
        float* dst = o == 1 ? a.dout[1] : a.dout[0];
        const int ldd = o == 1 ? a.lddout[1] : a.lddout[0];
#pragma unroll
        for (int p = 0; p < IPASS; ++p) {
          const int slot = tid + p * NT;
          const int r = slot / IQ, q = slot - r * IQ;
          if (r < pvalid && 4 * q < a.k_in) {  // k_in % 4 == 0 (launcher)
            float4 v = *reinterpret_cast<const float4*>(Outs + r * OP + o * KP + 4 * q);
            if constexpr (GB) {
              const uint2 h = narrow4(v.x, v.y, v.z, v.w);
              *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(dst) + ((unsigned)(prow0 + r) * (unsigned)ldd + 4u * q)) = h;
              v = widen4(h);  // the sums below describe the stored values
            } else {
              *reinterpret_cast<float4*>(dst + ((unsigned)(prow0 + r) * (unsigned)ldd + 4u * q)) = v;
            }
            if (a.psums != nullptr) {
              const float4 in = *reinterpret_cast<const float4*>(Ins + (o * kFRows + r) * IP + 4 * q);
              const float wgt = (NOPS == 2 && o == 0) ? deg : 1.f;
              ps1[0] = fmaf(wgt, v.x, ps1[0]); ps1[1] = fmaf(wgt, v.y, ps1[1]);
              ps1[2] = fmaf(wgt, v.z, ps1[2]); ps1[3] = fmaf(wgt, v.w, ps1[3]);
              ps2[0] = fmaf(v.x, in.x, ps2[0]); ps2[1] = fmaf(v.y, in.y, ps2[1]);
              ps2[2] = fmaf(v.z, in.z, ps2[2]); ps2[3] = fmaf(v.w, in.w, ps2[3]);
            }
          }
        }
      }
    }
  };

  // ---- prologue: sources of the first tile, molecule indices of the first two
  int seg, row0, valid;
  tile_info(tile_at(0), seg, row0, valid);
  int nseg = seg, nrow0 = row0, nvalid = valid;  // the tile after this one (itself when there is none)
  if (my_tiles > 1) tile_info(tile_at(1), nseg, nrow0, nvalid);
  load_mem(row0, valid, mem1);
  load_src(row0, valid);
  load_mem(nrow0, nvalid, mem1);  // mem1 now describes tile 1: its sources are requested in phase (b) of tile 0
  int cur_seg = -1;
  int prow0 = row0, pvalid = 0, pseg = 0;

  const bool stamp = a.diag != nullptr && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == NT / 64 - 1);
  unsigned long long d_a = 0, d_w1 = 0, d_b = 0, d_w2 = 0, t0 = 0, t1 = 0;
  for (int i = 0; i < my_tiles; ++i) {
    if (stamp) t0 = wall_clock64();
    if (seg != cur_seg) {
      if (cur_seg >= 0 && !is_dgrad) flush_w(cur_seg);
      cur_seg = seg;
      if constexpr (DGRAD) {
        // the segment's weight blocks as three bf16 images [operand][piece][input column][column of G]
        // (everyone passed the barrier that ended the previous tile: nobody reads the old images any more)
        constexpr int NE = NOPS * KP * NG;
        for (int e = tid; e < NE; e += NT) {
          int o, ko, c;
          if constexpr (TRANS) {  // w is NG x k_in: consecutive threads along k_in
            o = 0;
            c = e / KP;
            ko = e - c * KP;
          } else {                // w is k_in x NG: consecutive threads along NG
            o = e / (KP * NG);
            const int r = e - o * (KP * NG);
            ko = r / NG;
            c = r - ko * NG;
          }
          const int64_t woff = t_w_s[NOPS == 2 ? (o & 1) : 0][seg];
          float v = 0.f;
          if (woff >= 0 && ko < a.k_in)
            v = TRANS ? a.w[woff + (int64_t)c * a.k_in + ko] : a.w[woff + (int64_t)ko * NG + c];
          unsigned p1, p2, p3;
          split3(v, p1, p2, p3);
          unsigned short* dst = Wimg + ((size_t)(o * 3) * KP + ko) * WP + c;
          dst[0] = (unsigned short)(p1 >> 16);
          dst[(size_t)KP * WP] = (unsigned short)(p2 >> 16);
          dst[(size_t)2 * KP * WP] = (unsigned short)(p3 >> 16);
        }
      }
    }

    // ---- phase (a): the previous tile's input gradients out (they were written before the last barrier; this also
    // reads the previous In rows, so it comes first), then G and the In rows of this tile -> LDS
    if (i > 0) {
      store_out(prow0, pvalid, pseg);
      if (a.psums != nullptr && (i & 7) == 0) flush_psums();
    }
#pragma unroll
    for (int p = 0; p < GPASS; ++p) {
      const int r = gr + p * RPP;
      float4 dy4;
      if constexpr (GB && !RD) dy4 = widen4(pdy[p]);
      else dy4 = pdy[p];
      float dyv[4] = {dy4.x, dy4.y, dy4.z, dy4.w};
      const float4 x4 = act4(px[p]);
      const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
      if constexpr (RD) {
        const int rg = row0 + r;
        const float gm[4] = {pgm[p].x, pgm[p].y, pgm[p].z, pgm[p].w};
        const int av[4] = {parg[p].x, parg[p].y, parg[p].z, parg[p].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) dyv[q] = dyv[q] + (av[q] == rg ? gm[q] : 0.f);
      }
      const float4 cA4 = *reinterpret_cast<const float4*>(coef_s + 4 * gq);
      const float4 cB4 = *reinterpret_cast<const float4*>(coef_s + NG + 4 * gq);
      const float4 cC4 = *reinterpret_cast<const float4*>(coef_s + 2 * NG + 4 * gq);
      const float cA[4] = {cA4.x, cA4.y, cA4.z, cA4.w}, cB[4] = {cB4.x, cB4.y, cB4.z, cB4.w};
      const float cC[4] = {cC4.x, cC4.y, cC4.z, cC4.w};
      float o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v = fmaf(cA[q], dyv[q], fmaf(cB[q], xv[q], cC[q]));
        o[q] = (r < valid && xv[q] > 0.f) ? v : 0.f;
      }
      *reinterpret_cast<float4*>(Gs + r * GP + 4 * gq) = make_float4(o[0], o[1], o[2], o[3]);
    }
#pragma unroll
    for (int o = 0; o < NOPS; ++o) {
#pragma unroll
      for (int p = 0; p < IPASS; ++p) {
        const int slot = tid + p * NT;
        const int r = slot / IQ, q = slot - r * IQ;
        const int tail = a.k_in - 4 * q;
        float4 v = act4(pin[o][p]);
        const bool ok = r < valid;
        v.x = (ok && tail > 0) ? v.x : 0.f;
        v.y = (ok && tail > 1) ? v.y : 0.f;
        v.z = (ok && tail > 2) ? v.z : 0.f;
        v.w = (ok && tail > 3) ? v.w : 0.f;
        *reinterpret_cast<float4*>(Ins + (o * kFRows + r) * IP + 4 * q) = v;
      }
    }
    if (stamp) { t1 = wall_clock64(); d_a += t1 - t0; }
    __syncthreads();
    if (stamp) { t0 = wall_clock64(); d_w1 += t0 - t1; }

    // ---- phase (b): next tile's sources in flight, products from LDS
    int n2seg = nseg, n2row0 = nrow0, n2valid = nvalid;
    if (i + 2 < my_tiles) tile_info(tile_at(i + 2), n2seg, n2row0, n2valid);
    load_src(nrow0, nvalid);             // uses mem1 = molecule indices of the next tile (read when the loads issue)
    load_mem(n2row0, n2valid, mem1);     // ... which the indices of the tile after it then replace
    if (is_dgrad) {
      if constexpr (DGRAD) {
        // input gradients of this tile: `accs` starts from zero (the weight-gradient waves keep theirs across tiles)
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
          for (int k = 0; k < 16; ++k) accs[t][k] = 0.f;
        bool on[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const int ot = cg * TPW + t;
          on[t] = t_w_s[NOPS == 2 ? ((ot / KT) & 1) : 0][seg] >= 0;
        }
        // units (k-step, output tile), the LDS reads of unit u + 1 issued before the split and the MFMAs of unit u:
        // with two waves per SIMD nobody else hides a read's latency
        constexpr int NKS = NG / 16, NU = NKS * TPW;
        float4 glo, ghi;
        u32x4 wv[2][3];
        auto read_g = [&](int ks) {
          const float* grow = Gs + (rb * 32 + l31) * GP + ks * 16 + 8 * half;
          glo = *reinterpret_cast<const float4*>(grow);
          ghi = *reinterpret_cast<const float4*>(grow + 4);
        };
        auto read_w = [&](int u) {
          const int ks = u / TPW, t = u - ks * TPW;
          const int ot = cg * TPW + t;
          const int o = ot / KT, kt = ot - o * KT;
          const unsigned short* wrow = Wimg + ((size_t)(o * 3) * KP + kt * 32 + l31) * WP + ks * 16 + 8 * half;
          wv[u & 1][0] = *reinterpret_cast<const u32x4*>(wrow);
          wv[u & 1][1] = *reinterpret_cast<const u32x4*>(wrow + (size_t)KP * WP);
          if constexpr (!HB) wv[u & 1][2] = *reinterpret_cast<const u32x4*>(wrow + (size_t)2 * KP * WP);
        };
        read_g(0);
        read_w(0);
        Frag3 fg;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int ks = u / TPW, t = u - ks * TPW;
          if (t == 0) {  // this k-step's rows of G: split, then its registers take the next k-step's
            const float v[8] = {glo.x, glo.y, glo.z, glo.w, ghi.x, ghi.y, ghi.z, ghi.w};
            fg = split_frag(v);
          }
          if (u + 1 < NU) {
            read_w(u + 1);
            if (t == TPW - 1) read_g(ks + 1);
          }
          if (on[t]) {  // uniform
            const u32x4 w1 = wv[u & 1][0], w2 = wv[u & 1][1];
            // rows of G x input columns: lane = input column, registers = rows; small terms first
            if constexpr (!HB) {
              const u32x4 w3 = wv[u & 1][2];
              accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fg.p[2]), as_bf16x8(w1), accs[t], 0, 0, 0);
              accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fg.p[0]), as_bf16x8(w3), accs[t], 0, 0, 0);
              accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fg.p[1]), as_bf16x8(w2), accs[t], 0, 0, 0);
            }
            accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fg.p[1]), as_bf16x8(w1), accs[t], 0, 0, 0);
            accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fg.p[0]), as_bf16x8(w2), accs[t], 0, 0, 0);
            accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fg.p[0]), as_bf16x8(w1), accs[t], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);  // units stay in order: one set of fragments live besides the one in flight
        }
        // -> LDS, [row][tile * 32 + column]: 32 consecutive banks per half-wave, the halves 32 banks apart
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          float* orow = Outs + (rb * 32 + 4 * half) * OP + (cg * TPW + t) * 32 + l31;
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) orow[((reg & 3) + 8 * (reg >> 2)) * OP] = accs[t][reg];
        }
      }
    } else {
      const bool on = t_w_s[NOPS == 2 ? wo : 0][seg] >= 0;
      const float* irow = Ins + (NOPS == 2 ? wo : 0) * kFRows * IP;
      // Fragments in order: per 16-row step the G columns, then the KT column groups of In.  The eight LDS reads of
      // fragment f + 1 are issued before fragment f is split and multiplied: with two waves per SIMD nobody else
      // hides their latency.
      constexpr int FPS = 1 + KT, NF = STEPS * FPS;
      if constexpr (INTERLEAVE) {
      float raw[8];
      auto read_frag = [&](int f) {
        const int st_ = f / FPS, w_ = f - st_ * FPS;
        const float* src = w_ == 0 ? Gs + (16 * st_ + 8 * half) * GP + wj * 32 + l31            // zero beyond `valid`
                                   : irow + (16 * st_ + 8 * half) * IP + (w_ - 1) * 32 + l31;
        const int pitch = w_ == 0 ? GP : IP;
#pragma unroll
        for (int u = 0; u < 8; ++u) raw[u] = src[u * pitch];
      };
      // Software pipeline over fragments: in iteration f the split of fragment f + 1 (vector unit) is interleaved
      // with the six MFMAs of fragment f (a wave issues in order: behind a chain of dependent MFMAs nothing of its
      // own would issue for 6 x 32 cycles), and the LDS reads of fragment f + 2 follow the split that frees `raw`.
      read_frag(0);
      Frag3 fg, cur, nxt;
      {
        if (wo == 0) {
#pragma unroll
          for (int u = 0; u < 8; ++u) bsum += raw[u];
        }
        nxt = split_frag(raw);  // fragment 0 is a G fragment
        if (NF > 1) read_frag(1);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int w_ = f % FPS;
        cur = nxt;
        if (w_ == 0) fg = cur;
        const bool more = f + 1 < NF;
        if (more && (f + 1) % FPS == 0 && wo == 0) {
#pragma unroll
          for (int u = 0; u < 8; ++u) bsum += raw[u];
        }
        unsigned q[3][4];
        const bool next_is_g = (f + 1) % FPS == 0;  // (compile-time under the unrolling)
        auto pair_split = [&](int i) {
          if (!more) return;
          if (HB && !next_is_g) {  // a fragment of stored bf16 values: exact in one piece
            q[0][i] = pack_exact_bf16x2(raw[2 * i], raw[2 * i + 1]);
            q[1][i] = q[2][i] = 0u;
          } else {
            split3_pair(raw[2 * i], raw[2 * i + 1], q[0][i], q[1][i], q[2][i]);
          }
        };
        if (HB && w_ != 0 && on) {  // uniform: In in one piece x G in three
          const int t = w_ - 1;
          const Frag3& L = TRANS ? fg : cur;
          const Frag3& R = TRANS ? cur : fg;
          // (In is `cur`: TRANS -> the right operand R.p[0], else the left operand L.p[0]; G in its two leading pieces,
          // the smaller first)
          pair_split(0);
          pair_split(1);
          __builtin_amdgcn_sched_barrier(0);
          accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(TRANS ? L.p[1] : L.p[0]),
                                                            as_bf16x8(TRANS ? R.p[0] : R.p[1]), accs[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          pair_split(2);
          pair_split(3);
          __builtin_amdgcn_sched_barrier(0);
          if (more && f + 2 < NF) read_frag(f + 2);  // `raw` is free: the reads of the fragment after next
          __builtin_amdgcn_sched_barrier(0);
          accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[0]), as_bf16x8(R.p[0]), accs[t], 0, 0, 0);
        } else if (w_ != 0 && on) {  // uniform
          const int t = w_ - 1;
          const Frag3& L = TRANS ? fg : cur;
          const Frag3& R = TRANS ? cur : fg;
          // MFMA, a quarter of the next fragment's split, MFMA, ...: the order is pinned (the scheduler would put the
          // six dependent MFMAs back to back)
          accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[2]), as_bf16x8(R.p[0]), accs[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          pair_split(0);
          __builtin_amdgcn_sched_barrier(0);
          accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[0]), as_bf16x8(R.p[2]), accs[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          pair_split(1);
          __builtin_amdgcn_sched_barrier(0);
          accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[1]), as_bf16x8(R.p[1]), accs[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          pair_split(2);
          __builtin_amdgcn_sched_barrier(0);
          accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[1]), as_bf16x8(R.p[0]), accs[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          pair_split(3);
          __builtin_amdgcn_sched_barrier(0);
          accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[0]), as_bf16x8(R.p[1]), accs[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (more && f + 2 < NF) read_frag(f + 2);  // `raw` is free: the reads of the fragment after next
          __builtin_amdgcn_sched_barrier(0);
          accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[0]), as_bf16x8(R.p[0]), accs[t], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) pair_split(i);
          if (more && f + 2 < NF) read_frag(f + 2);
        }
        if (more) {
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) {
            nxt.p[pc].x = q[pc][0]; nxt.p[pc].y = q[pc][1]; nxt.p[pc].z = q[pc][2]; nxt.p[pc].w = q[pc][3];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      } else {
      float raw[2][8];
      auto read_frag = [&](int f) {
        const int st_ = f / FPS, w_ = f - st_ * FPS;
        const float* src = w_ == 0 ? Gs + (16 * st_ + 8 * half) * GP + wj * 32 + l31            // zero beyond `valid`
                                   : irow + (16 * st_ + 8 * half) * IP + (w_ - 1) * 32 + l31;
        const int pitch = w_ == 0 ? GP : IP;
#pragma unroll
        for (int u = 0; u < 8; ++u) raw[f & 1][u] = src[u * pitch];
      };
      read_frag(0);
      Frag3 fg;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int w_ = f % FPS;
        if (f + 1 < NF) read_frag(f + 1);
        if (w_ == 0) {
          if (wo == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) bsum += raw[f & 1][u];
          }
          fg = split_frag(raw[f & 1]);
        } else if (on) {  // uniform
          const int t = w_ - 1;
          if constexpr (HB) {  // In in one exact piece x G in three
            u32x4 fi;
            fi.x = pack_exact_bf16x2(raw[f & 1][0], raw[f & 1][1]); fi.y = pack_exact_bf16x2(raw[f & 1][2], raw[f & 1][3]);
            fi.z = pack_exact_bf16x2(raw[f & 1][4], raw[f & 1][5]); fi.w = pack_exact_bf16x2(raw[f & 1][6], raw[f & 1][7]);
#pragma unroll
            for (int pc = 1; pc >= 0; --pc)  // G in its two leading pieces (terms above 2^-16)
              accs[t] = TRANS ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fg.p[pc]), as_bf16x8(fi), accs[t], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fi), as_bf16x8(fg.p[pc]), accs[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            continue;
          }
          const Frag3 fa = split_frag(raw[f & 1]);
          const Frag3& L = TRANS ? fg : fa;
          const Frag3& R = TRANS ? fa : fg;
          accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[2]), as_bf16x8(R.p[0]), accs[t], 0, 0, 0);
          accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[0]), as_bf16x8(R.p[2]), accs[t], 0, 0, 0);
          accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[1]), as_bf16x8(R.p[1]), accs[t], 0, 0, 0);
          accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[1]), as_bf16x8(R.p[0]), accs[t], 0, 0, 0);
          accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[0]), as_bf16x8(R.p[1]), accs[t], 0, 0, 0);
          accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(L.p[0]), as_bf16x8(R.p[0]), accs[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      }
    }
    if (stamp) { t1 = wall_clock64(); d_b += t1 - t0; }
    __syncthreads();
    if (stamp) { t0 = wall_clock64(); d_w2 += t0 - t1; }
    prow0 = row0; pvalid = valid; pseg = seg;
    seg = nseg; row0 = nrow0; valid = nvalid;
    nseg = n2seg; nrow0 = n2row0; nvalid = n2valid;
  }
  store_out(prow0, pvalid, pseg);
  if (!is_dgrad) flush_w(cur_seg);
  if constexpr (DGRAD) {
    if (a.psums != nullptr) {
      flush_psums();
      __syncthreads();
      for (int c = tid; c < 2 * KP; c += NT) {
        const int which = c / KP, col = c - which * KP;
        if (col < a.k_in)
          atomicAdd(a.psums + (size_t)2 * a.k_in * (1 + (blockIdx.x % kBnReplicas)) + (size_t)which * a.k_in + col,
                    stat_s[which][col]);
      }
    }
  }
  if (stamp) {
    unsigned long long* d = a.diag + (wave == 0 ? 0 : 5);
    d[0] = d_a; d[1] = d_w1; d[2] = d_b; d[3] = d_w2; d[4] = (unsigned long long)my_tiles;
  }
}

static bool fused_bwd_on() {
  static const int env = getenv("GCMI_FUSED_BWD") ? atoi(getenv("GCMI_FUSED_BWD")) : 1;
  return env != 0;
}

static std::atomic<int> g_fused_bwd{1};
static std::atomic<int> g_fused_launches{0};
int fused_bwd_launches() { return g_fused_launches.load(std::memory_order_relaxed); }
void set_fused_bwd(int on) { g_fused_bwd.store(on ? 1 : 0, std::memory_order_relaxed); }
int get_fused_bwd() { return g_fused_bwd.load(std::memory_order_relaxed); }
bool fused_bwd_enabled() { return fused_bwd_on() && get_fused_bwd() != 0 && !gemm_exact_mode(); }

template <int NG, int KT, int NOPS, bool TRANS, bool RD, bool DGRAD, bool HB = false, bool GB = false>
static int launch_fused(const FusedTable& st, int n_tiles, const FusedArgs& a, hipStream_t sm) {
  constexpr int KP = KT * 32;
  size_t shmem = sizeof(float) * kFRows * (NG + 4) + sizeof(float) * (size_t)NOPS * kFRows * (KP + 4);
  if (DGRAD)
    shmem += sizeof(unsigned short) * (size_t)NOPS * 3 * KP * (NG + 8) + sizeof(float) * kFRows * (NOPS * KT * 32 + 8);
  auto kern = fused_bwd_kernel<NG, KT, NOPS, TRANS, RD, DGRAD, HB, GB>;
  static bool attr_done = false;  // per instantiation
  if (!attr_done) {
    // exactly what is asked for: the kernel also has a few hundred bytes of static LDS (the segment table)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)shmem) != hipSuccess) {
      (void)hipGetLastError();
      return GCMI_ERR_UNSUPPORTED;
    }
    attr_done = true;
  }
  const int per_cu = DGRAD ? 1 : 2;
  const int grid = std::min(n_tiles, 256 * per_cu);
  static const bool diag_on = getenv("GCMI_FUSED_DIAG") && atoi(getenv("GCMI_FUSED_DIAG")) != 0;
  static unsigned long long* d_diag = nullptr;
  FusedArgs aa = a;
  if (diag_on) {
    if (!d_diag && hipMalloc(&d_diag, 10 * sizeof(unsigned long long)) != hipSuccess) d_diag = nullptr;
    if (d_diag) (void)hipMemsetAsync(d_diag, 0, 10 * sizeof(unsigned long long), sm);
    aa.diag = d_diag;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(DGRAD ? 512 : 256), shmem, sm, st, n_tiles, aa, next_sweep_direction());
  GCMI_CHECK_LAUNCH("fused_bwd");
  if (diag_on && d_diag) {  // serialises the stream: diagnostics only
    unsigned long long h[10];
    if (hipStreamSynchronize(sm) == hipSuccess && hipMemcpy(h, d_diag, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
      fprintf(stderr, "fused_bwd<%d,%d,%d> tiles %llu | first wave: G phase %.2f us, wait %.2f, products %.2f, wait %.2f "
                      "| last wave: %.2f %.2f %.2f %.2f (per tile)\n", NG, KT, (int)DGRAD, h[4],
              h[0] * 0.01 / h[4], h[1] * 0.01 / h[4], h[2] * 0.01 / h[4], h[3] * 0.01 / h[4], h[5] * 0.01 / h[4],
              h[6] * 0.01 / h[4], h[7] * 0.01 / h[4], h[8] * 0.01 / h[4]);
  }
  g_fused_launches.fetch_add(1, std::memory_order_relaxed);
  return GCMI_OK;
}

static int make_table(int32_t n_seg, const int32_t* seg_begin, const int32_t* seg_end, const int64_t* w0_off,
                      const int64_t* w1_off, const int64_t* db_off, FusedTable* st) {
  memset(st, 0, sizeof(*st));
  st->n_seg = n_seg;
  int64_t tiles = 0;
  for (int s = 0; s < kFMaxSeg; ++s) {
    st->tile_start[s] = (int32_t)tiles;
    st->w_off[0][s] = st->w_off[1][s] = st->db_off[s] = -1;
    if (s < n_seg) {
      st->seg_begin[s] = seg_begin[s];
      st->seg_end[s] = seg_end[s];
      st->w_off[0][s] = w0_off ? w0_off[s] : -1;
      st->w_off[1][s] = w1_off ? w1_off[s] : -1;
      st->db_off[s] = db_off ? db_off[s] : -1;
      tiles += (seg_end[s] - seg_begin[s] + kFRows - 1) / kFRows;
    }
  }
  st->tile_start[kFMaxSeg] = (int32_t)tiles;
  return (int)tiles;
}

// GraphConv block: dW_rel += S^T G, dW_self += X^T G, dbsum += colsum G, and (d_ds != nullptr) dS = G W_rel^T,
// dXs = G W_self^T.  GCMI_ERR_UNSUPPORTED = shape not covered (the caller runs the separate kernels).
int fused_conv_bwd(int32_t n_seg, const int32_t* seg_begin, const int32_t* seg_end, const int64_t* w_rel,
                   const int64_t* w_self, const int64_t* b_off, const float* d_dy, int64_t lddy, const float* d_gc,
                   int64_t ldgc, const float* d_coef, int32_t width, const float* d_s, int64_t lds, const float* d_x,
                   int64_t ldx, int32_t k_in, const float* d_w, float* d_dw, float* d_dbsum, float* d_ds_out,
                   int64_t ldds, float* d_dxs_out, int64_t lddxs, double* d_psums, hipStream_t sm, int32_t act_bf16) {
  if (!fused_bwd_enabled() || n_seg > kFMaxSeg || width != 64) return GCMI_ERR_UNSUPPORTED;
  {  // 32-bit element offsets inside the kernel: every array below 2^30 elements
    int64_t rows = 0;
    for (int sgi = 0; sgi < n_seg; ++sgi) rows = std::max<int64_t>(rows, seg_end[sgi]);
    const int64_t ldmax = std::max(std::max(lddy, ldgc), std::max(std::max(lds, ldx), std::max(ldds, lddxs)));
    if (rows * ldmax >= (int64_t)1 << 30) return GCMI_ERR_UNSUPPORTED;
  }
  if (!aligned16(d_dy) || lddy % 4 || !aligned16(d_gc) || ldgc % 4) return GCMI_ERR_UNSUPPORTED;
  if (d_coef && !aligned16(d_coef)) return GCMI_ERR_UNSUPPORTED;
  // (act_bf16: d_gc, d_s and d_x point to bf16 rows, their leading dimensions count elements; 8-byte pieces)
  if (act_bf16 && (!aligned16(d_s) || lds % 4 || !aligned16(d_x) || ldx % 4)) return GCMI_ERR_UNSUPPORTED;
  const bool dgrad = d_ds_out != nullptr;
  FusedTable st;
  const int tiles = make_table(n_seg, seg_begin, seg_end, w_rel, w_self, b_off, &st);
  if (tiles == 0) return GCMI_OK;
  FusedArgs a;
  memset(&a, 0, sizeof(a));
  a.dy = d_dy; a.lddy = (int32_t)lddy; a.x = d_gc; a.ldx = (int32_t)ldgc; a.coef = d_coef;
  a.in[0] = d_s; a.ldin[0] = (int32_t)lds; a.in[1] = d_x; a.ldin[1] = (int32_t)ldx; a.k_in = k_in;
  a.w = d_w; a.dw = d_dw; a.db = d_dbsum;
  a.dout[0] = d_ds_out; a.lddout[0] = (int32_t)ldds; a.dout[1] = d_dxs_out; a.lddout[1] = (int32_t)lddxs;
  a.psums = dgrad ? d_psums : nullptr;
  if (dgrad) {
    if (k_in % 4 || ldds % 4 || lddxs % 4 || !aligned16(d_ds_out) || !aligned16(d_dxs_out)) return GCMI_ERR_UNSUPPORTED;
    // (act_bf16 == 2: dy, dS and dXs are bf16 rows too)
    if (k_in > 32 && k_in <= 64)
      return act_bf16 == 2 ? launch_fused<64, 2, 2, false, false, true, true, true>(st, tiles, a, sm)
             : act_bf16    ? launch_fused<64, 2, 2, false, false, true, true>(st, tiles, a, sm)
                           : launch_fused<64, 2, 2, false, false, true>(st, tiles, a, sm);
    return GCMI_ERR_UNSUPPORTED;
  }
  if (k_in > 32 && k_in <= 64)
    return act_bf16 == 2 ? launch_fused<64, 2, 2, false, false, false, true, true>(st, tiles, a, sm)
           : act_bf16    ? launch_fused<64, 2, 2, false, false, false, true>(st, tiles, a, sm)
                         : launch_fused<64, 2, 2, false, false, false>(st, tiles, a, sm);
  if (k_in > 64 && k_in <= 96)
    return act_bf16 == 2 ? launch_fused<64, 3, 2, false, false, false, true, true>(st, tiles, a, sm)
           : act_bf16    ? launch_fused<64, 3, 2, false, false, false, true>(st, tiles, a, sm)
                         : launch_fused<64, 3, 2, false, false, false>(st, tiles, a, sm);
  return GCMI_ERR_UNSUPPORTED;
}

// Dense layer behind the GraphGather: dy recomputed from the per-molecule gradient, dW (n_out x k_in, nn.Linear
// layout) += G^T P, db += colsum G, dP = G W.
int fused_dense_bwd(int64_t n_rows, const int32_t* d_membership, const float* d_g2, int64_t ldg2,
                    const int32_t* d_arg, const float* d_dense, int64_t ldd, const float* d_coef, int32_t width,
                    const float* d_p, int64_t ldp, int32_t k_in, const float* d_w, float* d_dw, float* d_db,
                    float* d_dp, int64_t lddp, double* d_psums, hipStream_t sm, int32_t act_bf16) {
  if (!fused_bwd_enabled() || width != 128 || k_in <= 32 || k_in > 64 || d_coef == nullptr) return GCMI_ERR_UNSUPPORTED;
  if (!aligned16(d_g2) || ldg2 % 4 || !aligned16(d_arg) || !aligned16(d_dense) || ldd % 4 || !aligned16(d_coef))
    return GCMI_ERR_UNSUPPORTED;
  if (n_rows <= 0 || n_rows > INT32_MAX || k_in % 4 || lddp % 4 || !aligned16(d_dp)) return GCMI_ERR_UNSUPPORTED;
  if (n_rows * std::max<int64_t>(std::max(ldd, ldg2), std::max(ldp, lddp)) >= (int64_t)1 << 30) return GCMI_ERR_UNSUPPORTED;
  const int32_t zero = 0, nn = (int32_t)n_rows;
  const int64_t off0 = 0;
  FusedTable st;
  const int tiles = make_table(1, &zero, &nn, &off0, nullptr, &off0, &st);
  FusedArgs a;
  memset(&a, 0, sizeof(a));
  a.x = d_dense; a.ldx = (int32_t)ldd; a.coef = d_coef;
  a.membership = d_membership; a.g2 = d_g2; a.ldg2 = (int32_t)ldg2; a.arg = d_arg;
  a.in[0] = d_p; a.ldin[0] = (int32_t)ldp; a.k_in = k_in;
  a.w = d_w; a.dw = d_dw; a.db = d_db;
  a.dout[0] = d_dp; a.lddout[0] = (int32_t)lddp;
  a.psums = d_psums;
  if (act_bf16) {  // d_dense and d_p point to bf16 rows
    if (!aligned16(d_p) || ldp % 4) return GCMI_ERR_UNSUPPORTED;
    if (act_bf16 == 2) return launch_fused<128, 2, 1, true, true, true, true, true>(st, tiles, a, sm);  // d_dp: bf16 rows
    return launch_fused<128, 2, 1, true, true, true, true>(st, tiles, a, sm);
  }
  return launch_fused<128, 2, 1, true, true, true>(st, tiles, a, sm);
}

}  // namespace gcmi
