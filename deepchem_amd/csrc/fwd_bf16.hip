// Forward product of a GraphConv / dense block over activations STORED AS bf16 (gcmi_model_desc.storage == 1):
//
//     out = relu([S | X] . [W_rel[d]; W_self[d]] + b[d])        GraphConv.forward (models/torch_models/layers.py:6204-6246)
//     out = relu(P . W^T + b)                                   nn.Linear + ReLU (graphconvmodel.py:222-223)
//
// plus, for the training forward, the column sums of out and out^2 (of the ROUNDED values: the statistics describe
// what is stored and what the backward reads) for the BatchNorm that follows.
//
// What bf16 storage changes against fwd_fused.hip, beyond half the bytes per row:
//   * a stored operand IS its own first bf16 piece: no three-way split of the activations (the vector work that bounds
//     the fp32 kernels), and the product needs the weight pieces x ONE operand piece = 3 MFMAs per k-step instead of 6
//     (fp32 weights split exactly as before, fp32 accumulation);
//   * the operand tile goes to LDS as it arrives (16-byte pieces of 8 elements, no conversion) and a fragment of
//     v_mfma_f32_32x32x16_bf16 is ONE ds_read_b128: half the LDS traffic and a quarter of the fragment instructions;
//   * LDS per workgroup is 25-35 KB (operand tile + output tile, the weight fragments live in registers as in
//     fwd_reg_kernel), so three or four four-wave workgroups share a CU and one's loads, stores and barriers overlap the
//     others' products.
// The weights are the matrix core's A operand and the activations its B operand, so the accumulator holds out^T
// (lane = row of the tile, registers = four runs of four consecutive output columns): bias, ReLU and the rounding are
// applied there, a lane writes 8-byte pieces of its row to an LDS output tile, and the tile leaves as whole rows
// (16 bytes per lane) at the start of the next iteration, where the BatchNorm sums are taken from the rounded values.
#include "common.h"
#include "split_bf16.h"

namespace gcmi {

constexpr int kHMaxSeg = 16;

struct FwdHTable {
  int32_t n_seg;
  int32_t seg_begin[kHMaxSeg];
  int32_t seg_end[kHMaxSeg];
  int32_t tile_start[kHMaxSeg + 1];
  int64_t w_off[2][kHMaxSeg];  // weight block of operand o; < 0: term absent
  int64_t b_off[kHMaxSeg];     // bias row; < 0: none
};

struct FwdHArgs {
  const bf16_t* in[2];
  int32_t ldin[2];       // elements, multiples of 8; columns [k_in, KO) of the rows are zero (or absent: ld < KO)
  int32_t k_in;          // columns of every operand (<= KO)
  const float* w[2];
  const float* bias;
  bf16_t* out;
  int32_t ldo;           // elements, multiple of 8
  int32_t relu;
  double* stats;         // bn.hip scratch layout, or nullptr
  const u32x4* wimg;     // the segments' weight fragments, split and in lane order (wprep_kernel below)
#ifdef GCMI_FWD_H_DIAG_BUILD  // diagnostic build only (tools/fwd_h_diag.sh): phase switches and phase clocks
  int32_t diag_flags;    // 1 no products, 2 no global stores, 4 no operand loads, 8 no epilogue
  unsigned long long* diag_out;
#endif
};

// ---- weight fragments, prepared once per launch.  The kernels keep a wave's weight fragments in registers and reload
// them when its tile range crosses into another degree's segment.  Read from the fp32 parameter block that is eight
// strided dword loads per fragment and lane, which the compiler (short of registers) issues one at a time, each behind
// a full wait: 64 dependent L2 round trips per segment change, 20-40 us that EVERY workgroup pays at its start.  This
// kernel splits every fragment of every segment once -- [segment][32-column tile][k-step][piece][lane] 16-byte entries
// -- so that a reload is NKS * NPW coalesced 1 KiB wave loads with one wait behind them.
template <bool TRANS>
__global__ void __launch_bounds__(256)
wprep_kernel(FwdHTable st, const float* __restrict__ w0, const float* __restrict__ w1, int k_in, int KO, int NOPS,
             int NOUT, u32x4* __restrict__ wimg) {
  const int NKS = NOPS * KO / 16, TW = NOUT / 32;
  const int total = st.n_seg * TW * NKS * 64;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int lane = e & 63;
    int t = e >> 6;
    const int ks = t % NKS; t /= NKS;
    const int tw = t % TW;
    const int seg = t / TW;
    const int n = tw * 32 + (lane & 31);
    const int c0 = ks * 16 + 8 * (lane >> 5);
    const int o = c0 >= KO ? 1 : 0;
    const int ck0 = c0 - o * KO;
    const int64_t woff = o == 1 ? pick_n(st.w_off[1], seg) : pick_n(st.w_off[0], seg);
    const float* w = o == 1 ? w1 : w0;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int ck = ck0 + q;
      const bool ok = woff >= 0 && ck < k_in && w != nullptr;
      v[q] = ok ? (TRANS ? w[woff + (int64_t)n * k_in + ck] : w[woff + (int64_t)ck * NOUT + n]) : 0.f;
    }
    const Frag3 f = split_frag(v);
    u32x4* dst = wimg + (size_t)((seg * TW + tw) * NKS + ks) * 3 * 64 + lane;
    dst[0] = f.p[0];
    dst[64] = f.p[1];
    dst[128] = f.p[2];
  }
}

// NOPS operands of KO (padded) columns each, NOUT output columns, NPW bf16 pieces kept of every weight (3: exact to
// 2^-24; 2: to 2^-16, far below the 2^-9 of the stored result); TRANS: weights stored NOUT x k_in (nn.Linear)
template <int NOPS, int KO, int NOUT, bool TRANS, int NPW>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NPW == 2 ? 3 : 2)))
fwd_h_kernel(FwdHTable st, int n_tiles, FwdHArgs a, int rev) {
  constexpr int NT = 256, ROWS = 64;
  constexpr int NC = NOPS * KO;               // contraction length
  constexpr int AP = NC + 8;                  // pitch of an operand row in LDS (bf16 elements): conflict-free b128 reads
  constexpr int NKS = NC / 16;
  constexpr int TW = NOUT / 32;               // 32-column tiles of the output
  constexpr int TPW = TW / 2;                 // ... per wave: waves = 2 row blocks x 2 column groups
  static_assert(TW % 2 == 0 && KO % 8 == 0 && NC % 16 == 0, "tile shapes");
  constexpr int IQ = KO / 8;                  // 16-byte pieces of an operand row
  constexpr int RQ = NOPS * IQ;               // ... of a tile row over all operands
  constexpr int IPASS = ROWS * RQ / NT;
  static_assert(ROWS * RQ % NT == 0, "tile loads divide evenly");
  constexpr int OPB = NOUT * 2 + 16;          // pitch of an output row in LDS (bytes)
  constexpr int OQ = NOUT / 8;                // 16-byte pieces of an output row
  constexpr int OPASS = ROWS * OQ / NT;
  static_assert(ROWS * OQ % NT == 0 && NT % OQ == 0, "a thread keeps its column piece over the passes");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  bf16_t* As = reinterpret_cast<bf16_t*>(lds_raw);                       // [ROWS][AP]
  unsigned char* Outs = lds_raw + (size_t)ROWS * AP * 2;                  // [ROWS][OPB]
  __shared__ int t_begin_s[kHMaxSeg], t_end_s[kHMaxSeg], t_tile_s[kHMaxSeg + 1];
  __shared__ long long t_w_s[2][kHMaxSeg], t_b_s[kHMaxSeg];
  __shared__ __attribute__((aligned(16))) float bias_s[NOUT];
  __shared__ double stat_s[2][NOUT];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int rb = wave & 1, twb = wave >> 1;   // rows rb*32.., column tiles twb, twb + 2, ...

  if (tid <= kHMaxSeg) {
    t_tile_s[tid] = pick_n(st.tile_start, tid);
    if (tid < kHMaxSeg) {
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
  auto tile_info = [&](int tile, int& seg, int& row0, int& valid) {
    int s = 0;
    for (int k = 1; k < n_seg; ++k) s += tile >= t_tile_s[k] ? 1 : 0;
    seg = __builtin_amdgcn_readfirstlane(s);  // LDS reads land in vector registers; these are uniform
    row0 = __builtin_amdgcn_readfirstlane(t_begin_s[seg] + (tile - t_tile_s[seg]) * ROWS);
    const int left = __builtin_amdgcn_readfirstlane(t_end_s[seg]) - row0;
    valid = left < ROWS ? left : ROWS;
  };

  // ---- prefetch registers: the next tile's operand rows, 16 bytes (8 elements) per lane, unconditional from clamped
  // addresses; a tile row is the RQ pieces of its operands side by side, as it will lie in LDS
  // (ext_vector_type, not HIP's uint4 struct: an array of those captured by the lambdas below goes to scratch memory)
  u32x4 pin[IPASS];
  auto slot_rj = [&](int p, int& r, int& j) {
    int slot = tid + p * NT;
    asm volatile("" : "+v"(slot));  // formed at each use: hoisted out of the tile loop these would be spilled
    r = slot / RQ;
    j = slot - r * RQ;
  };
  auto load_src = [&](int row0, int valid) {
#pragma unroll
    for (int p = 0; p < IPASS; ++p) {
      int r, j;
      slot_rj(p, r, j);
      const int o = NOPS == 2 ? (j >= IQ ? 1 : 0) : 0;
      const int q = j - o * IQ;
      const int ld = o == 1 ? a.ldin[1] : a.ldin[0];
      const int qc = 8 * q + 8 <= ld ? 8 * q : 0;  // a piece beyond the stored row (ld < KO): any finite bytes, its weights are zero
      const int rc = r < valid ? r : valid - 1;
      unsigned off = ((unsigned)(row0 + rc) * (unsigned)ld + (unsigned)qc) * 2u;
      asm volatile("" : "+v"(off));
      const char* base = reinterpret_cast<const char*>(o == 1 ? a.in[1] : a.in[0]);
      pin[p] = *reinterpret_cast<const u32x4*>(base + off);
    }
  };
  // the prefetched rows -> LDS as they are.  Rows beyond a ragged tile's end hold copies of its last row (their
  // results are never stored); columns [k_in, KO) meet zero weights and hold zeros or finite padding.
  auto write_as = [&]() {
#pragma unroll
    for (int p = 0; p < IPASS; ++p) {
      int r, j;
      slot_rj(p, r, j);
      *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(As) + (r * AP + j * 8) * 2) = pin[p];
    }
  };

  // ---- this wave's weight fragments of the segment (the matrix core's A operand: lane = output column, eight
  // consecutive contraction indices), split into NPW bf16 pieces, and the segment's bias row in LDS
  u32x4 wf[TPW][NKS][NPW];
  auto load_w = [&](int seg_) {
    // (wprep_kernel's images: [segment][32-column tile][k-step][piece][lane])
    const u32x4* base = a.wimg + (size_t)seg_ * TW * NKS * 3 * 64 + lane;
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int pc = 0; pc < NPW; ++pc)
          wf[j][ks][pc] = base[(size_t)(((twb + 2 * j) * NKS + ks) * 3 + pc) * 64];
  };

  // ---- the previous tile's output: LDS -> HBM as whole rows, and the BatchNorm sums of the ROUNDED values on the way
  // (fp32 partials per thread over eight tiles -- a thread keeps its column piece --, then fp64 in LDS)
  float ps1[8], ps2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ps1[i] = ps2[i] = 0.f;
  auto flush_stats = [&]() {
    const int q = tid % OQ;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      atomicAdd(&stat_s[0][8 * q + i], (double)ps1[i]);
      atomicAdd(&stat_s[1][8 * q + i], (double)ps2[i]);
      ps1[i] = ps2[i] = 0.f;
    }
  };
  auto store_out = [&](int prow0, int pvalid) {
#pragma unroll
    for (int p = 0; p < OPASS; ++p) {
      const int slot = tid + p * NT;
      const int r = slot / OQ, q = slot - r * OQ;
      if (r < pvalid) {
        const uint4 v = *reinterpret_cast<const uint4*>(Outs + r * OPB + q * 16);
        unsigned off = ((unsigned)(prow0 + r) * (unsigned)a.ldo + 8u * q) * 2u;
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(a.out) + off) = v;
        if (a.stats != nullptr) {
          float f[8];
          widen8(v, f);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            ps1[i] += f[i];
            ps2[i] = fmaf(f[i], f[i], ps2[i]);
          }
        }
      }
    }
  };

  // ---- this wave's output tile(s): products, bias, ReLU, rounding -> the LDS output tile
  auto products = [&]() {
    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[j][k] = 0.f;
    const unsigned char* arow = reinterpret_cast<const unsigned char*>(As) + ((rb * 32 + l31) * AP + 8 * half) * 2;
    u32x4 xa = *reinterpret_cast<const u32x4*>(arow);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const u32x4 x = xa;
      if (ks + 1 < NKS) xa = *reinterpret_cast<const u32x4*>(arow + (ks + 1) * 32);  // issued before this k-step's MFMAs
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        // output columns x rows: lane = row of the tile, registers = output columns; small terms first
#pragma unroll
        for (int pc = NPW - 1; pc >= 0; --pc)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wf[j][ks][pc]), as_bf16x8(x), acc[j], 0, 0, 0);
      }
    }
    unsigned char* orow = Outs + (rb * 32 + l31) * OPB;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = (twb + 2 * j) * 32 + 8 * g + 4 * half;
        const float4 bv = *reinterpret_cast<const float4*>(bias_s + c0);
        float v0 = acc[j][4 * g + 0] + bv.x, v1 = acc[j][4 * g + 1] + bv.y;
        float v2 = acc[j][4 * g + 2] + bv.z, v3 = acc[j][4 * g + 3] + bv.w;
        if (a.relu) {
          v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f;
          v2 = v2 > 0.f ? v2 : 0.f; v3 = v3 > 0.f ? v3 : 0.f;
        }
        *reinterpret_cast<uint2*>(orow + c0 * 2) = narrow4(v0, v1, v2, v3);
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
    // the prefetched rows first (their loads are the oldest entries of the memory queue), then the previous tile's
    // stores: no store sits between a load and the wait for it
    write_as();
    if (i > 0) {
      store_out(prow0, pvalid);
      if (a.stats != nullptr && (i & 7) == 0) flush_stats();
    }
    if (seg != cur_seg) {  // uniform; everyone passed the barrier that ended the previous tile
      const int64_t boff = t_b_s[seg];
      for (int n = tid; n < NOUT; n += NT) bias_s[n] = (a.bias != nullptr && boff >= 0) ? a.bias[boff + n] : 0.f;
    }
    __syncthreads();
    int n2seg = nseg, n2row0 = nrow0, n2valid = nvalid;
    if (i + 2 < my_tiles) tile_info(tile_at(i + 2), n2seg, n2row0, n2valid);
    load_src(nrow0, nvalid);
    if (seg != cur_seg) {
      cur_seg = seg;
      load_w(seg);
    }
    products();
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

// ------------------------------------------------------------------------------------------------------------------
// The same product with the operand rows brought in by LDS-DMA, two tiles ahead.
// fwd_h_kernel keeps ONE tile of prefetch per workgroup in registers: with two four-wave workgroups per CU an
// iteration is load issue -> ~0.5 us of products -> wait for loads that were issued half a microsecond ago, i.e. it runs
// at memory latency (~4 us per tile and workgroup, measured 150-190 us per launch = 3 TB/s).  Here the rows go
// HBM -> LDS directly (global_load_lds_dwordx4: no registers, per-lane source address, lane-linear LDS image) into a
// ring of D + 1 tile buffers, D tiles ahead, and a tile is waited for with a COUNTED s_waitcnt: the memory queue is in
// order, so "all but the youngest D * (stores of one tile) + (D - 1) * (loads of one tile)" is exactly "tile i has
// landed".  For that count to hold every tile issues the same number of loads and stores: a tile beyond the
// workgroup's last one is loaded again into a buffer nobody reads, rows beyond a ragged tile's end are stored to a
// per-thread dump slot.
// The LDS image has no padding (the DMA writes 64 consecutive 16-byte chunks per wave instruction); bank conflicts of
// the fragment reads (ds_read_b128: 16 lanes per cycle, one 16-byte chunk = one "bank quad" each) are avoided by the
// SOURCE side instead: chunk position jpos of row r holds piece j = jpos ^ f(r) of that row, f chosen per row length so
// that the 16 rows of a read group hit 16 different bank quads (RQ 16: r & 15; RQ 20: (r >> 2) & 3, which stays inside
// an aligned group of four pieces; RQ 8: (r >> 1) & 7).
typedef __attribute__((address_space(3))) void* lds_ptr_h;
__device__ __forceinline__ void glds16_h(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);  // wave-uniform by construction
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ u32x4 g_fwdh_dump[512 * 256];  // where the rows beyond a ragged tile's end are stored: one slot per thread

template <int NOPS, int KO, int NOUT, bool TRANS, int NPW>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
fwd_hd_kernel(FwdHTable st, int n_tiles, FwdHArgs a, int rev) {
  constexpr int NT = 256, ROWS = 64, D = 2, NBUF = D + 1;
  constexpr int NC = NOPS * KO;
  constexpr int NKS = NC / 16;
  constexpr int TW = NOUT / 32, TPW = TW / 2;
  static_assert(TW % 2 == 0 && KO % 8 == 0 && NC % 16 == 0, "tile shapes");
  constexpr int IQ = KO / 8, RQ = NOPS * IQ;            // 16-byte pieces of an operand row / of a tile row
  static_assert(RQ == 8 || RQ == 16 || RQ == 20, "a swizzle exists for these row lengths");
  constexpr int TILE_CHUNKS = ROWS * RQ;
  constexpr int TILE_BYTES = TILE_CHUNKS * 16;
  constexpr int LPW = TILE_CHUNKS / NT;                  // DMA instructions per wave and tile
  static_assert(TILE_CHUNKS % NT == 0, "tile loads divide evenly");
  constexpr int OPB = NOUT * 2 + 16, OQ = NOUT / 8, OPASS = ROWS * OQ / NT;
  static_assert(ROWS * OQ % NT == 0 && NT % OQ == 0, "a thread keeps its column piece over the passes");
  constexpr int kWaitSteady = D * OPASS + (D - 1) * LPW;  // younger than tile i's loads when iteration i starts
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char* As = lds_raw;                                         // [NBUF][ROWS][RQ] chunks
  unsigned char* Outs = lds_raw + (size_t)NBUF * TILE_BYTES;           // [ROWS][OPB]
  __shared__ int t_begin_s[kHMaxSeg], t_end_s[kHMaxSeg], t_tile_s[kHMaxSeg + 1];
  __shared__ long long t_w_s[2][kHMaxSeg], t_b_s[kHMaxSeg];
  __shared__ __attribute__((aligned(16))) float bias_s[NOUT];
  __shared__ double stat_s[2][NOUT];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int rb = wave & 1, twb = wave >> 1;
  auto swz = [](int r) { return RQ == 16 ? (r & 15) : RQ == 20 ? ((r >> 2) & 3) : ((r >> 1) & 7); };

  if (tid <= kHMaxSeg) {
    t_tile_s[tid] = pick_n(st.tile_start, tid);
    if (tid < kHMaxSeg) {
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
  const int t_begin = __builtin_amdgcn_readfirstlane((int)((int64_t)b * n_tiles / gridDim.x));
  const int t_end = __builtin_amdgcn_readfirstlane((int)((int64_t)(b + 1) * n_tiles / gridDim.x));
  const int my_tiles = t_end - t_begin;  // >= 1
  auto tile_at = [&](int i) { return rev ? t_end - 1 - i : t_begin + i; };
  // Tile -> (segment, first row, rows), kept INCREMENTALLY: a workgroup walks consecutive tiles, so a cursor moves to the
  // next segment now and then and is otherwise two scalar operations.  (Looking the segment up per tile -- a loop of
  // LDS reads over the segment starts, each landing in a vector register -- measured 1 200 cycles per look-up, two
  // look-ups per tile: a third of the tile loop.)
  struct Cursor { int seg, t0, t1, r0, r1; };
  auto cur_load = [&](Cursor& c) {
    c.t0 = __builtin_amdgcn_readfirstlane(t_tile_s[c.seg]);
    c.t1 = __builtin_amdgcn_readfirstlane(t_tile_s[c.seg + 1]);
    c.r0 = __builtin_amdgcn_readfirstlane(t_begin_s[c.seg]);
    c.r1 = __builtin_amdgcn_readfirstlane(t_end_s[c.seg]);
  };
  auto cur_init = [&](Cursor& c, int tile) {
    int sg = 0;
    for (int k = 1; k < n_seg; ++k) sg += tile >= t_tile_s[k] ? 1 : 0;
    c.seg = __builtin_amdgcn_readfirstlane(sg);
    cur_load(c);
  };
  auto cur_seek = [&](Cursor& c, int tile, int& row0, int& valid) {
    while (tile >= c.t1) { ++c.seg; cur_load(c); }  // (uniform; empty segments are stepped over)
    while (tile < c.t0) { --c.seg; cur_load(c); }
    row0 = c.r0 + (tile - c.t0) * ROWS;
    const int left = c.r1 - row0;
    valid = left < ROWS ? left : ROWS;
  };
  Cursor cur_issue, cur_run;
  cur_init(cur_issue, tile_at(0));
  cur_run = cur_issue;
  const unsigned as_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_h)As);

  // ---- LDS-DMA of tile `it` (clamped to the workgroup's last tile) into ring buffer it % NBUF: LPW instructions per
  // wave, always.  Rows beyond a ragged tile's end re-read its last row (finite values, never stored).
  // what does not depend on the tile is formed once: this lane's row of the tile, its byte column and its operand for
  // each of the LPW instructions of a tile
  int d_row[LPW];
  unsigned d_col[LPW], d_ldb[LPW];
  const char* d_base[LPW];
#pragma unroll
  for (int p = 0; p < LPW; ++p) {
    const int c = (p * 4 + wave) * 64 + lane;  // instruction t = 4 p + wave covers chunks [64 t, 64 t + 64)
    const int r = c / RQ, jpos = c - r * RQ;
    const int j = jpos ^ swz(r);
    const int o = NOPS == 2 ? (j >= IQ ? 1 : 0) : 0;
    const int q = j - o * IQ;
    const int ld = o == 1 ? a.ldin[1] : a.ldin[0];
    d_row[p] = r;
    d_col[p] = (unsigned)(8 * q + 8 <= ld ? 8 * q : 0) * 2u;
    d_ldb[p] = (unsigned)ld * 2u;
    d_base[p] = reinterpret_cast<const char*>(o == 1 ? a.in[1] : a.in[0]);
  }
  auto issue_tile = [&](int it) {
    int row0_, valid_;
    cur_seek(cur_issue, tile_at(it < my_tiles ? it : my_tiles - 1), row0_, valid_);
    const unsigned buf = as_base + (unsigned)(it % NBUF) * TILE_BYTES + (unsigned)wave * 1024u;
#pragma unroll
    for (int p = 0; p < LPW; ++p) {
      const int rc = d_row[p] < valid_ ? d_row[p] : valid_ - 1;
      const unsigned off = (unsigned)(row0_ + rc) * d_ldb[p] + d_col[p];
      glds16_h(d_base[p] + off, buf + (unsigned)p * 4096u);
    }
  };

  u32x4 wf[TPW][NKS][NPW];
  auto load_w = [&](int seg_) {
    // (wprep_kernel's images: [segment][32-column tile][k-step][piece][lane])
    const u32x4* base = a.wimg + (size_t)seg_ * TW * NKS * 3 * 64 + lane;
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int pc = 0; pc < NPW; ++pc)
          wf[j][ks][pc] = base[(size_t)(((twb + 2 * j) * NKS + ks) * 3 + pc) * 64];
  };

  float ps1[8], ps2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ps1[i] = ps2[i] = 0.f;
  auto flush_stats = [&]() {
    const int q = tid % OQ;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      atomicAdd(&stat_s[0][8 * q + i], (double)ps1[i]);
      atomicAdd(&stat_s[1][8 * q + i], (double)ps2[i]);
      ps1[i] = ps2[i] = 0.f;
    }
  };
  u32x4* const my_dump = &g_fwdh_dump[(blockIdx.x % 512) * 256 + tid];
  // OPASS stores per thread, always (rows beyond the tile's end go to the dump slot): the counted wait above relies on it
  auto store_out = [&](int row0_, int valid_) {
#pragma unroll
    for (int p = 0; p < OPASS; ++p) {
      const int slot = tid + p * NT;
      const int r = slot / OQ, q = slot - r * OQ;
      const bool live = r < valid_;
      const u32x4 v = *reinterpret_cast<const u32x4*>(Outs + r * OPB + q * 16);
      const unsigned off = ((unsigned)(row0_ + r) * (unsigned)a.ldo + 8u * q) * 2u;
      u32x4* dst = live ? reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.out) + off) : my_dump;
      *dst = v;
      if (a.stats != nullptr && live) {
        float f[8];
        widen8(uint4{v.x, v.y, v.z, v.w}, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          ps1[i] += f[i];
          ps2[i] = fmaf(f[i], f[i], ps2[i]);
        }
      }
    }
  };

  const int my_row = rb * 32 + l31;
  const int my_swz = swz(my_row);
  auto products = [&](int it) {
    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[j][k] = 0.f;
    const unsigned char* arow = As + (size_t)(it % NBUF) * TILE_BYTES + (size_t)my_row * RQ * 16;
    auto frag = [&](int ks) { return *reinterpret_cast<const u32x4*>(arow + (((2 * ks + half) ^ my_swz) * 16)); };
    u32x4 xa = frag(0);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const u32x4 x = xa;
      if (ks + 1 < NKS) xa = frag(ks + 1);
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
#pragma unroll
        for (int pc = NPW - 1; pc >= 0; --pc)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wf[j][ks][pc]), as_bf16x8(x), acc[j], 0, 0, 0);
      }
    }
    unsigned char* orow = Outs + my_row * OPB;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = (twb + 2 * j) * 32 + 8 * g + 4 * half;
        const float4 bv = *reinterpret_cast<const float4*>(bias_s + c0);
        float v0 = acc[j][4 * g + 0] + bv.x, v1 = acc[j][4 * g + 1] + bv.y;
        float v2 = acc[j][4 * g + 2] + bv.z, v3 = acc[j][4 * g + 3] + bv.w;
        if (a.relu) {
          v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f;
          v2 = v2 > 0.f ? v2 : 0.f; v3 = v3 > 0.f ? v3 : 0.f;
        }
        *reinterpret_cast<uint2*>(orow + c0 * 2) = narrow4(v0, v1, v2, v3);
      }
    }
  };

#ifdef GCMI_FWD_H_DIAG_BUILD
  const int dflags = a.diag_flags;
  const bool stamp = a.diag_out != nullptr && blockIdx.x == 0 && tid == 0;
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
#define DIAG_T(k) do { if (stamp) { t1 = __builtin_amdgcn_s_memtime(); tk[k] += t1 - t0; t0 = t1; } } while (0)
#define DIAG_ON(bit) (!(dflags & (bit)))
#else
#define DIAG_T(k) do { } while (0)
#define DIAG_ON(bit) true
#endif
#pragma unroll
  for (int d = 0; d < D; ++d) issue_tile(d);
  int cur_seg = -1;
#ifdef GCMI_FWD_H_DIAG_BUILD
  if (stamp) t0 = __builtin_amdgcn_s_memtime();
#endif
  for (int i = 0; i < my_tiles; ++i) {
    int row0, valid;
    cur_seek(cur_run, tile_at(i), row0, valid);
    const int seg = cur_run.seg;
    DIAG_T(0);
    // tile i has landed: in the steady state all but the D * OPASS stores and (D - 1) * LPW loads issued after its loads;
    // the first D iterations (no stores of earlier tiles in the queue yet) simply wait for everything
    if (i < D) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWaitSteady) : "memory");
    DIAG_T(1);
    if (seg != cur_seg) {  // uniform; everyone passed the barrier that ended the previous tile's products
      const int64_t boff = t_b_s[seg];
      for (int n = tid; n < NOUT; n += NT) bias_s[n] = (a.bias != nullptr && boff >= 0) ? a.bias[boff + n] : 0.f;
    }
    __syncthreads();  // every wave's part of tile i is in LDS; buffer (i + D) % NBUF (tile i - 1) and Outs are free
    DIAG_T(2);
    if (DIAG_ON(4)) issue_tile(i + D);
    DIAG_T(3);
    if (seg != cur_seg) {
      cur_seg = seg;
      load_w(seg);
    }
    if (DIAG_ON(1)) products(i);
    DIAG_T(4);
    __syncthreads();
    DIAG_T(5);
    if (DIAG_ON(2)) store_out(row0, valid);
    if (a.stats != nullptr && (i & 31) == 31) flush_stats();  // fp32 partials over <= 32 tiles x OPASS rows, then fp64
    DIAG_T(6);
  }
#ifdef GCMI_FWD_H_DIAG_BUILD
  if (stamp) {
    for (int k = 0; k < 7; ++k) a.diag_out[k] = tk[k];
    a.diag_out[7] = (unsigned long long)my_tiles;
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the look-ahead loads of tiles beyond the last one
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

template <int NOPS, int KO, int NOUT, bool TRANS, int NPW>
static int launch_fwd_hd(const FwdHTable& st, int n_tiles, const FwdHArgs& a, hipStream_t sm) {
  constexpr int RQ = NOPS * KO / 8;
  const size_t shmem = (size_t)3 * 64 * RQ * 16 + (size_t)64 * (NOUT * 2 + 16);
  auto kern = fwd_hd_kernel<NOPS, KO, NOUT, TRANS, NPW>;
  static bool attr_done = false;  // per instantiation
  static int per_cu = 1;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)shmem) != hipSuccess) {
      (void)hipGetLastError();
      return GCMI_ERR_UNSUPPORTED;
    }
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), 256, shmem) != hipSuccess ||
        occ < 1) {
      (void)hipGetLastError();
      occ = 2;
    }
    per_cu = std::min(occ, 2);  // (g_fwdh_dump is sized for 512 workgroups)
    if (const char* e = getenv("GCMI_FWD_H_PER_CU")) per_cu = std::max(1, std::min(atoi(e), 2));
    attr_done = true;
  }
  const int grid = std::min(n_tiles, 256 * per_cu);
#ifdef GCMI_FWD_H_DIAG_BUILD
  static unsigned long long* d_diag = nullptr;
  static const int dflags = getenv("GCMI_FWD_H_DIAG") ? atoi(getenv("GCMI_FWD_H_DIAG")) : 0;
  FwdHArgs aa = a;
  if (!d_diag && hipMalloc(&d_diag, 8 * sizeof(unsigned long long)) != hipSuccess) d_diag = nullptr;
  aa.diag_flags = dflags;
  aa.diag_out = d_diag;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, sm, st, n_tiles, aa, next_sweep_direction());
  GCMI_CHECK_LAUNCH("fwd_hd");
  static int printed = 0;
  if (d_diag && printed < 12) {  // serialises the stream: diagnostics only
    unsigned long long h[8];
    if (hipStreamSynchronize(sm) == hipSuccess && hipMemcpy(h, d_diag, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h[7])
      fprintf(stderr, "fwd_hd<%d,%d,%d> flags %d tiles %llu | cycles per tile: tile_info %.0f wait %.0f barrierA %.0f issue %.0f "
                      "products %.0f barrierB %.0f store %.0f\n", NOPS, KO, NOUT, dflags, h[7], (double)h[0] / h[7],
              (double)h[1] / h[7], (double)h[2] / h[7], (double)h[3] / h[7], (double)h[4] / h[7], (double)h[5] / h[7],
              (double)h[6] / h[7]);
    ++printed;
  }
  return GCMI_OK;
#else
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, sm, st, n_tiles, a, next_sweep_direction());
  GCMI_CHECK_LAUNCH("fwd_hd");
  return GCMI_OK;
#endif
}

// The same images for a caller outside this file (fwd_fused.hip's fwd_reg_kernel): [segment][32-column tile][k-step]
// [piece][lane] 16-byte entries, (n_seg * (n_out / 32) * (n_ops * ko / 16) * 3 * 64) of them, in d_scratch.
int fwd_weight_images(int32_t n_seg, const int64_t* w1_off, const int64_t* w2_off, const float* d_w1, const float* d_w2,
                      int32_t k_in, int32_t ko, int32_t n_ops, int32_t n_out, int32_t trans_w, float* d_scratch,
                      hipStream_t sm) {
  if (n_seg > kHMaxSeg || d_scratch == nullptr || !aligned16(d_scratch)) return GCMI_ERR_UNSUPPORTED;
  if ((int64_t)n_seg * (n_out / 32) * (n_ops * ko / 16) * 3 * 256 > kFwdHWimgFloats) return GCMI_ERR_UNSUPPORTED;
  FwdHTable st;
  memset(&st, 0, sizeof(st));
  st.n_seg = n_seg;
  for (int s = 0; s < kHMaxSeg; ++s) {
    st.w_off[0][s] = (s < n_seg && w1_off) ? w1_off[s] : -1;
    st.w_off[1][s] = (s < n_seg && w2_off && n_ops == 2) ? w2_off[s] : -1;
  }
  const int entries = n_seg * (n_out / 32) * (n_ops * ko / 16) * 64;
  const int blocks = std::min((entries + 255) / 256, 1024);
  u32x4* wimg = reinterpret_cast<u32x4*>(d_scratch);
  if (trans_w)
    hipLaunchKernelGGL(wprep_kernel<true>, dim3(blocks), dim3(256), 0, sm, st, d_w1, d_w2, k_in, ko, n_ops, n_out, wimg);
  else
    hipLaunchKernelGGL(wprep_kernel<false>, dim3(blocks), dim3(256), 0, sm, st, d_w1, d_w2, k_in, ko, n_ops, n_out, wimg);
  GCMI_CHECK_LAUNCH("fwd weight images");
  return GCMI_OK;
}

template <int NOPS, int KO, int NOUT, bool TRANS, int NPW>
static int launch_fwd_h(const FwdHTable& st, int n_tiles, const FwdHArgs& a, hipStream_t sm) {
  constexpr int NC = NOPS * KO;
  const size_t shmem = (size_t)64 * (NC + 8) * 2 + (size_t)64 * (NOUT * 2 + 16);
  auto kern = fwd_h_kernel<NOPS, KO, NOUT, TRANS, NPW>;
  static bool attr_done = false;  // per instantiation
  static int per_cu = 1;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)shmem) != hipSuccess) {
      (void)hipGetLastError();
      return GCMI_ERR_UNSUPPORTED;
    }
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), 256, shmem) != hipSuccess ||
        occ < 1) {
      (void)hipGetLastError();
      occ = 2;
    }
    per_cu = std::min(occ, 4);
    if (const char* e = getenv("GCMI_FWD_H_PER_CU")) per_cu = std::max(1, std::min(atoi(e), 8));
    attr_done = true;
  }
  const int grid = std::min(n_tiles, 256 * per_cu);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, sm, st, n_tiles, a, next_sweep_direction());
  GCMI_CHECK_LAUNCH("fwd_h");
  return GCMI_OK;
}

// The shapes of the default model: two operands of 65..80 columns -> 64 columns (the first GraphConv), two 64-column
// operands -> 64 columns (GraphConv over pooled rows), one 64-column operand -> 128 columns in nn.Linear layout (the
// atom-level dense layer).  Anything else: GCMI_ERR_UNSUPPORTED (bf16 storage is for these shapes).
int fwd_h_gemm(int32_t n_seg, const int32_t* seg_begin, const int32_t* seg_end, const bf16_t* d_a1, int64_t lda1,
               int32_t k1, const float* d_w1, const int64_t* w1_off, const bf16_t* d_a2, int64_t lda2, int32_t k2,
               const float* d_w2, const int64_t* w2_off, const float* d_bias, const int64_t* bias_off, int32_t n_out,
               int32_t trans_w, int32_t act, bf16_t* d_out, int64_t ldo, double* d_stats, float* d_wimg_scratch,
               hipStream_t sm) {
  if (n_seg > kHMaxSeg || d_wimg_scratch == nullptr || !aligned16(d_wimg_scratch) || (act != 0 && act != 1) || gemm_exact_mode()) return GCMI_ERR_UNSUPPORTED;
  const bool two = d_a1 != nullptr && d_a2 != nullptr;
  const bool conv = two && !trans_w && n_out == 64 && k1 == k2 && k1 > 32 && k1 <= 64;
  const bool conv80 = two && !trans_w && n_out == 64 && k1 == k2 && k1 > 64 && k1 <= 80;
  const bool dense = !two && d_a1 != nullptr && trans_w && n_out == 128 && k1 > 32 && k1 <= 64;
  if (!conv && !dense && !conv80) return GCMI_ERR_UNSUPPORTED;
  if (!aligned16(d_a1) || lda1 % 8 || (two && (!aligned16(d_a2) || lda2 % 8)) || !aligned16(d_out) || ldo % 8 ||
      ldo < n_out || (d_bias && !aligned16(d_bias)))
    return GCMI_ERR_UNSUPPORTED;
  int64_t rows = 0;
  for (int s = 0; s < n_seg; ++s) rows = std::max<int64_t>(rows, seg_end[s]);
  if (rows * std::max(std::max(lda1, two ? lda2 : 0), ldo) >= (int64_t)1 << 30) return GCMI_ERR_UNSUPPORTED;
  FwdHTable st;
  memset(&st, 0, sizeof(st));
  st.n_seg = n_seg;
  int64_t tiles = 0;
  for (int s = 0; s < kHMaxSeg; ++s) {
    st.tile_start[s] = (int32_t)tiles;
    st.w_off[0][s] = st.w_off[1][s] = st.b_off[s] = -1;
    if (s < n_seg) {
      st.seg_begin[s] = seg_begin[s];
      st.seg_end[s] = seg_end[s];
      st.w_off[0][s] = w1_off ? w1_off[s] : -1;
      st.w_off[1][s] = (two && w2_off) ? w2_off[s] : -1;
      st.b_off[s] = (d_bias && bias_off) ? bias_off[s] : -1;
      tiles += (seg_end[s] - seg_begin[s] + 63) / 64;
    }
  }
  st.tile_start[kHMaxSeg] = (int32_t)tiles;
  if (tiles == 0) return GCMI_OK;
  FwdHArgs a;
  memset(&a, 0, sizeof(a));
  a.in[0] = d_a1; a.ldin[0] = (int32_t)lda1; a.in[1] = d_a2; a.ldin[1] = (int32_t)lda2; a.k_in = k1;
  a.w[0] = d_w1; a.w[1] = d_w2; a.bias = d_bias; a.out = d_out; a.ldo = (int32_t)ldo; a.relu = act; a.stats = d_stats;
  {  // the segments' weight fragments, split once (kFwdHWimgFloats of scratch cover every shape above)
    u32x4* wimg = reinterpret_cast<u32x4*>(d_wimg_scratch);
    const int KO = conv80 ? 80 : 64, NOPS = two ? 2 : 1;
    const int entries = n_seg * (n_out / 32) * (NOPS * KO / 16) * 64;
    const int blocks = std::min((entries + 255) / 256, 1024);
    if (trans_w)
      hipLaunchKernelGGL(wprep_kernel<true>, dim3(blocks), dim3(256), 0, sm, st, d_w1, d_w2, k1, KO, NOPS, n_out, wimg);
    else
      hipLaunchKernelGGL(wprep_kernel<false>, dim3(blocks), dim3(256), 0, sm, st, d_w1, d_w2, k1, KO, NOPS, n_out, wimg);
    GCMI_CHECK_LAUNCH("fwd_h wprep");
    a.wimg = wimg;
  }
  static const int npw = getenv("GCMI_FWD_H_PIECES") ? atoi(getenv("GCMI_FWD_H_PIECES")) : 3;
  // operand rows by LDS-DMA two tiles ahead (default) or one tile ahead in registers (GCMI_FWD_H_DMA=0)
  static const int dma = getenv("GCMI_FWD_H_DMA") ? atoi(getenv("GCMI_FWD_H_DMA")) : 1;
  if (dma && npw != 2) {
    if (conv80) return launch_fwd_hd<2, 80, 64, false, 3>(st, (int)tiles, a, sm);
    if (conv) return launch_fwd_hd<2, 64, 64, false, 3>(st, (int)tiles, a, sm);
    return launch_fwd_hd<1, 64, 128, true, 3>(st, (int)tiles, a, sm);
  }
  if (dma) {
    if (conv80) return launch_fwd_hd<2, 80, 64, false, 2>(st, (int)tiles, a, sm);
    if (conv) return launch_fwd_hd<2, 64, 64, false, 2>(st, (int)tiles, a, sm);
    return launch_fwd_hd<1, 64, 128, true, 2>(st, (int)tiles, a, sm);
  }
  if (npw == 2) {
    if (conv80) return launch_fwd_h<2, 80, 64, false, 2>(st, (int)tiles, a, sm);
    if (conv) return launch_fwd_h<2, 64, 64, false, 2>(st, (int)tiles, a, sm);
    return launch_fwd_h<1, 64, 128, true, 2>(st, (int)tiles, a, sm);
  }
  if (conv80) return launch_fwd_h<2, 80, 64, false, 3>(st, (int)tiles, a, sm);
  if (conv) return launch_fwd_h<2, 64, 64, false, 3>(st, (int)tiles, a, sm);
  return launch_fwd_h<1, 64, 128, true, 3>(st, (int)tiles, a, sm);
}

}  // namespace gcmi
