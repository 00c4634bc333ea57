// LDS-window forms of the three gather kernels: GraphConv.sum_neigh
// (models/torch_models/layers.py:6236-6246), GraphPool.forward (:6319-6367) and its backward.
//
// Every neighbour of an atom belongs to the atom's own molecule.  The collation
// (gcmi_collate_plans) groups consecutive molecules into windows of ~win_cap atoms; because each
// degree block of the batch is sorted by molecule, the atoms of a window are <= 11 CONTIGUOUS row
// ranges (one per degree block).  A persistent workgroup walks windows:
//
//   * the rows of window i+1 stream HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, the
//     per-lane SOURCE address does the row-range lookup, the LDS image is slot-major and
//     lane-linear) while window i is being computed from the other LDS buffer;
//   * the neighbour lists of the window (uint16 LDS slots, window-major in HBM) arrive the same
//     way, so the compute phase touches HBM only to store results;
//   * the window descriptors (24 ints) are fetched three windows ahead into an LDS ring.
//
// HBM traffic drops from E*(4F+4) + N*4F (every neighbour row fetched once per edge) to
// N*4F + 2E + N*4F: each row is read ONCE; the (1+E/N)-fold re-reads are served by LDS.  The
// algorithmic figure of SURVEY.md 8d is therefore delivered above the HBM roofline.
#include "common.h"
#include "split_bf16.h"

namespace gcmi {

// 16-byte row stores of the window operations.  -DGCMI_WIN_NT=1 (A/B switch, tools/win_nt_ab.sh): as non-temporal stores
// (streamed past the caches: every output row of a window pass is written once and read by ANOTHER kernel).
typedef float win_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void win_store4(float* p, const float4 v) {
#if defined(GCMI_WIN_NT) && GCMI_WIN_NT
  const win_f4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<win_f4*>(p));
#else
  *reinterpret_cast<float4*>(p) = v;
#endif
}


constexpr int kND = GCMI_MAX_DEG + 1;
constexpr int kLdsPerCU = 160 * 1024;
constexpr int kRingBytes = 320;                 // 3 descriptors of GCMI_WIN_META_INTS ints, 16-byte padded
constexpr int kHeadBytes = kRingBytes + 2048;   // + 512 floats of per-op constants

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// LDS-DMA: 64 lanes x 16 (4) bytes from per-lane global addresses to LDS [dst, dst + 1024 (256)).
// Written as asm on purpose: for the builtin hipcc (ROCm 7.2) waits vmcnt(0) before EVERY later
// ds_read of the same LDS array (it cannot tell the buffer being filled from the one being read),
// which would serialise the fill of window i+1 with the compute of window i.  The kernel waits
// for these loads itself (wait_dma) before the barrier that publishes a buffer.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr_t)p);
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);  // wave-uniform by construction
#if defined(GCMI_WIN_NT) && GCMI_WIN_NT >= 2  // (A/B switch: the row loads non-temporal as well)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
#else
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
#endif
}
__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// The row ranges of one window, read from the LDS ring of window descriptors.
struct WinMeta {
  int rb[kND];      // global row of slot s (degree d) = rb[d] + s
  int sb[kND + 1];  // first slot of degree d; sb[kND] = atoms in the window
  int eoff;         // first entry of the window in win_edges (multiple of 8)
  int ne;           // edge entries of the window
};

__device__ __forceinline__ WinMeta read_meta(const int* ring_slot) {
  WinMeta m;
#pragma unroll
  for (int d = 0; d < kND; ++d) m.rb[d] = ring_slot[d];
  m.sb[0] = 0;
#pragma unroll
  for (int d = 1; d <= kND; ++d) m.sb[d] = ring_slot[kND - 1 + d];
  m.eoff = ring_slot[2 * kND];
  m.ne = ring_slot[2 * kND + 1];
  return m;
}

// slot -> (degree, global row, first entry of its neighbour list inside the window);
// maxd = highest degree present in the batch (uniform)
__device__ __forceinline__ void locate(const WinMeta& m, int maxd, int slot, int& d, int& row, int& eloc) {
  d = 0;
  int rb = m.rb[0], eb = 0, eacc = 0;
#pragma unroll
  for (int k = 1; k < kND; ++k) {
    if (k <= maxd) {
      eacc += (m.sb[k] - m.sb[k - 1]) * (k - 1);  // entries of the degrees below k
      const bool ge = slot >= m.sb[k];
      d = ge ? k : d;
      rb = ge ? m.rb[k] : rb;
      eb = ge ? eacc - m.sb[k] * k : eb;
    }
  }
  row = rb + slot;
  eloc = eb + slot * d;
}

__device__ __forceinline__ int row_of_slot(const WinMeta& m, int maxd, int slot) {
  int rb = m.rb[0];
#pragma unroll
  for (int k = 1; k < kND; ++k)
    if (k <= maxd) rb = slot >= m.sb[k] ? m.rb[k] : rb;
  return rb + slot;
}

// LDS image of one buffer: [tile: alloc*LPR float4][aux: alloc*LPR*4 bytes (optional)][edges]
struct Layout {
  int tile_bytes, aux_bytes, edge_bytes, maxd;
  __host__ __device__ int buf_bytes() const { return tile_bytes + aux_bytes + edge_bytes; }
};

// Issue the LDS-DMA of one window: rows of `x` (and of the byte matrix `aux`, F bytes per row)
// and the window's neighbour entries.  Nothing waits here.
// (x: rows of LPR 16-byte pieces -- 4 floats or 8 bf16 each --, ldx in BYTES: the DMA moves bytes, not elements)
// AUX: the byte matrix `aux` (one byte per element) rides along.  Four elements per piece (fp32 rows): one dword per
// piece, aux32[e].  Eight (bf16 rows, EPP 8): two dwords per piece by two DMA instructions, each of which writes 64
// lanes x 4 bytes side by side -- so per 64 pieces the image is [64 low dwords][64 high dwords] (aux_lo / aux_hi below).
template <int WT, int LPR, bool AUX, int EPP = 4>
__device__ __forceinline__ void stage(char* buf, const Layout& L, const WinMeta& m,
                                      const char* __restrict__ x, int64_t ldx,
                                      const uint8_t* __restrict__ aux,
                                      const uint16_t* __restrict__ edges) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int n16 = m.sb[kND] * LPR;
  const unsigned base = lds_addr(buf);
  for (int e0 = tid - lane; e0 < n16; e0 += WT) {  // e0: wave-uniform first chunk of this piece
    const int e = e0 + lane;
    if (e < n16) {
      const int slot = e / LPR;
      const int c = e - slot * LPR;
      const int row = row_of_slot(m, L.maxd, slot);
      glds16(x + (int64_t)row * ldx + c * 16, base + e0 * 16);
      if constexpr (AUX && EPP == 4) glds4(aux + (int64_t)row * (LPR * 4) + c * 4, base + L.tile_bytes + e0 * 4);
      if constexpr (AUX && EPP == 8) {
        glds4(aux + (int64_t)row * (LPR * 8) + c * 8, base + L.tile_bytes + (e0 >> 6) * 512);
        glds4(aux + (int64_t)row * (LPR * 8) + c * 8 + 4, base + L.tile_bytes + (e0 >> 6) * 512 + 256);
      }
    }
  }
  const int nq = (m.ne + 7) >> 3;  // 16-byte pieces of the neighbour entries
  const uint16_t* src = edges + m.eoff;
  for (int q0 = tid - lane; q0 < nq; q0 += WT) {
    const int q = q0 + lane;
    if (q < nq) glds16(src + (size_t)q * 8, base + L.tile_bytes + L.aux_bytes + q0 * 16);
  }
}

// The rows of a SECOND matrix of the window (same slots, same row pitch in pieces) into a tile of their own: ops whose
// compute phase would otherwise fetch them from HBM -- and wait, one in-order counter, for the next window's DMA with them.
template <int WT, int LPR>
__device__ __forceinline__ void stage_extra(char* dst, const Layout& L, const WinMeta& m, const char* __restrict__ x2,
                                            int64_t ldx2) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int n16 = m.sb[kND] * LPR;
  const unsigned base = lds_addr(dst);
  for (int e0 = tid - lane; e0 < n16; e0 += WT) {
    const int e = e0 + lane;
    if (e < n16) {
      const int slot = e / LPR;
      const int c = e - slot * LPR;
      glds16(x2 + (int64_t)row_of_slot(m, L.maxd, slot) * ldx2 + c * 16, base + e0 * 16);
    }
  }
}

// ---------------------------------------------------------------- per-window compute phases
// Rows this thread also needs from HBM in the compute phase (the old value of an accumulated output, the BatchNorm
// input of the statistics) are requested for up to kPre elements up front, unconditionally from clamped addresses,
// and used after the LDS work of all of them: the wait for them is also a wait for the LDS-DMA of the next window
// (one counter, in order), which the loop would wait for at its top anyway.
constexpr int kPre = 4;

struct NoState {};

// ACC: s += sum of the neighbours' rows (the transposed gather of a backward pass onto the self term)
template <bool ACC>
struct SumOp {
  float* __restrict__ s;
  int64_t lds;
  using State = NoState;
  static constexpr bool kExtraTile = false;
  static constexpr int kEPP = 4;  // elements per 16-byte piece of a tile row
  __device__ __forceinline__ bool skip(int) const { return false; }
  template <int WT>
  __device__ __forceinline__ void init(float*, int, State&) const {}
  template <int WT>
  __device__ __forceinline__ void finish(char*, int, State&) const {}
  template <int WT, int LPR>
  __device__ __forceinline__ void run(const char* buf, const Layout& L, const WinMeta& m, const float*, State&) const {
    const float4* tile = reinterpret_cast<const float4*>(buf);
    const uint16_t* ent = reinterpret_cast<const uint16_t*>(buf + L.tile_bytes + L.aux_bytes);
    const int n16 = m.sb[kND] * LPR;
    for (int e0 = threadIdx.x; e0 < n16; e0 += kPre * WT) {
      float4 old[kPre];
      if constexpr (ACC) {
#pragma unroll
        for (int k = 0; k < kPre; ++k) {
          const int e = e0 + k * WT < n16 ? e0 + k * WT : n16 - 1;
          const int slot = e / LPR;
          const int c = e - slot * LPR;
          old[k] = *reinterpret_cast<const float4*>(s + (int64_t)row_of_slot(m, L.maxd, slot) * lds + c * 4);
        }
      }
#pragma unroll
      for (int k = 0; k < kPre; ++k) {
        const int e = e0 + k * WT;
        if (e >= n16) break;
        const int slot = e / LPR;
        const int c = e - slot * LPR;
        int d, row, eloc;
        locate(m, L.maxd, slot, d, row, eloc);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);  // lone atoms: zero
        for (int j = 0; j < d; ++j) {
          const int sl = ent[eloc + j] & GCMI_WIN_MAX_SLOTS;
          const float4 v = tile[sl * LPR + c];
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if constexpr (ACC) {
          acc.x += old[k].x; acc.y += old[k].y; acc.z += old[k].z; acc.w += old[k].w;
        }
        win_store4(s + (int64_t)row * lds + c * 4, acc);
      }
    }
  }
};

template <bool BN>
struct MaxOp {
  const float* __restrict__ scale;
  const float* __restrict__ shift;
  float* __restrict__ out;
  int64_t ldo;
  uint8_t* __restrict__ arg;
  static constexpr bool kExtraTile = false;
  static constexpr int kEPP = 4;
  // measured at 1.2 M atoms, 64 columns (us per launch): 256 threads 131 / 137, 512: 151 / 156, 1 024: 137 / 137 (the
  // gather-sum is the other way round: 179 / 127 / 149 for 76 columns; the GraphPool backward 166 / 126 / 141)
  static constexpr int kThreads = 256;
  // the folded BatchNorm vectors live in LDS: a global load in the compute phase would make the
  // compiler wait for the LDS-DMA in flight as well
  using State = NoState;
  __device__ __forceinline__ bool skip(int) const { return false; }
  template <int WT>
  __device__ __forceinline__ void init(float* sh_lds, int n_feat, State&) const {
    if (BN)
      for (int i = threadIdx.x; i < n_feat; i += WT) {
        sh_lds[i] = scale[i];
        sh_lds[256 + i] = shift[i];
      }
  }
  template <int WT>
  __device__ __forceinline__ void finish(char*, int, State&) const {}
  template <int WT, int LPR>
  __device__ __forceinline__ void run(const char* buf, const Layout& L, const WinMeta& m,
                                      const float* sh_lds, State&) const {
    const float4* tile = reinterpret_cast<const float4*>(buf);
    const uint16_t* ent = reinterpret_cast<const uint16_t*>(buf + L.tile_bytes + L.aux_bytes);
    const int n16 = m.sb[kND] * LPR;
    for (int e = threadIdx.x; e < n16; e += WT) {
      const int slot = e / LPR;
      const int c = e - slot * LPR;
      int d, row, eloc;
      locate(m, L.maxd, slot, d, row, eloc);
      float4 sc, sh;
      if (BN) {
        sc = *reinterpret_cast<const float4*>(sh_lds + c * 4);
        sh = *reinterpret_cast<const float4*>(sh_lds + 256 + c * 4);
      }
      float4 best = tile[e];  // self first
      if (BN) {
        best.x = fmaf(best.x, sc.x, sh.x); best.y = fmaf(best.y, sc.y, sh.y);
        best.z = fmaf(best.z, sc.z, sh.z); best.w = fmaf(best.w, sc.w, sh.w);
      }
      uchar4 ba = make_uchar4(0, 0, 0, 0);
      for (int j = 0; j < d; ++j) {
        const int sl = ent[eloc + j] & GCMI_WIN_MAX_SLOTS;
        float4 v = tile[sl * LPR + c];
        if (BN) {
          v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
          v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        }
        const unsigned char a = (unsigned char)(j + 1);
        if (v.x > best.x) { best.x = v.x; ba.x = a; }  // strict >: the first maximum wins
        if (v.y > best.y) { best.y = v.y; ba.y = a; }
        if (v.z > best.z) { best.z = v.z; ba.z = a; }
        if (v.w > best.w) { best.w = v.w; ba.w = a; }
      }
      win_store4(out + (int64_t)row * ldo + c * 4, best);
      if (arg) *reinterpret_cast<uchar4*>(arg + (int64_t)row * (LPR * 4) + c * 4) = ba;
    }
  }
};

// dx[k] = dout[k]*[arg[k]==0] + sum_j dout[i_j]*[arg[i_j] == rev_pos(k,j)+1]: tile = dout rows,
// aux = arg rows of the window.
// STATS: dx is the gradient w.r.t. a BatchNorm output; the column sums its backward needs (sum dx, sum dx*xhat with
// xhat = (x - mean)*invstd of the BatchNorm input x, col_sums_kernel MODE 1 of bn.hip: same fp32 xhat, same fp64
// products and sums) are taken here from the values about to be stored, so dx is not read again for them.  A
// thread's column piece is the same for all its elements (WT % LPR == 0): eight fp64 accumulators per thread,
// combined per workgroup at the end and added to the replicated accumulators with one atomic per column.
template <bool STATS>
struct MaxBwdOp {
  float* __restrict__ dx;
  int64_t lddx;
  const float* __restrict__ x;      // STATS: BatchNorm input rows
  int64_t ldx;
  const float* __restrict__ mean;
  const float* __restrict__ invstd;
  double* __restrict__ sums;        // bn.hip scratch layout: [coef 2F][replica][sum F | sum of products F]
  // optional (bn_bwd_pool_impl): dx is needed only where the pooled BatchNorm sums are ill-conditioned
  const float* __restrict__ only_if_gamma;
  const float* __restrict__ only_if_beta;
  static constexpr bool kExtraTile = false;
  static constexpr int kEPP = 4;
  __device__ __forceinline__ bool skip(int n_feat) const {
    return only_if_gamma != nullptr && !bn_pool_ill_conditioned(only_if_gamma, only_if_beta, n_feat);
  }
  struct State {
    double s1[STATS ? 4 : 1], s2[STATS ? 4 : 1];
  };
  template <int WT>
  __device__ __forceinline__ void init(float* sh_lds, int n_feat, State& acc_) const {
    if constexpr (STATS) {
      for (int i = threadIdx.x; i < n_feat; i += WT) {
        sh_lds[i] = mean[i];
        sh_lds[256 + i] = invstd[i];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) acc_.s1[q] = acc_.s2[q] = 0.0;
    }
  }
  template <int WT, int LPR>
  __device__ __forceinline__ void run(const char* buf, const Layout& L, const WinMeta& m, const float* sh_lds,
                                      State& acc_) const {
    const float4* tile = reinterpret_cast<const float4*>(buf);
    const uchar4* atile = reinterpret_cast<const uchar4*>(buf + L.tile_bytes);
    const uint16_t* ent = reinterpret_cast<const uint16_t*>(buf + L.tile_bytes + L.aux_bytes);
    const int n16 = m.sb[kND] * LPR;
    for (int e0 = threadIdx.x; e0 < n16; e0 += kPre * WT) {
      float4 xr[kPre];
      if constexpr (STATS) {
#pragma unroll
        for (int k = 0; k < kPre; ++k) {
          const int e = e0 + k * WT < n16 ? e0 + k * WT : n16 - 1;
          const int slot = e / LPR;
          const int c = e - slot * LPR;
          xr[k] = *reinterpret_cast<const float4*>(x + (int64_t)row_of_slot(m, L.maxd, slot) * ldx + c * 4);
        }
      }
#pragma unroll
      for (int k = 0; k < kPre; ++k) {
        const int e = e0 + k * WT;
        if (e >= n16) break;
        const int slot = e / LPR;
        const int c = e - slot * LPR;
        int d, row, eloc;
        locate(m, L.maxd, slot, d, row, eloc);
        float4 g = tile[e];
        uchar4 a = atile[e];
        float4 acc;
        acc.x = a.x == 0 ? g.x : 0.f;
        acc.y = a.y == 0 ? g.y : 0.f;
        acc.z = a.z == 0 ? g.z : 0.f;
        acc.w = a.w == 0 ? g.w : 0.f;
        for (int j = 0; j < d; ++j) {
          const int en = ent[eloc + j];
          const int sl = en & GCMI_WIN_MAX_SLOTS;
          const unsigned char want = (unsigned char)((en >> GCMI_WIN_SLOT_BITS) + 1);
          g = tile[sl * LPR + c];
          a = atile[sl * LPR + c];
          acc.x += a.x == want ? g.x : 0.f;
          acc.y += a.y == want ? g.y : 0.f;
          acc.z += a.z == want ? g.z : 0.f;
          acc.w += a.w == want ? g.w : 0.f;
        }
        win_store4(dx + (int64_t)row * lddx + c * 4, acc);
        if constexpr (STATS) {
          const float4 mu = *reinterpret_cast<const float4*>(sh_lds + c * 4);
          const float4 is = *reinterpret_cast<const float4*>(sh_lds + 256 + c * 4);
          acc_.s1[0] += (double)acc.x; acc_.s2[0] += (double)acc.x * (double)((xr[k].x - mu.x) * is.x);
          acc_.s1[1] += (double)acc.y; acc_.s2[1] += (double)acc.y * (double)((xr[k].y - mu.y) * is.y);
          acc_.s1[2] += (double)acc.z; acc_.s2[2] += (double)acc.z * (double)((xr[k].z - mu.z) * is.z);
          acc_.s1[3] += (double)acc.w; acc_.s2[3] += (double)acc.w * (double)((xr[k].w - mu.w) * is.w);
        }
      }
    }
  }
  // after the last window (every thread of the workgroup arrives here): wave partials by shuffles over the lanes
  // that share a column piece, the waves' partials meet in the (now idle) window buffers, one fp64 atomic per column
  template <int WT>
  __device__ __forceinline__ void finish(char* smem, int lpr, State& acc_) const {
    if constexpr (STATS) {
      const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
      double v[8] = {acc_.s1[0], acc_.s1[1], acc_.s1[2], acc_.s1[3], acc_.s2[0], acc_.s2[1], acc_.s2[2], acc_.s2[3]};
      for (int o = lpr; o < 64; o <<= 1)  // lpr is 16 or 32: lanes l, l + lpr, ... hold the same piece
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += __shfl_xor(v[i], o, 64);
      __syncthreads();  // the window buffers are no longer read
      double* red = reinterpret_cast<double*>(smem);  // [wave][piece][8]
      if (lane < lpr)
#pragma unroll
        for (int i = 0; i < 8; ++i) red[(wave * lpr + lane) * 8 + i] = v[i];
      __syncthreads();
      const int n_feat = lpr * 4;
      for (int t = tid; t < 2 * n_feat; t += WT) {
        const int which = t / n_feat, col = t - which * n_feat;
        double tot = 0.0;
        for (int w = 0; w < WT / 64; ++w) tot += red[(w * lpr + (col >> 2)) * 8 + which * 4 + (col & 3)];
        atomicAdd(sums + (size_t)2 * n_feat * (1 + (blockIdx.x % kBnReplicas)) + (size_t)which * n_feat + col, tot);
      }
    }
  }
};

// The backward between two GraphConv blocks in one window pass:
//   dX[k]  = dXs[k] + sum_j dS[i_j]                 (SumOp<true>: the neighbour part onto the self part)
//   dy[k]  = dX[k]*[arg[k]==0] + sum_j dX[i_j]*[arg[i_j] == rev_pos(k,j)+1]       (MaxBwdOp of the block below)
// dX is the gradient of the pooled rows and nothing else reads it, so it lives in a third LDS tile only: it is neither
// written (N*F floats) nor read back (N*F) through HBM.  tile = dS rows, aux = arg rows of the block below.  The price
// is LDS: one workgroup per CU instead of two.  Oversized windows are not handled (the launcher refuses).
struct SumAccMaxBwdOp {
  const float* __restrict__ dxs;  // self part of dX (global rows)
  int64_t lddxs;
  float* __restrict__ dy;
  int64_t lddy;
  struct State {
    char* extra;  // the third tile of the window being computed: [slot][LPR] float4, the dXs rows on entry
  };
  static constexpr bool kExtraTile = true;
  // The third tile is double-buffered and filled by the LDS-DMA with the window's dXs rows (stage_extra): read from HBM
  // in the compute phase they waited, one in-order counter, for the NEXT window's DMA, so that no window's compute
  // overlapped the next one's load (281 us at 96-atom windows against 181 at 192-atom ones: a fixed ~3.8 us per window).
  static constexpr bool kExtraDma = true;
  static constexpr int kExtraScale = 2;
  static constexpr int kEPP = 4;
  __device__ __forceinline__ const char* extra_src() const { return reinterpret_cast<const char*>(dxs); }
  __device__ __forceinline__ int64_t extra_ld_bytes() const { return lddxs * 4; }
  __device__ __forceinline__ bool skip(int) const { return false; }
  template <int WT>
  __device__ __forceinline__ void init(float*, int, State&) const {}
  template <int WT>
  __device__ __forceinline__ void finish(char*, int, State&) const {}
  template <int WT, int LPR>
  __device__ __forceinline__ void run(const char* buf, const Layout& L, const WinMeta& m, const float*, State& st) const {
    const float4* tile = reinterpret_cast<const float4*>(buf);
    const uchar4* atile = reinterpret_cast<const uchar4*>(buf + L.tile_bytes);
    const uint16_t* ent = reinterpret_cast<const uint16_t*>(buf + L.tile_bytes + L.aux_bytes);
    float4* t2 = reinterpret_cast<float4*>(st.extra);
    const int n16 = m.sb[kND] * LPR;
    // ---- stage 1: dX of the window = gather(dS) + dXs, in place in the third tile
    for (int e = threadIdx.x; e < n16; e += WT) {
      const int slot = e / LPR;
      const int c = e - slot * LPR;
      int d, row, eloc;
      locate(m, L.maxd, slot, d, row, eloc);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j = 0; j < d; ++j) {
        const int sl = ent[eloc + j] & GCMI_WIN_MAX_SLOTS;
        const float4 v = tile[sl * LPR + c];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      const float4 old = t2[e];
      acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;  // same order as SumOp<true>
      t2[e] = acc;
    }
    __syncthreads();
    // ---- stage 2: the GraphPool backward over it
    for (int e = threadIdx.x; e < n16; e += WT) {
      const int slot = e / LPR;
      const int c = e - slot * LPR;
      int d, row, eloc;
      locate(m, L.maxd, slot, d, row, eloc);
      float4 g = t2[e];
      uchar4 a = atile[e];
      float4 acc;
      acc.x = a.x == 0 ? g.x : 0.f;
      acc.y = a.y == 0 ? g.y : 0.f;
      acc.z = a.z == 0 ? g.z : 0.f;
      acc.w = a.w == 0 ? g.w : 0.f;
      for (int j = 0; j < d; ++j) {
        const int en = ent[eloc + j];
        const int sl = en & GCMI_WIN_MAX_SLOTS;
        const unsigned char want = (unsigned char)((en >> GCMI_WIN_SLOT_BITS) + 1);
        g = t2[sl * LPR + c];
        a = atile[sl * LPR + c];
        acc.x += a.x == want ? g.x : 0.f;
        acc.y += a.y == want ? g.y : 0.f;
        acc.z += a.z == want ? g.z : 0.f;
        acc.w += a.w == want ? g.w : 0.f;
      }
      win_store4(dy + (int64_t)row * lddy + c * 4, acc);
    }
    // (the walker's barrier at the top of the next window comes before the third tile is written again)
  }
};

// ---------------------------------------------------------------- bf16 activation storage (gcmi_model_desc.storage == 1)
// The LDS-DMA moves bytes: a bf16 row of 64 is 8 pieces of 16 bytes, a piece holds 8 elements.  Sums and maxima are
// formed in fp32 from the widened elements and rounded once (v_cvt_pk_bf16_f32) when the row is stored.

// First GraphConv: the atom features arrive as fp32 rows (the caller's matrix); the sum of the neighbours' rows goes
// out as bf16, and so does a bf16 copy of the atom's OWN row (it is in LDS anyway), so that the product that follows
// and the backward read two bf16 operands and the fp32 matrix is read exactly once per step.  Output rows are `ldo`
// elements; the `pad4` groups of four columns behind the LPR pieces are zeroed (76 -> 80 columns: 16-byte rows).
struct SumOpFH {
  bf16_t* __restrict__ s;
  bf16_t* __restrict__ xcopy;
  int64_t ldo;
  int pad4;
  using State = NoState;
  static constexpr bool kExtraTile = false;
  static constexpr int kEPP = 4;
  __device__ __forceinline__ bool skip(int) const { return false; }
  template <int WT>
  __device__ __forceinline__ void init(float*, int, State&) const {}
  template <int WT>
  __device__ __forceinline__ void finish(char*, int, State&) const {}
  template <int WT, int LPR>
  __device__ __forceinline__ void run(const char* buf, const Layout& L, const WinMeta& m, const float*, State&) const {
    const float4* tile = reinterpret_cast<const float4*>(buf);
    const uint16_t* ent = reinterpret_cast<const uint16_t*>(buf + L.tile_bytes + L.aux_bytes);
    const int n16 = m.sb[kND] * LPR;
    for (int e = threadIdx.x; e < n16; e += WT) {
      const int slot = e / LPR;
      const int c = e - slot * LPR;
      int d, row, eloc;
      locate(m, L.maxd, slot, d, row, eloc);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);  // lone atoms: zero
      for (int j = 0; j < d; ++j) {
        const int sl = ent[eloc + j] & GCMI_WIN_MAX_SLOTS;
        const float4 v = tile[sl * LPR + c];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      const float4 own = tile[e];
      bf16_t* srow = s + (int64_t)row * ldo + c * 4;
      bf16_t* xrow = xcopy + (int64_t)row * ldo + c * 4;
      *reinterpret_cast<uint2*>(srow) = narrow4(acc.x, acc.y, acc.z, acc.w);
      *reinterpret_cast<uint2*>(xrow) = narrow4(own.x, own.y, own.z, own.w);
      if (c == LPR - 1) {
        for (int z = 1; z <= pad4; ++z) {
          *reinterpret_cast<uint2*>(srow + 4 * z) = make_uint2(0u, 0u);
          *reinterpret_cast<uint2*>(xrow + 4 * z) = make_uint2(0u, 0u);
        }
      }
    }
  }
};

// GraphConv.sum_neigh over rows stored as bf16 (the pooled rows of the block below): bf16 in, bf16 out
struct SumOpH {
  bf16_t* __restrict__ s;
  int64_t lds;
  using State = NoState;
  static constexpr bool kExtraTile = false;
  static constexpr int kEPP = 8;
  static constexpr int kThreads = 256;  // measured (64 columns, 1.2 M atoms): 58.6 us against 74.4 at 512 and 93.4 at 1 024
  __device__ __forceinline__ bool skip(int) const { return false; }
  template <int WT>
  __device__ __forceinline__ void init(float*, int, State&) const {}
  template <int WT>
  __device__ __forceinline__ void finish(char*, int, State&) const {}
  template <int WT, int LPR>
  __device__ __forceinline__ void run(const char* buf, const Layout& L, const WinMeta& m, const float*, State&) const {
    const uint4* tile = reinterpret_cast<const uint4*>(buf);
    const uint16_t* ent = reinterpret_cast<const uint16_t*>(buf + L.tile_bytes + L.aux_bytes);
    const int n16 = m.sb[kND] * LPR;
    for (int e = threadIdx.x; e < n16; e += WT) {
      const int slot = e / LPR;
      const int c = e - slot * LPR;
      int d, row, eloc;
      locate(m, L.maxd, slot, d, row, eloc);
      float acc[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = 0.f;
      for (int j = 0; j < d; ++j) {
        const int sl = ent[eloc + j] & GCMI_WIN_MAX_SLOTS;
        float v[8];
        widen8(tile[sl * LPR + c], v);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += v[q];
      }
      uint4 o;
      o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]);
      o.z = pack_bf16x2(acc[4], acc[5]); o.w = pack_bf16x2(acc[6], acc[7]);
      *reinterpret_cast<uint4*>(s + (int64_t)row * lds + c * 8) = o;
    }
  }
};

// GraphPool over bf16 rows with the folded BatchNorm applied on the fly: candidates y = x * scale + shift in fp32,
// the first maximum wins (self first, then neighbour order), the winner is rounded once when it is stored
template <bool BN>
struct MaxOpH {
  const float* __restrict__ scale;
  const float* __restrict__ shift;
  bf16_t* __restrict__ out;
  int64_t ldo;
  uint8_t* __restrict__ arg;
  static constexpr bool kExtraTile = false;
  static constexpr int kEPP = 8;
  using State = NoState;
  __device__ __forceinline__ bool skip(int) const { return false; }
  template <int WT>
  __device__ __forceinline__ void init(float* sh_lds, int n_feat, State&) const {
    if (BN)
      for (int i = threadIdx.x; i < n_feat; i += WT) {
        sh_lds[i] = scale[i];
        sh_lds[256 + i] = shift[i];
      }
  }
  template <int WT>
  __device__ __forceinline__ void finish(char*, int, State&) const {}
  template <int WT, int LPR>
  __device__ __forceinline__ void run(const char* buf, const Layout& L, const WinMeta& m,
                                      const float* sh_lds, State&) const {
    const uint4* tile = reinterpret_cast<const uint4*>(buf);
    const uint16_t* ent = reinterpret_cast<const uint16_t*>(buf + L.tile_bytes + L.aux_bytes);
    const int n16 = m.sb[kND] * LPR;
    for (int e = threadIdx.x; e < n16; e += WT) {
      const int slot = e / LPR;
      const int c = e - slot * LPR;
      int d, row, eloc;
      locate(m, L.maxd, slot, d, row, eloc);
      float sc[8], sh[8];
      if (BN) {
        const float4 a0 = *reinterpret_cast<const float4*>(sh_lds + c * 8);
        const float4 a1 = *reinterpret_cast<const float4*>(sh_lds + c * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(sh_lds + 256 + c * 8);
        const float4 b1 = *reinterpret_cast<const float4*>(sh_lds + 256 + c * 8 + 4);
        sc[0] = a0.x; sc[1] = a0.y; sc[2] = a0.z; sc[3] = a0.w; sc[4] = a1.x; sc[5] = a1.y; sc[6] = a1.z; sc[7] = a1.w;
        sh[0] = b0.x; sh[1] = b0.y; sh[2] = b0.z; sh[3] = b0.w; sh[4] = b1.x; sh[5] = b1.y; sh[6] = b1.z; sh[7] = b1.w;
      }
      float best[8];
      widen8(tile[e], best);  // self first
      unsigned char ba[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (BN) best[q] = fmaf(best[q], sc[q], sh[q]);
        ba[q] = 0;
      }
      for (int j = 0; j < d; ++j) {
        const int sl = ent[eloc + j] & GCMI_WIN_MAX_SLOTS;
        float v[8];
        widen8(tile[sl * LPR + c], v);
        const unsigned char a = (unsigned char)(j + 1);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (BN) v[q] = fmaf(v[q], sc[q], sh[q]);
          if (v[q] > best[q]) { best[q] = v[q]; ba[q] = a; }  // strict >: the first maximum wins
        }
      }
      uint4 o;
      o.x = pack_bf16x2(best[0], best[1]); o.y = pack_bf16x2(best[2], best[3]);
      o.z = pack_bf16x2(best[4], best[5]); o.w = pack_bf16x2(best[6], best[7]);
      *reinterpret_cast<uint4*>(out + (int64_t)row * ldo + c * 8) = o;
      if (arg) {
        uint2 av;
        av.x = (unsigned)ba[0] | ((unsigned)ba[1] << 8) | ((unsigned)ba[2] << 16) | ((unsigned)ba[3] << 24);
        av.y = (unsigned)ba[4] | ((unsigned)ba[5] << 8) | ((unsigned)ba[6] << 16) | ((unsigned)ba[7] << 24);
        *reinterpret_cast<uint2*>(arg + (int64_t)row * (LPR * 8) + c * 8) = av;
      }
    }
  }
};

// GraphPool of block l and GraphConv.sum_neigh of block l + 1 in ONE window pass (bf16 rows): the pooled rows of the
// window are formed in a third LDS tile (and stored, with their arg-max bytes, as MaxOpH stores them), then every atom
// sums its neighbours' pooled rows from that tile.  The pooled matrix is written once and NOT read back by a second
// launch (-128 bytes per atom and one launch); the third tile is half the size it would be in fp32, so two workgroups
// still share a CU.  Same values as MaxOpH followed by SumOpH (same candidates in the same order, the same rounded
// pooled values summed in the same neighbour order).  Oversized windows take the two separate ops over their own rows.
template <bool BN>
struct MaxSumOpH {
  const float* __restrict__ scale;
  const float* __restrict__ shift;
  bf16_t* __restrict__ out;   // pooled rows
  int64_t ldo;
  uint8_t* __restrict__ arg;
  bf16_t* __restrict__ s;     // neighbour sums of the pooled rows
  int64_t lds;
  struct State {
    char* extra;  // the third tile: [slot][LPR] 16-byte pieces of pooled rows
  };
  static constexpr bool kExtraTile = true;
  static constexpr int kEPP = 8;
  __device__ __forceinline__ bool skip(int) const { return false; }
  template <int WT>
  __device__ __forceinline__ void init(float* sh_lds, int n_feat, State&) const {
    if (BN)
      for (int i = threadIdx.x; i < n_feat; i += WT) {
        sh_lds[i] = scale[i];
        sh_lds[256 + i] = shift[i];
      }
  }
  template <int WT>
  __device__ __forceinline__ void finish(char*, int, State&) const {}
  template <int WT, int LPR>
  __device__ __forceinline__ void run(const char* buf, const Layout& L, const WinMeta& m, const float* sh_lds,
                                      State& st) const {
    const uint4* tile = reinterpret_cast<const uint4*>(buf);
    const uint16_t* ent = reinterpret_cast<const uint16_t*>(buf + L.tile_bytes + L.aux_bytes);
    uint4* t2 = reinterpret_cast<uint4*>(st.extra);
    const int n16 = m.sb[kND] * LPR;
    // ---- stage 1: the pooled rows of the window -> HBM and the third tile
    for (int e = threadIdx.x; e < n16; e += WT) {
      const int slot = e / LPR;
      const int c = e - slot * LPR;
      int d, row, eloc;
      locate(m, L.maxd, slot, d, row, eloc);
      float sc[8], sh[8];
      if (BN) {
        const float4 a0 = *reinterpret_cast<const float4*>(sh_lds + c * 8);
        const float4 a1 = *reinterpret_cast<const float4*>(sh_lds + c * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(sh_lds + 256 + c * 8);
        const float4 b1 = *reinterpret_cast<const float4*>(sh_lds + 256 + c * 8 + 4);
        sc[0] = a0.x; sc[1] = a0.y; sc[2] = a0.z; sc[3] = a0.w; sc[4] = a1.x; sc[5] = a1.y; sc[6] = a1.z; sc[7] = a1.w;
        sh[0] = b0.x; sh[1] = b0.y; sh[2] = b0.z; sh[3] = b0.w; sh[4] = b1.x; sh[5] = b1.y; sh[6] = b1.z; sh[7] = b1.w;
      }
      float best[8];
      widen8(tile[e], best);  // self first
      unsigned char ba[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (BN) best[q] = fmaf(best[q], sc[q], sh[q]);
        ba[q] = 0;
      }
      for (int j = 0; j < d; ++j) {
        const int sl = ent[eloc + j] & GCMI_WIN_MAX_SLOTS;
        float v[8];
        widen8(tile[sl * LPR + c], v);
        const unsigned char a = (unsigned char)(j + 1);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (BN) v[q] = fmaf(v[q], sc[q], sh[q]);
          if (v[q] > best[q]) { best[q] = v[q]; ba[q] = a; }  // strict >: the first maximum wins
        }
      }
      uint4 o;
      o.x = pack_bf16x2(best[0], best[1]); o.y = pack_bf16x2(best[2], best[3]);
      o.z = pack_bf16x2(best[4], best[5]); o.w = pack_bf16x2(best[6], best[7]);
      *reinterpret_cast<uint4*>(out + (int64_t)row * ldo + c * 8) = o;
      t2[e] = o;
      if (arg) {
        uint2 av;
        av.x = (unsigned)ba[0] | ((unsigned)ba[1] << 8) | ((unsigned)ba[2] << 16) | ((unsigned)ba[3] << 24);
        av.y = (unsigned)ba[4] | ((unsigned)ba[5] << 8) | ((unsigned)ba[6] << 16) | ((unsigned)ba[7] << 24);
        *reinterpret_cast<uint2*>(arg + (int64_t)row * (LPR * 8) + c * 8) = av;
      }
    }
    __syncthreads();
    // ---- stage 2: every atom sums its neighbours' pooled rows
    for (int e = threadIdx.x; e < n16; e += WT) {
      const int slot = e / LPR;
      const int c = e - slot * LPR;
      int d, row, eloc;
      locate(m, L.maxd, slot, d, row, eloc);
      float acc[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = 0.f;
      for (int j = 0; j < d; ++j) {
        const int sl = ent[eloc + j] & GCMI_WIN_MAX_SLOTS;
        float v[8];
        widen8(t2[sl * LPR + c], v);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += v[q];
      }
      uint4 o;
      o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]);
      o.z = pack_bf16x2(acc[4], acc[5]); o.w = pack_bf16x2(acc[6], acc[7]);
      *reinterpret_cast<uint4*>(s + (int64_t)row * lds + c * 8) = o;
    }
    // (the walker's barrier at the top of the next window comes before the third tile is written again)
  }
};

// ---------------------------------------------------------------- gradient streams in bf16 (storage == 2)
// The gradients that travel between kernels (dpool, dy, dS, dXs) as bf16 rows: the same ops as MaxBwdOp, SumOp<true> and
// SumAccMaxBwdOp over pieces of eight elements, sums in fp32, one rounding at the store.  The arg-max bytes of a piece
// are two dwords in the split image stage() writes (aux_lo / aux_hi).
__device__ __forceinline__ uint2 aux8(const char* buf, const Layout& L, int e) {
  const unsigned* a = reinterpret_cast<const unsigned*>(buf + L.tile_bytes) + (e >> 6) * 128 + (e & 63);
  return make_uint2(a[0], a[64]);
}
__device__ __forceinline__ unsigned char aux_byte(const uint2 a, int q) {
  return (unsigned char)((q < 4 ? a.x >> (8 * q) : a.y >> (8 * (q - 4))) & 255u);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  uint4 o;
  o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
  o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
  return o;
}

// dx[k] = dout[k]*[arg[k]==0] + sum_j dout[i_j]*[arg[i_j] == rev_pos(k,j)+1] over bf16 rows
struct MaxBwdOpH {
  bf16_t* __restrict__ dx;
  int64_t lddx;
  const float* __restrict__ only_if_gamma;  // optional (bn_bwd_pool_impl): dx only where the pooled sums are ill-conditioned
  const float* __restrict__ only_if_beta;
  using State = NoState;
  static constexpr bool kExtraTile = false;
  static constexpr int kEPP = 8;
  static constexpr int kThreads = 256;  // measured: 84.4 us against 92.6 at 512 and 119 at 1 024
  __device__ __forceinline__ bool skip(int n_feat) const {
    return only_if_gamma != nullptr && !bn_pool_ill_conditioned(only_if_gamma, only_if_beta, n_feat);
  }
  template <int WT>
  __device__ __forceinline__ void init(float*, int, State&) const {}
  template <int WT>
  __device__ __forceinline__ void finish(char*, int, State&) const {}
  template <int WT, int LPR>
  __device__ __forceinline__ void run(const char* buf, const Layout& L, const WinMeta& m, const float*, State&) const {
    const uint4* tile = reinterpret_cast<const uint4*>(buf);
    const uint16_t* ent = reinterpret_cast<const uint16_t*>(buf + L.tile_bytes + L.aux_bytes);
    const int n16 = m.sb[kND] * LPR;
    for (int e = threadIdx.x; e < n16; e += WT) {
      const int slot = e / LPR;
      const int c = e - slot * LPR;
      int d, row, eloc;
      locate(m, L.maxd, slot, d, row, eloc);
      float g[8], acc[8];
      widen8(tile[e], g);
      uint2 a = aux8(buf, L, e);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = aux_byte(a, q) == 0 ? g[q] : 0.f;
      for (int j = 0; j < d; ++j) {
        const int en = ent[eloc + j];
        const int sl = en & GCMI_WIN_MAX_SLOTS;
        const unsigned char want = (unsigned char)((en >> GCMI_WIN_SLOT_BITS) + 1);
        widen8(tile[sl * LPR + c], g);
        a = aux8(buf, L, sl * LPR + c);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += aux_byte(a, q) == want ? g[q] : 0.f;
      }
      *reinterpret_cast<uint4*>(dx + (int64_t)row * lddx + c * 8) = pack8(acc);
    }
  }
};

// s += sum of the neighbours' rows, bf16 in and out (the oversized windows of the two-stage pass)
struct SumAccOpH {
  bf16_t* __restrict__ s;
  int64_t lds;
  using State = NoState;
  static constexpr bool kExtraTile = false;
  static constexpr int kEPP = 8;
  __device__ __forceinline__ bool skip(int) const { return false; }
  template <int WT>
  __device__ __forceinline__ void init(float*, int, State&) const {}
  template <int WT>
  __device__ __forceinline__ void finish(char*, int, State&) const {}
  template <int WT, int LPR>
  __device__ __forceinline__ void run(const char* buf, const Layout& L, const WinMeta& m, const float*, State&) const {
    const uint4* tile = reinterpret_cast<const uint4*>(buf);
    const uint16_t* ent = reinterpret_cast<const uint16_t*>(buf + L.tile_bytes + L.aux_bytes);
    const int n16 = m.sb[kND] * LPR;
    for (int e = threadIdx.x; e < n16; e += WT) {
      const int slot = e / LPR;
      const int c = e - slot * LPR;
      int d, row, eloc;
      locate(m, L.maxd, slot, d, row, eloc);
      bf16_t* dst = s + (int64_t)row * lds + c * 8;
      const uint4 old = *reinterpret_cast<const uint4*>(dst);
      float acc[8], v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = 0.f;
      for (int j = 0; j < d; ++j) {
        widen8(tile[(ent[eloc + j] & GCMI_WIN_MAX_SLOTS) * LPR + c], v);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += v[q];
      }
      widen8(old, v);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += v[q];  // same order as SumAccMaxBwdOpH's first stage
      *reinterpret_cast<uint4*>(dst) = pack8(acc);
    }
  }
};

// SumAccMaxBwdOp over bf16 streams: tile = dS rows, aux = arg rows of the block below, dxs / dy bf16; dX of the window
// stays fp32 in the third LDS tile (it is never stored, so it is never rounded).  With tiles of half the size two
// 1 024-thread workgroups... do not fit a CU's thread limit, but two 512-thread ones do.
struct SumAccMaxBwdOpH {
  const bf16_t* __restrict__ dxs;  // self part of dX (global rows)
  int64_t lddxs;
  bf16_t* __restrict__ dy;
  int64_t lddy;
  struct State {
    char* extra;  // the third tile of the window being computed: [slot][LPR] pieces of 8 bf16, the dXs rows on entry
  };
  static constexpr bool kExtraTile = true;
  // As SumAccMaxBwdOp: the third tile comes by LDS-DMA with the window, one per buffer.  It is a bf16 tile and dX is
  // completed IN PLACE in it -- rounded to bf16 once, exactly what the two separate passes do when they write dX to
  // HBM as a bf16 matrix between them -- so the LDS footprint stays that of the former single fp32 tile (two 512-thread
  // workgroups per CU) and the compute phase reads nothing from HBM.
  static constexpr bool kExtraDma = true;
  static constexpr int kExtraScale = 2;
  static constexpr int kEPP = 8;
  __device__ __forceinline__ const char* extra_src() const { return reinterpret_cast<const char*>(dxs); }
  __device__ __forceinline__ int64_t extra_ld_bytes() const { return lddxs * 2; }
  __device__ __forceinline__ bool skip(int) const { return false; }
  template <int WT>
  __device__ __forceinline__ void init(float*, int, State&) const {}
  template <int WT>
  __device__ __forceinline__ void finish(char*, int, State&) const {}
  template <int WT, int LPR>
  __device__ __forceinline__ void run(const char* buf, const Layout& L, const WinMeta& m, const float*, State& st) const {
    const uint4* tile = reinterpret_cast<const uint4*>(buf);
    const uint16_t* ent = reinterpret_cast<const uint16_t*>(buf + L.tile_bytes + L.aux_bytes);
    uint4* t2 = reinterpret_cast<uint4*>(st.extra);
    const int n16 = m.sb[kND] * LPR;
    // ---- stage 1: dX of the window = gather(dS) + dXs (fp32 sums), rounded into the third tile
    for (int e = threadIdx.x; e < n16; e += WT) {
      const int slot = e / LPR;
      const int c = e - slot * LPR;
      int d, row, eloc;
      locate(m, L.maxd, slot, d, row, eloc);
      float acc[8], v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = 0.f;
      for (int j = 0; j < d; ++j) {
        widen8(tile[(ent[eloc + j] & GCMI_WIN_MAX_SLOTS) * LPR + c], v);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += v[q];
      }
      widen8(t2[e], v);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += v[q];
      t2[e] = pack8(acc);
    }
    __syncthreads();
    // ---- stage 2: the GraphPool backward over it
    for (int e = threadIdx.x; e < n16; e += WT) {
      const int slot = e / LPR;
      const int c = e - slot * LPR;
      int d, row, eloc;
      locate(m, L.maxd, slot, d, row, eloc);
      uint2 a = aux8(buf, L, e);
      float acc[8], g[8];
      widen8(t2[e], g);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = aux_byte(a, q) == 0 ? g[q] : 0.f;
      for (int j = 0; j < d; ++j) {
        const int en = ent[eloc + j];
        const int sl = en & GCMI_WIN_MAX_SLOTS;
        const unsigned char want = (unsigned char)((en >> GCMI_WIN_SLOT_BITS) + 1);
        widen8(t2[sl * LPR + c], g);
        a = aux8(buf, L, sl * LPR + c);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += aux_byte(a, q) == want ? g[q] : 0.f;
      }
      *reinterpret_cast<uint4*>(dy + (int64_t)row * lddy + c * 8) = pack8(acc);
    }
  }
};

// ---------------------------------------------------------------- the persistent window walker
// Workgroups [0, g_norm) walk the ordinary windows double-buffered; workgroups [g_norm, gridDim)
// walk the oversized windows (one big molecule each) using both buffers as one.
// GraphGather over the windows: out[b] = act([sum over the atoms of molecule b | max over them]) of the (folded-BatchNorm)
// rows (layers.py:6450-6479), with the arg-max rows and the raw sums the BatchNorm backward wants -- what
// readout_fwd_kernel (readout.hip) computes walking a molecule's <= 11 row runs in HBM, at 3.3 TB/s.  A window holds
// WHOLE molecules, so once its rows are in LDS every molecule of it is reduced from LDS: a thread = (molecule, 16-byte
// column piece), the molecule's rows in ascending row order (degree block by degree block: the order of
// readout_fwd_kernel, so sums are bit-identical and the first maximum wins as there).  The molecules of a window are
// the membership of its first and last rows; their row runs come from d_mol_runs.  (Those few global loads sit in the
// compute phase and wait behind the next window's DMA like the kPre loads of the accumulating ops.)
// F = the row width the outputs are laid out for; the tile holds the columns [col0, col0 + LPR * kEPP) of the rows (fp32
// rows of 128 columns do not fit two window buffers: two passes of 64).
template <bool BN, bool HB>
struct ReadoutOp {
  const float* __restrict__ scale;
  const float* __restrict__ shift;
  const int32_t* __restrict__ runs;
  const int32_t* __restrict__ membership;
  int n_deg, act, F, col0;
  float* __restrict__ out;
  int64_t ldo;
  int32_t* __restrict__ arg;
  float* __restrict__ rawsum;
  static constexpr bool kExtraTile = false;
  static constexpr int kEPP = HB ? 8 : 4;
  using State = NoState;
  __device__ __forceinline__ bool skip(int) const { return false; }
  template <int WT>
  __device__ __forceinline__ void init(float* sh_lds, int n_cols, State&) const {
    if (BN)
      for (int i = threadIdx.x; i < n_cols; i += WT) {
        sh_lds[i] = scale[col0 + i];
        sh_lds[256 + i] = shift[col0 + i];
      }
  }
  template <int WT>
  __device__ __forceinline__ void finish(char*, int, State&) const {}
  template <int WT, int LPR>
  __device__ __forceinline__ void run(const char* buf, const Layout& L, const WinMeta& m, const float* sh_lds, State&) const {
    constexpr int V = kEPP;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    // the window's molecules
    int m0 = INT_MAX, m1 = -1;
#pragma unroll
    for (int d = 0; d < kND; ++d) {
      if (d <= L.maxd && m.sb[d + 1] > m.sb[d]) {
        const int a = membership[m.rb[d] + m.sb[d]], b = membership[m.rb[d] + m.sb[d + 1] - 1];
        m0 = a < m0 ? a : m0;
        m1 = b > m1 ? b : m1;
      }
    }
    const int n_items = (m1 - m0 + 1) * LPR;
    for (int e = threadIdx.x; e < n_items; e += WT) {
      const int mi = e / LPR;
      const int c = e - mi * LPR;
      const int b = m0 + mi;
      const int2* rb = reinterpret_cast<const int2*>(runs + (int64_t)b * n_deg * 2);
      int2 run[kND];
#pragma unroll
      for (int d = 0; d < kND; ++d) run[d] = d < n_deg ? rb[d] : make_int2(0, 0);
      float sc[V], sh[V], sum[V], mx[V], raw[V], rawmx[V];
      int am[V];
#pragma unroll
      for (int q = 0; q < V; ++q) {
        sc[q] = BN ? sh_lds[c * V + q] : 1.f;
        sh[q] = BN ? sh_lds[256 + c * V + q] : 0.f;
        raw[q] = 0.f; rawmx[q] = 0.f; sum[q] = 0.f;
        mx[q] = -INFINITY;
        am[q] = -1;
      }
#pragma unroll
      for (int d = 0; d < kND; ++d) {
        for (int r = run[d].x; r < run[d].y; ++r) {
          const int slot = r - m.rb[d];
          float v[V];
          if constexpr (HB) {
            widen8(reinterpret_cast<const uint4*>(buf)[slot * LPR + c], v);
          } else {
            const float4 t = reinterpret_cast<const float4*>(buf)[slot * LPR + c];
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
          }
#pragma unroll
          for (int q = 0; q < V; ++q) {
            const float a = BN ? fmaf(v[q], sc[q], sh[q]) : v[q];
            raw[q] += v[q];
            sum[q] += a;
            if (a > mx[q]) { mx[q] = a; am[q] = r; rawmx[q] = v[q]; }
          }
        }
      }
      float* o = out + (int64_t)b * ldo + col0 + c * V;
#pragma unroll
      for (int h = 0; h < V / 4; ++h) {
        f32x4 so, mo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          so[q] = act == 1 ? tanhf(sum[4 * h + q]) : sum[4 * h + q];
          mo[q] = act == 1 ? tanhf(mx[4 * h + q]) : mx[4 * h + q];
        }
        *reinterpret_cast<f32x4*>(o + 4 * h) = so;
        *reinterpret_cast<f32x4*>(o + F + 4 * h) = mo;
        if (arg)
          *reinterpret_cast<i32x4*>(arg + (int64_t)b * F + col0 + c * V + 4 * h) =
              i32x4{am[4 * h], am[4 * h + 1], am[4 * h + 2], am[4 * h + 3]};
        if (rawsum) {  // [sum of the rows | value of the arg-max row], both BEFORE the folded BatchNorm
          float* rs = rawsum + (int64_t)b * 2 * F + col0 + c * V + 4 * h;
          *reinterpret_cast<f32x4*>(rs) = f32x4{raw[4 * h], raw[4 * h + 1], raw[4 * h + 2], raw[4 * h + 3]};
          *reinterpret_cast<f32x4*>(rs + F) = f32x4{rawmx[4 * h], rawmx[4 * h + 1], rawmx[4 * h + 2], rawmx[4 * h + 3]};
        }
      }
    }
  }
};

// ops whose third tile is filled by the DMA, one per window buffer (kExtraDma)
template <class Op>
static constexpr auto extra_dma(int) -> decltype(Op::kExtraDma) { return Op::kExtraDma; }
template <class Op>
static constexpr bool extra_dma(long) { return false; }

template <int WT, int LPR, bool AUX, class Op>
__global__ void __launch_bounds__(WT)
win_kernel(const int32_t* __restrict__ meta, const uint16_t* __restrict__ edges, int n_norm, int n_win,
           int g_norm, Layout L, Layout Lbig, const char* __restrict__ x, int64_t ldx,
           const uint8_t* __restrict__ aux, Op op, int rev) {
  // ALL LDS is one array (a second __shared__ object beside an LDS-DMA target makes hipcc wait
  // vmcnt(0) before every ds_read): [window descriptors: it, it+1, it+2][op constants][2 buffers]
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  int(*ring)[GCMI_WIN_META_INTS] = reinterpret_cast<int(*)[GCMI_WIN_META_INTS]>(smem_all);
  float* op_lds = reinterpret_cast<float*>(smem_all + kRingBytes);
  char* smem = smem_all + kHeadBytes;
  if (op.skip(LPR * Op::kEPP)) return;  // uniform over the grid
  typename Op::State ost;
  if constexpr (Op::kExtraTile) ost.extra = smem + 2 * L.buf_bytes();  // behind the two window buffers
  constexpr bool kXD = extra_dma<Op>(0);  // the third tile is per window, DMA-filled (two of them, like the buffers)
  op.template init<WT>(op_lds, LPR * Op::kEPP, ost);
  const int t = threadIdx.x;
  if ((int)blockIdx.x >= g_norm) {  // oversized windows: stage, wait, compute
    const int G = gridDim.x - g_norm;
    for (int w = n_norm + (int)blockIdx.x - g_norm; w < n_win; w += G) {
      if (t < GCMI_WIN_META_INTS) ring[0][t] = meta[(size_t)w * GCMI_WIN_META_INTS + t];
      __syncthreads();
      const WinMeta m = read_meta(ring[0]);
      stage<WT, LPR, AUX, Op::kEPP>(smem, Lbig, m, x, ldx, aux, edges);
      wait_dma();
      __syncthreads();
      op.template run<WT, LPR>(smem, Lbig, m, op_lds, ost);
      __syncthreads();
    }
    op.template finish<WT>(smem, LPR, ost);
    return;
  }
  const int G = g_norm;
  int w = blockIdx.x;  // < n_norm by construction of the grid
  // on alternate launches the ordinary windows are walked from the last one (next_sweep_direction, core.cpp): the
  // rows the previous kernel wrote last are the ones still in the Infinity Cache
  auto widx = [&](int v) { return (size_t)(rev ? n_norm - 1 - v : v); };
  const int bb = L.buf_bytes();
  int metareg = 0;
  if (t < GCMI_WIN_META_INTS) {
    ring[0][t] = meta[widx(w) * GCMI_WIN_META_INTS + t];
    if (w + G < n_norm) ring[1][t] = meta[widx(w + G) * GCMI_WIN_META_INTS + t];
    if (w + 2 * G < n_norm) metareg = meta[widx(w + 2 * G) * GCMI_WIN_META_INTS + t];
  }
  __syncthreads();
  stage<WT, LPR, AUX, Op::kEPP>(smem, L, read_meta(ring[0]), x, ldx, aux, edges);
  if constexpr (kXD) stage_extra<WT, LPR>(smem + 2 * bb, L, read_meta(ring[0]), op.extra_src(), op.extra_ld_bytes());
  int it = 0;
  for (;;) {
    wait_dma();
    __syncthreads();  // buffer it&1 has landed, the other one is free
    const bool has_next = w + G < n_norm;
    if (t < GCMI_WIN_META_INTS) {
      ring[(it + 2) % 3][t] = metareg;  // read from the next round on
      if (w + 3 * G < n_norm) metareg = meta[widx(w + 3 * G) * GCMI_WIN_META_INTS + t];
    }
    if (has_next) {
      stage<WT, LPR, AUX, Op::kEPP>(smem + ((it + 1) & 1) * bb, L, read_meta(ring[(it + 1) % 3]), x, ldx, aux, edges);
      if constexpr (kXD)
        stage_extra<WT, LPR>(smem + 2 * bb + ((it + 1) & 1) * L.tile_bytes, L, read_meta(ring[(it + 1) % 3]), op.extra_src(),
                             op.extra_ld_bytes());
    }
    if constexpr (kXD) ost.extra = smem + 2 * bb + (it & 1) * L.tile_bytes;
    op.template run<WT, LPR>(smem + (it & 1) * bb, L, read_meta(ring[it % 3]), op_lds, ost);
    if (!has_next) break;
    w += G;
    ++it;
  }
  op.template finish<WT>(smem, LPR, ost);
}

// ------------------------------------------------------------------ host-side dispatch helpers
static bool windows_disabled() {
  static int v = -1;
  if (v < 0) v = getenv("GCMI_NO_WINDOWS") != nullptr ? 1 : 0;
  return v == 1;
}

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

// (n_feat counts FLOATS per tile row: a bf16 row of n elements is a tile row of n / 2 "floats")
// aux: 0 none, 4 / 8 = aux bytes per 16-byte piece (fp32 / bf16 rows; stage())
static Layout make_layout(int alloc, int ecap, int maxd, int n_feat, int aux) {
  Layout L;
  L.tile_bytes = alloc * n_feat * 4;
  L.aux_bytes = aux == 4 ? (alloc * n_feat + 15) / 16 * 16 : aux == 8 ? (alloc * (n_feat / 4) + 63) / 64 * 512 : 0;
  L.edge_bytes = (ecap > 8 ? ecap : 8) * 2;
  L.maxd = maxd;
  return L;
}

struct WinPlan {
  Layout L, Lbig;
  size_t shmem;
  bool ok;
};

// LDS shapes: the two buffers of the ordinary windows must also hold one oversized window.
static WinPlan make_plan(const gcmi_graph* g, int n_feat, int aux) {
  int maxd = 0;
  for (int d = 1; d <= g->max_deg; ++d)
    if (g->deg_start[d + 1] > g->deg_start[d]) maxd = d;
  WinPlan p;
  p.Lbig = make_layout(g->win_alloc_big, g->win_ecap_big, maxd, n_feat, aux);
  int alloc = std::max(g->win_alloc, 1);
  p.L = make_layout(alloc, g->win_ecap, maxd, n_feat, aux);
  if (g->n_win_big > 0) {
    while (2 * p.L.buf_bytes() < p.Lbig.buf_bytes()) {
      alloc += 8;
      p.L = make_layout(alloc, g->win_ecap, maxd, n_feat, aux);
    }
  }
  p.shmem = 2 * (size_t)p.L.buf_bytes() + kHeadBytes;
  p.ok = p.shmem <= (size_t)kLdsPerCU;
  return p;
}

bool win_usable(const gcmi_graph* g, int n_feat, bool aux) {
  if (windows_disabled() || g->d_win_meta == nullptr || g->n_win <= 0) return false;
  if (g->d_win_edges == nullptr || g->n_win_big < 0 || g->n_win_big > g->n_win) return false;
  if (n_feat % 4 != 0 || n_feat > 256) return false;
  return make_plan(g, n_feat, aux ? 4 : 0).ok;
}

// bytes of an op's third LDS tile in units of the window tile (1 unless the op says otherwise: kExtraScale)
template <class Op>
static constexpr auto extra_scale(int) -> decltype(Op::kExtraScale) { return Op::kExtraScale; }
template <class Op>
static constexpr int extra_scale(long) { return 1; }

template <int WT, int LPR, bool AUX, class Op>
static int launch_wt(const gcmi_graph* g, const WinPlan& p, const char* x, int64_t ldx, const uint8_t* aux,
                     const Op& op, hipStream_t st, const char* what, int which = 0) {  // which: 0 all windows,
                                                                                        // 1 ordinary, 2 oversized only
  auto kern = win_kernel<WT, LPR, AUX, Op>;
  static bool attr_done = false;  // per instantiation
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kLdsPerCU) != hipSuccess) {
      (void)hipGetLastError();
      set_error("%s: cannot raise the dynamic LDS limit", what);
      return GCMI_ERR_LAUNCH;
    }
    attr_done = true;
  }
  const int n_norm = g->n_win - g->n_win_big;
  size_t shmem = p.shmem;
  if constexpr (Op::kExtraTile) {
    if (which != 1) {
      set_error("%s: oversized windows are not handled by the two-stage form", what);
      return GCMI_ERR_UNSUPPORTED;
    }
    shmem += (size_t)p.L.tile_bytes * extra_scale<Op>(0);
    if (shmem > (size_t)kLdsPerCU) return GCMI_ERR_UNSUPPORTED;
  }
  const int by_lds = (int)((size_t)kLdsPerCU / shmem);
  const int by_threads = 2048 / WT;
  const int per_cu = std::max(1, std::min(env_int("GCMI_WIN_PER_CU", 8), std::min(by_lds, by_threads)));
  const int g_norm = which == 2 ? 0 : std::min(n_norm, 256 * per_cu);
  const int g_big = which == 1 ? 0 : std::min(g->n_win_big, 64);
  if (g_norm + g_big == 0) return GCMI_OK;
  hipLaunchKernelGGL(kern, dim3(g_norm + g_big), dim3(WT), shmem, st, g->d_win_meta, g->d_win_edges, n_norm,
                     g->n_win, g_norm, p.L, p.Lbig, x, ldx, aux, op, next_sweep_direction_windows());
  GCMI_CHECK_LAUNCH(what);
  return GCMI_OK;
}

// threads per workgroup an op asks for (kThreads) unless GCMI_WIN_THREADS says otherwise; 512 when it has no preference
template <class Op>
static constexpr auto op_threads(int) -> decltype(Op::kThreads) { return Op::kThreads; }
template <class Op>
static constexpr int op_threads(long) { return 512; }

template <int LPR, bool AUX, class Op>
static int launch_lpr(const gcmi_graph* g, const WinPlan& p, const char* x, int64_t ldx, const uint8_t* aux,
                      const Op& op, hipStream_t st, const char* what, int which) {
  static const int wt_env = env_int("GCMI_WIN_THREADS", 0);
  const int wt = wt_env ? wt_env : op_threads<Op>(0);
  if constexpr (Op::kExtraTile) {
    // fp32 tiles: one workgroup per CU by LDS, so make it a full one (1 024 threads: 292 us against 346 at 512).  bf16
    // tiles are half the size and two 512-thread workgroups share a CU: 288 us against 367 at 1 024 (SumAccMaxBwdOpH)
    static const int wt2 = env_int("GCMI_WIN_THREADS_TWO_STAGE", 0);
    const int want = wt2 ? wt2 : (Op::kEPP == 8 ? 512 : 1024);
    if (want == 1024) return launch_wt<1024, LPR, AUX, Op>(g, p, x, ldx, aux, op, st, what, which);
    if (want == 768) return launch_wt<768, LPR, AUX, Op>(g, p, x, ldx, aux, op, st, what, which);
    if (want == 512) return launch_wt<512, LPR, AUX, Op>(g, p, x, ldx, aux, op, st, what, which);
  }
  if (wt == 1024) return launch_wt<1024, LPR, AUX, Op>(g, p, x, ldx, aux, op, st, what, which);
  if (wt == 256) return launch_wt<256, LPR, AUX, Op>(g, p, x, ldx, aux, op, st, what, which);
  return launch_wt<512, LPR, AUX, Op>(g, p, x, ldx, aux, op, st, what, which);
}

template <bool AUX, class Op>
static int launch(const gcmi_graph* g, int n_feat, const float* x, int64_t ldx, const uint8_t* aux,
                  const Op& op, hipStream_t st, const char* what, int which = 0) {
  const WinPlan p = make_plan(g, n_feat, AUX ? 4 : 0);
  const char* xb = reinterpret_cast<const char*>(x);
  switch (n_feat / 4) {
    case 16: return launch_lpr<16, AUX, Op>(g, p, xb, ldx * 4, aux, op, st, what, which);
    case 19: return launch_lpr<19, AUX, Op>(g, p, xb, ldx * 4, aux, op, st, what, which);
    case 32: return launch_lpr<32, AUX, Op>(g, p, xb, ldx * 4, aux, op, st, what, which);
    default: break;
  }
  set_error("%s: no window kernel for %d features", what, n_feat);
  return GCMI_ERR_UNSUPPORTED;
}

// rows of bf16: n_feat elements = n_feat / 8 pieces (64 -> 8, 80 -> 10, 128 -> 16)
template <class Op, bool AUX = false>
static int launch_h(const gcmi_graph* g, int n_feat, const bf16_t* x, int64_t ldx, const Op& op, hipStream_t st,
                    const char* what, int which = 0, const uint8_t* aux = nullptr) {
  const WinPlan p = make_plan(g, n_feat / 2, AUX ? 8 : 0);
  const char* xb = reinterpret_cast<const char*>(x);
  switch (n_feat / 8) {
    case 8: return launch_lpr<8, AUX, Op>(g, p, xb, ldx * 2, aux, op, st, what, which);
    case 10: return launch_lpr<10, AUX, Op>(g, p, xb, ldx * 2, aux, op, st, what, which);
    case 16: return launch_lpr<16, AUX, Op>(g, p, xb, ldx * 2, aux, op, st, what, which);
    default: break;
  }
  set_error("%s: no bf16 window kernel for %d features", what, n_feat);
  return GCMI_ERR_UNSUPPORTED;
}

// GraphGather forward over the windows (ReadoutOp): GCMI_ERR_UNSUPPORTED where the window form does not apply -- no
// window plan, molecules without atoms (the plan says how many molecules its windows cover: gcmi_graph.win_reserved[0]),
// other widths than 128 columns -- and the caller walks the row runs instead (readout.hip).
int win_readout(const gcmi_graph* g, const float* d_x, int64_t ldx, int n_feat, const float* d_scale, const float* d_shift,
                int act, float* d_out, int64_t ldo, int32_t* d_arg, float* d_rawsum, int x_bf16, hipStream_t st) {
  // Kept, OFF (GCMI_READOUT_WINDOWS=1 turns it on; results identical, tests pass with it): measured 311 + 297 us for the
  // two fp32 passes against readout_fwd_kernel's 187, and 737 us against 130 on bf16 rows.  A window is ~25 molecules, so
  // the compute phase has 25 x 16 serial chains of ~18 rows for 512 threads at ONE workgroup per CU (LDS), and its
  // run-bound loads wait behind the next window's DMA; the row-run walk keeps thousands of molecules in flight per CU.
  static const bool on = getenv("GCMI_READOUT_WINDOWS") && atoi(getenv("GCMI_READOUT_WINDOWS")) == 1;
  if (!on || n_feat != 128 || g->win_reserved[0] != g->n_mols || g->n_mols <= 0 || g->d_membership == nullptr ||
      g->d_mol_runs == nullptr || (reinterpret_cast<uintptr_t>(g->d_mol_runs) & 7u) || ldo % 4 || !aligned16(d_out) ||
      (d_arg && !aligned16(d_arg)) || (d_rawsum && !aligned16(d_rawsum)))
    return GCMI_ERR_UNSUPPORTED;
  const bool bn = d_scale != nullptr;
  if (x_bf16) {
    if (!win_usable_h(g, 128) || (reinterpret_cast<uintptr_t>(d_x) & 15u) || ldx % 8) return GCMI_ERR_UNSUPPORTED;
    const bf16_t* xh = reinterpret_cast<const bf16_t*>(d_x);
    if (bn) {
      ReadoutOp<true, true> op{d_scale, d_shift, g->d_mol_runs, g->d_membership, g->max_deg + 1, act, 128, 0, d_out, ldo, d_arg, d_rawsum};
      return launch_h<ReadoutOp<true, true>>(g, 128, xh, ldx, op, st, "win_readout");
    }
    ReadoutOp<false, true> op{nullptr, nullptr, g->d_mol_runs, g->d_membership, g->max_deg + 1, act, 128, 0, d_out, ldo, d_arg, d_rawsum};
    return launch_h<ReadoutOp<false, true>>(g, 128, xh, ldx, op, st, "win_readout");
  }
  // fp32 rows: two passes of 64 columns (512-byte rows do not fit two window buffers)
  if (!win_usable(g, 64, false) || !aligned16(d_x) || ldx % 4) return GCMI_ERR_UNSUPPORTED;
  for (int col0 = 0; col0 < 128; col0 += 64) {
    int rc;
    if (bn) {
      ReadoutOp<true, false> op{d_scale, d_shift, g->d_mol_runs, g->d_membership, g->max_deg + 1, act, 128, col0, d_out, ldo, d_arg, d_rawsum};
      rc = launch<false, ReadoutOp<true, false>>(g, 64, d_x + col0, ldx, nullptr, op, st, "win_readout");
    } else {
      ReadoutOp<false, false> op{nullptr, nullptr, g->d_mol_runs, g->d_membership, g->max_deg + 1, act, 128, col0, d_out, ldo, d_arg, d_rawsum};
      rc = launch<false, ReadoutOp<false, false>>(g, 64, d_x + col0, ldx, nullptr, op, st, "win_readout");
    }
    if (rc) return rc;
  }
  return GCMI_OK;
}

bool win_has_width(int n_feat) { return n_feat == 64 || n_feat == 76 || n_feat == 128; }

int win_gather_sum(const gcmi_graph* g, const float* d_x, int64_t ldx, int n_feat, float* d_s,
                   int64_t lds, hipStream_t st, bool accumulate) {
  if (accumulate) {
    SumOp<true> op{d_s, lds};
    return launch<false>(g, n_feat, d_x, ldx, nullptr, op, st, "win_gather_sum (accumulate)");
  }
  SumOp<false> op{d_s, lds};
  return launch<false>(g, n_feat, d_x, ldx, nullptr, op, st, "win_gather_sum");
}

int win_gather_max(const gcmi_graph* g, const float* d_x, int64_t ldx, int n_feat, const float* d_scale,
                   const float* d_shift, float* d_out, int64_t ldo, uint8_t* d_arg, hipStream_t st) {
  if (d_scale) {
    MaxOp<true> op{d_scale, d_shift, d_out, ldo, d_arg};
    return launch<false>(g, n_feat, d_x, ldx, nullptr, op, st, "win_gather_max");
  }
  MaxOp<false> op{nullptr, nullptr, d_out, ldo, d_arg};
  return launch<false>(g, n_feat, d_x, ldx, nullptr, op, st, "win_gather_max");
}

int win_gather_max_bwd(const gcmi_graph* g, const float* d_dout, int64_t lddo, int n_feat,
                       const uint8_t* d_arg, float* d_dx, int64_t lddx, hipStream_t st) {
  MaxBwdOp<false> op{d_dx, lddx, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
  return launch<true>(g, n_feat, d_dout, lddo, d_arg, op, st, "win_gather_max_bwd");
}

int win_gather_max_bwd_if_ill(const gcmi_graph* g, const float* d_dout, int64_t lddo, int n_feat, const uint8_t* d_arg,
                              float* d_dx, int64_t lddx, const float* d_gamma, const float* d_beta, hipStream_t st) {
  MaxBwdOp<false> op{d_dx, lddx, nullptr, 0, nullptr, nullptr, nullptr, d_gamma, d_beta};
  return launch<true>(g, n_feat, d_dout, lddo, d_arg, op, st, "win_gather_max_bwd (conditional)");
}

// ---- bf16 activation storage
bool win_usable_h(const gcmi_graph* g, int n_feat) {
  if (windows_disabled() || g->d_win_meta == nullptr || g->n_win <= 0) return false;
  if (g->d_win_edges == nullptr || g->n_win_big < 0 || g->n_win_big > g->n_win) return false;
  if (n_feat != 64 && n_feat != 80 && n_feat != 128) return false;
  return make_plan(g, n_feat / 2, 0).ok;
}

// fp32 rows of n_feat (76) columns -> bf16 neighbour sums and a bf16 copy of the rows, both `ldo` (80) wide
int win_gather_sum_fh(const gcmi_graph* g, const float* d_x, int64_t ldx, int n_feat, bf16_t* d_s, bf16_t* d_xcopy,
                      int64_t ldo, hipStream_t st) {
  if (n_feat % 4 != 0 || ldo < n_feat || (ldo - n_feat) % 4 != 0 || ldo % 8 != 0 || !aligned16(d_s) || !aligned16(d_xcopy)) {
    set_error("win_gather_sum (fp32 -> bf16): bad shape (n_feat %d, ldo %lld)", n_feat, (long long)ldo);
    return GCMI_ERR_UNSUPPORTED;
  }
  SumOpFH op{d_s, d_xcopy, ldo, (int)((ldo - n_feat) / 4)};
  return launch<false>(g, n_feat, d_x, ldx, nullptr, op, st, "win_gather_sum (fp32 -> bf16)");
}

int win_gather_sum_h(const gcmi_graph* g, const bf16_t* d_x, int64_t ldx, int n_feat, bf16_t* d_s, int64_t lds,
                     hipStream_t st) {
  SumOpH op{d_s, lds};
  return launch_h(g, n_feat, d_x, ldx, op, st, "win_gather_sum (bf16)");
}

int win_gather_max_h(const gcmi_graph* g, const bf16_t* d_x, int64_t ldx, int n_feat, const float* d_scale,
                     const float* d_shift, bf16_t* d_out, int64_t ldo, uint8_t* d_arg, hipStream_t st) {
  if (d_scale) {
    MaxOpH<true> op{d_scale, d_shift, d_out, ldo, d_arg};
    return launch_h(g, n_feat, d_x, ldx, op, st, "win_gather_max (bf16)");
  }
  MaxOpH<false> op{nullptr, nullptr, d_out, ldo, d_arg};
  return launch_h(g, n_feat, d_x, ldx, op, st, "win_gather_max (bf16)");
}

// ---- gradient streams in bf16 (storage == 2)
bool win_usable_gh(const gcmi_graph* g, int n_feat) {
  if (windows_disabled() || g->d_win_meta == nullptr || g->n_win <= 0) return false;
  if (g->d_win_edges == nullptr || g->n_win_big < 0 || g->n_win_big > g->n_win || n_feat != 64) return false;
  return make_plan(g, n_feat / 2, 8).ok;
}

int win_gather_max_bwd_h(const gcmi_graph* g, const bf16_t* d_dout, int64_t lddo, int n_feat, const uint8_t* d_arg,
                         bf16_t* d_dx, int64_t lddx, const float* only_if_gamma, const float* only_if_beta, hipStream_t st) {
  MaxBwdOpH op{d_dx, lddx, only_if_gamma, only_if_beta};
  return launch_h<MaxBwdOpH, true>(g, n_feat, d_dout, lddo, op, st, "win_gather_max_bwd (bf16)", 0, d_arg);
}

bool win_two_stage_usable_h(const gcmi_graph* g, int n_feat) {
  if (!win_usable_gh(g, n_feat)) return false;
  const WinPlan p = make_plan(g, n_feat / 2, 8);
  return p.shmem + (size_t)2 * p.L.tile_bytes <= (size_t)kLdsPerCU;
}

// d_dxs holds the self part of dX on entry (bf16); on return d_dy holds the GraphPool backward of the complete dX.
int win_gather_sumacc_max_bwd_h(const gcmi_graph* g, const bf16_t* d_ds, int64_t ldds, int n_feat, bf16_t* d_dxs,
                                int64_t lddxs, const uint8_t* d_arg, bf16_t* d_dy, int64_t lddy, hipStream_t st) {
  SumAccMaxBwdOpH op{d_dxs, lddxs, d_dy, lddy};
  int rc = launch_h<SumAccMaxBwdOpH, true>(g, n_feat, d_ds, ldds, op, st, "win_gather_sumacc_max_bwd (bf16)", 1, d_arg);
  if (rc || g->n_win_big == 0) return rc;
  SumAccOpH acc{d_dxs, lddxs};
  rc = launch_h(g, n_feat, d_ds, ldds, acc, st, "win_gather_sum (bf16, accumulate, oversized windows)", 2);
  if (rc) return rc;
  MaxBwdOpH mb{d_dy, lddy, nullptr, nullptr};
  return launch_h<MaxBwdOpH, true>(g, n_feat, d_dxs, lddxs, mb, st, "win_gather_max_bwd (bf16, oversized windows)", 2, d_arg);
}

// GraphPool of this block + sum_neigh of the next in one window pass (MaxSumOpH); the oversized windows (a molecule
// above the window cap each) take the two separate ops over their own rows
bool win_max_sum_usable_h(const gcmi_graph* g, int n_feat) {
  // measured on the 65 536-molecule step: 288 us for the fused pass against 196 + 72 for the two it replaces (the second
  // stage waits behind a barrier for the whole window's pooled rows) -- 153 MB less traffic, 20 us more time: off by
  // default, GCMI_FUSED_POOL_SUM=1 switches it on
  static const bool on = getenv("GCMI_FUSED_POOL_SUM") && atoi(getenv("GCMI_FUSED_POOL_SUM")) != 0;
  if (!on || !win_usable_h(g, n_feat)) return false;
  const WinPlan p = make_plan(g, n_feat / 2, 0);
  return p.shmem + (size_t)p.L.tile_bytes <= (size_t)kLdsPerCU;
}

int win_gather_max_sum_h(const gcmi_graph* g, const bf16_t* d_x, int64_t ldx, int n_feat, const float* d_scale,
                         const float* d_shift, bf16_t* d_out, int64_t ldo, uint8_t* d_arg, bf16_t* d_s, int64_t lds,
                         hipStream_t st) {
  int rc;
  if (d_scale) {
    MaxSumOpH<true> op{d_scale, d_shift, d_out, ldo, d_arg, d_s, lds};
    rc = launch_h(g, n_feat, d_x, ldx, op, st, "win_gather_max_sum (bf16)", 1);
  } else {
    MaxSumOpH<false> op{nullptr, nullptr, d_out, ldo, d_arg, d_s, lds};
    rc = launch_h(g, n_feat, d_x, ldx, op, st, "win_gather_max_sum (bf16)", 1);
  }
  if (rc || g->n_win_big == 0) return rc;
  if (d_scale) {
    MaxOpH<true> mx{d_scale, d_shift, d_out, ldo, d_arg};
    rc = launch_h(g, n_feat, d_x, ldx, mx, st, "win_gather_max (bf16, oversized windows)", 2);
  } else {
    MaxOpH<false> mx{nullptr, nullptr, d_out, ldo, d_arg};
    rc = launch_h(g, n_feat, d_x, ldx, mx, st, "win_gather_max (bf16, oversized windows)", 2);
  }
  if (rc) return rc;
  SumOpH sm{d_s, lds};
  return launch_h(g, n_feat, d_out, ldo, sm, st, "win_gather_sum (bf16, oversized windows)", 2);
}

// dy = GraphPool backward of (dXs + gather of dS), dX kept in LDS only.  GCMI_ERR_UNSUPPORTED: oversized windows in
// the batch, or no LDS for the third tile.
bool win_two_stage_usable(const gcmi_graph* g, int n_feat) {
  if (!win_has_width(n_feat) || !win_usable(g, n_feat, true)) return false;
  const WinPlan p = make_plan(g, n_feat, 4);
  return p.shmem + (size_t)2 * p.L.tile_bytes <= (size_t)kLdsPerCU;  // two third tiles (SumAccMaxBwdOp::kExtraScale)
}

// d_dxs holds the self part of dX on entry; on return d_dy holds the GraphPool backward of the complete dX.  The
// ordinary windows take the two-stage pass (dX in LDS only); the few oversized ones (a molecule above the window cap
// each) take the two separate passes over their own rows, which completes d_dxs there.
int win_gather_sumacc_max_bwd(const gcmi_graph* g, const float* d_ds, int64_t ldds, int n_feat, float* d_dxs,
                              int64_t lddxs, const uint8_t* d_arg, float* d_dy, int64_t lddy, hipStream_t st) {
  SumAccMaxBwdOp op{d_dxs, lddxs, d_dy, lddy};
  int rc = launch<true>(g, n_feat, d_ds, ldds, d_arg, op, st, "win_gather_sumacc_max_bwd", 1);
  if (rc || g->n_win_big == 0) return rc;
  SumOp<true> acc{d_dxs, lddxs};
  rc = launch<false>(g, n_feat, d_ds, ldds, nullptr, acc, st, "win_gather_sum (accumulate, oversized windows)", 2);
  if (rc) return rc;
  MaxBwdOp<false> mb{d_dy, lddy, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
  return launch<true>(g, n_feat, d_dxs, lddxs, d_arg, mb, st, "win_gather_max_bwd (oversized windows)", 2);
}

// threads per workgroup of the window kernels must be a multiple of the row's 16-byte pieces for the statistics form
bool win_stats_usable(const gcmi_graph* g, int n_feat) {
  return (n_feat == 64 || n_feat == 128) && win_usable(g, n_feat, true);
}

int win_gather_max_bwd_stats(const gcmi_graph* g, const float* d_dout, int64_t lddo, int n_feat, const uint8_t* d_arg,
                             float* d_dx, int64_t lddx, const float* d_x, int64_t ldx, const float* d_mean,
                             const float* d_invstd, double* d_sums, hipStream_t st) {
  MaxBwdOp<true> op{d_dx, lddx, d_x, ldx, d_mean, d_invstd, d_sums, nullptr, nullptr};
  return launch<true>(g, n_feat, d_dout, lddo, d_arg, op, st, "win_gather_max_bwd (statistics)");
}

}  // namespace gcmi
