// K4 readout: GraphGather = per-molecule [sum | max] over atom rows (+tanh),
// and the per-molecule row ranges ("mol runs") it walks.
//
// In a collated batch the atoms of molecule b are NOT contiguous: the batch is
// sorted by degree first, so b owns one short contiguous run of rows inside
// every degree block (membership is ascending inside a block).  The plan
// d_mol_runs[b][d] = [begin,end) lists those <= 11 runs; a group of F/4 lanes
// (one 16-byte column chunk per lane) then walks the runs of ONE molecule in
// ascending row order and keeps running sum / max / arg-max in registers: a
// segmented reduction with no atomics, no sort, deterministic summation order
// and the reference's first-maximum tie rule (lowest row wins,
// utils/pytorch_utils.py:524-526 -> torch.max(dim=0)).
// Bound: HBM.  Algorithmic bytes per launch: N*(4F+4) + B*8F.
#include <limits.h>
#include <math.h>

#include <atomic>

#include "common.h"
#include "split_bf16.h"

namespace gcmi {

constexpr int kRBlock = 256;

__global__ void __launch_bounds__(kRBlock)
mol_runs_kernel(DegTable t, int n_atoms, int n_mols, const int32_t* __restrict__ membership,
                int32_t* __restrict__ runs, int32_t* __restrict__ flag) {
  const int n_deg = t.max_deg + 1;
  for (int i = blockIdx.x * kRBlock + threadIdx.x; i < n_atoms; i += gridDim.x * kRBlock) {
    const int d = degree_of_row(t, i);
    int lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k <= GCMI_MAX_DEG; ++k) {
      if (k == d) {
        lo = t.deg_start[k];
        hi = t.deg_start[k + 1];
      }
    }
    const int b = membership[i];
    if (b < 0 || b >= n_mols) {
      if (flag) *flag = 1;
      continue;
    }
    const int prev = (i > lo) ? membership[i - 1] : -1;
    const int next = (i + 1 < hi) ? membership[i + 1] : INT_MAX;
    if (prev > b && flag) *flag = 1;
    int32_t* r = runs + ((int64_t)b * n_deg + d) * 2;
    if (prev != b) r[0] = i;
    if (next != b) r[1] = i + 1;
  }
}

// PRE (host: a lane group of 11..64 lanes inside one wave, one column chunk per lane): the molecule's run bounds
// are fetched once, one run per lane, and handed round the group with cross-lane reads; the rows are then walked
// as ONE sequence of four-row rounds across the runs, the loads of rounds k + 1 (and, DEPTH 3, k + 2) issued before
// round k is consumed.
// Same rows in the same order as the plain form (bit-identical sums, same first-maximum rule); what changes is the
// number of dependent memory round trips per molecule: 1 + ceil(atoms / 4) with two rounds in flight, against one
// per degree + one per four rows of every run, one at a time.
// HB: the rows are stored as bf16 (gcmi_model_desc.storage == 1; ldx counts elements): a lane's four columns are one
// 8-byte load, widened; sums, maxima and everything written stay fp32.
template <int V, bool BN, bool PRE, int DEPTH = 2, bool HB = false>
__global__ void __launch_bounds__(kRBlock)
readout_fwd_kernel(int n_mols, int n_deg, const int32_t* __restrict__ runs,
                   const float* __restrict__ x, int64_t ldx, int n_feat, int lpr, int gl,
                   const float* __restrict__ scale, const float* __restrict__ shift, int act,
                   float* __restrict__ out, int64_t ldo, int32_t* __restrict__ arg, float* __restrict__ rawsum,
                   int vec_out) {
  const int mpb = (int)blockDim.x / gl;  // molecules per workgroup
  const int grp = threadIdx.x / gl;
  const int lane = threadIdx.x - grp * gl;
  if (grp >= mpb) return;
  const int b = blockIdx.x * mpb + grp;
  if (b >= n_mols) return;
  const int32_t* rb = runs + (int64_t)b * n_deg * 2;
  for (int cc = lane; cc < lpr; cc += gl) {
    const int c = cc * V;
    float sc[V], sh[V], sum[V], mx[V], raw[V], rawmx[V];
    int am[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      sc[q] = BN ? scale[c + q] : 1.f;
      sh[q] = BN ? shift[c + q] : 0.f;
      raw[q] = 0.f;
      rawmx[q] = 0.f;
      sum[q] = 0.f;
      mx[q] = -INFINITY;
      am[q] = -1;
    }
    auto take = [&](float (&v)[4][V], const int (&idx)[4], int n) {
      // every row of the round is touched on every path (an empty asm): the waits for the round's loads are then
      // unconditional, and the compiler knows at the next round that none of these registers is still a load target
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < V; ++q) asm volatile("" : "+v"(v[u][q]));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u < n) {
          const int r = idx[u];
#pragma unroll
          for (int q = 0; q < V; ++q) {
            const float a = BN ? fmaf(v[u][q], sc[q], sh[q]) : v[u][q];
            raw[q] += v[u][q];
            sum[q] += a;
            if (a > mx[q]) { mx[q] = a; am[q] = r; rawmx[q] = v[u][q]; }
          }
        }
      }
    };
    if constexpr (PRE) {
      // lane d holds run d.  Loaded unconditionally and passed through an empty asm: the wait for this one load
      // then sits here, before any row is requested, and the cross-lane reads below carry no memory wait
      int2 mine = reinterpret_cast<const int2*>(rb)[lane < n_deg ? lane : 0];
      asm volatile("" : "+v"(mine.x), "+v"(mine.y));
      // cursor over the molecule's rows in ascending order: [rr, r1) is what is left of the current run, and it is
      // moved on to the next non-empty run as soon as it empties (rr < r1 <=> rows are left); uniform over the group
      int d = 0, rr = 0, r1 = 0, last = 0;
      auto seek = [&]() {
        while (rr >= r1 && d < n_deg) {
          rr = __shfl(mine.x, d, gl);
          r1 = __shfl(mine.y, d, gl);
          ++d;
        }
      };
      seek();
      // a round: the next (up to) four rows, whatever runs they lie in; slots past the end re-read a row already read
      auto issue = [&](float (&v)[4][V], int (&idx)[4], int& n) {
        n = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool ok = rr < r1;
          idx[u] = ok ? rr : last;
          if (ok) {
            last = rr;
            ++rr;
            ++n;
            seek();
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float4 tv;
          if constexpr (HB)
            tv = widen4(*reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(x) + (int64_t)idx[u] * ldx + c));
          else
            tv = *reinterpret_cast<const float4*>(x + (int64_t)idx[u] * ldx + c);
          v[u][0] = tv.x; v[u][1] = tv.y; v[u][2] = tv.z; v[u][3] = tv.w;
        }
      };
      float va[4][V], vb[4][V];
      int ia[4], ib[4], an = 0, bn = 0;
      issue(va, ia, an);
      // one exit, at the top: with a way out between the halves the compiler cannot count the loads in flight and
      // waits for all of them.  Inside, round B is never empty; the second A may be (at most one such per molecule)
      if constexpr (DEPTH == 2) {
        while (rr < r1) {
          issue(vb, ib, bn);
          take(va, ia, an);
          issue(va, ia, an);
          take(vb, ib, bn);
        }
        take(va, ia, an);
      } else {  // three rounds in flight
        float vc[4][V];
        int ic[4], cn = 0;
        issue(vb, ib, bn);
        while (rr < r1) {
          issue(vc, ic, cn);
          take(va, ia, an);
          issue(va, ia, an);
          take(vb, ib, bn);
          issue(vb, ib, bn);
          take(vc, ic, cn);
        }
        take(va, ia, an);
        take(vb, ib, bn);
      }
    } else
    for (int d = 0; d < n_deg; ++d) {
      const int r0 = rb[2 * d], r1 = rb[2 * d + 1];
      // four rows of the run per round, loads issued together, consumed in row order
      for (int rr = r0; rr < r1; rr += 4) {
        float v[4][V];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = rr + u < r1 ? rr + u : rr;
          if constexpr (V == 4) {
            float4 tv;
            if constexpr (HB)
              tv = widen4(*reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(x) + (int64_t)r * ldx + c));
            else
              tv = *reinterpret_cast<const float4*>(x + (int64_t)r * ldx + c);
            v[u][0] = tv.x; v[u][1] = tv.y; v[u][2] = tv.z; v[u][3] = tv.w;
          } else {
            v[u][0] = x[(int64_t)r * ldx + c];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = rr + u;
          if (r >= r1) break;
#pragma unroll
          for (int q = 0; q < V; ++q) {
            const float a = BN ? fmaf(v[u][q], sc[q], sh[q]) : v[u][q];
            raw[q] += v[u][q];
            sum[q] += a;
            if (a > mx[q]) { mx[q] = a; am[q] = r; rawmx[q] = v[u][q]; }
          }
        }
      }
    }
    float* o = out + (int64_t)b * ldo;
    float so[V], mo[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      so[q] = act == 1 ? tanhf(sum[q]) : sum[q];
      mo[q] = act == 1 ? tanhf(mx[q]) : mx[q];
    }
    if constexpr (V == 4) {
      if (vec_out) {  // uniform: 16-byte stores (four 4-byte stores per lane touch every line four times)
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<f32x4*>(o + c) = f32x4{so[0], so[1], so[2], so[3]};
        *reinterpret_cast<f32x4*>(o + n_feat + c) = f32x4{mo[0], mo[1], mo[2], mo[3]};
        if (arg) *reinterpret_cast<i32x4*>(arg + (int64_t)b * n_feat + c) = i32x4{am[0], am[1], am[2], am[3]};
        if (rawsum) {  // [sum of the rows | value of the arg-max row], both BEFORE the folded BatchNorm
          *reinterpret_cast<f32x4*>(rawsum + (int64_t)b * 2 * n_feat + c) = f32x4{raw[0], raw[1], raw[2], raw[3]};
          *reinterpret_cast<f32x4*>(rawsum + (int64_t)b * 2 * n_feat + n_feat + c) =
              f32x4{rawmx[0], rawmx[1], rawmx[2], rawmx[3]};
        }
        continue;
      }
    }
#pragma unroll
    for (int q = 0; q < V; ++q) {
      o[c + q] = so[q];
      o[n_feat + c + q] = mo[q];
      if (arg) arg[(int64_t)b * n_feat + c + q] = am[q];
      if (rawsum) {
        rawsum[(int64_t)b * 2 * n_feat + c + q] = raw[q];
        rawsum[(int64_t)b * 2 * n_feat + n_feat + c + q] = rawmx[q];
      }
    }
  }
}

template <int V>
__global__ void __launch_bounds__(kRBlock)
readout_bwd_kernel(int64_t slots, int lpr, int n_feat, const int32_t* __restrict__ membership,
                   const float* __restrict__ dout, int64_t lddo, const float* __restrict__ out,
                   int64_t ldo, int act, const int32_t* __restrict__ arg, float* __restrict__ dx,
                   int64_t lddx) {
  for (int64_t e = (int64_t)blockIdx.x * kRBlock + threadIdx.x; e < slots;
       e += (int64_t)gridDim.x * kRBlock) {
    const int i = (int)(e / lpr);
    const int c = (int)(e - (int64_t)i * lpr) * V;
    const int b = membership[i];
    float g[V], gs[V], gm[V], os[V], om[V];
    int am[V];
    const float* drow = dout + (int64_t)b * lddo;
    const float* orow = out + (int64_t)b * ldo;
    if constexpr (V == 4) {
      const float4 a4 = *reinterpret_cast<const float4*>(drow + c);
      const float4 b4 = *reinterpret_cast<const float4*>(drow + n_feat + c);
      const int4 i4 = *reinterpret_cast<const int4*>(arg + (int64_t)b * n_feat + c);
      gs[0] = a4.x; gs[1] = a4.y; gs[2] = a4.z; gs[3] = a4.w;
      gm[0] = b4.x; gm[1] = b4.y; gm[2] = b4.z; gm[3] = b4.w;
      am[0] = i4.x; am[1] = i4.y; am[2] = i4.z; am[3] = i4.w;
      if (act == 1) {
        const float4 c4 = *reinterpret_cast<const float4*>(orow + c);
        const float4 d4 = *reinterpret_cast<const float4*>(orow + n_feat + c);
        os[0] = c4.x; os[1] = c4.y; os[2] = c4.z; os[3] = c4.w;
        om[0] = d4.x; om[1] = d4.y; om[2] = d4.z; om[3] = d4.w;
      }
    } else {
      gs[0] = drow[c];
      gm[0] = drow[n_feat + c];
      am[0] = arg[(int64_t)b * n_feat + c];
      if (act == 1) {
        os[0] = orow[c];
        om[0] = orow[n_feat + c];
      }
    }
#pragma unroll
    for (int q = 0; q < V; ++q) {
      float s1 = gs[q], m1 = gm[q];
      if (act == 1) {
        s1 *= (1.f - os[q] * os[q]);
        m1 *= (1.f - om[q] * om[q]);
      }
      g[q] = s1 + ((am[q] == i) ? m1 : 0.f);
    }
    if constexpr (V == 4) {
      *reinterpret_cast<float4*>(dx + (int64_t)i * lddx + c) = make_float4(g[0], g[1], g[2], g[3]);
    } else {
      dx[(int64_t)i * lddx + c] = g[0];
    }
  }
}

// g[b, :] *= 1 - out[b, :]^2 over the 2F columns of every molecule (tanh derivative of the readout),
// in place: the per-molecule gradient the fused BatchNorm backward gathers from
__global__ void readout_grad_prep_kernel(float* __restrict__ g, int64_t ldg, const float* __restrict__ out,
                                         int64_t ldo, int64_t n_mols, int n2) {
  const int64_t total = n_mols * n2;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / n2;
    const int c = (int)(e - b * n2);
    const float o = out[b * ldo + c];
    g[b * ldg + c] *= (1.f - o * o);
  }
}

int readout_grad_prep(float* d_g, int64_t ldg, const float* d_out, int64_t ldo, int64_t n_mols, int n_feat,
                      hipStream_t st) {
  if (n_mols == 0) return GCMI_OK;
  hipLaunchKernelGGL(readout_grad_prep_kernel, dim3(grid_for(n_mols * 2 * n_feat, 256)), dim3(256), 0, st, d_g, ldg,
                     d_out, ldo, n_mols, 2 * n_feat);
  GCMI_CHECK_LAUNCH("readout_grad_prep");
  return GCMI_OK;
}

}  // namespace gcmi

using namespace gcmi;

extern "C" {

int gcmi_build_mol_runs(const gcmi_graph* g, int32_t* d_mol_runs, int32_t* d_flag, void* stream) {
  int rc = check_graph(g, false);
  if (rc) return rc;
  GCMI_CHECK_ARG(d_mol_runs != nullptr || g->n_mols == 0, "build_mol_runs: NULL output");
  GCMI_CHECK_ARG(g->n_atoms == 0 || g->d_membership != nullptr, "build_mol_runs: d_membership is NULL");
  hipStream_t st = (hipStream_t)stream;
  const size_t bytes = (size_t)g->n_mols * (g->max_deg + 1) * 2 * sizeof(int32_t);
  if (bytes) {
    if (hipMemsetAsync(d_mol_runs, 0, bytes, st) != hipSuccess) {
      set_error("build_mol_runs: memset failed");
      return GCMI_ERR_LAUNCH;
    }
  }
  if (d_flag && hipMemsetAsync(d_flag, 0, sizeof(int32_t), st) != hipSuccess) {
    set_error("build_mol_runs: memset failed");
    return GCMI_ERR_LAUNCH;
  }
  if (g->n_atoms == 0) return GCMI_OK;
  DegTable t = make_deg_table(g);
  hipLaunchKernelGGL(mol_runs_kernel, dim3(grid_for(g->n_atoms, kRBlock)), dim3(kRBlock), 0, st, t,
                     g->n_atoms, g->n_mols, g->d_membership, d_mol_runs, d_flag);
  GCMI_CHECK_LAUNCH("build_mol_runs");
  return GCMI_OK;
}

int gcmi_readout_fwd(const gcmi_graph* g, const float* d_x, int64_t ldx, int32_t n_feat,
                     const float* d_scale, const float* d_shift, int32_t act, float* d_out,
                     int64_t ldo, int32_t* d_arg, void* stream) {
  return gcmi::readout_fwd_impl(g, d_x, ldx, n_feat, d_scale, d_shift, act, d_out, ldo, d_arg, nullptr, stream);
}

}  // extern "C"

namespace gcmi {
static std::atomic<int> g_readout_pre{getenv("GCMI_READOUT_PRE") && atoi(getenv("GCMI_READOUT_PRE")) == 0 ? 0 : 1};
void set_readout_pipelined(int on) { g_readout_pre.store(on ? 1 : 0, std::memory_order_relaxed); }
int get_readout_pipelined() { return g_readout_pre.load(std::memory_order_relaxed); }

// d_rawsum (may be NULL): [n_mols x 2 n_feat] per-molecule [sums of the input rows | value of the arg-max row], both
// before the folded BatchNorm -- what
// the BatchNorm backward behind this readout needs to get its column sums without another pass over the atoms
int readout_fwd_impl(const gcmi_graph* g, const float* d_x, int64_t ldx, int32_t n_feat, const float* d_scale,
                     const float* d_shift, int32_t act, float* d_out, int64_t ldo, int32_t* d_arg, float* d_rawsum,
                     void* stream, int32_t x_bf16) {
  int rc = check_graph(g, false);
  if (rc) return rc;
  GCMI_CHECK_ARG(n_feat > 0 && ldx >= n_feat && ldo >= 2 * (int64_t)n_feat, "readout: bad n_feat/ld");
  GCMI_CHECK_ARG(g->n_mols == 0 || (d_out && g->d_mol_runs), "readout: NULL output or mol runs");
  GCMI_CHECK_ARG(g->n_atoms == 0 || d_x, "readout: NULL input");
  GCMI_CHECK_ARG((d_scale == nullptr) == (d_shift == nullptr), "readout: scale/shift must come together");
  GCMI_CHECK_ARG(act == 0 || act == 1, "readout: act must be 0 or 1");
  if (g->n_mols == 0) return GCMI_OK;
  hipStream_t st = (hipStream_t)stream;
  // (bf16 rows: 8-byte pieces, so the alignment test runs on half the element counts)
  const int V = x_bf16 ? ((reinterpret_cast<uintptr_t>(d_x) & 7u) == 0 && ldx % 4 == 0 && n_feat % 4 == 0 ? 4 : 1)
                       : vec_width(d_x, ldx, n_feat);
  if (x_bf16 && V != 4) {
    set_error("readout (bf16 rows): rows must be 8-byte addressable");
    return GCMI_ERR_UNSUPPORTED;
  }
  const int lpr = n_feat / V;
  const int gl = lpr < kRBlock ? lpr : kRBlock;
  const int tb = kRBlock;  // (64- and 128-thread workgroups measured 3 % slower with the pipelined walk)
  const int mpb = tb / gl;
  const int blocks = (g->n_mols + mpb - 1) / mpb;
  const bool bn = d_scale != nullptr;
  const int n_deg = g->max_deg + 1;
  const int vec_out = (n_feat % 4 == 0 && ldo % 4 == 0 && aligned16(d_out) && (d_arg == nullptr || aligned16(d_arg)) &&
                       (d_rawsum == nullptr || aligned16(d_rawsum)))
                          ? 1
                          : 0;
  TimedScope ts(GCMI_K_READOUT, st);
  if (g->n_atoms > 0) {  // the window form where it applies (rows staged once through LDS, molecules reduced there)
    const int wrc = win_readout(g, d_x, ldx, n_feat, d_scale, d_shift, act, d_out, ldo, d_arg, d_rawsum, x_bf16, st);
    if (wrc != GCMI_ERR_UNSUPPORTED) return wrc;
  }
  // run bounds in registers + two rounds in flight: lane groups of whole power-of-two size inside one wave that
  // can hold one run per lane (GCMI_OPT_READOUT_PIPELINED / GCMI_READOUT_PRE=0: the plain walk)
  const bool pre = get_readout_pipelined() != 0 && V == 4 && g->n_atoms > 0 && gl == lpr && gl >= n_deg && gl <= 64 && (gl & (gl - 1)) == 0 &&
                   (reinterpret_cast<uintptr_t>(g->d_mol_runs) & 7u) == 0;
#define LAUNCH_RO(VV, BB, PP)                                                                        \
  hipLaunchKernelGGL((readout_fwd_kernel<VV, BB, PP>), dim3(blocks), dim3(tb), 0, st, g->n_mols, \
                     n_deg, g->d_mol_runs, d_x, ldx, n_feat, lpr, gl, d_scale, d_shift, act,         \
                     d_out, ldo, d_arg, d_rawsum, vec_out)
  // rounds in flight: three (default; 179-181 us in the step) or two (GCMI_READOUT_DEPTH=2; 184-187 us)
  static const int depth_env = getenv("GCMI_READOUT_DEPTH") ? atoi(getenv("GCMI_READOUT_DEPTH")) : 3;
  if (x_bf16) {
    if (pre) {
      if (bn)
        hipLaunchKernelGGL((readout_fwd_kernel<4, true, true, 3, true>), dim3(blocks), dim3(tb), 0, st, g->n_mols, n_deg,
                           g->d_mol_runs, d_x, ldx, n_feat, lpr, gl, d_scale, d_shift, act, d_out, ldo, d_arg, d_rawsum,
                           vec_out);
      else
        hipLaunchKernelGGL((readout_fwd_kernel<4, false, true, 3, true>), dim3(blocks), dim3(tb), 0, st, g->n_mols, n_deg,
                           g->d_mol_runs, d_x, ldx, n_feat, lpr, gl, d_scale, d_shift, act, d_out, ldo, d_arg, d_rawsum,
                           vec_out);
    } else {
      if (bn)
        hipLaunchKernelGGL((readout_fwd_kernel<4, true, false, 2, true>), dim3(blocks), dim3(tb), 0, st, g->n_mols, n_deg,
                           g->d_mol_runs, d_x, ldx, n_feat, lpr, gl, d_scale, d_shift, act, d_out, ldo, d_arg, d_rawsum,
                           vec_out);
      else
        hipLaunchKernelGGL((readout_fwd_kernel<4, false, false, 2, true>), dim3(blocks), dim3(tb), 0, st, g->n_mols, n_deg,
                           g->d_mol_runs, d_x, ldx, n_feat, lpr, gl, d_scale, d_shift, act, d_out, ldo, d_arg, d_rawsum,
                           vec_out);
    }
  } else if (V == 4 && pre && depth_env != 2) {
    if (bn)
      hipLaunchKernelGGL((readout_fwd_kernel<4, true, true, 3>), dim3(blocks), dim3(tb), 0, st, g->n_mols, n_deg,
                         g->d_mol_runs, d_x, ldx, n_feat, lpr, gl, d_scale, d_shift, act, d_out, ldo, d_arg, d_rawsum,
                         vec_out);
    else
      hipLaunchKernelGGL((readout_fwd_kernel<4, false, true, 3>), dim3(blocks), dim3(tb), 0, st, g->n_mols, n_deg,
                         g->d_mol_runs, d_x, ldx, n_feat, lpr, gl, d_scale, d_shift, act, d_out, ldo, d_arg, d_rawsum,
                         vec_out);
  } else if (V == 4 && pre) {
    if (bn) LAUNCH_RO(4, true, true); else LAUNCH_RO(4, false, true);
  } else if (V == 4) {
    if (bn) LAUNCH_RO(4, true, false); else LAUNCH_RO(4, false, false);
  } else {
    if (bn) LAUNCH_RO(1, true, false); else LAUNCH_RO(1, false, false);
  }
#undef LAUNCH_RO
  GCMI_CHECK_LAUNCH("readout_fwd");
  return GCMI_OK;
}
}  // namespace gcmi

extern "C" {

int gcmi_readout_bwd(const gcmi_graph* g, const float* d_dout, int64_t lddo, const float* d_out,
                     int64_t ldo, int32_t n_feat, int32_t act, const int32_t* d_arg,
                     float* d_dx, int64_t lddx, void* stream) {
  int rc = check_graph(g, false);
  if (rc) return rc;
  GCMI_CHECK_ARG(n_feat > 0 && lddo >= 2 * (int64_t)n_feat && lddx >= n_feat, "readout_bwd: bad n_feat/ld");
  GCMI_CHECK_ARG(act == 0 || (d_out && ldo >= 2 * (int64_t)n_feat), "readout_bwd: saved output needed for tanh");
  if (g->n_atoms == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_dout && d_arg && d_dx && g->d_membership, "readout_bwd: NULL buffer");
  hipStream_t st = (hipStream_t)stream;
  const int V = (vec_width(d_dx, lddx, n_feat) == 4 && vec_width(d_dout, lddo, n_feat) == 4 &&
                 (act == 0 || vec_width(d_out, ldo, n_feat) == 4) && aligned16(d_arg))
                    ? 4
                    : 1;
  const int lpr = n_feat / V;
  const int64_t slots = (int64_t)g->n_atoms * lpr;
  if (V == 4)
    hipLaunchKernelGGL(readout_bwd_kernel<4>, dim3(grid_for(slots, kRBlock)), dim3(kRBlock), 0, st,
                       slots, lpr, n_feat, g->d_membership, d_dout, lddo, d_out, ldo, act, d_arg,
                       d_dx, lddx);
  else
    hipLaunchKernelGGL(readout_bwd_kernel<1>, dim3(grid_for(slots, kRBlock)), dim3(kRBlock), 0, st,
                       slots, lpr, n_feat, g->d_membership, d_dout, lddo, d_out, ldo, act, d_arg,
                       d_dx, lddx);
  GCMI_CHECK_LAUNCH("readout_bwd");
  return GCMI_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- atom codes -> feature rows
// The 75 columns of atom_features (deepchem/feat/graph_features.py:282-391) are five one-hot blocks, two small
// integers and a flag: eight bytes per atom (layout: deepchem_amd/feat/atom_codes.py).  Collation and the H2D copy
// move the codes; this kernel writes the float rows the model reads (columns >= 75 up to ldo are zero).
namespace gcmi {
__global__ void __launch_bounds__(256)
expand_codes_kernel(const uint8_t* __restrict__ codes, int64_t ldc, int64_t n_atoms, float* __restrict__ out,
                    int64_t ldo, int quads) {
  // grid-stride: grid_for caps the grid, a large batch has more quads than one sweep of it covers
  for (int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; slot < n_atoms * quads;
       slot += (int64_t)gridDim.x * blockDim.x) {
  const int64_t r = slot / quads;
  const int q = (int)(slot - r * quads);
  const uint2 w = *reinterpret_cast<const uint2*>(codes + r * ldc);
  const int sym = w.x & 255, deg = (w.x >> 8) & 255, imp = (w.x >> 16) & 255;
  const int chg = (int)(int8_t)(w.x >> 24);
  const int rad = w.y & 255, hyb = (w.y >> 8) & 255, aro = (w.y >> 16) & 255, toth = (w.y >> 24) & 255;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * q + e;
    float x = 0.f;
    if (c < 44) x = c == (sym < 43 ? sym : 43) ? 1.f : 0.f;
    else if (c < 55) x = (c - 44) == (deg < 10 ? deg : 10) ? 1.f : 0.f;
    else if (c < 62) x = (c - 55) == (imp < 6 ? imp : 6) ? 1.f : 0.f;
    else if (c == 62) x = (float)chg;
    else if (c == 63) x = (float)rad;
    else if (c < 69) x = (c - 64) == (hyb < 4 ? hyb : 4) ? 1.f : 0.f;
    else if (c == 69) x = (float)aro;
    else if (c < 75) x = (c - 70) == (toth < 4 ? toth : 4) ? 1.f : 0.f;
    v[e] = x;
  }
  *reinterpret_cast<float4*>(out + r * ldo + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
  }
}
}  // namespace gcmi

extern "C" int gcmi_expand_atom_codes(const uint8_t* d_codes, int64_t ldc, int64_t n_atoms, float* d_out, int64_t ldo,
                                      void* stream) {
  using namespace gcmi;
  GCMI_CHECK_ARG(n_atoms >= 0 && ldc >= 8 && ldc % 8 == 0 && ldo >= 76 && ldo % 4 == 0,
                 "expand_atom_codes: bad shape (8-byte aligned code rows, ldo >= 76 and a multiple of 4)");
  if (n_atoms == 0) return GCMI_OK;
  GCMI_CHECK_ARG(d_codes && d_out && aligned16(d_out) && (reinterpret_cast<uintptr_t>(d_codes) & 7u) == 0,
                 "expand_atom_codes: NULL or misaligned buffer");
  const int quads = (int)(ldo / 4);
  hipLaunchKernelGGL(expand_codes_kernel, dim3(grid_for(n_atoms * quads, 256)), dim3(256), 0, (hipStream_t)stream,
                     d_codes, ldc, n_atoms, d_out, ldo, quads);
  GCMI_CHECK_LAUNCH("expand_atom_codes");
  return GCMI_OK;
}
