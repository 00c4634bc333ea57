// Host-side batch collation: ConvMol.agglomerate_mols
// (deepchem/feat/mol_graphs.py:256-349) over a packed molecule set, written
// straight into the (pinned) staging arena the H2D copy reads, plus the plans the
// kernels use (per-molecule row runs, reverse edge slots, LDS windows).
//
// The reference sorts a concatenated degree vector and re-indexes neighbour
// lists molecule by molecule in Python (~2.5 ms per 100 molecules).  Because
// the batch order is (degree, batch position, atom id inside the molecule) and
// the atoms of a molecule only need a stable order inside each degree, no sort
// is needed at all: a (parallel) histogram pass gives every molecule's share of
// every degree block, a serial prefix over molecules turns that into per-molecule
// row bases (and closes the LDS windows), and a second parallel pass writes rows.
// The result is identical for any thread count.
#include <algorithm>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace {

constexpr int ND = GCMI_MAX_DEG + 1;

template <typename F>
void parallel_for(int64_t n, int n_threads, F&& fn) {
  if (n_threads <= 1 || n < 512) {
    fn(0, n, 0);
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t) th.emplace_back(fn, n * t / n_threads, n * (t + 1) / n_threads, t);
  for (auto& x : th) x.join();
}


struct Win {
  int32_t begin[ND], sb[ND + 1], eb[ND];  // first row, slot prefix, edge-entry prefix per degree
  int64_t eoff;                           // first entry in the window-major edge array
};

// What the serial pass over a batch's per-molecule degree histograms decides: where every molecule's rows start in
// every degree block, which LDS window it belongs to, and the window descriptors.
struct BatchPlan {
  std::vector<int32_t> base, mol_win;
  bool have_counts = false;        // deg_count_in / max_mol_in already summed by the caller's histogram pass
  int64_t deg_count_in[ND] = {0}, max_mol_in = 0;
  int32_t* base_ext = nullptr;     // when set, row bases / window ids are written here instead of the vectors
  int32_t* mol_win_ext = nullptr;
  std::vector<Win> wins;
  int64_t deg_start[ND + 1], edge_start[ND + 1];
  int64_t n_atoms = 0, n_edges = 0, win_entries = 0;
  bool want_win = false;
  int32_t n_win_big = 0, win_alloc = 0, win_ecap = 0, win_alloc_big = 0, win_ecap_big = 0;
  int32_t n_mols_with_atoms = 0;  // -> gcmi_graph.win_reserved[0]: every molecule of a window pass is one of these
};

// hist: n_sel x ND degree histograms in batch order.  out_win_meta (n_win descriptors) is written when windows are
// wanted; out_win_edges (may be NULL) only gets the zero padding at the end of every window's entries.
int plan_serial(const int32_t* hist_p, int64_t n_sel, int n_deg, int32_t win_cap, bool want_win, int64_t cap_atoms,
                int64_t cap_edges, int32_t* out_win_meta, uint16_t* out_win_edges, BatchPlan& P) {
  struct HistView {
    const int32_t* p;
    int32_t operator[](size_t i) const { return p[i]; }
  } hist{hist_p};
  int32_t win_alloc = 0, win_ecap = 0, win_alloc_big = 0, win_ecap_big = 0, n_win_big = 0;
  // serial prefix over molecules: degree-block starts, per-molecule row bases, windows
  // (rows of hist are zero from n_deg on, so the loops below run over all ND columns: fixed trip counts)
  int64_t deg_count[ND] = {0};
  int64_t max_mol = 0;
  if (P.have_counts) {
    for (int d = 0; d < ND; ++d) deg_count[d] = P.deg_count_in[d];
    max_mol = P.max_mol_in;
  } else {
    for (int64_t p = 0; p < n_sel; ++p) {
      int64_t sz = 0;
      for (int d = 0; d < ND; ++d) {
        deg_count[d] += hist[(size_t)p * ND + d];
        sz += hist[(size_t)p * ND + d];
      }
      max_mol = std::max(max_mol, sz);
    }
  }
  int64_t* deg_start = P.deg_start;
  int64_t* edge_start = P.edge_start;
  deg_start[0] = 0;
  edge_start[0] = 0;
  for (int d = 0; d < ND; ++d) {
    const int64_t nd = d < n_deg ? deg_count[d] : 0;
    deg_start[d + 1] = deg_start[d] + nd;
    edge_start[d + 1] = edge_start[d] + nd * d;
  }
  const int64_t n_atoms = deg_start[n_deg], n_edges = edge_start[n_deg];
  GCMI_CHECK_ARG(n_atoms < (1LL << 31) && n_edges < (1LL << 31), "collate: batch too large for int32 rows");
  GCMI_CHECK_ARG(n_atoms <= cap_atoms && n_edges <= cap_edges,
                 "collate: capacity (%lld atoms, %lld edges) < needed (%lld, %lld)",
                 (long long)cap_atoms, (long long)cap_edges, (long long)n_atoms, (long long)n_edges);
  // a molecule larger than win_cap gets a window of its own; slots are 12-bit in the edge entries
  if (want_win && max_mol > GCMI_WIN_MAX_SLOTS) want_win = false;
  if (!P.base_ext) P.base.assign((size_t)n_sel * ND, 0);
  if (!P.mol_win_ext) P.mol_win.assign(want_win ? (size_t)n_sel : 0, 0);
  int32_t* base = P.base_ext ? P.base_ext : P.base.data();  // first row of molecule p inside degree block d
  int32_t* mol_win = P.mol_win_ext ? P.mol_win_ext : P.mol_win.data();
  std::vector<Win>& wins = P.wins;
  wins.clear();
  {
    int64_t cursor[ND];
    for (int d = 0; d < ND; ++d) cursor[d] = deg_start[d];
    int64_t in_win = 0;
    int32_t with_atoms = 0;
    for (int64_t p = 0; p < n_sel; ++p) {
      const int32_t* hp = hist_p + (size_t)p * ND;
      int64_t sz = 0;
      for (int d = 0; d < ND; ++d) sz += hp[d];
      with_atoms += sz > 0 ? 1 : 0;
      if (want_win) {
        if (p == 0 || in_win + sz > win_cap) {  // open a new window at molecule p
          Win w;
          for (int d = 0; d < ND; ++d) {
            w.begin[d] = (int32_t)cursor[d < n_deg ? d : n_deg - 1];
            w.sb[d] = 0;  // holds the row COUNT until the window is closed
          }
          w.sb[ND] = 0;
          wins.push_back(w);
          in_win = 0;
        }
        in_win += sz;
        mol_win[(size_t)p] = (int32_t)wins.size() - 1;
        int32_t* sb = wins.back().sb;
        for (int d = 0; d < ND; ++d) sb[d] += hp[d];
      }
      int32_t* bp = base + (size_t)p * ND;
      for (int d = 0; d < ND; ++d) {
        bp[d] = (int32_t)cursor[d];
        cursor[d] += hp[d];
      }
      for (int d = n_deg; d < ND; ++d) bp[d] = 0;
    }
    P.n_mols_with_atoms = with_atoms;
    // close the windows: counts -> prefixes, edge offsets (every window padded to 8 entries = 16 B).
    // Descriptors are emitted with the ordinary windows first and the oversized ones (a single
    // molecule above win_cap) last, so the kernels can give the two classes different LDS shapes.
    int64_t eoff = 0;
    for (auto& W : wins) {
      int32_t acc = 0;
      for (int d = 0; d < ND; ++d) acc += W.sb[d];
      if (acc > win_cap) ++n_win_big;
    }
    size_t pos_norm = 0, pos_big = wins.size() - (size_t)n_win_big;
    for (size_t w = 0; w < wins.size(); ++w) {
      Win& W = wins[w];
      int32_t acc = 0, eacc = 0;
      for (int d = 0; d < ND; ++d) {
        const int32_t c = W.sb[d];
        W.sb[d] = acc;
        W.eb[d] = eacc;
        acc += c;
        eacc += c * d;
      }
      W.sb[ND] = acc;
      W.eoff = eoff;
      const int32_t padded = (eacc + 7) / 8 * 8;
      eoff += padded;
      const bool big = acc > win_cap;
      if (big) {
        win_alloc_big = std::max(win_alloc_big, acc);
        win_ecap_big = std::max(win_ecap_big, padded);
      } else {
        win_alloc = std::max(win_alloc, acc);
        win_ecap = std::max(win_ecap, padded);
      }
      int32_t* m = out_win_meta + (big ? pos_big++ : pos_norm++) * GCMI_WIN_META_INTS;
      for (int d = 0; d < ND; ++d) m[d] = W.begin[d] - W.sb[d];
      for (int d = 1; d <= ND; ++d) m[ND - 1 + d] = W.sb[d];
      m[2 * ND] = (int32_t)W.eoff;
      m[2 * ND + 1] = eacc;
      if (out_win_edges)
        for (int32_t q = eacc; q < padded; ++q) out_win_edges[W.eoff + q] = 0;
    }
    P.win_entries = eoff;
  }
  P.n_atoms = n_atoms;
  P.n_edges = n_edges;
  P.want_win = want_win;
  P.n_win_big = n_win_big;
  P.n_mols_with_atoms = P.wins.empty() ? 0 : P.n_mols_with_atoms;
  P.win_alloc = win_alloc;
  P.win_ecap = win_ecap;
  P.win_alloc_big = win_alloc_big;
  P.win_ecap_big = win_ecap_big;
  return GCMI_OK;
}

}  // namespace

extern "C" {

int gcmi_collate_sizes(const int64_t* atom_ptr, const int64_t* adj_ptr, const int64_t* sel,
                       int64_t n_sel, int64_t* out_n_atoms, int64_t* out_n_edges) {
  GCMI_CHECK_ARG(atom_ptr && adj_ptr && (sel || n_sel == 0) && out_n_atoms && out_n_edges,
                 "collate_sizes: NULL argument");
  int64_t na = 0, ne = 0;
  for (int64_t p = 0; p < n_sel; ++p) {
    const int64_t m = sel[p];
    GCMI_CHECK_ARG(m >= 0, "collate_sizes: negative molecule index");
    const int64_t a0 = atom_ptr[m], a1 = atom_ptr[m + 1];
    na += a1 - a0;
    ne += adj_ptr[a1] - adj_ptr[a0];
  }
  *out_n_atoms = na;
  *out_n_edges = ne;
  return GCMI_OK;
}

int gcmi_collate_plans(const float* atom_features, int64_t n_feat, const int64_t* atom_ptr,
                       const int64_t* adj_ptr, const int32_t* adj_idx, const int64_t* sel,
                       int64_t n_sel, int32_t max_deg, float* out_features, int64_t out_ld,
                       int64_t cap_atoms, int32_t* out_membership, int32_t* out_col_idx,
                       int64_t cap_edges, int32_t* out_mol_runs, uint8_t* out_rev_pos,
                       int32_t* out_symmetric, int32_t win_cap, int32_t* out_win_meta,
                       uint16_t* out_win_edges, gcmi_graph* graph) {
  GCMI_CHECK_ARG(atom_features && atom_ptr && adj_ptr && (sel || n_sel == 0) && graph,
                 "collate: NULL input");
  GCMI_CHECK_ARG(n_feat > 0 && out_ld >= n_feat, "collate: out_ld %lld < n_feat %lld",
                 (long long)out_ld, (long long)n_feat);
  GCMI_CHECK_ARG(max_deg >= 0 && max_deg <= GCMI_MAX_DEG, "collate: max_deg outside [0,%d]",
                 GCMI_MAX_DEG);
  GCMI_CHECK_ARG(n_sel >= 0 && n_sel < (1LL << 31), "collate: bad n_sel");
  bool want_win = win_cap > 0 && out_win_meta && out_win_edges;
  const int n_deg = max_deg + 1;
  // GCMI_COLLATE_THREADS: worker threads per call (default: up to 16, one per 256 molecules)
  static const unsigned thread_cap = getenv("GCMI_COLLATE_THREADS") ? (unsigned)std::max(1, atoi(getenv("GCMI_COLLATE_THREADS"))) : 16u;
  const int n_threads = (int)std::min<int64_t>(
      std::max<int64_t>(1, n_sel / 256),
      std::min<unsigned>(thread_cap, std::max(1u, std::thread::hardware_concurrency())));

  // pass 1 (parallel): degree histogram of every molecule
  std::vector<int32_t> hist((size_t)n_sel * ND, 0);
  std::vector<int> bad(std::max(1, n_threads), 0);
  parallel_for(n_sel, n_threads, [&](int64_t p0, int64_t p1, int t) {
    for (int64_t p = p0; p < p1; ++p) {
      const int64_t m = sel[p];
      int32_t* h = hist.data() + (size_t)p * ND;
      for (int64_t a = atom_ptr[m]; a < atom_ptr[m + 1]; ++a) {
        const int64_t d = adj_ptr[a + 1] - adj_ptr[a];
        if (d < 0 || d > max_deg) {
          bad[t] = 1;
          continue;
        }
        h[d]++;
      }
    }
  });
  for (int b : bad) GCMI_CHECK_ARG(!b, "collate: an atom has more than max_deg=%d neighbours", max_deg);

  // serial prefix over molecules: degree-block starts, per-molecule row bases, windows
  BatchPlan P;
  {
    const int rc = plan_serial(hist.data(), n_sel, n_deg, win_cap, want_win, cap_atoms, cap_edges, out_win_meta,
                               out_win_edges, P);
    if (rc != GCMI_OK) return rc;
  }
  GCMI_CHECK_ARG(P.n_atoms == 0 || (out_features && out_membership), "collate: NULL output");
  GCMI_CHECK_ARG(P.n_edges == 0 || (out_col_idx && adj_idx), "collate: NULL edge buffers");
  want_win = P.want_win;
  const int32_t* base = P.base.data();
  const int32_t* mol_win = P.mol_win.data();
  const std::vector<Win>& wins = P.wins;
  const int64_t* deg_start = P.deg_start;
  const int64_t* edge_start = P.edge_start;
  const int64_t n_atoms = P.n_atoms, n_edges = P.n_edges;
  const int32_t n_win_big = P.n_win_big, win_alloc = P.win_alloc, win_ecap = P.win_ecap,
                win_alloc_big = P.win_alloc_big, win_ecap_big = P.win_ecap_big;

  // pass 2 (parallel): rows, features, membership, neighbour tables, runs, reverse slots, LDS slots
  std::vector<int> bad2(std::max(1, n_threads), 0);
  std::vector<int> asym(std::max(1, n_threads), 0);
  parallel_for(n_sel, n_threads, [&](int64_t p0, int64_t p1, int t) {
    std::vector<int32_t> new_row;
    std::vector<int32_t> deg_of;
    for (int64_t p = p0; p < p1; ++p) {
      const int64_t m = sel[p];
      const int64_t a0 = atom_ptr[m], a1 = atom_ptr[m + 1];
      const int64_t n = a1 - a0;
      new_row.resize((size_t)n);
      deg_of.resize((size_t)n);
      int32_t cur[ND];
      for (int d = 0; d < ND; ++d) cur[d] = d < n_deg ? base[(size_t)p * ND + d] : 0;
      if (out_mol_runs) {
        int32_t* runs = out_mol_runs + p * n_deg * 2;
        for (int d = 0; d < n_deg; ++d) {
          const int32_t c = hist[(size_t)p * ND + d];
          runs[2 * d] = c ? cur[d] : 0;
          runs[2 * d + 1] = c ? cur[d] + c : 0;
        }
      }
      for (int64_t a = a0; a < a1; ++a) {
        const int d = (int)(adj_ptr[a + 1] - adj_ptr[a]);
        deg_of[(size_t)(a - a0)] = d;
        new_row[(size_t)(a - a0)] = cur[d]++;
      }
      const Win* W = want_win ? &wins[(size_t)mol_win[(size_t)p]] : nullptr;
      for (int64_t a = a0; a < a1; ++a) {
        const int64_t e0 = adj_ptr[a];
        const int d = deg_of[(size_t)(a - a0)];
        const int64_t row = new_row[(size_t)(a - a0)];
        float* dst = out_features + row * out_ld;
        memcpy(dst, atom_features + a * n_feat, sizeof(float) * (size_t)n_feat);
        for (int64_t f = n_feat; f < out_ld; ++f) dst[f] = 0.f;
        out_membership[row] = (int32_t)p;
        const int64_t eb = edge_start[d] + (row - deg_start[d]) * d;
        for (int j = 0; j < d; ++j) {
          const int64_t nb = adj_idx[e0 + j];
          if (nb < 0 || nb >= n) {
            bad2[t] = 1;
            out_col_idx[eb + j] = (int32_t)row;
            if (W) out_win_edges[W->eoff + W->eb[d] + (row - W->begin[d]) * d + j] = 0;
            if (out_rev_pos) out_rev_pos[eb + j] = 255;
            continue;
          }
          const int32_t nrow = new_row[(size_t)nb];
          out_col_idx[eb + j] = nrow;
          int found = 15;
          if (out_rev_pos || W) {
            // the n-th slot of `a` that points at nb pairs with the n-th slot of nb pointing at `a`
            int nth = 0;
            for (int q = 0; q < j; ++q) nth += adj_idx[e0 + q] == nb ? 1 : 0;
            const int64_t f0 = adj_ptr[a0 + nb], f1 = adj_ptr[a0 + nb + 1];
            found = -1;
            for (int64_t q = f0; q < f1; ++q) {
              if (adj_idx[q] == (int32_t)(a - a0)) {
                if (nth == 0) {
                  found = (int)(q - f0);
                  break;
                }
                --nth;
              }
            }
            if (found < 0) {
              asym[t] = 1;
              found = 15;
            }
            if (out_rev_pos) out_rev_pos[eb + j] = found == 15 ? (uint8_t)255 : (uint8_t)found;
          }
          if (W) {
            const int dn = deg_of[(size_t)nb];
            const int32_t slot = W->sb[dn] + (nrow - W->begin[dn]);
            out_win_edges[W->eoff + W->eb[d] + (row - W->begin[d]) * d + j] =
                (uint16_t)(slot | (found << GCMI_WIN_SLOT_BITS));
          }
        }
      }
    }
  });
  for (int b : bad2) GCMI_CHECK_ARG(!b, "collate: a neighbour id is outside its molecule");
  if (out_symmetric) {
    *out_symmetric = 1;
    for (int b : asym)
      if (b) *out_symmetric = 0;
  }
  graph->n_atoms = (int32_t)n_atoms;
  graph->n_edges = (int32_t)n_edges;
  graph->n_mols = (int32_t)n_sel;
  graph->max_deg = max_deg;
  for (int d = 0; d < GCMI_MAX_DEG + 2; ++d) {
    graph->deg_start[d] = (int32_t)deg_start[d <= n_deg ? d : n_deg];
    graph->edge_start[d] = (int32_t)edge_start[d <= n_deg ? d : n_deg];
  }
  graph->d_col_idx = nullptr;
  graph->d_membership = nullptr;
  graph->d_mol_runs = nullptr;
  graph->d_rev_pos = nullptr;
  graph->n_win = want_win ? (int32_t)wins.size() : 0;
  graph->n_win_big = want_win ? n_win_big : 0;
  graph->win_alloc = want_win ? win_alloc : 0;
  graph->win_ecap = want_win ? win_ecap : 0;
  graph->win_alloc_big = want_win ? win_alloc_big : 0;
  graph->win_ecap_big = want_win ? win_ecap_big : 0;
  graph->win_reserved[0] = want_win ? P.n_mols_with_atoms : 0;  // molecules the windows cover (readout over windows)
  graph->win_reserved[1] = 0;
  graph->d_win_meta = nullptr;
  graph->d_win_edges = nullptr;
  return GCMI_OK;
}

int gcmi_collate(const float* atom_features, int64_t n_feat, const int64_t* atom_ptr,
                 const int64_t* adj_ptr, const int32_t* adj_idx, const int64_t* sel,
                 int64_t n_sel, int32_t max_deg, float* out_features, int64_t out_ld,
                 int64_t cap_atoms, int32_t* out_membership, int32_t* out_col_idx,
                 int64_t cap_edges, int32_t* out_mol_runs, gcmi_graph* graph) {
  return gcmi_collate_plans(atom_features, n_feat, atom_ptr, adj_ptr, adj_idx, sel, n_sel, max_deg,
                            out_features, out_ld, cap_atoms, out_membership, out_col_idx, cap_edges,
                            out_mol_runs, nullptr, nullptr, 0, nullptr, nullptr, graph);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Collation on the device over a molecule set resident in HBM.
//
// Everything gcmi_collate_plans looks up inside a molecule does not depend on the batch: an atom's degree, its rank
// among the atoms of the same degree in its molecule (its row = the molecule's base row in that degree block + rank),
// and the reverse slot of every neighbour entry.  gcmi_molset_tables computes those once per set (host); the set then
// lives in HBM.  Per batch the host only does the serial pass over per-molecule histograms (gcmi_collate_plan:
// row bases, windows; a few MB instead of the batch's whole arena cross PCIe) and one kernel with a thread per atom
// writes the same arena gcmi_collate_plans writes, byte for byte.
// ---------------------------------------------------------------------------------------------------------------
namespace gcmi {
namespace collate_dev {  // named, so that the kernels show up under their names in rocprofv3 traces

constexpr int WD = GCMI_COLLATE_WIN_DESC_INTS;  // begin[ND] | sb[ND] | eb[ND] | eoff | 2 spare

struct RowsArgs {
  const float* feat;
  int64_t n_feat;
  const int64_t* adj_ptr;
  const int32_t* adj_idx;
  const int32_t* rank;
  const uint8_t* rev;
  const int32_t* base;
  const int32_t* mol_win;
  const int32_t* atom_off;
  const int64_t* atom0;
  const int32_t* desc;
  float* out_feat;
  int64_t out_ld;
  int32_t* membership;
  int32_t* col_idx;
  int32_t* mol_runs;
  uint8_t* rev_pos;
  uint16_t* win_edges;
  int64_t* src_atom;  // when set: row -> atom of the set, and the rows are copied by collate_features_kernel
  int32_t deg_start[ND + 1], edge_start[ND + 1];
  int32_t n_sel, n_atoms, n_deg, want_win;
};

// batch atom i (atoms counted molecule by molecule in batch order) -> its row and everything stored per row / per edge
__host__ __device__ inline void collate_atom(const RowsArgs& A, int32_t i) {
  int32_t lo = 0, hi = A.n_sel - 1;
  while (lo < hi) {  // the last molecule whose first atom is <= i (molecules without atoms share an offset)
    const int32_t mid = (lo + hi + 1) >> 1;
    if (A.atom_off[mid] <= i) lo = mid;
    else hi = mid - 1;
  }
  const int32_t p = lo;
  const int64_t a0 = A.atom0[p];
  const int64_t a = a0 + (i - A.atom_off[p]);
  const int64_t e0 = A.adj_ptr[a];
  const int d = (int)(A.adj_ptr[a + 1] - e0);
  const int32_t* base = A.base + (int64_t)p * ND;
  const int32_t row = base[d] + A.rank[a];
  if (A.src_atom) {
    A.src_atom[row] = a;
  } else {
    float* dst = A.out_feat + (int64_t)row * A.out_ld;
    const float* src = A.feat + a * A.n_feat;
    for (int64_t f = 0; f < A.n_feat; ++f) dst[f] = src[f];
    for (int64_t f = A.n_feat; f < A.out_ld; ++f) dst[f] = 0.f;
  }
  A.membership[row] = p;
  const int64_t eb = (int64_t)A.edge_start[d] + (int64_t)(row - A.deg_start[d]) * d;
  const int32_t* W = A.want_win ? A.desc + (int64_t)A.mol_win[p] * WD : nullptr;
  for (int j = 0; j < d; ++j) {
    const int64_t an = a0 + A.adj_idx[e0 + j];
    const int dn = (int)(A.adj_ptr[an + 1] - A.adj_ptr[an]);
    const int32_t nrow = base[dn] + A.rank[an];
    const int found = A.rev[e0 + j];
    A.col_idx[eb + j] = nrow;
    if (A.rev_pos) A.rev_pos[eb + j] = found == 15 ? (uint8_t)255 : (uint8_t)found;
    if (W) {
      const int32_t slot = W[ND + dn] + (nrow - W[dn]);
      A.win_edges[(int64_t)W[3 * ND] + W[2 * ND + d] + (int64_t)(row - W[d]) * d + j] =
          (uint16_t)(slot | (found << GCMI_WIN_SLOT_BITS));
    }
  }
}

// per-molecule row ranges of the readout: thread per (molecule, degree)
__host__ __device__ inline void collate_run(const RowsArgs& A, int32_t k) {
  const int32_t p = k / A.n_deg, d = k - p * A.n_deg;
  const int32_t b = A.base[(int64_t)p * ND + d];
  const int32_t e = p + 1 < A.n_sel ? A.base[(int64_t)(p + 1) * ND + d] : A.deg_start[d + 1];
  A.mol_runs[2 * (int64_t)k] = e > b ? b : 0;
  A.mol_runs[2 * (int64_t)k + 1] = e > b ? e : 0;
}

__global__ void __launch_bounds__(256) collate_rows_kernel(RowsArgs A) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < A.n_atoms) collate_atom(A, (int32_t)i);
}

// wide feature rows: consecutive lanes copy consecutive columns of a row (the per-atom thread only noted its source)
__global__ void __launch_bounds__(256) collate_features_kernel(RowsArgs A) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)A.n_atoms * A.out_ld) return;
  const int64_t row = t / A.out_ld, c = t - row * A.out_ld;
  A.out_feat[t] = c < A.n_feat ? A.feat[A.src_atom[row] * A.n_feat + c] : 0.f;
}

// the same with one 16-byte store per lane (out_ld % 4 == 0, 16-byte aligned output; the source rows of n_feat floats
// are only 4-byte aligned, so they are read as scalars): quads = out_ld / 4 lanes per row, 32-bit index arithmetic
__global__ void __launch_bounds__(256) collate_features_quad_kernel(RowsArgs A, uint32_t quads, uint32_t n_quads) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= n_quads) return;
  const uint32_t row = t / quads, c = (t - row * quads) * 4u;
  const float* src = A.feat + A.src_atom[row] * A.n_feat;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 v;
  v.x = c + 0 < (uint32_t)A.n_feat ? src[c + 0] : 0.f;
  v.y = c + 1 < (uint32_t)A.n_feat ? src[c + 1] : 0.f;
  v.z = c + 2 < (uint32_t)A.n_feat ? src[c + 2] : 0.f;
  v.w = c + 3 < (uint32_t)A.n_feat ? src[c + 3] : 0.f;
  *reinterpret_cast<f32x4*>(A.out_feat + (int64_t)row * A.out_ld + c) = v;
}

__global__ void __launch_bounds__(256) collate_runs_kernel(RowsArgs A) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k < (int64_t)A.n_sel * A.n_deg) collate_run(A, (int32_t)k);
}

int fill_rows_args(RowsArgs& A, const void* features, int64_t n_feat, const int64_t* adj_ptr, const int32_t* adj_idx,
                   const int32_t* rank, const uint8_t* rev, const int32_t* staging, const int64_t* off,
                   const gcmi_graph* plan, float* out_features, int64_t out_ld, int32_t* membership, int32_t* col_idx,
                   int32_t* mol_runs, uint8_t* rev_pos, uint16_t* win_edges) {
  GCMI_CHECK_ARG(staging && off && plan, "collate_rows: NULL plan");
  GCMI_CHECK_ARG(n_feat > 0 && out_ld >= n_feat, "collate_rows: out_ld %lld < n_feat %lld", (long long)out_ld,
                 (long long)n_feat);
  GCMI_CHECK_ARG(plan->n_atoms == 0 || (features && adj_ptr && rank && out_features && membership),
                 "collate_rows: NULL atom buffers");
  GCMI_CHECK_ARG(plan->n_edges == 0 || (adj_idx && rev && col_idx), "collate_rows: NULL edge buffers");
  GCMI_CHECK_ARG(plan->n_win == 0 || win_edges, "collate_rows: windows planned but no entry buffer");
  GCMI_CHECK_ARG(plan->max_deg >= 0 && plan->max_deg <= GCMI_MAX_DEG, "collate_rows: bad plan");
  A.feat = (const float*)features;
  A.n_feat = n_feat;
  A.adj_ptr = adj_ptr;
  A.adj_idx = adj_idx;
  A.rank = rank;
  A.rev = rev;
  A.base = staging + off[0];
  A.mol_win = staging + off[1];
  A.atom_off = staging + off[2];
  A.atom0 = (const int64_t*)(staging + off[3]);
  A.desc = staging + off[5];
  A.out_feat = out_features;
  A.out_ld = out_ld;
  A.membership = membership;
  A.col_idx = col_idx;
  A.mol_runs = mol_runs;
  A.rev_pos = rev_pos;
  A.win_edges = win_edges;
  A.src_atom = nullptr;
  for (int d = 0; d <= ND; ++d) {
    A.deg_start[d] = plan->deg_start[d];
    A.edge_start[d] = plan->edge_start[d];
  }
  A.n_sel = plan->n_mols;
  A.n_atoms = plan->n_atoms;
  A.n_deg = plan->max_deg + 1;
  A.want_win = plan->n_win > 0;
  return GCMI_OK;
}

}  // namespace collate_dev
}  // namespace gcmi
using namespace gcmi::collate_dev;

extern "C" {

int gcmi_molset_tables(const int64_t* atom_ptr, const int64_t* adj_ptr, const int32_t* adj_idx, int64_t n_mols,
                       int32_t max_deg, int32_t* out_mol_hist, int32_t* out_rank, uint8_t* out_rev,
                       int32_t* out_symmetric, int32_t n_threads) {
  GCMI_CHECK_ARG(atom_ptr && adj_ptr && out_mol_hist && out_symmetric && n_mols >= 0, "molset_tables: NULL argument");
  GCMI_CHECK_ARG(max_deg >= 0 && max_deg <= GCMI_MAX_DEG, "molset_tables: max_deg outside [0,%d]", GCMI_MAX_DEG);
  const int64_t n_atoms_all = n_mols ? atom_ptr[n_mols] : 0;
  GCMI_CHECK_ARG(n_atoms_all == 0 || (out_rank && (adj_ptr[n_atoms_all] == 0 || (adj_idx && out_rev))),
                 "molset_tables: NULL table");
  const int nt = (int)std::min<int64_t>(std::max<int64_t>(1, n_mols / 256),
                                        std::max(1, std::min<int>(n_threads > 0 ? n_threads : 16,
                                                                  (int)std::max(1u, std::thread::hardware_concurrency()))));
  std::vector<int> bad_deg(nt, 0), bad_nb(nt, 0), asym(nt, 0);
  parallel_for(n_mols, nt, [&](int64_t m0, int64_t m1, int t) {
    for (int64_t m = m0; m < m1; ++m) {
      const int64_t a0 = atom_ptr[m], a1 = atom_ptr[m + 1], n = a1 - a0;
      int32_t* h = out_mol_hist + m * ND;
      for (int d = 0; d < ND; ++d) h[d] = 0;
      for (int64_t a = a0; a < a1; ++a) {
        const int64_t d = adj_ptr[a + 1] - adj_ptr[a];
        if (d < 0 || d > max_deg) {
          bad_deg[t] = 1;
          out_rank[a] = 0;
          continue;
        }
        out_rank[a] = h[d]++;
      }
      if (bad_deg[t]) continue;
      for (int64_t a = a0; a < a1; ++a) {
        const int64_t e0 = adj_ptr[a], e1 = adj_ptr[a + 1];
        for (int64_t e = e0; e < e1; ++e) {
          const int64_t nb = adj_idx[e];
          if (nb < 0 || nb >= n) {
            bad_nb[t] = 1;
            out_rev[e] = 15;
            continue;
          }
          // the n-th slot of `a` that points at nb pairs with the n-th slot of nb pointing at `a`
          int nth = 0;
          for (int64_t q = e0; q < e; ++q) nth += adj_idx[q] == nb ? 1 : 0;
          const int64_t f0 = adj_ptr[a0 + nb], f1 = adj_ptr[a0 + nb + 1];
          int found = -1;
          for (int64_t q = f0; q < f1; ++q) {
            if (adj_idx[q] == (int32_t)(a - a0)) {
              if (nth == 0) {
                found = (int)(q - f0);
                break;
              }
              --nth;
            }
          }
          if (found < 0) {
            asym[t] = 1;
            found = 15;
          }
          out_rev[e] = (uint8_t)found;
        }
      }
    }
  });
  for (int b : bad_deg) GCMI_CHECK_ARG(!b, "collate: an atom has more than max_deg=%d neighbours", max_deg);
  for (int b : bad_nb) GCMI_CHECK_ARG(!b, "collate: a neighbour id is outside its molecule");
  *out_symmetric = 1;
  for (int b : asym)
    if (b) *out_symmetric = 0;
  return GCMI_OK;
}

int64_t gcmi_collate_plan_words(int64_t n_sel) {
  if (n_sel < 0) return 0;
  // base | mol_win | atom_off | (8-byte aligned) atom0 | win_meta | window descriptors, one window per molecule at most
  return n_sel * ND + n_sel + (n_sel + 1) + 1 + 2 * n_sel + n_sel * (GCMI_WIN_META_INTS + WD) + 8;
}

int gcmi_collate_plan(const int32_t* mol_hist, const int64_t* atom_ptr, const int64_t* sel, int64_t n_sel,
                      int32_t max_deg, int32_t win_cap, int32_t* staging, int64_t staging_words, int64_t* out_offsets,
                      gcmi_graph* graph) {
  GCMI_CHECK_ARG(mol_hist && atom_ptr && (sel || n_sel == 0) && staging && out_offsets && graph,
                 "collate_plan: NULL argument");
  GCMI_CHECK_ARG(max_deg >= 0 && max_deg <= GCMI_MAX_DEG, "collate_plan: max_deg outside [0,%d]", GCMI_MAX_DEG);
  GCMI_CHECK_ARG(n_sel >= 0 && n_sel < (1LL << 31), "collate_plan: bad n_sel");
  GCMI_CHECK_ARG(staging_words >= gcmi_collate_plan_words(n_sel), "collate_plan: staging holds %lld words, %lld needed",
                 (long long)staging_words, (long long)gcmi_collate_plan_words(n_sel));
  const int n_deg = max_deg + 1;
  const int64_t o_base = 0, o_win = n_sel * ND, o_off = o_win + n_sel;
  const int64_t o_a0 = (o_off + n_sel + 1 + 1) / 2 * 2, o_meta = o_a0 + 2 * n_sel;
  std::vector<int32_t> hist((size_t)n_sel * ND);
  std::vector<int32_t> size_of((size_t)n_sel);
  int32_t* atom_off = staging + o_off;
  int64_t* atom0 = reinterpret_cast<int64_t*>(staging + o_a0);
  // the histograms of the selected molecules are scattered over the set's table (44 bytes each, one cache miss per
  // molecule): gathered by a few threads with the next rows prefetched; everything after this is sequential
  const int nt = (int)std::min<int64_t>(std::max<int64_t>(1, n_sel / 8192),
                                        std::min(4u, std::max(1u, std::thread::hardware_concurrency())));
  std::vector<int> bad(nt, 0);
  std::vector<int64_t> part((size_t)nt * (ND + 1), 0);  // per thread: atoms per degree, largest molecule
  parallel_for(n_sel, nt, [&](int64_t p0, int64_t p1, int t) {
    int64_t dc[ND] = {0}, mx = 0;
    for (int64_t p = p0; p < p1; ++p) {
      const int64_t m = sel[p];
      if (m < 0) {
        bad[t] = 1;
        continue;
      }
      if (p + 12 < p1 && sel[p + 12] >= 0) {
        __builtin_prefetch(mol_hist + sel[p + 12] * ND);
        __builtin_prefetch(atom_ptr + sel[p + 12]);
      }
      const int32_t* h = mol_hist + m * ND;
      int32_t sz = 0;
      for (int d = 0; d < ND; ++d) {
        hist[(size_t)p * ND + d] = h[d];
        sz += h[d];
        dc[d] += h[d];
      }
      mx = std::max<int64_t>(mx, sz);
      if (sz != atom_ptr[m + 1] - atom_ptr[m]) bad[t] = 2;
      size_of[(size_t)p] = sz;
      atom0[p] = atom_ptr[m];
    }
    for (int d = 0; d < ND; ++d) part[(size_t)t * (ND + 1) + d] = dc[d];
    part[(size_t)t * (ND + 1) + ND] = mx;
  });
  for (int b : bad) {
    GCMI_CHECK_ARG(b != 1, "collate_plan: negative molecule index");
    GCMI_CHECK_ARG(b != 2, "collate_plan: a molecule's histogram does not match its atoms");
  }
  int64_t acc = 0;
  for (int64_t p = 0; p < n_sel; ++p) {
    atom_off[p] = (int32_t)acc;
    acc += size_of[(size_t)p];
    GCMI_CHECK_ARG(acc < (1LL << 31), "collate: batch too large for int32 rows");
  }
  atom_off[n_sel] = (int32_t)acc;
  BatchPlan P;
  P.have_counts = true;
  for (int t = 0; t < nt; ++t) {
    for (int d = 0; d < ND; ++d) P.deg_count_in[d] += part[(size_t)t * (ND + 1) + d];
    P.max_mol_in = std::max(P.max_mol_in, part[(size_t)t * (ND + 1) + ND]);
  }
  for (int d = max_deg + 1; d < ND; ++d)
    GCMI_CHECK_ARG(P.deg_count_in[d] == 0, "collate: an atom has more than max_deg=%d neighbours", max_deg);
  P.base_ext = staging + o_base;
  P.mol_win_ext = staging + o_win;
  const int rc = plan_serial(hist.data(), n_sel, n_deg, win_cap, win_cap > 0, (1LL << 31), (1LL << 31),
                             staging + o_meta, nullptr, P);
  if (rc != GCMI_OK) return rc;
  if (!P.want_win)
    for (int64_t p = 0; p < n_sel; ++p) staging[o_win + p] = 0;
  const int64_t n_win = P.want_win ? (int64_t)P.wins.size() : 0;
  const int64_t o_desc = o_meta + n_win * GCMI_WIN_META_INTS;
  for (int64_t w = 0; w < n_win; ++w) {
    const Win& W = P.wins[(size_t)w];
    int32_t* D = staging + o_desc + w * WD;
    for (int d = 0; d < ND; ++d) {
      D[d] = W.begin[d];
      D[ND + d] = W.sb[d];
      D[2 * ND + d] = W.eb[d];
    }
    GCMI_CHECK_ARG(W.eoff < (1LL << 31), "collate: window entries exceed int32");
    D[3 * ND] = (int32_t)W.eoff;
    D[3 * ND + 1] = D[3 * ND + 2] = 0;
  }
  out_offsets[0] = o_base;
  out_offsets[1] = o_win;
  out_offsets[2] = o_off;
  out_offsets[3] = o_a0;
  out_offsets[4] = o_meta;
  out_offsets[5] = o_desc;
  out_offsets[6] = o_desc + n_win * WD;  // words used
  out_offsets[7] = P.want_win ? P.win_entries : 0;
  graph->n_atoms = (int32_t)P.n_atoms;
  graph->n_edges = (int32_t)P.n_edges;
  graph->n_mols = (int32_t)n_sel;
  graph->max_deg = max_deg;
  for (int d = 0; d < GCMI_MAX_DEG + 2; ++d) {
    graph->deg_start[d] = (int32_t)P.deg_start[d <= n_deg ? d : n_deg];
    graph->edge_start[d] = (int32_t)P.edge_start[d <= n_deg ? d : n_deg];
  }
  graph->d_col_idx = nullptr;
  graph->d_membership = nullptr;
  graph->d_mol_runs = nullptr;
  graph->d_rev_pos = nullptr;
  graph->n_win = (int32_t)n_win;
  graph->n_win_big = P.want_win ? P.n_win_big : 0;
  graph->win_alloc = P.want_win ? P.win_alloc : 0;
  graph->win_ecap = P.want_win ? P.win_ecap : 0;
  graph->win_alloc_big = P.want_win ? P.win_alloc_big : 0;
  graph->win_ecap_big = P.want_win ? P.win_ecap_big : 0;
  graph->win_reserved[0] = P.want_win ? P.n_mols_with_atoms : 0;  // molecules the windows cover (readout over windows)
  graph->win_reserved[1] = 0;
  graph->d_win_meta = nullptr;
  graph->d_win_edges = nullptr;
  return GCMI_OK;
}

int gcmi_collate_rows(const void* d_features, int64_t n_feat, const int64_t* d_adj_ptr, const int32_t* d_adj_idx,
                      const int32_t* d_rank, const uint8_t* d_rev, const int32_t* d_staging, const int64_t* offsets,
                      const gcmi_graph* plan, float* d_out_features, int64_t out_ld, int32_t* d_membership,
                      int32_t* d_col_idx, int32_t* d_mol_runs, uint8_t* d_rev_pos, uint16_t* d_win_edges,
                      int64_t* d_src_atom, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RowsArgs A;
  const int rc = fill_rows_args(A, d_features, n_feat, d_adj_ptr, d_adj_idx, d_rank, d_rev, d_staging, offsets, plan,
                                d_out_features, out_ld, d_membership, d_col_idx, d_mol_runs, d_rev_pos, d_win_edges);
  if (rc != GCMI_OK) return rc;
  if (A.want_win && offsets[7] > 0)  // the padding entries at the end of every window
    if (hipMemsetAsync(d_win_edges, 0, (size_t)offsets[7] * sizeof(uint16_t), stream) != hipSuccess) {
      ::gcmi::set_error("collate_rows: hipMemsetAsync failed");
      return GCMI_ERR_LAUNCH;
    }
  A.src_atom = d_src_atom;
  if (A.n_atoms > 0) {
    collate_rows_kernel<<<(unsigned)((A.n_atoms + 255) / 256), 256, 0, stream>>>(A);
    GCMI_CHECK_LAUNCH("collate_rows_kernel");
    if (d_src_atom) {
      const int64_t n_el = (int64_t)A.n_atoms * A.out_ld;
      GCMI_CHECK_ARG(n_el / 256 < (1LL << 31), "collate_rows: feature block too large for one launch");
      if (A.out_ld % 4 == 0 && ((uintptr_t)A.out_feat & 15) == 0 && n_el / 4 < (1LL << 31)) {
        const uint32_t n_quads = (uint32_t)(n_el / 4);
        collate_features_quad_kernel<<<(n_quads + 255u) / 256u, 256, 0, stream>>>(A, (uint32_t)(A.out_ld / 4), n_quads);
      } else {
        collate_features_kernel<<<(unsigned)((n_el + 255) / 256), 256, 0, stream>>>(A);
      }
      GCMI_CHECK_LAUNCH("collate_features_kernel");
    }
  }
  const int64_t n_runs = (int64_t)A.n_sel * A.n_deg;
  if (d_mol_runs && n_runs > 0) {
    collate_runs_kernel<<<(unsigned)((n_runs + 255) / 256), 256, 0, stream>>>(A);
    GCMI_CHECK_LAUNCH("collate_runs_kernel");
  }
  return GCMI_OK;
}

/* The same per-atom routine run by host loops over HOST buffers: lets the CPU test suite check the row arithmetic of
   gcmi_collate_rows against gcmi_collate_plans without a GPU.  Not used by the library's own paths. */
int gcmi_collate_rows_host(const void* features, int64_t n_feat, const int64_t* adj_ptr, const int32_t* adj_idx,
                           const int32_t* rank, const uint8_t* rev, const int32_t* staging, const int64_t* offsets,
                           const gcmi_graph* plan, float* out_features, int64_t out_ld, int32_t* membership,
                           int32_t* col_idx, int32_t* mol_runs, uint8_t* rev_pos, uint16_t* win_edges) {
  RowsArgs A;
  const int rc = fill_rows_args(A, features, n_feat, adj_ptr, adj_idx, rank, rev, staging, offsets, plan, out_features,
                                out_ld, membership, col_idx, mol_runs, rev_pos, win_edges);
  if (rc != GCMI_OK) return rc;
  if (A.want_win)
    for (int64_t q = 0; q < offsets[7]; ++q) win_edges[q] = 0;
  for (int32_t i = 0; i < A.n_atoms; ++i) collate_atom(A, i);
  if (mol_runs)
    for (int64_t k = 0; k < (int64_t)A.n_sel * A.n_deg; ++k) collate_run(A, (int32_t)k);
  return GCMI_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------- many batches per call
// The host side of the small-batch engine (gcmi_small_fit / gcmi_small_predict): all batches of a chunk of an
// epoch collated in ONE call into ONE arena (one H2D copy), worker threads taking whole batches.  Per batch the
// output is exactly what gcmi_collate_plans writes (without LDS windows).
namespace {
inline int64_t up4w(int64_t n) { return (n + 3) / 4 * 4; }
}  // namespace

extern "C" {

int64_t gcmi_collate_batches_layout(const int64_t* atom_ptr, const int64_t* adj_ptr, const int64_t* sel,
                                    const int64_t* batch_ptr, int64_t n_batches, int64_t ld, int32_t max_deg,
                                    int64_t mols_out, int64_t* out_parts, int64_t* out_counts) {
  if (!(atom_ptr && adj_ptr && batch_ptr && out_parts && out_counts && (sel || n_batches == 0))) {
    gcmi::set_error("collate_batches_layout: NULL argument");
    return GCMI_ERR_ARG;
  }
  if (n_batches < 0 || ld <= 0 || max_deg < 0 || max_deg > GCMI_MAX_DEG) {
    gcmi::set_error("collate_batches_layout: bad n_batches / ld / max_deg");
    return GCMI_ERR_ARG;
  }
  const int n_deg = max_deg + 1;
  // feature rows of all batches first, batch after batch (every batch padded to an even number of rows so that
  // 8-byte code rows keep 16-byte alignment): ONE contiguous row array, so atom codes expand in one launch
  int64_t rows = 0;
  for (int64_t b = 0; b < n_batches; ++b) {
    int64_t na = 0, ne = 0;
    const int64_t n_sel = batch_ptr[b + 1] - batch_ptr[b];
    if (n_sel < 0 || (mols_out > 0 && n_sel > mols_out)) {
      gcmi::set_error("collate_batches_layout: batch %lld has %lld molecules (mols_out %lld)", (long long)b,
                      (long long)n_sel, (long long)mols_out);
      return GCMI_ERR_ARG;
    }
    const int rc = gcmi_collate_sizes(atom_ptr, adj_ptr, sel + batch_ptr[b], n_sel, &na, &ne);
    if (rc != GCMI_OK) return rc;
    out_counts[3 * b] = na;
    out_counts[3 * b + 1] = ne;
    out_counts[3 * b + 2] = rows;  // first feature row of the batch in the chunk-wide row array
    rows += (na + 1) / 2 * 2;
  }
  int64_t off = up4w(rows * ld);
  for (int64_t b = 0; b < n_batches; ++b) {
    const int64_t na = out_counts[3 * b], ne = out_counts[3 * b + 1];
    const int64_t n_sel = batch_ptr[b + 1] - batch_ptr[b];
    const int64_t n_out = mols_out > 0 ? mols_out : n_sel;
    int64_t* p = out_parts + 5 * b;
    p[0] = out_counts[3 * b + 2] * ld;  // features
    p[1] = off;                         // membership
    off += up4w(na);
    p[2] = off;  // col_idx
    off += up4w(ne);
    p[3] = off;  // mol_runs (n_out molecules)
    off += up4w(n_out * n_deg * 2);
    p[4] = off;  // rev_pos (bytes)
    off += up4w((ne + 3) / 4);
  }
  return off > 4 ? off : 4;
}

int gcmi_collate_batches(const float* atom_features, int64_t n_feat, const int64_t* atom_ptr, const int64_t* adj_ptr,
                         const int32_t* adj_idx, const int64_t* sel, const int64_t* batch_ptr, int64_t n_batches,
                         int32_t max_deg, int64_t ld, int64_t mols_out, float* arena, const int64_t* parts,
                         const int64_t* counts, gcmi_graph* graphs, int32_t* out_symmetric, int32_t n_threads) {
  GCMI_CHECK_ARG(atom_features && atom_ptr && adj_ptr && batch_ptr && arena && parts && counts && graphs &&
                     (sel || n_batches == 0),
                 "collate_batches: NULL argument");
  GCMI_CHECK_ARG(n_batches >= 0 && ld >= n_feat && n_feat > 0, "collate_batches: bad n_batches / ld");
  const int n_deg = max_deg + 1;
  const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
  int nt = n_threads > 0 ? n_threads : 16;
  nt = std::max(1, std::min(std::min(nt, hw), (int)std::min<int64_t>(n_batches, 64)));
  std::vector<int> status((size_t)nt, GCMI_OK);
  std::vector<std::string> messages((size_t)nt);
  int32_t* i32 = reinterpret_cast<int32_t*>(arena);
  auto work = [&](int t) {
    for (int64_t b = t; b < n_batches; b += nt) {
      const int64_t n_sel = batch_ptr[b + 1] - batch_ptr[b];
      const int64_t n_out = mols_out > 0 ? mols_out : n_sel;
      const int64_t* p = parts + 5 * b;
      int32_t sym = 1;
      int32_t* runs = i32 + p[3];
      const int rc = gcmi_collate_plans(atom_features, n_feat, atom_ptr, adj_ptr, adj_idx, sel + batch_ptr[b], n_sel,
                                        max_deg, arena + p[0], ld, counts[3 * b], i32 + p[1], i32 + p[2],
                                        counts[3 * b + 1], runs, reinterpret_cast<uint8_t*>(i32 + p[4]), &sym, 0,
                                        nullptr, nullptr, graphs + b);
      if (rc != GCMI_OK) {
        status[(size_t)t] = rc;
        messages[(size_t)t] = gcmi_last_error();
        return;
      }
      // GraphGather emits mols_out rows whatever the batch holds (layers.py:6469-6479): the molecules beyond the
      // selection are empty
      for (int64_t k = n_sel * n_deg * 2; k < n_out * n_deg * 2; ++k) runs[k] = 0;
      graphs[b].n_mols = (int32_t)n_out;
      if (out_symmetric) out_symmetric[b] = sym;
    }
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  for (int t = 0; t < nt; ++t)
    if (status[(size_t)t] != GCMI_OK) {
      gcmi::set_error("collate_batches: %s", messages[(size_t)t].c_str());
      return status[(size_t)t];
    }
  return GCMI_OK;
}

/* Fill the descriptor array gcmi_small_fit / gcmi_small_predict read from what gcmi_collate_batches produced, once
 * the arena is on the device: d_arena = device copy of the arena; d_features = the chunk-wide feature row array
 * (the arena itself for float rows, or the rows expanded from atom codes), feature_ld its leading dimension.
 * Labels / weights (may be NULL): chunk-wide arrays with one row of label_stride / weight_stride floats per
 * molecule SLOT (mols_out slots per batch); outputs likewise (predict). */
int gcmi_small_bind(gcmi_small_batch* out, const gcmi_graph* graphs, const int64_t* parts, const int64_t* counts,
                    int64_t n_batches, const float* d_arena, const float* d_features, int64_t feature_ld,
                    int64_t mols_out, const int64_t* n_rows, const float* d_labels, int64_t label_stride,
                    const float* d_weights, int64_t weight_stride, float* d_logits, float* d_probs,
                    int64_t logit_stride, float* d_fingerprint, int64_t fp_stride) {
  GCMI_CHECK_ARG(out && graphs && parts && counts && d_arena && d_features && n_rows, "small_bind: NULL argument");
  const int32_t* i32 = reinterpret_cast<const int32_t*>(d_arena);
  for (int64_t b = 0; b < n_batches; ++b) {
    gcmi_small_batch& s = out[b];
    memset(&s, 0, sizeof(s));
    s.graph = graphs[b];
    const int64_t* p = parts + 5 * b;
    s.graph.d_membership = i32 + p[1];
    s.graph.d_col_idx = i32 + p[2];
    s.graph.d_mol_runs = i32 + p[3];
    s.graph.d_rev_pos = reinterpret_cast<const uint8_t*>(i32 + p[4]);
    s.d_atom_features = d_features + counts[3 * b + 2] * feature_ld;
    s.ld_features = feature_ld;
    s.n_rows = n_rows[b];
    const int64_t slot0 = b * mols_out;
    s.d_labels = d_labels ? d_labels + slot0 * label_stride : nullptr;
    s.d_weights = d_weights ? d_weights + slot0 * weight_stride : nullptr;
    s.d_logits = d_logits ? d_logits + slot0 * logit_stride : nullptr;
    s.d_probs = d_probs ? d_probs + slot0 * logit_stride : nullptr;
    s.d_fingerprint = d_fingerprint ? d_fingerprint + slot0 * fp_stride : nullptr;
  }
  return GCMI_OK;
}

}  // extern "C"
