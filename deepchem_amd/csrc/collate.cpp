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
  int64_t deg_count[ND] = {0};
  int64_t max_mol = 0;
  for (int64_t p = 0; p < n_sel; ++p) {
    int64_t sz = 0;
    for (int d = 0; d < n_deg; ++d) {
      deg_count[d] += hist[(size_t)p * ND + d];
      sz += hist[(size_t)p * ND + d];
    }
    max_mol = std::max(max_mol, sz);
  }
  int64_t deg_start[ND + 1], edge_start[ND + 1];
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
  GCMI_CHECK_ARG(n_atoms == 0 || (out_features && out_membership), "collate: NULL output");
  GCMI_CHECK_ARG(n_edges == 0 || (out_col_idx && adj_idx), "collate: NULL edge buffers");
  // a molecule larger than win_cap gets a window of its own; slots are 12-bit in the edge entries
  if (want_win && max_mol > GCMI_WIN_MAX_SLOTS) want_win = false;
  std::vector<int32_t> base((size_t)n_sel * ND);  // first row of molecule p inside degree block d
  std::vector<int32_t> mol_win(want_win ? (size_t)n_sel : 0);
  struct Win {
    int32_t begin[ND], sb[ND + 1], eb[ND];  // first row, slot prefix, edge-entry prefix per degree
    int64_t eoff;                           // first entry in the window-major edge array
  };
  std::vector<Win> wins;
  int32_t win_alloc = 0, win_ecap = 0, win_alloc_big = 0, win_ecap_big = 0, n_win_big = 0;
  {
    int64_t cursor[ND];
    for (int d = 0; d < ND; ++d) cursor[d] = deg_start[d];
    int64_t in_win = 0;
    for (int64_t p = 0; p < n_sel; ++p) {
      int64_t sz = 0;
      for (int d = 0; d < n_deg; ++d) sz += hist[(size_t)p * ND + d];
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
        for (int d = 0; d < n_deg; ++d) wins.back().sb[d] += hist[(size_t)p * ND + d];
      }
      for (int d = 0; d < n_deg; ++d) {
        base[(size_t)p * ND + d] = (int32_t)cursor[d];
        cursor[d] += hist[(size_t)p * ND + d];
      }
    }
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
      for (int32_t q = eacc; q < padded; ++q) out_win_edges[W.eoff + q] = 0;
    }
  }

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
  graph->win_reserved[0] = graph->win_reserved[1] = 0;
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
