// Host-side batch collation: ConvMol.agglomerate_mols
// (deepchem/feat/mol_graphs.py:256-349) over a packed molecule set, written
// straight into the (pinned) staging arena the H2D copy reads.
//
// The reference sorts a concatenated degree vector and re-indexes neighbour
// lists molecule by molecule in Python (~2.5 ms per 100 molecules).  Because
// the batch order is (degree, batch position, atom id inside the molecule) and
// the atoms of a molecule only need a stable order inside each degree, no sort
// is needed at all: one counting pass gives the degree-block starts, a second
// pass hands out rows from one cursor per degree.  Molecules are split into
// contiguous chunks across threads; per-chunk degree histograms turn into
// per-chunk cursor bases by a prefix sum, so the result is identical for any
// thread count.
#include <algorithm>
#include <thread>
#include <vector>

#include "common.h"

namespace {

constexpr int ND = GCMI_MAX_DEG + 1;

struct Chunk {
  int64_t p0, p1;        // batch positions [p0, p1)
  int64_t count[ND];     // atoms per degree in this chunk
  int64_t base[ND];      // first row of this chunk inside each degree block
  int bad_degree = 0;
};

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

int gcmi_collate(const float* atom_features, int64_t n_feat, const int64_t* atom_ptr,
                 const int64_t* adj_ptr, const int32_t* adj_idx, const int64_t* sel,
                 int64_t n_sel, int32_t max_deg, float* out_features, int64_t out_ld,
                 int64_t cap_atoms, int32_t* out_membership, int32_t* out_col_idx,
                 int64_t cap_edges, int32_t* out_mol_runs, gcmi_graph* graph) {
  GCMI_CHECK_ARG(atom_features && atom_ptr && adj_ptr && (sel || n_sel == 0) && graph,
                 "collate: NULL input");
  GCMI_CHECK_ARG(n_feat > 0 && out_ld >= n_feat, "collate: out_ld %lld < n_feat %lld",
                 (long long)out_ld, (long long)n_feat);
  GCMI_CHECK_ARG(max_deg >= 0 && max_deg <= GCMI_MAX_DEG, "collate: max_deg outside [0,%d]",
                 GCMI_MAX_DEG);
  GCMI_CHECK_ARG(n_sel >= 0 && n_sel < (1LL << 31), "collate: bad n_sel");
  const int n_deg = max_deg + 1;

  int n_threads = (int)std::min<int64_t>(std::max<int64_t>(1, n_sel / 256),
                                         std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())));
  std::vector<Chunk> chunks(n_threads);
  for (int t = 0; t < n_threads; ++t) {
    chunks[t].p0 = n_sel * t / n_threads;
    chunks[t].p1 = n_sel * (t + 1) / n_threads;
  }
  auto run = [&](auto&& fn) {
    if (n_threads == 1) {
      fn(0);
      return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(fn, t);
    for (auto& x : th) x.join();
  };

  // pass 1: degree histogram per chunk
  run([&](int t) {
    Chunk& c = chunks[t];
    for (int d = 0; d < ND; ++d) c.count[d] = 0;
    for (int64_t p = c.p0; p < c.p1; ++p) {
      const int64_t m = sel[p];
      for (int64_t a = atom_ptr[m]; a < atom_ptr[m + 1]; ++a) {
        const int64_t d = adj_ptr[a + 1] - adj_ptr[a];
        if (d < 0 || d > max_deg) {
          c.bad_degree = 1;
          continue;
        }
        c.count[d]++;
      }
    }
  });
  int64_t deg_count[ND] = {0};
  for (auto& c : chunks) {
    GCMI_CHECK_ARG(!c.bad_degree, "collate: an atom has more than max_deg=%d neighbours", max_deg);
    for (int d = 0; d < n_deg; ++d) deg_count[d] += c.count[d];
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
  for (int d = 0; d < n_deg; ++d) {
    int64_t run_base = deg_start[d];
    for (auto& c : chunks) {
      c.base[d] = run_base;
      run_base += c.count[d];
    }
  }

  // pass 2: hand out rows, write features / membership / neighbour tables / runs
  run([&](int t) {
    Chunk& c = chunks[t];
    int64_t cursor[ND];
    for (int d = 0; d < ND; ++d) cursor[d] = d < n_deg ? c.base[d] : 0;
    std::vector<int32_t> new_row;
    for (int64_t p = c.p0; p < c.p1; ++p) {
      const int64_t m = sel[p];
      const int64_t a0 = atom_ptr[m], a1 = atom_ptr[m + 1];
      new_row.resize((size_t)(a1 - a0));
      int32_t* runs = out_mol_runs ? out_mol_runs + p * n_deg * 2 : nullptr;
      if (runs)
        for (int d = 0; d < n_deg; ++d) runs[2 * d] = runs[2 * d + 1] = (int32_t)cursor[d];
      for (int64_t a = a0; a < a1; ++a) {
        const int d = (int)(adj_ptr[a + 1] - adj_ptr[a]);
        new_row[(size_t)(a - a0)] = (int32_t)cursor[d]++;
      }
      if (runs)
        for (int d = 0; d < n_deg; ++d) {
          runs[2 * d + 1] = (int32_t)cursor[d];
          if (runs[2 * d] == runs[2 * d + 1]) runs[2 * d] = runs[2 * d + 1] = 0;
        }
      for (int64_t a = a0; a < a1; ++a) {
        const int64_t e0 = adj_ptr[a];
        const int d = (int)(adj_ptr[a + 1] - e0);
        const int64_t row = new_row[(size_t)(a - a0)];
        float* dst = out_features + row * out_ld;
        memcpy(dst, atom_features + a * n_feat, sizeof(float) * (size_t)n_feat);
        for (int64_t f = n_feat; f < out_ld; ++f) dst[f] = 0.f;
        out_membership[row] = (int32_t)p;
        int32_t* cols = out_col_idx + edge_start[d] + (row - deg_start[d]) * d;
        for (int j = 0; j < d; ++j) {
          const int64_t nb = adj_idx[e0 + j];
          if (nb < 0 || nb >= a1 - a0) {
            c.bad_degree = 2;
            cols[j] = (int32_t)row;
          } else {
            cols[j] = new_row[(size_t)nb];
          }
        }
      }
    }
  });

  for (auto& c : chunks)
    GCMI_CHECK_ARG(c.bad_degree != 2, "collate: a neighbour id is outside its molecule");
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
  return GCMI_OK;
}

}  // extern "C"
