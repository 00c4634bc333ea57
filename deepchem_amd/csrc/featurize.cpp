// Host-side SMILES -> atom / bond / pair feature arrays.
//
// Replaces, for SMILES input, what the reference computes per molecule in Python on top of an rdkit Mol:
//   atom_features                 deepchem/feat/graph_features.py:282-391   (75 columns)
//   bond_features                 graph_features.py:394-459                 (6 columns)
//   pair_features / find_distance graph_features.py:532-695                 (14 columns, all pairs)
//   ConvMolFeaturizer._featurize  graph_features.py:845-914   (node matrix + adjacency lists in bond order)
//   WeaveFeaturizer._featurize    graph_features.py:1037-1078
//   MolecularFeaturizer.featurize feat/base_classes.py:254-330 (a molecule that cannot be read is dropped to an
//                                 empty entry, not an exception)
//
// rdkit is not available here, so the chemistry rdkit's MolFromSmiles applies is implemented directly: explicit
// hydrogen removal, Kekulisation (perfect matching over the aromatic atoms with a free valence), valence check
// and implicit hydrogens, radicals of bracket atoms, ring perception (relevant cycles from Horton candidates,
// GF(2) elimination on bond bit vectors), rdkit's default aromaticity model (electron donor types, 4n+2 on single
// rings, then on fused pairs and triples along their outer bonds), conjugation and hybridisation.  The checker
// is oracle/smiles_oracle.py, an independent restatement with brute-force algorithms; see its header for what
// pins it (PARITY UNPINNED beyond those vectors).
//
// Output is the packed layout the collation code reads (molecule-major atom rows + CSR adjacency), filled by
// worker threads molecule by molecule; results do not depend on the thread count.
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "common.h"

namespace {

constexpr int kMaxAtomBonds = 12;
constexpr int kNumElements = 103;

const char* const kSymbols[kNumElements] = {
    "H",  "He", "Li", "Be", "B",  "C",  "N",  "O",  "F",  "Ne", "Na", "Mg", "Al", "Si", "P",  "S",  "Cl", "Ar",
    "K",  "Ca", "Sc", "Ti", "V",  "Cr", "Mn", "Fe", "Co", "Ni", "Cu", "Zn", "Ga", "Ge", "As", "Se", "Br", "Kr",
    "Rb", "Sr", "Y",  "Zr", "Nb", "Mo", "Tc", "Ru", "Rh", "Pd", "Ag", "Cd", "In", "Sn", "Sb", "Te", "I",  "Xe",
    "Cs", "Ba", "La", "Ce", "Pr", "Nd", "Pm", "Sm", "Eu", "Gd", "Tb", "Dy", "Ho", "Er", "Tm", "Yb", "Lu", "Hf",
    "Ta", "W",  "Re", "Os", "Ir", "Pt", "Au", "Hg", "Tl", "Pb", "Bi", "Po", "At", "Rn", "Fr", "Ra", "Ac", "Th",
    "Pa", "U",  "Np", "Pu", "Am", "Cm", "Bk", "Cf", "Es", "Fm", "Md", "No", "Lr"};

// column of each element in the reference's 44-symbol list (graph_features.py:322-367); 43 = 'Unknown'
const char* const kSymbolList[43] = {"C",  "N",  "O",  "S",  "F",  "Si", "P",  "Cl", "Br", "Mg", "Na",
                                     "Ca", "Fe", "As", "Al", "I",  "B",  "V",  "K",  "Tl", "Yb", "Sb",
                                     "Sn", "Ag", "Pd", "Co", "Se", "Ti", "Zn", "H",  "Li", "Ge", "Cu",
                                     "Au", "Ni", "Cd", "In", "Mn", "Zr", "Cr", "Pt", "Hg", "Pb"};

struct Tables {
  int8_t valences[kNumElements + 1][4];  // terminated by -2; -1 = "anything"
  int8_t n_outer[kNumElements + 1];
  int8_t symbol_col[kNumElements + 1];
  bool early[kNumElements + 1];
  Tables() {
    for (int z = 0; z <= kNumElements; ++z) {
      valences[z][0] = -1;
      valences[z][1] = valences[z][2] = valences[z][3] = -2;
      early[z] = false;
      symbol_col[z] = 43;
      n_outer[z] = static_cast<int8_t>(outer(z));
    }
    auto set = [&](int z, int a, int b = -2, int c = -2, int d = -2) {
      valences[z][0] = a; valences[z][1] = b; valences[z][2] = c; valences[z][3] = d;
    };
    set(1, 1); set(2, 0); set(3, 1, -1); set(4, 2); set(5, 3); set(6, 4); set(7, 3); set(8, 2); set(9, 1);
    set(10, 0); set(11, 1, -1); set(12, 2, -1); set(13, 3, 6); set(14, 4, 6); set(15, 3, 5, 7);
    set(16, 2, 4, 6); set(17, 1); set(18, 0); set(19, 1, -1); set(20, 2, -1); set(31, 3); set(32, 4);
    set(33, 3, 5, 7); set(34, 2, 4, 6); set(35, 1); set(36, 0); set(37, 1); set(38, 2); set(49, 3);
    set(50, 2, 4); set(51, 3, 5, 7); set(52, 2, 4, 6); set(53, 1, 3, 5); set(54, 0, 2, 4, 6); set(55, 1);
    set(56, 2); set(81, 3); set(82, 2, 4); set(83, 3, 5, 7); set(84, 2, 4, 6); set(85, 1, 3, 5); set(86, 0);
    for (int z : {3, 4, 5, 11, 12, 13, 19, 20, 31, 37, 38, 49, 55, 56, 81}) early[z] = true;
    for (int c = 0; c < 43; ++c)
      for (int z = 1; z <= kNumElements; ++z)
        if (!strcmp(kSymbols[z - 1], kSymbolList[c])) symbol_col[z] = static_cast<int8_t>(c);
  }
  static int outer(int z) {
    if (z <= 0) return 0;
    if (z <= 2) return z;
    if (z <= 10) return z - 2;
    if (z <= 18) return z - 10;
    auto wide = [](int k) { return k <= 11 ? k : (k == 12 ? 2 : k - 10); };
    if (z <= 36) return wide(z - 18);
    if (z <= 54) return wide(z - 36);
    for (int start : {55, 87}) {
      if (z >= start && z < start + 32) {
        int k = z - start + 1;
        if (k <= 2) return k;
        if (k <= 17) return 3;
        return wide(k - 14);
      }
    }
    return 0;
  }
  int n_valences(int z) const {
    int n = 0;
    while (n < 4 && valences[z][n] != -2) ++n;
    return n;
  }
  int default_valence(int z) const { return (z >= 1 && z <= kNumElements) ? valences[z][0] : -1; }
};
const Tables kT;

enum Hyb : uint8_t { HYB_UNSPECIFIED = 0, HYB_S, HYB_SP, HYB_SP2, HYB_SP3, HYB_SP3D, HYB_SP3D2 };
enum Donor : uint8_t { D_NONE = 0, D_VACANT, D_ONE, D_TWO };

struct MAtom {
  int16_t z;
  int16_t iso;
  int8_t charge;
  uint8_t ex_h, imp_h, rad, hyb, nb;
  bool written_arom, bracket, arom;
  int32_t bond[kMaxAtomBonds];
};

struct MBond {
  int32_t a, b;
  int8_t order;  // 0 while an aromatic bond waits for kekulisation
  bool written_arom, arom, conj, ring;
  int other(int i) const { return i == a ? b : a; }
};

struct Ring {
  std::vector<int> atoms, bonds;
};

struct Cand {
  int len;
  std::vector<uint64_t> bits;
  std::vector<int> atoms, bonds;
};

struct Mol {
  std::vector<MAtom> atoms;
  std::vector<MBond> bonds;
  std::vector<Ring> rings;
  const char* error = nullptr;

  int degree(int i) const { return atoms[i].nb; }
  int total_h(int i) const { return atoms[i].imp_h + atoms[i].ex_h; }
  int explicit_valence(int i) const {
    int v = atoms[i].ex_h;
    for (int k = 0; k < atoms[i].nb; ++k) v += bonds[atoms[i].bond[k]].order;
    return v;
  }
  int total_valence(int i) const { return explicit_valence(i) + atoms[i].imp_h; }
  bool fail(const char* why) {
    error = why;
    return false;
  }
};

// ----------------------------------------------------------------------------------------------- parsing

int element_of(const char* s, int len) {
  for (int z = 1; z <= kNumElements; ++z)
    if (static_cast<int>(strlen(kSymbols[z - 1])) == len && !strncmp(kSymbols[z - 1], s, len)) return z;
  return 0;
}

int aromatic_element(const char* s, int* used) {
  if ((s[0] == 's' && s[1] == 'e')) { *used = 2; return 34; }
  if ((s[0] == 'a' && s[1] == 's')) { *used = 2; return 33; }
  if ((s[0] == 't' && s[1] == 'e')) { *used = 2; return 52; }
  *used = 1;
  switch (s[0]) {
    case 'b': return 5;
    case 'c': return 6;
    case 'n': return 7;
    case 'o': return 8;
    case 'p': return 15;
    case 's': return 16;
    default: return 0;
  }
}

inline bool is_digit(char c) { return c >= '0' && c <= '9'; }
inline bool is_lower(char c) { return c >= 'a' && c <= 'z'; }
inline bool is_upper(char c) { return c >= 'A' && c <= 'Z'; }

MAtom new_atom(int z, bool arom, bool bracket) {
  MAtom a;
  memset(&a, 0, sizeof(a));
  a.z = static_cast<int16_t>(z);
  a.written_arom = arom;
  a.bracket = bracket;
  return a;
}

bool read_bracket(const char* s, int len, MAtom* out) {
  int k = 0, iso = 0;
  while (k < len && is_digit(s[k])) iso = std::min(iso * 10 + (s[k++] - '0'), 30000);
  if (k >= len) return false;
  int z = 0;
  bool arom = false;
  if (is_lower(s[k])) {
    int used = 0;
    z = aromatic_element(s + k, &used);
    if (used == 2 && k + 1 >= len) z = 0;
    if (!z) return false;
    arom = true;
    k += used;
  } else if (is_upper(s[k])) {
    int n = (k + 1 < len && is_lower(s[k + 1])) ? 2 : 1;
    z = element_of(s + k, n);
    if (!z) return false;
    k += n;
  } else {
    return false;
  }
  if (k < len && s[k] == '@') {
    ++k;
    if (k < len && s[k] == '@') {
      ++k;
    } else if (k + 1 < len && ((s[k] == 'T' && s[k + 1] == 'H') || (s[k] == 'A' && s[k + 1] == 'L') ||
                               (s[k] == 'S' && s[k + 1] == 'P') || (s[k] == 'T' && s[k + 1] == 'B') ||
                               (s[k] == 'O' && s[k + 1] == 'H'))) {
      k += 2;
      while (k < len && is_digit(s[k])) ++k;
    }
  }
  int h = 0;
  if (k < len && s[k] == 'H') {
    ++k;
    if (k < len && is_digit(s[k])) {
      h = 0;
      while (k < len && is_digit(s[k])) h = std::min(h * 10 + (s[k++] - '0'), 100);
    } else {
      h = 1;
    }
  }
  int charge = 0;
  if (k < len && (s[k] == '+' || s[k] == '-')) {
    const char sign = s[k];
    int run = 0;
    while (k < len && s[k] == sign) { ++run; ++k; }
    if (k < len && is_digit(s[k])) {
      int v = 0;
      while (k < len && is_digit(s[k])) v = std::min(v * 10 + (s[k++] - '0'), 100);
      run = v;
    }
    charge = sign == '+' ? run : -run;
  }
  if (k < len && s[k] == ':') {
    ++k;
    while (k < len && is_digit(s[k])) ++k;
  }
  if (k != len) return false;
  *out = new_atom(z, arom, true);
  out->iso = static_cast<int16_t>(iso);
  out->ex_h = static_cast<uint8_t>(h);
  out->charge = static_cast<int8_t>(std::max(-100, std::min(100, charge)));
  return true;
}

struct BondSpec {
  int8_t order;  // -1 = not given
  bool arom;
};

bool add_bond(Mol& m, int a, int b, BondSpec spec) {
  if (a == b) return m.fail("bond from an atom to itself");
  MAtom& A = m.atoms[a];
  MAtom& B = m.atoms[b];
  for (int k = 0; k < A.nb; ++k)
    if (m.bonds[A.bond[k]].other(a) == b) return m.fail("two bonds between the same atoms");
  if (A.nb >= kMaxAtomBonds || B.nb >= kMaxAtomBonds) return m.fail("too many bonds on one atom");
  MBond bd;
  bd.a = a;
  bd.b = b;
  bd.arom = bd.conj = bd.ring = false;
  if (spec.order < 0 && !spec.arom) {
    const bool arom = A.written_arom && B.written_arom;
    bd.order = arom ? 0 : 1;
    bd.written_arom = arom;
  } else {
    bd.order = spec.arom ? 0 : spec.order;
    bd.written_arom = spec.arom;
  }
  const int idx = static_cast<int>(m.bonds.size());
  m.bonds.push_back(bd);
  A.bond[A.nb++] = idx;
  B.bond[B.nb++] = idx;
  return true;
}

bool parse(const char* s, Mol& m) {
  m.atoms.clear();
  m.bonds.clear();
  m.rings.clear();
  m.error = nullptr;
  while (*s == ' ' || *s == '\t' || *s == '\n' || *s == '\r') ++s;
  int n = static_cast<int>(strlen(s));
  while (n > 0 && (s[n - 1] == ' ' || s[n - 1] == '\t' || s[n - 1] == '\n' || s[n - 1] == '\r')) --n;
  std::vector<int> stack;
  struct Open { int atom; BondSpec spec; bool open; };
  Open rings[100];
  for (auto& r : rings) r.open = false;
  int n_open = 0;
  int prev = -1;
  bool have_pending = false;
  BondSpec pending{-1, false};
  const BondSpec none{-1, false};
  int i = 0;
  while (i < n) {
    const char ch = s[i];
    if (ch == '(') {
      if (prev < 0) return m.fail("branch before any atom");
      stack.push_back(prev);
      ++i;
    } else if (ch == ')') {
      if (stack.empty()) return m.fail("unbalanced ')'");
      prev = stack.back();
      stack.pop_back();
      ++i;
    } else if (ch == '-' || ch == '/' || ch == '\\') {
      pending = {1, false}; have_pending = true; ++i;
    } else if (ch == '=') {
      pending = {2, false}; have_pending = true; ++i;
    } else if (ch == '#') {
      pending = {3, false}; have_pending = true; ++i;
    } else if (ch == ':') {
      pending = {-1, true}; have_pending = true; ++i;
    } else if (ch == '.') {
      if (have_pending) return m.fail("bond symbol before '.'");
      prev = -1;
      ++i;
    } else if (is_digit(ch) || ch == '%') {
      int num;
      if (ch == '%') {
        if (i + 2 >= n || !is_digit(s[i + 1]) || !is_digit(s[i + 2])) return m.fail("bad %nn ring closure");
        num = (s[i + 1] - '0') * 10 + (s[i + 2] - '0');
        i += 3;
      } else {
        num = ch - '0';
        ++i;
      }
      if (prev < 0) return m.fail("ring closure before any atom");
      if (rings[num].open) {
        rings[num].open = false;
        --n_open;
        if (!add_bond(m, rings[num].atom, prev, have_pending ? pending : rings[num].spec)) return false;
      } else {
        rings[num] = {prev, have_pending ? pending : none, true};
        ++n_open;
      }
      have_pending = false;
    } else {
      MAtom atom;
      if (ch == '[') {
        int j = i + 1;
        while (j < n && s[j] != ']') ++j;
        if (j >= n) return m.fail("unclosed bracket atom");
        if (!read_bracket(s + i + 1, j - i - 1, &atom)) return m.fail("cannot read bracket atom");
        i = j + 1;
      } else if (ch == 'C' && i + 1 < n && s[i + 1] == 'l') {
        atom = new_atom(17, false, false); i += 2;
      } else if (ch == 'B' && i + 1 < n && s[i + 1] == 'r') {
        atom = new_atom(35, false, false); i += 2;
      } else {
        int z = 0;
        bool arom = false;
        switch (ch) {
          case 'B': z = 5; break;
          case 'C': z = 6; break;
          case 'N': z = 7; break;
          case 'O': z = 8; break;
          case 'P': z = 15; break;
          case 'S': z = 16; break;
          case 'F': z = 9; break;
          case 'I': z = 53; break;
          case 'b': z = 5; arom = true; break;
          case 'c': z = 6; arom = true; break;
          case 'n': z = 7; arom = true; break;
          case 'o': z = 8; arom = true; break;
          case 'p': z = 15; arom = true; break;
          case 's': z = 16; arom = true; break;
          default: return m.fail("unexpected character");
        }
        atom = new_atom(z, arom, false);
        ++i;
      }
      m.atoms.push_back(atom);
      const int idx = static_cast<int>(m.atoms.size()) - 1;
      if (prev >= 0) {
        if (!add_bond(m, prev, idx, have_pending ? pending : none)) return false;
      } else if (have_pending) {
        return m.fail("bond symbol without a left atom");
      }
      have_pending = false;
      prev = idx;
    }
  }
  if (!stack.empty()) return m.fail("unbalanced '('");
  if (n_open) return m.fail("unclosed ring bond");
  if (have_pending) return m.fail("dangling bond symbol");
  if (m.atoms.empty()) return m.fail("no atoms");
  return true;
}

// rdkit's RemoveHs as MolFromSmiles applies it: a plain [H] with one single bond to a heavy atom turns into a
// hydrogen count (explicit on bracket atoms, implicit otherwise).
void remove_explicit_hydrogens(Mol& m, std::vector<int>& remap) {
  const int n = static_cast<int>(m.atoms.size());
  bool any = false;
  remap.assign(n, 0);
  for (int i = 0; i < n; ++i) {
    const MAtom& a = m.atoms[i];
    if (a.z == 1 && a.iso == 0 && a.charge == 0 && a.nb == 1 && a.ex_h == 0) {
      const MBond& b = m.bonds[a.bond[0]];
      if (m.atoms[b.other(i)].z != 1 && b.order == 1) {
        remap[i] = -1;
        any = true;
      }
    }
  }
  if (!any) return;
  std::vector<MAtom> atoms;
  for (int i = 0; i < n; ++i) {
    if (remap[i] < 0) continue;
    remap[i] = static_cast<int>(atoms.size());
    MAtom a = m.atoms[i];
    a.nb = 0;
    atoms.push_back(a);
  }
  std::vector<MBond> bonds;
  for (const MBond& b : m.bonds) {
    if (remap[b.a] < 0 || remap[b.b] < 0) {
      const int heavy = remap[b.a] < 0 ? b.b : b.a;
      if (m.atoms[heavy].bracket) atoms[remap[heavy]].ex_h++;
      continue;
    }
    MBond nb = b;
    nb.a = remap[b.a];
    nb.b = remap[b.b];
    const int idx = static_cast<int>(bonds.size());
    bonds.push_back(nb);
    atoms[nb.a].bond[atoms[nb.a].nb++] = idx;
    atoms[nb.b].bond[atoms[nb.b].nb++] = idx;
  }
  m.atoms.swap(atoms);
  m.bonds.swap(bonds);
}

// ------------------------------------------------------------------------------------------------ cleanup

int raw_valence(const Mol& m, int i) {  // before kekulisation: waiting aromatic bonds count 1.5
  int twice = 2 * m.atoms[i].ex_h;
  for (int k = 0; k < m.atoms[i].nb; ++k) {
    const int o = m.bonds[m.atoms[i].bond[k]].order;
    twice += o == 0 ? 3 : 2 * o;
  }
  return twice / 2;
}

// rdkit's first sanitisation step: neutral hypervalent N, P and halogens written with double bonds become
// charge-separated (CN(=O)=O -> C[N+](=O)[O-], CN=N#N -> CN=[N+]=[N-], C=P(=O)(C)C -> C=[P+]([O-])(C)C,
// OCl(=O)(=O)=O -> O[Cl+3]([O-])([O-])[O-]).
void cleanup(Mol& m) {
  const int n = static_cast<int>(m.atoms.size());
  for (int i = 0; i < n; ++i) {
    MAtom& a = m.atoms[i];
    if (a.charge != 0) continue;
    if (a.z == 7) {
      if (raw_valence(m, i) != 5) continue;
      for (int k = 0; k < a.nb; ++k) {
        MBond& b = m.bonds[a.bond[k]];
        MAtom& nb = m.atoms[b.other(i)];
        if (nb.z == 8 && nb.charge == 0 && b.order == 2) {
          b.order = 1; a.charge = 1; nb.charge = -1;
          break;
        }
        if (nb.z == 7 && nb.charge == 0 && b.order == 3) {
          b.order = 2; a.charge = 1; nb.charge = -1;
          break;
        }
      }
    } else if (a.z == 15) {
      if (raw_valence(m, i) != 5) continue;
      int dbl_o = -1;
      bool to_c_or_n = false;
      for (int k = 0; k < a.nb; ++k) {
        const MBond& b = m.bonds[a.bond[k]];
        const MAtom& nb = m.atoms[b.other(i)];
        if (nb.z == 8 && nb.charge == 0 && b.order == 2) dbl_o = a.bond[k];
        else if ((nb.z == 6 || nb.z == 7) && nb.nb >= 2 && b.order == 2) to_c_or_n = true;
      }
      if (dbl_o >= 0 && to_c_or_n) {
        MBond& b = m.bonds[dbl_o];
        b.order = 1;
        a.charge = 1;
        m.atoms[b.other(i)].charge = -1;
      }
    } else if (a.z == 17 || a.z == 35 || a.z == 53) {
      const int ev = raw_valence(m, i);
      if (ev != 3 && ev != 5 && ev != 7) continue;
      bool all_o = true;
      for (int k = 0; k < a.nb; ++k) all_o &= m.atoms[m.bonds[a.bond[k]].other(i)].z == 8;
      if (!all_o) continue;
      for (int k = 0; k < a.nb; ++k) {
        MBond& b = m.bonds[a.bond[k]];
        if (b.order == 2) {
          b.order = 1;
          m.atoms[b.other(i)].charge = -1;
          a.charge++;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------- rings

// Ring bonds = non-bridge edges (one depth-first pass with low-links, iterative).
void mark_ring_bonds(Mol& m, std::vector<int>& disc, std::vector<int>& low, std::vector<int>& via,
                     std::vector<int>& it, std::vector<int>& stack) {
  const int n = static_cast<int>(m.atoms.size());
  for (MBond& b : m.bonds) b.ring = true;
  disc.assign(n, -1);
  low.assign(n, 0);
  via.assign(n, -1);
  it.assign(n, 0);
  int timer = 0;
  for (int root = 0; root < n; ++root) {
    if (disc[root] >= 0) continue;
    stack.clear();
    stack.push_back(root);
    disc[root] = low[root] = timer++;
    while (!stack.empty()) {
      const int u = stack.back();
      if (it[u] < m.atoms[u].nb) {
        const int bi = m.atoms[u].bond[it[u]++];
        if (bi == via[u]) continue;
        const int v = m.bonds[bi].other(u);
        if (disc[v] < 0) {
          disc[v] = low[v] = timer++;
          via[v] = bi;
          stack.push_back(v);
        } else {
          low[u] = std::min(low[u], disc[v]);
        }
      } else {
        stack.pop_back();
        if (via[u] >= 0) {
          const int p = m.bonds[via[u]].other(u);
          low[p] = std::min(low[p], low[u]);
          if (low[u] > disc[p]) m.bonds[via[u]].ring = false;
        }
      }
    }
  }
}

struct RingScratch {
  std::vector<int> dist, parent, pbond, queue, stamp, ring_atoms;
  std::vector<Cand> cands;
  std::vector<uint64_t> seen_hash;
  std::vector<std::vector<uint64_t>> basis;  // row per pivot bit (empty = none)
  std::vector<uint64_t> tmp;
};

inline uint64_t mix64(int v) {  // splitmix64 finaliser
  uint64_t z = (static_cast<uint64_t>(v) + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

bool reduce_vec(const std::vector<std::vector<uint64_t>>& basis, std::vector<uint64_t>& v) {
  const int W = static_cast<int>(v.size());
  for (int w = W - 1; w >= 0; --w) {
    while (v[w]) {
      const int bit = 63 - __builtin_clzll(v[w]);
      const auto& row = basis[w * 64 + bit];
      if (row.empty()) return true;  // independent
      for (int k = 0; k <= w; ++k) v[k] ^= row[k];
    }
  }
  return false;
}

void insert_vec(std::vector<std::vector<uint64_t>>& basis, std::vector<uint64_t>& v, int* rank) {
  if (!reduce_vec(basis, v)) return;
  const int W = static_cast<int>(v.size());
  for (int w = W - 1; w >= 0; --w) {
    if (v[w]) {
      basis[w * 64 + 63 - __builtin_clzll(v[w])] = v;
      ++*rank;
      return;
    }
  }
}

// Relevant cycles: Horton candidates (shortest path to both ends of an edge from every root, two tie-break
// orders), sorted by length; a candidate stays if it is independent of all strictly shorter cycles.
void find_rings(Mol& m, RingScratch& rs) {
  m.rings.clear();
  const int n = static_cast<int>(m.atoms.size());
  const int nb = static_cast<int>(m.bonds.size());
  rs.ring_atoms.clear();
  int n_ring_bonds = 0;
  for (const MBond& b : m.bonds) n_ring_bonds += b.ring;
  if (!n_ring_bonds) return;
  for (int i = 0; i < n; ++i) {
    bool r = false;
    for (int k = 0; k < m.atoms[i].nb; ++k) r |= m.bonds[m.atoms[i].bond[k]].ring;
    if (r) rs.ring_atoms.push_back(i);
  }
  const int W = (nb + 63) / 64;
  rs.cands.clear();
  rs.seen_hash.clear();
  rs.dist.assign(n, -1);
  rs.parent.assign(n, -1);
  rs.pbond.assign(n, -1);
  rs.stamp.assign(n, -1);
  int n_comp = 0;
  {
    // components of the ring subgraph (for the cycle rank)
    std::vector<int>& seen = rs.dist;
    for (int r : rs.ring_atoms) {
      if (seen[r] >= 0) continue;
      ++n_comp;
      rs.queue.clear();
      rs.queue.push_back(r);
      seen[r] = 0;
      for (size_t h = 0; h < rs.queue.size(); ++h) {
        const int u = rs.queue[h];
        for (int k = 0; k < m.atoms[u].nb; ++k) {
          const MBond& b = m.bonds[m.atoms[u].bond[k]];
          if (!b.ring) continue;
          const int v = b.other(u);
          if (seen[v] < 0) { seen[v] = 0; rs.queue.push_back(v); }
        }
      }
    }
  }
  const int rank_target = n_ring_bonds - static_cast<int>(rs.ring_atoms.size()) + n_comp;
  int stamp_id = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int root : rs.ring_atoms) {
      for (int a : rs.ring_atoms) rs.dist[a] = -1;
      rs.queue.clear();
      rs.queue.push_back(root);
      rs.dist[root] = 0;
      rs.parent[root] = -1;
      rs.pbond[root] = -1;
      for (size_t h = 0; h < rs.queue.size(); ++h) {
        const int u = rs.queue[h];
        const int deg = m.atoms[u].nb;
        for (int kk = 0; kk < deg; ++kk) {
          const int k = pass ? deg - 1 - kk : kk;
          const int bi = m.atoms[u].bond[k];
          if (!m.bonds[bi].ring) continue;
          const int v = m.bonds[bi].other(u);
          if (rs.dist[v] < 0) {
            rs.dist[v] = rs.dist[u] + 1;
            rs.parent[v] = u;
            rs.pbond[v] = bi;
            rs.queue.push_back(v);
          }
        }
      }
      for (int bi = 0; bi < nb; ++bi) {
        const MBond& b = m.bonds[bi];
        if (!b.ring) continue;
        const int x = b.a, y = b.b;
        if (rs.dist[x] < 0 || rs.dist[y] < 0) continue;  // other component
        if (rs.pbond[x] == bi || rs.pbond[y] == bi) continue;
        // paths root->x and root->y must share only the root
        ++stamp_id;
        for (int u = x; u != root; u = rs.parent[u]) rs.stamp[u] = stamp_id;
        bool disjoint = true;
        for (int u = y; u != root; u = rs.parent[u])
          if (rs.stamp[u] == stamp_id) { disjoint = false; break; }
        if (!disjoint) continue;
        // most candidates repeat a cycle already found from another root: filter on an order-free 64-bit
        // hash of the bond set (a sum of mixed bond ids) before building the candidate
        uint64_t hsh = mix64(bi);
        for (int u = x; u != root; u = rs.parent[u]) hsh += mix64(rs.pbond[u]);
        for (int u = y; u != root; u = rs.parent[u]) hsh += mix64(rs.pbond[u]);
        if (std::find(rs.seen_hash.begin(), rs.seen_hash.end(), hsh) != rs.seen_hash.end()) continue;
        rs.seen_hash.push_back(hsh);
        Cand c;
        c.len = rs.dist[x] + rs.dist[y] + 1;
        c.bits.assign(W, 0);
        c.bits[bi >> 6] |= 1ull << (bi & 63);
        c.bonds.push_back(bi);
        c.atoms.push_back(root);
        for (int u = x; u != root; u = rs.parent[u]) {
          c.bits[rs.pbond[u] >> 6] |= 1ull << (rs.pbond[u] & 63);
          c.bonds.push_back(rs.pbond[u]);
          c.atoms.push_back(u);
        }
        for (int u = y; u != root; u = rs.parent[u]) {
          c.bits[rs.pbond[u] >> 6] |= 1ull << (rs.pbond[u] & 63);
          c.bonds.push_back(rs.pbond[u]);
          c.atoms.push_back(u);
        }
        rs.cands.push_back(std::move(c));
      }
    }
  }
  std::sort(rs.cands.begin(), rs.cands.end(), [](const Cand& p, const Cand& q) {
    if (p.len != q.len) return p.len < q.len;
    return p.bits < q.bits;
  });
  rs.cands.erase(std::unique(rs.cands.begin(), rs.cands.end(),
                             [](const Cand& p, const Cand& q) { return p.len == q.len && p.bits == q.bits; }),
                 rs.cands.end());
  rs.basis.assign(static_cast<size_t>(W) * 64, std::vector<uint64_t>());
  int rank = 0;
  size_t k = 0;
  while (k < rs.cands.size() && rank < rank_target) {
    const int len = rs.cands[k].len;
    size_t e = k;
    while (e < rs.cands.size() && rs.cands[e].len == len) ++e;
    for (size_t c = k; c < e; ++c) {
      rs.tmp = rs.cands[c].bits;
      if (reduce_vec(rs.basis, rs.tmp)) {
        Ring r;
        r.atoms = rs.cands[c].atoms;
        r.bonds = rs.cands[c].bonds;
        m.rings.push_back(std::move(r));
      }
    }
    for (size_t c = k; c < e; ++c) {
      rs.tmp = rs.cands[c].bits;
      insert_vec(rs.basis, rs.tmp, &rank);
    }
    k = e;
  }
  // canonical order (size, sorted bond ids): the fused-ring search walks ring combinations in this order
  for (Ring& r : m.rings) std::sort(r.bonds.begin(), r.bonds.end());
  std::sort(m.rings.begin(), m.rings.end(), [](const Ring& p, const Ring& q) {
    if (p.bonds.size() != q.bonds.size()) return p.bonds.size() < q.bonds.size();
    return p.bonds < q.bonds;
  });
}

// ---------------------------------------------------------------------------------------------- kekulise

// smallest charge-adjusted allowed valence >= sigma, or -1
int fitting_valence(const MAtom& a, int sigma) {
  int chg = a.charge;
  if (kT.early[a.z]) chg = -chg;
  if (a.z == 6 && chg > 0) chg = -chg;
  for (int k = 0; k < 4 && kT.valences[a.z][k] != -2; ++k) {
    const int v = kT.valences[a.z][k];
    if (v < 0) continue;
    if (v + chg >= sigma) return v + chg;
  }
  return -1;
}

struct KekuleScratch {
  std::vector<uint8_t> needs, free_;
  std::vector<int> chosen, list;
  int64_t budget;
};

bool kekule_solve(Mol& m, KekuleScratch& ks, int n_free) {
  if (n_free == 0) return true;
  if (--ks.budget < 0) return false;
  int best = -1, best_n = 1 << 30;
  for (int i : ks.list) {
    if (!ks.free_[i]) continue;
    int opts = 0;
    const MAtom& a = m.atoms[i];
    for (int k = 0; k < a.nb; ++k) {
      const MBond& b = m.bonds[a.bond[k]];
      if (b.order == 0 && ks.free_[b.other(i)]) ++opts;
    }
    if (opts < best_n) {
      best = i;
      best_n = opts;
      if (!opts) return false;
      if (opts == 1) break;
    }
  }
  const MAtom& a = m.atoms[best];
  for (int k = 0; k < a.nb; ++k) {
    const int bi = a.bond[k];
    const MBond& b = m.bonds[bi];
    const int j = b.other(best);
    if (b.order != 0 || !ks.free_[j]) continue;
    ks.free_[best] = ks.free_[j] = 0;
    ks.chosen.push_back(bi);
    if (kekule_solve(m, ks, n_free - 2)) return true;
    ks.chosen.pop_back();
    ks.free_[best] = ks.free_[j] = 1;
  }
  return false;
}

bool kekulize(Mol& m, KekuleScratch& ks) {
  const int n = static_cast<int>(m.atoms.size());
  for (int i = 0; i < n; ++i) {
    const MAtom& a = m.atoms[i];
    if (!a.written_arom) continue;
    bool in_ring = false;
    for (int k = 0; k < a.nb; ++k) in_ring |= m.bonds[a.bond[k]].ring;
    if (!in_ring) return m.fail("non-ring atom marked aromatic");
  }
  bool any = false;
  for (MBond& b : m.bonds) {
    if (b.order == 0 && !b.ring) {
      b.order = 1;
      b.written_arom = false;
    }
    any |= b.order == 0;
  }
  if (!any) return true;
  ks.needs.assign(n, 0);
  ks.list.clear();
  for (int i = 0; i < n; ++i) {
    const MAtom& a = m.atoms[i];
    int sigma = a.ex_h;
    bool has = false;
    for (int k = 0; k < a.nb; ++k) {
      const MBond& b = m.bonds[a.bond[k]];
      has |= b.order == 0;
      sigma += b.order == 0 ? 1 : b.order;
    }
    if (!has) continue;
    const int target = fitting_valence(a, sigma);
    if (target >= 0 && target - sigma >= 1) {
      ks.needs[i] = 1;
      ks.list.push_back(i);
    }
  }
  ks.free_ = ks.needs;
  ks.chosen.clear();
  ks.budget = 2000000;
  if (ks.list.size() % 2 || !kekule_solve(m, ks, static_cast<int>(ks.list.size())))
    return m.fail("cannot kekulize");
  for (MBond& b : m.bonds)
    if (b.order == 0) b.order = 1;
  for (int bi : ks.chosen) m.bonds[bi].order = 2;
  return true;
}

// --------------------------------------------------------------------------------------- valence, radicals

bool assign_valence(Mol& m) {
  const int n = static_cast<int>(m.atoms.size());
  for (int i = 0; i < n; ++i) {
    MAtom& a = m.atoms[i];
    const int ev = m.explicit_valence(i);
    const int nv = kT.n_valences(a.z);
    const int last = kT.valences[a.z][nv - 1];
    const int effective = kT.n_outer[a.z] >= 4 ? ev - a.charge : ev + a.charge;
    if (last > 0 && effective > last) return m.fail("valence too high");
    a.imp_h = 0;
    if (a.bracket) continue;
    const int target = fitting_valence(a, ev);
    if (target < 0) {
      if (last == -1) continue;
      return m.fail("valence too high");
    }
    a.imp_h = static_cast<uint8_t>(target - ev);
  }
  return true;
}

void assign_radicals(Mol& m) {
  const int n = static_cast<int>(m.atoms.size());
  for (int i = 0; i < n; ++i) {
    MAtom& a = m.atoms[i];
    a.rad = 0;
    if (!a.bracket) continue;
    const int nv = kT.n_valences(a.z);
    if (nv == 1 && kT.valences[a.z][0] == -1) continue;
    const int n_outer = kT.n_outer[a.z];
    const int total = m.explicit_valence(i);
    const int base = a.z <= 2 ? 2 : 8;
    int late = base - n_outer - total + a.charge;
    if (late < 0) {
      late = 0;
      if (nv > 1) {
        for (int k = 0; k < nv; ++k) {
          const int v = kT.valences[a.z][k];
          if (v - total + a.charge >= 0) { late = v - total + a.charge; break; }
        }
      }
    }
    const int early = n_outer - total - a.charge;
    if (early >= 0) late = std::min(late, early);
    a.rad = static_cast<uint8_t>(std::max(0, std::min(late, 255)));
  }
}

// ------------------------------------------------------------------------------------------- aromaticity

int count_atom_elec(const Mol& m, int i) {
  const MAtom& a = m.atoms[i];
  const int dv = kT.default_valence(a.z);
  if (dv <= 1) return 0;
  const int degree = m.degree(i) + m.total_h(i);
  if (degree > 3) return -1;
  const int nlp = std::max(kT.n_outer[a.z] - dv - a.charge, 0);
  int res = (dv - degree) + nlp - a.rad;
  if (res > 1 && m.explicit_valence(i) - m.degree(i) > 1) res = 1;
  return res;
}

bool more_electronegative(int z1, int z2) {
  const int n1 = kT.n_outer[z1], n2 = kT.n_outer[z2];
  return n1 > n2 || (n1 == n2 && z1 < z2);
}

Donor donor_type(const Mol& m, int i) {
  const MAtom& a = m.atoms[i];
  int nelec = count_atom_elec(m, i);
  int exo = -1;
  bool cyc = false, multiple = false;
  for (int k = 0; k < a.nb; ++k) {
    const MBond& b = m.bonds[a.bond[k]];
    if (b.order >= 2) {
      multiple = true;
      if (b.ring) cyc = true;
      else if (exo < 0) exo = b.other(i);
    }
  }
  if (nelec < 0) return D_NONE;
  if (nelec == 0) {
    if (exo >= 0) return D_VACANT;
    if (cyc) return D_ONE;
    return D_NONE;
  }
  if (nelec == 1) {
    if (exo >= 0) return more_electronegative(m.atoms[exo].z, a.z) ? D_VACANT : D_ONE;
    if (multiple) return D_ONE;
    if (a.charge == 1) return D_VACANT;
    return D_NONE;
  }
  if (exo >= 0 && more_electronegative(m.atoms[exo].z, a.z)) --nelec;
  return (nelec % 2) ? D_ONE : D_TWO;
}

bool arom_candidate(const Mol& m, int i, Donor d) {
  const MAtom& a = m.atoms[i];
  if (a.z > 18 && a.z != 34 && a.z != 52) return false;
  if (d == D_NONE) return false;
  const int dv = kT.default_valence(a.z);
  const int zc = a.z - a.charge;
  const int dvc = (zc >= 1 && zc <= kNumElements) ? kT.default_valence(zc) : -1;
  if (dv > 0 && m.total_valence(i) > dvc) return false;
  int n_mult = 0;
  for (int k = 0; k < a.nb; ++k) n_mult += m.bonds[a.bond[k]].order >= 2;
  if (m.explicit_valence(i) - m.degree(i) > 1 && n_mult > 1) return false;
  return true;
}

struct AromScratch {
  std::vector<uint8_t> donor, cand, done, in_combo;
  std::vector<int> ring_ids, count, touched, atoms_list, seen_atoms;
  std::vector<std::vector<int>> fused;
};

bool huckel(const std::vector<int>& atoms, const std::vector<uint8_t>& donor) {
  int n = 0;
  for (int x : atoms) n += donor[x] == D_ONE ? 1 : (donor[x] == D_TWO ? 2 : 0);
  if (n >= 6) return (n - 2) % 4 == 0;
  return n == 2;
}

void try_combo(Mol& m, AromScratch& as, const int* combo, int size) {
  as.touched.clear();
  for (int c = 0; c < size; ++c)
    for (int b : m.rings[as.ring_ids[combo[c]]].bonds) {
      if (as.count[b]++ == 0) as.touched.push_back(b);
    }
  as.atoms_list.clear();
  for (int b : as.touched) {
    if (as.count[b] != 1) continue;
    for (int x : {m.bonds[b].a, m.bonds[b].b}) {
      if (!as.seen_atoms[x]) {
        as.seen_atoms[x] = 1;
        as.atoms_list.push_back(x);
      }
    }
  }
  if (huckel(as.atoms_list, as.donor)) {
    for (int b : as.touched)
      if (as.count[b] == 1) m.bonds[b].arom = true;
    for (int x : as.atoms_list) m.atoms[x].arom = true;
    for (int c = 0; c < size; ++c) as.done[combo[c]] = 1;
  }
  for (int x : as.atoms_list) as.seen_atoms[x] = 0;
  for (int b : as.touched) as.count[b] = 0;
}

void perceive_aromaticity(Mol& m, AromScratch& as) {
  const int n = static_cast<int>(m.atoms.size());
  for (MAtom& a : m.atoms) a.arom = false;
  for (MBond& b : m.bonds) b.arom = false;
  if (m.rings.empty()) return;
  as.donor.resize(n);
  as.cand.resize(n);
  for (int i = 0; i < n; ++i) {
    as.donor[i] = donor_type(m, i);
    as.cand[i] = arom_candidate(m, i, static_cast<Donor>(as.donor[i]));
  }
  as.ring_ids.clear();
  for (size_t r = 0; r < m.rings.size(); ++r) {
    bool ok = true;
    for (int x : m.rings[r].atoms) ok &= as.cand[x] != 0;
    if (ok) as.ring_ids.push_back(static_cast<int>(r));
  }
  const int R = static_cast<int>(as.ring_ids.size());
  if (!R) return;
  as.count.assign(m.bonds.size(), 0);
  as.seen_atoms.assign(n, 0);
  as.done.assign(R, 0);
  as.fused.assign(R, std::vector<int>());
  // rings sharing a bond
  for (int i = 0; i < R; ++i) {
    for (int b : m.rings[as.ring_ids[i]].bonds) as.count[b] = 1;
    for (int j = 0; j < R; ++j) {
      if (i == j) continue;
      bool share = false;
      for (int b : m.rings[as.ring_ids[j]].bonds) share |= as.count[b] != 0;
      if (share) as.fused[i].push_back(j);
    }
    for (int b : m.rings[as.ring_ids[i]].bonds) as.count[b] = 0;
  }
  int n_done = 0;
  auto all_done = [&]() {
    n_done = 0;
    for (int i = 0; i < R; ++i) n_done += as.done[i];
    return n_done == R;
  };
  int combo[3];
  for (int i = 0; i < R; ++i) {
    combo[0] = i;
    try_combo(m, as, combo, 1);
  }
  if (all_done()) return;
  auto adjacent = [&](int i, int j) {
    return std::find(as.fused[i].begin(), as.fused[i].end(), j) != as.fused[i].end();
  };
  for (int i = 0; i < R; ++i)
    for (int j = i + 1; j < R; ++j) {
      if (!adjacent(i, j)) continue;
      if (as.done[i] && as.done[j]) continue;
      combo[0] = i; combo[1] = j;
      try_combo(m, as, combo, 2);
    }
  if (all_done()) return;
  for (int i = 0; i < R; ++i)
    for (int j = i + 1; j < R; ++j)
      for (int k = j + 1; k < R; ++k) {
        const int links = adjacent(i, j) + adjacent(i, k) + adjacent(j, k);
        if (links < 2) continue;  // three rings are connected iff at least two of the pairs touch
        if (as.done[i] && as.done[j] && as.done[k]) continue;
        combo[0] = i; combo[1] = j; combo[2] = k;
        try_combo(m, as, combo, 3);
      }
}

// --------------------------------------------------------------------------- conjugation, hybridisation

void mark_conjugation(Mol& m) {
  const int n = static_cast<int>(m.atoms.size());
  for (MBond& b : m.bonds) b.conj = b.arom;
  for (int i = 0; i < n; ++i) {
    const MAtom& a = m.atoms[i];
    const int sbo = m.degree(i) + m.total_h(i);
    if (sbo < 2 || sbo > 3) continue;
    for (int k1 = 0; k1 < a.nb; ++k1) {
      MBond& b1 = m.bonds[a.bond[k1]];
      if (!(b1.arom || b1.order >= 2)) continue;  // valence contribution below 1.5
      for (int k2 = 0; k2 < a.nb; ++k2) {
        if (k1 == k2) continue;
        MBond& b2 = m.bonds[a.bond[k2]];
        const int j = b2.other(i);
        if (m.degree(j) + m.total_h(j) > 3) continue;
        const int zj = m.atoms[j].z;
        const int no = kT.n_outer[zj];
        if ((zj <= 10 || (no != 5 && no != 6)) && count_atom_elec(m, j) > 0) {
          b1.conj = true;
          b2.conj = true;
        }
      }
    }
  }
}

void set_hybridization(Mol& m) {
  const int n = static_cast<int>(m.atoms.size());
  for (int i = 0; i < n; ++i) {
    MAtom& a = m.atoms[i];
    const int deg = m.degree(i) + m.total_h(i);
    int norbs;
    if (a.z <= 1) {
      norbs = deg;
    } else {
      const int no = kT.n_outer[a.z];
      const int tv = m.total_valence(i);
      const int free_e = no - (tv + a.charge);
      if (tv + no - a.charge < 8) norbs = deg + (free_e - a.rad) / 2 + a.rad;
      else norbs = deg + free_e / 2;
    }
    uint8_t h;
    if (norbs <= 1) h = HYB_S;
    else if (norbs == 2) h = HYB_SP;
    else if (norbs == 3) h = HYB_SP2;
    else if (norbs == 4) {
      bool conj = false;
      for (int k = 0; k < a.nb; ++k) conj |= m.bonds[a.bond[k]].conj;
      h = (deg > 3 || !conj) ? HYB_SP3 : HYB_SP2;
    } else if (norbs == 5) h = HYB_SP3D;
    else if (norbs == 6) h = HYB_SP3D2;
    else h = HYB_UNSPECIFIED;
    a.hyb = h;
  }
}

struct Scratch {
  Mol mol;
  std::vector<int> v0, v1, v2, v3, v4;
  RingScratch rs;
  KekuleScratch ks;
  AromScratch as;
  std::vector<int> bfs, bfs_dist;
};

// validate_only: stop once it is known that the molecule can be featurized (sizes pass)
bool mol_from_smiles(const char* smiles, Scratch& s, bool validate_only = false) {
  Mol& m = s.mol;
  if (!smiles || !parse(smiles, m)) return false;
  remove_explicit_hydrogens(m, s.v0);
  for (const MAtom& a : m.atoms)
    if (a.nb > GCMI_MAX_DEG) return m.fail("atom degree above the largest degree table");
  cleanup(m);
  mark_ring_bonds(m, s.v0, s.v1, s.v2, s.v3, s.v4);
  if (!kekulize(m, s.ks)) return false;
  if (!assign_valence(m)) return false;
  if (validate_only) return true;
  assign_radicals(m);
  find_rings(m, s.rs);
  perceive_aromaticity(m, s.as);
  mark_conjugation(m);
  set_hybridization(m);
  return true;
}

// --------------------------------------------------------------------------------------------- features

inline void one_hot_unk(float* out, int n, int x) { out[(x >= 0 && x < n) ? x : n - 1] = 1.f; }

void write_atom_features(const Mol& m, int i, float* out) {
  const MAtom& a = m.atoms[i];
  memset(out, 0, sizeof(float) * GCMI_ATOM_FEATURES);
  out[kT.symbol_col[a.z]] = 1.f;
  out[44 + std::min<int>(m.degree(i), 10)] = 1.f;  // degrees above 10 cannot be collated anyway
  one_hot_unk(out + 55, 7, a.imp_h);
  out[62] = static_cast<float>(a.charge);
  out[63] = static_cast<float>(a.rad);
  int hcol = 4;
  switch (a.hyb) {
    case HYB_SP: hcol = 0; break;
    case HYB_SP2: hcol = 1; break;
    case HYB_SP3: hcol = 2; break;
    case HYB_SP3D: hcol = 3; break;
    default: hcol = 4; break;
  }
  out[64 + hcol] = 1.f;
  out[69] = a.arom ? 1.f : 0.f;
  one_hot_unk(out + 70, 5, m.total_h(i));
}

void write_bond_features(const MBond& b, float* out) {
  out[0] = (!b.arom && b.order == 1) ? 1.f : 0.f;
  out[1] = (!b.arom && b.order == 2) ? 1.f : 0.f;
  out[2] = (!b.arom && b.order == 3) ? 1.f : 0.f;
  out[3] = b.arom ? 1.f : 0.f;
  out[4] = b.conj ? 1.f : 0.f;
  out[5] = b.ring ? 1.f : 0.f;
}

void write_pair_features(const Mol& m, Scratch& s, float* out) {
  const int n = static_cast<int>(m.atoms.size());
  const int F = GCMI_PAIR_FEATURES;
  memset(out, 0, sizeof(float) * static_cast<size_t>(n) * n * F);
  for (const MBond& b : m.bonds) {
    write_bond_features(b, out + (static_cast<size_t>(b.a) * n + b.b) * F);
    write_bond_features(b, out + (static_cast<size_t>(b.b) * n + b.a) * F);
  }
  for (const Ring& r : m.rings)
    for (int x : r.atoms)
      for (int y : r.atoms)
        if (x != y) out[(static_cast<size_t>(x) * n + y) * F + 6] = 1.f;
  s.bfs_dist.assign(n, -1);
  for (int a1 = 0; a1 < n; ++a1) {
    s.bfs.clear();
    s.bfs.push_back(a1);
    s.bfs_dist[a1] = 0;
    for (size_t h = 0; h < s.bfs.size(); ++h) {
      const int u = s.bfs[h];
      if (s.bfs_dist[u] >= 7) continue;
      for (int k = 0; k < m.atoms[u].nb; ++k) {
        const int v = m.bonds[m.atoms[u].bond[k]].other(u);
        if (s.bfs_dist[v] < 0) {
          s.bfs_dist[v] = s.bfs_dist[u] + 1;
          s.bfs.push_back(v);
          out[(static_cast<size_t>(a1) * n + v) * F + 7 + s.bfs_dist[v] - 1] = 1.f;
        }
      }
    }
    for (int u : s.bfs) s.bfs_dist[u] = -1;
  }
}

template <typename F>
void parallel_molecules(int64_t n, int n_threads, F&& fn) {
  if (n_threads <= 1 || n < 64) {
    fn(0, n);
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t) th.emplace_back(fn, n * t / n_threads, n * (t + 1) / n_threads);
  for (auto& x : th) x.join();
}

}  // namespace

extern "C" {

int gcmi_smiles_sizes(const char* const* smiles, int64_t n, int32_t* n_atoms, int32_t* n_bonds, int n_threads) {
  GCMI_CHECK_ARG((smiles || n == 0) && n >= 0 && n_atoms && n_bonds, "smiles_sizes: NULL argument");
  parallel_molecules(n, n_threads, [&](int64_t lo, int64_t hi) {
    Scratch s;
    for (int64_t i = lo; i < hi; ++i) {
      if (mol_from_smiles(smiles[i], s, true)) {
        n_atoms[i] = static_cast<int32_t>(s.mol.atoms.size());
        n_bonds[i] = static_cast<int32_t>(s.mol.bonds.size());
      } else {
        n_atoms[i] = -1;
        n_bonds[i] = 0;
      }
    }
  });
  return GCMI_OK;
}

int gcmi_smiles_featurize(const char* const* smiles, int64_t n, const int64_t* atom_off, const int64_t* bond_off,
                          const int64_t* pair_off, float* atom_features, int32_t* adj_degree, int32_t* adj_idx,
                          int32_t* bond_atoms, float* bond_features, float* pair_features, int32_t* atom_props,
                          int n_threads) {
  GCMI_CHECK_ARG((smiles || n == 0) && n >= 0 && atom_off && bond_off, "smiles_featurize: NULL argument");
  GCMI_CHECK_ARG(!pair_features || pair_off, "smiles_featurize: pair_features without pair_off");
  int bad = 0;
  parallel_molecules(n, n_threads, [&](int64_t lo, int64_t hi) {
    Scratch s;
    for (int64_t i = lo; i < hi; ++i) {
      const int64_t na = atom_off[i + 1] - atom_off[i];
      const int64_t nbd = bond_off[i + 1] - bond_off[i];
      if (na == 0 && nbd == 0) continue;  // a molecule gcmi_smiles_sizes rejected
      if (!mol_from_smiles(smiles[i], s) || static_cast<int64_t>(s.mol.atoms.size()) != na ||
          static_cast<int64_t>(s.mol.bonds.size()) != nbd) {
        __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED);
        continue;
      }
      const Mol& m = s.mol;
      const int64_t a0 = atom_off[i], b0 = bond_off[i];
      for (int a = 0; a < na; ++a) {
        if (atom_features) write_atom_features(m, a, atom_features + (a0 + a) * GCMI_ATOM_FEATURES);
        if (adj_degree) adj_degree[a0 + a] = m.atoms[a].nb;
        if (atom_props) {
          int32_t* p = atom_props + (a0 + a) * 8;
          const MAtom& A = m.atoms[a];
          p[0] = A.z; p[1] = A.nb; p[2] = A.imp_h; p[3] = A.ex_h; p[4] = A.charge; p[5] = A.rad; p[6] = A.hyb;
          p[7] = A.arom;
        }
      }
      if (adj_idx) {
        // neighbours in bond order: what appending both directions bond by bond gives (graph_features.py:897-904)
        int32_t* out = adj_idx + 2 * b0;
        for (int a = 0; a < na; ++a)
          for (int k = 0; k < m.atoms[a].nb; ++k) *out++ = m.bonds[m.atoms[a].bond[k]].other(a);
      }
      for (int b = 0; b < nbd; ++b) {
        if (bond_atoms) {
          bond_atoms[(b0 + b) * 2] = m.bonds[b].a;
          bond_atoms[(b0 + b) * 2 + 1] = m.bonds[b].b;
        }
        if (bond_features) write_bond_features(m.bonds[b], bond_features + (b0 + b) * GCMI_BOND_FEATURES);
      }
      if (pair_features) write_pair_features(m, s, pair_features + pair_off[i] * GCMI_PAIR_FEATURES);
    }
  });
  if (bad) {
    ::gcmi::set_error("smiles_featurize: %d molecule(s) do not match the sizes passed in", bad);
    return GCMI_ERR_ARG;
  }
  return GCMI_OK;
}

const char* gcmi_smiles_check(const char* smiles) {
  static thread_local Scratch s;
  if (mol_from_smiles(smiles, s, true)) return nullptr;
  return s.mol.error ? s.mol.error : "cannot read SMILES";
}

}  // extern "C"
