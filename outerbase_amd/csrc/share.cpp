// Shared sub-products for the term-per-lane kernels: the "star" tables of a term set.
//
// The reference multiplies every term out on its own (prodmm_ / tprodmm_, src/linalg.cpp:57-131,
// 286-355: one pass over a term's non-zero levels per term), and so did the term-per-lane kernels
// of rounds 1-4: W LDS column reads per (term, row).  But selectterms' output is downward-closed
// (src/modandbase.cpp:419-436: a candidate is pushed only when all its parents are selected), so
// the terms come in large families that differ in ONE factor.  A *star* is four terms that share
// all factors but one: the lane that owns it reads the P shared factors once, keeps their product
// in a register and reads only the four distinct factors -- P + 4 reads and P + 3 multiplies for
// four terms instead of 4 (P + 1) reads and 4 P multiplies.
//
// Host side (here): partition the p_pad terms into p_pad / 4 stars.
//   * every term t with factors F_t is a child of the |F_t| families F_t \ {f}; families are taken
//     greedily (longest shared part first, then the family that leaves the fewest members over),
//     whole groups of four at a time, members that do not fill a group stay free for their other
//     families;
//   * what is left over (no family with four free members; any term set, downward-closed or not)
//     is listed (k_star multiplies those few terms out in its middle step) and also packed four
//     at a time into *plain* stars -- no shared part, S column slots per term -- behind the family
//     stars, for kernels that take whole star-waves only;
//   * family stars are ordered by falling number of shared factors and cut into star-waves of 64
//     (one per wave), filled up with empty stars; a star-wave's shape (P, S) is the maximum over its
//     members (shorter prefixes are padded with the ones column at the front);
//   * a seeded local search then places the stars (half-wave, order of the four terms) and numbers
//     the used columns for few LDS bank conflicts (below).
// Device side: TlStar in device_common.h runs the read pipeline of a shape; the kernels pick the
// instantiation per star-wave by a wave-uniform branch.
//
// At the headline term set (d = 20, p = 4096, 12 177 factors in all) a block issues 97 column
// reads per row for the 4052 family terms + 16 for the plain star-wave of the 44 left-over ones,
// instead of 194; d = 8 mat25pow with six-factor terms 100 + 24 instead of 224; BASELINE
// configs[4]'s 16 384 terms 384 + 16 instead of 768 (obhip_terms_share_tables reports these numbers
// for any term set; tests/test_host_logic.py checks the tables themselves).
#include <algorithm>
#include <map>
#include <queue>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "obhip_internal.h"

namespace obhip {

namespace {

using Key = std::vector<uint16_t>;  // sorted used-column indices (the ones column left out)

struct KeyHash {
  size_t operator()(const Key &k) const {
    uint64_t h = 0x9e3779b97f4a7c15ull ^ k.size();
    for (uint16_t v : k) h = (h ^ v) * 0x100000001b3ull + 0x9e3779b97f4a7c15ull;
    return (size_t)(h ^ (h >> 29));
  }
};

struct Star {
  int P = 0;             // shared factors (0 for a plain star)
  bool plain = false;
  Key prefix;            // the shared factors, ascending
  uint32_t term[4];      // term indices (slots of the caller's order)
  int nnz_max = 0;       // plain: most factors of a member
};

// ---- LDS bank conflicts ---------------------------------------------------------------------
// The tile is [column][65 doubles]: at a given row, column u lies in bank pair u mod 32, a
// ds_read_b64 serves the lanes 0-31 and 32-63 in one LDS cycle each when, within the half, equal
// columns aside (a broadcast), no two columns share a bank pair; every further column on a busy
// pair costs a cycle.  With the terms in selection order (rounds 1-4) neighbouring lanes mostly
// read the same column (1.05 cycles per half); the own factors of a star-wave's 64 stars are all
// over the place (1.4).  So, after the stars are formed: a seeded local search over (i) which
// half-wave a star sits in -- stars are exchanged only with stars of the same shape, so the
// star-waves keep their shapes -- and (ii) the order of a star's four terms, minimising the LDS
// cycles of one row summed over all reads of all star-waves.
struct Layout {
  uint64_t W, nst;
  const std::vector<Key> *fac;
  std::vector<Star> *stars;
  std::vector<int> wP, wS;          // shape per star-wave
  std::vector<uint16_t> sc;         // nst x 4 W: col(i, j) of the current arrangement (refresh)
  std::vector<uint16_t> relabel;    // column -> the index it gets in the LDS tile (its bank pair: & 31)
  void refresh(uint64_t i) {
    for (int j = 0; j < slots(i / 64); ++j) sc[i * 4 * W + j] = col(i, j);
  }
  // column read by star i in slot j of its star-wave's shape
  uint16_t col(uint64_t i, int j) const {
    const Star &s = (*stars)[i];
    const int P = wP[i / 64], S = wS[i / 64];
    if (P > 0) {
      if (j < P) {
        const int e = j - (P - s.P);
        return e >= 0 ? s.prefix[e] : 0;
      }
      return own(s, j - P);
    }
    const int t = j / S, e = j % S;
    const Key &f = (*fac)[s.term[t]];
    const int k = e - (S - (int)f.size());
    return k >= 0 ? f[k] : 0;
  }
  uint16_t own(const Star &s, int t) const {  // the one factor of term t not in the shared part
    const Key &f = (*fac)[s.term[t]];
    uint16_t o = 0;
    size_t a = 0;
    for (size_t e = 0; e < f.size(); ++e) {
      if (a < s.prefix.size() && s.prefix[a] == f[e])
        ++a;
      else
        o = f[e];
    }
    return o;
  }
  int slots(uint64_t w) const { return wP[w] + 4 * wS[w]; }
  // LDS cycles of slot j over the 32 stars of half-wave h (stars 32 h .. 32 h + 31)
  int cycles(uint64_t h, int j) const {
    uint16_t seen[32][8];
    int cnt[32] = {0};
    int worst = 1;
    for (uint64_t i = 32 * h; i < 32 * h + 32; ++i) {
      const uint16_t c = sc[i * 4 * W + j];
      const int b = relabel[c] & 31;
      bool dup = false;
      for (int k = 0; k < cnt[b] && k < 8; ++k) dup = dup || seen[b][k] == c;
      if (dup) continue;
      if (cnt[b] < 8) seen[b][cnt[b]] = c;
      worst = std::max(worst, ++cnt[b]);
    }
    return worst;
  }
  int half_cost(uint64_t h) const {
    int t = 0;
    for (int j = 0; j < slots(h / 2); ++j) t += cycles(h, j);
    return t;
  }
};

uint64_t layout_cost(const Layout &L) {
  uint64_t t = 0;
  for (uint64_t h = 0; h < L.nst / 32; ++h) t += (uint64_t)L.half_cost(h);
  return t;
}

// Which tile index (hence bank pair) a column gets is the library's choice too: columns that are
// read by different lanes of one half-wave in one slot should sit on different bank pairs.  From
// the arrangement at hand: cooc[c][c'] = number of (half-wave, slot) reads in which both occur;
// columns by falling weight, each to the bank pair with room left where it meets the least
// co-occurrence (indices 1 .. ncol - 1 are dealt in order to the 32 bank pairs; 0, the ones column,
// stays).
void renumber_columns(Layout &L, uint64_t ncol) {
  std::vector<uint32_t> cooc(ncol * ncol, 0);
  std::vector<uint16_t> seen;
  for (uint64_t h = 0; h < L.nst / 32; ++h)
    for (int j = 0; j < L.slots(h / 2); ++j) {
      seen.clear();
      for (uint64_t i = 32 * h; i < 32 * h + 32; ++i) {
        const uint16_t c = L.sc[i * 4 * L.W + j];
        if (std::find(seen.begin(), seen.end(), c) == seen.end()) seen.push_back(c);
      }
      for (size_t a = 0; a < seen.size(); ++a)
        for (size_t b = a + 1; b < seen.size(); ++b) {
          cooc[(size_t)seen[a] * ncol + seen[b]] += 1;
          cooc[(size_t)seen[b] * ncol + seen[a]] += 1;
        }
    }
  std::vector<uint64_t> weight(ncol, 0);
  for (uint64_t c = 0; c < ncol; ++c)
    for (uint64_t e = 0; e < ncol; ++e) weight[c] += cooc[c * ncol + e];
  std::vector<uint32_t> order;
  for (uint64_t c = 1; c < ncol; ++c) order.push_back((uint32_t)c);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return weight[a] > weight[b]; });
  int cap[32];
  for (int b = 0; b < 32; ++b) cap[b] = 0;
  for (uint64_t l = 1; l < ncol; ++l) cap[l & 31] += 1;
  std::vector<std::vector<uint32_t>> members(32);
  members[0].push_back(0);  // the ones column
  for (uint32_t c : order) {
    int best = -1;
    uint64_t bestcost = 0;
    for (int b = 0; b < 32; ++b) {
      if (cap[b] == 0) continue;
      uint64_t cost = 0;
      for (uint32_t e : members[b]) cost += cooc[(size_t)c * ncol + e];
      if (best < 0 || cost < bestcost || (cost == bestcost && cap[b] > cap[best])) {
        best = b;
        bestcost = cost;
      }
    }
    members[best].push_back(c);
    cap[best] -= 1;
  }
  L.relabel.assign(ncol, 0);
  for (int b = 0; b < 32; ++b) {
    uint32_t next = b == 0 ? 32 : (uint32_t)b;  // (index 0 is taken by the ones column)
    for (uint32_t c : members[b]) {
      if (c == 0) continue;
      L.relabel[c] = (uint16_t)next;
      next += 32;
    }
  }
}

void reduce_conflicts(Layout &L, uint64_t moves_per_star) {
  std::vector<Star> &st = *L.stars;
  const uint64_t nh = L.nst / 32;
  // classes of exchangeable stars: same kind, same number of shared factors / of slots needed
  auto cls = [&](const Star &s) { return s.plain ? 100 + s.nnz_max : s.P; };
  std::map<int, std::vector<uint32_t>> members;
  for (uint64_t i = 0; i < L.nst; ++i) members[cls(st[i])].push_back((uint32_t)i);
  uint64_t rng = 0x243f6a8885a308d3ull;
  auto next = [&]() {
    rng += 0x9e3779b97f4a7c15ull;
    uint64_t z = rng;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  };
  std::vector<int> hc(nh);
  for (uint64_t h = 0; h < nh; ++h) hc[h] = L.half_cost(h);
  const uint64_t moves = std::min<uint64_t>(400000, moves_per_star * L.nst);
  for (uint64_t it = 0; it < moves; ++it) {
    const uint64_t r = next();
    const uint64_t i = (r >> 8) % L.nst;
    if ((r & 3) == 0) {  // reorder the terms of star i
      const int a = (int)((r >> 40) & 3), b = (int)((r >> 44) & 3);
      if (a == b) continue;
      const uint64_t h = i / 32;
      std::swap(st[i].term[a], st[i].term[b]);
      L.refresh(i);
      const int c = L.half_cost(h);
      if (c <= hc[h]) {
        hc[h] = c;
      } else {
        std::swap(st[i].term[a], st[i].term[b]);
        L.refresh(i);
      }
    } else {  // exchange star i with another star of its class in a different half-wave
      const std::vector<uint32_t> &m = members[cls(st[i])];
      const uint64_t k = m[(r >> 36) % m.size()];
      const uint64_t hi = i / 32, hk = k / 32;
      if (hi == hk) continue;
      std::swap(st[i], st[k]);
      L.refresh(i);
      L.refresh(k);
      const int ci = L.half_cost(hi), ck = L.half_cost(hk);
      if (ci + ck <= hc[hi] + hc[hk]) {
        hc[hi] = ci;
        hc[hk] = ck;
      } else {
        std::swap(st[i], st[k]);
        L.refresh(i);
        L.refresh(k);
      }
    }
  }
}

}  // namespace

bool share_wanted() {
  static const bool off = getenv("OBHIP_SHARE") && atoi(getenv("OBHIP_SHARE")) == 0;
  return !off;
}

// hc: p_pad x W used-column indices per term (0 = the ones column; obhip_terms::prepare's table)
// renumber: also choose the columns' indices (out.relabel: old -> new; the tables are written in
// the new indices, the caller renumbers its own column lists)
int build_share_tables(const uint16_t *hc, uint64_t p_pad, uint64_t W, ShareTables &out, bool renumber,
                       uint64_t ncol_in) {
  out = ShareTables();
  if (p_pad == 0 || p_pad % 256 != 0 || W < 2 || W > 8 || W % 2) return 0;  // (not shareable: ok stays false)
  std::vector<Key> fac(p_pad);
  for (uint64_t k = 0; k < p_pad; ++k) {
    for (uint64_t j = 0; j < W; ++j)
      if (hc[k * W + j]) fac[k].push_back(hc[k * W + j]);
    std::sort(fac[k].begin(), fac[k].end());
  }
  // families: shared part -> members.  A term without factors (the constant term, the padding
  // terms) is a child of the empty family with the ones column as its own factor.
  std::unordered_map<Key, std::vector<uint32_t>, KeyHash> fam;
  fam.reserve(p_pad * 3);
  for (uint64_t k = 0; k < p_pad; ++k) {
    const Key &f = fac[k];
    if (f.empty()) {
      fam[Key()].push_back((uint32_t)k);
      continue;
    }
    for (size_t drop = 0; drop < f.size(); ++drop) {
      if (drop > 0 && f[drop] == f[drop - 1]) continue;  // (a repeated factor: one family)
      Key q;
      q.reserve(f.size() - 1);
      for (size_t j = 0; j < f.size(); ++j)
        if (j != drop) q.push_back(f[j]);
      fam[q].push_back((uint32_t)k);
    }
  }
  // greedy: (longest shared part, fewest left over, largest) first; lazily re-keyed
  using Prio = std::tuple<int, int, int, const Key *>;  // (-|q|, free % 4, -free, key)
  auto cmp = [](const Prio &a, const Prio &b) {
    if (std::get<0>(a) != std::get<0>(b)) return std::get<0>(a) > std::get<0>(b);
    if (std::get<1>(a) != std::get<1>(b)) return std::get<1>(a) > std::get<1>(b);
    if (std::get<2>(a) != std::get<2>(b)) return std::get<2>(a) > std::get<2>(b);
    return *std::get<3>(a) > *std::get<3>(b);  // deterministic whatever the hash order
  };
  std::priority_queue<Prio, std::vector<Prio>, decltype(cmp)> pq(cmp);
  for (auto &kv : fam)
    if (kv.second.size() >= 4)
      pq.push(Prio(-(int)kv.first.size(), (int)(kv.second.size() % 4), -(int)kv.second.size(), &kv.first));
  std::vector<char> taken(p_pad, 0);
  std::vector<Star> stars;
  stars.reserve(p_pad / 4 + 128);
  std::vector<uint32_t> fr;
  while (!pq.empty()) {
    const Prio top = pq.top();
    pq.pop();
    const Key &q = *std::get<3>(top);
    std::vector<uint32_t> &mem = fam[q];
    fr.clear();
    for (uint32_t k : mem)
      if (!taken[k]) fr.push_back(k);
    if (fr.size() < 4) continue;
    if (-(int)fr.size() != std::get<2>(top)) {  // members went to other families meanwhile
      pq.push(Prio(-(int)q.size(), (int)(fr.size() % 4), -(int)fr.size(), &q));
      continue;
    }
    // members in the order of their own factor, so that neighbouring stars read neighbouring columns
    std::sort(fr.begin(), fr.end(), [&](uint32_t a, uint32_t b) {
      if (fac[a] != fac[b]) return fac[a] < fac[b];
      return a < b;
    });
    for (size_t i = 0; i + 4 <= fr.size(); i += 4) {
      Star s;
      s.P = (int)q.size();
      s.prefix = q;
      for (int j = 0; j < 4; ++j) {
        s.term[j] = fr[i + j];
        taken[fr[i + j]] = 1;
      }
      stars.push_back(std::move(s));
    }
  }
  // What is left over (no family with four free members): kept as a list -- k_hm3 multiplies those
  // few terms out in its middle step, lane = (row, term) -- and, for the kernels that take whole
  // star-waves only, four at a time in *plain* stars behind the family stars.
  std::vector<uint32_t> rest;
  for (uint64_t k = 0; k < p_pad; ++k)
    if (!taken[k]) rest.push_back((uint32_t)k);
  std::stable_sort(rest.begin(), rest.end(),
                   [&](uint32_t a, uint32_t b) { return fac[a].size() > fac[b].size(); });
  out.nleft = rest.size();
  out.left_term.assign(rest.begin(), rest.end());
  out.left_cols.assign(rest.size() * W, 0);
  for (size_t i = 0; i < rest.size(); ++i) {
    const Key &f = fac[rest[i]];
    for (size_t e = 0; e < f.size(); ++e) out.left_cols[i * W + (W - f.size()) + e] = f[e];  // (relabelled below)
  }
  // family stars by falling number of reads, equal shapes by shared part (the lanes of a wave then
  // read the same shared columns: LDS broadcasts); star-waves filled up with empty stars of
  // virtual terms (no factors, index >= p_pad: the kernels skip them)
  std::stable_sort(stars.begin(), stars.end(), [&](const Star &a, const Star &b) {
    if (a.P != b.P) return a.P > b.P;
    return a.prefix < b.prefix;
  });
  uint32_t next_virtual = (uint32_t)p_pad;
  auto empty_star = [&](bool plain) {
    Star s;
    s.plain = plain;
    for (int j = 0; j < 4; ++j) s.term[j] = next_virtual++;
    return s;
  };
  while (stars.size() % 64) stars.push_back(empty_star(false));
  const uint64_t nswf = stars.size() / 64;
  for (size_t i = 0; i < rest.size(); i += 4) {
    Star s;
    s.plain = true;
    for (int j = 0; j < 4; ++j) {
      s.term[j] = i + j < rest.size() ? rest[i + j] : next_virtual++;
      if (i + j < rest.size()) s.nnz_max = std::max<int>(s.nnz_max, (int)fac[rest[i + j]].size());
    }
    stars.push_back(std::move(s));
  }
  while (stars.size() % 64) stars.push_back(empty_star(true));
  fac.resize(next_virtual);  // (the virtual terms: no factors)
  const uint64_t nst = stars.size();
  const uint64_t NA = 4 * W;
  const uint64_t nsw = nst / 64;
  out.nsw_family = nswf;
  out.nsw_plain = nsw - nswf;
  out.cols.assign(nst * NA, 0);
  out.term.resize(nst * 4);
  out.shape.resize(nsw);
  const int s_short = W >= 4 ? (int)W - 2 : (int)W;  // the two plain widths the kernels instantiate
  Layout L;
  L.W = W;
  L.nst = nst;
  L.fac = &fac;
  L.stars = &stars;
  L.wP.resize(nsw);
  L.wS.resize(nsw);
  for (uint64_t w = 0; w < nsw; ++w) {
    const bool plain = w >= nswf;
    int pmax = 1, nzmax = 1;
    for (uint64_t i = w * 64; i < (w + 1) * 64; ++i) {
      const Star &s = stars[i];
      pmax = std::max(pmax, s.P);
      nzmax = std::max(nzmax, s.nnz_max);
    }
    L.wP[w] = plain ? 0 : pmax;
    L.wS[w] = plain ? (nzmax > s_short ? (int)W : s_short) : 1;
    out.shape[w] = (uint32_t)L.wP[w] | ((uint32_t)L.wS[w] << 8);
    (plain ? out.reads_left : out.reads) += (uint64_t)(L.wP[w] + 4 * L.wS[w]);
  }
  // which half-wave a star sits in and the order of its terms: fewest LDS bank conflicts
  uint64_t ncol = std::max<uint64_t>(1, ncol_in);  // (the caller's column count, or what the terms use)
  for (uint64_t k = 0; k < p_pad; ++k)
    for (uint16_t c : fac[k]) ncol = std::max<uint64_t>(ncol, (uint64_t)c + 1);
  L.relabel.resize(ncol);
  for (uint64_t c = 0; c < ncol; ++c) L.relabel[c] = (uint16_t)c;
  L.sc.assign(nst * NA, 0);
  for (uint64_t i = 0; i < nst; ++i) L.refresh(i);
  out.lds_cycles0 = layout_cost(L);
  // (moves per star: OBHIP_SHARE_MOVES, default 30.  At the headline terms, LDS cycles per row /
  // host time of the tables: 10 moves 247 / 0.05 s, 30: 245 / 0.12 s, 60: 240 / 0.2 s, 150: 234 / 0.4 s,
  // from 311 unsearched; the kernels gain 0.4 % from 258 -> 231, a new term set pays the host time once)
  static const uint64_t mps = getenv("OBHIP_SHARE_MOVES") ? (uint64_t)std::max(0, atoi(getenv("OBHIP_SHARE_MOVES"))) : 30;
  // ... and which bank pair a column lies on (the caller renumbers its used columns accordingly):
  // a short search for the co-occurrence counts, the numbering, then the arrangement for the new banks
  if (renumber) {
    reduce_conflicts(L, std::min<uint64_t>(mps, 20));
    renumber_columns(L, ncol);
  }
  reduce_conflicts(L, mps);
  out.lds_cycles = layout_cost(L);
  out.relabel = L.relabel;
  // ad[0 .. P): the shared factors, right-aligned (ones in front); ad[P + j S .. P + (j + 1) S):
  // term j's own factor(s), right-aligned, ascending
  for (uint64_t i = 0; i < nst; ++i) {
    for (int j = 0; j < 4; ++j) out.term[i * 4 + j] = stars[i].term[j] < p_pad ? stars[i].term[j] : 0xffffffffu;
    for (int j = 0; j < L.slots(i / 64); ++j) out.cols[i * NA + j] = L.relabel[L.sc[i * NA + j]];
  }
  // what the nnz-sorted plain scheme of rounds 1-4 issues at 4 terms per lane (for the record)
  {
    std::vector<int> nz(p_pad);
    for (uint64_t k = 0; k < p_pad; ++k) nz[k] = (int)fac[k].size();
    std::sort(nz.begin(), nz.end(), std::greater<int>());
    for (uint64_t w = 0; w < p_pad / 256; ++w) {
      int a = 1, b = 1;
      for (uint64_t i = 0; i < 128; ++i) a = std::max(a, nz[w * 256 + i]);
      for (uint64_t i = 128; i < 256; ++i) b = std::max(b, nz[w * 256 + i]);
      const int lo = std::max(1, (int)W - 3);
      out.reads_plain += 2 * std::max(a, lo) + 2 * std::max(b, lo);
    }
  }
  for (uint16_t &c : out.left_cols) c = L.relabel[c];
  out.nstars = nst;
  out.ok = true;
  return 0;
}

}  // namespace obhip
