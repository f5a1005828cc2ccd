// Library plumbing of libobhip: error string, device/stream selection,
// per-kernel hipEvent profiling, the device view of a model (ModelDev) and the
// device tables of a `terms` matrix.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <map>
#include <mutex>

#include "obhip_internal.h"

namespace obhip {

static thread_local std::string g_err;
static thread_local hipStream_t g_stream = nullptr;

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
  char buf[512];
  std::snprintf(buf, sizeof buf, "HIP error %d (%s) in %s at %s:%d", (int)e,
                hipGetErrorString(e), what, file, line);
  g_err = buf;
  return e == hipErrorNoDevice ? OBHIP_ERR_NO_DEVICE : OBHIP_ERR_HIP;
}

int require_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(OBHIP_ERR_NO_DEVICE,
                "no HIP device visible: libobhip has no CPU fallback");
  }
  return 0;
}

hipStream_t cur_stream() { return g_stream; }
void set_cur_stream(hipStream_t s) { g_stream = s; }

// ---- device memory pool (see obhip_internal.h) ------------------------------------------
namespace {
struct PoolBlock {
  void *p;
  int device;
  hipStream_t stream;
};
std::mutex g_pool_mu;
std::multimap<size_t, PoolBlock> g_pool;
size_t g_pool_bytes = 0;
size_t pool_cap() {
  static const size_t cap = [] {
    const char *e = getenv("OBHIP_POOL_MB");
    if (e) return (size_t)std::max(0, atoi(e)) << 20;
    // an eighth of the device's memory (36 GB of the MI355X's 288), at least 8 GB: obfit frees
    // and re-allocates the gradient basis of its rows (6.5 GB at n = 1e6, 8 dimensions) on every
    // function evaluation, and a block of more than half the cap goes back to the driver
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) tot = 0;
    return std::max<size_t>((size_t)8192 << 20, tot / 8);
  }();
  return cap;
}
}  // namespace

void pool_trim() {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (auto &kv : g_pool) (void)hipFree(kv.second.p);
  g_pool.clear();
  g_pool_bytes = 0;
}

int pool_alloc(void **p, size_t bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto range = g_pool.equal_range(bytes);
    for (auto it = range.first; it != range.second; ++it)
      if (it->second.device == dev && it->second.stream == cur_stream()) {
        *p = it->second.p;
        g_pool.erase(it);
        g_pool_bytes -= bytes;
        return 0;
      }
  }
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    pool_trim();
    e = hipMalloc(p, bytes);
  }
  if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
  return 0;
}

void pool_free(void *p, size_t bytes, hipStream_t stream) {
  if (!p) return;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (bytes > pool_cap() / 2) {
    (void)hipFree(p);
    return;
  }
  g_pool.emplace(bytes, PoolBlock{p, dev, stream});
  g_pool_bytes += bytes;
  while (g_pool_bytes > pool_cap() && !g_pool.empty()) {  // largest first
    auto it = std::prev(g_pool.end());
    (void)hipFree(it->second.p);
    g_pool_bytes -= it->first;
    g_pool.erase(it);
  }
}


// ---- profiling ----------------------------------------------------------------
struct ProfEntry {
  uint64_t launches = 0;
  double total_ms = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};
std::atomic<uint64_t> g_host_syncs{0};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::map<std::string, ProfEntry> g_prof;

ProfScope::ProfScope(const char *n) : name(n) {
  if (!g_prof_on) return;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return;
  active = true;
  (void)hipEventRecord(e0, cur_stream());
}

ProfScope::~ProfScope() {
  if (!active) return;
  (void)hipEventRecord(e1, cur_stream());
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof[name].pending.emplace_back(e0, e1);
}

static void prof_drain(ProfEntry &pe) {
  for (auto &ev : pe.pending) {
    if (hipEventSynchronize(ev.second) == hipSuccess) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
        pe.launches += 1;
        pe.total_ms += ms;
      }
    }
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  pe.pending.clear();
}

// ---- ModelDev -------------------------------------------------------------------
int ModelDev::build(const obhip_model &m, const std::vector<int64_t> &cap_in) {
  if (!m.knots_set) return fail(OBHIP_ERR_STATE, "knots not set");
  const uint64_t d = m.d;
  // nothing changed since the last build (same model state, same level caps): the device tables
  // stand -- obhip_basis_rebuild on new rows or a repeated fit must not pay for them again
  if (dims.p && model_version == m.version && built_for == &m && dims_h.size() == d) {
    bool same = cap.size() == d;
    for (uint64_t l = 0; same && l < d; ++l) {
      int64_t c = cap_in.empty() ? (int64_t)m.m_of(l) - 1 : cap_in[l];
      c = std::max<int64_t>(0, std::min<int64_t>(c, (int64_t)m.m_of(l) - 1));
      same = c == cap[l];
    }
    if (same) return 0;
  }
  cap.resize(d);
  dims_h.resize(d);
  std::vector<double> hka(m.M()), hkb(m.M()), hkc(m.M());
  std::vector<double> hrot, htab;
  std::vector<int> table_dims;  // per dimension: itself when it gets interval tables, else -1
  uint64_t ccol = 1;  // compact column 0 is the all-ones column
  for (uint64_t l = 0; l < d; ++l) {
    const uint64_t ml = m.m_of(l), o = m.knotptst[l];
    int64_t c = cap_in.empty() ? (int64_t)ml - 1 : cap_in[l];
    if (c < 0) c = 0;
    if (c > (int64_t)ml - 1) c = (int64_t)ml - 1;
    cap[l] = c;
    DimDesc &D = dims_h[l];
    D.kind = m.kinds[l];
    D.m = (int)ml;
    D.koff = (int)o;
    D.ncol = (int)c + 1;
    D.ncolp = (D.ncol + 7) / 8 * 8;
    D.rotoff = (int)hrot.size();
    D.ccol0 = (int)ccol;
    D.tab = -1;
    D.gwin = 0;
    D.g0 = D.ginv = 0.0;
    ccol += (uint64_t)c;
    const double *hy = &m.hyp[m.hypst[l]];
    const double a = 2.0, b = 0.25;  // covfuncs.h:42,53-54,66
    if (D.kind == OBHIP_COV_MAT25 || D.kind == OBHIP_COV_MAT25POW) {
      if (D.kind == OBHIP_COV_MAT25) {
        D.p0 = std::exp(a * hy[0]);  // expLS, covfuncs.cpp:114
        D.p1 = 0;
      } else {
        D.p0 = std::exp(b * hy[1]);              // powv, covfuncs.cpp:198
        D.p1 = std::exp(a * hy[0] + b * hy[1]);  // expLS, :199
      }
      // t_j, centred on the middle of the knots (device_common.h): u_j = t_j - c
      double tmin = 0, tmax = 0;
      for (uint64_t j = 0; j < ml; ++j) {
        const double t = D.kind == OBHIP_COV_MAT25 ? m.knotpt[o + j] / D.p0
                                                   : std::pow(m.knotpt[o + j], D.p0) / D.p1;
        hka[o + j] = t;
        tmin = j == 0 ? t : std::min(tmin, t);
        tmax = j == 0 ? t : std::max(tmax, t);
      }
      D.p2 = 0.5 * (tmin + tmax);
      bool safe = std::isfinite(D.p2);
      for (uint64_t j = 0; j < ml; ++j) {
        const double u = hka[o + j] - D.p2;
        hka[o + j] = u;
        hkb[o + j] = std::exp(u);
        hkc[o + j] = std::exp(-u);
        safe = safe && std::fabs(u) < 150.0;
      }
      // knots spread too far for the separable exponentials: one exp per knot on the device
      if (!safe) D.kind = D.kind == OBHIP_COV_MAT25 ? 3 : 4;  // kCovMat25Direct / kCovMat25PowDirect
      // (build_dim_tab bisects in 7 steps; tables beyond the LDS buffer of k_build_basis are not
      // built: the knot loop is as fast as gathering them from global memory)
      const uint64_t tab_doubles = (ml + 1) / 2 * 2 + (ml + 1) * (uint64_t)D.ncol * 6;
      table_dims.push_back(safe && ml <= 127 && tab_doubles <= (uint64_t)kIntervalTabMax ? (int)l : -1);
    } else {
      table_dims.push_back(-1);
      D.p0 = std::exp(a * hy[0]);  // expLSs, covfuncs.cpp:290
      D.p1 = std::exp(a * hy[1]);  // expLSc, :291
      D.p2 = 0;
      for (uint64_t j = 0; j < ml; ++j) {
        hka[o + j] = std::sin(m.knotpt[o + j]) / D.p0;
        hkb[o + j] = std::cos(m.knotpt[o + j]) / D.p1;
        hkc[o + j] = 0;
      }
    }
    // rot block [m][ncolp], zero padded
    hrot.resize(hrot.size() + ml * D.ncolp, 0.0);
    for (uint64_t j = 0; j < ml; ++j)
      for (int cc = 0; cc < D.ncol; ++cc)
        hrot[D.rotoff + j * D.ncolp + cc] = m.rotmat[(o + cc) * m.mmax + j];
    // Interval tables (mat25 / mat25pow, device_common.h build_dim_tab): the kernel is
    // (1 + h + h^2 / 3) e^{-h} with h = |u(x) - u_j|, so between two neighbouring knots the whole
    // knot sum R[c] = sum_j k(x, knot_j) rot[j][c] is e^{-t} (quadratic in t) + e^{+t} (quadratic in
    // t), t = u(x) - (the largest u_j <= u(x)): with the knots sorted by u, J of them <= u(x),
    // ref = u_(J-1) (u_(0) for J = 0) and d_j = |u_j - ref|,
    //   h = t + d_j (j < J):  (1 + h + h^2/3) e^{-h} = e^{-t} e^{-d_j} [(1 + d_j + d_j^2/3) + t (1 + 2 d_j/3) + t^2/3]
    //   h = d_j - t (j >= J): (1 + h + h^2/3) e^{-h} = e^{+t} e^{-d_j} [(1 + d_j + d_j^2/3) - t (1 + 2 d_j/3) + t^2/3]
    // Six sums over the knots per (interval, level), taken here once per hyper-parameter update
    // in extended precision; the device then spends O(1) per (row, level) instead of a pass over
    // the knots.  Every weight e^{-d_j} is <= 1 and every bracket term positive: what cancels is
    // what cancels in the knot sum itself (the signs of rot), nothing more.
    if (table_dims.back() >= 0) {
      std::vector<int> ord(ml);
      for (uint64_t j = 0; j < ml; ++j) ord[j] = (int)j;
      std::stable_sort(ord.begin(), ord.end(), [&](int a2, int b2) { return hka[o + a2] < hka[o + b2]; });
      const uint64_t mu = (ml + 1) / 2 * 2;  // sorted u, padded to an even length (16-byte aligned tables)
      if (htab.size() % 2) htab.push_back(0.0);
      D.tab = (int)htab.size();
      htab.resize(htab.size() + mu + (ml + 1) * (uint64_t)D.ncol * 6, 0.0);
      double *us = &htab[D.tab];
      for (uint64_t j = 0; j < ml; ++j) us[j] = hka[o + ord[j]];
      for (uint64_t j = ml; j < mu; ++j) us[j] = us[ml - 1];
      // Interval search.  J = number of knots with u_j <= u(x).  For (nearly) equidistant knots --
      // the reference tests' grid, obfit's quantile knots of evenly spread inputs -- the guess
      // J0 = floor((u - u_0) (m - 1) / (u_{m-1} - u_0)) + 1 is off by at most one, and the device
      // settles J with 2 E independent reads of the sorted knots around J0 (one round trip)
      // instead of seven dependent bisection steps.  Whether that holds is CHECKED here, for every
      // interval, with the very expression the device evaluates: the guess at both ends of the
      // interval [u_{J-1}, u_J) must lie within E - 1 of J (one interval of slack for u values
      // the device computes a rounding away from a knot).  Otherwise gwin stays 0: bisection.
      if (ml >= 4 && us[ml - 1] > us[0]) {
        const double g0 = us[0], ginv = (double)(ml - 1) / (us[ml - 1] - us[0]);
        auto guess = [&](double u) {
          double q = std::floor((u - g0) * ginv) + 1.0;
          q = std::min(std::max(q, 0.0), (double)ml);
          return (int64_t)q;
        };
        int64_t worst = 0;
        for (uint64_t J = 0; J <= ml; ++J) {
          // u in [us[J-1], us[J]) (J = 0: below the first knot, J = m: from the last knot on)
          const double a_ = J >= 1 ? us[J - 1] : us[0] - 1.0 / ginv;
          const double b_ = J < ml ? std::nextafter(us[J], -INFINITY) : us[ml - 1] + 1.0 / ginv;
          if (J < ml && J >= 1 && !(us[J] > us[J - 1])) continue;  // (an empty interval: repeated knots)
          worst = std::max<int64_t>(worst, (int64_t)std::llabs(guess(a_) - (int64_t)J));
          worst = std::max<int64_t>(worst, (int64_t)std::llabs(guess(b_) - (int64_t)J));
        }
        if (worst <= 1) {
          D.gwin = (int)worst + 1;  // 1 or 2: reads J0 - E .. J0 + E - 1
          D.g0 = g0;
          D.ginv = ginv;
        }
      }
      double *cf = us + mu;
      // e^{-d_j} = e^{u_j} e^{-ref} (j < J) or e^{-u_j} e^{ref}: |u| < 150, far from the range's end
      std::vector<long double> ep(ml), em(ml);
      for (uint64_t j = 0; j < ml; ++j) {
        ep[j] = expl((long double)us[j]);
        em[j] = expl(-(long double)us[j]);
      }
      for (uint64_t J = 0; J <= ml; ++J) {
        const uint64_t jr = J >= 1 ? J - 1 : 0;
        const double ref = us[jr];
        for (int cc = 0; cc < D.ncol; ++cc) {
          long double A0 = 0, A1 = 0, A2 = 0, B0 = 0, B1 = 0, B2 = 0;
          for (uint64_t j = 0; j < ml; ++j) {
            const long double dj = j < J ? (long double)ref - us[j] : (long double)us[j] - ref;
            const long double w = (j < J ? ep[j] * em[jr] : em[j] * ep[jr]) *
                                  (long double)m.rotmat[(o + cc) * m.mmax + ord[j]];
            const long double q0 = 1 + dj + dj * dj / 3, q1 = 1 + 2 * dj / 3;
            if (j < J) {
              A0 += w * q0;
              A1 += w * q1;
              A2 += w / 3;
            } else {
              B0 += w * q0;
              B1 += w * q1;
              B2 += w / 3;
            }
          }
          double *e = cf + (J * D.ncol + cc) * 6;
          e[0] = (double)A0, e[1] = (double)A1, e[2] = (double)A2;
          e[3] = (double)B0, e[4] = (double)B1, e[5] = (double)B2;
        }
      }
    }
  }
  Mc = ccol;
  OB_TRY(dims.upload(dims_h.data(), d));
  OB_TRY(ka.upload(hka.data(), hka.size()));
  OB_TRY(kb.upload(hkb.data(), hkb.size()));
  OB_TRY(kc.upload(hkc.data(), hkc.size()));
  OB_TRY(rot.upload(hrot.data(), hrot.size()));
  if (htab.empty()) htab.assign(2, 0.0);
  OB_TRY(tab.upload(htab.data(), htab.size()));
  model_version = m.version;
  built_for = &m;
  return 0;
}

}  // namespace obhip

using namespace obhip;

// ---- terms device tables ----------------------------------------------------------
int obhip_terms::prepare(const std::vector<int64_t> &cap,
                         const std::vector<DimDesc> &dims) {
  if (cap == cached_cap && cols.p) return 0;
  for (uint64_t l = 0; l < d; ++l)
    if (maxlev[l] > cap[l])
      return fail(OBHIP_ERR_INVALID,
                  "terms use level " + std::to_string(maxlev[l]) + " in dimension " +
                      std::to_string(l + 1) + " but the basis was built with level cap " +
                      std::to_string(cap[l]));
  // used compact columns (ones column always first)
  std::vector<uint32_t> used;
  used.push_back(0);
  for (uint64_t l = 0; l < d; ++l) {
    std::vector<char> seen(maxlev[l] + 1, 0);
    for (uint64_t k = 0; k < p; ++k) seen[lev[k * d + l]] = 1;
    for (int64_t t = 1; t <= maxlev[l]; ++t)
      if (seen[t]) used.push_back((uint32_t)(dims[l].ccol0 + t - 1));
  }
  std::sort(used.begin() + 1, used.end());
  if (used.size() > 65535) return fail(OBHIP_ERR_INVALID, "terms use more than 65535 basis columns");
  Mu = used.size();
  std::map<uint32_t, uint16_t> pos;
  for (size_t u = 0; u < used.size(); ++u) pos[used[u]] = (uint16_t)u;
  // column lists are read as packed pairs of uint16 (one dword), so W is even;
  // unused slots point at used-column 0, the all-ones column.
  W = std::max<uint64_t>(2, (max_nnz + 1) / 2 * 2);
  p_pad = (p + 255) / 256 * 256;
  std::vector<uint16_t> hc(p_pad * W, 0);
  // Right-aligned: the ones sit in the leading slots, the factors keep their dimension order
  // (so every product rounds as before).  In the term-per-lane kernels the 32 lanes of a
  // bank group read slot j of 32 consecutive terms; right-aligned, those are far more often
  // the same column (a broadcast) -- 1.05 instead of 1.17 LDS cycles per read at C3.
  for (uint64_t k = 0; k < p; ++k) {
    uint64_t nnz = 0;
    for (uint64_t l = 0; l < d; ++l) nnz += lev[k * d + l] > 0;
    uint64_t w = W - nnz;
    for (uint64_t l = 0; l < d; ++l) {
      const uint32_t t = lev[k * d + l];
      if (t > 0) hc[k * W + w++] = pos[(uint32_t)(dims[l].ccol0 + t - 1)];
    }
  }
  // shared sub-products: stars of four terms that differ in one factor (share.cpp).  The star
  // layout also picks the tile index of every used column (its LDS bank pair): `used` and the
  // column lists above are renumbered to match before anything is uploaded.
  sh = obhip::ShareTables();
  if (!no_share) OB_TRY(obhip::build_share_tables(hc.data(), p_pad, W, sh, true, used.size()));
  if (sh.ok) {
    if (sh.relabel.size() != used.size()) return fail(OBHIP_ERR_STATE, "share tables: column count");
    std::vector<uint32_t> used2(used.size());
    for (size_t u = 0; u < used.size(); ++u) used2[sh.relabel[u]] = used[u];
    used.swap(used2);
    for (uint16_t &c : hc) c = sh.relabel[c];
  }
  OB_TRY(cols.upload(hc.data(), hc.size()));
  // Slot order of the term-per-lane kernels: by falling number of factors, so that the 64 x NU
  // consecutive slots a wave owns hold terms of (nearly) one length and the wave reads only
  // that many column slots per term; the padding terms p .. p_pad keep their places.
  {
    std::vector<uint32_t> nz(p, 0), order(p_pad);
    for (uint64_t k = 0; k < p; ++k)
      for (uint64_t l = 0; l < d; ++l) nz[k] += lev[k * d + l] > 0;
    for (uint64_t k = 0; k < p_pad; ++k) order[k] = (uint32_t)k;
    std::stable_sort(order.begin(), order.begin() + p,
                     [&](uint32_t a, uint32_t b) { return nz[a] > nz[b]; });
    OB_TRY(sperm.upload(order.data(), order.size()));
  }
  if (sh.ok) {
    OB_TRY(sh_cols.upload(sh.cols.data(), sh.cols.size()));
    OB_TRY(sh_term.upload(sh.term.data(), sh.term.size()));
    OB_TRY(sh_shape.upload(sh.shape.data(), sh.shape.size()));
    if (sh.left_term.empty()) {  // (never an empty upload: the kernels take the pointers)
      sh.left_term.push_back(0);
      sh.left_cols.assign(W, 0);
    }
    OB_TRY(sh_left_term.upload(sh.left_term.data(), sh.left_term.size()));
    OB_TRY(sh_left_cols.upload(sh.left_cols.data(), sh.left_cols.size()));
    std::vector<uint16_t>().swap(sh.cols);
    std::vector<uint32_t>().swap(sh.term);
    std::vector<uint32_t>().swap(sh.left_term);
    std::vector<uint16_t>().swap(sh.left_cols);
  }
  OB_TRY(ucol.upload(used.data(), used.size()));
  uint64_t Mc = 1;
  for (uint64_t l = 0; l < d; ++l) Mc += (uint64_t)cap[l];
  std::vector<int32_t> hpos(Mc, -1);
  for (size_t u = 0; u < used.size(); ++u) hpos[used[u]] = (int32_t)u;
  OB_TRY(cpos.upload(hpos.data(), hpos.size()));
  cached_cap = cap;
  return 0;
}


namespace obhip {

// hipFuncAttributeMaxDynamicSharedMemorySize once per (device, kernel) and only when a launch
// needs more than what was granted before -- not on every launch
int ensure_dyn_lds(const void *kernel, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void *>, size_t> granted;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  size_t &g = granted[{dev, kernel}];
  if (bytes <= g || bytes <= 64 * 1024) return 0;
  OB_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  g = bytes;
  return 0;
}

uint64_t next_model_version() {
  static std::atomic<uint64_t> counter{0};
  return ++counter;
}

// compute units of a device (cached per device, not per process)
int device_cus(int device) {
  static std::atomic<int> ncu[64];
  const int slot = device >= 0 && device < 64 ? device : 0;
  int v = ncu[slot].load();
  if (!v) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || v <= 0)
      v = 256;
    ncu[slot].store(v);
  }
  return v;
}

}  // namespace obhip

extern "C" {

int obhip_abi_version(void) { return 5; }

const char *obhip_last_error(void) { return g_err.c_str(); }

int obhip_device_count(int *count) {
  if (!count) return fail(OBHIP_ERR_INVALID, "null argument");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count = n;
  return 0;
}

int obhip_set_device(int device) {
  OB_TRY(require_device());
  OB_HIP(hipSetDevice(device));
  return 0;
}

int obhip_set_stream(void *hip_stream) {
  g_stream = (hipStream_t)hip_stream;
  return 0;
}

int obhip_trim_pool(void) {
  OB_TRY(require_device());
  OB_HIP(hipStreamSynchronize(cur_stream()));
  pool_trim();
  return 0;
}

int obhip_synchronize(void) {
  OB_TRY(require_device());
  OB_HIP(hipStreamSynchronize(cur_stream()));
  return 0;
}

int obhip_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  return 0;
}

int obhip_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto &kv : g_prof) prof_drain(kv.second);
  g_prof.clear();
  return 0;
}

int obhip_profile_get(const char *kernel, uint64_t *launches, double *total_ms) {
  if (!kernel) return fail(OBHIP_ERR_INVALID, "null argument");
  if (std::string(kernel) == "host_syncs") {  // blocking waits for the device since the library was loaded
    if (launches) *launches = obhip::g_host_syncs.load();
    if (total_ms) *total_ms = 0;
    return 0;
  }
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (std::string(kernel) == "*") {  // all profiled scopes together
    uint64_t n = 0;
    double ms = 0;
    for (auto &e : g_prof) {
      prof_drain(e.second);
      n += e.second.launches;
      ms += e.second.total_ms;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    return 0;
  }
  auto it = g_prof.find(kernel);
  if (it == g_prof.end()) {
    if (launches) *launches = 0;
    if (total_ms) *total_ms = 0;
    return 0;
  }
  prof_drain(it->second);
  if (launches) *launches = it->second.launches;
  if (total_ms) *total_ms = it->second.total_ms;
  return 0;
}

int obhip_terms_create(obhip_terms **out, const obhip_model *m,
                       const uint64_t *terms, uint64_t p) {
  if (!out || !m || !terms || p == 0) return fail(OBHIP_ERR_INVALID, "terms_create: bad argument");
  if (!m->knots_set) return fail(OBHIP_ERR_STATE, "knots not set");
  obhip_terms *t = new obhip_terms();
  static std::atomic<uint64_t> next_uid{1};
  t->uid = next_uid++;
  t->p = p;
  t->d = m->d;
  t->lev.resize(p * m->d);
  t->maxlev.assign(m->d, 0);
  for (uint64_t k = 0; k < p; ++k) {
    uint64_t nnz = 0;
    for (uint64_t l = 0; l < m->d; ++l) {
      const uint64_t v = terms[l * p + k];
      if (v >= m->m_of(l)) {
        delete t;
        return fail(OBHIP_ERR_INVALID, "terms_create: level out of range");
      }
      t->lev[k * m->d + l] = (uint32_t)v;
      t->maxlev[l] = std::max<int64_t>(t->maxlev[l], (int64_t)v);
      nnz += v > 0;
    }
    t->nnz_total += nnz;
    t->max_nnz = std::max(t->max_nnz, nnz);
  }
  *out = t;
  return 0;
}

int obhip_terms_destroy(obhip_terms *t) {
  delete t;
  return 0;
}

int obhip_terms_info(const obhip_terms *t, uint64_t *p, uint64_t *d,
                     uint64_t *nnz_total, uint64_t *max_nnz) {
  if (!t) return fail(OBHIP_ERR_INVALID, "null terms");
  if (p) *p = t->p;
  if (d) *d = t->d;
  if (nnz_total) *nnz_total = t->nnz_total;
  if (max_nnz) *max_nnz = t->max_nnz;
  return 0;
}

int obhip_terms_maxlevels(const obhip_terms *t, int64_t *levels) {
  if (!t || !levels) return fail(OBHIP_ERR_INVALID, "null argument");
  std::copy(t->maxlev.begin(), t->maxlev.end(), levels);
  return 0;
}

int obhip_terms_share_tables(const obhip_terms *t, uint64_t *info, uint32_t *term, uint32_t *shape,
                             uint16_t *factor, uint32_t *left) {
  if (!t || !info) return fail(OBHIP_ERR_INVALID, "null argument");
  // the same tables obhip_terms::prepare uploads, on factor ids that do not depend on a basis'
  // column layout: 1 + levels of the dimensions before + level - 1
  std::vector<uint64_t> off(t->d + 1, 1);
  for (uint64_t l = 0; l < t->d; ++l) off[l + 1] = off[l] + (uint64_t)t->maxlev[l];
  if (off[t->d] > 65535) return fail(OBHIP_ERR_INVALID, "terms use more than 65535 basis columns");
  const uint64_t W = std::max<uint64_t>(2, (t->max_nnz + 1) / 2 * 2);
  const uint64_t p_pad = (t->p + 255) / 256 * 256;
  std::vector<uint16_t> hc(p_pad * W, 0);
  for (uint64_t k = 0; k < t->p; ++k) {
    uint64_t nnz = 0;
    for (uint64_t l = 0; l < t->d; ++l) nnz += t->lev[k * t->d + l] > 0;
    uint64_t w = W - nnz;
    for (uint64_t l = 0; l < t->d; ++l)
      if (t->lev[k * t->d + l] > 0) hc[k * W + w++] = (uint16_t)(off[l] + t->lev[k * t->d + l] - 1);
  }
  obhip::ShareTables sh;
  OB_TRY(obhip::build_share_tables(hc.data(), p_pad, W, sh));
  if (!sh.ok) return fail(OBHIP_ERR_INVALID, "terms of more than 8 factors are not grouped into stars");
  info[0] = p_pad;
  info[1] = sh.nleft;
  info[2] = sh.reads;
  info[3] = sh.reads_plain;
  info[4] = W;
  info[5] = sh.lds_cycles;
  info[6] = sh.lds_cycles0;
  info[7] = sh.nsw_family;
  info[8] = sh.nsw_plain;
  info[9] = sh.reads_left;
  {  // ... and with the library's choice of the columns' tile indices (what obhip_terms::prepare uploads)
    obhip::ShareTables sh2;
    OB_TRY(obhip::build_share_tables(hc.data(), p_pad, W, sh2, true, off[t->d]));
    info[10] = sh2.lds_cycles;
  }
  if (term) std::copy(sh.term.begin(), sh.term.end(), term);
  if (shape) std::copy(sh.shape.begin(), sh.shape.end(), shape);
  if (factor) std::copy(sh.cols.begin(), sh.cols.end(), factor);
  if (left) std::copy(sh.left_term.begin(), sh.left_term.end(), left);
  return 0;
}

int obhip_malloc(void **d_ptr, uint64_t bytes) {
  if (!d_ptr) return fail(OBHIP_ERR_INVALID, "null argument");
  OB_TRY(require_device());
  OB_HIP(hipMalloc(d_ptr, bytes));
  return 0;
}

int obhip_free(void *d_ptr) {
  if (d_ptr) OB_HIP(hipFree(d_ptr));
  return 0;
}

int obhip_memcpy_h2d(void *d_dst, const void *src, uint64_t bytes) {
  OB_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, cur_stream()));
  OB_HIP(hipStreamSynchronize(cur_stream()));
  return 0;
}

int obhip_memcpy_d2d(void *d_dst, const void *d_src, uint64_t bytes) {
  OB_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, cur_stream()));
  return 0;
}

int obhip_memcpy_d2h(void *dst, const void *d_src, uint64_t bytes) {
  OB_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, cur_stream()));
  OB_HIP(hipStreamSynchronize(cur_stream()));
  return 0;
}

}  // extern "C"
