// B^T (c_a B a + c_b y) in ONE pass over the basis, second generation (k_hm2).
//
// The Hessian product of the PCG (loglik_gauss::hessmult, src/lpdfs/loglik_gauss.cpp:137-145:
// B^T (B p)) and the gradient pass of its update() (loglik_gauss.cpp:117-125: yhat = B theta,
// B^T (e^{-2 sigma} (y - yhat))), which lpdf::optcg (src/fit.cpp:71-85) calls once per
// iteration each.  Same algebra as k_hm_tl (kernels_prod.hip): a block holds ALL terms
// (lane = term, NU terms per lane), takes a 64-row tile in sub-chunks of 4 rows, keeps the
// NU x 4 term products of a sub-chunk in registers,
//   1. prod[u][r] by the hand-issued LDS read pipeline (TlPipe), s[r] += a_u prod[u][r];
//   2. s[r] over the 64 lanes (permlane swaps + DPP) and, through LDS and one s_barrier, over
//      the waves: tot_r = sum_k a_k prod_k(row r);
//   3. w_r = c_a s_r^2 tot_r + c_b s_r y_r,  acc[u] += prod[u][r] w_r.
// What round 3's profile showed (LdsUtil 39 %, VALUBusy 55 % at two waves per SIMD, 50-84
// spilled VGPRs): the product phase was bound by the few LDS reads a wave can keep in flight,
// not by the LDS pipe, and every sub-chunk ended in a chain of LDS-pipe round trips
// (ds_bpermute shuffles) behind the barrier.  Changes here:
//   * the next tile is prefetched by LDS-direct loads (global_load_lds_dword: 256 contiguous
//     bytes per wave instruction, two per column at the conflict-free pitch of 65 doubles) into
//     a SECOND tile buffer -- no prefetch registers (32 VGPRs in k_hm_tl), no staging ds_writes;
//   * the register budget that frees goes into occupancy: 16 waves x 4 terms per lane at
//     <= 128 VGPRs (four waves per SIMD, 12 reads in flight each) instead of 8 waves x 8 terms;
//   * the row weights vA = c_a s^2, vB = c_b s y live in LDS (written once per tile) and the wave
//     partials are laid out [row][wave], so the post-barrier phase is one LDS read, four DPP
//     rotate-adds and v_readlane -- no ds_bpermute;
//   * steps 1 and 3 ride in the read pipeline (Hm2Ctx below).
// Measured at the headline terms: 1.70 -> 1.43 ms, of which the read
// pipelines alone are 0.99 ms (LDS pipe ~80 % busy) and the reduction / exchange phases alone 0.52.
// Limits: terms of at most 6 factors (W2 <= 3), p_pad <= 4096, two tiles of the used columns in
// LDS (hm2_supports below); anything else takes k_hm_tl or the two-kernel form.
#include "obhip_internal.h"
#include "device_common.h"

namespace obhip {

namespace {

constexpr int kHm2Chunk = 4;
constexpr int kHm2RedSlots = 64;  // [4 rows][16 waves] partial sums of one sub-chunk

// Step 3 of sub-chunk i (acc[u] += prod[u][r] w_r) is not done behind the barrier: the weights
// stay in four scalar register pairs (wprev) and every product of sub-chunk i is accumulated
// immediately before the read pipeline of sub-chunk i + 1 overwrites it -- 16 multiply-adds per
// wave that now issue under the LDS-bound phase instead of in a VALU-only phase during which the
// LDS pipe idles (all waves of the block are in step, so nothing else would fill it).  The first
// sub-chunk accumulates zeros (prod = 0, wprev = 0); the kernel's epilogue flushes the last one.
// Step 1's sums s[r] += a_u prod[u][r] ride in the pipeline as well when the registers allow
// (SIN: a_u and s[r] in 16 VGPRs); otherwise the coefficients live in LDS and the sums are taken
// after the pipeline has drained (the update() form, which also carries y, yhat and the residual
// sum, and 6-factor terms at 4 per lane).
template <int W, int NU, bool SIN>
struct Hm2Ctx {
  uint32_t ad[NU][W];
  double acc[NU];
  double prod[NU][kHm2Chunk];
  double wprev[kHm2Chunk];   // wave-uniform (v_readlane results)
  double av[SIN ? NU : 1];   // the coefficients a_u of this lane's terms
  double s[kHm2Chunk];       // sum_u a_u prod[u][r] of the sub-chunk in the pipeline
  template <int RR>
  __device__ __forceinline__ void row() {}
  template <int RR, int UNIT>
  __device__ __forceinline__ void use(double v) {
    acc[UNIT] = fma(prod[UNIT][RR], wprev[RR], acc[UNIT]);
    prod[UNIT][RR] = v;
    if constexpr (SIN) s[RR] = fma(v, av[UNIT], s[RR]);
  }
};

// (Terms go to the lanes in the order of obhip_terms::sperm, falling number of factors, NU
// consecutive runs of 64 per wave, as in k_tmm_tl -- so the first waves of a block have the long
// terms and the last ones the short terms.  Evening that out (every wave one run from the long and
// one from the short end) was measured SLOWER, 1.484 -> 1.521 ms at the headline terms: with all
// waves equally long their reduction phases coincide and the LDS pipe idles through them, whereas
// unequal waves reduce while the long ones still read.)

// one wave instruction pair: the 512 bytes of a basis column (64 rows) from g to LDS at l
// (lane l: dword at g + 4 l -> LDS l + 4 l, then the same 256 bytes on)
__device__ __forceinline__ void hm2_dma_col(const char *g /* uniform */, uint32_t voff /* 4 lane */,
                                            uint32_t l /* uniform */) {
// m0 is written here: on the clobber list so that the compiler never assumes a value of its own
// survives the statement (round-4 advice; m0 is a reserved register, hence the diagnostic)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1\n\t"
               "global_load_lds_dword %0, %1 offset:256"
               :: "v"(voff), "s"(g), "s"(l) : "memory", "m0");
#pragma clang diagnostic pop
}

// one sub-chunk of 4 rows (tile rows rc .. rc + 3; the addresses in c.ad point ROW0 rows before).
// wea / web: column reads per term of the first / second half of the lane's units (the pipeline
// variant is picked by a wave-uniform branch per sub-chunk, as in k_tmm_tl.  The joins of those
// branches cost ~10 v_mov_b64 each, 13 % of the kernel's VALU instructions; making the choice once
// per tile instead -- nine instantiations of the whole row loop behind one dispatch -- was built and
// measured: 70-200 spilled registers, some reloaded inside the loops, 1.43 -> 2.19 ms.)
template <int W, int NU, int ROW0, int INFL, bool RO, bool SIN>
__device__ __forceinline__ void hm2_subchunk(Hm2Ctx<W, NU, SIN> &c, bool live, int wea, int web, double *red_half,
                                             const double *wts /* [2][64] of this tile */,
                                             const double *avl /* !SIN: a of this lane's unit 0 */, int avstride,
                                             int wave, int lane, int rc, double &totrow) {
#pragma unroll
  for (int r = 0; r < kHm2Chunk; ++r) c.s[r] = 0.0;
  if (live) {  // (a wave without terms keeps prod = 0)
    if constexpr (NU == 1) {
      tl_run_half<W, 1, kHm2Chunk, INFL, 0, ROW0>(c, wea);
    } else {
      tl_run_half<W, NU / 2, kHm2Chunk, INFL, 0, ROW0>(c, wea);
      tl_run_half<W, NU - NU / 2, kHm2Chunk, INFL, NU / 2, ROW0>(c, web);
    }
  }
  if constexpr (!SIN) {  // s[r] = sum_u a_u prod[u][r] of this lane's terms, coefficients from LDS
    double av[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) av[u] = avl[u * avstride];
#pragma unroll
    for (int r = 0; r < kHm2Chunk; ++r) c.s[r] = av[0] * c.prod[0][r];
#pragma unroll
    for (int u = 1; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < kHm2Chunk; ++r) c.s[r] = fma(av[u], c.prod[u][r], c.s[r]);
  }
  // the row weights of the 4 rows, row r in the 16-lane row r (independent of the sums: issued
  // ahead of the barrier)
  const double vA = wts[rc + (lane >> 4)];
  double vB = 0.0;
  if (RO) vB = wts[64 + rc + (lane >> 4)];
  // s[0..3] over the 64 lanes: 16-lane row q ends with the sum of s[q]
  static_assert(kHm2Chunk == 4, "the butterfly below reduces 4 rows");
  double v = swap16_sum(swap32_sum(c.s[0], c.s[2]), swap32_sum(c.s[1], c.s[3]));
  v = row16_ror_add<8>(v);
  v = row16_ror_add<4>(v);
  v = row16_ror_add<2>(v);
  v = row16_ror_add<1>(v);
  if ((lane & 15) == 0) red_half[(lane >> 4) * 16 + wave] = v;  // [row][wave]
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (not vmcnt: the next tile's loads stay in flight)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // over the waves: lane l reads the partial of row l / 16, wave l % 16 (slots of absent waves are
  // zero): four rotate-adds within the rows of 16 lanes and EVERY lane l holds tot of row l / 16
  double t = red_half[lane];
  t = row16_ror_add<8>(t);
  t = row16_ror_add<4>(t);
  t = row16_ror_add<2>(t);
  t = row16_ror_add<1>(t);
  const double wl = RO ? fma(vA, t, vB) : vA * t;
  if (RO) {  // lane = row keeps sum_k a_k prod_k of its row: row j's total sits in lanes 16 j ..
    const double tj = __shfl(t, 16 * (lane & 3), 64);
    if ((lane & ~3) == rc) totrow = tj;
  }
#pragma unroll
  for (int r = 0; r < kHm2Chunk; ++r) c.wprev[r] = readlane_f64(wl, 16 * r);
}

// the 64 rows of the staged tile: 8 x two sub-chunks; step_last moves the addresses to row 0 of the
// other tile buffer after the last rows
template <int W, int NU, int INFL, bool RO, bool SIN>
__device__ __forceinline__ void hm2_tile(Hm2Ctx<W, NU, SIN> &c, bool live, int wea, int web, double *red, const double *wt,
                                         const double *avl, int avstride, int wave, int lane,
                                         int32_t step_last, double &totrow) {
#pragma unroll 1
  for (int rc = 0; rc < kTileRows; rc += 2 * kHm2Chunk) {
    // two sub-chunks per address update: the second reads at immediate row offsets 4 .. 7; the
    // cross-wave sums alternate between the two halves of red
    hm2_subchunk<W, NU, 0, INFL, RO, SIN>(c, live, wea, web, red, wt, avl, avstride, wave, lane, rc, totrow);
    hm2_subchunk<W, NU, kHm2Chunk, INFL, RO, SIN>(c, live, wea, web, red + kHm2RedSlots, wt, avl, avstride,
                                                  wave, lane, rc + kHm2Chunk, totrow);
    const int32_t step = rc + 2 * kHm2Chunk < kTileRows ? 2 * kHm2Chunk * 8 : step_last;
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int j = 0; j < W; ++j) {
        c.ad[u][j] += (uint32_t)step;
        asm volatile("" : "+v"(c.ad[u][j]));
      }
  }
}

// RO: the update() form (y, yhat, sum of squared residuals); without it the Hessian product
template <int W2, int NU, int WAVES, int INFL, bool RO, bool SIN>
__global__ void __launch_bounds__(WAVES * 64, WAVES / 4)
k_hm2(const double *__restrict__ bm, const double *__restrict__ scale, const uint32_t *__restrict__ ucol,
      int Mu, uint64_t Mc, const uint32_t *__restrict__ colsw, const uint32_t *__restrict__ sperm,
      const double *__restrict__ a, int p, const double *__restrict__ y, double ca, double cb, uint64_t n,
      uint64_t ntiles, uint64_t tiles_per_split, uint64_t p_pad, double *__restrict__ part,
      double *__restrict__ yhat, double *__restrict__ sspart, const double *__restrict__ stop0,
      const double *__restrict__ stop1) {
  // a launch enqueued before the host has read the step's break conditions (the PCG loop of
  // api.cpp does that, to keep the GPU busy through the host round trip): nothing to do when the
  // iteration it was meant for will not happen
  if (stop0 != nullptr && (*stop0 != 0.0 || *stop1 != 0.0)) return;
  extern __shared__ double lds[];
  constexpr int W = 2 * W2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile_doubles = Mu * kTlPitch;
  double *wts = lds + 2 * (size_t)tile_doubles;  // [2 buffers][vA 64 | vB 64]
  double *red = wts + 2 * 128;                    // [2 halves][4 rows][16 waves]
  double *avs = red + 2 * kHm2RedSlots;           // !SIN: [NU][WAVES * 64], a of the term in slot (wave, u, lane)
  constexpr int avstride = WAVES * 64;
  const double *avl = avs + wave * 64 + lane;

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double *)lds;
  const uint32_t tile_bytes = (uint32_t)tile_doubles * 8u;
  const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);

  Hm2Ctx<W, NU, SIN> c;
  int nza = 1, nzb = 1;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const uint64_t slot = tl_slot<NU>(0, wave, u, lane);
    const bool ok = slot < p_pad;
    const uint64_t k = ok ? sperm[slot] : 0;
    c.acc[u] = 0.0;
#pragma unroll
    for (int r = 0; r < kHm2Chunk; ++r) c.prod[u][r] = 0.0;
    const double au = ok && k < (uint64_t)p ? a[k] : 0.0;
    if constexpr (SIN)
      c.av[u] = au;
    else
      avs[u * avstride + wave * 64 + lane] = au;
    uint32_t cw[W2];
#pragma unroll
    for (int w = 0; w < W2; ++w) {
      cw[w] = ok ? colsw[k * W2 + w] : 0u;  // column 0 = ones
      c.ad[u][2 * w] = lds0 + (cw[w] & 0xffffu) * (kTlPitch * 8);
      c.ad[u][2 * w + 1] = lds0 + (cw[w] >> 16) * (kTlPitch * 8);
    }
    if (NU == 1 || u < NU / 2)
      nza = max(nza, tl_nnz<W2>(cw));
    else
      nzb = max(nzb, tl_nnz<W2>(cw));
  }
#pragma unroll
  for (int r = 0; r < kHm2Chunk; ++r) c.wprev[r] = 0.0;
  const int wea = tl_variant<W>(wave_max_i32(nza)), web = tl_variant<W>(wave_max_i32(nzb));
  const bool live = ((uint64_t)wave * NU) * 64 < p_pad;  // (whole waves beyond p_pad: zeros)
  if (threadIdx.x < 2 * kHm2RedSlots) red[threadIdx.x] = 0.0;  // slots of absent waves stay zero

  // next tile -> the other buffer, by LDS-direct loads; the last wave also fetches the row weights
  // (scale and y are requested BEFORE the LDS-direct loads and only used at the top of the next
  // tile: the compiler's own vmcnt bookkeeping does not see the inline-asm loads, so a use right
  // here would wait for all of them)
  double scn = 0.0, yn = 0.0;
  auto prefetch = [&](uint64_t tile, int bsel) {
    if (wave == WAVES - 1) {
      const uint64_t row = tile * kTileRows + lane;
      scn = yn = 0.0;
      if (row < n) {
        scn = scale[row];
        if (RO) yn = y[row];
      }
    }
    const char *tb = (const char *)(bm + tile * Mc * kTileRows);
    const uint32_t l0 = lds0 + (bsel ? tile_bytes : 0u);
    for (int u = wave; u < Mu; u += WAVES) {
      const uint32_t col = __builtin_amdgcn_readfirstlane(ucol[u]);
      // (the address is wave-uniform; said explicitly, so that it reaches the "s" operands of the
      // asm in scalar registers whatever the optimiser made of the surrounding code)
      const uint64_t ga = (uint64_t)(tb + (size_t)col * (kTileRows * 8));
      const uint64_t gu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ga >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ga);
      hm2_dma_col((const char *)gu, (uint32_t)lane * 4u,
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(l0 + (uint32_t)u * (kTlPitch * 8))));
    }
  };
  if (t0 < t1) prefetch(t0, 0);
  double ssacc = 0.0;  // wave 0: sum over its rows of (yhat - y)^2

  for (uint64_t tile = t0; tile < t1; ++tile) {
    const int bsel = (int)((tile - t0) & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the tile has landed
    if (wave == WAVES - 1) {  // per row (lane = row): vA = c_a s^2, vB = c_b s y  ->  w = vA tot + vB
      wts[bsel * 128 + lane] = ca * scn * scn;
      if (RO) wts[bsel * 128 + 64 + lane] = cb * scn * yn;
    }
    __syncthreads();  // tile and weights complete; every wave is done with the other buffer
    if (tile + 1 < t1) prefetch(tile + 1, bsel ^ 1);
    const double *wt = wts + bsel * 128;
    double totrow = 0.0;  // lane = row: sum_k a_k prod_k of this tile's row
    // after the last rows: row 0 of the other buffer
    const int32_t step_last = -(kTileRows - 2 * kHm2Chunk) * 8 + (bsel ? -(int32_t)tile_bytes : (int32_t)tile_bytes);
    hm2_tile<W, NU, INFL, RO, SIN>(c, live, wea, web, red, wt, avl, avstride, wave, lane, step_last, totrow);
    if (RO && wave == 0) {
      const uint64_t row = tile * kTileRows + lane;
      if (row < n) {
        const double yh = scale[row] * totrow;
        if (yhat != nullptr) yhat[row] = yh;
        const double dlt = yh - y[row];
        ssacc = fma(dlt, dlt, ssacc);
      }
    }
  }
  // the last sub-chunk's products have not been accumulated yet
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int r = 0; r < kHm2Chunk; ++r) c.acc[u] = fma(c.prod[u][r], c.wprev[r], c.acc[u]);
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const uint64_t slot = tl_slot<NU>(0, wave, u, lane);
    if (slot < p_pad) part[(uint64_t)blockIdx.x * p_pad + sperm[slot]] = c.acc[u];
  }
  if (RO && wave == 0 && sspart != nullptr) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ssacc += __shfl_xor(ssacc, off, 64);
    if (lane == 0) sspart[blockIdx.x] = ssacc;
  }
}

template <int W2, int NU, int WAVES, int INFL, bool SIN>
int run_hm2(const obhip_basis &b, obhip_terms &t, const double *d_a, const double *d_y, double ca, double cb,
            double *part, double *d_yhat, double *sspart, unsigned nsplit, uint64_t ntiles, uint64_t tps,
            size_t lds, const double *stop0, const double *stop1) {
  // (the update() form carries y, yhat and the residual sum: 8 reads in flight keep it free of spills)
#define OB_HM2_LAUNCH(RO_)                                                                                  \
  do {                                                                                                      \
    constexpr int IF = RO_ && NU >= 4 ? 8 : INFL;                                                           \
    constexpr bool SI = SIN && !(RO_ && NU >= 4);                                                           \
    OB_TRY(ensure_dyn_lds((const void *)k_hm2<W2, NU, WAVES, IF, RO_, SI>, lds));                           \
    hipLaunchKernelGGL((k_hm2<W2, NU, WAVES, IF, RO_, SI>), dim3(nsplit), dim3(WAVES * 64), lds, cur_stream(), \
                       b.bm.p, b.scale.p, t.ucol.p, (int)t.Mu, b.md.Mc, (const uint32_t *)t.cols.p,         \
                       t.sperm.p, d_a, (int)t.p, d_y, ca, cb, b.n, ntiles, tps, t.p_pad, part, d_yhat,      \
                       sspart, stop0, stop1);                                                               \
  } while (0)
  if (d_y != nullptr)
    OB_HM2_LAUNCH(true);
  else
    OB_HM2_LAUNCH(false);
#undef OB_HM2_LAUNCH
  OB_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// two tile buffers, the row weights, the wave partials and -- for the instantiations that keep
// it in LDS (launch_hm2 below: 4 terms per lane in the update() form or with 6-factor terms) -- the
// coefficient vector, one slot per lane and unit
size_t hm2_lds_bytes(const obhip_terms &t, bool ro, int variant) {
  const int w2 = (int)(t.W / 2);
  const int nu = t.p_pad <= 1024 ? 1 : (t.p_pad <= 2048 ? 2 : 4);
  size_t slots = 0;
  if (variant == 2 || variant == 3)
    slots = ro ? 4608 : 0;  // (12 waves x 6 units, 8 x 8: the experiments at 4-factor terms)
  else if ((w2 >= 2 && variant == 5) || (w2 == 3 && variant == 7) || (nu == 4 && ro))
    slots = 4096;  // the instantiations that keep the coefficients in LDS (launch_hm2): variants 5
                   // and 7 are 16 waves x 4 units whatever p_pad is (round-4 advice), the update()
                   // form at 4 units per lane
  return ((size_t)2 * t.Mu * kTlPitch + 2 * 128 + 2 * kHm2RedSlots + slots) * sizeof(double);
}

// the terms this kernel takes in the form asked for (launch_hessmult_fused asks before it falls
// back to k_hm_tl): up to 147 used columns for the Hessian product at 4-factor terms, 118 where the
// coefficients live in LDS too
bool hm2_supports(const obhip_terms &t, bool ro, int variant) {
  const int w2 = (int)(t.W / 2);
  return w2 >= 1 && w2 <= 3 && t.p_pad <= 4096 && hm2_lds_bytes(t, ro, variant) <= (size_t)156 * 1024;
}

// variant: 0 = automatic; experiments at 4-factor terms (OBHIP_HM2_VARIANT): 2 = 12 waves x 6
// terms per lane, 3 = 8 x 8, 5 = 16 x 4 with the coefficients in LDS and 12 reads in flight,
// 6 = 16 x 4, sums in the pipeline, 12 reads in flight (2 spilled registers)
int launch_hm2(const obhip_basis &b, obhip_terms &t, const double *d_a, const double *d_y, double ca,
               double cb, double *part, double *d_yhat, double *sspart, unsigned nsplit, uint64_t ntiles,
               uint64_t tps, int variant, const double *stop0, const double *stop1) {
  const size_t lds = hm2_lds_bytes(t, d_y != nullptr, variant);
  const int w2 = (int)(t.W / 2);
  const uint64_t pp = t.p_pad;
#define OB_HM2(W2_, NU_, WAVES_, INFL_, SIN_) \
  return run_hm2<W2_, NU_, WAVES_, INFL_, SIN_>(b, t, d_a, d_y, ca, cb, part, d_yhat, sspart, nsplit, ntiles, tps, lds, \
                                                stop0, stop1)
  if (w2 == 2) {
    if (variant == 2 && pp <= 12 * 6 * 64) OB_HM2(2, 6, 12, 12, true);
    if (variant == 3 && pp <= 8 * 8 * 64) OB_HM2(2, 8, 8, 12, true);
    if (variant == 5) OB_HM2(2, 4, 16, 12, false);
    if (variant == 6) OB_HM2(2, 4, 16, 12, true);
    if (pp <= 16 * 1 * 64) OB_HM2(2, 1, 16, 12, true);
    if (pp <= 16 * 2 * 64) OB_HM2(2, 2, 16, 12, true);
    OB_HM2(2, 4, 16, 8, true);
  }
  if (w2 == 3) {  // terms of 5 and 6 factors (obfit's eight-dimensional examples)
    // (n = 1e6, p = 4096, d = 8, tools/hm_bench.py: sums outside the pipeline, 8 reads in flight 1.797 ms,
    // 12 in flight 1.780, sums inside -- 128 VGPRs, two of them spilled outside the loops -- 1.723)
    if (variant == 5) OB_HM2(3, 4, 16, 12, false);
    if (variant == 7) OB_HM2(3, 4, 16, 8, false);
    if (pp <= 16 * 1 * 64) OB_HM2(3, 1, 16, 12, true);
    if (pp <= 16 * 2 * 64) OB_HM2(3, 2, 16, 12, true);
    OB_HM2(3, 4, 16, 8, true);
  }
  if (pp <= 16 * 1 * 64) OB_HM2(1, 1, 16, 12, true);
  if (pp <= 16 * 2 * 64) OB_HM2(1, 2, 16, 12, true);
  OB_HM2(1, 4, 16, 12, true);
#undef OB_HM2
}

}  // namespace obhip
