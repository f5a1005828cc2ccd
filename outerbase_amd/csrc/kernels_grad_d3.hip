// The fused hyper-gradient pass k_tmm_d3: for the terms that HAVE a dimension, the delta parts of both
// contractions a likelihood needs of d B / d hyp -- w^T d(B a)/dhyp (loglik_gauss.cpp:120-127,
// modandbase.cpp:663-760 matmul_gradhyp) and sum_i d(B^2)_ik/dhyp (diaghessgradhyp,
// loglik_gauss.cpp:158-161; basematsq_gradhyp modandbase.cpp:588-590) -- in one sweep per group of
// dimensions.  The groups' tables are built in kernels_grad.hip (build_d3_groups), the dense part
// every term has is k_tmm_ge0 there.  (Its own translation unit: the instantiations compile for two
// minutes.)
#include <hip/hip_runtime.h>

#include "obhip_internal.h"
#include "device_common.h"

namespace obhip {

namespace {

// out[r0 + y][k] = sum of the row-split partials laid out [split][nc][p_pad], rows r0 + blockIdx.y
__global__ void k_d3_reduce(const double *__restrict__ part, int nsplit, int nc, int r0, uint64_t p_pad,
                            int p, double *__restrict__ out /* [nc][p] */) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int h = r0 + blockIdx.y;
  if (k >= (uint64_t)p) return;
  double s = 0.0;
  for (int r = 0; r < nsplit; ++r) s += part[((uint64_t)r * nc + h) * p_pad + k];
  out[(uint64_t)h * p + k] = s;
}

// ---- the sparse part of both hyper-gradient contractions in ONE pass per dimension group --------
// What the likelihoods need of the terms that HAVE a hyper-parameter's dimension l (levels t > 0
// there) is, with o = the product of the term's OTHER factors, b = its factor in l and
// delta_h = ge_h[t] - b ge_h[0] (so that dB/dhyp_h = ge_h[0] B + o delta_h, mm_gradhyp_dev):
//   u1[k, h] = sum_i w1_i s_i   o delta_h           (B_delta^T w1: gradhyp = yhat_gradhyp^T r,
//                                                    loglik_gauss.cpp:127; tmatmul_gradhyp)
//   u2[k, h] = sum_i w2_i s_i^2 (o delta_h) (o b)   (half the delta part of d(B^2)/dhyp: the squared
//                                                    stores' delta_sq = 2 b delta, basematsq_gradhyp
//                                                    modandbase.cpp:588-590; diaghessgradhyp :158-161)
// Rounds 1-3 ran one restricted k_mm / k_tmm pass per hyper-parameter and store for this, each
// re-reading the other factors: 4 |A| column reads per (term with |A| factors, dimension of two
// hyper-parameters, row).  Here a view-term reads its |A| - 1 other factors, b and the delta
// columns of both hyper-parameters ONCE -- |A| + 2 reads -- and feeds four lane accumulators.
// The row's basescale s and second weight w2 are folded into the tile while it is staged (delta
// columns times s, the dimension's own factor times s w2; the group's tables give those their own
// LDS columns), so that the inner loop multiplies by no weight but w1.
// Skeleton (tile staging, lane = term, hand-issued reads) as k_tmm_tl; the pipeline hands the
// last 1 + NH factors over singly (TlPipe<..., TAIL>).  MODE bit 0: u1, bit 1: u2.
template <int W, int NU, int NH, int MODE>
struct D3Ctx {
  uint32_t ad[NU][W];
  double acc1[NU][NH], acc2[NU][NH];
  double vs;  // first weight of row = lane
  double vr;  // of the current row, wave-uniform
  int rc;
  template <int RR>
  __device__ __forceinline__ void row() {
    if constexpr ((MODE & 1) != 0) vr = readlane_f64(vs, rc + RR);
  }
  // buf[S] = the dimension's own factor (staged times s w2), buf[S + 1 + h] = the delta column of
  // hyper-parameter h (staged times s): the row's basescale and second weight cost no instruction
  // here, the first weight is the multiplier of the accumulation into u1
  template <int RR, int UNIT, int S, bool LEAD>
  __device__ __forceinline__ void use_tail(double o, double (&buf)[12]) {
    double t2 = 0.0;
    if constexpr ((MODE & 2) != 0) t2 = LEAD ? o * buf[S] : buf[S];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const double g = LEAD ? o * buf[S + 1 + h] : buf[S + 1 + h];
      if constexpr ((MODE & 1) != 0) acc1[UNIT][h] = fma(g, vr, acc1[UNIT][h]);
      if constexpr ((MODE & 2) != 0) acc2[UNIT][h] = fma(g, t2, acc2[UNIT][h]);
    }
  }
};


// slot of unit u: the runs of 64 sorted view-terms are dealt to the blocks along p in turn (wave w of
// block y takes runs (w gridDim.y + y) NU ...), so that every block holds the same mix of long and
// short terms -- with tl_slot the first block held the longest and set the time of the launch
template <int NU>
__device__ __forceinline__ uint64_t d3_slot(int wave, int u, int lane) {
  return (((uint64_t)wave * gridDim.y + blockIdx.y) * NU + u) * 64 + lane;
}

template <int W2, int NH, int NU, int MODE, bool PF, int NW>
__global__ void __launch_bounds__(NW * 64, NW / 2)
k_tmm_d3(const double *__restrict__ bm, const double *__restrict__ scale,
         const uint32_t *__restrict__ ucol, int Mu, uint64_t Mc,
         const uint32_t *__restrict__ colsw, const uint32_t *__restrict__ sperm,
         const double *__restrict__ w1, const double *__restrict__ w2, uint64_t n, uint64_t ntiles,
         uint64_t tiles_per_split, uint64_t p_pad, double *__restrict__ part /* [split][2 NH][p_pad] */) {
  extern __shared__ double lds[];
  constexpr int W = 2 * W2, TAIL = 1 + NH, PRE = kD3Pre * 8 / NW;
  static_assert(TAIL <= W, "");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t t0 = (uint64_t)blockIdx.x * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);

  D3Ctx<W, NU, NH, MODE> c;
  int nza = TAIL, nzb = TAIL;
  bool live = false;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const uint64_t slot = d3_slot<NU>(wave, u, lane);
    const bool ok = slot < p_pad;
    live = live || ok;
    const uint64_t k = ok ? sperm[slot] : 0;
#pragma unroll
    for (int h = 0; h < NH; ++h) c.acc1[u][h] = c.acc2[u][h] = 0.0;
    uint32_t cw[W2];
#pragma unroll
    for (int w = 0; w < W2; ++w) {
      cw[w] = ok ? colsw[k * W2 + w] : 0u;  // column 0 = ones
      c.ad[u][2 * w] = (cw[w] & 0xffffu) * (kTlPitch * 8);
      c.ad[u][2 * w + 1] = (cw[w] >> 16) * (kTlPitch * 8);
    }
    if (NU == 1 || u < NU / 2)
      nza = max(nza, tl_nnz<W2>(cw));
    else
      nzb = max(nzb, tl_nnz<W2>(cw));
  }
  const int wea = tl_variant_tail<W, TAIL>(wave_max_i32(nza)), web = tl_variant_tail<W, TAIL>(wave_max_i32(nzb));
  live = wave_max_i32(live ? 1 : 0) != 0;

  // this wave stages columns u = wave + 8 q of every tile: their tile offsets, wave-uniform
  // (lu: tile offset of the column with its staging class in bits 28-29 of the column entry)
  int lu[PF ? PRE : 1], cl[PF ? PRE : 1];
  double pre[PF ? PRE : 1];
  if (PF) {
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
      const int u = wave + NW * q;
      const uint32_t e = u < Mu ? (uint32_t)__builtin_amdgcn_readfirstlane((int)ucol[u]) : 0u;
      lu[q] = (int)(e & kD3ColMask) * kTileRows;
      cl[q] = (int)(e >> 28);
    }
  }
  // vsn: first weight of row = lane; scn, scw2n: the multipliers of the delta columns and of the
  // dimension's own factor (0 beyond the last row: padded rows contribute nothing)
  double vsn = 0.0, scn = 0.0, scw2n = 0.0;
  auto weights = [&](uint64_t tile) {
    const uint64_t row = tile * kTileRows + lane;
    vsn = scn = scw2n = 0.0;
    if (row < n) {
      scn = scale[row];
      if ((MODE & 1) != 0) vsn = w1[row];
      if ((MODE & 2) != 0) scw2n = (w2 ? w2[row] : 1.0) * scn;
    }
  };
  auto fetch = [&](uint64_t tile) {
    const double *src = bm + tile * Mc * kTileRows + lane;
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
      const int u = wave + NW * q;
      pre[q] = u < Mu ? src[lu[q]] : 0.0;
    }
    weights(tile);
  };
  if (PF && t0 < t1) fetch(t0);

  for (uint64_t tile = t0; tile < t1; ++tile) {
    __syncthreads();  // every wave is done with the previous tile
    if (PF) {
#pragma unroll
      for (int q = 0; q < PRE; ++q) {
        const int u = wave + NW * q;
        const double mlt = cl[q] == 0 ? 1.0 : (cl[q] == 1 ? scn : scw2n);
        if (u < Mu) lds[u * kTlPitch + lane] = pre[q] * mlt;
      }
    } else {  // (the other block of the CU computes meanwhile)
      const double *src = bm + tile * Mc * kTileRows + lane;
      weights(tile);
      for (int u = wave; u < Mu; u += NW) {
        const uint32_t e = ucol[u];
        const double mlt = (e >> 28) == 0 ? 1.0 : ((e >> 28) == 1 ? scn : scw2n);
        lds[u * kTlPitch + lane] = src[(size_t)(e & kD3ColMask) * kTileRows] * mlt;
      }
    }
    c.vs = vsn;
    __syncthreads();
    if (PF && tile + 1 < t1) fetch(tile + 1);
    if (!live) continue;  // (whole waves beyond p_pad in the last block along p)
    constexpr int kInflight = NU * W >= 32 ? 8 : 12;
#pragma unroll 1
    for (int rc = 0; rc < kTileRows; rc += kTlChunk) {
      c.rc = rc;
      if constexpr (NU <= 2) {  // one run: the row weights are read once per row, not per half
        tl_run_tail<W, NU, kTlChunk, kInflight, 0, TAIL>(c, max(wea, web));
      } else {
        tl_run_tail<W, NU / 2, kTlChunk, kInflight, 0, TAIL>(c, wea);
        tl_run_tail<W, NU / 2, kTlChunk, kInflight, NU / 2, TAIL>(c, web);
      }
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int j = 0; j < W; ++j) {
          c.ad[u][j] += (rc + kTlChunk < kTileRows) ? kTlChunk * 8 : -(kTileRows - kTlChunk) * 8;
          asm volatile("" : "+v"(c.ad[u][j]));
        }
    }
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const uint64_t slot = d3_slot<NU>(wave, u, lane);
    if (slot < p_pad) {
      const uint64_t k = sperm[slot];
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        if ((MODE & 1) != 0) part[((uint64_t)blockIdx.x * 2 * NH + h) * p_pad + k] = c.acc1[u][h];
        if ((MODE & 2) != 0) part[((uint64_t)blockIdx.x * 2 * NH + NH + h) * p_pad + k] = c.acc2[u][h];
      }
    }
  }
}

}  // namespace

namespace {
template <int W2, int NH, int NU, int MODE, bool PF, int NW>
int run_tmm_d3(const obhip_basis &src, const obhip_terms &v, const double *d_w1, const double *d_w2, dim3 grid,
               uint64_t ntiles, uint64_t tps, double *part) {
  const size_t lds = v.Mu * kTlPitch * sizeof(double);
  if (lds > 64 * 1024) OB_TRY(ensure_dyn_lds((const void *)k_tmm_d3<W2, NH, NU, MODE, PF, NW>, lds));
  hipLaunchKernelGGL((k_tmm_d3<W2, NH, NU, MODE, PF, NW>), grid, dim3(NW * 64), lds, cur_stream(), src.bm.p,
                     src.scale.p, v.ucol.p, (int)v.Mu, src.md.Mc, (const uint32_t *)v.cols.p, v.sperm.p, d_w1,
                     d_w2, src.n, ntiles, tps, v.p_pad, part);
  OB_HIP(hipGetLastError());
  return 0;
}
template <int W2, int NH, int NU, bool PF, int NW>
int run_tmm_d3_mode(int mode, const obhip_basis &src, const obhip_terms &v, const double *d_w1, const double *d_w2,
                    dim3 grid, uint64_t ntiles, uint64_t tps, double *part) {
  if (mode == 1) return run_tmm_d3<W2, NH, NU, 1, PF, NW>(src, v, d_w1, d_w2, grid, ntiles, tps, part);
  if (mode == 2) return run_tmm_d3<W2, NH, NU, 2, PF, NW>(src, v, d_w1, d_w2, grid, ntiles, tps, part);
  return run_tmm_d3<W2, NH, NU, 3, PF, NW>(src, v, d_w1, d_w2, grid, ntiles, tps, part);
}
}  // namespace

// d_out (device, [2 nh][v.p]: u1 of the group's first hyper-parameter, of its second, u2 likewise;
// the rows a mode does not compute are left alone) for one group of build_d3_groups
int launch_tmm_d3(obhip_basis &b, const obhip_terms::GeD3 &g, int mode, const double *d_w1, const double *d_w2,
                  double *d_out) {
  const obhip_basis &src = *b.grad->gb;
  const obhip_terms &v = *g.v;
  const uint64_t ntiles = b.n_pad / kTileRows;
  // 8 waves x 4 view-terms per lane, the tile loaded between the barriers (the other block of the CU
  // computes meanwhile; with prefetch registers this shape spills 77).  OBHIP_D3_VARIANT=0 (A/B runs):
  // 2 per lane, the tile prefetched into registers -- 1.21 against 1.11 ms per launch at d = 8, 1.96
  // against 1.91 at C3 (half the row-weight v_readlanes per view-term).  Also measured: 2 per lane
  // without prefetch 1.27, 16 waves x 1 per lane 1.42 (before the weights went into the staged columns).
  static const int variant = getenv("OBHIP_D3_VARIANT") ? atoi(getenv("OBHIP_D3_VARIANT")) : 1;
  const int nw = 8;
  const int nu = variant == 1 && v.p_pad > 1024 ? 4 : 2;
  const uint64_t tpb = (uint64_t)nw * nu * 64;
  const uint64_t pblocks = (v.p_pad + tpb - 1) / tpb;
  uint64_t nsplit = std::max<uint64_t>(1, (uint64_t)device_cus(b.device) * 2 / pblocks);
  nsplit = std::min(nsplit, std::max<uint64_t>(1, ntiles / 4));
  const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  const int nc = 2 * g.nh;
  double *part = nullptr;
  OB_TRY(b.workspace(nsplit * nc * v.p_pad * sizeof(double), (void **)&part));
  const dim3 grid((unsigned)nsplit, (unsigned)pblocks);
  {
    ProfScope ps("tmm_d3");
#define OB_D3(W2_, NH_)                                                                                      \
  do {                                                                                                       \
    if (nu == 4)                                                                                             \
      OB_TRY((run_tmm_d3_mode<W2_, NH_, 4, false, 8>(mode, src, v, d_w1, d_w2, grid, ntiles, tps, part)));   \
    else                                                                                                     \
      OB_TRY((run_tmm_d3_mode<W2_, NH_, 2, true, 8>(mode, src, v, d_w1, d_w2, grid, ntiles, tps, part)));    \
  } while (0)
    const int w2 = (int)(v.W / 2);
    if (g.nh == 1) {
      if (w2 == 2) OB_D3(2, 1); else if (w2 == 3) OB_D3(3, 1); else OB_D3(4, 1);
    } else {
      if (w2 == 2) OB_D3(2, 2); else if (w2 == 3) OB_D3(3, 2); else OB_D3(4, 2);
    }
#undef OB_D3
  }
  // the rows of the other mode hold whatever the workspace held: reduce only what was computed
  const int r0 = (mode & 1) ? 0 : g.nh, r1 = (mode & 2) ? nc : g.nh;
  hipLaunchKernelGGL(k_d3_reduce, dim3((unsigned)((v.p + 255) / 256), (unsigned)(r1 - r0)), dim3(256), 0,
                     cur_stream(), part, (int)nsplit, nc, r0, v.p_pad, (int)v.p, d_out);
  OB_HIP(hipGetLastError());
  return 0;
}

}  // namespace obhip
