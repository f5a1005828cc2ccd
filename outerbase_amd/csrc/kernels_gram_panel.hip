// G = B^T B on the FP64 matrix cores of gfx950 (v_mfma_f64_4x4x4_4b_f64) from a
// design matrix materialised ONCE in HBM -- the default Gram path when memory allows.
//
// Replaces loglik_std's getm_ + basismat.t() * basismat (src/linalg.cpp:647-715,
// src/lpdfs/loglik_std.cpp:48,170-173).  Like the reference it forms B, but row-major
// and only as a staging area for the matrix cores (32.8 GB at n = 1e6, p = 4096 --
// 11 % of the 288 GB of one MI355X).
//
// Why not generate the operand panels inside the Gram kernel (kernels_gram_mfma4.hip
// does, and stays as the low-memory fallback): measured with tools/issue_bench.hip, a
// wave that shares a SIMD with two waves saturating the FP64 matrix pipe gets one
// instruction issued every 50-270 cycles even at s_setprio 3.  The ~190 instructions
// per 16-row chunk of an in-kernel producer (LDS gathers, multiplies, panel stores)
// therefore take longer than the 4096 MFMA cycles of the chunk and the consumers
// wait at the barrier (1500 of 5400 cycles per chunk, s_memtime stamps).  Here the
// producers only move bytes: 8 global_load_dwordx4 + 8 ds_write_b128 per chunk.
//
// k_materialize_rows: B[row][term] row-major, p_pad columns.  Lane = row for the
//   Hadamard products (same LDS tile and register-resident term tables as k_mm), a
//   per-wave 32-term LDS transpose, then 256-byte row segments to HBM.
// k_gram_panel: tile pair (I, J), row range split over gridDim.y.  12 waves: 8
//   consumers exactly as in k_gram_mfma4 (64 x 32 wave tiles, 32 accumulators,
//   operands from triple-buffered LDS panels [row][term], pitch 272 doubles); 4
//   producers copy the chunk's two 16 x 128 panels from B, two chunks ahead.
#include "obhip_internal.h"
#include "device_common.h"

namespace obhip {

int launch_gram_reduce(const double *part, int npairs, int nsplit, int nb, int p, double *d_G);

namespace {

constexpr int kGT = 128;            // output tile edge (terms)
constexpr int kCR = 16;             // rows per chunk
constexpr int kTP = 2 * 128 + 16;   // LDS panel row pitch (doubles), see k_gram_mfma4
constexpr int kSteps = kCR / 4;
constexpr int kItems = 8;           // 16-byte items per producer thread and chunk
constexpr int kTB = 65;             // transpose buffer pitch

typedef double d2 __attribute__((ext_vector_type(2)));

// ---- B row-major ------------------------------------------------------------------------
template <int W2>
__global__ void __launch_bounds__(256)
k_materialize_rows(const double *__restrict__ bm, const double *__restrict__ scale,
                   const uint32_t *__restrict__ ucol, int Mu, uint64_t Mc,
                   const uint32_t *__restrict__ colsw, int W2rt, uint64_t p_pad,
                   double *__restrict__ out) {
  extern __shared__ double lds[];
  double *tb = lds + (size_t)Mu * kTileRows;  // [4 waves][32 terms][65]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t tile = blockIdx.x;
  stage_tile<false, false>(lds, bm + tile * Mc * kTileRows, ucol, Mu, threadIdx.x, 256);
  __syncthreads();
  const double s = scale[tile * kTileRows + lane];  // 0 in padding rows
  double *mytb = tb + wave * 32 * kTB;
  const int ngroups = (int)(p_pad / 64);
  for (int g = wave; g < ngroups; g += 4) {
    const int k0 = g * 64;
    uint32_t cw[W2 > 0 ? W2 : 1];
    if constexpr (W2 > 0) load_cw(cw, colsw, k0 + lane);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int t = 0; t < 32; ++t) {
        double v;
        if constexpr (W2 > 0)
          v = term_prod_rl<W2>(lds, cw, h * 32 + t, lane, s);
        else
          v = term_prod_mem(lds, colsw + (size_t)(k0 + h * 32 + t) * W2rt, W2rt, lane, s);
        mytb[t * kTB + lane] = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS writes have landed
      // rows in pairs: lanes 0-31 -> row 2i, lanes 32-63 -> row 2i+1; 32 terms = 256 B
      double *dst = out + (tile * kTileRows + (lane >> 5)) * p_pad + k0 + h * 32 + (lane & 31);
      const double *src = mytb + (lane & 31) * kTB + (lane >> 5);
#pragma unroll 8
      for (int i = 0; i < 32; ++i) dst[(size_t)(2 * i) * p_pad] = src[2 * i];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the buffer is reused
    }
  }
}

// ---- Gram from the materialised B ---------------------------------------------------------
__global__ void __launch_bounds__(768, 3)
k_gram_panel(const double *__restrict__ B, uint64_t p_pad, int nb, uint64_t ntiles,
             uint64_t tiles_per_split, double *__restrict__ part) {
  extern __shared__ double T[];  // [3][16][272]
  constexpr int tsz = kCR * kTP;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 8;

  int I = 0, rem = blockIdx.x;
  while (rem >= nb - I) {
    rem -= nb - I;
    ++I;
  }
  const int J = I + rem;

  const uint64_t t0 = (uint64_t)blockIdx.y * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);
  const int nchunks = (int)(t1 > t0 ? (t1 - t0) * (kTileRows / kCR) : 0);

  if (producer) {
    // item e = pt + 256 q (q < 8): panel = e >> 10 (0: block I, 1: block J), row =
    // (e >> 6) & 15, 16-byte column pair c2 = e & 63: one wave instruction moves one
    // 1-KB row segment (fully coalesced), and lands as 64 consecutive ds_write_b128.
    const int pt = tid - 512;
    const int c2 = pt & 63, rq = pt >> 6;  // rq = 0..3: rows rq, rq+4, rq+8, rq+12
    const double *srcA = B + (t0 * kTileRows + rq) * p_pad + (uint64_t)I * kGT + 2 * c2;
    const double *srcB = B + (t0 * kTileRows + rq) * p_pad + (uint64_t)J * kGT + 2 * c2;
    const int dA = rq * kTP + 2 * c2, dB = rq * kTP + kGT + 2 * c2;
    d2 preA[kItems], preB[kItems];
    auto fetch = [&](int ch, d2 (&pre)[kItems]) {
      const uint64_t off = (uint64_t)ch * kCR * p_pad;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pre[q] = *(const d2 *)(srcA + off + (uint64_t)(4 * q) * p_pad);
        pre[4 + q] = *(const d2 *)(srcB + off + (uint64_t)(4 * q) * p_pad);
      }
    };
    auto put = [&](int tbuf, const d2 (&pre)[kItems]) {
      double *dst = T + tbuf * tsz;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *(d2 *)(dst + dA + 4 * q * kTP) = pre[q];
        *(d2 *)(dst + dB + 4 * q * kTP) = pre[4 + q];
      }
    };
    // Invariant at the top of iteration c: T[c % 3], T[(c+1) % 3] hold chunks c, c+1;
    // this iteration's stage (A on even c, B on odd c) holds chunk c+2, the other c+3.
    if (nchunks > 0) fetch(0, preA);
    if (nchunks > 1) fetch(1, preB);
    if (nchunks > 0) put(0, preA);
    if (nchunks > 2) fetch(2, preA);
    if (nchunks > 1) put(1, preB);
    if (nchunks > 3) fetch(3, preB);
    __syncthreads();  // (P)
    int t2 = 2;  // (c + 2) % 3
    for (int c = 0; c < nchunks; c += 2) {
      if (c + 2 < nchunks) put(t2, preA);
      if (c + 4 < nchunks) fetch(c + 4, preA);
      t2 = t2 == 2 ? 0 : t2 + 1;
      __syncthreads();
      if (c + 1 >= nchunks) break;
      if (c + 3 < nchunks) put(t2, preB);
      if (c + 5 < nchunks) fetch(c + 5, preB);
      t2 = t2 == 2 ? 0 : t2 + 1;
      __syncthreads();
    }
    return;
  }

  // consumers: identical to k_gram_mfma4
  const int wm = wave >> 2, wn = wave & 3;
  const int mk = lane >> 4, mblk = (lane >> 2) & 3, me = lane & 3;
  double acc[4][2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
  const int abase = mk * kTP + wm * 64 + mblk * 4 + me;
  int bbase[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bbase[r] = mk * kTP + kGT + wn * 32 + ((mblk + r) & 3) * 4 + me;
  auto load_ops = [&](const double *tp, int step, double (&a)[4], double (&b)[2][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = tp[abase + i * 16 + 4 * step * kTP];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) b[j][r] = tp[bbase[r] + j * 16 + 4 * step * kTP];
  };
  auto mfma_step = [&](const double (&a)[4], const double (&b)[2][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j][r], acc[i][j][r], 0, 0, 0);
  };

  __syncthreads();  // (P)
  double a0[4], b0[2][4], a1[4], b1[2][4];
  if (nchunks > 0) load_ops(T, 0, a0, b0);
  int tc = 0;  // c % 3
  for (int c = 0; c < nchunks; ++c) {
    const double *tp = T + tc * tsz;
    tc = tc == 2 ? 0 : tc + 1;
    const double *tnext = T + tc * tsz;  // complete since the previous barrier
#pragma unroll
    for (int s = 0; s < kSteps; s += 2) {
      load_ops(tp, s + 1, a1, b1);
      mfma_step(a0, b0);
      if (s + 2 < kSteps)
        load_ops(tp, s + 2, a0, b0);
      else if (c + 1 < nchunks)
        load_ops(tnext, 0, a0, b0);
      mfma_step(a1, b1);
    }
    __syncthreads();
  }

  double *out = part + ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * (kGT * kGT);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * 64 + i * 16 + mblk * 4 + mk;
        const int col = wn * 32 + j * 16 + ((mblk + r) & 3) * 4 + me;
        out[row * kGT + col] = acc[i][j][r];
      }
}

template <int W2>
int run_materialize(const obhip_basis &b, obhip_terms &t, double *d_B) {
  const size_t lds = (t.Mu * kTileRows + 4 * 32 * kTB) * sizeof(double);
  OB_HIP(hipFuncSetAttribute((const void *)k_materialize_rows<W2>,
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_materialize_rows<W2>, dim3((unsigned)(b.n_pad / kTileRows)), dim3(256), lds,
                     cur_stream(), b.bm.p, b.scale.p, t.ucol.p, (int)t.Mu, b.md.Mc,
                     (const uint32_t *)t.cols.p, (int)(t.W / 2), t.p_pad, d_B);
  OB_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// the design matrix needs n_pad * p_pad doubles; use it only when it takes at most half
// of the free HBM (the fused kernel needs none)
bool gram_panel_supports(const obhip_basis &b, const obhip_terms &t) {
  if (t.Mu > 280) return false;
  const size_t need = (size_t)b.n_pad * t.p_pad * sizeof(double);
  if (b.bmat.n * sizeof(double) >= need) return true;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
  return need <= free_b / 2;
}

int launch_gram_panel(const obhip_basis &bc, obhip_terms &t, double *d_G) {
  obhip_basis &b = const_cast<obhip_basis &>(bc);
  if (!gram_panel_supports(b, t))
    return fail(OBHIP_ERR_INVALID, "materialised-B Gram kernel: not enough free HBM for n x p doubles");
  const size_t need = (size_t)b.n_pad * t.p_pad;
  if (b.bmat.n < need) OB_TRY(b.bmat.alloc(need));
  {
    ProfScope ps("materialize_B");
    switch (t.W / 2) {
      case 1: OB_TRY(run_materialize<1>(b, t, b.bmat.p)); break;
      case 2: OB_TRY(run_materialize<2>(b, t, b.bmat.p)); break;
      case 3: OB_TRY(run_materialize<3>(b, t, b.bmat.p)); break;
      case 4: OB_TRY(run_materialize<4>(b, t, b.bmat.p)); break;
      default: OB_TRY(run_materialize<0>(b, t, b.bmat.p)); break;
    }
  }
  const int nb = (int)((t.p + kGT - 1) / kGT);
  const int npairs = nb * (nb + 1) / 2;
  const uint64_t ntiles = b.n_pad / kTileRows;
  uint64_t nsplit = std::max<uint64_t>(1, (4096 + npairs - 1) / npairs);
  nsplit = std::min(nsplit, std::max<uint64_t>(1, ntiles / 8));
  const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  double *part = nullptr;
  OB_TRY(b.workspace((size_t)nsplit * npairs * kGT * kGT * sizeof(double), (void **)&part));
  const size_t lds = (size_t)3 * kCR * kTP * sizeof(double);
  OB_HIP(hipFuncSetAttribute((const void *)k_gram_panel, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)lds));
  {
    ProfScope ps("gram");
    hipLaunchKernelGGL(k_gram_panel, dim3((unsigned)npairs, (unsigned)nsplit), dim3(768), lds,
                       cur_stream(), b.bmat.p, t.p_pad, nb, ntiles, tps, part);
    OB_HIP(hipGetLastError());
  }
  return launch_gram_reduce(part, npairs, (int)nsplit, nb, (int)t.p, d_G);
}

}  // namespace obhip
