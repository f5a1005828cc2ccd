// G = B^T B on the FP64 matrix cores of gfx950 with v_mfma_f64_4x4x4_4b_f64,
// warp-specialised: the default Gram kernel.
//
// Replaces loglik_std::hess (src/lpdfs/loglik_std.cpp:170-173, basismat.t() *
// basismat on the design matrix materialised by getm_, src/linalg.cpp:647-715).
//
// Why this instruction: measured on MI355X (tools/mfma4x4_bench.hip,
// tools/fp64_pipes_bench.hip; profiles/r01_fp64_pipes_microbench.txt) the 4x4x4
// four-block FP64 MFMA sustains 70-77 TFLOP/s (the 78.6 TF datasheet rate) with
// one to three waves per SIMD once 64 independent accumulators are in flight,
// whereas v_mfma_f64_16x16x4_f64 tops out at 36-49 TFLOP/s.  Its operands are
// one f64 per lane, lane = 16 k + 4 blk + e (k = row of the K = 4 step, blk = one
// of four independent 4 x 4 x 4 blocks, e = row of A / column of B inside the
// block); the result lane 16 i + 4 blk + j holds D_blk[i][j] (probed on the
// device; CBSZ/ABID block broadcast is ignored by this opcode).  All 16
// (row group, column group) pairings of an A and a B register are reached by
// reading the B operand four times from LDS with its blocks rotated.
//
// Decomposition (tiling contract shared with the other Gram kernels: 128 x 128
// output tiles over upper-triangular tile pairs, rows split over gridDim.y,
// partials summed in a fixed order by k_gram_reduce):
//   - workgroup = 12 waves, one per CU (3 waves per SIMD).  Waves 0-7 are
//     CONSUMERS: each owns a 64 x 32 wave tile = 4 A registers x 2 B registers x 4
//     rotations = 32 accumulators, and per K = 4 step issues 32 MFMAs on 12
//     operands read with ds_read_b64; two consumers share a SIMD so that one
//     wave's LDS issue slots and barrier waits hide behind the other's MFMAs
//     (a single wave cannot overlap its own ds_read issue with its FP64 MFMAs:
//     measured 58 TFLOP/s consumer-only at one wave per SIMD) (conflict-free: 16 terms x 2 rows at pitch 18 doubles hit 32
//     different 8-byte slots).  Waves 8-11 are PRODUCERS: thread t owns term t of
//     the tile pair; per 16-row chunk they write the chunk's slice of the used
//     basemat columns to LDS (global loads issued two chunks earlier, so HBM
//     latency never stalls anyone), and build the operand panels T[term][16 rows]
//     for the next chunk (A block scaled by basescale^2): W runs of 16 rows
//     (ds_read_b128) multiplied together.
//   - panels are triple buffered, slices double buffered; ONE workgroup barrier per
//     chunk.  In iteration c the consumers read T[c % 3] (and pre-load the first
//     operands of T[(c+1) % 3], complete since the previous barrier) while the
//     producers write T[(c+2) % 3] from slice[c & 1] and refill slice[(c+1) & 1].
//   - basemat is never re-laid-out for this: a chunk's slice is 128 contiguous
//     bytes per column in the tile-blocked HBM layout.
#include "obhip_internal.h"
#include "device_common.h"

namespace obhip {


namespace {

constexpr int kGT = 128;    // output tile edge (terms)
constexpr int kCR = 16;     // rows per chunk
constexpr int kLD = 18;     // padded leading dimension (doubles) of a slice column
constexpr int kTP = 2 * 128 + 16;  // panel row pitch (doubles): [row][term], +128 B so that
                                   // rows k and k+1 fall in different bank halves
constexpr int kMaxPre = 8;  // prefetch registers per producer thread and stage => Mu <= 128
constexpr int kChunksPerTile = kTileRows / kCR;
constexpr int kSteps = kCR / 4;

typedef double d2 __attribute__((ext_vector_type(2)));

template <int W>
__global__ void __launch_bounds__(768, 3)
k_gram_mfma4(const double *__restrict__ bm, const double *__restrict__ scale,
             const uint32_t *__restrict__ ucol, int Mu, uint64_t Mc,
             const uint16_t *__restrict__ cols, int nb, uint64_t ntiles,
             uint64_t tiles_per_split, double *__restrict__ part) {
  extern __shared__ double lds[];
  const int subsz = (Mu + 1) * kLD;  // slice: Mu columns + basescale^2 as column Mu
  const int tsz = kCR * kTP;         // panels [row][term]: A block terms 0..127, B block 128..255
  double *sub = lds;                 // [2][Mu + 1][18]
  double *T = sub + 2 * subsz;       // [3][16][272]
  int *lu = (int *)(T + 3 * tsz);    // [Mu] ucol[u] * 64

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
  const int nchunks = (int)(t1 > t0 ? (t1 - t0) * kChunksPerTile : 0);

  for (int u = tid; u < Mu; u += 768) lu[u] = (int)ucol[u] * kTileRows;
  __syncthreads();

  if (producer) {
    // ---------------------------------------------------------------- producers
    __builtin_amdgcn_s_setprio(3);  // tiny instruction stream that must never be the late arrival
    const int pt = tid - 512;   // term of the tile pair owned by this thread
    const bool isA = pt < kGT;
    const int term = isA ? I * kGT + pt : J * kGT + (pt - kGT);
    int coff[W];
#pragma unroll
    for (int w = 0; w < W; ++w) coff[w] = (int)cols[(size_t)term * W + w] * kLD;
    const int soff = Mu * kLD;  // the basescale^2 pseudo-column

    // slice element (u, r) of chunk ch: bm[(tile*Mc + ucol[u])*64 + (ch%4)*16 + r];
    // thread pt moves row r = pt & 15 of columns u = (pt >> 4) + 16 q
    const int pr = pt & 15, pu = pt >> 4;
    int goff[kMaxPre];  // ucol[u] * 64 of this thread's columns (clamped: loads are branch-free)
#pragma unroll
    for (int q = 0; q < kMaxPre; ++q) goff[q] = lu[min(pu + 16 * q, Mu - 1)];
    double preA[kMaxPre], preB[kMaxPre];
    double presA = 0.0, presB = 0.0;
    auto fetch = [&](int ch, double (&pre)[kMaxPre], double &pres) {
      const uint64_t tile = t0 + ch / kChunksPerTile;
      const int roff = (ch % kChunksPerTile) * kCR + pr;
      const double *src = bm + tile * Mc * kTileRows + roff;
#pragma unroll
      for (int q = 0; q < kMaxPre; ++q) pre[q] = src[goff[q]];
      pres = scale[tile * kTileRows + roff];
    };
    auto put_slice = [&](int buf, const double (&pre)[kMaxPre], double pres) {
      double *dst = sub + buf * subsz;
#pragma unroll
      for (int q = 0; q < kMaxPre; ++q) {
        const int u = pu + 16 * q;
        if (u < Mu) dst[u * kLD + pr] = pre[q];
      }
      if (pt < kCR) dst[soff + pt] = pres * pres;
    };
    // the LDS reads of half a panel row (8 rows, all W columns) are issued before the
    // first multiply: two LDS round trips per chunk instead of one per column
    auto gen_panel = [&](int buf, int tbuf) {
      const double *src = sub + buf * subsz;
      double *dst = T + tbuf * tsz + pt;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        d2 c[W][kCR / 4];
        d2 v[kCR / 4];
#pragma unroll
        for (int r = 0; r < kCR / 4; ++r) v[r] = *(const d2 *)(src + (isA ? soff : 0) + 8 * h + 2 * r);
#pragma unroll
        for (int w = 0; w < W; ++w)
#pragma unroll
          for (int r = 0; r < kCR / 4; ++r) c[w][r] = *(const d2 *)(src + coff[w] + 8 * h + 2 * r);
#pragma unroll
        for (int w = 0; w < W; ++w)
#pragma unroll
          for (int r = 0; r < kCR / 4; ++r) v[r] *= c[w][r];
#pragma unroll
        for (int r = 0; r < kCR / 4; ++r) {
          dst[(8 * h + 2 * r) * kTP] = v[r].x;
          dst[(8 * h + 2 * r + 1) * kTP] = v[r].y;
        }
      }
    };

    // Invariant at the top of iteration c: panels T[c % 3] and T[(c+1) % 3] are
    // complete (chunks c, c+1), slice[c & 1] holds chunk c+2, the register stage of
    // this iteration's parity (B on even c, A on odd c) holds chunk c+3 and the other
    // stage chunk c+4.  Iteration c: panels of chunk c+2 -> T[(c+2) % 3], slice of
    // chunk c+3 -> slice[(c+1) & 1], fetch chunk c+5.
    if (nchunks > 0) fetch(0, preA, presA);
    if (nchunks > 1) fetch(1, preB, presB);
    if (nchunks > 0) put_slice(0, preA, presA);
    if (nchunks > 2) fetch(2, preA, presA);
    __syncthreads();  // (P1)
    if (nchunks > 0) gen_panel(0, 0);
    if (nchunks > 1) put_slice(1, preB, presB);
    if (nchunks > 3) fetch(3, preB, presB);
    __syncthreads();  // (P2)
    if (nchunks > 1) gen_panel(1, 1);
    if (nchunks > 2) put_slice(0, preA, presA);
    if (nchunks > 4) fetch(4, preA, presA);
    __syncthreads();  // (P3)
    int t2 = 2;  // (c + 2) % 3
    for (int c = 0; c < nchunks; c += 2) {
      if (c + 2 < nchunks) gen_panel(0, t2);
      if (c + 3 < nchunks) put_slice(1, preB, presB);
      if (c + 5 < nchunks) fetch(c + 5, preB, presB);
      t2 = t2 == 2 ? 0 : t2 + 1;
      __syncthreads();
      if (c + 1 >= nchunks) break;
      if (c + 3 < nchunks) gen_panel(1, t2);
      if (c + 4 < nchunks) put_slice(0, preA, presA);
      if (c + 6 < nchunks) fetch(c + 6, preA, presA);
      t2 = t2 == 2 ? 0 : t2 + 1;
      __syncthreads();
    }
    return;
  }

  // ------------------------------------------------------------------ consumers
  const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 consumer waves, 64 x 32 each
  const int mk = lane >> 4, mblk = (lane >> 2) & 3, me = lane & 3;
  double acc[4][2][4];  // [A register][B register][block rotation]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
  // operand addresses (doubles, relative to a panel buffer, step 0); register i / j
  // of an operand set is a compile-time offset of 16 panel rows from these bases
  const int abase = mk * kTP + wm * 64 + mblk * 4 + me;
  int bbase[4];  // one per block rotation
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

  __syncthreads();  // (P1)
  __syncthreads();  // (P2)
  __syncthreads();  // (P3)
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
        load_ops(tnext, 0, a0, b0);  // first operands of the next chunk cross the barrier
      mfma_step(a1, b1);
    }
    __syncthreads();
  }

  // result lanes: lane = 16 i + 4 blk + j holds D_blk[i][j]
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

template <int W>
int run_gram_mfma4(const obhip_basis &b, obhip_terms &t, const GramSink &sink) {
  const int nb = (int)((t.p + kGT - 1) / kGT);
  const int npairs = nb * (nb + 1) / 2;
  const uint64_t ntiles = b.n_pad / kTileRows;
  // one workgroup per CU: ~16 rounds over 256 CUs
  uint64_t nsplit = std::max<uint64_t>(1, (4096 + npairs - 1) / npairs);
  nsplit = std::min(nsplit, std::max<uint64_t>(1, ntiles / 8));
  const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  double *part = nullptr;
  OB_TRY(const_cast<obhip_basis &>(b).workspace((size_t)nsplit * npairs * kGT * kGT * sizeof(double),
                                                (void **)&part));
  const size_t lds = (2 * (t.Mu + 1) * kLD + 3 * kCR * kTP) * sizeof(double) + t.Mu * sizeof(int);
  OB_TRY(ensure_dyn_lds((const void *)k_gram_mfma4<W>, lds));
  {
    ProfScope ps("gram");
    hipLaunchKernelGGL(k_gram_mfma4<W>, dim3((unsigned)npairs, (unsigned)nsplit), dim3(768), lds,
                       cur_stream(), b.bm.p, b.scale.p, t.ucol.p, (int)t.Mu, b.md.Mc, t.cols.p, nb,
                       ntiles, tps, part);
    OB_HIP(hipGetLastError());
  }
  return launch_gram_reduce(part, npairs, (int)nsplit, nb, (int)t.p, sink, false, true);
}

}  // namespace

bool gram_mfma4_supports(const obhip_terms &t) { return t.Mu <= 16 * (uint64_t)kMaxPre && t.W <= 8; }

int launch_gram_mfma4(const obhip_basis &b, obhip_terms &t, const GramSink &sink) {
  if (!gram_mfma4_supports(t))
    return fail(OBHIP_ERR_INVALID,
                "4x4x4 matrix-core Gram kernel: at most 128 basis columns and 8 non-zero levels per term");
  switch (t.W) {
    case 2: return run_gram_mfma4<2>(b, t, sink);
    case 4: return run_gram_mfma4<4>(b, t, sink);
    case 6: return run_gram_mfma4<6>(b, t, sink);
    default: return run_gram_mfma4<8>(b, t, sink);
  }
}

}  // namespace obhip
