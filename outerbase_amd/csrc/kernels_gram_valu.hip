// G = B^T B on the FP64 vector pipe of gfx950 (register-tiled, LDS-staged).
//
// Why not the matrix cores: measured on MI355X (tools/fp64_pipes_bench.hip,
// profiles/r01_fp64_pipes_microbench.txt) v_mfma_f64_16x16x4_f64 tops out at
// 36 TFLOP/s with 1-2 waves per SIMD and 49 TFLOP/s with 8, while plain
// v_fma_f64 sustains 70-76 TFLOP/s (97 % of the 78.6 TF datasheet figure)
// already at 1-2 waves per SIMD, and the two do not add up when mixed.  The
// MFMA kernel (kernels_gram.hip) is kept as an alternative back end.
//
// Decomposition (same tiling contract as kernels_gram.hip: 128 x 128 output
// tiles over upper-triangular tile pairs, rows split over gridDim.y, partials
// summed by k_gram_reduce):
//   - rows are consumed in chunks of 16.  Per chunk the workgroup (4 waves)
//       1. writes the 16-row slice of the used basemat columns to LDS (its
//          global loads were issued one chunk earlier and sit in registers
//          during the previous chunk's arithmetic),
//       2. generates the two operand panels T[256 terms][16 rows]: thread t
//          owns term t of the tile pair (A block 0..127 scaled by basescale^2,
//          B block 128..255), its W column addresses live in registers, and it
//          multiplies W runs of 16 consecutive rows (ds_read_b128) together,
//       3. accumulates: each lane owns an 8 x 8 register tile of the 64 x 64
//          wave tile; per pair of rows it reads 8 + 8 operand pairs with
//          ds_read_b128 and issues 128 v_fma_f64.
//     Two barriers per chunk (after 1 and after 2); the slice buffer and the
//     panels are never written while another wave can still read them.
//   - LDS layouts: slice sub[column][18] and panels T[term][18] doubles (16
//     rows + 2 pad).  Lane (ly, lx) = (lane >> 3, lane & 7) owns output rows
//     {16c + 2ly + e} and columns {16c + 2lx + e}, c = 0..3, e = 0..1, of the
//     wave tile, so for a fixed (c, e) the 8 distinct addresses of a
//     ds_read_b128 are 288 bytes apart = 32 bytes apart modulo the 256-byte
//     bank row: conflict-free, and lanes sharing ly (or lx) read one address
//     (broadcast).  The 144-byte row pitch also spreads the per-thread panel
//     writes and the per-term slice reads over all banks.
#include "obhip_internal.h"
#include "device_common.h"

namespace obhip {

int launch_gram_reduce(const double *part, int npairs, int nsplit, int nb, int p, double *d_G);

namespace {

constexpr int kGT = 128;    // output tile edge (terms)
constexpr int kCR = 8;      // rows per chunk
constexpr int kLD = 10;     // padded leading dimension (doubles) of slice and panels
constexpr int kMaxPre = 4;  // prefetch registers per thread => Mu <= 32 * 4 columns
constexpr int kChunksPerTile = kTileRows / kCR;

typedef double d2 __attribute__((ext_vector_type(2)));

template <int W>
__global__ void __launch_bounds__(256, 2)
k_gram_valu(const double *__restrict__ bm, const double *__restrict__ scale,
            const uint32_t *__restrict__ ucol, int Mu, uint64_t Mc,
            const uint16_t *__restrict__ cols, int nb, uint64_t ntiles,
            uint64_t tiles_per_split, double *__restrict__ part) {
  extern __shared__ double lds[];
  // two generations of everything: [buf][...]
  const int subsz = (Mu + 1) * kLD;       // slice: Mu columns + basescale^2 as column Mu
  const int tsz = 2 * kGT * kLD;          // panels: A block then B block
  double *sub = lds;                      // [2][Mu + 1][10]
  double *T = sub + 2 * subsz;            // [2][256][10]
  int *lu = (int *)(T + 2 * tsz);         // [Mu] ucol[u] * 64

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ly = lane >> 3, lx = lane & 7;

  int I = 0, rem = blockIdx.x;
  while (rem >= nb - I) {
    rem -= nb - I;
    ++I;
  }
  const int J = I + rem;

  // this thread's term of the tile pair; offsets (doubles) of its columns in a slice
  const bool isA = tid < kGT;
  const int term = isA ? I * kGT + tid : J * kGT + (tid - kGT);
  int coff[W];
#pragma unroll
  for (int w = 0; w < W; ++w) coff[w] = (int)cols[(size_t)term * W + w] * kLD;
  const int soff = Mu * kLD;  // the basescale^2 pseudo-column

  double acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.0;

  const uint64_t t0 = (uint64_t)blockIdx.y * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);
  const int nchunks = (int)(t1 > t0 ? (t1 - t0) * kChunksPerTile : 0);

  for (int u = tid; u < Mu; u += 256) lu[u] = (int)ucol[u] * kTileRows;
  __syncthreads();

  // slice element (u, r) of chunk ch lives at
  //   bm[(tile * Mc + ucol[u]) * 64 + (ch % 8) * 8 + r];
  // thread t moves row r = t & 7 of columns u = (t >> 3) + 32 q.
  const int pr = tid & 7, pu = tid >> 3;
  double pre[kMaxPre];
  double pres = 0.0;  // basescale of row pr (threads 0..7)
  auto prefetch = [&](int ch) {
    const uint64_t tile = t0 + ch / kChunksPerTile;
    const int roff = (ch % kChunksPerTile) * kCR + pr;
    const double *src = bm + tile * Mc * kTileRows + roff;
#pragma unroll
    for (int q = 0; q < kMaxPre; ++q) {
      const int u = pu + 32 * q;
      pre[q] = u < Mu ? src[lu[u]] : 0.0;
    }
    if (tid < kCR) pres = scale[tile * kTileRows + roff];
  };
  auto put_slice = [&](int buf) {
    double *dst = sub + buf * subsz;
#pragma unroll
    for (int q = 0; q < kMaxPre; ++q) {
      const int u = pu + 32 * q;
      if (u < Mu) dst[u * kLD + pr] = pre[q];
    }
    if (tid < kCR) dst[soff + tid] = pres * pres;
  };
  // panel row of this thread's term for the chunk whose slice is in `buf`
  auto gen_panel = [&](int buf) {
    const double *src = sub + buf * subsz;
    d2 v[kCR / 2];
    if (isA) {
#pragma unroll
      for (int r = 0; r < kCR / 2; ++r) v[r] = *(const d2 *)(src + soff + 2 * r);
    } else {
#pragma unroll
      for (int r = 0; r < kCR / 2; ++r) v[r] = d2{1.0, 1.0};
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
      d2 c[kCR / 2];
#pragma unroll
      for (int r = 0; r < kCR / 2; ++r) c[r] = *(const d2 *)(src + coff[w] + 2 * r);
#pragma unroll
      for (int r = 0; r < kCR / 2; ++r) v[r] *= c[r];
    }
    double *dst = T + buf * tsz + tid * kLD;
#pragma unroll
    for (int r = 0; r < kCR / 2; ++r) *(d2 *)(dst + 2 * r) = v[r];
  };

  // prologue: establish the loop invariant at the first barrier
  if (nchunks > 0) {
    prefetch(0);
    put_slice(0);
    if (nchunks > 1) prefetch(1);
    __syncthreads();
    gen_panel(0);
    if (nchunks > 1) put_slice(1);
    if (nchunks > 2) prefetch(2);
  }
  __syncthreads();

  const int aoff = (wm * 64 + 2 * ly) * kLD;
  const int boff = (kGT + wn * 64 + 2 * lx) * kLD;

  // Invariant at the top of iteration c: panels T[c & 1] are complete, slice
  // sub[(c + 1) & 1] holds chunk c + 1, `pre` holds (or is fetching) chunk c + 2.
  // One barrier per iteration; nothing written in iteration c is read before it.
  for (int c = 0; c < nchunks; ++c) {
    const int cur = c & 1, nxt = cur ^ 1;
    // (b) next chunk's panels from its slice
    if (c + 1 < nchunks) gen_panel(nxt);
    // (c) chunk c + 2's slice replaces chunk c's, then fetch chunk c + 3
    if (c + 2 < nchunks) put_slice(cur);
    if (c + 3 < nchunks) prefetch(c + 3);
    // (a) rank-8 update of the register tiles from T[cur], software pipelined:
    //     the B operands of row pair kp + 1 and the A operand two steps ahead are
    //     in flight while the 16 FMAs of the current (kp, i) step issue.
    {
      const double *ta = T + cur * tsz + aoff;
      const double *tb = T + cur * tsz + boff;
      auto lda = [&](int s) -> d2 {  // step s = kp * 8 + i
        const int kp = s >> 3, i = s & 7;
        return *(const d2 *)(ta + ((i >> 1) * 16 + (i & 1)) * kLD + 2 * kp);
      };
      auto ldb = [&](int kp, int j) -> d2 {
        return *(const d2 *)(tb + ((j >> 1) * 16 + (j & 1)) * kLD + 2 * kp);
      };
      d2 bcur[8], bnxt[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) bcur[j] = ldb(0, j);
      d2 a0 = lda(0), a1 = lda(1), a2 = lda(2), a3;
#pragma unroll
      for (int kp = 0; kp < kCR / 2; ++kp) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int s = kp * 8 + i;
          if (s + 3 < 8 * (kCR / 2)) a3 = lda(s + 3);
          if (kp + 1 < kCR / 2) bnxt[i] = ldb(kp + 1, i);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            acc[i][j] = fma(a0.x, bcur[j].x, acc[i][j]);
            acc[i][j] = fma(a0.y, bcur[j].y, acc[i][j]);
          }
          a0 = a1;
          a1 = a2;
          a2 = a3;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) bcur[j] = bnxt[j];
      }
    }
    __syncthreads();
  }

  double *out = part + ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * (kGT * kGT);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = wm * 64 + (i >> 1) * 16 + 2 * ly + (i & 1);
      const int col = wn * 64 + (j >> 1) * 16 + 2 * lx + (j & 1);
      out[row * kGT + col] = acc[i][j];
    }
}

template <int W>
int run_gram_valu(const obhip_basis &b, obhip_terms &t, double *d_G) {
  const int nb = (int)((t.p + kGT - 1) / kGT);
  const int npairs = nb * (nb + 1) / 2;
  const uint64_t ntiles = b.n_pad / kTileRows;
  uint64_t nsplit = std::max<uint64_t>(1, (8192 + npairs - 1) / npairs);
  nsplit = std::min(nsplit, std::max<uint64_t>(1, ntiles / 8));
  const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  double *part = nullptr;
  OB_TRY(const_cast<obhip_basis &>(b).workspace((size_t)nsplit * npairs * kGT * kGT * sizeof(double),
                                                (void **)&part));
  const size_t lds = 2 * ((t.Mu + 1) * kLD + 2 * kGT * kLD) * sizeof(double) + t.Mu * sizeof(int);
  if (lds > 64 * 1024)
    OB_HIP(hipFuncSetAttribute((const void *)k_gram_valu<W>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  {
    ProfScope ps("gram");
    hipLaunchKernelGGL(k_gram_valu<W>, dim3((unsigned)npairs, (unsigned)nsplit), dim3(256), lds,
                       cur_stream(), b.bm.p, b.scale.p, t.ucol.p, (int)t.Mu, b.md.Mc, t.cols.p, nb,
                       ntiles, tps, part);
    OB_HIP(hipGetLastError());
  }
  return launch_gram_reduce(part, npairs, (int)nsplit, nb, (int)t.p, d_G);
}

}  // namespace

bool gram_valu_supports(const obhip_terms &t) {
  return t.Mu <= 32 * (uint64_t)kMaxPre && t.W <= 8;
}

int launch_gram_valu(const obhip_basis &b, obhip_terms &t, double *d_G) {
  if (!gram_valu_supports(t))
    return fail(OBHIP_ERR_INVALID,
                "vector-pipe Gram kernel: at most 128 basis columns and 8 non-zero levels per term");
  switch (t.W) {
    case 2: return run_gram_valu<2>(b, t, d_G);
    case 4: return run_gram_valu<4>(b, t, d_G);
    case 6: return run_gram_valu<6>(b, t, d_G);
    default: return run_gram_valu<8>(b, t, d_G);
  }
}

}  // namespace obhip
