// G = B^T B on the FP64 matrix cores of gfx950, without ever forming B.
//
// Replaces loglik_std::hess (src/lpdfs/loglik_std.cpp:170-173, the
// basismat.t() * basismat gemm on the materialised n x p design matrix built by
// getm_, src/linalg.cpp:647-715).  At n = 1e6, p = 4096 the design matrix would
// be 32.8 GB; here each workgroup regenerates the operand fragments it needs
// from a 64-row LDS tile of the factored basis (<= 100 columns at that config).
//
// Decomposition
//   - output tile 128 x 128 terms per workgroup, upper-triangular tile pairs
//     (I <= J) only; 4 waves as 2 x 2, each wave owns 64 x 64 = 4 x 4 MFMA
//     tiles of v_mfma_f64_16x16x4_f64 (16 accumulators x 4 f64 = 128 VGPRs);
//   - the n dimension (the MFMA "k" index) is split over gridDim.y workgroups
//     per tile pair; each writes its 128 x 128 partial, k_gram_reduce sums the
//     partials in a fixed order (bit-reproducible, no atomics) and mirrors the
//     result into the lower triangle;
//   - MFMA operand element of lane l: A[i = l & 15][k = l >> 4] =
//     B[row 4s + (l >> 4)][term i0 + (l & 15)].  The term is a per-lane
//     constant for the whole kernel, so its column list lives in registers as
//     pre-swizzled LDS addresses; a fragment element costs W ds_read_b64 + W
//     v_mul_f64 and is never staged anywhere.  basescale^2 is folded into the
//     A operand only.
#include "obhip_internal.h"
#include "device_common.h"

namespace obhip {

int launch_gram_reduce(const double *part, int npairs, int nsplit, int nb, int p, double *d_G);

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int kGT = 128;  // output tile edge (terms)

// byte address of element (u, r = 4s + q): ((u*64) | swz(u)) ^ r, times 8.
// The lane-constant part ((u*64 | swz) ^ q) * 8 is precomputed; the wave-
// uniform part (4s * 8) is XORed in per step.
__device__ __forceinline__ int col_addr(int u, int q) {
  return (((u * kTileRows) | tile_swz(u)) ^ q) * 8;
}

template <int W>
__global__ void __launch_bounds__(256, 2)
k_gram(const double *__restrict__ bm, const double *__restrict__ scale,
       const uint32_t *__restrict__ ucol, int Mu, uint64_t Mc, const uint16_t *__restrict__ cols,
       int nb, uint64_t ntiles, uint64_t tiles_per_split, double *__restrict__ part) {
  extern __shared__ double lds[];
  double *s2 = lds + (size_t)Mu * kTileRows;  // basescale^2 of the 64 rows
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int t16 = lane & 15, q = lane >> 4;

  // decode the upper-triangular pair index
  int I = 0, rem = blockIdx.x;
  while (rem >= nb - I) {
    rem -= nb - I;
    ++I;
  }
  const int J = I + rem;

  int ca[4][W], cb[4][W];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int ka = I * kGT + wm * 64 + f * 16 + t16;
    const int kb = J * kGT + wn * 64 + f * 16 + t16;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      ca[f][w] = col_addr(cols[(size_t)ka * W + w], q);
      cb[f][w] = col_addr(cols[(size_t)kb * W + w], q);
    }
  }

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

  const uint64_t t0 = (uint64_t)blockIdx.y * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);
  const char *ldsb = (const char *)lds;
  for (uint64_t tile = t0; tile < t1; ++tile) {
    __syncthreads();
    stage_tile<false, true>(lds, bm + tile * Mc * kTileRows, ucol, Mu, threadIdx.x, 256);
    if (threadIdx.x < kTileRows) {
      const double s = scale[tile * kTileRows + threadIdx.x];
      s2[threadIdx.x] = s * s;
    }
    __syncthreads();
#pragma unroll 2
    for (int s = 0; s < 16; ++s) {
      const int rx = s * 32;  // (4 s) * 8 bytes
      double a[4], b[4];
      const double sv = s2[4 * s + q];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        double va = sv, vb = 1.0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
          va *= *(const double *)(ldsb + (ca[f][w] ^ rx));
          vb *= *(const double *)(ldsb + (cb[f][w] ^ rx));
        }
        a[f] = va;
        b[f] = vb;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  double *out = part + ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * (kGT * kGT);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * 64 + i * 16 + q + 4 * r;
        const int col = wn * 64 + j * 16 + t16;
        out[row * kGT + col] = acc[i][j][r];
      }
}

// sum the row-split partials and scatter tile (I, J) and its mirror into the
// full symmetric p x p matrix.
__global__ void __launch_bounds__(256)
k_gram_reduce(const double *__restrict__ part, int npairs, int nsplit, int nb, int p,
              double *__restrict__ G) {
  __shared__ double S[64 * 65];  // one 64 x 64 quadrant, so that the mirror goes out in rows too
  int I = 0, rem = blockIdx.x;
  while (rem >= nb - I) {
    rem -= nb - I;
    ++I;
  }
  const int J = I + rem;
  const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
  for (int qd = 0; qd < 4; ++qd) {
    const int qr = qd >> 1, qc = qd & 1;
    if (I == J && qr > qc) continue;  // diagonal tiles: the lower-left quadrant is the mirror
    // sum of the row-split partials, 64 consecutive doubles per wave load
    for (int r = r4; r < 64; r += 4) {
      const int e = (qr * 64 + r) * kGT + qc * 64 + c;
      double s = 0.0;
#pragma unroll 8
      for (int k = 0; k < nsplit; ++k) s += part[((uint64_t)k * npairs + blockIdx.x) * (kGT * kGT) + e];
      S[r * 65 + c] = s;
    }
    __syncthreads();
    const int gi0 = I * kGT + qr * 64, gj0 = J * kGT + qc * 64;
    for (int r = r4; r < 64; r += 4) {
      // G[gi0 + r][gj0 + c] and, mirrored, G[gj0 + r][gi0 + c] = S[c][r]
      if (gi0 + r < p && gj0 + c < p && !(I == J && qr == qc && c < r))
        G[(uint64_t)(gi0 + r) * p + gj0 + c] = S[r * 65 + c];
      if (gj0 + r < p && gi0 + c < p && !(I == J && qr == qc && c >= r))
        G[(uint64_t)(gj0 + r) * p + gi0 + c] = S[c * 65 + r];
    }
    __syncthreads();
  }
}

template <int W>
int run_gram(const obhip_basis &b, obhip_terms &t, double *d_G) {
  const int nb = (int)((t.p + kGT - 1) / kGT);
  const int npairs = nb * (nb + 1) / 2;
  const uint64_t ntiles = b.n_pad / kTileRows;
  // enough workgroups for ~16 rounds over 256 CUs x 2 resident blocks
  uint64_t nsplit = std::max<uint64_t>(1, (8192 + npairs - 1) / npairs);
  nsplit = std::min(nsplit, std::max<uint64_t>(1, ntiles / 8));
  const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
  nsplit = (ntiles + tps - 1) / tps;
  double *part = nullptr;
  OB_TRY(const_cast<obhip_basis &>(b).workspace(
      (size_t)nsplit * npairs * kGT * kGT * sizeof(double), (void **)&part));
  const size_t lds = (t.Mu * kTileRows + kTileRows) * sizeof(double);
  if (lds > 64 * 1024)
    OB_HIP(hipFuncSetAttribute((const void *)k_gram<W>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
  {
    ProfScope ps("gram");
    hipLaunchKernelGGL(k_gram<W>, dim3((unsigned)npairs, (unsigned)nsplit), dim3(256), lds,
                       cur_stream(), b.bm.p, b.scale.p, t.ucol.p, (int)t.Mu, b.md.Mc, t.cols.p, nb,
                       ntiles, tps, part);
    OB_HIP(hipGetLastError());
  }
  return launch_gram_reduce(part, npairs, (int)nsplit, nb, (int)t.p, d_G);
}

}  // namespace

int launch_gram_reduce(const double *part, int npairs, int nsplit, int nb, int p, double *d_G) {
  ProfScope ps("gram_reduce");
  hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)npairs), dim3(256), 0, cur_stream(), part, npairs,
                     nsplit, nb, p, d_G);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_gram_valu(const obhip_basis &b, obhip_terms &t, double *d_G);
bool gram_valu_supports(const obhip_terms &t);
int launch_gram_mfma4(const obhip_basis &b, obhip_terms &t, double *d_G);
bool gram_mfma4_supports(const obhip_terms &t);

int launch_gram_panel(const obhip_basis &b, obhip_terms &t, double *d_G);
bool gram_panel_supports(const obhip_basis &b, const obhip_terms &t);

// 0 = automatic (materialised-B 4x4x4 matrix-core kernel when n x p doubles fit in
// half of the free HBM, else the fused 4x4x4 kernel when the terms fit it, else 16x16x4),
// 1 = v_mfma_f64_16x16x4_f64 kernel, 2 = v_fma_f64 register-tiled kernel,
// 3 = fused v_mfma_f64_4x4x4_4b_f64 kernel, 4 = materialised-B 4x4x4 kernel
static int g_gram_backend = 0;
void set_gram_backend(int b) { g_gram_backend = b; }
int get_gram_backend() { return g_gram_backend; }

int launch_gram(const obhip_basis &b, obhip_terms &t, double *d_G) {
  OB_TRY(t.prepare(b.md.cap, b.md.dims_h));
  if (g_gram_backend == 4 || (g_gram_backend == 0 && gram_panel_supports(b, t)))
    return launch_gram_panel(b, t, d_G);
  if (g_gram_backend == 3 || (g_gram_backend == 0 && gram_mfma4_supports(t)))
    return launch_gram_mfma4(b, t, d_G);
  if (g_gram_backend == 2) return launch_gram_valu(b, t, d_G);
  if (t.Mu > 300)
    return fail(OBHIP_ERR_INVALID, "terms touch too many basis columns for the LDS tile");
  // the column-list table is padded to a multiple of 256 terms, which covers
  // the 128-term tiles
  switch (t.W) {
    case 2: return run_gram<2>(b, t, d_G);
    case 4: return run_gram<4>(b, t, d_G);
    case 6: return run_gram<6>(b, t, d_G);
    case 8: return run_gram<8>(b, t, d_G);
    default:
      return fail(OBHIP_ERR_INVALID,
                  "Gram kernel supports terms with at most 8 non-zero levels; use the CG back end");
  }
}

}  // namespace obhip
