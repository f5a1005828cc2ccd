// Dense SPD solve on gfx950: blocked right-looking Cholesky with the forward
// substitution carried along as an extra matrix row, then a blocked backward
// substitution.  Replaces `coeff += solve(h, r)` of lpdf::optnewton
// (src/fit.cpp:120; Armadillo -> LAPACK in the reference).
//
// H is p x p, row-major, full symmetric storage on entry; on exit its lower
// triangle holds L (H = L L^T), the strict upper triangle is scratch.
//
// Several 64-column panels per trailing update (two from p = 4096, four from p = 8192): panel j, a
// strip update of the next 64 columns only with the panels of the pass so far, panel j + 1, ...,
// then ONE pass over the trailing matrix with all panels of the pass (k = 128 or 256, staged
// through LDS in parts).  The trailing update reads and writes the whole trailing triangle; every
// doubling of the panels per pass halves that traffic.
//   k_chol_panel2: every workgroup re-factorises the 64 x 64 diagonal block itself
//                  (cheaper than a launch boundary) -- four 16-column blocks, wave 0 in
//                  registers, the columns to the right on the matrix cores by all waves --
//                  and solves its 64 panel rows against L_jj^T on the matrix cores (details at
//                  the kernel).  One extra "row" is the right-hand side z, which turns the
//                  forward substitution L z = rhs into part of the panel solve.  The solved
//                  rows also go to a k-major scratch copy Wt[k][row] for the update.
//   k_chol_update: trailing update A22 -= L21 L21^T (lower tiles only) on
//                  v_mfma_f64_16x16x4_f64, 64 x 64 tiles (three workgroups per CU; 128 x 128
//                  only beyond 5000 tiles), operands straight from Wt in 16-byte pieces; plus
//                  the matching update of z.  Strip form: the 64 columns of the next panel only.
// Backward: k_chol_back per block from the last to the first: theta_j = L_jj^-T z_j with the
// inverses of the 16 x 16 diagonal sub-blocks the panel step kept, then
// z[0:j) -= L[j, 0:j)^T theta_j.
#include <utility>

#include "obhip_internal.h"

namespace obhip {

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NB = 64;
constexpr int LDP = NB + 1;  // padded LDS leading dimension

__device__ __forceinline__ double readlane_d(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

// ---- panel step ---------------------------------------------------------------------------------
// One workgroup (4 waves) per 64 panel rows; block 0 handles the diagonal block only, the
// last block the rhs row z (which makes the forward substitution L z = rhs part of the panel
// solve).  Every workgroup re-factorises the 64 x 64 diagonal block itself (cheaper than a
// launch boundary):
//   1. wave 0: unblocked right-looking Cholesky, lane = row, 64 columns of the row in
//      registers.  The next pivot's only dependence on the current column goes through a
//      v_readlane (register path); the rest of the rank-1 update reads the finished column
//      back from LDS as broadcast ds_read_b128 and overlaps the next column's rsqrt chain.
//      Waves 1-3 meanwhile stage the workgroup's panel rows in LDS.
//   2. wave w inverts the 16 x 16 diagonal sub-block L_ww (lane = column of the inverse).
//   3. blocked triangular solve X L_jj^T = A on v_mfma_f64_16x16x4_f64, wave w owns 16
//      rows: X_b = (A_b - sum_{c<b} X_c L_bc^T) inv(L_bb)^T for the four 16-column blocks,
//      results passed from the MFMA output layout to the operand layout through LDS.
typedef double d2v __attribute__((ext_vector_type(2)));
constexpr int LT = NB + 2;   // column-buffer pitch (even: 16-byte aligned pairs)
constexpr int SP = 17;       // pitch of the 16 x 16 scratch blocks

__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  return y;
}

// Column C0 + C of the in-register factorisation of the 16 columns [C0, C0 + 16) (lane =
// row); a template so that every index below is a compile-time constant (a[] must stay in
// registers).  The columns stay UNNORMALISED while the block is being eliminated (u = the
// column as the earlier columns left it, w = u / pivot, a[k] -= w u_k): the chain from one
// pivot to the next is then a reciprocal with its two Newton steps, one multiply, one FMA and
// the v_readlane of the next diagonal element; the square roots wait for potrf_block.
constexpr int PB = 16;  // columns per in-register block of the diagonal-block factorisation
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
// the part of a column's rank-1 update that goes through LDS, applied one column later
struct PotrfPending {
  double w;      // u / pivot of that column, lane = row
  double v[PB];  // its u[C0 .. C0 + 16), broadcast
};
// ties the listed registers to this point of the instruction stream: what uses them cannot be
// scheduled above it (left alone, the scheduler hoists the first delayed FMA above the
// reciprocal of the next column, and with it the wait for the LDS read it depends on)
template <int K0>
__device__ __forceinline__ void potrf_pin(double &w, double (&v)[PB]) {
  if constexpr (K0 + 8 <= PB)
    asm volatile("" : "+v"(w), "+v"(v[K0]), "+v"(v[K0 + 1]), "+v"(v[K0 + 2]), "+v"(v[K0 + 3]),
                      "+v"(v[K0 + 4]), "+v"(v[K0 + 5]), "+v"(v[K0 + 6]), "+v"(v[K0 + 7]));
  else if constexpr (K0 + 4 <= PB)
    asm volatile("" : "+v"(w), "+v"(v[K0]), "+v"(v[K0 + 1]), "+v"(v[K0 + 2]), "+v"(v[K0 + 3]));
  else if constexpr (K0 + 2 <= PB)
    asm volatile("" : "+v"(w), "+v"(v[K0]), "+v"(v[K0 + 1]));
  else if constexpr (K0 + 1 <= PB)
    asm volatile("" : "+v"(w), "+v"(v[K0]));
  if constexpr (K0 + 8 <= PB) potrf_pin<K0 + 8>(w, v);
  else if constexpr (K0 + 4 <= PB) potrf_pin<K0 + 4>(w, v);
  else if constexpr (K0 + 2 <= PB) potrf_pin<K0 + 2>(w, v);
}
template <int C, int C0>
__device__ __forceinline__ void potrf_col(double (&a)[PB], double &piv, double &pv, bool &bad,
                                          double *Lt, int lane, PotrfPending &pin, PotrfPending &pout) {
  constexpr int c = C0 + C;
  const double u = lane < c ? 0.0 : a[C];
  a[C] = u;
  // The column goes to LDS and its 16 block entries are requested back as broadcast reads
  // right away; they are consumed a column later (pout), when the round trip is long over.
  if constexpr (C + 2 < PB) {
    Lt[c * LT + lane] = u;  // (potrf_block overwrites the row with the normalised column)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = (C + 2) / 2; i < PB / 2; ++i) {
      const d2v t = *(const d2v *)&Lt[c * LT + C0 + 2 * i];
      pout.v[2 * i] = t.x;
      pout.v[2 * i + 1] = t.y;
    }
  }
  // (the reads are on their way before the reciprocal starts)
  asm volatile("" : "+s"(piv) : : "memory");
  if (!(piv > 0.0)) bad = true;
  pv = lane == c ? piv : pv;
  const double r = rcp_nr(piv);
  double w = u * r;
  pout.w = w;
  // Column C - 1 updated a[C] (register path) before it handed over the pivot; what it owes
  // the columns to the right of this one is in pin, and is consumed only now that this
  // column's reciprocal is through (left alone, the scheduler hoists these FMAs, and the wait
  // for their LDS reads, into the pivot chain).
  if constexpr (C >= 1 && C + 1 < PB) {
    potrf_pin<C + 1>(w, pin.v);
    a[C + 1] = fma(-pin.w, pin.v[C + 1], a[C + 1]);
  }
  if constexpr (C + 1 < PB) {
    // the next pivot depends on this column only through u[c+1]: register path
    a[C + 1] = fma(-w, readlane_d(u, c + 1), a[C + 1]);
    piv = readlane_d(a[C + 1], c + 1);
  }
  if constexpr (C >= 1) {
#pragma unroll
    for (int k = C + 2; k < PB; ++k) a[k] = fma(-pin.w, pin.v[k], a[k]);
  }
}
template <int C0, int... Cs>
__device__ __forceinline__ void potrf_cols(std::integer_sequence<int, Cs...>, double (&a)[PB],
                                           double &piv, double &pv, bool &bad, double *Lt, int lane) {
  PotrfPending p0, p1;  // written by the even / odd columns, read by the next one
  (potrf_col<Cs, C0>(a, piv, pv, bad, Lt, lane, (Cs & 1) ? p0 : p1, (Cs & 1) ? p1 : p0), ...);
}
template <int C0, int... Cs>
__device__ __forceinline__ void potrf_store(std::integer_sequence<int, Cs...>, const double (&a)[PB],
                                            double rsv, double sqv, double *Lt, int lane) {
  // L[lane][C0 + C] = u / sqrt(pivot) below the diagonal, sqrt(pivot) on it, 0 above (u is)
  ((Lt[(C0 + Cs) * LT + lane] = lane == C0 + Cs ? sqv : a[Cs] * readlane_d(rsv, C0 + Cs)), ...);
}
// wave 0: columns [C0, C0 + 16) of the 64 x 64 block P (rows >= C0 matter), L -> Lt, 1/diag -> dinv
template <int C0>
__device__ __forceinline__ void potrf_block(const double *P, bool &bad, double *Lt, double *dinv,
                                            int lane) {
  double a[PB];
#pragma unroll
  for (int k = 0; k < PB; ++k) a[k] = P[lane * LDP + C0 + k];
  double piv = readlane_d(a[0], C0), pv = 1.0;  // pv: lane c keeps the pivot of column c
  potrf_cols<C0>(std::make_integer_sequence<int, PB>{}, a, piv, pv, bad, Lt, lane);
  // the 16 square roots at once, lane = column
  const double rsv = rsqrt_nr(pv);
  double sqv = pv * rsv;
  sqv = fma(fma(-sqv, sqv, pv), 0.5 * rsv, sqv);
  if (lane >= C0 && lane < C0 + PB) dinv[lane] = rsv;
  potrf_store<C0>(std::make_integer_sequence<int, PB>{}, a, rsv, sqv, Lt, lane);
}
// all four waves: P[r][c] -= sum_{k in [C0, C0 + 16)} L[r][k] L[c][k] for the 16 x 16 tiles
// (rt >= ct) right of and below the block, on v_mfma_f64_16x16x4_f64, operands from Lt
template <int C0>
__device__ __forceinline__ void potrf_trailing(double *P, const double *Lt, int wave, int lane) {
  constexpr int B1 = C0 / 16 + 1, NT = 4 - B1;  // tiles per side
  const int t16 = lane & 15, q = lane >> 4;
  int t = 0;
#pragma unroll
  for (int rt = B1; rt < 4; ++rt)
#pragma unroll
    for (int ct = B1; ct <= rt; ++ct, ++t) {
      if ((t & 3) != wave) continue;
      d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double av = Lt[(C0 + 4 * s + q) * LT + 16 * rt + t16];  // L[16 rt + m][C0 + k]
        const double bv = Lt[(C0 + 4 * s + q) * LT + 16 * ct + t16];  // L[16 ct + n][C0 + k]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) P[(16 * rt + q + 4 * r) * LDP + 16 * ct + t16] -= acc[r];
    }
  (void)NT;
}

__global__ void __launch_bounds__(256)
k_chol_panel2(double *__restrict__ H, double *__restrict__ z, double *__restrict__ Wt, int pw, int p,
              int j0, int *__restrict__ info, double *__restrict__ Ljj, int wt_row0,
              double *__restrict__ Iinv) {
  __shared__ __attribute__((aligned(16))) double Lt[NB * LT];  // Lt[c][k] = L[k][c], 0 for k < c
  __shared__ double P[NB * LDP];    // diagonal block; later X (solved rows), per wave 16 rows
  __shared__ double Ap[NB * LDP];   // panel rows
  __shared__ double Iw[4 * 16 * SP];  // inverses of the four 16 x 16 diagonal sub-blocks
  __shared__ double Ts[4 * 16 * SP];  // per-wave transpose scratch
  __shared__ double dinv[NB];         // 1 / L[i][i]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jb = min(NB, p - j0);

  // diagonal block -> LDS (identity padding beyond jb), 16 rows per wave, coalesced
  {
    const double *src = H + (size_t)j0 * p + j0 + min(lane, jb - 1);
    double t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = src[(size_t)min(wave * 16 + i, jb - 1) * p];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = wave * 16 + i;
      P[r * LDP + lane] = (r < jb && lane < jb) ? t[i] : ((r == lane) ? 1.0 : 0.0);
    }
  }
  __syncthreads();

  const bool is_z = blockIdx.x == gridDim.x - 1;
  const int r0 = j0 + NB * (int)blockIdx.x;
  const int nrows = is_z ? 1 : min(NB, p - r0);

  // The diagonal block in four blocks of 16 columns: wave 0 factorises a block in registers
  // (the serial part: a pivot chain of ~170 cycles per column), then all four waves apply it to
  // the columns to its right on the matrix cores.  (One wave doing the whole rank-1 update of
  // every column, 64 columns wide, took 555 cycles per column: 36 of the step's 52 thousand.)
  // Waves 1-3 meanwhile fetch the workgroup's panel rows: the loads go out now, the values
  // are parked in registers and written to Ap before the last barrier.
  bool bad = false;
  double stage[24];
  const int lcs = min(lane, jb - 1);
  if (wave != 0 && blockIdx.x != 0) {
    const double *src = is_z ? z + j0 + lcs : H + (size_t)min(r0, p - 1) * p + j0 + lcs;
    const size_t pitch = is_z ? 0 : (size_t)p;
    const int rmax = max(nrows - 1, 0);
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        stage[it * 8 + i] = src[(size_t)min((wave - 1) * 8 + it * 24 + i, rmax) * pitch];
  }
  if (wave == 0) potrf_block<0>(P, bad, Lt, dinv, lane);
  __syncthreads();
  potrf_trailing<0>(P, Lt, wave, lane);
  __syncthreads();
  if (wave == 0) potrf_block<16>(P, bad, Lt, dinv, lane);
  __syncthreads();
  potrf_trailing<16>(P, Lt, wave, lane);
  __syncthreads();
  if (wave == 0) potrf_block<32>(P, bad, Lt, dinv, lane);
  __syncthreads();
  potrf_trailing<32>(P, Lt, wave, lane);
  __syncthreads();
  if (wave == 0) {
    potrf_block<48>(P, bad, Lt, dinv, lane);
    if (bad && blockIdx.x == 0 && lane == 0) atomicCAS(info, 0, j0 + 1);  // the first bad block stays (later ones inherit NaNs)
  } else if (blockIdx.x != 0) {
    // panel rows of this workgroup -> Ap (zero padding), waves 1-3
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = (wave - 1) * 8 + it * 24 + i;
        if (r < NB) Ap[r * LDP + lane] = (r < nrows && lane < jb) ? stage[it * 8 + i] : 0.0;
      }
  }
  __syncthreads();  // Lt, dinv, Ap complete; P (diagonal block) is free

  if (blockIdx.x == 0) {
    // L_jj goes to a scratch block, NOT into H: the other workgroups of this launch read the
    // diagonal block of H to re-factorise it, and with more workgroups than the GPU holds
    // at once (p >= 16384 on 256 CUs) late ones would see it half overwritten.  The next
    // launch in the stream (k_chol_update or a copy) moves it into place.
    for (int e = tid; e < NB * NB; e += 256) Ljj[e] = (e & 63) <= (e >> 6) ? Lt[(e & 63) * LT + (e >> 6)] : 0.0;
    return;
  }

  // inverse of the 16 x 16 diagonal sub-block `wave`: lane j < 16 solves L_ww x = e_j
  {
    const int j = lane & 15, o = 16 * wave;
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (i == j) ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const double xi = x[i] * dinv[o + i];
      x[i] = xi;
#pragma unroll
      for (int m = i + 1; m < 16; ++m) x[m] = fma(-Lt[(o + i) * LT + o + m], xi, x[m]);
    }
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) Iw[wave * 16 * SP + i * SP + j] = x[i];
    }
  }
  __syncthreads();
  // the four 16 x 16 inverses are kept for the backward substitution (k_chol_back*)
  if (is_z)
    for (int e = tid; e < 4 * 256; e += 256)
      Iinv[(size_t)(j0 / NB) * 1024 + e] = Iw[(e >> 8) * 16 * SP + ((e >> 4) & 15) * SP + (e & 15)];

  // blocked solve, wave w owns rows 16 w .. 16 w + 15; everything below is wave-local
  const int t16 = lane & 15, q = lane >> 4;
  double *X = P + wave * 16 * LDP;    // solved blocks, operand layout source [m][64]
  double *T = Ts + wave * 16 * SP;    // [m][n] scratch
  const double *A = Ap + wave * 16 * LDP;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < b; ++c)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double av = X[t16 * LDP + 16 * c + 4 * s + q];               // X_c[m][k]
        const double bv = Lt[(16 * c + 4 * s + q) * LT + 16 * b + t16];    // L[16b+n][16c+k]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) T[(q + 4 * r) * SP + t16] = A[(q + 4 * r) * LDP + 16 * b + t16] - acc[r];
    d4 xb = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const double av = T[t16 * SP + 4 * s + q];                           // T[m][k]
      const double bv = Iw[b * 16 * SP + t16 * SP + 4 * s + q];            // inv(L_bb)[n][k]
      xb = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, xb, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 16 + q + 4 * r, col = 16 * b + t16;
      X[(q + 4 * r) * LDP + col] = xb[r];
      if (row < nrows && col < jb) {
        if (is_z)
          z[j0 + col] = xb[r];
        else
          H[(size_t)(r0 + row) * p + j0 + col] = xb[r];
      }
    }
  }
  // k-major copy of the solved rows for the trailing update: Wt[k][row], 128-byte segments
  if (!is_z) {
    const int m = lane & 15, kq = lane >> 4;
    if (wave * 16 + m < nrows) {
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const int k = 4 * kk + kq;
        Wt[(size_t)(wt_row0 + k) * pw + r0 + wave * 16 + m] = X[m * LDP + k];
      }
    }
  }
}

// Trailing update of one TS x TS lower tile (TS = 128, or 64 once the trailing matrix has too
// few 128-tiles to go round the CUs): D = H - L21_i L21_j^T.  Latency-shaped: the panel was
// written by the previous kernel (other XCDs' L2s), so every dependent global access is a trip
// to Infinity Cache / HBM.  Everything the block needs is therefore requested up front -- both
// TS x KH operand panels (16-byte pieces of the k-major copy, staged in LDS) and the H tile
// itself, straight into the accumulators -- so the block pays one memory round trip, its MFMAs
// from LDS (256 per wave at TS = 128, k = 64: 16 thousand cycles of the CU's matrix pipe), and
// a store.
// KH = k per staged part (32).  Workgroups per CU by LDS: TS 128 -> 2, TS 64 -> 3 (with KH = 64 and
// one 128-tile workgroup per CU the update of p = 16384 took 50.8 instead of 43.5 ms).
template <int TS, int KH>
__global__ void __launch_bounds__(256, TS == 64 ? 3 : 2)
k_chol_update(double *__restrict__ H, double *__restrict__ z, const double *__restrict__ Wt, int pw,
              int p, int t0 /* first trailing row / column */, int kparts /* staged halves of 64 k */,
              int strip /* 1: only the columns [t0, t0 + 64) */, int ntiles,
              int ljj_j0 /* where the panel's L_jj goes */, int zj0 /* panel whose z part is applied */,
              int zk0 /* its first row in Wt */, const double *__restrict__ Ljj) {
  constexpr int UP = TS + 16;   // LDS pitch of a staged panel row (doubles)
  constexpr int HW = TS / 2;    // wave tile edge; also the 16-byte pieces per staged row
  constexpr int F = TS / 32;    // 16 x 16 MFMA tiles per wave-tile edge
  __shared__ __attribute__((aligned(16))) double Sa[KH * UP];
  __shared__ __attribute__((aligned(16))) double Sb[KH * UP];
  if ((int)blockIdx.x >= ntiles) {
    if ((int)blockIdx.x == ntiles)  // L_jj from the panel step's scratch block into place
      for (int e = threadIdx.x; e < NB * NB; e += 256)
        if ((e & 63) <= (e >> 6) && ljj_j0 + (e >> 6) < p && ljj_j0 + (e & 63) < p)
          H[(size_t)(ljj_j0 + (e >> 6)) * p + ljj_j0 + (e & 63)] = Ljj[e];
    // z[c] -= sum_k z[zj0 + k] * L[c][zj0 + k] for c >= zj0 + 64, all 64 loads in flight at once
    const int c = zj0 + NB + ((int)blockIdx.x - ntiles) * 256 + (int)threadIdx.x;
    if (c < p) {
      double w[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) w[k] = Wt[(size_t)(zk0 + k) * pw + c];
      double s = z[c];
#pragma unroll
      for (int k = 0; k < NB; ++k) s = fma(-z[zj0 + k], w[k], s);
      z[c] = s;
    }
    return;
  }
  // lower-triangular tile pair (bi >= bj); strip form: the tiles (bi, 0)
  int bi = 0, bj = 0;
  if (strip) {
    bi = blockIdx.x;
  } else {
    int rem = blockIdx.x;
    while (rem > bi) {
      rem -= bi + 1;
      ++bi;
    }
    bj = rem;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // strictly upper quadrant of a diagonal tile: no work; strip: the first 64 columns only
  const bool active = strip ? (wn * HW < NB) : !(bi == bj && wn > wm);
  const int t16 = lane & 15, q = lane >> 4;
  const int rbase = t0 + bi * TS + wm * HW, cbase = t0 + bj * TS + wn * HW;

  // H tile into the accumulators (rows / columns beyond p or above the diagonal: zero)
  d4 acc[F][F];
#pragma unroll
  for (int i = 0; i < F; ++i)
#pragma unroll
    for (int j = 0; j < F; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rbase + i * 16 + q + 4 * r, col = cbase + j * 16 + t16;
        acc[i][j][r] = (active && row < p && col <= row) ? H[(size_t)row * p + col] : 0.0;
      }
  constexpr int NU = KH * HW / 256;  // 16-byte pieces per thread and panel part
  const int nparts = kparts * (NB / KH);
  // panels: KH k x TS rows per part, in 16-byte pieces; the pieces of part + 1 are requested
  // before the MFMAs of part, so that only the first part's round trip is exposed
  d2v ga[NU], gb[NU];
  auto fetch = [&](int part) {
    const double *srca = Wt + (size_t)part * KH * pw + t0 + bi * TS;
    const double *srcb = Wt + (size_t)part * KH * pw + t0 + bj * TS;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int e = tid + 256 * u, k = e / HW, c2 = e % HW;
      ga[u] = *(const d2v *)(srca + (size_t)k * pw + 2 * c2);
      gb[u] = *(const d2v *)(srcb + (size_t)k * pw + 2 * c2);
    }
  };
  fetch(0);
  for (int part = 0; part < nparts; ++part) {
    if (part > 0) __syncthreads();  // the previous part has been consumed
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int e = tid + 256 * u, k = e / HW, c2 = e % HW;
      *(d2v *)&Sa[k * UP + 2 * c2] = ga[u];
      *(d2v *)&Sb[k * UP + 2 * c2] = gb[u];
    }
    __syncthreads();
    if (part + 1 < nparts) fetch(part + 1);
    if (active) {
      const double *pa = Sa + q * UP + wm * HW + t16;
      const double *pb = Sb + q * UP + wn * HW + t16;
#pragma unroll 4
      for (int s = 0; s < KH / 4; ++s) {
        double a[F], b[F];
#pragma unroll
        for (int f = 0; f < F; ++f) {
          a[f] = -pa[4 * s * UP + 16 * f];
          b[f] = pb[4 * s * UP + 16 * f];
        }
#pragma unroll
        for (int i = 0; i < F; ++i)
#pragma unroll
          for (int j = 0; j < F; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int i = 0; i < F; ++i)
#pragma unroll
    for (int j = 0; j < F; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rbase + i * 16 + q + 4 * r;
        const int col = cbase + j * 16 + t16;
        if (row < p && col <= row) H[(size_t)row * p + col] = acc[i][j][r];
      }
}

// L^T theta = z for one 64 x 64 diagonal block (Ld, LDS) from the last 16 unknowns to the first,
// with the inverses of its four 16 x 16 diagonal sub-blocks as the panel step left them (Iv:
// [4][16][16] in LDS): theta_b = inv(L_bb)^T z_b, then z[i] -= sum_k L[16 b + k][i] theta_b[k] for
// the unknowns above.  Lane = unknown.  Two groups of 16 v_readlane pairs and 16 FMAs (in four
// partial sums) per sub-block instead of 64 dependent readlane - multiply - FMA steps.
__device__ __forceinline__ double back_solve64(const double *Ld, const double *Iv, double zk, int lane) {
  double th = 0.0;
  const int il = lane & 15;
#pragma unroll
  for (int b = 3; b >= 0; --b) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 16; ++r)
      acc[r & 3] = fma(Iv[b * 256 + r * 16 + il], readlane_d(zk, 16 * b + r), acc[r & 3]);
    const double tb = (acc[0] + acc[1]) + (acc[2] + acc[3]);  // theta_b[il] in every 16-lane row
    if ((lane >> 4) == b) th = tb;
    if (b > 0) {
      double upd[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k = 0; k < 16; ++k)
        upd[k & 3] = fma(Ld[(16 * b + k) * LDP + lane], readlane_d(tb, k), upd[k & 3]);
      if (lane < 16 * b) zk -= (upd[0] + upd[1]) + (upd[2] + upd[3]);
    }
  }
  return th;
}

template <int CB>  // columns of the z update per workgroup
__global__ void __launch_bounds__(256)
k_chol_back(const double *__restrict__ L, double *__restrict__ z, double *__restrict__ theta, int p,
            int j0, const double *__restrict__ Iinv) {
  constexpr int KG = 256 / CB, KR = NB / KG;  // groups of threads along k, rows per group
  __shared__ double Ld[NB * LDP];
  __shared__ double Iv[4 * 256];
  __shared__ double th[NB];
  __shared__ double red[256];
  const int jb = min(NB, p - j0);
  // A workgroup owns 64 columns c < j0 of the update z[c] -= sum_k L[j0 + k][c] theta_k: thread
  // (kg, cl) takes column cl and the 16 rows k = 16 kg .. 16 kg + 15 (a single CU pulls 32 KB
  // instead of 128, four times as many CUs pull).  The rows do not depend on theta, so they
  // are fetched now and arrive under the solve.
  const int cl = threadIdx.x % CB, kg = threadIdx.x / CB;
  const int c = (int)blockIdx.x * CB + cl;
  double lc[KR];
  if (c < j0) {
#pragma unroll
    for (int k = 0; k < KR; ++k) lc[k] = L[(size_t)(j0 + min(KR * kg + k, jb - 1)) * p + c];
  }
  {
    double t[NB * NB / 256], ti[4];
#pragma unroll
    for (int i = 0; i < NB * NB / 256; ++i) {
      const int e = threadIdx.x + i * 256;
      t[i] = L[(size_t)(j0 + min(e / NB, jb - 1)) * p + j0 + min(e % NB, jb - 1)];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) ti[i] = Iinv[(size_t)(j0 / NB) * 1024 + threadIdx.x + i * 256];
#pragma unroll
    for (int i = 0; i < NB * NB / 256; ++i) {
      const int e = threadIdx.x + i * 256;
      const int r = e / NB, cc = e % NB;
      Ld[r * LDP + cc] = (r < jb && cc < jb && cc <= r) ? t[i] : ((r == cc) ? 1.0 : 0.0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) Iv[threadIdx.x + i * 256] = ti[i];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const double zk = back_solve64(Ld, Iv, lane < jb ? z[j0 + lane] : 0.0, lane);
    th[lane] = lane < jb ? zk : 0.0;  // rows past the end (ragged last block) add nothing
    if (blockIdx.x == 0 && lane < jb) theta[j0 + lane] = zk;
  }
  __syncthreads();
  double sum = 0.0;
  if (c < j0) {
#pragma unroll
    for (int k = 0; k < KR; ++k) sum = fma(lc[k], th[KR * kg + k], sum);
  }
  red[kg * CB + cl] = sum;
  __syncthreads();
  if (kg == 0 && c < j0) {
    double tot = 0.0;
#pragma unroll
    for (int g = 0; g < KG; ++g) tot += red[g * CB + cl];
    z[c] -= tot;
  }
}

__global__ void k_form_hessian(double *__restrict__ G, const double *__restrict__ prec, double e2,
                               int p, double *__restrict__ diagH) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)p * p) return;
  const int i = (int)(idx / p), j = (int)(idx % p);
  double v = e2 * G[idx];
  if (i == j) {
    v += prec[i];
    if (diagH) diagH[i] = v;
  }
  G[idx] = v;
}

}  // namespace

// z (p), info (64 doubles reserved), k-major copies of the panels of a pass Wt (64 rows of
// chol_pitch(p) doubles each), scratch block for L_jj (64 x 64), the inverses of the 16 x 16
// diagonal sub-blocks (1024 doubles per 64 columns)
static uint64_t chol_pitch(uint64_t p) { return (p + 127) / 128 * 128 + 128; }
// Panels of the trailing pass that starts with m rows and columns left (OBHIP_CHOL_PANELS: 1, 2, 4
// or 8 throughout, for A/B runs).  The more panels, the fewer passes over the trailing matrix (HBM:
// 8 flops per byte at two panels) but the more work in the strips between the panels, which a few
// workgroups do: eight while the trailing matrix is large, two once it is small.
static int chol_panels_at(uint64_t p, uint64_t m) {
  static const int forced = [] {
    const char *e = getenv("OBHIP_CHOL_PANELS");
    const int v = e ? atoi(e) : 0;
    return v == 1 || v == 2 || v == 4 || v == 8 ? v : 0;
  }();
  static const uint64_t m8 = getenv("OBHIP_CHOL_M8") ? strtoull(getenv("OBHIP_CHOL_M8"), nullptr, 10) : 8192;
  static const uint64_t m4 = getenv("OBHIP_CHOL_M4") ? strtoull(getenv("OBHIP_CHOL_M4"), nullptr, 10) : 4096;
  if (forced) return forced;
  if (p < 3072) return 1;  // (p = 1024 / 2048: 0.421 / 0.889 ms with one panel per pass, 0.434 / 0.894 with two; 3072: 1.474 / 1.441)
  return m >= m8 ? 8 : (m >= m4 ? 4 : 2);
}
// (the most any pass takes: the first)
static int chol_panels(uint64_t p) { return chol_panels_at(p, p); }
uint64_t newton_workspace_bytes(uint64_t p) {
  return (p + 64 + (uint64_t)chol_panels(p) * NB * chol_pitch(p) + NB * NB + ((p + NB - 1) / NB) * 1024) *
         sizeof(double);
}

int launch_form_hessian(uint64_t p, double *d_G, const double *d_prec, double e2, double *d_diagH) {
  ProfScope ps("form_hessian");
  const size_t total = (size_t)p * p;
  hipLaunchKernelGGL(k_form_hessian, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     cur_stream(), d_G, d_prec, e2, (int)p, d_diagH);
  OB_HIP(hipGetLastError());
  return 0;
}

int launch_newton_solve(uint64_t p64, double *d_H, const double *d_rhs, double *d_theta, void *d_ws,
                        uint64_t ws_bytes) {
  if (ws_bytes < newton_workspace_bytes(p64)) return fail(OBHIP_ERR_INVALID, "workspace too small");
  if (p64 > (1u << 30)) return fail(OBHIP_ERR_INVALID, "p too large");
  const int p = (int)p64;
  double *z = (double *)d_ws;
  int *info = (int *)(z + p);
  double *Wt = z + p + 64;
  const int pw = (int)chol_pitch(p64);
  double *Ljj = Wt + (size_t)chol_panels(p64) * NB * pw;
  double *Iinv = Ljj + NB * NB;  // [p / 64][4][16][16]: inverses of the 16 x 16 diagonal sub-blocks
  hipStream_t st = cur_stream();
  OB_HIP(hipMemsetAsync(Wt, 0, sizeof(double) * chol_panels(p64) * NB * pw, st));  // rows beyond p stay zero
  OB_HIP(hipMemcpyAsync(z, d_rhs, sizeof(double) * p, hipMemcpyDeviceToDevice, st));
  OB_HIP(hipMemsetAsync(info, 0, sizeof(int), st));
  {
    ProfScope ps("cholesky");
    // L_jj of the last panel of a pass: nothing follows that would move it into place
    auto place_ljj = [&](int j0) -> int {
      const int jb = std::min(NB, p - j0);
      OB_HIP(hipMemcpy2DAsync(d_H + (size_t)j0 * p + j0, (size_t)p * sizeof(double), Ljj,
                              NB * sizeof(double), jb * sizeof(double), jb, hipMemcpyDeviceToDevice,
                              st));
      return 0;
    };
    auto panel = [&](int j0, int wt_row0) {
      const int nrowblk = (p - j0 + NB - 1) / NB;  // block 0 = diagonal block
      hipLaunchKernelGGL(k_chol_panel2, dim3((unsigned)(nrowblk + 1)), dim3(256), 0, st, d_H, z, Wt,
                         pw, p, j0, info, Ljj, wt_row0, Iinv);
    };
    // Several panels per trailing pass (chol_panels): panel, then a strip update of the next 64
    // columns only with all panels of the pass so far (left-looking inside the pass), ..., then ONE
    // pass over the trailing matrix with all of them (k = 64 x panels).  The trailing update reads
    // and writes the whole trailing triangle, 8 flops per byte at k = 128: two panels halve that
    // traffic against one (p = 16384: 60.1 -> 41.2 ms; p = 4096: 2.26 -> 2.17 ms), four halve it again
    // (36.0 ms), eight again (34.6); by the rows left at the start of a pass (chol_panels_at) 33.1 ms.
    // (Strips with 64 instead of 32 k per staged part: 33.7 against 33.3 ms, not kept.)
    // 64 x 64 tiles (a quarter of the MFMAs and loads per workgroup, three workgroups per CU)
    // unless there are thousands of 128 x 128 ones: p = 4096 2.67 -> 2.27 ms with them
    // throughout, p = 16384 43.8 -> 41.0 ms with them below 5000 tiles
    const int t64_below = getenv("OBHIP_CHOL_T64") ? atoi(getenv("OBHIP_CHOL_T64")) : 5000;
    // trailing rows / columns from t0 on with the panel(s) in Wt; strip: the next 64 columns only
    auto update = [&](int t0, int kparts, int strip, int ljj_j0, int zj0, int zk0) -> int {
      const int m = p - t0;
      const int nt128 = (m + 127) / 128, work128 = strip ? nt128 : nt128 * (nt128 + 1) / 2;
      const int nz = (m + 255) / 256;
      if (work128 < t64_below) {
        const int nt = (m + 63) / 64, work = strip ? nt : nt * (nt + 1) / 2;
        hipLaunchKernelGGL((k_chol_update<64, 32>), dim3((unsigned)(work + nz)), dim3(256), 0, st, d_H, z,
                           Wt, pw, p, t0, kparts, strip, work, ljj_j0, zj0, zk0, Ljj);
      } else {
        hipLaunchKernelGGL((k_chol_update<128, 32>), dim3((unsigned)(work128 + nz)), dim3(256), 0, st, d_H,
                           z, Wt, pw, p, t0, kparts, strip, work128, ljj_j0, zj0, zk0, Ljj);
      }
      return 0;
    };
    for (int j0 = 0, done = 0, npan = 1; !done && j0 < p; j0 += npan * NB) {
      npan = chol_panels_at(p64, (uint64_t)(p - j0));
      for (int i = 0; i < npan; ++i) {
        const int jp = j0 + i * NB;
        panel(jp, i * NB);
        if (p - (jp + NB) <= 0) {
          OB_TRY(place_ljj(jp));
          done = 1;
          break;
        }
        // not the last panel of the pass: the next panel's 64 columns (all rows below) with the
        // panels so far; the last: the trailing matrix once with all panels.  Either way z for
        // everything below with this panel, and this panel's L_jj into place.
        OB_TRY(update(jp + NB, i + 1, i + 1 < npan ? 1 : 0, jp, jp, i * NB));
      }
    }
    OB_HIP(hipGetLastError());
  }
  {
    ProfScope ps("backsolve");
    // blocks of 64 columns from the last to the first.  (Two blocks per launch, with the
    // hand-over inside the workgroup, was measured: 0.43 instead of 0.39 ms at p = 4096 -- the
    // launches are not what a step costs.)
    for (int j0 = (p - 1) / NB * NB; j0 >= 0; j0 -= NB) {
      // 64 columns of the z update per workgroup (32: 0.29 instead of 0.30 ms at p = 4096 but
      // 1.43 instead of 1.36 at 16384; 256, one thread per column: 0.39 / 1.85)
      const int nblk = std::max(1, (j0 + NB - 1) / NB);
      hipLaunchKernelGGL(k_chol_back<64>, dim3((unsigned)nblk), dim3(256), 0, st, d_H, z, d_theta, p, j0, Iinv);
    }
    OB_HIP(hipGetLastError());
  }
  int h_info = 0;
  OB_HIP(hipMemcpyAsync(&h_info, info, sizeof(int), hipMemcpyDeviceToHost, st));
  OB_HIP(hipStreamSynchronize(st));
  if (h_info != 0)
    return fail(OBHIP_ERR_NUMERIC, "Hessian is not positive definite (block starting at column " +
                                       std::to_string(h_info - 1) + ")");
  return 0;
}

}  // namespace obhip
