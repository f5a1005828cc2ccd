// G = B^T B on the FP64 matrix cores of gfx950 (v_mfma_f64_4x4x4_4b_f64) from a
// design matrix materialised ONCE in HBM -- the default Gram path when memory allows.
//
// Replaces loglik_std's getm_ + basismat.t() * basismat (src/linalg.cpp:647-715,
// src/lpdfs/loglik_std.cpp:48,170-173).  Like the reference it forms B, but row-major
// and only as a staging area for the matrix cores (32.8 GB at n = 1e6, p = 4096 --
// 11 % of the 288 GB of one MI355X).
//
// Why not generate the operand panels inside the Gram kernel (kernels_gram_mfma4.hip
// does, and stays as the low-memory fallback): anything that is not an MFMA is expensive
// next to a saturated FP64 matrix pipe.  tools/issue_bench.hip: a third wave on such a
// SIMD gets one instruction issued every 50-270 cycles even at s_setprio 3;
// tools/mfma4x4_lds_bench.hip: every integer VALU instruction inside the MFMA waves' own
// stream costs 7-14 matrix-pipe cycles.  So the hot loop here holds MFMA, LDS-read, VMEM
// and SALU instructions only:
//   * panels come from B by LDS-direct loads (global_load_lds_dwordx4: one wave instruction
//     moves one 1-KB row segment of a 128-term panel, no registers, no VALU), issued by
//     the MFMA waves themselves one chunk ahead and tracked with vmcnt;
//   * operand reads are hand-issued ds_read_b64 with immediate offsets (a dynamic buffer
//     index would need VALU address arithmetic) and hand-placed s_waitcnt lgkmcnt;
//   * two 4-wave blocks are resident per CU, so that one block's barrier / load waits are
//     covered by the other block's wave on the same SIMD;
//   * blocks that share an XCD (an L2) get tile pairs from a compact square of the pair
//     triangle, and the row split fills whole rounds of CU slots.
// Measured at n = 1e6, p = 4096 (profiles/r02_*): 231-238 ms depending on the GPU of the pool
// for n p (p + 1) flop = 71.4-72.5 TFLOP/s = 0.90-0.92 of the 78.6 TFLOP/s FP64 matrix peak;
// MfmaUtil 94-96 %, ~8390 of the ideal 8192 matrix-pipe cycles per chunk, the rest is clock
// (2.25-2.33 GHz under this load, not 2.4).
//
// k_materialize_rows: B[row][term] row-major, p_pad columns.  Lane = row for the
//   Hadamard products (same LDS tile and register-resident term tables as k_mm), a
//   per-wave 32-term LDS transpose, then 256-byte row segments to HBM.  (Terms of more than 8
//   factors; the default copy is k_materialize_tl in kernels_prod.hip.)
// k_atb_dma2: tile pair (I, J) of 128 x 128 terms and a run of row tiles per task, tasks in
//   an XCD-aware table (build_task_order); 4 waves with 64 x 64 wave tiles (64 accumulators),
//   operands from two LDS panel buffers [16 rows][272], row-split partials reduced by
//   k_gram_reduce (kernels_gram.hip).  The same body serves C = A^T Bm for two operands
//   (launch_atb: row norms for predr_std, stored products for the marginal adjustment).
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>

#include <memory>
#include "obhip_internal.h"
#include "device_common.h"

namespace obhip {

namespace {

constexpr int kGT = 128;            // output tile edge (terms)
constexpr int kCR = 16;             // rows per chunk
constexpr int kTP = 2 * 128 + 16;   // LDS panel row pitch (doubles), see k_gram_mfma4
constexpr int kTB = 65;             // transpose buffer pitch

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
// one wave instruction: 1 KB from gbase + 16 lane -> LDS at lds_addr + 16 lane
__device__ __forceinline__ void lds_dma_1k(const char *gbase /* uniform */, uint32_t voff,
                                           uint32_t lds_addr /* uniform */) {
// m0 is written here: on the clobber list so that the compiler never assumes a value of its own
// survives the statement (round-4 advice; m0 is a reserved register, hence the diagnostic)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff), "s"(gbase), "s"(lds_addr) : "memory", "m0");
#pragma clang diagnostic pop
}

template <int OFF>
__device__ __forceinline__ double lds_rd(uint32_t addr) {
  double v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// 4 waves per block with 64 x 64 wave tiles and two blocks resident per CU: while the waves
// of one block sit at their barrier (or wait for their LDS-direct loads), the other block's
// wave on the same SIMD owns the matrix pipe.  Two panel buffers per block (2 x 34 KB x 2
// blocks = 139 KB of LDS).  lgkmcnt holds 4 bits, so a K step's 20 operand reads go out in
// two halves (12 + 8) with at most 15 newer reads behind any wait.  The "+v" operands tie
// each s_waitcnt to the registers it releases, so the compiler cannot move an MFMA that
// reads them above it.
__device__ __forceinline__ void lds_wait12(double (&a)[4], double (&b)[4][4]) {
  // a[0..3], b[0..1][0..3] usable; 8 newer reads may stay in flight
  asm volatile("s_waitcnt lgkmcnt(8)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0][0]), "+v"(b[0][1]),
                 "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]),
                 "+v"(b[1][3]));
}
template <int KEEP>
__device__ __forceinline__ void lds_wait8(double (&b)[4][4]) {
  // b[2..3][0..3] usable; KEEP newer reads may stay in flight
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(b[2][0]), "+v"(b[2][1]), "+v"(b[2][2]), "+v"(b[2][3]), "+v"(b[3][0]),
                 "+v"(b[3][1]), "+v"(b[3][2]), "+v"(b[3][3])
               : "n"(KEEP));
}
template <int OFF>
__device__ __forceinline__ void lds_rd_h0(uint32_t aaddr, const uint32_t (&baddr)[4],
                                          double (&a)[4], double (&b)[4][4]) {
  a[0] = lds_rd<OFF>(aaddr);
  a[1] = lds_rd<OFF + 128>(aaddr);
  a[2] = lds_rd<OFF + 256>(aaddr);
  a[3] = lds_rd<OFF + 384>(aaddr);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    b[0][r] = lds_rd<OFF>(baddr[r]);
    b[1][r] = lds_rd<OFF + 128>(baddr[r]);
  }
}
template <int OFF>
__device__ __forceinline__ void lds_rd_h1(const uint32_t (&baddr)[4], double (&b)[4][4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    b[2][r] = lds_rd<OFF + 256>(baddr[r]);
    b[3][r] = lds_rd<OFF + 384>(baddr[r]);
  }
}

// One kernel body, three uses (MODE), all C = A^T Bm with the contraction index k the SLOW
// index of both operands (rows of k, 128 contiguous output indices per 1-KB panel row):
//   kAtbGram   A = Bm = the staged design matrix: tile pair (I <= J) of B^T B over the rows
//              of one row split; the partial tile goes to out[(split, pair)]
//   kAtbNorm   out[J][i] = sum over the 128 columns of tile J of C[i][c]^2 (C never stored):
//              || L^-1 b_i ||^2 of predr_std with A = B^T (term-major), Bm = L^-T
//   kAtbStore  C[I-tile rows][J-tile columns] stored row-major with leading dimension ldo
// For the last two the k range of column tile J ends at (J + 1) * 128 when `tri` says Bm is
// upper triangular.  DBG: one block reports its s_memtime / s_memrealtime span (clock and
// matrix-pipe cycles per chunk under load, OBHIP_GRAM_DBG=1); production carries none of it.
constexpr int kAtbGram = 0, kAtbNorm = 1, kAtbStore = 2;
__host__ __device__ inline uint64_t atb_task(uint64_t I, uint64_t J, uint64_t y, uint64_t type = 0) {
  return I | (J << 24) | (y << 48) | (type << 62);
}
constexpr uint64_t kAtbNoTask = ~0ull;
// What the four waves of a block do with the 256 staged columns [panel I | panel J] of a chunk:
// per wave one byte = A operand's 64-column group (2 bits) | B operand's (2 bits) << 2 |
// quadrant row << 4 | quadrant column << 5 | output tile (0: (I, I), 1: (J, J), 2: (I, J)) << 6.
//   type 0: the four quadrants of tile pair (I, J).
//   types 1-3 (Gram only): a diagonal tile (I, I) needs three 64 x 64 wave tiles, not four (its
//   lower-left quadrant is the mirror of the upper-right one), so FOUR consecutive diagonal tiles
//   i .. i + 3 are twelve wave tiles = three blocks, each staging two neighbouring panels:
//   (i, i + 1) takes i's three and (lo, lo) of i + 1; (i + 1, i + 2) the other two of i + 1 and
//   two of i + 2; (i + 2, i + 3) the last of i + 2 and i + 3's three.  One block in four of the
//   diagonal's saved: 1.5 % of all blocks at p = 4096.
__device__ __forceinline__ uint32_t atb_wave_code(uint32_t type, int wave) {
  constexpr uint32_t W = 0x80u;  // output tile (I, J)
  const uint32_t t0 = (W | 0x08u) | (W | 0x2cu) << 8 | (W | 0x19u) << 16 | (W | 0x3du) << 24;
  const uint32_t t1 = 0x00u | 0x24u << 8 | 0x35u << 16 | 0x4au << 24;
  const uint32_t t2 = 0x24u | 0x35u << 8 | 0x4au << 16 | 0x6eu << 24;
  const uint32_t t3 = 0x35u | 0x4au << 8 | 0x6eu << 16 | 0x7fu << 24;
  const uint32_t tab = type == 0 ? t0 : (type == 1 ? t1 : (type == 2 ? t2 : t3));
  return (tab >> (8 * wave)) & 0xffu;
}

template <int MODE, bool DBG>
__global__ void __launch_bounds__(256, 2)
k_atb_dma2(const double *__restrict__ A, uint64_t ldA, const double *__restrict__ Bm, uint64_t ldB,
           int nb, int npairs, uint64_t ntiles, uint64_t split_base /* Gram: tiles per row split ... */,
           int tri /* Gram: ... the first `tri` splits take one more tile */,
           const uint64_t *__restrict__ tasks, double *__restrict__ part, uint64_t ldo,
           unsigned long long *dbgout) {
  extern __shared__ double T[];  // [2][16][272]
  constexpr int tszb = kCR * kTP * 8;  // bytes per buffer
  unsigned long long st0 = 0, sr0 = 0;
  if constexpr (DBG) {
    st0 = __builtin_amdgcn_s_memtime();
    sr0 = __builtin_amdgcn_s_memrealtime();
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD-aware order: tasks[] hands every XCD (blocks x, x + 8, ... share one, and with it an
  // L2) whole (square of tile pairs, row split) units, see build_task_order
  const uint64_t task = tasks[blockIdx.x];
  if (task == kAtbNoTask) return;  // padding of the shorter per-XCD sequences
  const int I = (int)(task & 0xffffff), J = (int)((task >> 24) & 0xffffff), ysplit = (int)((task >> 48) & 0x3fff);
  const uint32_t wcode = atb_wave_code(MODE == kAtbGram ? (uint32_t)(task >> 62) : 0u, wave);

  uint64_t t0, t1;  // range of 64-row tiles of k
  if constexpr (MODE == kAtbGram) {
    // the row tiles in nsplit runs that differ by at most one tile (every block of a round
    // the same length: 1954 tiles in 32 splits are 2 x 62 + 30 x 61, not 31 x 62 + 32)
    const uint64_t y = (uint64_t)ysplit, rem = (uint64_t)tri;
    t0 = y * split_base + min(y, rem);
    t1 = t0 + split_base + (y < rem ? 1 : 0);
  } else {
    t0 = 0;
    t1 = tri ? min(ntiles, (uint64_t)(J + 1) * (kGT / kTileRows)) : ntiles;
  }
  const int nchunks = (int)(t1 > t0 ? (t1 - t0) * (kTileRows / kCR) : 0);

  // wave w moves rows w, w + 4, w + 8, w + 12 of both panels of a chunk
  const uint32_t ldsT = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double *)T;
  const uint32_t lds0 = ldsT + wave * kTP * 8;
  const uint32_t voff = lane * 16;
  const char *gA = (const char *)(A + (t0 * kTileRows + wave) * ldA + (uint64_t)I * kGT);
  const char *gB = (const char *)(Bm + (t0 * kTileRows + wave) * ldB + (uint64_t)J * kGT);
  const uint64_t pitch4 = 4 * ldA * sizeof(double), pitch16 = 4 * pitch4;
  const uint64_t pitch4b = MODE == kAtbGram ? pitch4 : 4 * ldB * sizeof(double), pitch16b = 4 * pitch4b;
  // The 8 blocks of a unit that share a panel (same I or same J) walk the row chunks of the
  // split with starts rotated by (I + J) mod 8 chunks: requests for one line that reach the L2
  // together are NOT merged (each goes out to the fabric), so sharers in lockstep miss
  // together; one chunk (~3.6 us) apart, the first brings the line in and seven hit.  The
  // window of 8 chunks x 16 panels x 16 KB = 2 MB stays inside the 4-MB L2.
  const int rot = nchunks > 0 ? ((I + J) & 7) % nchunks : 0;
  auto issue = [&](int ch, int buf) {
    const uint32_t l = lds0 + buf * tszb;
    int cr = ch + rot;
    cr = cr >= nchunks ? cr - nchunks : cr;
    const char *a = gA + (uint64_t)cr * pitch16, *b = gB + (uint64_t)cr * pitch16b;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      lds_dma_1k(a + q * pitch4, voff, l + q * 4 * kTP * 8);
      lds_dma_1k(b + q * pitch4b, voff, l + q * 4 * kTP * 8 + kGT * 8);
    }
  };

  const int wm = (int)((wcode >> 4) & 1), wn = (int)((wcode >> 5) & 1);  // quadrant of the output tile
  const int acol = (int)(wcode & 3) * 64, bcol = (int)((wcode >> 2) & 3) * 64;  // operand columns in the staged row
  const int mk = lane >> 4, mblk = (lane >> 2) & 3, me = lane & 3;
  double acc[4][4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
  const uint32_t aaddr = ldsT + (mk * kTP + acol + mblk * 4 + me) * 8;
  uint32_t baddr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    baddr[r] = ldsT + (mk * kTP + bcol + ((mblk + r) & 3) * 4 + me) * 8;
  auto mfma_half = [&](const double (&a)[4], const double (&b)[4][4], int j0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = j0; j < j0 + 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j][r], acc[i][j][r], 0, 0, 0);
  };

  if (nchunks > 0) issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  unsigned long long sr1 = 0, sr2 = 0;
  if constexpr (DBG) sr1 = __builtin_amdgcn_s_memrealtime();
  double a0[4], b0[4][4], a1[4], b1[4][4];
  constexpr int stp = 4 * kTP * 8;  // bytes per K step (4 rows)

  // one K step on the operand set `cur`; the next step's reads go out in two halves
#define OB_STEP(cur_a, cur_b, nxt_a, nxt_b, NEXT_OFF, HAS_NEXT)                              \
  lds_wait12(cur_a, cur_b);                                                               \
  if (HAS_NEXT) lds_rd_h0<NEXT_OFF>(aaddr, baddr, nxt_a, nxt_b);                             \
  mfma_half(cur_a, cur_b, 0);                                                                \
  lds_wait8<(HAS_NEXT) ? 12 : 0>(cur_b);                                                     \
  if (HAS_NEXT) lds_rd_h1<NEXT_OFF>(baddr, nxt_b);                                           \
  mfma_half(cur_a, cur_b, 2);

#define OB_CHUNK_BODY2(BUF)                                                                  \
  {                                                                                          \
    constexpr int ob = (BUF) * tszb;                                                         \
    if (c + 1 < nchunks) issue(c + 1, (BUF) ^ 1); /* free since the barrier */               \
    lds_rd_h0<ob>(aaddr, baddr, a0, b0);                                                     \
    lds_rd_h1<ob>(baddr, b0);                                                                \
    OB_STEP(a0, b0, a1, b1, ob + stp, true)                                                  \
    OB_STEP(a1, b1, a0, b0, ob + 2 * stp, true)                                              \
    OB_STEP(a0, b0, a1, b1, ob + 3 * stp, true)                                              \
    OB_STEP(a1, b1, a0, b0, 0, false)                                                        \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* own part of chunk c + 1 landed */    \
    __builtin_amdgcn_s_barrier();                                                            \
    ++c;                                                                                     \
  }

  for (int c = 0; c < nchunks;) {
    OB_CHUNK_BODY2(0)
    if (c >= nchunks) break;
    OB_CHUNK_BODY2(1)
  }
#undef OB_CHUNK_BODY2
#undef OB_STEP
  if constexpr (DBG) sr2 = __builtin_amdgcn_s_memrealtime();

  if constexpr (MODE == kAtbGram) {
    // output tile of this wave: (I, J), or the diagonal tile of I or of J
    const int sel = (int)(wcode >> 6);
    const int oi = sel == 1 ? J : I, oj = sel == 0 ? I : J;
    const int slot = oi * nb - oi * (oi - 1) / 2 + (oj - oi);  // row-major index in the upper triangle
    double *out = part + ((uint64_t)ysplit * npairs + slot) * (kGT * kGT);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wm * 64 + i * 16 + mblk * 4 + mk;
          const int col = wn * 64 + j * 16 + ((mblk + r) & 3) * 4 + me;
          out[row * kGT + col] = acc[i][j][r];
        }
  } else if constexpr (MODE == kAtbStore) {
    double *out = part + ((uint64_t)I * kGT) * ldo + (uint64_t)J * kGT;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wm * 64 + i * 16 + mblk * 4 + mk;
          const int col = wn * 64 + j * 16 + ((mblk + r) & 3) * 4 + me;
          out[(uint64_t)row * ldo + col] = acc[i][j][r];
        }
  } else {
    // per output row the sum of squares over this wave's 64 columns: 16 values per lane and
    // row group i, then the 4 lanes (me) that hold the other columns of the row; the two
    // waves (wn) that share the rows meet in LDS (the panel buffers are free: every wave is
    // past the loop's last barrier)
    double *red = T;  // [2][128]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) s = fma(acc[i][j][r], acc[i][j][r], s);
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      if (me == 0) red[wn * kGT + wm * 64 + i * 16 + mblk * 4 + mk] = s;
    }
    __syncthreads();
    if (tid < kGT) part[(uint64_t)J * ldo + (uint64_t)I * kGT + tid] = red[tid] + red[kGT + tid];
  }
  if constexpr (DBG) {
    // one block of the first round of 512, one of the middle, one of the last
    const unsigned which = blockIdx.x == 7 ? 1u : blockIdx.x == gridDim.x / 2 + 7 ? 0u
                           : blockIdx.x + 512 - 7 == gridDim.x - (gridDim.x % 512 ? gridDim.x % 512 : 0) ? 2u : 3u;
    if (dbgout && which < 3 && tid == 0) {
      dbgout[3 * which] = __builtin_amdgcn_s_memtime() - st0;
      dbgout[3 * which + 1] = __builtin_amdgcn_s_memrealtime() - sr0;
      dbgout[3 * which + 2] = nchunks;
    }
    // a block of the middle round and its successor one round on: start, first chunk landed,
    // loop done, stores done (100 MHz)
    const bool mid0 = blockIdx.x == gridDim.x / 2 + 7, mid1 = blockIdx.x == gridDim.x / 2 + 7 + 512;
    if (dbgout && (mid0 || mid1) && tid == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned long long *o = dbgout + 9 + (mid1 ? 4 : 0);
      o[0] = sr0;
      o[1] = sr1;
      o[2] = sr2;
      o[3] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

// Workgroups go to the 8 XCDs round-robin by linear id, so blocks x, x + 8, x + 16, ... share
// one 4-MB L2, and an XCD keeps 64 of them resident (32 CUs x 2).  The work is cut into units
// of (8 x 8 square of the tile-pair triangle, row split): the 64 blocks of a unit read 16
// panel column blocks over the same rows, so a panel row fetched by one block is an L2 hit
// for the others.  Every XCD gets whole units, one after the other (diagonal squares hold
// 36 pairs; the next unit fills the slots they leave), and all XCDs the same number of blocks
// (padding tasks exit at once).  At p = 4096: 10 squares x 32 splits = 320 units, 2112 blocks
// per XCD = 33 rounds of 64.  (The first version dealt every XCD a 66-pair run of the sorted
// pair list per split: runs straddle squares and touch 16-24 column blocks.)
//
// When the units do not pack into the 64 slots of an XCD -- one 34-block square per split at
// p = 1024: two units are 68 blocks, and the four that do not fit wait for a whole second round --
// the units are dealt CONTINUOUSLY instead (`continuous`): the task list in unit order is cut
// into 8 equal runs, a unit may straddle two XCDs (its panels are then fetched by both).
// gram_xcd_blocks tells the row-split choice what either dealing gives an XCD.
constexpr int kXcd = 8, kSq = 8;
struct GramUnit {
  int bi, bj, y, size;
};
void gram_units(int nb, int nsplit, bool diag4, std::vector<GramUnit> &units) {
  const int nsq = (nb + kSq - 1) / kSq;
  for (int y = 0; y < nsplit; ++y)
    for (int bi = 0; bi < nsq; ++bi)
      for (int bj = bi; bj < nsq; ++bj) {
        int size = 0;
        for (int i = bi * kSq; i < std::min(nb, (bi + 1) * kSq); ++i)
          for (int j = std::max(i, bj * kSq); j < std::min(nb, (bj + 1) * kSq); ++j)
            if (!(i == j && diag4 && i / 4 * 4 + 3 < nb && i % 4 == 3)) ++size;
        if (size) units.push_back({bi, bj, y, size});
      }
}
// blocks of the busiest XCD
uint64_t gram_xcd_blocks(int nb, int nsplit, bool diag4, bool continuous) {
  std::vector<GramUnit> units;
  gram_units(nb, 1, diag4, units);  // (every split has the same units)
  uint64_t per_split = 0;
  for (const GramUnit &u : units) per_split += (uint64_t)u.size;
  if (continuous) return (per_split * (uint64_t)nsplit + kXcd - 1) / kXcd;
  std::vector<int> sizes;
  for (int y = 0; y < nsplit; ++y)
    for (const GramUnit &u : units) sizes.push_back(u.size);
  std::stable_sort(sizes.begin(), sizes.end(), [](int a, int b) { return a > b; });
  uint64_t load[kXcd] = {0};
  for (int s : sizes) {
    int k = 0;
    for (int q = 1; q < kXcd; ++q)
      if (load[q] < load[k]) k = q;
    load[k] += (uint64_t)s;
  }
  return *std::max_element(load, load + kXcd);
}
void build_task_order(int nb, int nsplit, bool diag4, bool continuous, std::vector<uint64_t> &tab) {
  typedef GramUnit Unit;
  std::vector<Unit> units;
  gram_units(nb, nsplit, diag4, units);
  // largest units first, each to the XCD with the fewest blocks so far (stable: the order
  // of equal-sized units keeps squares of one split together)
  if (!continuous)
    std::stable_sort(units.begin(), units.end(), [](const Unit &a, const Unit &b) { return a.size > b.size; });
  uint64_t total = 0;
  for (const Unit &u : units) total += (uint64_t)u.size;
  const uint64_t run = (total + kXcd - 1) / kXcd;  // continuous: tasks per XCD
  uint64_t dealt = 0;
  std::vector<std::vector<uint64_t>> seq(kXcd);
  for (const Unit &u : units) {
    int k = 0;
    for (int q = 1; q < kXcd; ++q)
      if (seq[q].size() < seq[k].size()) k = q;
    if (continuous) {  // task by task: the unit's tasks go to XCD dealt / run
      for (int i = u.bi * kSq; i < std::min(nb, (u.bi + 1) * kSq); ++i)
        for (int j = std::max(i, u.bj * kSq); j < std::min(nb, (u.bj + 1) * kSq); ++j) {
          if (i == j && diag4) {
            const int g0 = i / 4 * 4;
            if (g0 + 3 < nb) {
              if (i - g0 < 3) seq[(dealt++) / run].push_back(atb_task(i, i + 1, u.y, 1 + (i - g0)));
              continue;
            }
          }
          seq[(dealt++) / run].push_back(atb_task(i, j, u.y));
        }
      continue;
    }
    for (int i = u.bi * kSq; i < std::min(nb, (u.bi + 1) * kSq); ++i)
      for (int j = std::max(i, u.bj * kSq); j < std::min(nb, (u.bj + 1) * kSq); ++j) {
        if (i == j && diag4) {
          // four consecutive diagonal tiles as three blocks (atb_wave_code); squares are 8 wide,
          // so a group never straddles two of them
          const int g0 = i / 4 * 4;
          if (g0 + 3 < nb) {
            if (i - g0 < 3) seq[k].push_back(atb_task(i, i + 1, u.y, 1 + (i - g0)));
            continue;
          }
        }
        seq[k].push_back(atb_task(i, j, u.y));
      }
  }
  size_t len = 0;
  for (auto &q : seq) len = std::max(len, q.size());
  tab.assign(len * kXcd, kAtbNoTask);
  for (int k = 0; k < kXcd; ++k)
    for (size_t m = 0; m < seq[k].size(); ++m) tab[m * kXcd + k] = seq[k][m];
}

template <int W2>
int run_materialize(const obhip_basis &b, obhip_terms &t, double *d_B) {
  const size_t lds = (t.Mu * kTileRows + 4 * 32 * kTB) * sizeof(double);
  OB_TRY(ensure_dyn_lds((const void *)k_materialize_rows<W2>, lds));
  hipLaunchKernelGGL(k_materialize_rows<W2>, dim3((unsigned)(b.n_pad / kTileRows)), dim3(256), lds,
                     cur_stream(), b.bm.p, b.scale.p, t.ucol.p, (int)t.Mu, b.md.Mc,
                     (const uint32_t *)t.cols.p, (int)(t.W / 2), t.p_pad, d_B);
  OB_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// The staged design matrix needs n_pad * p_pad doubles.  It is kept whole (and with the basis,
// for the next fit on the same terms) when it takes at most half of the free HBM; otherwise
// the Gram is accumulated over row chunks staged one after the other.
bool gram_panel_supports(const obhip_basis &b, const obhip_terms &t) {
  const size_t need = (size_t)b.n_pad * t.p_pad * sizeof(double);
  if (b.bmat.n * sizeof(double) >= need) return true;
  if (getenv("OBHIP_GRAM_CHUNK_ROWS")) return false;  // tests: force the chunked path
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
  return need <= free_b / 2;
}

namespace {

// rows [tile0 * 64, (tile0 + ntiles) * 64) of a basis as a basis of its own for the product
// kernels (they use bm, scale, Mc, n, n_pad only); owns nothing
struct RowView {
  obhip_basis v;
  RowView(const obhip_basis &b, uint64_t tile0, uint64_t ntiles) {
    v.model = b.model;
    v.d = b.d;
    v.device = b.device;
    v.md.Mc = b.md.Mc;
    v.n_pad = ntiles * kTileRows;
    const uint64_t r0 = tile0 * kTileRows;
    v.n = r0 >= b.n ? 0 : std::min<uint64_t>(v.n_pad, b.n - r0);
    v.bm.p = b.bm.p + tile0 * b.md.Mc * kTileRows;
    v.scale.p = b.scale.p + r0;
  }
  ~RowView() {
    v.bm.p = nullptr;  // borrowed
    v.scale.p = nullptr;
  }
};

int materialize_any(const obhip_basis &b, obhip_terms &t, double *d_B, GramFuse *fuse = nullptr) {
  if (t.Mu > 280 || getenv("OBHIP_FORCE_GENERIC")) return launch_materialize_generic(b, t, d_B);
  static const bool lane_row = getenv("OBHIP_MATERIALIZE_LANE_ROW") != nullptr;
  if (!lane_row && materialize_tl_supports(t)) {
    // the fit's B^T y rides along (whole basis only: the chunked path passes no request)
    const bool f = fuse && fuse->y && fuse->g && !fuse->done;
    OB_TRY(launch_materialize_tl(b, t, d_B, f ? fuse->y : nullptr, f ? fuse->g : nullptr));
    if (f) fuse->done = true;
    return 0;
  }
  switch (t.W / 2) {
    case 1: return run_materialize<1>(b, t, d_B);
    case 2: return run_materialize<2>(b, t, d_B);
    case 3: return run_materialize<3>(b, t, d_B);
    case 4: return run_materialize<4>(b, t, d_B);
    default: return run_materialize<0>(b, t, d_B);
  }
}

// partial tiles of B^T B over the ntiles row tiles of d_B, reduced into d_G
int gram_of_staged(obhip_basis &b, const double *d_B, uint64_t ntiles, obhip_terms &t,
                   const GramSink &sink, bool accumulate, bool last) {
  // OBHIP_GRAM_DBG=1: print one block's s_memtime / s_memrealtime span
  const bool dbg = getenv("OBHIP_GRAM_DBG") && atoi(getenv("OBHIP_GRAM_DBG")) != 0;
  const int nb = (int)((t.p + kGT - 1) / kGT);
  const int npairs = nb * (nb + 1) / 2;
  // Row split: two blocks per CU at a time and all blocks (nearly) equally long, so the launch
  // takes ceil(blocks / slots) rounds of tiles-per-split each; pick the split that minimises
  // that product (528 pairs x 32 splits = 33 x 512 exactly on MI355X).
  const int ncu = device_cus(b.device);
  const bool diag4 = !(getenv("OBHIP_GRAM_DIAG4") && atoi(getenv("OBHIP_GRAM_DIAG4")) == 0);
  // blocks per row split: four consecutive diagonal tiles take three blocks (atb_wave_code)
  const uint64_t bps = (uint64_t)npairs - (diag4 ? (uint64_t)(nb / 4) : 0);
  const uint64_t slots = 2 * (uint64_t)ncu;
  uint64_t nsplit = 1;
  const uint64_t max_split = std::max<uint64_t>(
      1, std::min<uint64_t>({64, ntiles / 8, (4ull << 30) / ((uint64_t)npairs * kGT * kGT * 8)}));
  // (the splits differ by at most one tile and the longer ones are few and dealt out first,
  // so a round costs the average length; one tile charged for a block's first chunk, its
  // partial-tile store and the hand-over of the slot: ~7 us measured, OBHIP_GRAM_DBG)
  // A last round that fills at most half of the slots leaves its blocks alone on their CUs,
  // where they run ~1.7x as fast (0.6 of a round); and every split costs k_gram_reduce one
  // more partial of every tile pair to read (128 KB at ~4 TB/s = 0.0022 of the 14.6 us a block
  // takes per row tile).  Both measured at p = 4096 (16 vs 32 splits at 125 000 rows: equal
  // Gram times, half the reduction).
  // The rounds are those of the busiest XCD (64 of the slots each): the task table hands the XCDs
  // whole units (build_task_order), and when those do not pack into 64 slots -- p = 1024: one
  // 34-block unit per split, two of them = 68 -- a whole round is spent on the overhang (measured:
  // 2.56 ms where the model without XCDs promised 1.5).  Both dealings are priced, the continuous
  // one 3 % dearer for the panels its straddling units fetch twice.
  double bestc = 1e300;
  bool continuous = false;
  const uint64_t xslots = slots / kXcd;
  // (the choice depends on the shape only: priced once per shape, not once per fit)
  static std::mutex split_mu;
  static std::map<std::tuple<int, uint64_t, uint64_t, uint64_t, bool>, std::pair<uint64_t, bool>> split_cache;
  const auto split_key = std::make_tuple(nb, ntiles, max_split, slots, diag4);
  bool cached = false;
  {
    std::lock_guard<std::mutex> lk(split_mu);
    auto it = split_cache.find(split_key);
    if (it != split_cache.end()) {
      nsplit = it->second.first;
      continuous = it->second.second;
      cached = true;
    }
  }
  for (uint64_t ns = 1; !cached && ns <= max_split; ++ns) {
    const uint64_t blocks = ns * bps;
    for (int mode = 0; mode < 2; ++mode) {
      const uint64_t per = gram_xcd_blocks(nb, (int)ns, diag4, mode == 1);
      const uint64_t full = per / xslots, tail = per % xslots;
      const double rounds = (double)full + (tail == 0 ? 0.0 : (2 * tail <= xslots ? 0.6 : 1.0));
      double cost = rounds * ((double)ntiles / (double)ns + 1.0) + 0.0022 * (double)blocks;
      if (mode == 1) cost *= 1.03;
      if (cost < bestc - 1e-9) {
        bestc = cost;
        nsplit = ns;
        continuous = mode == 1;
      }
    }
  }
  if (!cached) {
    std::lock_guard<std::mutex> lk(split_mu);
    split_cache[split_key] = std::make_pair(nsplit, continuous);
  }
  if (const char *e = getenv("OBHIP_GRAM_NSPLIT")) nsplit = std::max<uint64_t>(1, std::min<uint64_t>(max_split, atoi(e)));
  if (const char *e = getenv("OBHIP_GRAM_CONTINUOUS")) continuous = atoi(e) != 0;
  double *part = nullptr;
  OB_TRY(b.workspace((size_t)nsplit * npairs * kGT * kGT * sizeof(double) + 256, (void **)&part));
  unsigned long long *dbgout =
      dbg ? (unsigned long long *)(part + (size_t)nsplit * npairs * kGT * kGT) : nullptr;
  if (b.gram_pairs_nb != nb || b.gram_pairs_ns != (int)nsplit || b.gram_pairs_diag4 != diag4 ||
      b.gram_pairs_cont != continuous) {
    std::vector<uint64_t> tab;
    build_task_order(nb, (int)nsplit, diag4, continuous, tab);
    b.gram_pairs_diag4 = diag4;
    b.gram_pairs_cont = continuous;
    OB_TRY(b.gram_pairs.upload(tab.data(), tab.size()));
    b.gram_pairs_nb = nb;
    b.gram_pairs_ns = (int)nsplit;
  }
  const unsigned nblocks = (unsigned)b.gram_pairs.n;
  const size_t lds = (size_t)2 * kCR * kTP * sizeof(double);
  {
    ProfScope ps("gram");
    if (dbg) {
      OB_TRY(ensure_dyn_lds((const void *)k_atb_dma2<kAtbGram, true>, lds));
      hipLaunchKernelGGL((k_atb_dma2<kAtbGram, true>), dim3(nblocks), dim3(256), lds, cur_stream(), d_B,
                         t.p_pad, d_B, t.p_pad, nb, npairs, ntiles, ntiles / nsplit, (int)(ntiles % nsplit), b.gram_pairs.p, part,
                         (uint64_t)0, dbgout);
    } else {
      OB_TRY(ensure_dyn_lds((const void *)k_atb_dma2<kAtbGram, false>, lds));
      hipLaunchKernelGGL((k_atb_dma2<kAtbGram, false>), dim3(nblocks), dim3(256), lds, cur_stream(),
                         d_B, t.p_pad, d_B, t.p_pad, nb, npairs, ntiles, ntiles / nsplit, (int)(ntiles % nsplit), b.gram_pairs.p, part,
                         (uint64_t)0, nullptr);
    }
    OB_HIP(hipGetLastError());
  }
  if (dbgout) {
    unsigned long long h[17];
    OB_HIP(hipMemcpy(h, dbgout, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[gram dbg] middle-round block: first chunk %.2f us, loop %.2f us, stores %.2f us; the block one round "
                    "on starts %.2f us after this one ended (its first chunk %.2f us)\n",
            0.01 * (double)(h[10] - h[9]), 0.01 * (double)(h[11] - h[10]), 0.01 * (double)(h[12] - h[11]),
            0.01 * (double)((long long)h[13] - (long long)h[12]), 0.01 * (double)(h[14] - h[13]));
    fprintf(stderr, "[gram dbg] clock of a block of the first / middle / last round: %.0f / %.0f / %.0f MHz\n",
            h[4] ? 100.0 * h[3] / h[4] : 0.0, h[1] ? 100.0 * h[0] / h[1] : 0.0, h[7] ? 100.0 * h[6] / h[7] : 0.0);
    fprintf(stderr,
            "[gram dbg] one block: %llu shader-clock ticks in %llu x 10 ns -> %.0f MHz, %.0f ticks "
            "per 16-row chunk (8192 = matrix pipe saturated by two blocks)\n",
            h[0], h[1], h[1] ? 100.0 * h[0] / h[1] : 0.0, h[2] ? (double)h[0] / h[2] : 0.0);
  }
  return launch_gram_reduce(part, npairs, (int)nsplit, nb, (int)t.p, sink, accumulate, last);
}

}  // namespace

// d_B: n_pad x p_pad doubles, row-major (= column-major p_pad x n_pad); padding rows are 0
int launch_materialize_rows(const obhip_basis &b, obhip_terms &t, double *d_B, GramFuse *fuse) {
  OB_TRY(t.prepare(b.md.cap, b.md.dims_h));
  ProfScope ps("materialize_B");
  return materialize_any(b, t, d_B, fuse);
}

// b.bmat = row-major design matrix of (b, t); kept until the basis is rebuilt or other
// terms need the buffer
int ensure_bmat(obhip_basis &b, obhip_terms &t, GramFuse *fuse) {
  if (b.bmat_terms == t.uid && t.uid != 0) return 0;
  OB_TRY(t.prepare(b.md.cap, b.md.dims_h));
  if (!gram_panel_supports(b, t))
    return fail(OBHIP_ERR_INVALID, "not enough free HBM for the n x p design matrix");
  const size_t need = (size_t)b.n_pad * t.p_pad;
  if (b.bmat.n < need) OB_TRY(b.bmat.alloc(need));
  b.bmat_terms = 0;
  OB_TRY(launch_materialize_rows(b, t, b.bmat.p, fuse));
  b.bmat_terms = t.uid;
  return 0;
}

int launch_gram_panel(const obhip_basis &bc, obhip_terms &t, const GramSink &sink, GramFuse *fuse) {
  obhip_basis &b = const_cast<obhip_basis &>(bc);
  const uint64_t ntiles = b.n_pad / kTileRows;
  if (gram_panel_supports(b, t)) {
    OB_TRY(ensure_bmat(b, t, fuse));
    return gram_of_staged(b, b.bmat.p, ntiles, t, sink, false, true);
  }
  // Not enough memory for all rows at once: stage and contract row chunks one after the
  // other, the later ones accumulating into G.  Chunk = a quarter of the free HBM (at most
  // 16 GB, at least 64 tiles so that the launch still fills the GPU).
  size_t free_b = 0, total_b = 0;
  OB_HIP(hipMemGetInfo(&free_b, &total_b));
  const size_t row_bytes = (size_t)t.p_pad * kTileRows * sizeof(double);
  uint64_t ctiles = std::min<size_t>(free_b / 4, (size_t)16 << 30) / row_bytes;
  if (const char *e = getenv("OBHIP_GRAM_CHUNK_ROWS")) ctiles = (uint64_t)atoll(e) / kTileRows;
  ctiles = std::max<uint64_t>(ctiles, 1);
  ctiles = std::min(ctiles, ntiles);
  if ((size_t)ctiles * row_bytes > free_b)
    return fail(OBHIP_ERR_HIP, "not enough free HBM for even one row chunk of the design matrix");
  b.bmat_terms = 0;  // the buffer no longer holds the whole matrix of any terms
  if (b.bmat.n < (size_t)ctiles * kTileRows * t.p_pad) OB_TRY(b.bmat.alloc((size_t)ctiles * kTileRows * t.p_pad));
  // OBHIP_GRAM_OVERLAP=1 (A/B, round-4 verdict): chunk k + 1 staged on a second stream, into a second
  // buffer, while chunk k is multiplied.  Measured at the headline with two and four chunks
  // (DESIGN.md section 10.8): nothing to gain -- the Gram's two workgroups per CU leave no LDS for a
  // staging workgroup beside them, so the copy runs in slots the Gram gives up.  Off by default.
  static const bool overlap = getenv("OBHIP_GRAM_OVERLAP") && atoi(getenv("OBHIP_GRAM_OVERLAP")) != 0;
  if (overlap && ctiles < ntiles) {
    DevBuf<double> second;
    OB_TRY(second.alloc((size_t)ctiles * kTileRows * t.p_pad));
    hipStream_t mainst = cur_stream(), side = nullptr;
    OB_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t ev0, evM[2], evG[2];
    OB_HIP(hipEventCreateWithFlags(&ev0, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) {
      OB_HIP(hipEventCreateWithFlags(&evM[i], hipEventDisableTiming));
      OB_HIP(hipEventCreateWithFlags(&evG[i], hipEventDisableTiming));
    }
    OB_HIP(hipEventRecord(ev0, mainst));  // the basis is built on the caller's stream
    OB_HIP(hipStreamWaitEvent(side, ev0, 0));
    int rc = 0;
    uint64_t c = 0;
    for (uint64_t t0 = 0; t0 < ntiles && rc == 0; t0 += ctiles, ++c) {
      const uint64_t nt = std::min(ctiles, ntiles - t0);
      const int buf = (int)(c & 1);
      double *Bp = buf ? second.p : b.bmat.p;
      RowView view(b, t0, nt);
      if (c >= 2) (void)hipStreamWaitEvent(side, evG[buf], 0);  // the product that read this buffer is done
      set_cur_stream(side);
      {
        ProfScope ps("materialize_B");
        rc = materialize_any(view.v, t, Bp);
      }
      (void)hipEventRecord(evM[buf], side);
      set_cur_stream(mainst);
      if (rc) break;
      (void)hipStreamWaitEvent(mainst, evM[buf], 0);
      rc = gram_of_staged(b, Bp, nt, t, sink, t0 != 0, t0 + nt >= ntiles);
      (void)hipEventRecord(evG[buf], mainst);
    }
    (void)hipStreamSynchronize(side);
    (void)hipStreamSynchronize(mainst);  // (the second buffer goes back to the pool below)
    (void)hipEventDestroy(ev0);
    for (int i = 0; i < 2; ++i) {
      (void)hipEventDestroy(evM[i]);
      (void)hipEventDestroy(evG[i]);
    }
    (void)hipStreamDestroy(side);
    return rc;
  }
  for (uint64_t t0 = 0; t0 < ntiles; t0 += ctiles) {
    const uint64_t nt = std::min(ctiles, ntiles - t0);
    RowView view(b, t0, nt);
    {
      ProfScope ps("materialize_B");
      OB_TRY(materialize_any(view.v, t, b.bmat.p));
    }
    OB_TRY(gram_of_staged(b, b.bmat.p, nt, t, sink, t0 != 0, t0 + nt >= ntiles));
  }
  return 0;
}

// C = A^T Bm on the matrix cores for two different operands, k the slow index of both:
// A is K x M (leading dimension ldA), Bm is K x N (ldB); K a multiple of 64 and M, N multiples
// of 128 (callers pad with zeros).  tri: Bm[k][c] = 0 for k > c (column tile J needs
// k < (J + 1) * 128 only).  mode kAtbNorm: out[J * ldo + i] = sum_{c in tile J} C[i][c]^2;
// mode kAtbStore: out[i * ldo + c] = C[i][c].  Tasks go to the XCDs in 8 x 8 squares of
// (row tile, column tile) like the Gram's.
int launch_atb(int mode, const double *A, uint64_t ldA, uint64_t M, const double *Bm, uint64_t ldB,
               uint64_t N, uint64_t K, bool tri, double *out, uint64_t ldo) {
  if (M % kGT || N % kGT || K % kTileRows)
    return fail(OBHIP_ERR_INVALID, "launch_atb: sizes must be padded to the tile");
  const uint64_t mt = M / kGT, nt = N / kGT;
  if (mt >= (1ull << 24) || nt >= (1ull << 24)) return fail(OBHIP_ERR_INVALID, "launch_atb: too many tiles");
  // The task table depends on (mt, nt) only and a predictor asks for the same one at every
  // var() call: built and uploaded once per device and shape and kept (no per-call upload, no
  // stream synchronisation for a local buffer) -- the kAtbCache most recently used shapes, so
  // that a long-lived process predicting on ever-changing batch sizes does not grow without
  // bound (a table is mt * nt * 8 bytes: megabytes at 1e6 rows).  Entries are shared_ptrs: a
  // caller keeps its table alive from the look-up until its launch has been enqueued, whatever
  // another thread evicts meanwhile (round-4 advice: handles of different threads may launch
  // concurrently).  The evicting thread waits for the device THE ENTRY BELONGS TO (a launch on
  // any of its streams may still be reading the table) outside the lock; that happens once per
  // kAtbCache new shapes at most.
  constexpr size_t kAtbCache = 8;
  using Tab = std::shared_ptr<DevBuf<uint64_t>>;
  static std::mutex mu;
  static std::map<std::tuple<int, uint64_t, uint64_t>, Tab> cache;
  static std::vector<std::tuple<int, uint64_t, uint64_t>> lru;  // least recently used first
  int dev = 0;
  (void)hipGetDevice(&dev);
  Tab dtabp, victim;
  int victim_dev = dev;
  {
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_tuple(dev, mt, nt);
    lru.erase(std::remove(lru.begin(), lru.end(), key), lru.end());
    lru.push_back(key);
    if (lru.size() > kAtbCache) {
      auto it = cache.find(lru.front());
      if (it != cache.end()) {
        victim = std::move(it->second);
        victim_dev = std::get<0>(lru.front());
        cache.erase(it);
      }
      lru.erase(lru.begin());
    }
    Tab &slot = cache[key];
    if (!slot) {
      // units of up to 8 x 8 tiles, column-tile squares of equal k range together, dealt to the
      // XCD with the fewest blocks so far
      std::vector<std::vector<uint64_t>> seq(kXcd);
      for (uint64_t bj = 0; bj * kSq < nt; ++bj)
        for (uint64_t bi = 0; bi * kSq < mt; ++bi) {
          int k = 0;
          for (int q = 1; q < kXcd; ++q)
            if (seq[q].size() < seq[k].size()) k = q;
          for (uint64_t i = bi * kSq; i < std::min(mt, (bi + 1) * kSq); ++i)
            for (uint64_t j = bj * kSq; j < std::min(nt, (bj + 1) * kSq); ++j) seq[k].push_back(atb_task(i, j, 0));
        }
      size_t len = 0;
      for (auto &q : seq) len = std::max(len, q.size());
      std::vector<uint64_t> tab(len * kXcd, kAtbNoTask);
      for (int k = 0; k < kXcd; ++k)
        for (size_t m = 0; m < seq[k].size(); ++m) tab[m * kXcd + k] = seq[k][m];
      slot = std::make_shared<DevBuf<uint64_t>>();  // freed by its last holder, after eviction
      const int rc = slot->upload(tab.data(), tab.size());
      if (rc) {
        cache.erase(key);
        lru.pop_back();
        return rc;
      }
    }
    dtabp = slot;
  }
  if (victim) {  // (its own device's work first: the table may be in use by a launch in flight)
    if (victim_dev != dev) (void)hipSetDevice(victim_dev);
    (void)hipDeviceSynchronize();
    victim.reset();
    if (victim_dev != dev) (void)hipSetDevice(dev);
  }
  DevBuf<uint64_t> &dtab = *dtabp;
  const size_t ntasks = dtab.n;
  const size_t lds = (size_t)2 * kCR * kTP * sizeof(double);
  const uint64_t ktiles = K / kTileRows;
  if (mode == kAtbNorm) {
    OB_TRY(ensure_dyn_lds((const void *)k_atb_dma2<kAtbNorm, false>, lds));
    hipLaunchKernelGGL((k_atb_dma2<kAtbNorm, false>), dim3((unsigned)ntasks), dim3(256), lds,
                       cur_stream(), A, ldA, Bm, ldB, 0, 0, ktiles, ktiles, tri ? 1 : 0, dtab.p, out,
                       ldo, nullptr);
  } else if (mode == kAtbStore) {
    OB_TRY(ensure_dyn_lds((const void *)k_atb_dma2<kAtbStore, false>, lds));
    hipLaunchKernelGGL((k_atb_dma2<kAtbStore, false>), dim3((unsigned)ntasks), dim3(256), lds,
                       cur_stream(), A, ldA, Bm, ldB, 0, 0, ktiles, ktiles, tri ? 1 : 0, dtab.p, out,
                       ldo, nullptr);
  } else {
    return fail(OBHIP_ERR_INVALID, "launch_atb: unknown mode");
  }
  OB_HIP(hipGetLastError());
  return 0;
}

}  // namespace obhip
