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
//   operands from a ring of 4 LDS panel buffers [row][term], pitch 272 doubles); 4
//   producers copy the chunk's two 16 x 128 panels from B, two chunks ahead in
//   registers; the hand-over uses LDS counters, not barriers.
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
// Producer/consumer hand-over goes through LDS counters instead of workgroup barriers: a
// barrier makes the 8 consumer waves drain the matrix pipe in lock-step once per chunk.
// ready[b] counts producer waves that have filled buffer b (4 per use), done[b] counts
// consumer waves that have issued their last read of it (8 per use).  LDS instructions of
// one wave execute in order, so a counter update is ordered behind that wave's earlier
// panel writes / operand reads.
constexpr int kNB = 4;  // panel buffers

__device__ __forceinline__ uint32_t flag_load(const uint32_t *p) {
  return __builtin_amdgcn_readfirstlane(
      __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void flag_wait(const uint32_t *p, uint32_t need) {
  while (flag_load(p) < need) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void flag_signal(uint32_t *p, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ void __launch_bounds__(768, 3)
k_gram_panel(const double *__restrict__ B, uint64_t p_pad, int nb, uint64_t ntiles,
             uint64_t tiles_per_split, double *__restrict__ part, int dbg) {
  extern __shared__ double T[];  // [kNB][16][272] + counters
  constexpr int tsz = kCR * kTP;
  uint32_t *ready = (uint32_t *)(T + kNB * tsz);
  uint32_t *done = ready + kNB;
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

  if (tid < 2 * kNB) ready[tid] = 0;
  __syncthreads();  // the only workgroup barrier

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
      const uint64_t off = (dbg & 2) ? 0 : (uint64_t)ch * kCR * p_pad;
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
    // registers hold chunks c (this iteration's stage) and c+1 (the other stage)
    if (nchunks > 0) fetch(0, preA);
    if (nchunks > 1) fetch(1, preB);
    int buf = 0;
    uint32_t round = 0;  // c / kNB
    auto stage = [&](int c, d2 (&pre)[kItems]) {
      if (round > 0) flag_wait(done + buf, 8u * round);
      put(buf, pre);
      flag_signal(ready + buf, lane);
      if (c + 2 < nchunks) fetch(c + 2, pre);
      if (++buf == kNB) {
        buf = 0;
        ++round;
      }
    };
    for (int c = 0; c < nchunks; c += 2) {
      stage(c, preA);
      if (c + 1 < nchunks) stage(c + 1, preB);
    }
    return;
  }

  // consumers: 64 x 32 wave tiles as in k_gram_mfma4
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

  double a0[4], b0[2][4], a1[4], b1[2][4];
  int buf = 0;
  uint32_t round = 0;
  if (nchunks > 0) {
    flag_wait(ready + 0, 4u);
    load_ops(T, 0, a0, b0);
  }
  for (int c = 0; c < nchunks; ++c) {
    const double *tp = T + buf * tsz;
    uint32_t *mydone = done + buf;
    if (++buf == kNB) {
      buf = 0;
      ++round;
    }
    load_ops(tp, 1, a1, b1);
    mfma_step(a0, b0);
    load_ops(tp, 2, a0, b0);
    mfma_step(a1, b1);
    load_ops(tp, 3, a1, b1);
    flag_signal(mydone, lane);  // last read of this buffer is in the LDS queue
    mfma_step(a0, b0);
    if (c + 1 < nchunks) {
      flag_wait(ready + buf, 4u * (round + 1));
      load_ops(T + buf * tsz, 0, a0, b0);
    }
    mfma_step(a1, b1);
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

// ---- Gram from the materialised B, panels moved by LDS-direct loads ---------------------------
// No producer waves at all: every consumer wave issues four global_load_lds_dwordx4 per chunk
// (one instruction = one 1-KB row segment of a panel, global -> LDS without registers or
// VALU work), kNB - 1 chunks ahead, and tracks them with vmcnt.  VMEM and SALU instructions
// issue beside the wave's own MFMA stream, so the copy costs no matrix-pipe time.
__device__ __forceinline__ void lds_dma_1k(const char *gbase /* uniform */, uint32_t voff,
                                           uint32_t lds_addr /* uniform */) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff), "s"(gbase), "s"(lds_addr) : "memory");
}

// Operand reads are hand-issued ds_read_b64 with immediate offsets and hand-placed
// s_waitcnt: a dynamic buffer index would cost ~20 integer VALU instructions per chunk for
// LDS addresses, and tools/mfma4x4_lds_bench.hip shows each VALU instruction inside an
// FP64-MFMA-saturated stream costs ~7-14 matrix-pipe cycles.  The loop below holds MFMA,
// LDS, VMEM and SALU instructions only.
template <int OFF>
__device__ __forceinline__ double lds_rd(uint32_t addr) {
  double v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ void lds_rd_ops(uint32_t aaddr, const uint32_t (&baddr)[4],
                                           double (&a)[4], double (&b)[2][4]) {
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
// the 12 operands become usable here; KEEP = newer LDS reads that may stay in flight
template <int KEEP>
__device__ __forceinline__ void lds_wait_ops(double (&a)[4], double (&b)[2][4]) {
  asm volatile("s_waitcnt lgkmcnt(%12)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0][0]), "+v"(b[0][1]),
                 "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]),
                 "+v"(b[1][3])
               : "n"(KEEP));
}

__global__ void __launch_bounds__(512, 2)
k_gram_dma(const double *__restrict__ B, uint64_t p_pad, int nb, uint64_t ntiles,
           uint64_t tiles_per_split, double *__restrict__ part, unsigned long long *dbgout) {
  extern __shared__ double T[];  // [kNB][16][272]
  static_assert(kNB == 4, "buffer offsets and vmcnt immediates assume 4 buffers");
  constexpr int tszb = kCR * kTP * 8;  // bytes per buffer
  const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int I = 0, rem = blockIdx.x;
  while (rem >= nb - I) {
    rem -= nb - I;
    ++I;
  }
  const int J = I + rem;

  const uint64_t t0 = (uint64_t)blockIdx.y * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);
  const int nchunks = (int)(t1 > t0 ? (t1 - t0) * (kTileRows / kCR) : 0);

  // wave w moves rows w and w + 8 of both panels of a chunk
  const uint32_t ldsT = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double *)T;
  const uint32_t lds0 = ldsT + wave * kTP * 8;
  const uint32_t voff = lane * 16;
  const char *gA = (const char *)(B + (t0 * kTileRows + wave) * p_pad + (uint64_t)I * kGT);
  const char *gB = (const char *)(B + (t0 * kTileRows + wave) * p_pad + (uint64_t)J * kGT);
  const uint64_t pitch8 = 8 * p_pad * sizeof(double), pitch16 = 2 * pitch8;
  auto issue = [&](int ch, int buf) {
    const uint32_t l = lds0 + buf * tszb;
    const uint64_t off = (uint64_t)ch * pitch16;
    lds_dma_1k(gA + off, voff, l);
    lds_dma_1k(gB + off, voff, l + kGT * 8);
    lds_dma_1k(gA + off + pitch8, voff, l + 8 * kTP * 8);
    lds_dma_1k(gB + off + pitch8, voff, l + 8 * kTP * 8 + kGT * 8);
  };

  const int wm = wave >> 2, wn = wave & 3;
  const int mk = lane >> 4, mblk = (lane >> 2) & 3, me = lane & 3;
  double acc[4][2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
  // byte addresses of this lane's operands in buffers 0 (set 0) and 2 (set 1); buffers 1
  // and 3, the K steps and the 16-term sub-blocks are immediate offsets (< 64 KB)
  uint32_t aaddr[2], baddr[2][4];
#pragma unroll
  for (int set = 0; set < 2; ++set) {
    aaddr[set] = ldsT + set * 2 * tszb + (mk * kTP + wm * 64 + mblk * 4 + me) * 8;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      baddr[set][r] =
          ldsT + set * 2 * tszb + (mk * kTP + kGT + wn * 32 + ((mblk + r) & 3) * 4 + me) * 8;
  }
  auto mfma_step = [&](const double (&a)[4], const double (&b)[2][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j][r], acc[i][j][r], 0, 0, 0);
  };

  // Invariant at the top of the body of chunk c: chunks c and c + 1 have landed and are
  // visible to every wave; DMA loads are in flight up to chunk c + 2; the reads of chunk
  // c's step-0 operands (a0, b0) are in the LDS queue.
  for (int ch = 0; ch < kNB - 1 && ch < nchunks; ++ch) issue(ch, ch);
  if (nchunks >= kNB - 1)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  double a0[4], b0[2][4], a1[4], b1[2][4];
  constexpr int stp = 4 * kTP * 8;  // bytes per K step (4 rows)
  if (nchunks > 0) lds_rd_ops<0>(aaddr[0], baddr[0], a0, b0);

#define OB_CHUNK_BODY(BUF)                                                                   \
  {                                                                                          \
    constexpr int set = (BUF) >> 1, ob = ((BUF) & 1) * tszb;                                 \
    constexpr int nset = (((BUF) + 1) & 3) >> 1, nob = (((BUF) + 1) & 1) * tszb;             \
    const bool more = c + kNB - 1 < nchunks;                                                 \
    if (more) issue(c + kNB - 1, ((BUF) + 3) & 3); /* buffer of chunk c - 1, free now */     \
    lds_rd_ops<ob + stp>(aaddr[set], baddr[set], a1, b1);                                    \
    lds_wait_ops<12>(a0, b0);                                                                \
    mfma_step(a0, b0);                                                                       \
    lds_rd_ops<ob + 2 * stp>(aaddr[set], baddr[set], a0, b0);                                \
    lds_wait_ops<12>(a1, b1);                                                                \
    mfma_step(a1, b1);                                                                       \
    lds_rd_ops<ob + 3 * stp>(aaddr[set], baddr[set], a1, b1);                                \
    lds_wait_ops<12>(a0, b0);                                                                \
    mfma_step(a0, b0);                                                                       \
    /* step 0 of the next chunk; past the last chunk this reads a stale buffer, unused */    \
    lds_rd_ops<nob>(aaddr[nset], baddr[nset], a0, b0);                                       \
    lds_wait_ops<12>(a1, b1);                                                                \
    mfma_step(a1, b1);                                                                       \
    if (more)                                                                                \
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); /* own part of chunk c + 2 landed */  \
    else                                                                                     \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
    __builtin_amdgcn_s_barrier();                                                            \
    ++c;                                                                                     \
  }

  for (int c = 0; c < nchunks;) {
    OB_CHUNK_BODY(0)
    if (c >= nchunks) break;
    OB_CHUNK_BODY(1)
    if (c >= nchunks) break;
    OB_CHUNK_BODY(2)
    if (c >= nchunks) break;
    OB_CHUNK_BODY(3)
  }
#undef OB_CHUNK_BODY
  if (nchunks > 0) lds_wait_ops<0>(a0, b0);  // the trailing (unused) reads have landed

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
  if (dbgout && blockIdx.x == 7 && blockIdx.y == 3 && tid == 0) {
    dbgout[0] = __builtin_amdgcn_s_memtime() - st0;
    dbgout[1] = __builtin_amdgcn_s_memrealtime() - sr0;
    dbgout[2] = nchunks;
  }
}

// ---- two desynchronised blocks per CU --------------------------------------------------------
// Same data flow as k_gram_dma, but 4 waves per block with 64 x 64 wave tiles (64
// accumulators) and two blocks resident per CU: while the waves of one block sit at their
// barrier (or wait for their LDS-direct loads), the other block's wave on the same SIMD owns
// the matrix pipe.  Two panel buffers per block (2 x 34 KB x 2 blocks = 139 KB of LDS).
// lgkmcnt holds 4 bits, so a K step's 20 operand reads go out in two halves (12 + 8) with at
// most 15 newer reads behind any wait.
__device__ __forceinline__ void lds_wait12(int keep8, double (&a)[4], double (&b)[4][4]) {
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

__global__ void __launch_bounds__(256, 2)
k_gram_dma2(const double *__restrict__ B, uint64_t p_pad, int nb, uint64_t ntiles,
            uint64_t tiles_per_split, const uint32_t *__restrict__ pairs,
            double *__restrict__ part, unsigned long long *dbgout) {
  extern __shared__ double T[];  // [2][16][272]
  constexpr int tszb = kCR * kTP * 8;  // bytes per buffer
  const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD-aware order: pairs[] deals compact squares of the (I, J) triangle to the blocks
  // that share an XCD (and with it an L2), see build_pair_order
  const uint32_t ij = pairs[blockIdx.x];
  const int I = ij & 0xffff, J = ij >> 16;
  const int slot = I * nb - I * (I - 1) / 2 + (J - I);  // row-major index in the upper triangle

  const uint64_t t0 = (uint64_t)blockIdx.y * tiles_per_split;
  const uint64_t t1 = min(ntiles, t0 + tiles_per_split);
  const int nchunks = (int)(t1 > t0 ? (t1 - t0) * (kTileRows / kCR) : 0);

  // wave w moves rows w, w + 4, w + 8, w + 12 of both panels of a chunk
  const uint32_t ldsT = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double *)T;
  const uint32_t lds0 = ldsT + wave * kTP * 8;
  const uint32_t voff = lane * 16;
  const char *gA = (const char *)(B + (t0 * kTileRows + wave) * p_pad + (uint64_t)I * kGT);
  const char *gB = (const char *)(B + (t0 * kTileRows + wave) * p_pad + (uint64_t)J * kGT);
  const uint64_t pitch4 = 4 * p_pad * sizeof(double), pitch16 = 4 * pitch4;
  auto issue = [&](int ch, int buf) {
    const uint32_t l = lds0 + buf * tszb;
    if (dbgout && dbgout[7]) ch = 0;  // debug: L2-resident source
    const char *a = gA + (uint64_t)ch * pitch16, *b = gB + (uint64_t)ch * pitch16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      lds_dma_1k(a + q * pitch4, voff, l + q * 4 * kTP * 8);
      lds_dma_1k(b + q * pitch4, voff, l + q * 4 * kTP * 8 + kGT * 8);
    }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int mk = lane >> 4, mblk = (lane >> 2) & 3, me = lane & 3;
  double acc[4][4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
  const uint32_t aaddr = ldsT + (mk * kTP + wm * 64 + mblk * 4 + me) * 8;
  uint32_t baddr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    baddr[r] = ldsT + (mk * kTP + kGT + wn * 64 + ((mblk + r) & 3) * 4 + me) * 8;
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
  double a0[4], b0[4][4], a1[4], b1[4][4];
  constexpr int stp = 4 * kTP * 8;  // bytes per K step (4 rows)

  // one K step on the operand set `cur`; the next step's reads go out in two halves
#define OB_STEP(cur_a, cur_b, nxt_a, nxt_b, NEXT_OFF, HAS_NEXT)                              \
  lds_wait12(0, cur_a, cur_b);                                                               \
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

  double *out = part + ((uint64_t)blockIdx.y * gridDim.x + slot) * (kGT * kGT);
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
  if (dbgout && blockIdx.x == 7 && blockIdx.y == 3 && tid == 0) {
    dbgout[0] = __builtin_amdgcn_s_memtime() - st0;
    dbgout[1] = __builtin_amdgcn_s_memrealtime() - sr0;
    dbgout[2] = nchunks;
  }
}

// Workgroups go to the 8 XCDs round-robin by linear id, so blocks x, x + 8, x + 16, ... of
// a row split share one 4-MB L2.  Give each XCD a contiguous run of the tile pairs sorted
// square-major (8 x 8 squares of the (I, J) triangle): its ~66 resident blocks then touch
// ~16-24 of the 32 column blocks instead of all of them, and panel rows fetched by one
// block are L2 hits for the others.
constexpr int kXcd = 8;
void build_pair_order(int nb, std::vector<uint32_t> &tab) {
  const int npairs = nb * (nb + 1) / 2;
  const int group = (npairs + kXcd - 1) / kXcd;
  int sq = 1;
  while ((sq + 1) * (sq + 1) <= group) ++sq;
  std::vector<uint32_t> sorted;
  sorted.reserve(npairs);
  for (int bi = 0; bi * sq < nb; ++bi)
    for (int bj = bi; bj * sq < nb; ++bj)
      for (int i = bi * sq; i < std::min(nb, (bi + 1) * sq); ++i)
        for (int j = std::max(i, bj * sq); j < std::min(nb, (bj + 1) * sq); ++j)
          sorted.push_back((uint32_t)i | ((uint32_t)j << 16));
  tab.assign(npairs, 0);
  if (getenv("OBHIP_GRAM_ORDER") && atoi(getenv("OBHIP_GRAM_ORDER")) == 0) {  // tuning aid
    int x = 0;
    for (int i = 0; i < nb; ++i)
      for (int j = i; j < nb; ++j) tab[x++] = (uint32_t)i | ((uint32_t)j << 16);
    return;
  }
  // block x = kXcd * m + k runs on XCD k: hand it element m of XCD k's run
  int next = 0;
  for (int k = 0; k < kXcd; ++k)
    for (int x = k; x < npairs; x += kXcd) tab[x] = sorted[next++];
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
  const int dbg = getenv("OBHIP_GRAM_DBG") ? atoi(getenv("OBHIP_GRAM_DBG")) : 0;
  if (dbg & 1) OB_HIP(hipMemsetAsync(b.bmat.p, 0, need * sizeof(double), cur_stream()));
  const int nb = (int)((t.p + kGT - 1) / kGT);
  const int npairs = nb * (nb + 1) / 2;
  const uint64_t ntiles = b.n_pad / kTileRows;
  // Row split: one block per CU at a time and all blocks equally long, so the launch takes
  // ceil(blocks / CUs) rounds of tiles-per-split each; pick the split that minimises that
  // product (528 pairs x 16 splits = 33 x 256 exactly on MI355X), 2 tiles per block charged
  // for its prologue and partial-tile write.
  static int ncu = 0;
  if (!ncu) {
    OB_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, b.device));
    if (ncu <= 0) ncu = 256;
  }
  const bool two_blocks = !(dbg & 8) && !(dbg & 4);  // k_gram_dma2: two resident blocks per CU
  const uint64_t slots = (uint64_t)ncu * (two_blocks ? 2 : 1);
  uint64_t nsplit = 1, best = ~0ull;
  const uint64_t max_split =
      std::max<uint64_t>(1, std::min<uint64_t>({64, ntiles / 8, (4ull << 30) / ((uint64_t)npairs * kGT * kGT * 8)}));
  for (uint64_t ns = 1; ns <= max_split; ++ns) {
    const uint64_t tp = (ntiles + ns - 1) / ns, nse = (ntiles + tp - 1) / tp;
    const uint64_t rounds = (nse * npairs + slots - 1) / slots, cost = rounds * (tp + 2);
    if (cost < best) {
      best = cost;
      nsplit = nse;
    }
  }
  if (getenv("OBHIP_GRAM_NSPLIT")) nsplit = std::max(1, atoi(getenv("OBHIP_GRAM_NSPLIT")));
  const uint64_t tps = (ntiles + nsplit - 1) / nsplit;
  double *part = nullptr;
  OB_TRY(b.workspace((size_t)nsplit * npairs * kGT * kGT * sizeof(double) + 64, (void **)&part));
  unsigned long long *dbgout =
      (dbg & 16) ? (unsigned long long *)(part + (size_t)nsplit * npairs * kGT * kGT) : nullptr;
  if (dbgout) {
    unsigned long long flag = (dbg & 2) ? 1 : 0;
    OB_HIP(hipMemcpy(dbgout + 7, &flag, 8, hipMemcpyHostToDevice));
  }
  const size_t lds = (size_t)kNB * kCR * kTP * sizeof(double) + 2 * kNB * sizeof(uint32_t);
  OB_HIP(hipFuncSetAttribute((const void *)k_gram_panel, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)lds));
  if (!(dbg & 4)) {
    ProfScope ps("gram");
    if (two_blocks) {
      const size_t ldsd = (size_t)2 * kCR * kTP * sizeof(double);
      OB_HIP(hipFuncSetAttribute((const void *)k_gram_dma2,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd));
      if (b.gram_pairs_nb != nb) {
        std::vector<uint32_t> tab;
        build_pair_order(nb, tab);
        OB_TRY(b.gram_pairs.upload(tab.data(), tab.size()));
        b.gram_pairs_nb = nb;
      }
      hipLaunchKernelGGL(k_gram_dma2, dim3((unsigned)npairs, (unsigned)nsplit), dim3(256), ldsd,
                         cur_stream(), b.bmat.p, t.p_pad, nb, ntiles, tps, b.gram_pairs.p, part,
                         dbgout);
    } else {
      const size_t ldsd = (size_t)kNB * kCR * kTP * sizeof(double);
      OB_HIP(hipFuncSetAttribute((const void *)k_gram_dma,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd));
      hipLaunchKernelGGL(k_gram_dma, dim3((unsigned)npairs, (unsigned)nsplit), dim3(512), ldsd,
                         cur_stream(), b.bmat.p, t.p_pad, nb, ntiles, tps, part, dbgout);
    }
    OB_HIP(hipGetLastError());
    if (dbgout) {
      unsigned long long h[3];
      OB_HIP(hipMemcpy(h, dbgout, sizeof(h), hipMemcpyDeviceToHost));
      fprintf(stderr, "[gram dbg] block: %llu memtime ticks, %llu realtime ticks (100 MHz) -> %.1f MHz if memtime = shader clock; %.1f ticks/chunk\n", h[0], h[1], h[1] ? 100.0 * h[0] / h[1] : 0.0, h[2] ? (double)h[0] / h[2] : 0.0);
    }
  } else {
    ProfScope ps("gram");
    hipLaunchKernelGGL(k_gram_panel, dim3((unsigned)npairs, (unsigned)nsplit), dim3(768), lds,
                       cur_stream(), b.bmat.p, t.p_pad, nb, ntiles, tps, part, dbg);
    OB_HIP(hipGetLastError());
  }
  return launch_gram_reduce(part, npairs, (int)nsplit, nb, (int)t.p, d_G);
}

}  // namespace obhip
