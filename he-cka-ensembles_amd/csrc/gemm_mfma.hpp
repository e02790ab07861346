// fp32 MFMA GEMM core for gfx950 (MI355X): 256-thread workgroups (4 waves of 64 lanes, 2x2),
// LDS-tiled operands, v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chains), register-staged
// double buffering with one barrier per K-tile.
//
// Every GEMM-shaped step of the HE-CKA path instantiates this core (hecka_kernels.hip):
//   gram   (NT, symmetric, centred)   Kt_m = (X_m - 1 sbar_m^T)(X_m - 1 sbar_m^T)^T
//   pair   (NT, split-K)              S    = Cflat Cflat^T
//   comb   (NN)                       dC   = A Cflat   (+ epilogue -> W)
//   dgram  (NN, batched)              dX_m = W_m X_m
//
// Operand conventions (all row-major fp32):
//   A      : [rowsA, K]   k contiguous, leading dimension lda
//   B (NT) : [rowsB, K]   k contiguous            -> out[i,j] = sum_k A[i,k] B[j,k]
//   B (NN) : [K, cols]    n contiguous            -> out[i,j] = sum_k A[i,k] B[k,j]
//
// MFMA operand map (32x32x2, lane l, r=l&31, h=l>>5): A holds A[r][k=h], B holds B[k=h][r].
// The sum over k is order independent, so inside each group of 8 k's lane-half h takes
// k = 4h+s at step s (s=0..3): one 16-byte LDS read feeds four MFMAs (A and B use the same map).
// Accumulator map: reg q of a 32x32 tile is row (q&3)+8*(q>>2)+4*h, column r.
//
// Schedule of the steady-state K-tile (measured on MI355X, see DESIGN.md "GEMM core"): hipcc groups the
// global loads, the LDS restage and the fragment reads into phases around the MFMA block, and two
// co-resident waves of a SIMD then run in lockstep (both in their non-MFMA phase at once): 73-80 % MFMA
// occupancy.  Here the side work is cut into pieces and one piece is placed in front of every small
// chunk of MFMAs, pinned with sched_barrier(0), so it issues in the shadow of the 64-cycle MFMAs:
//   k-group 0      : global loads of the NEXT tile (one per chunk) + fragment reads of k-group 1
//   k-group 1..    : fragment reads of the next k-group
//   k-group KB-2   : LDS writes of the next tile into the other buffer (their data was loaded in group 0)
#pragma once
#include <hip/hip_runtime.h>

namespace hecka {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBK = 32;        // default K-tile depth (split-K chunks are multiples of this)
constexpr int kThreads = 256;

struct GemmOperands {
  const float* A;
  const float* B;
  long long strideA;  // batch stride (elements)
  long long strideB;
  int lda, ldb;
  int rowsA;  // valid rows of A (output rows)
  int colsB;  // valid output columns (NT: rows of B; NN: columns of B)
  int K;      // full reduction length
  int kchunk; // reduction length handled per blockIdx.z (multiple of kBK); == K rounded up if no split
};

template <int TM, int TN, bool NN, int BK>
struct GemmSmem {
  static constexpr int LDK = BK + 4;  // padded LDS row (floats) of a k-contiguous tile: conflict-free b128 reads
  static constexpr int kA = TM * LDK;
  static constexpr int kB = NN ? (BK * TN) : (TN * LDK);
  alignas(16) float a[2][kA];
  alignas(16) float b[2][kB];
};

// Loads a float4 of a k-contiguous row; zero outside [0,rows) x [k, kend).  VEC: 16-byte path allowed.
template <bool VEC>
__device__ __forceinline__ f32x4 load_row4(const float* base, int ld, int row, int rows, int k, int kend) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (row < rows) {
    const float* p = base + (size_t)row * ld + k;
    if (VEC) {
      if (k < kend) v = *reinterpret_cast<const f32x4*>(p);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k + e < kend) v[e] = p[e];
    }
  }
  return v;
}

// One workgroup's tile of the product.  `Pro` may transform staged A/B values (centering); the caller
// consumes the accumulators through for_each() after run().
//   FULL : the caller guarantees every tile is interior (rows/cols multiples of the tile, K-range a
//          multiple of BK, 16-byte aligned rows): per-thread source pointers are formed once and bumped per
//          K-tile and the loads carry no bounds checks (no exec-mask branches in the main loop).
//   CHAIN: the k range is accumulated in local chains of kChainTiles K-tiles that start from zero and are added to the
//          running sum with v_add_f32 (blocked summation): the rounding error of a long K grows with the number of chains
//          plus the chain length instead of K (measured at K = 2294: 2.0e-6 -> 6e-7 of the largest entry).
constexpr int kChainTiles = 4;
template <int TM, int TN, bool NN, bool VEC, class Pro, int BK = kBK, bool FULL = false, bool CHAIN = false>
struct GemmTile {
  static_assert(BK % 8 == 0 && (TM * BK / 4) % kThreads == 0 && (TN * BK / 4) % kThreads == 0, "tile shape");
  static constexpr int LDK = BK + 4;
  static constexpr int C4 = BK / 4;                   // float4 per k-contiguous tile row
  static constexpr int KB = BK / 8;                   // k-groups (8 k's = 4 MFMA steps) per tile
  static constexpr int WM = TM / 2, WN = TN / 2;      // per-wave tile
  static constexpr int MI = WM / 32, NI = WN / 32;    // 32x32 MFMA tiles per wave
  static constexpr int A4 = TM * C4 / kThreads;       // float4 staged per thread for A
  static constexpr int B4 = TN * C4 / kThreads;       // ... and for B (both layouts)
  static constexpr int CH = 4 * MI;                   // MFMA chunks (NI MFMAs each) per k-group
  static constexpr int NG = A4 + B4;                  // global-load / LDS-write pieces per tile
  static constexpr int GPC = (NG + CH - 1) / CH;      // ... per chunk
  static constexpr int NR = MI + (NN ? 4 * NI : NI);  // fragment-read pieces per k-group
  static constexpr int RPC = (NR + CH - 1) / CH;      // ... per chunk
  static constexpr int WKB = KB >= 3 ? KB - 2 : KB - 1;  // k-group that carries the LDS writes
  using Smem = GemmSmem<TM, TN, NN, BK>;

  f32x16 acc[MI][NI];

  __device__ __forceinline__ void run(Smem& sm, const GemmOperands& g, const float* Ab, const float* Bb,
                                      int row0, int col0, int kbeg, int kend, Pro& pro) {
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    f32x4 ra[A4], rb[B4];
    const float* pa[A4];
    const float* pb[B4];
    if (FULL) {
#pragma unroll
      for (int i = 0; i < A4; ++i) {
        const int idx = t + kThreads * i;
        pa[i] = Ab + (size_t)(row0 + idx / C4) * g.lda + kbeg + 4 * (idx % C4);
      }
#pragma unroll
      for (int i = 0; i < B4; ++i) {
        const int idx = t + kThreads * i;
        pb[i] = NN ? Bb + (size_t)(kbeg + idx / (TN / 4)) * g.ldb + col0 + 4 * (idx % (TN / 4))
                   : Bb + (size_t)(col0 + idx / C4) * g.ldb + kbeg + 4 * (idx % C4);
      }
    }
    const size_t bstep = NN ? (size_t)BK * g.ldb : (size_t)BK;

    // piece p of the global loads of the K-tile starting at k0 (p < A4: A, else B); p == 0 also prefetches
    // whatever the prologue needs for this tile
    auto gload_piece = [&](int p, int k0) __attribute__((always_inline)) {
      if (p == 0) pro.prefetch(k0 + 4 * (t % C4), kend);  // every float4 a thread stages sits in k-column t % C4
      if (p < A4) {
        const int i = p, idx = t + kThreads * i;
        if (FULL) {
          ra[i] = *reinterpret_cast<const f32x4*>(pa[i]);
          pa[i] += BK;
        } else {
          ra[i] = load_row4<VEC>(Ab, g.lda, row0 + idx / C4, g.rowsA, k0 + 4 * (idx % C4), kend);
        }
      } else {
        const int i = p - A4, idx = t + kThreads * i;
        if (FULL) {
          rb[i] = *reinterpret_cast<const f32x4*>(pb[i]);
          pb[i] += bstep;
        } else if (!NN) {
          rb[i] = load_row4<VEC>(Bb, g.ldb, col0 + idx / C4, g.colsB, k0 + 4 * (idx % C4), kend);
        } else {
          // B[k0+r][col0 + 4*c4 ..]: n contiguous; rows are k (bounded by kend), columns by colsB
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          const int kk = k0 + idx / (TN / 4), n = col0 + 4 * (idx % (TN / 4));
          if (kk < kend) {
            const float* ptr = Bb + (size_t)kk * g.ldb + n;
            if (VEC) {
              if (n < g.colsB) v = *reinterpret_cast<const f32x4*>(ptr);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n + e < g.colsB) v[e] = ptr[e];
            }
          }
          rb[i] = v;
        }
      }
    };
    // piece p of the LDS restage; the prologue transform is applied here, long after the load was issued
    auto lstore_piece = [&](int p, int buf) __attribute__((always_inline)) {
      if (p < A4) {
        const int i = p, idx = t + kThreads * i;
        pro.stageA(ra[i], i, row0 + idx / C4);
        *reinterpret_cast<f32x4*>(&sm.a[buf][(idx / C4) * LDK + 4 * (idx % C4)]) = ra[i];
      } else {
        const int i = p - A4, idx = t + kThreads * i;
        if (!NN) {
          pro.stageB(rb[i], i, col0 + idx / C4);
          *reinterpret_cast<f32x4*>(&sm.b[buf][(idx / C4) * LDK + 4 * (idx % C4)]) = rb[i];
        } else {
          *reinterpret_cast<f32x4*>(&sm.b[buf][(idx / (TN / 4)) * TN + 4 * (idx % (TN / 4))]) = rb[i];
        }
      }
    };

    const int ntiles = (kend - kbeg + BK - 1) / BK;
    if (ntiles <= 0) return;
#pragma unroll
    for (int p = 0; p < NG; ++p) gload_piece(p, kbeg);
#pragma unroll
    for (int p = 0; p < NG; ++p) lstore_piece(p, 0);
    __syncthreads();

    f32x4 af[2][MI], bf[2][NI];
    // piece p of the fragment reads of k-group kb out of LDS buffer `buf` into register set `set`
    auto ldfrag_piece = [&](int p, int kb, int buf, int set) __attribute__((always_inline)) {
      if (p < MI) {
        af[set][p] = *reinterpret_cast<const f32x4*>(&sm.a[buf][(wr * WM + p * 32 + l31) * LDK + kb * 8 + 4 * h]);
      } else if (!NN) {
        const int j = p - MI;
        bf[set][j] = *reinterpret_cast<const f32x4*>(&sm.b[buf][(wc * WN + j * 32 + l31) * LDK + kb * 8 + 4 * h]);
      } else {
        const int j = (p - MI) >> 2, s = (p - MI) & 3;
        bf[set][j][s] = sm.b[buf][(kb * 8 + 4 * h + s) * TN + wc * WN + j * 32 + l31];
      }
    };

    // One K-tile out of LDS buffer `cur`; when `more`, also fetches the next tile (k0n) into buffer cur^1.
    auto tile_body = [&](int cur, bool more, int k0n) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < NR; ++p) ldfrag_piece(p, 0, cur, 0);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          __builtin_amdgcn_sched_barrier(0);
          if (kb + 1 < KB) {
#pragma unroll
            for (int p = c * RPC; p < (c + 1) * RPC && p < NR; ++p) ldfrag_piece(p, kb + 1, cur, (kb + 1) & 1);
          }
          if (more && kb == 0) {
#pragma unroll
            for (int p = c * GPC; p < (c + 1) * GPC && p < NG; ++p) gload_piece(p, k0n);
          }
          if (more && kb == WKB) {
#pragma unroll
            for (int p = c * GPC; p < (c + 1) * GPC && p < NG; ++p) lstore_piece(p, cur ^ 1);
          }
          __builtin_amdgcn_sched_barrier(0);
          const int s = c / MI, i = c % MI;
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kb & 1][i][s], bf[kb & 1][j][s], acc[i][j], 0, 0, 0);
        }
      }
    };

    f32x16 tot[CHAIN ? MI : 1][CHAIN ? NI : 1];
    if (CHAIN) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) tot[i][j][q] = 0.f;
    }
    // steady state: the last tile is peeled so that the body has no data-dependent branches
    for (int kt = 0; kt + 1 < ntiles; ++kt) {
      tile_body(kt & 1, true, kbeg + (kt + 1) * BK);
      if (CHAIN && (kt % kChainTiles) == kChainTiles - 1) {
#pragma unroll
        for (int i = 0; i < (CHAIN ? MI : 0); ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              tot[i][j][q] += acc[i][j][q];
              acc[i][j][q] = 0.f;
            }
      }
      __syncthreads();  // tile kt+1 is in buffer (kt+1)&1; buffer kt&1 may be overwritten from the next body on
    }
    tile_body((ntiles - 1) & 1, false, 0);
    if (CHAIN) {
#pragma unroll
      for (int i = 0; i < (CHAIN ? MI : 0); ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[i][j][q] += tot[i][j][q];
    }
  }

  // Visits every accumulator element this lane owns: f(local_row, local_col, value).
  template <class F>
  __device__ __forceinline__ void for_each(F&& f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q)
          f(wr * WM + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h, wc * WN + j * 32 + l31, acc[i][j][q]);
  }
  // Same, also passing the row-tile index i and the register index q (compile-time constants after unrolling).
  template <class F>
  __device__ __forceinline__ void for_each_idx(F&& f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q)
          f(i, q, wr * WM + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h, wc * WN + j * 32 + l31, acc[i][j][q]);
  }
};

struct NoPrologue {
  __device__ __forceinline__ void prefetch(int, int) {}
  __device__ __forceinline__ void stageA(f32x4&, int, int) {}
  __device__ __forceinline__ void stageB(f32x4&, int, int) {}
};

}  // namespace hecka
