// fp32 MFMA GEMM core for gfx950 (MI355X): 256-thread workgroups (4 waves of 64 lanes, 2x2),
// LDS-tiled operands, v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chains), register-staged
// double buffering with one barrier per 32-deep K-tile.
//
// Every GEMM-shaped step of the HE-CKA path instantiates this core:
//   gram   (NT, symmetric, centred)   Kt_m = (X_m - 1 sbar_m^T)(X_m - 1 sbar_m^T)^T      [gram_fwd.hip]
//   pair   (NT, split-K)              S    = Cflat Cflat^T                              [pairwise.hip]
//   comb   (NN)                       dC   = A Cflat   (+ epilogue -> W)                [combine.hip]
//   dgram  (NN, batched)              dX_m = W_m X_m                                    [gram_bwd.hip]
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
#pragma once
#include <hip/hip_runtime.h>

namespace hecka {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBK = 32;        // K-tile depth
constexpr int kLDK = kBK + 4;  // padded LDS row (floats) for k-contiguous tiles: conflict-free b128 reads
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

template <int TM, int TN, bool NN>
struct GemmSmem {
  static constexpr int kA = TM * kLDK;
  static constexpr int kB = NN ? (kBK * TN) : (TN * kLDK);
  float a[2][kA];
  float b[2][kB];
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

// One workgroup's tile of the product.  `Pro` may transform staged A/B values (centering);
// the caller consumes the accumulators through `acc` after the call.
template <int TM, int TN, bool NN, bool VEC, class Pro>
struct GemmTile {
  static constexpr int WM = TM / 2, WN = TN / 2;      // per-wave tile
  static constexpr int MI = WM / 32, NI = WN / 32;    // 32x32 MFMA tiles per wave
  static constexpr int A4 = TM / 32;                  // float4 staged per thread for A
  static constexpr int B4 = TN / 32;                  // ... and for B (both layouts)
  using Smem = GemmSmem<TM, TN, NN>;

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

    auto gload = [&](int k0) {
#pragma unroll
      for (int i = 0; i < A4; ++i) {
        const int idx = t + kThreads * i;
        const int r = idx >> 3, c4 = idx & 7;
        ra[i] = load_row4<VEC>(Ab, g.lda, row0 + r, g.rowsA, k0 + 4 * c4, kend);
      }
      pro.prefetch(k0 + 4 * (t & 7), kend);  // every float4 this thread stages sits in k-column (t & 7)
      if (!NN) {
#pragma unroll
        for (int i = 0; i < B4; ++i) {
          const int idx = t + kThreads * i;
          const int r = idx >> 3, c4 = idx & 7;
          rb[i] = load_row4<VEC>(Bb, g.ldb, col0 + r, g.colsB, k0 + 4 * c4, kend);
        }
      } else {
#pragma unroll
        for (int i = 0; i < B4; ++i) {
          const int idx = t + kThreads * i;
          const int r = idx / (TN / 4), c4 = idx % (TN / 4);
          // B[k0+r][col0 + 4*c4 ..]: n contiguous; rows are k (bounded by kend), columns by colsB
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          const int kk = k0 + r, n = col0 + 4 * c4;
          if (kk < kend) {
            const float* p = Bb + (size_t)kk * g.ldb + n;
            if (VEC) {
              if (n < g.colsB) v = *reinterpret_cast<const f32x4*>(p);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n + e < g.colsB) v[e] = p[e];
            }
          }
          rb[i] = v;
        }
      }
    };
    // The prologue transform is applied HERE (after the MFMAs of the previous tile), never in gload:
    // touching a freshly loaded register there would put the global-load latency in front of the MFMAs.
    auto lstore = [&](int buf) {
#pragma unroll
      for (int i = 0; i < A4; ++i) {
        const int idx = t + kThreads * i;
        pro.stageA(ra[i], i, row0 + (idx >> 3));
        *reinterpret_cast<f32x4*>(&sm.a[buf][(idx >> 3) * kLDK + 4 * (idx & 7)]) = ra[i];
      }
#pragma unroll
      for (int i = 0; i < B4; ++i) {
        const int idx = t + kThreads * i;
        if (!NN) pro.stageB(rb[i], i, col0 + (idx >> 3));
        if (!NN)
          *reinterpret_cast<f32x4*>(&sm.b[buf][(idx >> 3) * kLDK + 4 * (idx & 7)]) = rb[i];
        else
          *reinterpret_cast<f32x4*>(&sm.b[buf][(idx / (TN / 4)) * TN + 4 * (idx % (TN / 4))]) = rb[i];
      }
    };

    const int ntiles = (kend - kbeg + kBK - 1) / kBK;
    if (ntiles <= 0) return;
    gload(kbeg);
    lstore(0);
    __syncthreads();

    for (int kt = 0; kt < ntiles; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < ntiles) gload(kbeg + (kt + 1) * kBK);  // in flight during the MFMAs below

      const float* as = sm.a[cur];
      const float* bs = sm.b[cur];
#pragma unroll
      for (int kb = 0; kb < kBK / 8; ++kb) {
        f32x4 af[MI], bf[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
          af[i] = *reinterpret_cast<const f32x4*>(&as[(wr * WM + i * 32 + l31) * kLDK + kb * 8 + 4 * h]);
        if (!NN) {
#pragma unroll
          for (int j = 0; j < NI; ++j)
            bf[j] = *reinterpret_cast<const f32x4*>(&bs[(wc * WN + j * 32 + l31) * kLDK + kb * 8 + 4 * h]);
        } else {
#pragma unroll
          for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) bf[j][s] = bs[(kb * 8 + 4 * h + s) * TN + wc * WN + j * 32 + l31];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
      }

      if (kt + 1 < ntiles) {
        lstore(cur ^ 1);  // the other buffer was last read one iteration ago, behind the barrier below
        __syncthreads();
      }
    }
  }

  // Visits every accumulator element this lane owns: f(local_row, local_col, value&).
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
};

struct NoPrologue {
  __device__ __forceinline__ void prefetch(int, int) {}
  __device__ __forceinline__ void stageA(f32x4&, int, int) {}
  __device__ __forceinline__ void stageB(f32x4&, int, int) {}
};

}  // namespace hecka
