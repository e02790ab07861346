// fp32-accurate GEMM on the fp16 matrix cores from PRE-SPLIT operands ("planes").
//
// Every fp32 operand x is stored once in HBM as two fp16 planes  x * s = h + l  (h = rne16(x s), l = rne16(x s - h),
// s a power of two that brings the operand into fp16 range: 22 significand bits), and a.b is rebuilt from the three
// leading products hh + hl + lh on v_mfma_f32_32x32x16_f16.  Compared with splitting inside the GEMM
// (gemm_bf16x6.hpp) the k-loop carries no VALU split work and no ds_write: the planes go L2 -> LDS directly
// (global_load_lds, 16 B per lane), and the wave's issue stream is ds_read + MFMA only.
//
// HBM layout of a plane ("panel layout"): an operand [R rows][K] is stored as [K/32 panels][R rows][32 k] fp16, so
// the rows x 32-k tile of one k-tile is CONTIGUOUS (full 128-byte lines, eight per wave-load), and so is a
// 32-row x 32-col block of an operand used as [k][cols] (the NN case: its rows are the k dimension).
//
// Tile 256x128, 512 threads (4x2 waves of 64x64), BK = 32, THREE LDS stages of 48 KiB (one workgroup per CU, two
// waves per SIMD).  The L2 -> CU path sustains about 70 GB/s per CU (measured, load-only ablation of the 128x128
// version: 35 B/clk/CU), which a 128x128 tile would need in full at the MFMA rate; 256x128 needs 3/4 of it.
// LDS images are lane-linear per wave-load (that is how LDS-DMA writes):
//   row image (operand rows = GEMM rows): 64-byte rows; 16-byte chunk c of row r sits at chunk c ^ ((r >> 2) & 3)
//     (the permutation is applied to the per-lane SOURCE address, inside one 64-byte segment): ds_read_b128 of a
//     32-row x 8-k fragment is conflict-free;
//   column image (NN: operand rows = k): [4 column panels][32 k-rows][64 B], unswizzled; fragments come out through
//     ds_read_b64_tr_b16 (hardware transpose read) - a 32-lane half reads 4 k-rows x 64 B = 256 contiguous bytes.
// Pipeline: loads run two k-tiles ahead (counted vmcnt, raw s_barrier, one barrier per k-tile); fragments of k-step
// g+1 are read while the 12 MFMAs of k-step g issue, across the barrier as well.
// The 16-bit MFMA truncates its sums, so at most four k-steps (64 k) are chained in a LOCAL accumulator that starts
// from zero and is folded into the running sum with a round-to-nearest fp32 FMA; the fold multiplies by a per-chunk
// weight (fold(kchunk)) so the operands may carry a different power-of-two scale every 64 k.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../gemm_mfma.hpp"

namespace hecka {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int kPlThreads = 512;
constexpr int kPlTM = 256, kPlTN = 128, kPlBK = 32;
constexpr int kPlABytes = kPlTM * kPlBK * 2;                 // one A plane of one stage: 16 KiB
constexpr int kPlBBytes = kPlTN * kPlBK * 2;                 // one B plane of one stage:  8 KiB
constexpr int kPlStageBytes = 2 * kPlABytes + 2 * kPlBBytes; // 48 KiB: [A hi][A lo][B hi][B lo]
constexpr int kPlStages = 3;
constexpr int kPlSmemBytes = kPlStages * kPlStageBytes;      // 144 KiB (dynamic LDS)

#define HECKA_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef __attribute__((address_space(3))) unsigned char lds_u8;

// NN = false: out[i,j] = sum_k A[i,k] B[j,k]   (both operands [rows][K] planes)
// NN = true : out[i,j] = sum_k A[i,k] B[k,j]   (B planes are those of the [K][cols] matrix: its rows are k)
// Requirements (checked by the host): tile fully inside the operands, K % 64 == 0.
template <bool NN>
struct GemmTilePlanes {
  static constexpr int MI = 2, NI = 2;
  f32x16 acc[MI][NI];

  // smem: kPlSmemBytes of LDS, 1 KiB aligned.  Ah/Al/Bh/Bl: this member's planes (fp16, panel layout); rowsA / rowsB:
  // rows of the A / B operand matrices (the panel pitch is rows * 64 B); (row0, col0): tile origin; K: depth;
  // fold(kc) -> weight of 64-k chunk kc
  template <class Fold>
  __device__ __forceinline__ void run(lds_u8* smem, const _Float16* Ah, const _Float16* Al, int rowsA,
                                      const _Float16* Bh, const _Float16* Bl, int rowsB, int row0, int col0, int K,
                                      Fold&& fold) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // ---- staging: wave-loads of 1 KiB = 16 rows of 64 B; A plane = 16 loads (waves take w, w + 8), B plane = 8 ----
    const int rrow = lane >> 2, rchk = lane & 3;
    const char* srcA[2][2];   // [plane][which of this wave's two loads], byte pointers at k = 0
    const char* srcB[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = 16 * (wave + 8 * u) + rrow;
      const size_t offA = (size_t)(row0 + r) * 64 + (size_t)((rchk ^ ((r >> 2) & 3)) * 16);
      srcA[0][u] = reinterpret_cast<const char*>(Ah) + offA;
      srcA[1][u] = reinterpret_cast<const char*>(Al) + offA;
    }
    {
      size_t offB;
      if (NN) {   // load i: column panel i >> 1, k-rows 16 (i & 1) .. + 15
        offB = ((size_t)(col0 / 32 + (wave >> 1)) * rowsB + 16 * (wave & 1) + rrow) * 64 + (size_t)(rchk * 16);
      } else {
        const int r = 16 * wave + rrow;
        offB = (size_t)(col0 + r) * 64 + (size_t)((rchk ^ ((r >> 2) & 3)) * 16);
      }
      srcB[0] = reinterpret_cast<const char*>(Bh) + offB;
      srcB[1] = reinterpret_cast<const char*>(Bl) + offB;
    }
    const size_t stepA = (size_t)rowsA * 64, stepB = NN ? (size_t)kPlBK * 64 : (size_t)rowsB * 64;
    const int ntiles = K / kPlBK;
    // one of the 6 wave-loads of k-tile kt: g = 3 p + {0, 1: A loads, 2: B load}
    auto stage1 = [&](int kt, int g) __attribute__((always_inline)) {
      lds_u8* sb = smem + (kt % kPlStages) * kPlStageBytes;
      const int p = g / 3, u = g % 3;
      if (u < 2)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>(srcA[p][u] + kt * stepA),
                                         (__attribute__((address_space(3))) void*)(sb + p * kPlABytes + (wave + 8 * u) * 1024), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>(srcB[p] + kt * stepB),
                                         (__attribute__((address_space(3))) void*)(sb + 2 * kPlABytes + p * kPlBBytes + wave * 1024), 16, 0, 0);
    };
    auto stage = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
      for (int g = 0; g < 6; ++g) stage1(kt, g);
    };

    // ---- fragment read offsets (bytes from the stage base); k-step s flips bit 5 (chunk 2 s + h) ----
    int offa[2], offb[2];   // [s]
    {
      const int r = wr * 64 + l31;   // second row block: + 32 rows = + 2048 B (same swizzle: (r >> 2) & 3 unchanged)
      offa[0] = r * 64 + ((h ^ ((r >> 2) & 3)) * 16);
      offa[1] = offa[0] ^ 32;
    }
    if (NN) {
      // transposed reads; lane 4 q + p of 16-lane group g addresses k-row 8 (g >> 1) + q (+ 4 e + 16 s), 8 bytes at
      // column 16 (g & 1) + 4 p of column panel 2 wc + j
      const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
      offb[0] = 2 * kPlABytes + (2 * wc) * 2048 + (8 * (g >> 1) + q) * 64 + (16 * (g & 1) + 4 * p) * 2;
      offb[1] = offb[0] + 16 * 64;
    } else {
      const int r = wc * 64 + l31;
      offb[0] = 2 * kPlABytes + r * 64 + ((h ^ ((r >> 2) & 3)) * 16);
      offb[1] = offb[0] ^ 32;
    }

    h16x8 fa[2][MI][2], fb[2][NI][2];   // [register set][block][plane]
    f32x16 loc[MI][NI];
    // fragment read r of 8 for k-step (kt, s) into register set S: r < 4: A block r >> 1 plane r & 1; else B likewise
    auto read1 = [&](auto SET, const lds_u8* sb, int s, int r) __attribute__((always_inline)) {
      constexpr int S = decltype(SET)::value;
      const int blk = (r >> 1) & 1, pl = r & 1;
      if (r < 4) {
        fa[S][blk][pl] = *reinterpret_cast<const __attribute__((address_space(3))) h16x8*>(sb + offa[s] + blk * 2048 + pl * kPlABytes);
      } else if (NN) {
        s16x4 v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e)
          v[e] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(
              sb + offb[s] + blk * 2048 + e * 256 + pl * kPlBBytes));
        const s16x8 w = {v[0][0], v[0][1], v[0][2], v[0][3], v[1][0], v[1][1], v[1][2], v[1][3]};
        fb[S][blk][pl] = __builtin_bit_cast(h16x8, w);
      } else {
        fb[S][blk][pl] = *reinterpret_cast<const __attribute__((address_space(3))) h16x8*>(sb + offb[s] + blk * 2048 + pl * kPlBBytes);
      }
    };
    auto read_frags = [&](auto SET, int kt, int s) __attribute__((always_inline)) {
      const lds_u8* sb = smem + (kt % kPlStages) * kPlStageBytes;
#pragma unroll
      for (int r = 0; r < 8; ++r) read1(SET, sb, s, r);
    };
    // barrier that opens k-tile kt: this wave's loads of tile kt have landed (those of tile kt + 1 may stay in
    // flight), every wave has finished reading tile kt - 1, whose buffer then takes tile kt + 2
    auto open_tile = [&](int kt) __attribute__((always_inline)) {
#ifndef PL_ABL_NOLOAD
      if (kt + 1 < ntiles) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
      __builtin_amdgcn_s_barrier();
    };
    // One slot: the 12 MFMAs of the k-step held in register set S, with the side work spread one piece per MFMA
    // gap: the 8 fragment reads of the next k-step (kt_r, s_r) into the other set, and (LOADS) the 6 wave-loads of
    // k-tile kt_l.  FIRST: the local chain starts from zero.
    auto slot = [&](auto SET, auto FIRST, bool reads, int kt_r, int s_r, bool loads, int kt_l) __attribute__((always_inline)) {
      constexpr int S = decltype(SET)::value;
      using OTHER = std::integral_constant<int, 1 - S>;
      const lds_u8* sb = smem + (kt_r % kPlStages) * kPlStageBytes;
#pragma unroll
      for (int n = 0; n < 12; ++n) {
        const int e = n >> 2, i = (n >> 1) & 1, j = n & 1;
        const int pa = e == 2 ? 1 : 0, pb = e == 1 ? 1 : 0;   // hh, hl, lh
        f32x16 c = loc[i][j];
        if (decltype(FIRST)::value && e == 0) {
#pragma unroll
          for (int q = 0; q < 16; ++q) c[q] = 0.f;
        }
#ifdef PL_ABL_NOMFMA
        if (e == 0) { c[0] += (float)fa[S][i][0][0] + (float)fa[S][i][1][1]; c[1] += (float)fb[S][j][0][0] + (float)fb[S][j][1][1]; }
        loc[i][j] = c;
#else
        loc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][i][pa], fb[S][j][pb], c, 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (n < 8) { if (reads) read1(OTHER{}, sb, s_r, n); }
#ifndef PL_ABL_NOLOAD
        if (n >= 1 && (n & 1)) { if (loads) stage1(kt_l, n >> 1); }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    std::true_type T;
    std::false_type F;

    stage(0);
    if (ntiles > 1) stage(1);
    open_tile(0);
    read_frags(I0{}, 0, 0);
    if (ntiles > 2) stage(2);
    // k-steps g = 2 kt + s; the slot of step g reads the fragments of step g + 1; the slot behind the barrier that
    // opens tile kt + 1 also issues the loads of tile kt + 3
    for (int kc = 0; kc < ntiles / 2; ++kc) {
      const int kt = 2 * kc;
      slot(I0{}, T, true, kt, 1, false, 0);                                   // step (kt, 0)
      open_tile(kt + 1);
      slot(I1{}, F, true, kt + 1, 0, kt + 3 < ntiles, kt + 3);                // step (kt, 1)
      slot(I0{}, F, true, kt + 1, 1, false, 0);                               // step (kt + 1, 0)
      const bool more = kt + 2 < ntiles;
      if (more) open_tile(kt + 2);
      slot(I1{}, F, more, kt + 2, 0, kt + 4 < ntiles, kt + 4);                // step (kt + 1, 1)
      const float w = fold(kc);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[i][j][q] = __builtin_fmaf(loc[i][j][q], w, acc[i][j][q]);
    }
  }

  template <class Fn>
  __device__ __forceinline__ void for_each(Fn&& f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q)
          f(wr * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h, wc * 64 + j * 32 + l31, acc[i][j][q]);
  }
};

}  // namespace hecka
