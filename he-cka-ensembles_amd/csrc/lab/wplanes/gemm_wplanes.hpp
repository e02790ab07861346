// dX = W X with the A operand (W) PRE-SPLIT into two fp16 planes and the B operand (X, fp32, [k][n]) split in the loop.
//
// In gemm_bf16x6.hpp every workgroup splits both of its operand tiles on the way into LDS; the wave's issue stream
// (VALU split + ds_write + MFMA) is what limits it.  For dX = W X the W tile of a member is re-split by every one of
// the P/128 column tiles, although W is tiny (4 B^2 bytes per member) and is written by a kernel of ours: the combine
// kernel now stores W already scaled and split (two fp16 planes in "panel layout": [B/16 k-panels][B rows][16 k], so
// the 128-row x 16-k tile of a k-tile is 4 KiB contiguous), and here the planes go L2 -> LDS with global_load_lds:
// no registers, no VALU, no ds_write for A.  That halves the split work per MFMA.
//
// Tile 128x128, 256 threads (2x2 waves of 64x64), BK = 16, two workgroups per CU - as gemm_bf16x6.hpp (K = B is
// short: two workgroups per CU hide each other's prologue and epilogue).
// EVERY global read of the loop is an LDS-DMA: hipcc (ROCm 7.2) drains the whole vector-memory queue (vmcnt(0)) at the
// first use of an ordinary load's result while an LDS-DMA is in flight, which would undo the prefetch - so the fp32 X
// tile also lands in LDS raw ([16 k][128 n] fp32, three stages), and the threads split it from there (a wave reads 256
// contiguous bytes of one k-row: conflict-free ds_read_b32) into the padded fp16 planes the B fragments are read from.
// With nothing but LDS-DMA in the queue, the one wait per k-tile is an exact count: s_waitcnt vmcnt(4).
// LDS: A three stages x two planes x 4 KiB (32-byte rows, 16-byte chunk c of row r stored at c ^ ((r >> 4) & 1):
// lane-linear per wave-load, conflict-free ds_read_b128), raw X three stages x 8 KiB, B planes two stages x two planes
// of padded 48-byte rows: 72 KiB.  Loads run two (A) and three (raw X) k-tiles ahead, raw s_barrier, one per k-tile;
// the split + LDS write of B tile t+1 is spread over the MFMA gaps of tile t.  Local 64-k accumulation chains folded
// with round-to-nearest fp32 adds, as in gemm_bf16x6.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../gemm_bf16x6.hpp"

namespace hecka {

typedef __attribute__((address_space(3))) unsigned char lds_u8;

struct WPSmem {
  alignas(1024) unsigned char a[3][2][4096];              // [stage][plane][128 rows x 32 B]
  alignas(1024) float braw[3][16 * 128];                  // [stage][k][n] fp32, as loaded
  alignas(16) unsigned short b[2][2][128 * kX6LD];        // [stage][plane][128 cols x 24 halfs]
};

struct GemmTileWP {
  static constexpr int MI = 2, NI = 2;
  f32x16 acc[MI][NI];

  // Wh / Wl: this member's planes (fp16, panel layout, rowsW = B rows per panel); Xb: this member's X [K][ldb] fp32,
  // tile origin (row0, col0); K = B (multiple of 64, >= 64); sb: power-of-two scale of X
  __device__ __forceinline__ void run(WPSmem& sm, const _Float16* Wh, const _Float16* Wl, int rowsW, const float* Xb,
                                      int ldb, int row0, int col0, int K, float sb) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5;
    f32x16 loc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const int ntiles = K / kX6BK;

    // ---- A: one wave-load (1 KiB = 32 rows x 32 B) per plane per wave per k-tile ----
    const char* srcA[2];
    {
      const int r = 32 * wave + (lane >> 1);
      const size_t off = (size_t)(row0 + r) * 32 + (size_t)(((lane & 1) ^ ((r >> 4) & 1)) * 16);
      srcA[0] = reinterpret_cast<const char*>(Wh) + off;
      srcA[1] = reinterpret_cast<const char*>(Wl) + off;
    }
    const size_t stepA = (size_t)rowsW * 32;   // bytes between k-panels
    lds_u8* smA = (lds_u8*)&sm.a[0][0][0];
    auto glds_a = [&](int kt) __attribute__((always_inline)) {
      lds_u8* dst = smA + (kt % 3) * 8192 + wave * 1024;
#pragma unroll
      for (int p = 0; p < 2; ++p)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>(srcA[p] + kt * stepA),
                                         (__attribute__((address_space(3))) void*)(dst + p * 4096), 16, 0, 0);
    };
    int offa[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int r = wr * 64 + i * 32 + l31;
      offa[i] = r * 32 + ((h ^ ((r >> 4) & 1)) * 16);
    }

    // ---- B raw: the 16 x 128 fp32 tile = 8 wave-loads of 2 k-rows x 512 B; waves take loads w and w + 4 ----
    const char* srcB[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int kr = 2 * (wave + 4 * u) + (lane >> 5);
      srcB[u] = reinterpret_cast<const char*>(Xb) + ((size_t)kr * ldb + col0 + 4 * (lane & 31)) * 4;
    }
    const size_t stepB = (size_t)kX6BK * ldb * 4;
    lds_u8* smR = (lds_u8*)&sm.braw[0][0];
    auto glds_b = [&](int kt) __attribute__((always_inline)) {
      lds_u8* dst = smR + (kt % 3) * 8192;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>(srcB[u] + kt * stepB),
                                         (__attribute__((address_space(3))) void*)(dst + (wave + 4 * u) * 1024), 16, 0, 0);
    };
    // split: thread t takes column n = t & 127, k = 8 (t >> 7) + 4 quad + c of the raw tile -> 4 halfs per plane at
    // [n][8 (t >> 7) + 4 quad] of the padded plane image
    const int roff = (8 * (t >> 7)) * 128 + (t & 127);   // floats
    int boff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) boff[i] = (t & 127) * kX6LD + 8 * (t >> 7) + 4 * i;
    auto split_b = [&](int kt, int quad, u32x2* qv) __attribute__((always_inline)) {
      const float* raw = &sm.braw[kt % 3][roff + quad * 4 * 128];
      f32x4 v;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = raw[c * 128] * sb;
      split2h(v, 0, qv[0], qv[1], 0);
      split2h(v, 2, qv[0], qv[1], 1);
    };
    auto write_b = [&](int st, int quad, const u32x2* qv) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 2; ++q) *reinterpret_cast<u32x2*>(&sm.b[st][q][boff[quad]]) = qv[q];
    };

    // prologue.  LDS-DMA issue order: raw0 A0 raw1 A1 raw2 (two wave-loads each)
    glds_b(0);
    glds_a(0);
    glds_b(1);
    glds_a(1);
    glds_b(2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // raw0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
      u32x2 qv[2];
#pragma unroll
      for (int quad = 0; quad < 2; ++quad) {
        split_b(0, quad, qv);
        write_b(0, quad, qv);
      }
    }
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");   // A0, raw1 (A1, raw2 may fly); B planes of tile 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // one k-tile: 12 MFMAs (hh, hl, lh on the four 32x32 blocks) out of A stage kt % 3 and B plane stage kt & 1.
    // top: LDS-DMA of A tile kt+2 and raw X tile kt+3 | gaps 0..3: split / write of the two B quads of tile kt+1
    // STEADY: tiles kt+1 .. kt+3 exist (no branches); otherwise guarded.
    auto iter = [&](auto FIRST, auto STEADY, int kt) __attribute__((always_inline)) {
      constexpr bool G = !decltype(STEADY)::value;
      const int st = kt & 1;
      const bool more = !G || kt + 1 < ntiles;
      const bool loadb = !G || kt + 3 < ntiles;
      const bool loada = !G || kt + 2 < ntiles;
      if (loada) glds_a(kt + 2);
      if (loadb) glds_b(kt + 3);
      const lds_u8* pa = smA + (kt % 3) * 8192;
      f16x8 fa[MI][2], fb[NI][2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
          fa[i][p] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>(pa + p * 4096 + offa[i]);
#pragma unroll
        for (int j = 0; j < NI; ++j)
          fb[j][p] = *reinterpret_cast<const f16x8*>(&sm.b[st][p][(wc * 64 + j * 32 + l31) * kX6LD + 8 * h]);
      }
      __builtin_amdgcn_sched_barrier(0);
      constexpr int PA[3] = {0, 0, 1};
      constexpr int PB[3] = {0, 1, 0};
      u32x2 qv[2];
#pragma unroll
      for (int n = 0; n < 12; ++n) {
        const int e = n >> 2, i = (n >> 1) & 1, j = n & 1;
        {
          f32x16 c = loc[i][j];
          if (decltype(FIRST)::value && e == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) c[q] = 0.f;
          }
          loc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][PA[e]], fb[j][PB[e]], c, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (n < 4 && more) {
          if ((n & 1) == 0) split_b(kt + 1, n >> 1, qv);
          else write_b(st ^ 1, n >> 1, qv);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // tile kt+1 must be complete in LDS: A tile kt+1 and raw X tile kt+2 (only this iteration's two pairs of
      // LDS-DMA are younger in the queue) and this wave's B plane writes
      if (!G || kt + 3 < ntiles) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else if (kt + 2 < ntiles) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    auto fold = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[i][j][q] += loc[i][j][q];   // v_add_f32, round to nearest
    };
    std::true_type T;
    std::false_type F;
    int kt = 0;
    for (; kt + 7 <= ntiles; kt += 4) {   // steady state: tiles kt .. kt+3 all have a tile kt+3 behind them
      iter(T, T, kt);
      iter(F, T, kt + 1);
      iter(F, T, kt + 2);
      iter(F, T, kt + 3);
      fold();
    }
    while (kt < ntiles) {
      iter(T, F, kt++);
      if (kt < ntiles) iter(F, F, kt++);
      if (kt < ntiles) iter(F, F, kt++);
      if (kt < ntiles) iter(F, F, kt++);
      fold();
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
