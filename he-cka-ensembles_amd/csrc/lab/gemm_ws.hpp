// Wave-specialised f16x3 GEMM for dX = W X (NN, K = B <= 1024: one accumulation chain, no local folds).
//
// The tile core in gemm_bf16x6.hpp gives every wave the whole job: global loads, the fp16 split (VALU), LDS writes,
// fragment reads and MFMAs share one in-order issue stream, and a stall on any of them (a load that has not landed, the
// LDS write queue, the barrier) also stops that wave's MFMAs: measured 46 % matrix-pipe occupancy.  Here a workgroup
// of eight waves - two per SIMD - splits the roles:
//   consumers (waves 0-3, 2 x 2 over a 256 x 128 tile, 128 x 64 each): ds_read fragments + MFMA, nothing else; the
//       fragments of k-tile g+1 are read (second register set) between the 24 MFMAs of k-tile g;
//   producers (waves 4-7): buffer loads three k-tiles ahead (three register sets), scale + split each fp32 value into
//       two fp16 terms, write the planes of k-tile g+2 into a ring of three LDS stages.
// A SIMD holds one consumer and one producer, so the producer's VALU / LDS / VMEM instructions issue in the shadow of the
// consumer's 32-cycle MFMAs instead of in front of them.  One s_barrier per k-tile keeps the two roles within one tile
// of each other.  The workgroup is PERSISTENT: it walks a list of output tiles and the k-tile stream runs on across
// tile boundaries (the producers already stage the next tile's first k-tiles while the consumers store the finished one).
//
// LDS (96 KiB, one workgroup per CU): A planes [256 rows][16 k] fp16 at a 48-byte pitch (conflict-free ds_read_b128);
// X planes as they lie in memory, [4 column panels][16 k-rows][32 columns] - a thread stages four consecutive COLUMNS
// of one k-row (one 16-byte load: whole 128-byte lines per 8 lanes) and the consumers read the MFMA B fragments with
// the transposing ds_read_b64_tr_b16.  The products of one k-tile are issued in the order of the tile core
// (hh for every block, then h.l, then l.h), so results are bit-identical to GemmTileX6<.., FOLD = false>.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../gemm_bf16x6.hpp"

namespace hecka {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) unsigned char lds_u8;

constexpr int kWsThreads = 512;
constexpr int kWsTM = 256, kWsTN = 128, kWsBK = 16, kWsStages = 3;
constexpr int kWsAPlane = kWsTM * 48;            // bytes: 256 rows x 48 B
constexpr int kWsBPlane = 4 * 16 * 64;           // bytes: 4 panels x 16 k-rows x 64 B
constexpr int kWsStage = 2 * kWsAPlane + 2 * kWsBPlane;   // 32768 B: [A hi][A lo][X hi][X lo]
constexpr int kWsSmem = kWsStages * kWsStage;    // 98304 B
constexpr int kWsMaxTiles = 1024;                // output tiles per workgroup (16 KiB table behind the stages)
constexpr int kWsSmemTotal = kWsSmem + kWsMaxTiles * 16;

struct WsTile {      // one output tile of the persistent walk
  int m, ti, tj;
};

// n-th tile of this workgroup.  With M % 8 == 0 workgroup w (on XCD w % 8) takes members w % 8, w % 8 + 8, ... and the
// (gridDim.x / 8) workgroups of an XCD walk those members' tiles side by side, row tiles fastest: at any moment an XCD
// works on ONE member - its W (1 MiB at B = 512) stays in that XCD's L2 and every X panel is fetched once.
__device__ __forceinline__ bool ws_tile(int n, int M, int tilesM, int tilesN, bool xcd_map, WsTile& t) {
  const int per = tilesM * tilesN;
  long long L;
  if (xcd_map) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, nj = gridDim.x >> 3;
    L = (long long)j + (long long)n * nj;
    if (L >= (long long)(M >> 3) * per) return false;
    t.m = (int)(L / per) * 8 + xcd;
  } else {
    L = (long long)blockIdx.x + (long long)n * gridDim.x;
    if (L >= (long long)M * per) return false;
    t.m = (int)(L / per);
  }
  const int tl = (int)(L % per);
  t.ti = tl % tilesM;
  t.tj = tl / tilesM;
  return true;
}

// barrier between the roles: the producers' LDS writes (and the consumers' LDS reads) are complete, global loads stay in
// flight (no vmcnt wait)
__device__ __forceinline__ void ws_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// EX: element kind of X / dX (fp32 here; the 16-bit variants keep the tile core).
__device__ __forceinline__ void ws_gram_bwd_body(lds_u8* smem, const float* __restrict__ W, const float* __restrict__ X,
                                                 const unsigned* __restrict__ wmax, const float* __restrict__ xmax,
                                                 float* __restrict__ dX, int M, int B, int P, int tilesM, int tilesN,
                                                 bool xcd_map, unsigned long long* dbg = nullptr) {
#ifdef WS_DBG
  unsigned long long t_bar = 0, t_epi = 0, t_work = 0, t0_, t_all = clock64();
#define WS_T0() t0_ = clock64()
#define WS_ACC(v) v += clock64() - t0_
#else
#define WS_T0()
#define WS_ACC(v)
#endif
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int KT = B / kWsBK;                       // k-tiles per output tile
  // this workgroup's list of output tiles in LDS behind the stages - (m << 16 | ti << 12 | tj), scale of W, scale of X,
  // 1 / (their product) - so that the index arithmetic (64-bit divisions) and the reads of wmax / xmax happen once here
  // and not in the k-tile stream (a global load there would drain the producers' prefetch queue: vmcnt(0))
  __attribute__((address_space(3))) u32x4* tab = (__attribute__((address_space(3))) u32x4*)(smem + kWsSmem);
  int ntile = 0;
  {
    const int per = tilesM * tilesN;
    const long long total = xcd_map ? (long long)(M >> 3) * per : (long long)M * per;
    const long long first = xcd_map ? (blockIdx.x >> 3) : blockIdx.x, step = xcd_map ? (gridDim.x >> 3) : gridDim.x;
    if (first < total) ntile = (int)((total - first + step - 1) / step);
    if (ntile > kWsMaxTiles) ntile = kWsMaxTiles;      // (the host never launches more)
    for (int n = t; n < ntile; n += kWsThreads) {
      WsTile q;
      ws_tile(n, M, tilesM, tilesN, xcd_map, q);
      const float sa = pow2_scale(__uint_as_float(wmax[q.m]), 14), sb = pow2_scale(xmax[q.m], 14);
      tab[n] = u32x4{((unsigned)q.m << 16) | ((unsigned)q.ti << 12) | (unsigned)q.tj, __float_as_uint(sa), __float_as_uint(sb),
                     __float_as_uint(1.0f / (sa * sb))};
    }
    __syncthreads();
  }
  if (ntile == 0) return;
#ifndef WS_NO_SKEW
  // The workgroups of a launch would otherwise run in step and all store their finished tiles at the same moment (32 MiB
  // in one burst, then nothing for a whole tile): start them spread over one tile's duration.
  {
    const int slot = xcd_map ? ((blockIdx.x >> 3) & 31) : (blockIdx.x & 31);
    for (int i = 0; i < slot * KT / 32; ++i) __builtin_amdgcn_s_sleep(24);      // one step ~ (k-tiles per tile) * 800 cycles / 32
  }
#endif
  auto tile_at = [&](int n, WsTile& q, float& sa, float& sb, float& unscale) __attribute__((always_inline)) {
    const u32x4 e4 = tab[n];
    const unsigned e = (unsigned)__builtin_amdgcn_readfirstlane((int)e4[0]);
    q.m = (int)(e >> 16);
    q.ti = (int)((e >> 12) & 15u);
    q.tj = (int)(e & 4095u);
    sa = __uint_as_float(e4[1]);
    sb = __uint_as_float(e4[2]);
    unscale = __uint_as_float(e4[3]);
  };
  const int G = ntile * KT;                       // k-tiles in this workgroup's stream
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

  if (wave < 4) {
    // ------------------------------------------------------------------ consumers
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const int offa = ((wr * 128 + l31) * 48 + 16 * h);                     // + i * 32 * 48, + plane * kWsAPlane
    int offb;
    {
      const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
      offb = 2 * kWsAPlane + (2 * wc) * (16 * 64) + (8 * (g >> 1) + q) * 64 + (16 * (g & 1) + 4 * p) * 2;   // + j * 1024, + e * 256, + plane * kWsBPlane
    }
    // ONE set of fragment registers; every group is re-read as soon as its last MFMA of the k-tile has issued.  Product
    // order inside a k-tile: A.l B.h | A.h B.h | A.h B.l, so that
    //   gaps 0-3  : A.h of THIS k-tile (first used by MFMA 8)      gaps 4-7  : B.l of this k-tile (first used by MFMA 16)
    //   gaps 8-11 : A.l of the NEXT k-tile (free since MFMA 7)     gaps 16-19: B.h of the next k-tile (free since MFMA 15)
    f16x8 fa[4][2];                                // [block][plane]
    s16x4 fbv[2][2][2];                            // [block][plane][k half]
    auto rdA = [&](const lds_u8* sb, int blk, int pl) __attribute__((always_inline)) {
      fa[blk][pl] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>(sb + offa + blk * (32 * 48) + pl * kWsAPlane);
    };
    auto rdB = [&](const lds_u8* sb, int blk, int pl, int e) __attribute__((always_inline)) {
      fbv[blk][pl][e] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(sb + offb + blk * 1024 + e * 256 + pl * kWsBPlane));
    };
    auto ktile = [&](auto FIRST, const lds_u8* sbc, bool reads, const lds_u8* sbn) __attribute__((always_inline)) {
      constexpr int PA[3] = {1, 0, 0};
      constexpr int PB[3] = {0, 0, 1};
#pragma unroll
      for (int n = 0; n < 24; ++n) {
        const int e = n >> 3, i = (n >> 1) & 3, j = n & 1;
        const s16x4 b0 = fbv[j][PB[e]][0], b1 = fbv[j][PB[e]][1];
        const s16x8 bw = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#ifdef WS_ABL_NOMFMA
        if (n % 8 == 0) acc[i][j][0] += (float)fa[i][PA[e]][0] + (float)__builtin_bit_cast(f16x8, bw)[1];
#else
        {
          f32x16 c = acc[i][j];
          if (decltype(FIRST)::value && e == 0) {      // first k-tile of an output tile: the chain starts from zero
#pragma unroll
            for (int q = 0; q < 16; ++q) c[q] = 0.f;
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][PA[e]], __builtin_bit_cast(f16x8, bw), c, 0, 0, 0);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (n < 4) rdA(sbc, n, 0);
        else if (n < 8) rdB(sbc, (n - 4) >> 1, 1, (n - 4) & 1);
        else if (n < 12) { if (reads) rdA(sbn, n - 8, 1); }
        else if (n >= 16 && n < 20) { if (reads) rdB(sbn, (n - 16) >> 1, 0, (n - 16) & 1); }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto epilogue = [&](int n) __attribute__((always_inline)) {
      WsTile q;
      float sa_, sb_, unscale;
      tile_at(n, q, sa_, sb_, unscale);
      // buffer stores: per-lane byte offset in a VGPR, the row in a scalar offset (no 64-bit address arithmetic per store)
      const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(dX + (size_t)q.m * B * P, 0, (int)((unsigned)B * P * 4u), 0x00020000);
      const unsigned vo = ((unsigned)(4 * h) * P + q.tj * kWsTN + wc * 64 + l31) * 4u;
      const unsigned so = (unsigned)(q.ti * kWsTM + wr * 128) * P * 4u;
      // one block of 16 registers at a time: 16 multiplies into 16 DIFFERENT temporaries, then the 16 stores (a single
      // temporary would make every multiply wait until the previous store has read it: measured 60 cycles per store)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 tmp;
#pragma unroll
          for (int r = 0; r < 16; ++r) tmp[r] = acc[i][j][r] * unscale;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(tmp[r]), rsO, (int)(vo + j * 128u),
                                                  (int)(so + (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * P * 4u), 0);
          __builtin_amdgcn_sched_barrier(0);
        }
    };

    ws_barrier();                                  // b_0: k-tiles 0 and 1 are staged
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) rdA(smem, blk, 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) rdB(smem, r >> 1, 0, r & 1);
    int kt = 0, tile = 0, st = 0;                                   // position of k-tile g inside its output tile; its LDS stage
    for (int g = 0; g < G; ++g) {
      const int stn = st + 1 == kWsStages ? 0 : st + 1;
      WS_T0();
      if (kt == 0) ktile(std::true_type{}, smem + st * kWsStage, g + 1 < G, smem + stn * kWsStage);
      else ktile(std::false_type{}, smem + st * kWsStage, g + 1 < G, smem + stn * kWsStage);
      WS_ACC(t_work);
#ifdef WS_ABL_NOEPI
      if (++kt == KT) { kt = 0; if (acc[0][0][0] == 123.456f) epilogue(tile++); }
#else
      if (++kt == KT) { WS_T0(); epilogue(tile++); kt = 0; WS_ACC(t_epi); }
#endif
      st = stn;
      WS_T0();
      ws_barrier();
      WS_ACC(t_bar);
    }
#ifdef WS_DBG
    if (dbg && lane == 0) {
      unsigned long long* d = dbg + ((size_t)blockIdx.x * 8 + wave) * 4;
      d[0] = clock64() - t_all; d[1] = t_work; d[2] = t_bar; d[3] = t_epi;
    }
#endif
  } else {
    // ------------------------------------------------------------------ producers
    const int pt = t - 256, pw = pt >> 6;
#ifndef WS_NO_PRIO
    __builtin_amdgcn_s_setprio(3);     // the producers' short instructions go ahead of the consumers' queued MFMAs
#endif
    const int kq = pt & 3, srow = pt >> 2;           // A: quad kq of rows srow + 64 i
    const int r8 = lane >> 3, c8 = lane & 7;         // X: k-row r8 (+ 8 half), columns 4 c8 .. + 3 of a panel
    f32x4 ra[3][4], rb[3][2];
    // per-thread constant offsets
    unsigned aoff[4], boff[2];
    int awr[4], bwr[2];                              // LDS byte offsets inside a stage
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      aoff[i] = ((unsigned)(srow + 64 * i) * B + 4 * kq) * 4u;
      awr[i] = (srow + 64 * i) * 48 + 8 * kq;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int v = pw * 2 + u, panel = v >> 1, half = v & 1;
      boff[u] = ((unsigned)(8 * half + r8) * P + 32 * panel + 4 * c8) * 4u;
      bwr[u] = 2 * kWsAPlane + (panel * 16 + 8 * half + r8) * 64 + 8 * c8;
    }
    // state of the load stream (k-tile gl = next to load) and of the staging stream
    int l_tile = 0, l_kt = 0;
    __amdgpu_buffer_rsrc_t rsA, rsB;
    unsigned sA = 0, sB = 0;                          // scalar byte offsets of the current load position
    float l_sa = 1.f, l_sb = 1.f;
    float s_sa[3], s_sb[3];                           // scales that go with each register set
    auto open_tile = [&](int n) __attribute__((always_inline)) {
      WsTile q;
      float us_;
      tile_at(n < ntile ? n : ntile - 1, q, l_sa, l_sb, us_);
      rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W + (size_t)q.m * B * B), 0, (int)((unsigned)B * B * 4u), 0x00020000);
      rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X + (size_t)q.m * B * P), 0, (int)((unsigned)B * P * 4u), 0x00020000);
      sA = (unsigned)(q.ti * kWsTM) * B * 4u;
      sB = (unsigned)(q.tj * kWsTN) * 4u;
    };
    auto gload = [&](auto SET) __attribute__((always_inline)) {     // loads of the next k-tile of the stream into set S
      constexpr int S = decltype(SET)::value;
      if (l_kt == 0) open_tile(l_tile);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        ra[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)aoff[i], (int)(sA + (unsigned)l_kt * 64u), 0));
#pragma unroll
      for (int u = 0; u < 2; ++u)
        rb[S][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)boff[u], (int)(sB + (unsigned)l_kt * 16u * P * 4u), 0));
      s_sa[S] = l_sa;
      s_sb[S] = l_sb;
      if (++l_kt == KT) { l_kt = 0; ++l_tile; }
    };
    auto stage = [&](auto SET, lds_u8* sb) __attribute__((always_inline)) {   // split + LDS write of set S into stage sb
      constexpr int S = decltype(SET)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u32x2 p[2];
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ra[S][i][e] * s_sa[S];
        split2h(v, 0, p[0], p[1], 0);
        split2h(v, 2, p[0], p[1], 1);
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(sb + awr[i]) = p[0];
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(sb + awr[i] + kWsAPlane) = p[1];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        u32x2 p[2];
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rb[S][u][e] * s_sb[S];
        split2h(v, 0, p[0], p[1], 0);
        split2h(v, 2, p[0], p[1], 1);
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(sb + bwr[u]) = p[0];
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(sb + bwr[u] + kWsBPlane) = p[1];
      }
    };
    // The stream is UNCONDITIONAL (no branch around a load or a staging step: hipcc then counts the loads in flight
    // exactly and waits with vmcnt(12+), i.e. only for the oldest of three k-tiles): past the end of the stream it
    // re-reads the last output tile and stages into LDS stages no consumer reads any more.
    // prologue (G >= 16): loads of k-tiles 0..2, stage 0 and 1 (their sets take the loads of 3 and 4), barrier b_0
    gload(I0{});
    gload(I1{});
    gload(I2{});
    stage(I0{}, smem);
    gload(I0{});
    stage(I1{}, smem + kWsStage);
    gload(I1{});
    ws_barrier();                                  // b_0
    // iteration g: stage k-tile g + 2 (register set and LDS stage (g + 2) % 3), load k-tile g + 5 into the same set
    auto iter = [&](auto SET) __attribute__((always_inline)) {
      constexpr int S = decltype(SET)::value;
      WS_T0();
#ifndef WS_ABL_NOSTAGE
      stage(SET, smem + S * kWsStage);
#else
      if (ra[S][0][0] == 123.456f) stage(SET, smem + S * kWsStage);
#endif
      WS_ACC(t_work);
      WS_T0();
#ifndef WS_ABL_NOLOAD
      gload(SET);
#endif
      WS_ACC(t_epi);
      WS_T0();
      ws_barrier();
      WS_ACC(t_bar);
    };
    int g = 0;
    for (; g + 3 <= G; g += 3) {          // (g + 2) % 3 == 2, 0, 1 for g % 3 == 0, 1, 2
      iter(I2{});
      iter(I0{});
      iter(I1{});
    }
    if (g < G) { iter(I2{}); ++g; }
    if (g < G) { iter(I0{}); ++g; }
#ifdef WS_DBG
    if (dbg && lane == 0) {
      unsigned long long* d = dbg + ((size_t)blockIdx.x * 8 + wave) * 4;
      d[0] = clock64() - t_all; d[1] = t_work; d[2] = t_bar; d[3] = t_epi;
    }
#endif
  }
}

}  // namespace hecka
