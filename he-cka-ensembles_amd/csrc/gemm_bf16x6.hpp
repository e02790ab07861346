// fp32 GEMM on the bf16 matrix cores ("bf16x6"): every fp32 operand is split EXACTLY into three bf16 terms
//     x = h + m + l,   h = trunc16(x), m = trunc16(x - h), l = trunc16(x - h - m)          (24 significand bits)
// and a.b is accumulated in fp32 from the six leading products  hh + hm + mh + mm + hl + lh  (the three dropped
// ones are below 2^-23 |a||b|).  On gfx950 v_mfma_f32_32x32x16_bf16 retires 16 k per 32 cycles against 2 k per
// 64 cycles for v_mfma_f32_32x32x2_f32, so six bf16 passes cost 12 cycles per k instead of 32: 2.67x the
// fp32-MFMA rate.
//
// Accuracy.  The bf16 MFMA adds its 16 products and the accumulator with TRUNCATION (measured: bias -4e-6 of
// sum|a.b| at K = 8192 when one accumulator runs over the whole K).  The tile therefore accumulates at most
// kX6Fold k-tiles (64 k) in a LOCAL accumulator that starts from zero, and folds it into the running sum with
// IEEE round-to-nearest v_add_f32: the truncation then acts on 1/128 of the magnitude and the running sum sees
// K/64 correctly rounded adds.  Measured max error / sum|a.b| at K = 8192: 6.5e-7 (fp32 MFMA path: 2.5e-6).
//
// Structure: 256 threads (4 waves, 2x2), tile 128x128, BK = 16, two LDS stages (2 x 36 KB, two workgroups per
// CU).  The split is done ONCE per staged element (registers -> three bf16 planes in LDS); LDS rows are 16 bf16
// + 8 pad (48 B: conflict-free ds_read_b128).  Global loads run two k-tiles ahead in two register sets; split +
// LDS write of tile t+1 is interleaved with the 24 MFMAs of tile t; one barrier per tile.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "gemm_mfma.hpp"

namespace hecka {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kX6BK = 16;
constexpr int kX6LD = 24;   // bf16 per padded LDS row (48 B)
constexpr int kX6Fold = 4;  // k-tiles per local accumulation chain
#ifndef HECKA_X3_LATE_L
#define HECKA_X3_LATE_L 1
#endif
#ifndef HECKA_X3_PREFETCH
#define HECKA_X3_PREFETCH 1
#endif

// SCHEME 0 "bf16x6": three bf16 planes, six products.  SCHEME 1 "f16x3": x*2^e = h + l with two fp16 terms
// (22 significand bits), three products hh + hl + lh; half the MFMA work and 2/3 of the LDS traffic, but the
// operand must be pre-scaled into fp16 range by the caller's prologue (Pro::scaleA / scaleB, powers of two).
template <int TM, int TN, int NP>
struct X6Smem {
  alignas(16) unsigned short a[2][NP][TM * kX6LD];
  alignas(16) unsigned short b[2][NP][TN * kX6LD];
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// fp16 two-term split of two floats (already scaled into range): h = rne16(x), l = rne16(x - h); x - h is exact in
// fp32 and |x - h - l| <= 2^-24 |x| (v_cvt_pk_f16_f32 rounds to nearest: no truncation bias)
template <class V>
__device__ __forceinline__ void split2h(const f32x4& v, int q0, V& vh, V& vl, int qo) {
  const f32x2 x = {v[q0], v[q0 + 1]};
  const f16x2 h = __builtin_convertvector(x, f16x2);
  const f32x2 r = {x[0] - (float)h[0], x[1] - (float)h[1]};
  const f16x2 l = __builtin_convertvector(r, f16x2);
  vh[qo] = __builtin_bit_cast(unsigned, h);
  vl[qo] = __builtin_bit_cast(unsigned, l);
}

// Operand element kinds.  16-bit operands (bf16 / fp16 features) are loaded as they lie in HBM (half the bytes) and
// widened to fp32 in registers right before the prologue; everything behind that is the fp32 pipeline.
constexpr int kElemF32 = 0, kElemBF16 = 1, kElemF16 = 2;
template <int E> constexpr int elem_bytes() { return E == kElemF32 ? 4 : 2; }
template <int E>
__device__ __forceinline__ float widen16(unsigned u) {   // low 16 bits of u
  if (E == kElemBF16) return __uint_as_float(u << 16);
  return (float)__builtin_bit_cast(_Float16, (unsigned short)u);
}
// A staged quad of a 16-bit operand sits in the fp32 register set as raw bits: two packed dwords in v[0], v[1] (k
// contiguous loads) or four zero-extended halves (GATHER: the transposing loads of an NN operand).
template <int E, bool GATHER>
__device__ __forceinline__ void widen_quad(f32x4& v) {
  if (E == kElemF32) return;
  if (GATHER) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = widen16<E>(__float_as_uint(v[e]));
  } else {
    const unsigned u0 = __float_as_uint(v[0]), u1 = __float_as_uint(v[1]);
    if (E == kElemBF16) {
      v = f32x4{__uint_as_float(u0 << 16), __uint_as_float(u0 & 0xFFFF0000u), __uint_as_float(u1 << 16),
                __uint_as_float(u1 & 0xFFFF0000u)};
    } else {
      v = f32x4{widen16<E>(u0), widen16<E>(u0 >> 16), widen16<E>(u1), widen16<E>(u1 >> 16)};
    }
  }
}

// power-of-two scale s with bound * s <= 2^headroom (exact to apply and to undo); 1 for zero / non-finite bounds
__device__ __forceinline__ float pow2_scale(float bound, int headroom) {
  const unsigned b = __float_as_uint(bound) & 0x7FFFFFFFu;
  const int be = (int)(b >> 23);
  if (be == 0 || be == 255) return 1.0f;
  int k = headroom - (be - 126);      // bound < 2^(be-126)
  k = k < -60 ? -60 : (k > 60 ? 60 : k);
  return __uint_as_float((unsigned)(127 + k) << 23);
}

// two of the four floats of v (elements q0, q0+1) -> one packed dword per plane
template <class V>
__device__ __forceinline__ void split2(const f32x4& v, int q0, V& vh, V& vm, V& vl, int qo) {
  // plain (unpacked) f32 subtracts: packed-f32 VALU ops beside MFMAs cost several times their issue slot
  const float x0 = v[q0], x1 = v[q0 + 1];
  const unsigned b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
  const float r0 = x0 - __uint_as_float(b0 & 0xFFFF0000u), r1 = x1 - __uint_as_float(b1 & 0xFFFF0000u);   // exact
  const unsigned a0 = __float_as_uint(r0), a1 = __float_as_uint(r1);
  const float t0 = r0 - __uint_as_float(a0 & 0xFFFF0000u), t1 = r1 - __uint_as_float(a1 & 0xFFFF0000u);   // exact
  vh[qo] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);   // high halves of (x1, x0)
  vm[qo] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
  vl[qo] = __builtin_amdgcn_perm(__float_as_uint(t1), __float_as_uint(t0), 0x07060302u);
}

// A prologue may supply the NN operand IMPLICITLY (Pro::kImplicitB = true): instead of row k of a [K, cols] matrix the
// tile asks pro.bvoff(k) for the byte offset of element (k, this thread's column) in the buffer rsB is made over - an
// out-of-range offset reads as zero (buffer bounds check).  The convolution kernels gather image patches this way.
template <class P, class = void>
struct implicit_b : std::false_type {};
template <class P>
struct implicit_b<P, std::void_t<decltype(P::kImplicitB)>> : std::bool_constant<P::kImplicitB> {};

// TAP-MAJOR implicit NN operand (Pro::kImplicitBT = true): the K axis is ordered (tap, channel) and a k-tile holds 16
// channels of ONE tap, so whether the tap falls inside the image is decided once per k-tile and thread - pro.tile(ktile, vo,
// so) gives the thread's byte offset (out of range -> a bounds-checked offset that reads as zero) and the wave-uniform
// byte offset of the tile's first channel; channel e of the tile is pro.hw4 * e further on, in the SCALAR offset of the
// load: no vector instruction per gathered element.
template <class P, class = void>
struct implicit_bt : std::false_type {};
template <class P>
struct implicit_bt<P, std::void_t<decltype(P::kImplicitBT)>> : std::bool_constant<P::kImplicitBT> {};

// Likewise for an NT product (Pro::kImplicitNT = true): both operands come from pro - pro.seek(kt) positions it on k-tile
// kt (calls come with kt increasing by one), pro.avoff(i) is the byte offset of this thread's A quad of row block i,
// pro.btoff(i, c) that of element c of its B quad (element-wise, so that a quad may straddle an image border).
template <class P, class = void>
struct implicit_nt : std::false_type {};
template <class P>
struct implicit_nt<P, std::void_t<decltype(P::kImplicitNT)>> : std::bool_constant<P::kImplicitNT> {};
// ... whose B quad is four CONSECUTIVE elements in memory (Pro::kQuadB = true): ONE 16-byte load at pro.btoffq(SET, i); the
// prologue keeps what to zero of it (elements pushed over an image row's edge) per register set and applies it in stageB.
template <class P, class = void>
struct quad_nt : std::false_type {};
template <class P>
struct quad_nt<P, std::void_t<decltype(P::kQuadB)>> : std::bool_constant<P::kQuadB> {};

// Product tile out[i,j] = sum_k A[i,k] B(k,j).  RAGGED = false: interior tiles only (rows/cols multiples of the tile,
// (kend - kbeg) a multiple of 16, 16-byte aligned A rows).  RAGGED = true: any row / column count and any k range that is
// a multiple of 4 for the k-contiguous operands (whole quads; row starts stay 16-byte aligned) - the whole address of a
// load sits in the bounds-checked VGPR offset, so rows past the operand's end read as zero, quads past kend are sent
// out of range on purpose, and the caller masks the rows / columns of the result that do not exist.  Operands below
// 2 GiB.  (The reference's own batch sizes - 300, 100, 400 - are ragged.)  A is [rows, K] (k contiguous).  NN = false: B is [cols, K] (k contiguous,
// 16-byte aligned rows); NN = true: B is [K, cols] (cols contiguous) and is transposed on the way into LDS - a thread
// gathers 8 k of ONE column with dword loads (a wave-load = 256 contiguous bytes of one k-row).
// Pro: transform of the staged fp32 values (centring), NT only.
// FOLD = false: one accumulation chain over the whole K (no local chains, 64 fewer VGPRs: a third workgroup per CU);
// only for short K - the 16-bit MFMA's truncation bias grows with the chain (about -5e-10 of sum|a.b| per k).
// EA / EB: element kinds of the operands in HBM.  BLOW = false (f16x3 only): the staged B values are exact in ONE fp16
// term (16-bit features times a power of two), so B has no low plane and the product list is hh + lh.
// ALOW = false: the same for A (both false: one product, hh).
template <int TM, int TN, bool NN, class Pro, int SCHEME = 0, bool FOLD = true, int EA = kElemF32, int EB = kElemF32,
          bool BLOW = true, bool ALOW = true, bool SAMEAB = false, bool RAGGED = false, bool ASLICED = false, int NT = kThreads>
struct GemmTileX6 {
  // NT threads: 256 (four waves, 2 x 2) or 512 (eight waves, 2 x 4, with TN = 256: a 256 x 256 tile stages a third
  // fewer operand bytes per flop than two 256 x 128 tiles and half the transposing gathers)
  static_assert(NT == 2 * TN, "one thread per column and k-half of the NN operand: NT = 2 TN");
  static constexpr int RPP = NT / 4;            // rows of a k-contiguous operand staged per pass
  static constexpr int WCOLS = NT / 128;        // wave grid: 2 rows x WCOLS columns
  // ASLICED: the rows of A lie in blocks of 2^a_rshift rows, block b of this member at element offset b * a_blk (the
  // member-sharded path keeps W in the order the all-to-all delivers it: [slice, member, rows of the slice, B])
  static_assert(!RAGGED || (!implicit_b<Pro>::value && !implicit_nt<Pro>::value && !implicit_bt<Pro>::value),
                "ragged tiles: explicit operands only");
  static constexpr unsigned kOOB = 0x80000000u;   // a byte offset no (< 2 GiB) operand reaches: the load returns zero
  // SAMEAB (NT, diagonal tile of a symmetric product): the B rows ARE the A rows - nothing is loaded, split or staged
  // for B, its fragments come from the A planes
  static_assert(!SAMEAB || (!NN && TM == TN), "SAMEAB: NT product with a square tile");
  static_assert((BLOW && ALOW) || SCHEME == 1, "single-plane operands: f16x3 only");
  static constexpr int SA = elem_bytes<EA>(), SB = elem_bytes<EB>();
  static constexpr int NP = SCHEME == 1 ? 2 : 3;
  static constexpr int WM = TM / 2, WN = TN / (NT / 128), MI = WM / 32, NI = WN / 32;
  static constexpr int CA = TM * 4 / NT, CB = TN * 4 / NT;  // 4-k quads staged per thread per tile
  static_assert(CA >= 1 && CB >= 1, "tile too small");
  using Smem = X6Smem<TM, TN, NP>;
  f32x16 acc[MI][NI];

  // Ab / Bb: this member's operands, bytesA / bytesB their sizes (< 4 GiB: offsets are 32-bit)
  __device__ __forceinline__ void run(Smem& sm, const void* Ab, unsigned bytesA, int lda, const void* Bb,
                                      unsigned bytesB, int ldb, int row0, int col0, int kbeg, int kend, Pro& pro,
                                      int a_rshift = 31, unsigned a_blk = 0) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave / (NT / 128), wc = wave % (NT / 128), l31 = lane & 31, h = lane >> 5;
    f32x16 loc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // staging map: four consecutive lanes cover the 64-byte k-tile segment of one row (one wave-load = 16 whole
    // segments); thread t stages quad kq = t & 3 of rows (t >> 2) + 64 i.  Buffer loads: the operand's base sits in
    // a resource descriptor, the tile/k position in a scalar byte offset, the thread's place in ONE 32-bit VGPR
    // offset per quad - no 64-bit address arithmetic in the loop.
    const int kq = t & 3, srow = t >> 2;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(Ab), 0, (int)bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(Bb), 0, (int)bytesB, 0x00020000);
    const unsigned sA0 = ((ASLICED ? 0u : (unsigned)row0 * lda) + kbeg) * (unsigned)SA;
    const unsigned sB0 = (NN ? ((unsigned)kbeg * ldb + col0) : ((unsigned)col0 * ldb + kbeg)) * (unsigned)SB;
    unsigned aoff[CA], bgoff[CB];
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      if (ASLICED) {
        const unsigned r = (unsigned)(row0 + srow + RPP * i);
        aoff[i] = ((r >> a_rshift) * a_blk + (r & ((1u << a_rshift) - 1u)) * lda + 4 * kq) * (unsigned)SA;
      } else {
        aoff[i] = ((unsigned)(srow + RPP * i) * lda + 4 * kq) * (unsigned)SA;
      }
    }
#pragma unroll
    for (int i = 0; i < CB; ++i)
      bgoff[i] = (NN ? (unsigned)(8 * (t / TN) + 4 * i) * ldb + (t % TN) : (unsigned)(srow + RPP * i) * ldb + 4 * kq) *
                 (unsigned)SB;
    const int klen = kend - kbeg;
    const int ntiles = RAGGED ? (klen + kX6BK - 1) / kX6BK : klen / kX6BK;
    if (ntiles <= 0) return;
    auto& smb = *[&]() {      // the planes the B fragments are read from
      if constexpr (SAMEAB) return &sm.a;
      else return &sm.b;
    }();

    f32x4 ra[2][CA], rb[2][CB];  // two register sets: tiles t+1 and t+2 in flight
    auto gload_a = [&](auto SET, int kt) __attribute__((always_inline)) {
      constexpr int S = decltype(SET)::value;
      const unsigned so = sA0 + (unsigned)kt * (unsigned)(kX6BK * SA);
      if constexpr (implicit_nt<Pro>::value) pro.seek(kt);
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        const unsigned vo = RAGGED ? (kt * kX6BK + 4 * kq < klen ? aoff[i] + so : kOOB) : aoff[i];
        const unsigned so_ = RAGGED ? 0u : so;
        if constexpr (implicit_nt<Pro>::value) {
          ra[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)pro.avoff(i), 0, 0));
        } else if (EA == kElemF32) {
          ra[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)vo, (int)so_, 0));
        } else {
          const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rsA, (int)vo, (int)so_, 0);
          ra[S][i][0] = __uint_as_float(r[0]);
          ra[S][i][1] = __uint_as_float(r[1]);
        }
      }
    };
    auto gload_b = [&](auto SET, int kt) __attribute__((always_inline)) {
      constexpr int S = decltype(SET)::value;
      pro.prefetch(SET, kbeg + kt * kX6BK + 4 * kq);
      if (SAMEAB) return;
      if constexpr (implicit_bt<Pro>::value) {
        static_assert(NN, "tap-major implicit operand: NN products");
        unsigned vo, so;
        pro.tile(__builtin_amdgcn_readfirstlane(kbeg / kX6BK + kt), vo, so);
        so += (unsigned)__builtin_amdgcn_readfirstlane(8 * (t / TN)) * pro.hw4;
#pragma unroll
        for (int i = 0; i < CB; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            rb[S][i][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsB, (int)vo, (int)(so + (unsigned)(4 * i + c) * pro.hw4), 0));
        return;
      }
#pragma unroll
      for (int i = 0; i < CB; ++i) {
        if (NN) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if constexpr (implicit_b<Pro>::value) {
              const int k = __builtin_amdgcn_readfirstlane(kbeg + kt * kX6BK + 8 * (t / TN) + 4 * i + c);   // wave-uniform
              rb[S][i][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsB, (int)pro.bvoff(k), 0, 0));
              continue;
            }
            int so = (int)(sB0 + (unsigned)(kt * kX6BK + c) * ldb * (unsigned)SB);
            unsigned vo = bgoff[i];
            if (RAGGED) {
              vo = kt * kX6BK + 8 * (t / TN) + 4 * i + c < klen ? bgoff[i] + (unsigned)so : kOOB;
              so = 0;
            }
            if (EB == kElemF32)
              rb[S][i][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsB, (int)vo, so, 0));
            else
              rb[S][i][c] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(rsB, (int)vo, so, 0));
          }
        } else if constexpr (quad_nt<Pro>::value) {
          rb[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)pro.btoffq(SET, i), 0, 0));
        } else if constexpr (implicit_nt<Pro>::value) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            rb[S][i][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsB, (int)pro.btoff(i, c), 0, 0));
        } else {
          int so = (int)(sB0 + (unsigned)kt * (unsigned)(kX6BK * SB));
          unsigned vo = bgoff[i];
          if (RAGGED) {
            vo = kt * kX6BK + 4 * kq < klen ? bgoff[i] + (unsigned)so : kOOB;
            so = 0;
          }
          if (EB == kElemF32) {
            rb[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)vo, so, 0));
          } else {
            const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rsB, (int)vo, so, 0);
            rb[S][i][0] = __uint_as_float(r[0]);
            rb[S][i][1] = __uint_as_float(r[1]);
          }
        }
      }
    };
    // bf16 offsets of this thread's quads in a plane
    const int soff = srow * kX6LD + 4 * kq;
    int boff[CB];
#pragma unroll
    for (int i = 0; i < CB; ++i) boff[i] = NN ? (t % TN) * kX6LD + 8 * (t / TN) + 4 * i : soff + RPP * i * kX6LD;
    auto stage_a = [&](auto SET, int st) __attribute__((always_inline)) {   // split + LDS write of the A quads
      constexpr int S = decltype(SET)::value;
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        u32x2 p[3];
        widen_quad<EA, false>(ra[S][i]);
        pro.stageA(SET, ra[S][i], i);
        if (SCHEME == 1) {
          split2h(ra[S][i], 0, p[0], p[1], 0);
          split2h(ra[S][i], 2, p[0], p[1], 1);
        } else {
          split2(ra[S][i], 0, p[0], p[1], p[2], 0);
          split2(ra[S][i], 2, p[0], p[1], p[2], 1);
        }
#pragma unroll
        for (int q = 0; q < (ALOW ? NP : 1); ++q) *reinterpret_cast<u32x2*>(&sm.a[st][q][soff + RPP * i * kX6LD]) = p[q];
      }
    };
    auto stage_b = [&](auto SET, int st) __attribute__((always_inline)) {
      constexpr int S = decltype(SET)::value;
      if (SAMEAB) return;
#pragma unroll
      for (int i = 0; i < CB; ++i) {
        u32x2 p[3];
        widen_quad<EB, NN>(rb[S][i]);
        pro.stageB(SET, rb[S][i], i);
        if (SCHEME == 1) {
          split2h(rb[S][i], 0, p[0], p[1], 0);
          split2h(rb[S][i], 2, p[0], p[1], 1);
        } else {
          split2(rb[S][i], 0, p[0], p[1], p[2], 0);
          split2(rb[S][i], 2, p[0], p[1], p[2], 1);
        }
#pragma unroll
        for (int q = 0; q < (BLOW ? NP : 1); ++q) *reinterpret_cast<u32x2*>(&sm.b[st][q][boff[i]]) = p[q];
      }
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    gload_a(S0{}, 0);
    gload_b(S0{}, 0);
    if (ntiles > 1) { gload_a(S1{}, 1); gload_b(S1{}, 1); }
    stage_a(S0{}, 0);
    stage_b(S0{}, 0);
    if (ntiles > 2) { gload_a(S0{}, 2); gload_b(S0{}, 2); }
    __syncthreads();
    // PREF (f16x3, both operands with two planes, at least four MFMA blocks per wave): the fragments live across k-tiles.
    // A.h sits in slot 0 and A.l in slot 1; B.h alternates between slot (kt & 1) and B.l takes the other one.  The high
    // planes of tile kt + 1 are read during the LAST product group of tile kt (into the registers of A.h and B.l, dead by
    // then), behind a barrier that sits in the middle of the k-tile instead of at its end: no wave starts a k-tile by
    // waiting for LDS, and the eight waves of a one-workgroup-per-CU tile no longer hit the LDS together.
    constexpr bool PREF = HECKA_X3_PREFETCH && SCHEME == 1 && ALOW && BLOW && (MI * NI >= 4) &&
                          !(TM == 256 && TN == 128);   // (the four-wave 256-row tile has no registers to spare: it would spill)
    f16x8 pfa[PREF ? MI : 1][2], pfb[PREF ? NI : 1][2];
    if constexpr (PREF) {
#pragma unroll
      for (int j = 0; j < NI; ++j)
        pfb[j][0] = *reinterpret_cast<const f16x8*>(&smb[0][0][(wc * WN + j * 32 + l31) * kX6LD + 8 * h]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
        pfa[i][0] = *reinterpret_cast<const f16x8*>(&sm.a[0][0][(wr * WM + i * 32 + l31) * kX6LD + 8 * h]);
    }

    // one k-tile: the 24 MFMAs of tile kt (LDS stage kt & 1), with the split + LDS write of tile kt+1 (register set
    // NSET) and the global loads of tile kt+3 spread one piece per MFMA gap (a piece = 10 VALU, three ds_write_b64
    // or the loads of one operand: about one 32-cycle MFMA).  FIRST: the local chain starts from zero.  GUARD: tail tiles, where
    // tile kt+1 / kt+3 may not exist; the steady-state instance has no branch inside.
    static_assert(SCHEME == 1 || (TM == 128 && TN == 128), "the bf16x6 schedule is written for the 128x128 tile");
    static_assert((TM == 256 || TM == 128 || TM == 64) && (TN == 128 || (TN == 256 && TM == 256)), "tile: 256, 128 or 64 rows by 128 columns, or 256 x 256");
    auto iter = [&](auto FIRST, auto NSET, auto GUARD, int kt) __attribute__((always_inline)) {
      constexpr int S = decltype(NSET)::value;
      constexpr bool G = decltype(GUARD)::value;
      const int st = kt & 1;
      const bool more = !G || kt + 1 < ntiles;
      const bool load = !G || kt + 3 < ntiles;
      // fragments: six 8-register groups (A/B x h/m/l) in FOUR register slots - each group is read one product
      // group (4 MFMAs) before its first use, into the slot of a group whose last MFMA has issued:
      //   products  lh hh mh mm hm hl ;  top: Al Bh Ah | gap 0: Am | gap 4: Bm (slot of Al) | gap 12: Bl (slot of Bh)
      bf16x8 fa[MI][3], fb[NI][3];
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
      constexpr int PB[6] = {0, 0, 0, 1, 1, 2};
      auto lda_ = [&](int p) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
          fa[i][p] = *reinterpret_cast<const bf16x8*>(&sm.a[st][p][(wr * WM + i * 32 + l31) * kX6LD + 8 * h]);
      };
      auto ldb_ = [&](int p) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
          fb[j][p] = *reinterpret_cast<const bf16x8*>(&smb[st][p][(wc * WN + j * 32 + l31) * kX6LD + 8 * h]);
      };
      lda_(2);
      ldb_(0);
      lda_(0);
      __builtin_amdgcn_sched_barrier(0);
      u32x2 qv[3];
#pragma unroll
      for (int n = 0; n < 24; ++n) {
        const int e = n >> 2, i = (n >> 1) & 1, j = n & 1;
        {
          f32x16 c = loc[i][j];
          if (decltype(FIRST)::value && e == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) c[q] = 0.f;
          }
          loc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA[e]], fb[j][PB[e]], c, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // gaps: 0 Am | 1-3 A quad 0 | 4 Bm | 5-7 A quad 1 | 8 A loads | 9-11 B quad 0 | 12 Bl | 13-15 B quad 1 | 16 B loads
        if (n == 0) lda_(1);
        else if (n == 4) ldb_(1);
        else if (n == 12) ldb_(2);
        else if (n == 8) { if (load) gload_a(NSET, kt + 3); }
        else if (n == 16) { if (load) gload_b(NSET, kt + 3); }
        else if (n < 16) {
          const bool isA = n < 8;
          const int quad = (n >> 2) & 1, step = (n & 3) - 1;
          f32x4& src = isA ? ra[S][quad] : rb[S][quad];
          if (more && (isA || !SAMEAB)) {
            if (step == 0) {
              if (isA) { widen_quad<EA, false>(src); pro.stageA(NSET, src, quad); }
              else { widen_quad<EB, NN>(src); pro.stageB(NSET, src, quad); }
              split2(src, 0, qv[0], qv[1], qv[2], 0);
            } else if (step == 1) {
              split2(src, 2, qv[0], qv[1], qv[2], 1);
            } else {
#pragma unroll
              for (int q = 0; q < 3; ++q) {
                if (isA) *reinterpret_cast<u32x2*>(&sm.a[st ^ 1][q][soff + RPP * quad * kX6LD]) = qv[q];
                else *reinterpret_cast<u32x2*>(&sm.b[st ^ 1][q][boff[quad]]) = qv[q];
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    };
    // f16x3 k-tile: 3 MI NI MFMAs (hh, hl, lh on the wave's 32x32 tiles; 12 for the 128x128 tile); all fragment groups are
    // read at the top; the staging work of the next tile goes one item per MFMA gap, in this order:
    //   A quads (split, write each) | A loads | B quads (split, write each) | B loads
    // (128x128: gaps 0-3 A, 4 A loads, 5-8 B, 9 B loads; a 64-row tile has 6 MFMAs and 8 items - the last two follow
    // the last MFMA)
    auto iter16 = [&](auto FIRST, auto NSET, auto GUARD, int kt) __attribute__((always_inline)) {
      constexpr int S = decltype(NSET)::value;
      constexpr bool G = decltype(GUARD)::value;
      constexpr int NM = 3 * MI * NI, NITEMS = 2 * CA + 1 + 2 * CB + 1;
      const int st = kt & 1;
      const bool more = !G || kt + 1 < ntiles;
      const bool load = !G || kt + 3 < ntiles;
      f16x8 fa[MI][2], fb[NI][2];
      auto rd_a = [&](int p) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
          fa[i][p] = *reinterpret_cast<const f16x8*>(&sm.a[st][p][(wr * WM + i * 32 + l31) * kX6LD + 8 * h]);
      };
      auto rd_b = [&](int p) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
          fb[j][p] = *reinterpret_cast<const f16x8*>(&smb[st][p][(wc * WN + j * 32 + l31) * kX6LD + 8 * h]);
      };
      // HECKA_X3_LATE_L: only the high planes are read before the first MFMA; the low planes (first used by the second
      // and third product group) are read in the first two MFMA gaps - half the fragment-read burst at the top of the
      // k-tile overlaps the hh products
      constexpr bool LATE = HECKA_X3_LATE_L && (MI * NI >= 4);
      rd_b(0);      // (B first: the first MFMA then waits for NI + 1 reads, not MI + NI)
      rd_a(0);
      if (!LATE) {
        if (ALOW) rd_a(1);
        if (BLOW) rd_b(1);
      }
      __builtin_amdgcn_sched_barrier(0);
      constexpr int PA[3] = {0, 0, 1};
      constexpr int PB[3] = {0, 1, 0};
      u32x2 qv[2];
#pragma unroll
      for (int n = 0; n < (NM > NITEMS ? NM : NITEMS); ++n) {
        if (n < NM) {
          const int e = n / (MI * NI), i = (n % (MI * NI)) / NI, j = n % NI;
          if (e == 0 || (e == 1 && BLOW) || (e == 2 && ALOW)) {   // e = 1 is A.h x B.l, e = 2 is A.l x B.h
            if (FOLD) {
              f32x16 c = loc[i][j];
              if (decltype(FIRST)::value && e == 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) c[q] = 0.f;
              }
              loc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][PA[e]], fb[j][PB[e]], c, 0, 0, 0);
            } else {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][PA[e]], fb[j][PB[e]], acc[i][j], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (LATE && n == 0 && BLOW) rd_b(1);
        if (LATE && n == 1 && ALOW) rd_a(1);
        if (n == 2 * CA) { if (load) gload_a(NSET, kt + 3); }
        else if (n == 2 * CA + 1 + 2 * CB) { if (load) gload_b(NSET, kt + 3); }
        else if (n < NITEMS) {
          const bool isA = n < 2 * CA;
          const int r = isA ? n : n - (2 * CA + 1), quad = r >> 1, step = r & 1;
          f32x4& src = isA ? ra[S][quad] : rb[S][quad];
          if (more && (isA || !SAMEAB)) {
            if (step == 0) {
              if (isA) { widen_quad<EA, false>(src); pro.stageA(NSET, src, quad); }
              else { widen_quad<EB, NN>(src); pro.stageB(NSET, src, quad); }
              split2h(src, 0, qv[0], qv[1], 0);
              split2h(src, 2, qv[0], qv[1], 1);
            } else {
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                if (isA) { if (q == 0 || ALOW) *reinterpret_cast<u32x2*>(&sm.a[st ^ 1][q][soff + RPP * quad * kX6LD]) = qv[q]; }
                else if (q == 0 || BLOW) *reinterpret_cast<u32x2*>(&sm.b[st ^ 1][q][boff[quad]]) = qv[q];
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    };
    auto iter16p = [&](auto FIRST, auto NSET, auto GUARD, int kt) __attribute__((always_inline)) {
      constexpr int S = decltype(NSET)::value;
      constexpr bool G = decltype(GUARD)::value;
      constexpr int NM = 3 * MI * NI, NITEMS = 2 * CA + 1 + 2 * CB + 1;
      constexpr int BH = S == 1 ? 0 : 1, BL = 1 - BH;      // NSET is S1 on even k-tiles: B.h in slot kt & 1
      constexpr int BAR = (2 * MI * NI - 1 > NITEMS - 1) ? 2 * MI * NI - 1 : NITEMS - 1;   // gap that holds the barrier
      constexpr int NG = (NM > BAR + 3) ? NM : BAR + 3;
      const int st = kt & 1;
      const bool more = !G || kt + 1 < ntiles;
      const bool load = !G || kt + 3 < ntiles;
      u32x2 qv[2];
#pragma unroll
      for (int n = 0; n < NG; ++n) {
        if (n < NM) {
          const int e = n / (MI * NI), i = (n % (MI * NI)) / NI, j = n % NI;
          const f16x8 av = pfa[i][e == 2 ? 1 : 0], bv = pfb[j][e == 1 ? BL : BH];   // hh, A.h x B.l, A.l x B.h
          if (FOLD) {
            f32x16 c = loc[i][j];
            if (decltype(FIRST)::value && e == 0) {
#pragma unroll
              for (int q = 0; q < 16; ++q) c[q] = 0.f;
            }
            loc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[i][j], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (n == 0) {
#pragma unroll
          for (int j = 0; j < NI; ++j)
            pfb[j][BL] = *reinterpret_cast<const f16x8*>(&smb[st][1][(wc * WN + j * 32 + l31) * kX6LD + 8 * h]);
        }
        if (n == 1) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
            pfa[i][1] = *reinterpret_cast<const f16x8*>(&sm.a[st][1][(wr * WM + i * 32 + l31) * kX6LD + 8 * h]);
        }
        if (n == 2 * CA) { if (load) gload_a(NSET, kt + 3); }
        else if (n == 2 * CA + 1 + 2 * CB) { if (load) gload_b(NSET, kt + 3); }
        else if (n < NITEMS) {
          const bool isA = n < 2 * CA;
          const int r = isA ? n : n - (2 * CA + 1), quad = r >> 1, step = r & 1;
          f32x4& src = isA ? ra[S][quad] : rb[S][quad];
          if (more && (isA || !SAMEAB)) {
            if (step == 0) {
              if (isA) { widen_quad<EA, false>(src); pro.stageA(NSET, src, quad); }
              else { widen_quad<EB, NN>(src); pro.stageB(NSET, src, quad); }
              split2h(src, 0, qv[0], qv[1], 0);
              split2h(src, 2, qv[0], qv[1], 1);
            } else {
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                if (isA) *reinterpret_cast<u32x2*>(&sm.a[st ^ 1][q][soff + RPP * quad * kX6LD]) = qv[q];
                else *reinterpret_cast<u32x2*>(&sm.b[st ^ 1][q][boff[quad]]) = qv[q];
              }
            }
          }
        }
        if (n == BAR) __syncthreads();      // tile kt + 1 is complete in stage st ^ 1; stage st has no reader left
        if (n == BAR + 1 && more) {
#pragma unroll
          for (int j = 0; j < NI; ++j)
            pfb[j][BL] = *reinterpret_cast<const f16x8*>(&smb[st ^ 1][0][(wc * WN + j * 32 + l31) * kX6LD + 8 * h]);
        }
        if (n == BAR + 2 && more) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
            pfa[i][0] = *reinterpret_cast<const f16x8*>(&sm.a[st ^ 1][0][(wr * WM + i * 32 + l31) * kX6LD + 8 * h]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto fold = [&]() __attribute__((always_inline)) {
      if (!FOLD) return;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[i][j][q] += loc[i][j][q];   // v_add_f32, round to nearest
    };
    std::true_type T;
    std::false_type F;
    auto step = [&](auto FIRST, auto NSET, auto GUARD, int kt_) __attribute__((always_inline)) {
      if constexpr (PREF) iter16p(FIRST, NSET, GUARD, kt_);
      else if constexpr (SCHEME == 1) iter16(FIRST, NSET, GUARD, kt_);
      else iter(FIRST, NSET, GUARD, kt_);
    };
    int kt = 0;
    for (; kt + 7 <= ntiles; kt += 4) {   // steady state: tiles kt .. kt+3 all have a tile kt+3 behind them
      step(T, S1{}, F, kt);
      step(F, S0{}, F, kt + 1);
      step(F, S1{}, F, kt + 2);
      step(F, S0{}, F, kt + 3);
      fold();
    }
    while (kt < ntiles) {
      step(T, S1{}, T, kt++);
      if (kt < ntiles) step(F, S0{}, T, kt++);
      if (kt < ntiles) step(F, S1{}, T, kt++);
      if (kt < ntiles) step(F, S0{}, T, kt++);
      fold();
    }
  }

  template <class Fn>
  __device__ __forceinline__ void for_each(Fn&& f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / (NT / 128), wc = wave % (NT / 128), l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q)
          f(wr * WM + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h, wc * WN + j * 32 + l31, acc[i][j][q]);
  }
};

struct NoPrologueX6 {
  template <class S> __device__ __forceinline__ void prefetch(S, int) {}
  template <class S> __device__ __forceinline__ void stageA(S, f32x4&, int) {}
  template <class S> __device__ __forceinline__ void stageB(S, f32x4&, int) {}
};

}  // namespace hecka
