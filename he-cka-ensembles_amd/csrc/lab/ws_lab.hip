// Lab: wave-specialised dX = W X (gemm_ws.hpp) against the tile core (GemmTileX6<256 | 128, 128, NN, f16x3, no fold>):
// bit-equality of the results and speed.   ./ws_lab P [M] [iters]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "gemm_ws.hpp"
using namespace hecka;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) line %d\n", (int)e_, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct ScalePro {
  float sa, sb;
  template <class S> __device__ __forceinline__ void prefetch(S, int) {}
  template <class S> __device__ __forceinline__ void stageA(S, f32x4& v, int) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= sa;
  }
  template <class S> __device__ __forceinline__ void stageB(S, f32x4& v, int) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= sb;
  }
};

template <int TM>
__global__ __launch_bounds__(kThreads, TM == 256 ? 2 : 3) void tile_kernel(const float* W, const float* X, const unsigned* wmax,
                                                                           const float* xmax, float* dX, int B, int P,
                                                                           int tilesM, int tilesN) {
  using Tile = GemmTileX6<TM, 128, true, ScalePro, 1, false>;
  __shared__ typename Tile::Smem sm;
  const int w = blockIdx.x, per = tilesM * tilesN, slot = w >> 3;
  const int m = (slot / per) * 8 + (w & 7), tl = slot % per;
  const int ti = tl % tilesM, tj = tl / tilesM;
  ScalePro pro;
  pro.sa = pow2_scale(__uint_as_float(wmax[m]), 14);
  pro.sb = pow2_scale(xmax[m], 14);
  const float unscale = 1.0f / (pro.sa * pro.sb);
  Tile tile;
  tile.run(sm, W + (size_t)m * B * B, (unsigned)B * B * 4u, B, X + (size_t)m * B * P, (unsigned)B * P * 4u, P, ti * TM, tj * 128, 0, B, pro);
  tile.for_each([&](int lr, int lc, float v) { dX[(size_t)m * B * P + (size_t)(ti * TM + lr) * P + tj * 128 + lc] = v * unscale; });
}

__global__ __launch_bounds__(kWsThreads, 1) void ws_kernel(const float* W, const float* X, const unsigned* wmax, const float* xmax,
                                                          float* dX, int M, int B, int P, int tilesM, int tilesN, int xcd_map,
                                                          unsigned long long* dbg) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[kWsSmemTotal];
  ws_gram_bwd_body((lds_u8*)smem, W, X, wmax, xmax, dX, M, B, P, tilesM, tilesN, xcd_map != 0, dbg);
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 2048, M = argc > 2 ? atoi(argv[2]) : 64, iters = argc > 3 ? atoi(argv[3]) : 10, B = 512;
  const int nwg = argc > 4 ? atoi(argv[4]) : 256;
  float *W, *X, *o1, *o2, *xmax;
  unsigned* wmax;
  CK(hipMalloc(&W, (size_t)M * B * B * 4)); CK(hipMalloc(&X, (size_t)M * B * P * 4));
  CK(hipMalloc(&o1, (size_t)M * B * P * 4)); CK(hipMalloc(&o2, (size_t)M * B * P * 4));
  CK(hipMalloc(&xmax, M * 4)); CK(hipMalloc(&wmax, M * 4));
  std::vector<float> hw((size_t)M * B * B), hx((size_t)M * B * P), hxm(M), hwm(M);
  srand(5);
  for (int m = 0; m < M; ++m) {
    float mw = 0, mx = 0;
    const float sc = powf(3.f, (float)(m % 7) - 3.f);
    for (size_t i = 0; i < (size_t)B * B; ++i) { float u = (rand() % 20001) / 10000.f - 1.f; float v = sc * u * u * u; hw[(size_t)m * B * B + i] = v; mw = fmaxf(mw, fabsf(v)); }
    for (size_t i = 0; i < (size_t)B * P; ++i) { float u = (rand() % 20001) / 10000.f - 1.f; float v = u + 0.25f; hx[(size_t)m * B * P + i] = v; mx = fmaxf(mx, fabsf(v)); }
    hwm[m] = mw; hxm[m] = mx;
  }
  CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(xmax, hxm.data(), M * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(wmax, hwm.data(), M * 4, hipMemcpyHostToDevice));
  CK(hipMemset(o1, 0xff, (size_t)M * B * P * 4)); CK(hipMemset(o2, 0xff, (size_t)M * B * P * 4));
  const int tilesN = P / 128;
  unsigned long long* dbg; CK(hipMalloc(&dbg, 256 * 8 * 4 * 8)); CK(hipMemset(dbg, 0, 256 * 8 * 4 * 8));
  auto run_tile256 = [&]() { hipLaunchKernelGGL(tile_kernel<256>, dim3(M * (B / 256) * tilesN), dim3(kThreads), 0, 0, W, X, wmax, xmax, o1, B, P, B / 256, tilesN); };
  auto run_tile128 = [&]() { hipLaunchKernelGGL(tile_kernel<128>, dim3(M * (B / 128) * tilesN), dim3(kThreads), 0, 0, W, X, wmax, xmax, o1, B, P, B / 128, tilesN); };
  auto run_ws = [&]() { hipLaunchKernelGGL(ws_kernel, dim3(nwg), dim3(kWsThreads), 0, 0, W, X, wmax, xmax, o2, M, B, P, B / 256, tilesN, (int)(M % 8 == 0 && nwg % 8 == 0), dbg); };
  run_tile256(); CK(hipDeviceSynchronize());
  run_ws(); CK(hipDeviceSynchronize());
  std::vector<float> r1((size_t)M * B * P), r2((size_t)M * B * P);
  CK(hipMemcpy(r1.data(), o1, r1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), o2, r2.size() * 4, hipMemcpyDeviceToHost));
  double maxd = 0, maxv = 0; size_t nbad = 0, first = (size_t)-1;
  for (size_t i = 0; i < r1.size(); ++i) {
    const double d = fabs((double)r1[i] - (double)r2[i]);
    if (!(d == 0)) { ++nbad; if (first == (size_t)-1) first = i; }
    if (d > maxd) maxd = d;
    if (fabs(r1[i]) > maxv) maxv = fabs(r1[i]);
  }
  printf("ws vs tile256: max |diff| %.3g of max %.3g, %zu differing elements", maxd, maxv, nbad);
  if (nbad) printf(" (first at m %zu row %zu col %zu: %g vs %g)", first / ((size_t)B * P), (first / P) % B, first % P, r1[first], r2[first]);
  printf("\n");
  // fp64 spot check of the ws result
  double worst = 0;
  for (int s = 0; s < 64; ++s) {
    const int m = rand() % M, r = rand() % B, c = rand() % P;
    double e = 0, sa = 0;
    for (int k = 0; k < B; ++k) { const double a = hw[(size_t)m * B * B + (size_t)r * B + k], b = hx[(size_t)m * B * P + (size_t)k * P + c]; e += a * b; sa += fabs(a * b); }
    worst = fmax(worst, fabs(e - r2[(size_t)m * B * P + (size_t)r * P + c]) / sa);
  }
  printf("ws vs fp64: worst |err| / sum|w x| over 64 samples %.3g\n", worst);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto fn) {
    fn(); fn(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    printf("%-10s P=%d M=%d: %.1f us  %.1f TFLOP/s (algorithmic)\n", name, P, M, ms * 1e3, 2.0 * M * B * B * P / ms / 1e9);
  };
  timeit("tile128", run_tile128);
  timeit("tile256", run_tile256);
  timeit("ws", run_ws);
  {
    std::vector<unsigned long long> hd(256 * 8 * 4);
    CK(hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost));
    for (int w : {0, 1, 4, 5}) {
      double a[4] = {0, 0, 0, 0};
      for (int b = 0; b < nwg; ++b) for (int k = 0; k < 4; ++k) a[k] += (double)hd[((size_t)b * 8 + w) * 4 + k] / nwg;
      printf("wave %d (%s): total %.0f  work %.0f  barrier %.0f  %s %.0f   [clock64 ticks, mean over workgroups]\n", w, w < 4 ? "consumer" : "producer",
             a[0], a[1], a[2], w < 4 ? "epilogue" : "loads-issue", a[3]);
    }
  }
  return 0;
}
