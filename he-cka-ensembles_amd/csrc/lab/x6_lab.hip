// Lab: accuracy and speed of the bf16x6 fp32-emulating GEMM (NT, out = A A^T per batch) vs the fp32-MFMA core.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "../gemm_bf16x6.hpp"
using namespace hecka;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

// bit 30 of P: every member reads member 0's data (L2-resident working set)
__device__ __forceinline__ int P_alias_mod(int P) { return (P >> 30) & 1 ? 1 : 1 << 20; }
template <bool XCDMAP, int SCHEME = 0>
__global__ __launch_bounds__(kThreads, 2) void x6_kernel(const float* X, int B, int P, float* out, unsigned long long* dbg) {
  using Tile = GemmTileX6<128, 128, false, NoPrologueX6, SCHEME>;
  __shared__ typename Tile::Smem sm;
  const int nt = B / 128;
  int tl = blockIdx.x, m = blockIdx.y;
  if (XCDMAP) {   // all tiles of a member on one XCD (workgroup w runs on XCD w % 8), dispatched back to back
    const int w = blockIdx.y * gridDim.x + blockIdx.x, xcd = w & 7, slot = w >> 3, per = nt * nt;
    m = (slot / per) * 8 + xcd;
    tl = slot % per;
  }
  const int ti = tl / nt, tj = tl % nt;
  const float* Xm = X + (size_t)(m % P_alias_mod(P)) * B * (P & 0xFFFFFF);
  NoPrologueX6 pro;
  Tile tile;
  const int Pm = P & 0xFFFFFF;
  tile.run(sm, Xm, (unsigned)B * Pm * 4u, Pm, Xm, (unsigned)B * Pm * 4u, Pm, ti * 128, tj * 128, 0, Pm, pro);
  float* o = out + (size_t)m * B * B;
  tile.for_each([&](int lr, int lc, float v) { o[(size_t)(ti * 128 + lr) * B + tj * 128 + lc] = v; });
}
__global__ __launch_bounds__(kThreads, 2) void f32_kernel(GemmOperands g, float* out) {
  using Tile = GemmTile<128, 128, false, true, NoPrologue, 32, true>;
  __shared__ typename Tile::Smem sm;
  const int nt = g.rowsA / 128, ti = blockIdx.x / nt, tj = blockIdx.x % nt, m = blockIdx.y;
  NoPrologue pro;
  Tile tile;
  tile.run(sm, g, g.A + (size_t)m * g.strideA, g.B + (size_t)m * g.strideB, ti * 128, tj * 128, 0, g.K, pro);
  float* o = out + (size_t)m * g.rowsA * g.colsB;
  tile.for_each([&](int lr, int lc, float v) { o[(size_t)(ti * 128 + lr) * g.colsB + tj * 128 + lc] = v; });
}

int main(int argc, char** argv) {
  const int M = argc > 2 ? atoi(argv[2]) : 64, B = 512, P = argc > 1 ? atoi(argv[1]) : 2048;
  float *X, *o1, *o2, *o3; const float mean = argc > 3 ? atof(argv[3]) : 0.25f;
  CK(hipMalloc(&X, (size_t)M * B * P * 4)); CK(hipMalloc(&o1, (size_t)M * B * B * 4)); CK(hipMalloc(&o2, (size_t)M * B * B * 4)); CK(hipMalloc(&o3, (size_t)M * B * B * 4));
  std::vector<float> h((size_t)M * B * P);
  srand(3);
  for (auto& v : h) { float u = (rand() % 20001) / 10000.f - 1.f; v = u * u * u + mean; }   // skewed, non-zero mean
  CK(hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  unsigned long long* dbg; CK(hipMalloc(&dbg, 8 * 32 * 8)); CK(hipMemset(dbg, 0, 8 * 32 * 8));
  GemmOperands g; g.A = X; g.B = X; g.strideA = g.strideB = (long long)B * P; g.lda = g.ldb = P; g.rowsA = g.colsB = B; g.K = P; g.kchunk = P;
  dim3 grid((B / 128) * (B / 128), M);
  auto run_x6 = [&]() { hipLaunchKernelGGL((x6_kernel<true, 1>), grid, dim3(kThreads), 0, 0, X, B, P, o1, dbg); };
  const int alias = argc > 4 ? atoi(argv[4]) : 0;
  const int dynlds = argc > 5 ? atoi(argv[5]) : 0;
  auto run_x6l = [&]() { hipLaunchKernelGGL(x6_kernel<true>, grid, dim3(kThreads), dynlds, 0, X, B, P | (alias << 30), o3, dbg); };
  auto run_f32 = [&]() { hipLaunchKernelGGL(f32_kernel, grid, dim3(kThreads), 0, 0, g, o2); };
  run_x6(); run_f32(); run_x6l(); CK(hipDeviceSynchronize()); CK(hipGetLastError());
  // accuracy vs fp64 on a sample of entries of member 0 and M-1
  std::vector<float> r1((size_t)B * B), r2((size_t)B * B), r3((size_t)B * B);
  double e1 = 0, e2 = 0, e3 = 0, nrm = 0, b1 = 0, b2 = 0, b3 = 0;
  for (int mm : {0, M - 1}) {
    CK(hipMemcpy(r1.data(), o1 + (size_t)mm * B * B, r1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r2.data(), o2 + (size_t)mm * B * B, r2.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r3.data(), o3 + (size_t)mm * B * B, r3.size() * 4, hipMemcpyDeviceToHost));
    const float* xm = h.data() + (size_t)mm * B * P;
    for (int s = 0; s < 4000; ++s) {
      const int a = rand() % B, b = rand() % B;
      double ref = 0, sabs = 0;
      for (int k = 0; k < P; ++k) { ref += (double)xm[(size_t)a * P + k] * xm[(size_t)b * P + k]; sabs += fabs((double)xm[(size_t)a * P + k] * xm[(size_t)b * P + k]); }
      e1 = std::max(e1, fabs(r1[(size_t)a * B + b] - ref) / sabs);
      e2 = std::max(e2, fabs(r2[(size_t)a * B + b] - ref) / sabs);
      e3 = std::max(e3, fabs(r3[(size_t)a * B + b] - ref) / sabs);
      b1 += (r1[(size_t)a * B + b] - ref) / sabs; b2 += (r2[(size_t)a * B + b] - ref) / sabs; b3 += (r3[(size_t)a * B + b] - ref) / sabs;
      nrm = std::max(nrm, fabs(ref));
    }
  }
  printf("max |err| / sum|a.b|:  f16x3 %.3e   fp32-mfma %.3e  bf16x6 %.3e\n", e1, e2, e3);
  printf("mean err / sum|a.b|:   f16x3 %.3e   fp32-mfma %.3e  bf16x6 %.3e\n", b1 / 8000, b2 / 8000, b3 / 8000);
  { int nb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, x6_kernel<true>, kThreads, 0)); printf("x6 kernel: %d workgroups per CU\n", nb);
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f32_kernel, kThreads, 0)); printf("f32 kernel: %d workgroups per CU\n", nb); }
  hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
  const double fl = 2.0 * M * B * (double)B * P;
  for (int v = 0; v < 3; ++v) {
    std::vector<float> ts;
    for (int r = 0; r < 6; ++r) {
      CK(hipEventRecord(ev0, 0));
      for (int i = 0; i < 3; ++i) { if (v == 0) run_x6(); else if (v == 1) run_f32(); else run_x6l(); }
      CK(hipEventRecord(ev1, 0)); CK(hipEventSynchronize(ev1));
      float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
      if (r) ts.push_back(ms / 3);
    }
    std::sort(ts.begin(), ts.end());
    printf("%s  med %8.1f us  %6.1f TF(fp32-equivalent)\n", v == 0 ? "f16x3    " : v == 1 ? "fp32-mfma" : "x6-xcdmap", ts[2] * 1e3, fl / ts[2] / 1e9);
  }
  return 0;
}
