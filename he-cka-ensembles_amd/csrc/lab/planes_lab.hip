// Lab: speed and accuracy of the pre-split "planes" GEMM (gemm_planes.hpp) - NT (X X^T) and NN (W X).
//   planes_lab [P] [M]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "gemm_planes.hpp"
using namespace hecka;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

// x * s -> (h, l) in panel layout [member][cols/32][rows][32], one thread per 8 elements; x is [members][rows][cols]
__global__ void split_kernel(const float* x, size_t n8, size_t per_member8, int rows, int cols, const float* scale, _Float16* hi, _Float16* lo) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const float s = scale[i / per_member8];
  const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  h16x8 h, l;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float xs = v[q] * s;
    h[q] = (_Float16)xs;
    l[q] = (_Float16)(xs - (float)h[q]);
  }
  const size_t m = i / per_member8, e = (i % per_member8) * 8, r = e / cols, c = e % cols;
  const size_t o = m * per_member8 * 8 + ((c / 32) * rows + r) * 32 + c % 32;
  *reinterpret_cast<h16x8*>(hi + o) = h;
  *reinterpret_cast<h16x8*>(lo + o) = l;
}

__device__ __forceinline__ void member_tile(int per, int& m, int& tl) {
  const int w = blockIdx.y * gridDim.x + blockIdx.x, xcd = w & 7, slot = w >> 3;
  m = (slot / per) * 8 + xcd;
  tl = slot % per;
}

// out[m] = X_m X_m^T / s^2 ; all tiles
extern __shared__ __attribute__((aligned(1024))) unsigned char dyn_smem[];
__global__ __launch_bounds__(512, 2) void nt_kernel(const _Float16* Xh, const _Float16* Xl, const float* scale, int B, int P, float* out) {
  lds_u8* sm = (lds_u8*)dyn_smem;
  const int ntr = B / 256, ntc = B / 128;
  int m, tl;
  member_tile(ntr * ntc, m, tl);
  const int ti = tl / ntc, tj = tl % ntc;
  const size_t mo = (size_t)m * B * P;
  GemmTilePlanes<false> tile;
  tile.run(sm, Xh + mo, Xl + mo, B, Xh + mo, Xl + mo, B, ti * 256, tj * 128, P, [](int) { return 1.0f; });
  const float inv = 1.0f / (scale[m] * scale[m]);
  float* o = out + (size_t)m * B * B;
  tile.for_each([&](int lr, int lc, float v) { o[(size_t)(ti * 256 + lr) * B + tj * 128 + lc] = v * inv; });
}

// out[m] = W_m X_m / (sw sx): W [B,B] planes, X [B,P] planes
__global__ __launch_bounds__(512, 2) void nn_kernel(const _Float16* Wh, const _Float16* Wl, const float* wscale, const _Float16* Xh,
                                                    const _Float16* Xl, const float* xscale, int B, int P, float* out) {
  lds_u8* sm = (lds_u8*)dyn_smem;
  const int ntc = P / 128, ntr = B / 256;
  int m, tl;
  member_tile(ntc * ntr, m, tl);
  const int ti = tl % ntr, tj = tl / ntr;
  const size_t mo = (size_t)m * B * P, wo = (size_t)m * B * B;
  GemmTilePlanes<true> tile;
  tile.run(sm, Wh + wo, Wl + wo, B, Xh + mo, Xl + mo, B, ti * 256, tj * 128, B, [](int) { return 1.0f; });
  const float inv = 1.0f / (wscale[m] * xscale[m]);
  float* o = out + mo;
  tile.for_each([&](int lr, int lc, float v) { o[(size_t)(ti * 256 + lr) * P + tj * 128 + lc] = v * inv; });
}

static float pow2_scale_host(float bound) {   // s with bound * s in [2^13, 2^14)
  int e;
  frexpf(bound, &e);   // bound = f * 2^e, f in [0.5, 1)
  return ldexpf(1.0f, 14 - e);
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 2048, M = argc > 2 ? atoi(argv[2]) : 64, B = 512;
  const size_t nx = (size_t)M * B * P, nw = (size_t)M * B * B;
  std::vector<float> hx(nx), hw(nw), sx(M), sw(M);
  srand(3);
  for (auto& v : hx) { float u = (rand() % 20001) / 10000.f - 1.f; v = u * u * u + 0.25f; }
  for (auto& v : hw) { float u = (rand() % 20001) / 10000.f - 1.f; v = u * u * u * 3.f - 0.1f; }
  for (int m = 0; m < M; ++m) {
    float mx = 0, mw = 0;
    for (size_t i = 0; i < (size_t)B * P; ++i) mx = std::max(mx, fabsf(hx[(size_t)m * B * P + i]));
    for (size_t i = 0; i < (size_t)B * B; ++i) mw = std::max(mw, fabsf(hw[(size_t)m * B * B + i]));
    sx[m] = pow2_scale_host(mx); sw[m] = pow2_scale_host(mw);
  }
  float *X, *W, *dsx, *dsw, *o_nt, *o_nn;
  _Float16 *Xh, *Xl, *Wh, *Wl;
  CK(hipMalloc(&X, nx * 4)); CK(hipMalloc(&W, nw * 4)); CK(hipMalloc(&dsx, M * 4)); CK(hipMalloc(&dsw, M * 4));
  CK(hipMalloc(&Xh, nx * 2)); CK(hipMalloc(&Xl, nx * 2)); CK(hipMalloc(&Wh, nw * 2)); CK(hipMalloc(&Wl, nw * 2));
  CK(hipMalloc(&o_nt, nw * 4)); CK(hipMalloc(&o_nn, nx * 4));
  CK(hipMemcpy(X, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hw.data(), nw * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsx, sx.data(), M * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsw, sw.data(), M * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)((nx / 8 + 255) / 256)), dim3(256), 0, 0, X, nx / 8, (size_t)B * P / 8, B, P, dsx, Xh, Xl);
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)((nw / 8 + 255) / 256)), dim3(256), 0, 0, W, nw / 8, (size_t)B * B / 8, B, B, dsw, Wh, Wl);
  CK(hipDeviceSynchronize());
  dim3 gnt((B / 256) * (B / 128), M), gnn((B / 256) * (P / 128), M);
  CK(hipFuncSetAttribute((const void*)nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kPlSmemBytes));
  CK(hipFuncSetAttribute((const void*)nn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kPlSmemBytes));
  auto run_nt = [&]() { hipLaunchKernelGGL(nt_kernel, gnt, dim3(512), kPlSmemBytes, 0, Xh, Xl, dsx, B, P, o_nt); };
  auto run_nn = [&]() { hipLaunchKernelGGL(nn_kernel, gnn, dim3(512), kPlSmemBytes, 0, Wh, Wl, dsw, Xh, Xl, dsx, B, P, o_nn); };
  run_nt(); run_nn(); CK(hipDeviceSynchronize()); CK(hipGetLastError());
  // accuracy vs fp64 on sampled entries of members 0 and M-1
  double e_nt = 0, e_nn = 0, b_nt = 0, b_nn = 0;
  std::vector<float> r_nt((size_t)B * B), r_nn((size_t)B * P);
  for (int mm : {0, M - 1}) {
    CK(hipMemcpy(r_nt.data(), o_nt + (size_t)mm * B * B, r_nt.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r_nn.data(), o_nn + (size_t)mm * B * P, r_nn.size() * 4, hipMemcpyDeviceToHost));
    const float* xm = hx.data() + (size_t)mm * B * P;
    const float* wm = hw.data() + (size_t)mm * B * B;
    for (int s = 0; s < 4000; ++s) {
      const int a = rand() % B, b = rand() % B, p = rand() % P;
      double ref = 0, sabs = 0;
      for (int k = 0; k < P; ++k) { const double t = (double)xm[(size_t)a * P + k] * xm[(size_t)b * P + k]; ref += t; sabs += fabs(t); }
      e_nt = std::max(e_nt, fabs(r_nt[(size_t)a * B + b] - ref) / sabs); b_nt += (r_nt[(size_t)a * B + b] - ref) / sabs;
      ref = 0; sabs = 0;
      for (int k = 0; k < B; ++k) { const double t = (double)wm[(size_t)a * B + k] * xm[(size_t)k * P + p]; ref += t; sabs += fabs(t); }
      e_nn = std::max(e_nn, fabs(r_nn[(size_t)a * P + p] - ref) / sabs); b_nn += (r_nn[(size_t)a * P + p] - ref) / sabs;
    }
  }
  printf("max |err| / sum|a.b|:  NT %.3e   NN %.3e     mean err: NT %.3e  NN %.3e\n", e_nt, e_nn, b_nt / 8000, b_nn / 8000);
  { int nb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nt_kernel, 512, kPlSmemBytes)); printf("nt: %d workgroups per CU; ", nb);
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nn_kernel, 512, kPlSmemBytes)); printf("nn: %d workgroups per CU\n", nb); }
  hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
  const double fl = 2.0 * M * B * (double)B * P;
  for (int v = 0; v < 2; ++v) {
    std::vector<float> ts;
    for (int r = 0; r < 6; ++r) {
      CK(hipEventRecord(ev0, 0));
      for (int i = 0; i < 3; ++i) { if (v == 0) run_nt(); else run_nn(); }
      CK(hipEventRecord(ev1, 0)); CK(hipEventSynchronize(ev1));
      float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
      if (r) ts.push_back(ms / 3);
    }
    std::sort(ts.begin(), ts.end());
    printf("%s  med %8.1f us  %6.1f TF(fp32-equivalent, all tiles)\n", v == 0 ? "planes NT" : "planes NN", ts[2] * 1e3, fl / ts[2] / 1e9);
  }
  return 0;
}
