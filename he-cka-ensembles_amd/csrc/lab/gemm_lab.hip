// Development lab (not part of libhecka): times variants / ablations of the fp32 MFMA GEMM core on the
// dX = W X shape in ONE process, interleaved rounds (cdna_hip_programming.md 5.4 rule 24).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o /tmp/gemm_lab gemm_lab.hip && /tmp/gemm_lab [P]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "../gemm_mfma.hpp"
using namespace hecka;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

template <int TM, int TN, int BK, int OCC, bool FULL, int PRIO = 0>
__global__ __launch_bounds__(kThreads, OCC) void nn_kernel(GemmOperands g, float* __restrict__ out, int tilesN) {
  using Tile = GemmTile<TM, TN, true, true, NoPrologue, BK, FULL>;
  __shared__ typename Tile::Smem sm;
  const int ti = blockIdx.x / tilesN, tj = blockIdx.x % tilesN, m = blockIdx.y;
  if (PRIO) {
    const unsigned id = blockIdx.x + gridDim.x * blockIdx.y;
    if ((PRIO == 1 ? id : (id >> 8)) & 1) __builtin_amdgcn_s_setprio(1);
  }
  NoPrologue pro;
  Tile tile;
  tile.run(sm, g, g.A + (size_t)m * g.strideA, g.B + (size_t)m * g.strideB, ti * TM, tj * TN, 0, g.K, pro);
  float* o = out + (size_t)m * g.rowsA * g.colsB;
  {
    tile.for_each([&](int lr, int lc, float v) {
      const int row = ti * TM + lr, col = tj * TN + lc;
      if (row < g.rowsA && col < g.colsB) o[(size_t)row * g.colsB + col] = v;
    });
  }
}

struct Variant { const char* name; void (*launch)(GemmOperands, float*, int M, hipStream_t); };

template <int TM, int TN, int BK, int OCC, bool FULL, int PRIO = 0>
void launch(GemmOperands g, float* out, int M, hipStream_t st) {
  const int tilesM = (g.rowsA + TM - 1) / TM, tilesN = (g.colsB + TN - 1) / TN;
  hipLaunchKernelGGL((nn_kernel<TM, TN, BK, OCC, FULL, PRIO>), dim3(tilesM * tilesN, M), dim3(kThreads), 0, st, g, out, tilesN);
}

int main(int argc, char** argv) {
  const int M = 64, B = 512, P = argc > 1 ? atoi(argv[1]) : 2048;
  float *W, *X, *dX;
  CK(hipMalloc(&W, (size_t)M * B * B * 4)); CK(hipMalloc(&X, (size_t)M * B * P * 4)); CK(hipMalloc(&dX, (size_t)M * B * P * 4));
  std::vector<float> h((size_t)M * B * P);
  srand(1);
  for (auto& v : h) v = (float)(rand() % 20001) / 10000.f - 1.f;
  CK(hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data(), (size_t)M * B * B * 4, hipMemcpyHostToDevice));
  GemmOperands g;
  g.A = W; g.B = X; g.strideA = (long long)B * B; g.strideB = (long long)B * P; g.lda = B; g.ldb = P;
  g.rowsA = B; g.colsB = P; g.K = B; g.kchunk = B;
  Variant vs[] = {
    {"128x128 bk32 occ2        ", launch<128, 128, 32, 2, false>},
    {"128x128 bk32 occ2 FULL   ", launch<128, 128, 32, 2, true>},
    {"128x128 bk32 FULL prio1  ", launch<128, 128, 32, 2, true, 1>},
    {"128x128 bk32 FULL prio8  ", launch<128, 128, 32, 2, true, 2>},
    {"128x128 bk16 occ3 FULL   ", launch<128, 128, 16, 3, true>},
    {"256x256 bk32 occ1 FULL   ", launch<256, 256, 32, 1, true>},
    {"256x256 bk16 occ1 FULL   ", launch<256, 256, 16, 1, true>},
    {"256x128 bk32 occ1 FULL   ", launch<256, 128, 32, 1, true>},
    {"128x256 bk32 occ1 FULL   ", launch<128, 256, 32, 1, true>},
  };
  const int nv = sizeof(vs) / sizeof(vs[0]), rounds = 5;
  std::vector<std::vector<float>> t(nv);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < rounds + 1; ++r)
    for (int v = 0; v < nv; ++v) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 3; ++i) vs[v].launch(g, dX, M, 0);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r) t[v].push_back(ms / 3);
    }
  CK(hipGetLastError());
  const double fl = 2.0 * M * B * (double)B * P;
  for (int v = 0; v < nv; ++v) {
    std::sort(t[v].begin(), t[v].end());
    printf("%s  med %8.1f us  %6.1f TF   min %8.1f us\n", vs[v].name, t[v][rounds / 2] * 1e3, fl / t[v][rounds / 2] / 1e9, t[v][0] * 1e3);
  }
  return 0;
}
