// Scratch microbenchmark: random 256-B row gathers from a table >> Infinity Cache.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// G lanes per row (float4 each), U rows in flight per group
template <int G, int U>
__global__ __launch_bounds__(256) void gather_v(const float* __restrict__ tab, const int* __restrict__ idx, int64_t n, int D, float* out) {
  const int lane = threadIdx.x & 63, lig = lane % G;
  constexpr int GPW = 64 / G;
  const int64_t group = (((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6) * GPW + lane / G;
  const int64_t ngroups = (((int64_t)gridDim.x * 256) >> 6) * GPW;
  float acc = 0.f;
  for (int64_t r0 = group * U; r0 < n; r0 += ngroups * U) {
    float4 v[U];
    int id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) id[u] = (r0 + u < n) ? idx[r0 + u] : 0;
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4*>(tab + (int64_t)id[u] * D + lig * 4);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const int D = 64;
  for (int64_t rows : {1000000LL, 4000000LL}) {
    float* tab; CK(hipMalloc(&tab, rows * D * 4)); CK(hipMemset(tab, 0, rows * D * 4));
    const int64_t n = 3 * 65536;  // rows fetched per launch (one training batch)
    const int L = 32;  // distinct index sets
    std::vector<int> h(n * L); for (auto& x : h) x = (int)(((uint64_t)rand() * 2654435761ULL + rand()) % rows);
    int* idx; CK(hipMalloc(&idx, n * L * 4)); CK(hipMemcpy(idx, h.data(), n * L * 4, hipMemcpyHostToDevice));
    float* out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
      for (int w = 0; w < 2; ++w) launch(0);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int l = 0; l < L; ++l) launch(l);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("rows=%lld %-28s %.1f us/launch  %.2f TB/s\n", (long long)rows, name, ms / L * 1e3, n * 256.0 / (ms / L * 1e-3) / 1e12);
    };
#define RUN(G, U, GRID) run("G" #G " U" #U " grid" #GRID, [&](int l) { hipLaunchKernelGGL((gather_v<G, U>), dim3(GRID), dim3(256), 0, 0, tab, idx + (int64_t)l * n, n, D, out); })
    RUN(16, 1, 2048); RUN(16, 1, 4096); RUN(16, 1, 12288); RUN(16, 2, 2048); RUN(16, 4, 2048); RUN(16, 4, 1024); RUN(16, 8, 1024); RUN(16, 8, 512);
    RUN(16, 3, 4096); RUN(16, 3, 2048);
    CK(hipFree(tab)); CK(hipFree(idx)); CK(hipFree(out));
  }
  return 0;
}
