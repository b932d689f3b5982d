// Scratch: bisect what slows the fused scoring kernel relative to a bare gather.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int G> __device__ __forceinline__ float gsum(float v) {
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int LIN, int MATH, int STAGE, int I64>
__global__ __launch_bounds__(256) void k1like(const float* __restrict__ U, const float* __restrict__ I, const float* __restrict__ UL,
    const float* __restrict__ IL, const int* __restrict__ ui, const int* __restrict__ pi, const int* __restrict__ ni, int64_t B, int D,
    float* __restrict__ sc, float* __restrict__ du) {
  constexpr int G = 16, TPW = 4;
  const int lane = threadIdx.x & 63, lig = lane % G;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwave = ((int64_t)gridDim.x * 256) >> 6;
  const int64_t niter = (B + TPW - 1) / TPW;
  for (int64_t it = wave; it < niter; it += nwave) {
    const int64_t t = it * TPW + lane / G;
    if (t >= B) continue;
    int64_t u = ui[t], p = pi[t], n = ni[t];
    float4 a = *reinterpret_cast<const float4*>(U + u * D + lig * 4);
    float4 b = *reinterpret_cast<const float4*>(I + p * D + lig * 4);
    float4 c = *reinterpret_cast<const float4*>(I + n * D + lig * 4);
    float ul = 0, pl = 0, nl = 0;
    if (LIN) { ul = UL[u]; pl = IL[p]; nl = IL[n]; }
    float pp = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w, pn = a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
    if (MATH) { pp = gsum<G>(pp); pn = gsum<G>(pn); pp = 1.f / (1.f + expf(-(pp + ul + pl))); pn = 1.f / (1.f + expf(-(pn + ul + nl))); }
    if (STAGE) { float4 g = make_float4(pp * b.x + pn * c.x, pp * b.y + pn * c.y, pp * b.z + pn * c.z, pp * b.w + pn * c.w);
      *reinterpret_cast<float4*>(du + t * D + lig * 4) = g; }
    if (lig == 0) { sc[t] = pp + ul; sc[B + t] = pn + pl + nl; }
  }
}
int main() {
  const int D = 64; const int64_t NU = 1000000, NI = 100000, B = 65536; const int L = 48;
  float *U, *I, *UL, *IL, *sc, *du; CK(hipMalloc(&U, NU * D * 4)); CK(hipMalloc(&I, NI * D * 4)); CK(hipMalloc(&UL, NU * 4)); CK(hipMalloc(&IL, NI * 4));
  CK(hipMemset(U, 0, NU * D * 4)); CK(hipMemset(I, 0, NI * D * 4)); CK(hipMemset(UL, 0, NU * 4)); CK(hipMemset(IL, 0, NI * 4));
  CK(hipMalloc(&sc, 2 * B * 4)); CK(hipMalloc(&du, B * D * 4));
  std::vector<int> hu(B * L), hp(B * L), hn(B * L);
  for (auto& x : hu) x = (int)(((uint64_t)rand() * 2654435761ULL + rand()) % NU);
  for (auto& x : hp) x = (int)(((uint64_t)rand() * 2654435761ULL + rand()) % NI);
  for (auto& x : hn) x = (int)(((uint64_t)rand() * 2654435761ULL + rand()) % NI);
  int *du_, *dp, *dn; CK(hipMalloc(&du_, B * L * 4)); CK(hipMalloc(&dp, B * L * 4)); CK(hipMalloc(&dn, B * L * 4));
  CK(hipMemcpy(du_, hu.data(), B * L * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dp, hp.data(), B * L * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dn, hn.data(), B * L * 4, hipMemcpyHostToDevice));
  float* big; CK(hipMalloc(&big, 1LL << 30));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch, bool evict) {
    float tot = 0;
    for (int l = 0; l < L; ++l) {
      if (evict) CK(hipMemsetAsync(big, l, 1LL << 30, 0));
      CK(hipEventRecord(e0)); launch(l); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (l >= 4) tot += ms;
    }
    printf("%-40s %s %.1f us\n", name, evict ? "evicted" : "warm   ", tot / (L - 4) * 1e3);
  };
#define RUN(LIN, MATH, STAGE, GRID) for (int ev = 0; ev < 2; ++ev) run("LIN" #LIN " MATH" #MATH " STAGE" #STAGE " grid" #GRID, [&](int l) { \
    hipLaunchKernelGGL((k1like<LIN, MATH, STAGE, 0>), dim3(GRID), dim3(256), 0, 0, U, I, UL, IL, du_ + (int64_t)l * B, dp + (int64_t)l * B, dn + (int64_t)l * B, B, D, sc, du); }, ev)
  RUN(0, 0, 0, 2048); RUN(1, 0, 0, 2048); RUN(1, 1, 0, 2048); RUN(1, 1, 1, 2048); RUN(1, 1, 1, 4096);
  return 0;
}
