// Scratch: rocprim::segmented_radix_sort_pairs (17-bit keys, fixed 131072-element segments) vs one global radix sort on
// (segment, key) — the two ways to group each batch's item references by row.
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <vector>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
struct Off { unsigned seg; __host__ __device__ unsigned operator()(unsigned i) const { return i * seg; } };
int main(int argc, char** argv) {
  const unsigned nseg = argc > 1 ? atoi(argv[1]) : 512, seg = argc > 2 ? atoi(argv[2]) : 131072, bits = argc > 3 ? atoi(argv[3]) : 17;
  const size_t n = (size_t)nseg * seg;
  std::vector<unsigned> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (unsigned)((i * 2654435761u) >> 7) % (bits == 17 ? 100000u : 1000000u);
  unsigned *k0, *k1; uint2 *v0, *v1;  unsigned *w0, *w1;
  hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * 8); hipMalloc(&v1, n * 8); hipMalloc(&w0, n * 4); hipMalloc(&w1, n * 4);
  hipMemcpy(k0, h.data(), n * 4, hipMemcpyHostToDevice);
  auto offs = rocprim::make_transform_iterator(rocprim::counting_iterator<unsigned>(0), Off{seg});
  auto offe = rocprim::make_transform_iterator(rocprim::counting_iterator<unsigned>(1), Off{seg});
  size_t t1 = 0, t2 = 0, t3 = 0;
  rocprim::segmented_radix_sort_pairs(nullptr, t1, k0, k1, v0, v1, n, nseg, offs, offe, 0, bits, 0);
  rocprim::radix_sort_pairs(nullptr, t2, k0, k1, v0, v1, n, 0, 26, 0);
  rocprim::segmented_radix_sort_pairs(nullptr, t3, k0, k1, w0, w1, n, nseg, offs, offe, 0, bits, 0);
  void* tmp; hipMalloc(&tmp, (t1 > t2 ? t1 : t2) + t3 + 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); rocprim::segmented_radix_sort_pairs(tmp, t1, k0, k1, v0, v1, n, nseg, offs, offe, 0, bits, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("segmented 17-bit, 8-B payload: %.3f ms\n", ms);
    hipEventRecord(e0); rocprim::segmented_radix_sort_pairs(tmp, t3, k0, k1, w0, w1, n, nseg, offs, offe, 0, bits, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("segmented 17-bit, 4-B payload: %.3f ms\n", ms);
#define TRY(R, BS, IPT)                                                                                              \
    {                                                                                                                \
      using cfg = rocprim::segmented_radix_sort_config<R, rocprim::kernel_config<BS, IPT>>;                          \
      size_t tt = 0;                                                                                                 \
      hipError_t er = rocprim::segmented_radix_sort_pairs<cfg>(nullptr, tt, k0, k1, w0, w1, n, nseg, offs, offe, 0, bits, 0); \
      if (er == hipSuccess && tt <= (t1 > t2 ? t1 : t2) + t3) {                                                      \
        hipEventRecord(e0);                                                                                          \
        er = rocprim::segmented_radix_sort_pairs<cfg>(tmp, tt, k0, k1, w0, w1, n, nseg, offs, offe, 0, bits, 0);      \
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);                               \
        printf("segmented cfg<%d,%d,%d> 4-B payload: %.3f ms (%s)\n", R, BS, IPT, ms, hipGetErrorString(er));       \
      } else printf("cfg<%d,%d,%d> skipped (%s, temp %zu)\n", R, BS, IPT, hipGetErrorString(er), tt);               \
    }
    if (rep == 1) {
      {
        using cfg = rocprim::segmented_radix_sort_config<9, rocprim::kernel_config<1024, 8>>;
        size_t tt = 0;
        rocprim::segmented_radix_sort_pairs<cfg>(nullptr, tt, k0, k1, v0, v1, n, nseg, offs, offe, 0, bits, 0);
        hipEventRecord(e0); rocprim::segmented_radix_sort_pairs<cfg>(tmp, tt, k0, k1, v0, v1, n, nseg, offs, offe, 0, bits, 0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("segmented cfg<9,1024,8> 8-B payload: %.3f ms\n", ms);
      }
      TRY(9, 1024, 8) TRY(9, 1024, 16) TRY(9, 1024, 12) TRY(10, 1024, 8) TRY(10, 1024, 16)
    }
    hipEventRecord(e0); rocprim::radix_sort_pairs(tmp, t2, k0, k1, v0, v1, n, 0, 26, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("global 26-bit, 8-B payload:    %.3f ms\n", ms);
    hipEventRecord(e0); rocprim::radix_sort_pairs(tmp, t2, k0, k1, w0, w1, n, 0, 26, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("global 26-bit, 4-B payload:    %.3f ms\n", ms);
    hipEventRecord(e0); rocprim::radix_sort_pairs(tmp, t2, k0, k1, w0, w1, n, 0, 24, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("global 24-bit, 4-B payload:    %.3f ms\n", ms);
  }
  return 0;
}
