// Host-to-device rates: pageable hipMemcpy, page-locked hipMemcpy, and the host-side copy
// (pageable -> page-locked) with 1..16 threads.  hipcc -O3 h2d_probe.cpp -o h2d_probe -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t n = size_t(2) << 30;
  char *pageable = static_cast<char *>(malloc(n));
  memset(pageable, 1, n);
  char *pinned; (void)hipHostMalloc(reinterpret_cast<void **>(&pinned), n, hipHostMallocDefault);
  memset(pinned, 2, n);
  void *dev; (void)hipMalloc(&dev, n);
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now(); (void)hipMemcpy(dev, pageable, n, hipMemcpyHostToDevice); double t1 = now();
    printf("pageable hipMemcpy: %.1f GB/s\n", n / (t1 - t0) / 1e9);
    t0 = now(); (void)hipMemcpy(dev, pinned, n, hipMemcpyHostToDevice); t1 = now();
    printf("page-locked hipMemcpy: %.1f GB/s\n", n / (t1 - t0) / 1e9);
  }
  for (int threads : {1, 2, 4, 8, 16}) {
    double t0 = now();
    std::vector<std::thread> pool;
    const size_t part = n / threads;
    for (int t = 0; t < threads; ++t)
      pool.emplace_back([=] { memcpy(pinned + t * part, pageable + t * part, part); });
    for (auto &th : pool) th.join();
    double t1 = now();
    printf("host copy pageable -> page-locked, %d threads: %.1f GB/s\n", threads, n / (t1 - t0) / 1e9);
  }
  return 0;
}
