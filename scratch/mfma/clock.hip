// Shader clock actually held while v_mfma_f64_16x16x4_f64 runs back to back, against a
// float32 vector loop: s_memtime counts shader cycles, s_memrealtime a constant 100 MHz.
// hipcc --offload-arch=gfx950 -O3 clock.hip -o clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int kMode>
__global__ __launch_bounds__(256) void spin(double *out, unsigned long long *clk, int iters,
                                            double a0, double b0) {
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  if (kMode == 0) {
    v4d acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = v4d{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b + j, acc[j], 0, 0, 0);
    }
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  } else if (kMode == 2) {
    double acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = 0.0;
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j)
        acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b + j, acc[j], 0, 0, 0);
    }
    for (int j = 0; j < 16; ++j) s += acc[j];
  } else if (kMode == 3) {
    double f[16];
    for (int j = 0; j < 16; ++j) f[j] = a0 + threadIdx.x + j;
    for (int i = 0; i < iters * 2; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) f[j] = fma(f[j], b0, 1.0);
    }
    for (int j = 0; j < 16; ++j) s += f[j];
  } else {
    float f[16];
    for (int j = 0; j < 16; ++j) f[j] = static_cast<float>(a0) + threadIdx.x + j;
    const float m = static_cast<float>(b0);
    for (int i = 0; i < iters * 8; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) f[j] = f[j] * m + 1.0f;
    }
    for (int j = 0; j < 16; ++j) s += f[j];
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int kMode>
int run(const char *name, int blocks, int iters) {
  double *out; unsigned long long *clk;
  CHECK(hipMalloc(&out, sizeof(double) * blocks * 256));
  CHECK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) spin<kMode><<<blocks, 256>>>(out, clk, iters, 1.0, 0.999);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  spin<kMode><<<blocks, 256>>>(out, clk, iters, 1.0, 0.999);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(2 * blocks);
  CHECK(hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
  double cyc = 0, real = 0;
  for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; real += h[2 * b + 1]; }
  printf("%s blocks=%d: %.2f ms, s_memtime/s_memrealtime = %.3f (x100 MHz = %.0f MHz if s_memtime counts shader cycles)\n",
         name, blocks, ms, cyc / real, 100.0 * cyc / real);
  CHECK(hipFree(out)); CHECK(hipFree(clk));
  return 0;
}
int main() {
  if (run<1>("f32 fma  ", 1024, 20000)) return 1;
  if (run<0>("f64 mfma ", 1024, 20000)) return 1;
  if (run<0>("f64 mfma ", 1024, 100000)) return 1;
  if (run<1>("f32 fma  ", 1024, 100000)) return 1;
  // 16 x (4 blocks of 4x4x4) = 4096 fma per wave and iteration; 8 x 16x16x4 = 8192
  if (run<2>("f64 mfma 4x4x4 ", 1024, 100000)) return 1;
  if (run<2>("f64 mfma 4x4x4 ", 2048, 100000)) return 1;
  if (run<2>("f64 mfma 4x4x4 ", 512, 100000)) return 1;   // 2 waves per SIMD
  if (run<2>("f64 mfma 4x4x4 ", 256, 100000)) return 1;   // 1 wave per SIMD
  if (run<2>("f64 mfma 4x4x4 ", 768, 100000)) return 1;   // 3 waves per SIMD
  // 32 v_fma_f64 per lane and iteration
  if (run<3>("f64 v_fma", 2048, 100000)) return 1;
  return 0;
}
