// v_mfma_f64_4x4x4_4b_f64 fed from LDS: one ds_read_b64 per instruction (or per two), the
// pattern of the band kernel's product loop without anything else around it.
// hipcc --offload-arch=gfx950 -O3 lds_mfma.hip -o lds_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)
constexpr int kPitch = 400;
constexpr int kRows = 12;

template <int kPerRead, bool kBarrier, int kStage>
__global__ __launch_bounds__(512) void spin(double *out, int iters, const float *src, long n_src) {
  extern __shared__ double img[];  // [2][kRows][kPitch]
  for (int i = threadIdx.x; i < 2 * kRows * kPitch; i += 512) img[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kk = lane >> 4, fr = lane & 15;
  double acc[33];
  for (int j = 0; j < 33; ++j) acc[j] = 0.0;
  int buf = 0;
  float pre[9];
  for (int e = 0; e < 9; ++e) pre[e] = 1.0f;
  // kStage: 0 nothing, 1 conversions + LDS stores, 2 also global loads (9 floats per thread)
  long cursor = (static_cast<long>(blockIdx.x) * 512 + threadIdx.x) * 3;
  for (int it = 0; it < iters; ++it) {
    if (kStage >= 1) {
#pragma unroll
      for (int e = 0; e < 9; ++e)
        img[((buf ^ 1) * kRows + (wave & 3) * 3 + e / 3) * kPitch + (wave >> 2) * 192 + lane + 64 * (e % 3)] =
            static_cast<double>(pre[e]);
    }
    if (kStage == 2) {
#pragma unroll
      for (int e = 0; e < 9; ++e) pre[e] = src[(cursor + (e % 3) + 192 * (e / 3)) & (n_src - 1)];
      cursor += 512 * 3 * 7;
    }
    if (kStage == 3) {  // the same bytes as three 12-byte loads
      struct F3 { float x, y, z; };
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const F3 v = *reinterpret_cast<const F3 *>(src + ((cursor + 192 * q) & (n_src - 1)));
        pre[3 * q] = v.x; pre[3 * q + 1] = v.y; pre[3 * q + 2] = v.z;
      }
      cursor += 512 * 3 * 7;
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const double *row = img + (buf * kRows + 4 * g + kk) * kPitch;
      const double a = row[16 * wave + fr];
      const double *shifted = row + 128 + 16 * wave + fr;
      double b[33];
#pragma unroll
      for (int u = 0; u < 6; ++u) b[u] = shifted[4 * u];
#pragma unroll
      for (int u = 0; u < 33; ++u) {
        if (u + 6 < 33) b[u + 6] = (kPerRead == 1 || (u & 1) == 0) ? shifted[4 * (u + 6)] : b[u + 5];
        if (u > 0) asm volatile("" ::"v"(b[u - 1]));
        acc[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b[u], acc[u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (kBarrier) __syncthreads();
    buf ^= 1;
  }
  double s = 0;
  for (int j = 0; j < 33; ++j) s += acc[j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int kPerRead, bool kBarrier, int kStage>
int run(const char *name, int blocks, int iters) {
  float *src; const long n_src = 2L << 20;  // 8 MB: stays in the L2s
  // 16 floats of slack: a 12-byte load may start at the last index
  CHECK(hipMalloc(&src, sizeof(float) * (n_src + 16)));
  CHECK(hipMemset(src, 0, sizeof(float) * (n_src + 16)));
  double *out;
  CHECK(hipMalloc(&out, sizeof(double) * blocks * 512));
  const size_t lds = sizeof(double) * 2 * kRows * kPitch;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(spin<kPerRead, kBarrier, kStage>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) spin<kPerRead, kBarrier, kStage><<<blocks, 512, lds>>>(out, iters, src, n_src);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  spin<kPerRead, kBarrier, kStage><<<blocks, 512, lds>>>(out, iters, src, n_src);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = 2.0 * 256 * 99 * 8.0 * blocks * iters;
  printf("%s blocks=%d: %.2f ms, %.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
  CHECK(hipFree(out)); CHECK(hipFree(src));
  return 0;
}
int main() {
  if (run<1, false, 0>("one read per mfma, no barrier            ", 512, 2000)) return 1;
  if (run<1, true, 0>("one read per mfma, barrier               ", 512, 2000)) return 1;
  if (run<2, true, 0>("one read per two, barrier                ", 512, 2000)) return 1;
  if (run<1, true, 1>("one read per mfma, barrier, stores       ", 512, 2000)) return 1;
  if (run<1, true, 2>("one read per mfma, barrier, stores, loads", 512, 2000)) return 1;
  if (run<1, false, 2>("one read per mfma, stores, loads         ", 512, 2000)) return 1;
  if (run<2, true, 2>("one read per two, barrier, stores, loads ", 512, 2000)) return 1;
  if (run<1, true, 3>("one read per mfma, barrier, stores, x3 loads", 512, 2000)) return 1;
  return 0;
}
