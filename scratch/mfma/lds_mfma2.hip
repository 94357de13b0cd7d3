// As lds_mfma.hip, with the stores and loads of a round moved to two extra waves of the
// block (waves 8 and 9) that issue no matrix instructions.
// hipcc --offload-arch=gfx950 -O3 lds_mfma2.hip -o lds_mfma2
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)
constexpr int kPitch = 400;
constexpr int kRows = 12;
constexpr int kSrc = 2 << 20;  // floats, 8 MB: stays in the L2s (+16 floats of slack)

__global__ __launch_bounds__(640) void spin(double *out, int iters, const float *src) {
  extern __shared__ double img[];  // [2][kRows][kPitch]
  for (int i = threadIdx.x; i < 2 * kRows * kPitch; i += 640) img[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int buf = 0;
  if (wave >= 8) {
    // producers: 4608 values per round over 128 lanes = 36 per lane, as 12 loads of 12 bytes
    struct F3 { float x, y, z; };
    const int p = (wave - 8) * 64 + lane;  // 0..127
    F3 pre[12];
    for (int e = 0; e < 12; ++e) pre[e] = F3{1.0f, 1.0f, 1.0f};
    long cursor = (static_cast<long>(blockIdx.x) * 128 + p) * 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int e = 0; e < 12; ++e) {
        // 4 series x 384 frames: item = e * 128 + p
        const int item = e * 128 + p, series = item / 384, frame = item % 384;
        double *dst = img + ((buf ^ 1) * kRows + series * 3) * kPitch + frame;
        dst[0] = static_cast<double>(pre[e].x);
        dst[kPitch] = static_cast<double>(pre[e].y);
        dst[2 * kPitch] = static_cast<double>(pre[e].z);
      }
#pragma unroll
      for (int e = 0; e < 12; ++e)
        pre[e] = *reinterpret_cast<const F3 *>(src + ((cursor + 384 * e) & (kSrc - 1)));
      cursor += 128 * 3 * 7;
      __syncthreads();
      buf ^= 1;
    }
    return;
  }
  const int kk = lane >> 4, fr = lane & 15;
  double acc[33];
  for (int j = 0; j < 33; ++j) acc[j] = 0.0;
  for (int it = 0; it < iters; ++it) {
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
        if (u + 6 < 33) b[u + 6] = shifted[4 * (u + 6)];
        if (u > 0) asm volatile("" ::"v"(b[u - 1]));
        acc[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b[u], acc[u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    buf ^= 1;
  }
  double s = 0;
  for (int j = 0; j < 33; ++j) s += acc[j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  const int blocks = 512, iters = 2000;
  double *out; float *src;
  CHECK(hipMalloc(&out, sizeof(double) * blocks * 512));
  CHECK(hipMalloc(&src, sizeof(float) * (kSrc + 16)));
  CHECK(hipMemset(src, 0, sizeof(float) * (kSrc + 16)));
  const size_t lds = sizeof(double) * 2 * kRows * kPitch;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(spin),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  int per_cu = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, spin, 640, lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) spin<<<blocks, 640, lds>>>(out, iters, src);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  spin<<<blocks, 640, lds>>>(out, iters, src);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = 2.0 * 256 * 99 * 8.0 * blocks * iters;
  printf("8 matrix waves + 2 staging waves, blocks per CU by the API %d: %.2f ms, %.1f TFLOP/s\n",
         per_cu, ms, flops / ms / 1e9);
  return 0;
}
