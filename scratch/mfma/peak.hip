// Sustained rate of v_mfma_f64_16x16x4_f64 with nothing else going on: the ceiling the band
// kernel can be held against.  hipcc --offload-arch=gfx950 -O3 peak.hip -o peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int kAcc>
__global__ __launch_bounds__(256) void spin(double *out, int iters, double a0, double b0) {
  v4d acc[kAcc];
  for (int j = 0; j < kAcc; ++j) acc[j] = v4d{0, 0, 0, 0};
  double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < kAcc; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
  }
  double s = 0;
  for (int j = 0; j < kAcc; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int kAcc>
void run(int blocks, int iters) {
  double *out;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  spin<kAcc><<<blocks, 256>>>(out, iters, 1.0, 2.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  spin<kAcc><<<blocks, 256>>>(out, iters, 1.0, 2.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2048.0 * kAcc * iters * 4.0 * blocks;  // per wave, 4 waves per block
  printf("acc=%d blocks=%d: %.2f ms, %.1f TFLOP/s\n", kAcc, blocks, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<4>(256 * 2, 20000);
  run<8>(256 * 2, 10000);
  run<17>(256 * 2, 5000);
  run<8>(256 * 4, 10000);
  run<8>(256 * 1, 10000);
  run<8>(256 * 4, 100000);  // 140 ms: the rate once clocks have settled
  run<8>(256 * 2, 100000);
  return 0;
}
