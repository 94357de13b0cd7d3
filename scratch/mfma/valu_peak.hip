// Sustained rate of v_fma_f64 (and v_add_f64 + v_fma_f64 pairs, the MSD kernel's mix) with
// operands in registers.  hipcc --offload-arch=gfx950 -O3 valu_peak.hip -o valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
template <int kMode>
__global__ __launch_bounds__(256) void spin(double *out, int iters, double a0) {
  double acc[16], x[16];
  for (int j = 0; j < 16; ++j) { acc[j] = j; x[j] = a0 + threadIdx.x * 1e-3 + j; }
  const double b = a0 * 0.5;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (kMode == 0) {
        acc[j] = __builtin_fma(x[j], b, acc[j]);
      } else {
        const double d = x[j] - acc[j];
        acc[j] = __builtin_fma(d, d, acc[j]);
      }
    }
  }
  double s = 0;
  for (int j = 0; j < 16; ++j) s += acc[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int kMode>
void run(int blocks, int iters) {
  double *out;
  (void)hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  spin<kMode><<<blocks, 256>>>(out, iters, 1.0);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  spin<kMode><<<blocks, 256>>>(out, iters, 1.0);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double instr = (kMode == 0 ? 1.0 : 2.0) * 16.0 * iters * 256.0 * blocks;
  printf("mode=%d blocks=%d: %.2f ms, %.2f T lane-instructions/s (fma counted as 2 flop: %.1f TFLOP/s)\n",
         kMode, blocks, ms, instr / ms / 1e9, (kMode == 0 ? 2.0 : 1.5) * instr / ms / 1e9);
  (void)hipFree(out);
}
int main() {
  run<0>(256 * 8, 20000);
  run<1>(256 * 8, 10000);
  run<0>(256 * 4, 20000);
  return 0;
}
