// Lane layout of v_mfma_f64_4x4x4_4b_f64 found by one-hot operands: wave (la, lb) sets
// a = 1 on lane la and b = 1 on lane lb; the lane of D that becomes 1 (if any) tells which
// A and B elements meet and where their product lands.
// hipcc --offload-arch=gfx950 -O3 probe4.hip -o probe4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int *hit) {
  const int la = blockIdx.x / 64, lb = blockIdx.x % 64, lane = threadIdx.x;
  const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  if (d != 0.0) hit[blockIdx.x] = lane;
}
int main() {
  int *hit;
  if (hipMalloc(&hit, 4096 * sizeof(int)) != hipSuccess) return 1;
  if (hipMemset(hit, 0xff, 4096 * sizeof(int)) != hipSuccess) return 1;
  probe<<<4096, 64>>>(hit);
  std::vector<int> h(4096);
  if (hipMemcpy(h.data(), hit, 4096 * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  for (int la = 0; la < 64; ++la) {
    printf("a lane %2d meets b lanes:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb] >= 0) printf(" %d->d%d", lb, h[la * 64 + lb]);
    printf("\n");
  }
  return 0;
}
