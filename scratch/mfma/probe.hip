#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void probe(const double *A, const double *B, double *D) {
  // A[16][4] row-major, B[4][16] row-major
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + (l / 16)];
  const double b = B[(l / 16) * 16 + (l % 16)];
  v4d c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}
int main() {
  double hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 64; ++i) { hA[i] = (i * 7 % 13) + 1; hB[i] = (i * 5 % 11) - 3; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i*4+k]*hB[k*16+j]; ref[i*16+j] = s; }
  double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
  hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
  int ok1 = 1, ok2 = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    if (hD[l*4+r] != ref[(4*(l/16)+r)*16 + l%16]) ok1 = 0;
    if (hD[l*4+r] != ref[((l/16)+4*r)*16 + l%16]) ok2 = 0;
  }
  printf("layout row=4*(l/16)+r: %d   layout row=(l/16)+4*r: %d\n", ok1, ok2);
  return 0;
}
