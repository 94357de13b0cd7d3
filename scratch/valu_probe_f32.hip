// Sustained issue rate of the vector instructions the RDF pair loop is built from, on gfx950,
// operands in registers.  One block per CU is forced through its LDS allocation, so the
// occupancy (waves per SIMD) is what the label says.  Every kernel runs 16 independent
// dependency chains per lane; three warm-up launches bring the clocks out of idle.
// Prints wave64 instructions per second over the chip and the equivalent cycles per
// instruction per SIMD at 2.4 GHz (1024 SIMDs).
//   hipcc --offload-arch=gfx950 -O3 valu_probe_f32.hip -o valu_probe_f32
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define OPS(X)                                                                      \
  X(fma_f32, "v_fma_f32 %0, %0, %2, %3", 1)                                         \
  X(fma_f32_sgpr, "v_fma_f32 %0, %0, %4, %3", 1)                                    \
  X(mul_f32, "v_mul_f32 %0, %0, %2", 1)                                             \
  X(add_f32, "v_add_f32 %0, %0, %3", 1)                                             \
  X(sub_f32, "v_sub_f32 %0, %0, %3", 1)                                             \
  X(sub_f32_abs, "v_sub_f32_e64 %0, %2, |%0|", 1)                                   \
  X(sub_f32_sgpr, "v_sub_f32 %0, %4, %0", 1)                                        \
  X(min_f32, "v_min_f32 %0, %0, %2", 1)                                             \
  X(max_f32, "v_max_f32 %0, %0, %3", 1)                                             \
  X(med3_f32, "v_med3_f32 %0, %0, %2, %3", 1)                                       \
  X(min3_f32, "v_min3_f32 %0, %0, %2, %3", 1)                                       \
  X(sqrt_f32, "v_sqrt_f32 %0, %0", 1)                                               \
  X(rsq_f32, "v_rsq_f32 %0, %0", 1)                                                 \
  X(fract_f32, "v_fract_f32 %0, %0", 1)                                             \
  X(floor_f32, "v_floor_f32 %0, %0", 1)                                             \
  X(rndne_f32, "v_rndne_f32 %0, %0", 1)                                             \
  X(cvt_i32_f32, "v_cvt_i32_f32 %0, %0", 1)                                         \
  X(cvt_u32_f32, "v_cvt_u32_f32 %0, %0", 1)                                         \
  X(cvt_f32_i32, "v_cvt_f32_i32 %0, %0", 1)                                         \
  X(cvt_f32_u32, "v_cvt_f32_u32 %0, %0", 1)                                         \
  X(sub_u32, "v_sub_u32 %0, %0, %2", 1)                                             \
  X(sub_u32_sgpr, "v_sub_u32 %0, %4, %0", 1)                                        \
  X(and_b32, "v_and_b32 %0, %0, %2", 1)                                             \
  X(lshrrev_b32, "v_lshrrev_b32 %0, 1, %0", 1)                                      \
  X(lshlrev_b32, "v_lshlrev_b32 %0, 1, %0", 1)                                      \
  X(bfe_u32, "v_bfe_u32 %0, %0, 3, 12", 1)                                          \
  X(min_i32, "v_min_i32 %0, %0, %2", 1)                                             \
  X(min_u32, "v_min_u32 %0, %0, %2", 1)                                             \
  X(mad_i32_i24, "v_mad_i32_i24 %0, %0, %2, %3", 1)                                 \
  X(mul_lo_u32, "v_mul_lo_u32 %0, %0, %2", 1)                                       \
  X(and_or_b32, "v_and_or_b32 %0, %0, %2, %3", 1)                                   \
  X(cmp_lt_f32, "v_cmp_lt_f32 vcc, %0, %2", 1)                                      \
  X(cmp_lt_u32, "v_cmp_lt_u32 vcc, %0, %2", 1)                                      \
  X(cmp_lt_f32_sgpr_dst, "v_cmp_lt_f32_e64 s[20:21], %0, %2", 1)                    \
  X(cndmask, "v_cndmask_b32 %0, %0, %2, vcc", 1)                                    \
  X(cmp_cndmask, "v_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %3, vcc", 2)    \
  X(pk_fma_f32, "v_pk_fma_f32 %1, %1, %5, %6", 1)                                   \
  X(pk_mul_f32, "v_pk_mul_f32 %1, %1, %5", 1)                                       \
  X(pk_add_f32, "v_pk_add_f32 %1, %1, %6", 1)                                       \
  X(pk_add_f32_sgpr, "v_pk_add_f32 %1, %1, %7", 1)                                  \
  X(pk_add_f32_neg, "v_pk_add_f32 %1, %1, %6 neg_lo:[1,0] neg_hi:[1,0]", 1)         \
  X(mov_b32, "v_mov_b32 %0, %2", 1)                                                 \
  X(ds_add_u32, "ds_add_u32 %8, %9", 1)

enum Kind {
#define X(name, text, n) k_##name,
  OPS(X)
#undef X
      kKinds
};
const char *kNames[kKinds] = {
#define X(name, text, n) #name,
    OPS(X)
#undef X
};
const int kInstPerStmt[kKinds] = {
#define X(name, text, n) n,
    OPS(X)
#undef X
};

template <int kKind>
__global__ __launch_bounds__(1024) void spin(float *out, int iters, float a0, float sb) {
  extern __shared__ unsigned lds_pad[];  // sized by the host so that one block fills a CU
  float acc[16];
  v2f pacc[16];
  for (int j = 0; j < 16; ++j) {
    acc[j] = a0 + j + threadIdx.x * 1e-3f;
    pacc[j] = v2f{acc[j], acc[j] + 0.5f};
  }
  const float b = a0 * 0.999f, c = a0 * 1e-3f;
  const v2f pb = {b, b}, pc = {c, c};
  const v2f spair = {sb, sb + 1.f};
  // histogram-like LDS target: lanes spread over 4096 words like bins of a smooth g(r)
  const unsigned lds_addr = ((threadIdx.x * 37u + blockIdx.x) & 4095u) * 4u;
  const unsigned one = 1u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
#define X(name, text, n)                                                             \
  if (kKind == k_##name)                                                             \
    asm volatile(text                                                                \
                 : "+v"(acc[j]), "+v"(pacc[j])                                       \
                 : "v"(b), "v"(c), "s"(sb), "v"(pb), "v"(pc), "s"(spair),            \
                   "v"(lds_addr + 64u * j), "v"(one)                                 \
                 : "vcc", "s20", "s21", "memory");
      OPS(X)
#undef X
    }
  }
  float s = 0;
  for (int j = 0; j < 16; ++j) s += acc[j] + pacc[j].x + pacc[j].y;
  __syncthreads();
  if (a0 < 0.f) s += lds_pad[threadIdx.x];  // keeps the allocation alive
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int kKind>
void run(int waves_per_simd, int iters) {
  const int blocks_per_cu = waves_per_simd == 8 ? 2 : 1;
  const int threads = 256 * waves_per_simd / blocks_per_cu;
  const int blocks = 256 * blocks_per_cu;
  const size_t lds = blocks_per_cu == 2 ? 70 * 1024 : 100 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(spin<kKind>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int waves = blocks * threads / 64;
  float *out;
  (void)hipMalloc(&out, sizeof(float) * blocks * threads);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) spin<kKind><<<blocks, threads, lds>>>(out, iters, 1.0f, 0.5f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  spin<kKind><<<blocks, threads, lds>>>(out, iters, 1.0f, 0.5f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double wave_inst = (double)kInstPerStmt[kKind] * 16.0 * iters * waves;
  printf("%-20s waves/SIMD=%d  %7.3f ms  wave-inst/s=%.3e  cycles per inst per SIMD at 2.4 GHz=%.2f\n",
         kNames[kKind], waves_per_simd, ms, wave_inst / (ms * 1e-3),
         ms * 1e-3 * 2.4e9 * 1024.0 / wave_inst);
  (void)hipFree(out);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
}

int main(int argc, char **argv) {
  const int iters = 100000;
#define X(name, text, n)            \
  run<k_##name>(4, iters / 4);      \
  run<k_##name>(8, iters / 8);
  OPS(X)
#undef X
  return 0;
}
