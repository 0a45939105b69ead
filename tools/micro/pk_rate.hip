// Micro-benchmark: issue cost of packed fp32 VALU instructions (v_pk_fma/mul/add_f32) against their scalar forms on
// gfx950, 4 waves per SIMD, dependent chains of ILP independent accumulators per wave.
// build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o pk_rate pk_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <class T> struct W { static constexpr int n = 1; };
template <> struct W<v2f> { static constexpr int n = 2; };
template <> struct W<double> { static constexpr int n = 1; };

template <class T, int OP, int ILP>
__global__ void k(float* out, int iters, float a, float b) {
  T x[ILP];
#pragma unroll
  for (int q = 0; q < ILP; q++) x[q] = T(threadIdx.x * 1e-3f + q);
  T A = T(a), B = T(b);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
#pragma unroll
      for (int q = 0; q < ILP; q++) {
        if (OP == 0) x[q] = x[q] * A + B;       // fma (contracted)
        if (OP == 1) x[q] = x[q] * A;           // mul
        if (OP == 2) x[q] = x[q] + B;           // add
        if constexpr (OP == 3 && sizeof(T) == 4) x[q] = __builtin_amdgcn_rcpf(x[q]);                 // v_rcp_f32
        if constexpr (OP == 4 && sizeof(T) == 4) x[q] = __builtin_amdgcn_sqrtf(x[q]);                // v_sqrt_f32
        if constexpr (OP == 5 && W<T>::n == 1) x[q] = x[q] > B ? A : x[q];                         // v_cmp + v_cndmask
        if constexpr (OP == 6 && sizeof(T) == 4) {                                                   // one v_rcp_f32 among three fmas
          x[q] = (r & 3) == 0 ? __builtin_amdgcn_rcpf(x[q]) : x[q] * A + B;
        }
      }
    }
  }
  T s = T(0.f);
#pragma unroll
  for (int q = 0; q < ILP; q++) s += x[q];
  float r;
  if constexpr (W<T>::n == 2) r = s.x + s.y; else r = (float)s;
  if (r == 123.456f) out[0] = r;
}

template <class T, int OP, int ILP>
void run(const char* name, float* d) {
  const int cus = 256, iters = 2048, wps = 4;
  dim3 block(64 * 4 * wps), grid(cus);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k<T, OP, ILP>), grid, block, 0, 0, d, iters, 1.0001f, 1e-6f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL((k<T, OP, ILP>), grid, block, 0, 0, d, iters, 1.0001f, 1e-6f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double insts_per_simd = (double)iters * 16 * ILP * wps;
  printf("%-22s ILP %d : %.2f ns per wave-instruction per SIMD, %.2f ns per scalar-equivalent op\n", name, ILP,
         ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd / W<T>::n);
}

int main() {
  float* d;
  hipMalloc(&d, 4096);
  run<float, 0, 1>("v_fma_f32", d);    run<float, 0, 4>("v_fma_f32", d);
  run<float, 1, 1>("v_mul_f32", d);    run<float, 1, 4>("v_mul_f32", d);
  run<float, 2, 1>("v_add_f32", d);    run<float, 2, 4>("v_add_f32", d);
  run<v2f, 0, 1>("v_pk_fma_f32", d);   run<v2f, 0, 4>("v_pk_fma_f32", d);
  run<v2f, 1, 1>("v_pk_mul_f32", d);   run<v2f, 1, 4>("v_pk_mul_f32", d);
  run<v2f, 2, 1>("v_pk_add_f32", d);   run<v2f, 2, 4>("v_pk_add_f32", d);
  // transcendentals and selects (what the WENO weights are made of besides multiply-adds)
  run<float, 3, 1>("v_rcp_f32", d);    run<float, 3, 4>("v_rcp_f32", d);
  run<float, 4, 1>("v_sqrt_f32", d);   run<float, 4, 4>("v_sqrt_f32", d);
  run<float, 5, 1>("v_cmp+v_cndmask", d); run<float, 5, 4>("v_cmp+v_cndmask", d);
  run<float, 6, 4>("1 rcp : 3 fma", d);
  // fp64 (the pressure kernel's arithmetic)
  run<double, 0, 1>("v_fma_f64", d);   run<double, 0, 4>("v_fma_f64", d);
  run<double, 1, 4>("v_mul_f64", d);   run<double, 2, 4>("v_add_f64", d);
  return 0;
}
