// Micro-benchmark: sustained issue rate of plain fp32 VALU instructions on gfx950 as a function of waves per SIMD and
// of the instruction-level parallelism inside one wave.  Answers: is a stencil kernel that retires one wave64 VALU
// instruction per 4 cycles per SIMD saturated, or does the SIMD take one per 2 cycles when several waves are ready?
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ void k_fma(float* out, int iters, float a, float b) {
  float x[ILP];
#pragma unroll
  for (int q = 0; q < ILP; q++) x[q] = threadIdx.x * 1e-3f + q;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
#pragma unroll
      for (int q = 0; q < ILP; q++) x[q] = __builtin_fmaf(x[q], a, b);   // ILP independent chains
    }
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < ILP; q++) s += x[q];
  if (s == 123.456f) out[0] = s;
}

template <int ILP>
void run(int waves_per_simd, float* d) {
  const int cus = 256, iters = 4096;
  // one block = waves_per_simd * 4 waves -> one block per CU fills every SIMD with `waves_per_simd` waves
  dim3 block(64 * 4 * waves_per_simd), grid(cus);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_fma<ILP>, grid, block, 0, 0, d, 16, 1.0001f, 1e-6f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_fma<ILP>, grid, block, 0, 0, d, iters, 1.0001f, 1e-6f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_simd = (double)iters * 16 * ILP * waves_per_simd;
  const double tflops = (double)iters * 16 * ILP * 2 * 64 * 4 * waves_per_simd * cus / (ms * 1e-3) / 1e12;
  printf("ILP %d  waves/SIMD %d : %.3f ms  %.2f ns per wave-instruction per SIMD  (%.1f TFLOP/s)\n", ILP,
         waves_per_simd, ms, ms * 1e6 / insts_per_simd, tflops);
}

int main() {
  float* d;
  hipMalloc(&d, 4096);
  for (int w : {1, 2, 3, 4, 8}) run<1>(w, d);
  for (int w : {1, 2, 3, 4, 8}) run<4>(w, d);
  for (int w : {1, 2, 4}) run<8>(w, d);
  int clk = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("reported peak clock %d kHz: at 2.4 GHz, 4 cycles = 1.67 ns and 2 cycles = 0.83 ns\n", clk);
  return 0;
}
