// Microbenchmark: do MFMA and VALU streams overlap on one SIMD of an MI355X (a) across two waves, (b) inside one wave?
// 512-thread workgroups (8 waves: waves w and w+4 share a SIMD), one workgroup per CU.
// mode 0: every wave MFMA only        mode 1: every wave VALU only (fma + exp mix like the attention loops)
// mode 2: waves 0-3 MFMA, waves 4-7 VALU (cross-wave overlap)   mode 3: every wave: 1 MFMA then K VALU, repeated
// mode 4: waves 0-3 only (one wave per SIMD) running mode 3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE, int K>
__global__ void __launch_bounds__(512) bench(float* out, unsigned long long* cycles, int iters) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * (lane - j)); }
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (lane + i);
  const float c = 0.999f, d = 0.0001f;
  // mode 5: as mode 2 with the VALU waves at s_setprio 3;  mode 6: roles swapped (waves 0-3 VALU, 4-7 MFMA);
  // mode 7: roles swapped and the MFMA waves at s_setprio 3
  const bool do_mfma = MODE == 0 || ((MODE == 2 || MODE == 5) && w < 4) || ((MODE == 6 || MODE == 7) && w >= 4) || MODE == 3 || MODE == 4;
  const bool do_valu = MODE == 1 || ((MODE == 2 || MODE == 5) && w >= 4) || ((MODE == 6 || MODE == 7) && w < 4) || MODE == 3 || MODE == 4;
  if (MODE == 4 && w >= 4) return;
  // mode 8/9/10: waves 0-3 MFMA, each followed by scalar s_nop padding of K cycles (K = 16 / 24 / 28 via template K);
  //              waves 4-7 VALU groups of 5.  Does the padding free the vector issue port for the partner wave?
  if (MODE == 8) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (w < 4) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (K == 16) asm volatile("s_nop 15");
          if (K == 3) asm volatile("s_nop 3");
          if (K == 4) asm volatile("s_nop 4");
          if (K == 5) asm volatile("s_nop 5");
          if (K == 6) asm volatile("s_nop 6");
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            const int i = (m * 5 + q) & 15;
            if ((q & 3) == 3) v[i] = __builtin_amdgcn_exp2f(v[i]);
            else v[i] = __builtin_fmaf(v[i], c, d);
          }
      }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cycles[blockIdx.x * 8 + w] = t1 - t0;
    return;
  }
  if (MODE == 5 && w >= 4) __builtin_amdgcn_s_setprio(3);
  if (MODE == 7 && w >= 4) __builtin_amdgcn_s_setprio(3);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 3 || MODE == 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < K; ++q) {
          const int i = (m * K + q) & 15;
          if ((q & 3) == 3) v[i] = __builtin_amdgcn_exp2f(v[i]);
          else v[i] = __builtin_fmaf(v[i], c, d);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
    if (do_mfma) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        }
      }
    }
    if (do_valu) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
          for (int q = 0; q < K; ++q) {
            const int i = (m * K + q) & 15;
            if ((q & 3) == 3) v[i] = __builtin_amdgcn_exp2f(v[i]);
            else v[i] = __builtin_fmaf(v[i], c, d);
          }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0) cycles[blockIdx.x * 8 + w] = t1 - t0;
}

template <int MODE, int K> void run(const char* name, float* out, unsigned long long* cyc, int iters) {
  hipMemset(cyc, 0, 256 * 8 * 8);
  bench<MODE, K><<<256, 512>>>(out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  bench<MODE, K><<<256, 512>>>(out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double lo = 0, hi = 0; int nlo = 0, nhi = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) { double c = (double)h[b * 8 + w]; if (c == 0) continue; if (w < 4) { lo += c; ++nlo; } else { hi += c; ++nhi; } }
  const double per = 8.0 * iters;   // MFMAs (or VALU groups of K) per wave
  printf("%-44s K=%2d  %.3f ms   waves0-3: %.1f cyc per MFMA-slot   waves4-7: %.1f\n", name, K, ms,
         nlo ? lo / nlo / per : 0.0, nhi ? hi / nhi / per : 0.0);
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 2000;
  run<0, 5>("all waves MFMA only (2 waves/SIMD)", out, cyc, iters);
  run<1, 5>("all waves VALU only, 5 per slot", out, cyc, iters);
  run<1, 8>("all waves VALU only, 8 per slot", out, cyc, iters);
  run<2, 5>("waves0-3 MFMA | waves4-7 VALU (5/slot)", out, cyc, iters);
  run<2, 8>("waves0-3 MFMA | waves4-7 VALU (8/slot)", out, cyc, iters);
  run<2, 12>("waves0-3 MFMA | waves4-7 VALU (12/slot)", out, cyc, iters);
  run<5, 5>("w0-3 MFMA | w4-7 VALU(5) at prio 3", out, cyc, iters);
  run<5, 8>("w0-3 MFMA | w4-7 VALU(8) at prio 3", out, cyc, iters);
  run<6, 5>("w0-3 VALU(5) | w4-7 MFMA", out, cyc, iters);
  run<6, 8>("w0-3 VALU(8) | w4-7 MFMA", out, cyc, iters);
  run<7, 5>("w0-3 VALU(5) | w4-7 MFMA at prio 3", out, cyc, iters);
  run<8, 16>("w0-3 MFMA + s_nop 15 | w4-7 VALU(5)", out, cyc, iters);
  run<8, 3>("w0-3 MFMA + s_nop 3 | w4-7 VALU(5)", out, cyc, iters);
  run<8, 4>("w0-3 MFMA + s_nop 4 | w4-7 VALU(5)", out, cyc, iters);
  run<8, 5>("w0-3 MFMA + s_nop 5 | w4-7 VALU(5)", out, cyc, iters);
  run<8, 6>("w0-3 MFMA + s_nop 6 | w4-7 VALU(5)", out, cyc, iters);
  run<3, 3>("every wave: MFMA + 3 VALU interleaved", out, cyc, iters);
  run<3, 5>("every wave: MFMA + 5 VALU interleaved", out, cyc, iters);
  run<3, 8>("every wave: MFMA + 8 VALU interleaved", out, cyc, iters);
  run<4, 3>("one wave/SIMD: MFMA + 3 VALU interleaved", out, cyc, iters);
  run<4, 5>("one wave/SIMD: MFMA + 5 VALU interleaved", out, cyc, iters);
  run<4, 8>("one wave/SIMD: MFMA + 8 VALU interleaved", out, cyc, iters);
  return 0;
}
