// Sustained MFMA rate of one MI355X under load: what the chip actually delivers when every SIMD issues MFMAs back to
// back on RANDOM operands (the clock it holds depends on the data and on the MFMA shape; zeros run much faster).
// SURVEY.md section 8(d) asks for this number next to the nominal 2.5 PFLOP/s bf16 / 157.3 TFLOP/s fp32 peaks.
//   build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 mfma_peak.hip -o mfma_peak && ./mfma_peak
// Per variant: wall TFLOP/s over >= 0.5 s of back-to-back launches, cycles per MFMA (s_memtime) and the in-kernel clock
// (s_memtime / s_memrealtime, 100 MHz reference).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// SHAPE 0: v_mfma_f32_32x32x16_bf16, 1: v_mfma_f32_16x16x32_bf16, 2: v_mfma_f32_32x32x2_f32
// LDS 0: operands stay in registers; 1: the A operand of every MFMA is re-read from LDS (ds_read_b128)
template <int SHAPE, int LDS>
__global__ void __launch_bounds__(512) peak(const float* __restrict__ seed, float* __restrict__ out,
                                            unsigned long long* __restrict__ stamps, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  // random operands (different per lane and per register)
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      a[i][j] = (__bf16)seed[(tid * 64 + i * 8 + j) & 65535];
      b[i][j] = (__bf16)seed[(tid * 64 + 32 + i * 8 + j) & 65535];
    }
  for (int i = tid; i < 64 * 1024 / 4; i += blockDim.x) ((float*)smem)[i] = seed[i & 65535];
  __syncthreads();
  const char* lbase = smem + (tid & 63) * 16 + (tid >> 6) * 4096;
  f32x16 c32[4];
  f32x4 c16[8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) c32[i][j] = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 4; ++j) c16[i][j] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      bf16x8 av = a[u & 3];
      if (LDS) av = *(const bf16x8*)(lbase + ((u * 1024 + it * 64) & 3072));
      if (SHAPE == 0) {
        c32[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[u & 3], c32[u & 3], 0, 0, 0);
      } else if (SHAPE == 1) {
        c16[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b[u & 3], c16[u], 0, 0, 0);
      } else {
        const float af = __builtin_bit_cast(float, __builtin_shufflevector(av, av, 0, 1)), bf = seed[0] + (float)u;
        c32[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, c32[u & 3], 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += c32[i][j];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 4; ++j) s += c16[i][j];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (lane == 0) {
    stamps[(blockIdx.x * 8 + (tid >> 6)) * 2] = t1 - t0;
    stamps[(blockIdx.x * 8 + (tid >> 6)) * 2 + 1] = r1 - r0;
  }
}

// Power model probe: 32x32x16 MFMA (SHAPE 0) or 16x16x32 (SHAPE 1) with, per 32 MFMA-cycles, NL ds_read_b128 (distinct
// addresses, consumed as MFMA operands) and NV "softmax elements" (mul, exp2, mul, cvt_pk, xor = 24 issue cycles each).
template <int SHAPE, int NL, int NV>
__global__ void __launch_bounds__(512) mix(const float* __restrict__ seed, float* __restrict__ out,
                                           unsigned long long* __restrict__ stamps, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  bf16x8 a[4], b[4];
  float w[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      a[i][j] = (__bf16)seed[(tid * 64 + i * 8 + j) & 65535];
      b[i][j] = (__bf16)seed[(tid * 64 + 32 + i * 8 + j) & 65535];
    }
  for (int i = 0; i < 16; ++i) w[i] = seed[(tid * 16 + i) & 65535];
  for (int i = tid; i < 64 * 1024 / 4; i += blockDim.x) ((float*)smem)[i] = seed[i & 65535];
  __syncthreads();
  const char* lbase = smem + lane * 16;
  f32x16 c32[4];
  f32x4 c16[8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) c32[i][j] = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 4; ++j) c16[i][j] = 0.f;
  unsigned int x = 0;
  float cit = seed[1];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  bf16x8 avn = a[0], bvn = b[0];
  for (int it = 0; it < iters; ++it) {
    cit += 0.0009765625f;
    const int rot = (it * 8192) & 0xffff;
#pragma unroll
    for (int u = 0; u < 8; ++u) {   // one u = 32 MFMA-cycles; LDS operands are requested one slot ahead
      bf16x8 av = a[u & 3], bv = b[u & 3];
      if (NL >= 1) { av = avn; avn = *(const bf16x8*)(lbase + ((rot + u * 1024) & 0xffff)); }
      if (NL >= 2) { bv = bvn; bvn = *(const bf16x8*)(lbase + ((rot + u * 1024 + 32768) & 0xffff)); }
      if (SHAPE == 0) {
        c32[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c32[u & 3], 0, 0, 0);
      } else {
        c16[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c16[u], 0, 0, 0);
        c16[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv, av, c16[u], 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        const int i = (u * NV + e) & 15;
        const float p = __builtin_amdgcn_exp2f(w[i] * cit);
        const float q = p * w[(i + 5) & 15];
        typedef __attribute__((ext_vector_type(2))) float f2;
        typedef __attribute__((ext_vector_type(2))) __bf16 b2;
        f2 pr = {p, q};
        x ^= __builtin_bit_cast(unsigned int, __builtin_convertvector(pr, b2));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = (float)x;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += c32[i][j];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 4; ++j) s += c16[i][j];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (lane == 0) {
    stamps[(blockIdx.x * 8 + (tid >> 6)) * 2] = t1 - t0;
    stamps[(blockIdx.x * 8 + (tid >> 6)) * 2 + 1] = r1 - r0;
  }
}

template <typename F>
void run_generic(const char* name, F launch, double flops_per_launch, int threads, int blocks, double slots_per_wave,
                 unsigned long long* stamps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) launch();
  hipDeviceSynchronize();
  int reps = 50;
  float ms = 0.f;
  for (int round = 0; round < 2; ++round) {
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (round == 0) reps = std::max(20, (int)(400.0f / (ms / reps)));
  }
  std::vector<unsigned long long> h(blocks * 8 * 2);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> clk, cyc;
  for (int i = 0; i < blocks * (threads / 64); ++i) {
    const int blk = i / (threads / 64), w = i % (threads / 64);
    const double c = (double)h[(blk * 8 + w) * 2], r = (double)h[(blk * 8 + w) * 2 + 1];
    if (r > 0) {
      clk.push_back(c / r * 0.1);
      cyc.push_back(c / slots_per_wave);
    }
  }
  std::sort(clk.begin(), clk.end());
  std::sort(cyc.begin(), cyc.end());
  printf("%-58s %8.1f TFLOP/s   %.1f cyc per 32-MFMA-cycle slot per wave   clock %.3f GHz\n", name,
         flops_per_launch / (ms / reps * 1e-3) / 1e12, cyc[cyc.size() / 2], clk[clk.size() / 2]);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

template <int SHAPE, int NL, int NV>
void run_mix(int waves_per_simd, const float* seed, float* out, unsigned long long* stamps) {
  const int threads = 256 * waves_per_simd, blocks = 256, iters = 4000;
  char name[128];
  snprintf(name, sizeof(name), "%s + %d ds_read_b128 + %d softmax elems /slot, %d w/SIMD", SHAPE == 0 ? "32x32x16" : "2x 16x16x32",
           NL, NV, waves_per_simd);
  const double flops = (double)blocks * (threads / 64) * iters * 8.0 * 2.0 * 32 * 32 * 16;
  run_generic(name, [&]() { mix<SHAPE, NL, NV><<<blocks, threads>>>(seed, out, stamps, iters); }, flops, threads, blocks,
              iters * 8.0, stamps);
}

template <int SHAPE, int LDS>
void run(const char* name, int waves_per_simd, const float* seed, float* out, unsigned long long* stamps, bool zeros) {
  const int threads = 256 * waves_per_simd, blocks = 256, iters = 4000;
  const double flops_per_mfma = SHAPE == 0 ? 2.0 * 32 * 32 * 16 : (SHAPE == 1 ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 2);
  const double mfmas = (double)blocks * (threads / 64) * iters * 8.0;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) peak<SHAPE, LDS><<<blocks, threads>>>(seed, out, stamps, iters);
  hipDeviceSynchronize();
  // >= 0.5 s of back-to-back launches so the clock settles; time the second half
  int reps = 50;
  float ms = 0.f;
  for (int round = 0; round < 2; ++round) {
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) peak<SHAPE, LDS><<<blocks, threads>>>(seed, out, stamps, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (round == 0) reps = std::max(20, (int)(500.0f / (ms / reps)));
  }
  std::vector<unsigned long long> h(blocks * 8 * 2);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> clk, cyc;
  for (int i = 0; i < blocks * (threads / 64); ++i) {
    const int blk = i / (threads / 64), w = i % (threads / 64);
    const double c = (double)h[(blk * 8 + w) * 2], r = (double)h[(blk * 8 + w) * 2 + 1];
    if (r > 0) {
      clk.push_back(c / r * 0.1);
      cyc.push_back(c / (iters * 8.0));
    }
  }
  std::sort(clk.begin(), clk.end());
  std::sort(cyc.begin(), cyc.end());
  const double tf = mfmas * flops_per_mfma / (ms / reps * 1e-3) / 1e12;
  printf("%-34s %s  waves/SIMD %d  %8.1f TFLOP/s   %.1f cyc/MFMA/wave   in-kernel clock %.3f GHz\n", name,
         zeros ? "zeros " : "random", waves_per_simd, tf, cyc[cyc.size() / 2], clk[clk.size() / 2]);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

int main() {
  float *seed, *zero, *out;
  unsigned long long* stamps;
  hipMalloc(&seed, 65536 * 4);
  hipMalloc(&zero, 65536 * 4);
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&stamps, 256 * 8 * 2 * 8);
  std::vector<float> h(65536);
  srand(1234);
  for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;   // U(-1,1), like the attention operands
  hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(zero, 0, 65536 * 4);
  run<0, 0>("bf16 32x32x16, register operands", 1, seed, out, stamps, false);
  run<0, 0>("bf16 32x32x16, register operands", 2, seed, out, stamps, false);
  run<0, 1>("bf16 32x32x16, A from LDS", 2, seed, out, stamps, false);
  run<1, 0>("bf16 16x16x32, register operands", 1, seed, out, stamps, false);
  run<1, 0>("bf16 16x16x32, register operands", 2, seed, out, stamps, false);
  run<1, 1>("bf16 16x16x32, A from LDS", 2, seed, out, stamps, false);
  run<0, 0>("bf16 32x32x16, register operands", 2, zero, out, stamps, true);
  run<1, 0>("bf16 16x16x32, register operands", 2, zero, out, stamps, true);
  run<2, 0>("fp32 32x32x2, register operands", 2, seed, out, stamps, false);
  run_mix<0, 0, 0>(2, seed, out, stamps);
  run_mix<0, 1, 0>(2, seed, out, stamps);
  run_mix<0, 2, 0>(2, seed, out, stamps);
  run_mix<0, 0, 1>(2, seed, out, stamps);
  run_mix<0, 1, 1>(2, seed, out, stamps);
  run_mix<0, 2, 1>(2, seed, out, stamps);
  run_mix<0, 0, 2>(2, seed, out, stamps);
  run_mix<0, 1, 1>(1, seed, out, stamps);
  run_mix<1, 0, 0>(2, seed, out, stamps);
  run_mix<1, 1, 0>(2, seed, out, stamps);
  run_mix<1, 1, 1>(2, seed, out, stamps);
  run_mix<1, 2, 1>(2, seed, out, stamps);
  return 0;
}
