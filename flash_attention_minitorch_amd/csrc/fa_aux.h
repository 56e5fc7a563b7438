// Measurement aid (sustained MFMA rate) and layout probes (tests only).
// Part of the kernel set described in fa_kernels.h (included from there, inside its include order).
#pragma once
#include "fa_common.h"

namespace fa {

// ---------------------------------------------------------------------------------------------
// Measurement aid (bench.py "sustained_peak"): a bare v_mfma_f32_32x32x16_bf16 loop on pseudo-random operands in (-1, 1),
// two waves per SIMD on every CU -- what the chip sustains under power on data like the attention operands (SURVEY.md
// section 8d asks for this next to the nominal peak).  Writes per-wave cycles and 100 MHz ticks for the in-kernel clock.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) mfma_peak_kernel(float* __restrict__ sink, unsigned long long* __restrict__ stamps,
                                                        int iters) {
  const int tid = threadIdx.x, lane = tid & 63;
  uint32_t st = 0x9E3779B9u * (uint32_t)(blockIdx.x * 512 + tid + 1);
  auto rnd = [&]() {   // xorshift32 -> uniform in (-1, 1)
    st ^= st << 13; st ^= st >> 17; st ^= st << 5;
    return (float)(int32_t)st * (1.0f / 2147483648.0f);
  };
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) { a[i][j] = (bf16_t)rnd(); b[i][j] = (bf16_t)rnd(); }
  f32x16 c[4];
  for (int i = 0; i < 4; ++i) c[i] = zero16();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 3], b[(u + (u >> 2)) & 3], c[u & 3], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += c[i][j];
  sink[blockIdx.x * 512 + tid] = s;
  if (lane == 0) {
    stamps[(blockIdx.x * 8 + (tid >> 6)) * 2] = t1 - t0;
    stamps[(blockIdx.x * 8 + (tid >> 6)) * 2 + 1] = r1 - r0;
  }
}

// ---------------------------------------------------------------------------------------------
// Scale guard (fa_common.h: guard_skip): the largest squared row norm of q (blockIdx.y = 0) and of k (1), as GUARD_SLOTS partial
// maxima each.  A row is D contiguous bf16 elements in both layouts ([BH][N][d] and [B][N][H][d]), so the kernel sees `rows` rows of
// D elements whatever the layout (zero-padded columns add nothing).  HBM-bound: one pass over both tensors, 16 bytes per lane.
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(256) scale_guard_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, long rows,
                                                          float* __restrict__ guard) {
  constexpr int LPR = D / 8;   // lanes per row, 8 elements each
  static_assert(256 % LPR == 0 && 64 % LPR == 0, "a row's lanes sit in one wave");
  const bf16_t* p = blockIdx.y ? k : q;
  const long total = rows * LPR, stride = (long)gridDim.x * 256;
  float mx = 0.f;
  constexpr int U = 8;   // 16-byte loads in flight per lane: 16 MiB across the chip (4 gave 8 MiB = 4.6 TB/s at HBM latency)
  for (long c0 = (long)blockIdx.x * 256 + threadIdx.x; c0 < total; c0 += U * stride) {
    bf16x8 f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long c = c0 + u * stride;
      f[u] = c < total ? *reinterpret_cast<const bf16x8*>(p + 8 * c) : Atom<bf16_t>::zero();
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) ss = __builtin_fmaf((float)f[u][j], (float)f[u][j], ss);
#pragma unroll
      for (int off = LPR / 2; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
      mx = fmaxf(mx, ss);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) guard[blockIdx.y * GUARD_SLOTS + blockIdx.x] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}

// ---------------------------------------------------------------------------------------------
// Layout probes (tests only): dump what the atoms read so the lane maps are checked against exact data.
// 256 threads stage the tile (as the real kernels do); wave 0 runs the probes.
// ---------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ void __launch_bounds__(256)
probe_kernel(const T* __restrict__ tile_in /*[64][D]*/, const T* __restrict__ b_in /*[32][D]*/,
             float* __restrict__ row_out /*[D/16][64][8]*/, float* __restrict__ tr_out /*[D/32][4][64][8]*/,
             float* __restrict__ mma_out /*[2][64][16]*/, float* __restrict__ swap_out /*[2][64]*/) {
  using A = Atom<T>;
  typedef typename A::frag frag;
  __shared__ __attribute__((aligned(16))) char smem_raw[A::template tile_bytes<D>(64)];
  lds_char* smem = (lds_char*)smem_raw;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  TileStager<T, D, 64, 256> st;
  st.init(tid, D);
  st.load(make_rsrc(tile_in, 64 * D * sizeof(T)), 0);
  st.store(smem);
  __syncthreads();
  if (tid >= 64) return;
  const LaneAddr ra = A::template row_addr<D>(lane);
  const LaneAddr ta = A::template tr_addr<D>(lane);
#pragma unroll
  for (int kc = 0; kc < D / 16; ++kc) {
    frag f = A::template row_frag<D>(smem, ra, 32, kc);  // rows 32..63
    for (int j = 0; j < 8; ++j) row_out[(kc * 64 + lane) * 8 + j] = (float)f[j];
  }
#pragma unroll
  for (int ct = 0; ct < D / 32; ++ct)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      frag f = A::template tr_frag<D>(smem, ta, 16 * s, ct);
      for (int j = 0; j < 8; ++j) tr_out[((ct * 4 + s) * 64 + lane) * 8 + j] = (float)f[j];
    }
  // X = tile[0:32] . b^T  (32 x 32, sum over D); then through pack / tr_frag:
  // Y[c][n] = sum_m tile[m][c] * X[m][n]  for c < 32  (A operand = tr_frag of the tile, B operand = pack(X)).
  f32x16 x = zero16();
#pragma unroll
  for (int kc = 0; kc < D / 16; ++kc)
    A::mma(x, A::template row_frag<D>(smem, ra, 0, kc), A::load_global(b_in + (size_t)r * D + 16 * kc + 8 * h));
  f32x16 y = zero16();
#pragma unroll
  for (int s = 0; s < 2; ++s) A::mma(y, A::template tr_frag<D>(smem, ta, 16 * s, 0), A::pack(x, s));
  for (int i = 0; i < 16; ++i) {
    mma_out[lane * 16 + i] = x[i];
    mma_out[(64 + lane) * 16 + i] = y[i];
  }
  swap_out[lane] = xhalf_max((float)lane);
  swap_out[64 + lane] = xhalf_sum((float)lane);
}


}  // namespace fa
