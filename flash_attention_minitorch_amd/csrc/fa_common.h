// Constants and small helpers shared by the FlashAttention kernels (fa_fwd.h, fa_bwd_dkdv.h, fa_bwd_dq.h, fa_aux.h);
// the overview of the kernel set is in fa_kernels.h.
#pragma once
#include <type_traits>

#include "fa_atoms.h"

namespace fa {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int AUX_FA1 = 1;  // l = sum exp(s - m), m = row max            (src/flash_attn_fw.cu:259-276)
constexpr int AUX_FA2 = 2;  // l = logsumexp, m untouched                  (src/flash_attn2_fw.cu:279-294)

template <int V> using ic = std::integral_constant<int, V>;

// Diagnostic builds only (MODE == 9 instantiation of the dK/dV kernel): per-wave cycle totals per loop phase,
// written to a buffer of their own that no other code reads.  The real kernels execute no stamp.
__device__ unsigned long long g_phase_cycles[8 * 8192];
// The lane id, recomputed on the spot (two VALU instructions the compiler can neither hoist nor keep alive across a loop): lane
// constants that are only needed in front of and behind a register-starved pipeline are re-derived instead of being carried (and
// spilled) through it (cdna_hip_programming.md, Appendix B: 'recompute per block (v_mbcnt)').
FA_DEV int lane_fresh() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

FA_DEV unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

template <typename T> FA_DEV typename Atom<T>::frag load_frag_buf(rsrc_t rs, int byte_off);
template <> FA_DEV bf16x8 load_frag_buf<bf16_t>(rsrc_t rs, int byte_off) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
}
template <> FA_DEV f32x8 load_frag_buf<float>(rsrc_t rs, int byte_off) {
  f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
  f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off + 16, 0, 0));
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Scale guard.  The MFMA-slot kernels fold c = tau*log2(e) into one bf16 operand (fa_kernels.h, "Scaling"): one more 2^-9 relative
// rounding of every q_d (k_d), which moves a score (log2 units) by sum_d c q_d k_d eps_d, root-sum-square estimate
// 2^-9 / sqrt(3) * c * |q| |k|.  scale_guard_kernel reduces the largest squared row norms of q and of k of a call to GUARD_SLOTS
// partial maxima each; a kernel launched under the guard reads them on entry (two 16-byte loads per lane, L2 hits), and returns at
// once unless the estimate for the call's largest rows is on its side of the budget (Layout::guard_want: 0 = within, 1 = beyond).
// Everything is wave-uniform after the reduction; a launch without a guard (Layout::guard == nullptr) pays one scalar branch.
constexpr int GUARD_SLOTS = 256;
FA_DEV bool guard_beyond(const Layout& L) {   // the call's operands are beyond the folded scale's budget (wave-uniform)
  const int lane = threadIdx.x & 63;
  const f32x4 a = *reinterpret_cast<const f32x4*>(L.guard + 4 * lane);
  const f32x4 b = *reinterpret_cast<const f32x4*>(L.guard + GUARD_SLOTS + 4 * lane);
  float qm = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), km = fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    qm = fmaxf(qm, __shfl_xor(qm, off));
    km = fmaxf(km, __shfl_xor(km, off));
  }
  const bool beyond = !(L.guard_coef * __builtin_sqrtf(qm * km) <= 1.0f);   // (NaN / Inf inputs count as beyond)
  return __builtin_amdgcn_readfirstlane((int)beyond) != 0;
}
FA_DEV bool guard_skip(const Layout& L) {
  // (want 2: this launch PRODUCES the guard, see guard_produce; want 3: it takes both sides itself, see scale_exact)
  if (L.guard == nullptr || L.guard_want >= 2) return false;
  return guard_beyond(L) != (L.guard_want != 0);
}
// The backward slot kernels: which copy of the sweep does this launch run?  (Layout::scale_sel)
FA_DEV bool scale_exact(const Layout& L) {
  if (L.scale_sel != 2 || L.guard == nullptr) return L.scale_sel != 0;
  return guard_beyond(L);
}

// The forward can produce the guard itself instead of a separate pass over q and k (Layout::guard_want == 2, a zero-filled guard):
// every wave of a mask-free slot forward holds the fragments of its 32 query rows anyway and loads the 32 KEY rows of the same
// indices beside them (the query blocks of a head cover 0..N-1, so the key rows of a head are covered exactly once as well); the
// wave's largest squared row norms go to slot (blockIdx & 255) with an atomic max on the float bits (non-negative floats order like
// unsigned integers; a NaN row reads as "beyond the budget").  Optimistic: the launch itself runs with the folded scale, and its
// fp32-scaling twin, launched behind it, redoes the call if the finished guard says so.  qs / ks: the lane's half-row sums of squares.
// (512-thread workgroups: eight waves.)
// Returns true when the WORKGROUP's own rows (scratch: 64 bytes of its LDS, two barriers) are already beyond the budget: the call's
// maxima can only be larger, the twin will redo the call, and the whole workgroup leaves at once instead of sweeping for nothing
// (operands that are large throughout, the usual case, cost the optimistic launch a few microseconds, not a forward).
FA_DEV bool guard_produce(const Layout& L, float qs, float ks, lds_char* scratch) {
  qs = xhalf_sum(qs);
  ks = xhalf_sum(ks);
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) {
    qs = fmaxf(qs, __shfl_xor(qs, off));
    ks = fmaxf(ks, __shfl_xor(ks, off));
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    unsigned* g = reinterpret_cast<unsigned*>(const_cast<float*>(L.guard)) + (blockIdx.x & (GUARD_SLOTS - 1));
    atomicMax(g, __float_as_uint(qs));
    atomicMax(g + GUARD_SLOTS, __float_as_uint(ks));
    *FA_LDS(float, scratch + 4 * w) = qs;
    *FA_LDS(float, scratch + 32 + 4 * w) = ks;
  }
  __syncthreads();
  float qm = 0.f, km = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    qm = fmaxf(qm, *FA_LDS(float, scratch + 4 * i));
    km = fmaxf(km, *FA_LDS(float, scratch + 32 + 4 * i));
  }
  __syncthreads();   // (the scratch is part of the stage ring: a later stage's LDS-DMA lands there)
  return !(L.guard_coef * __builtin_sqrtf(qm * km) <= 1.0f);
}
template <typename F> FA_DEV float frag_sumsq(const F& f) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s = __builtin_fmaf((float)f[j], (float)f[j], s);
  return s;
}

// Forward epilogue: 4 consecutive columns of one output row, fp32 (the default and the parity path) or bf16 (Layout::out_bf16: one
// rounding of the fp32 result, 2^-9 relative: for consumers that want a bf16 activation, e.g. the sharded gather of configs[4]).
FA_DEV void store_out4(float* o, size_t elem_off, const f32x4& val, int out_bf16) {
  if (out_bf16) {
    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(o) + elem_off) = __builtin_convertvector(val, bf16x4);
  } else {
    *reinterpret_cast<f32x4*>(o + elem_off) = val;
  }
}

FA_DEV f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

}  // namespace fa
