// FlashAttention forward for MI355X (gfx950), fp32, d = 64, SMALL launches: the keys of a 32-query block are split over the four waves
// of its workgroup.  Part of the kernel set described in fa_kernels.h.
//
// fwd_kernel gives a wave 32 queries and all N keys, so a launch that does not fill the chip takes one wave's whole sweep however few
// workgroups it has (B = 1, H = 8, N = 1024: 64 workgroups, 0.078 ms -- as long as B = 4).  Here a workgroup is ONE 32-query block: wave
// w takes the 32-key tiles w, w + 4, w + 8, ... (interleaved: level under the causal mask too), each through a wave-private pair of LDS
// tiles (no barrier in the sweep), with the classic online softmax of the reference (src/flash_attn_fw.cu:163-245: running maximum,
// O and l rescaled); at the end the four partial (O, l, m) meet in LDS and wave 0 combines them: m = max m_w, weights 2^(c (m_w - m)).
// Four times the workgroups, a quarter of the sweep each.  No key mask, no dropout, fp32 output.
#pragma once
#include "fa_common.h"

namespace fa {

template <int D>
__global__ void __launch_bounds__(256)
fwd_splitk_f32_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, float* __restrict__ o,
                      float* __restrict__ aux_l, float* __restrict__ aux_m, int N, int nqb, int BH, Layout lay, int causal,
                      int aux_mode, float tau) {
  if (guard_skip(lay)) return;
  static_assert(D == 64, "laid out for d = 64");
  using A = Atom<float>;
  typedef typename A::frag frag;
  constexpr int KC = D / 16, DT = D / 32, BN = 32;
  constexpr int TB = A::template tile_bytes<D>(BN);   // 8704 B; also the size of a wave's partial: 64 lanes x 34 words
  __shared__ __attribute__((aligned(16))) char smem_raw[4 * 2 * TB];
  lds_char* smem = (lds_char*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bh, qb;
  map_block(blockIdx.x, BH, nqb, bh, qb);
  const int q0 = qb * 32, qrow = q0 + r;
  const size_t base = head_base(lay, bh);
  const int ld = lay.ld;
  const uint32_t mat_bytes = ((uint32_t)(N - 1) * ld + D) * 4u;
  const rsrc_t qrs = make_rsrc(q + base, mat_bytes), krs = make_rsrc(k + base, mat_bytes), vrs = make_rsrc(v + base, mat_bytes);
  const float c = tau * LOG2E;

  frag qf[KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) qf[kc] = load_frag_buf<float>(qrs, (qrow * ld + 16 * kc + 8 * h) * 4);
  f32x16 acc_o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc_o[dt] = zero16();
  float m_run = -INFINITY, l_run = 0.f;   // raw score units; this lane's partial row sum

  lds_char* tk = smem + w * 2 * TB;
  lds_char* tv = tk + TB;
  const LaneAddr ra = A::template row_addr<D>(lane);
  const LaneAddr ta = A::template tr_addr<D>(lane);
  const int kmax = causal ? min(N, q0 + 32) : N;
  const int nt = (kmax + BN - 1) / BN;
  TileStager<float, D, BN, 64> sk, sv;   // one wave moves its own tiles
  sk.init(lane, ld);
  sv.init(lane, ld);
  int t = w;
  if (t < nt) {
    sk.load(krs, t * BN);
    sv.load(vrs, t * BN);
  }
  for (; t < nt; t += 4) {
    sk.store(tk);
    sv.store(tv);
    if (t + 4 < nt) {   // the wave's next tile is in flight under this one's products
      sk.load(krs, (t + 4) * BN);
      sv.load(vrs, (t + 4) * BN);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the tiles are wave-private: program order, no barrier
    const int kbase = t * BN;
    f32x16 s = zero16();
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) A::mma(s, A::template row_frag<D>(tk, ra, 0, kc), qf[kc]);
    if ((kbase + BN > N) || (causal && kbase + BN - 1 > q0)) {   // wave-uniform
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = kbase + acc_row(i, h);
        if (key >= N || (causal && key > qrow)) s[i] = -INFINITY;
      }
    }
    float mx = s[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
    const float m_new = fmaxf(m_run, xhalf_max(mx));
    const float nm = (m_new == -INFINITY) ? 0.f : -m_new * c;   // (every key so far masked: any finite reference, P = 0)
    const float alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m_run, c, nm));
    float rs = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], c, nm));
      rs += s[i];
    }
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_o[dt][i] *= alpha;
    }
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) A::mma(acc_o[dt], A::template tr_frag<D>(tv, ta, 16 * s2, dt), A::pack(s, s2));
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  const float l_w = xhalf_sum(l_run);

  // the four partials meet in LDS (a wave's 64 x 34 words fit its own K tile), wave 0 combines
  constexpr int PW = 2 * TB;   // bytes from wave to wave; [lane][34]: 32 accumulator registers, m, l
  {
    lds_char* mine = smem + w * PW + lane * 34 * 4;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) *FA_LDS(float, mine + (16 * dt + i) * 4) = acc_o[dt][i];
    *FA_LDS(float, mine + 32 * 4) = m_run;
    *FA_LDS(float, mine + 33 * 4) = l_w;
  }
  __syncthreads();
  if (w != 0) return;
  float m_all = m_run;
#pragma unroll
  for (int u = 1; u < 4; ++u) m_all = fmaxf(m_all, *FA_LDS(float, smem + u * PW + (lane * 34 + 32) * 4));
  // (a causal row always sees key 0 and N >= 1: m_all is finite)
  float wgt = __builtin_amdgcn_exp2f((m_run - m_all) * c);
  float l_tot = l_w * wgt;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc_o[dt][i] *= wgt;
#pragma unroll
  for (int u = 1; u < 4; ++u) {
    lds_char* pu = smem + u * PW + lane * 34 * 4;
    const float mu = *FA_LDS(float, pu + 32 * 4);
    wgt = (mu == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((mu - m_all) * c);   // (a wave without an admissible key)
    l_tot += *FA_LDS(float, pu + 33 * 4) * wgt;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc_o[dt][i] += *FA_LDS(float, pu + (16 * dt + i) * 4) * wgt;
  }
  if (qrow >= N) return;
  const float inv = 1.0f / l_tot;
  const size_t orow = base + (size_t)qrow * ld;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 val = {acc_o[dt][4 * g] * inv, acc_o[dt][4 * g + 1] * inv, acc_o[dt][4 * g + 2] * inv, acc_o[dt][4 * g + 3] * inv};
      *reinterpret_cast<f32x4*>(o + orow + 32 * dt + 8 * g + 4 * h) = val;
    }
  if (h == 0) {
    const size_t ri = (size_t)bh * N + qrow;
    if (aux_mode == AUX_FA1) {   // l = sum exp(tau*s - m), m = tau * rowmax(s)
      aux_l[ri] = l_tot;
      aux_m[ri] = m_all * tau;
    } else {
      aux_l[ri] = m_all * tau + __logf(l_tot);
    }
  }
}

}  // namespace fa
