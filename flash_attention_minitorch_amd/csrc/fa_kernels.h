// FlashAttention forward / backward kernels for MI355X (gfx950), written against fa_atoms.h.
//
// Replaces the device kernels of the reference (results, not mechanism):
//   flash_attn_fw<T>  FA-1  src/flash_attn_fw.cu:22-287      flash_attn_bw<T>  FA-1  src/flash_attn_bw.cu:20-261
//   flash_attn_fw<T>  FA-2  src/flash_attn2_fw.cu:22-297     flash_attn_bw<T>  FA-2  src/flash_attn2_bw.cu:20-263
//
// Layout everywhere: row-major contiguous [BH][N][D] for q,k,v,o,dO,dq,dk,dv and [BH][N] for row statistics
// (SURVEY.md section 8).  tau = sqrt(1/D) (src/flash_attn_fw.cu:37).
//
// Forward (one kernel for FA-1 and FA-2 side outputs): a workgroup = 4 waves = 128 query rows, each wave 32 rows.
// Q fragments stay in registers; K/V tiles of BN keys are staged through LDS (double buffered, loads issued before
// the MFMA phase and written after it).  S^T = K Q^T is computed with the QUERY on the lane, so a lane owns whole
// (half) rows of the softmax: row max / row sum are register reductions plus one v_permlane32_swap, and the
// exponentiated tile is directly the B operand of O^T += V^T P^T (no LDS round trip for P).
//
// Backward = preprocess (delta = rowsum(dO*O), -L*log2e) + a key-stationary dK/dV kernel (S, dP with the KEY on
// the lane; P and dS feed dV^T += dO^T P and dK^T += Q^T dS from registers) + a query-stationary dQ kernel
// (S^T, dP^T with the query on the lane; dQ^T += K^T dS^T).  No atomics: results are bitwise reproducible
// (the reference's FA-2 backward uses atomicAdd for dQ, src/flash_attn2_bw.cu:228).
//
// Tile loops are unrolled by two so the LDS double-buffer index is a compile-time constant: every LDS address is
// a per-lane register computed once plus an instruction immediate, and every global tile load is a buffer load
// whose tile offset is a scalar operand (out-of-range rows read as zero) -- no address VALU inside the loops.
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

FA_DEV f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// ---------------------------------------------------------------------------------------------
// Forward.  P = exp2(c*s - c*m_ref) with c = tau*log2(e) applied in fp32 (one fma per score: pre-scaling Q or K
// in bf16 was measured to cost up to 3.7e-3 max-abs on O at small N -- the rounding is the same for every key of a
// row, so it does not average out).  m_ref is a per-row REFERENCE, not the running maximum: it is only moved
// (O, l rescaled) when some P of the row would exceed 2^6, which fp32 / bf16 hold at full relative precision; the
// steady state computes neither a row maximum nor a rescale (time ~ MFMA + VALU on this chip: they barely co-issue).
// Row sums stay on the VALU in fp32: summing the bf16-rounded P on the MFMA (ones . P^T) was measured 4 % faster
// but puts P's 2^-9 quantisation into L = m + log(l), which the backward then exponentiates (dV error 2.7e-3 on
// causal rows with few keys).
// ---------------------------------------------------------------------------------------------
constexpr float MAX_DEFER_SUM = 64.0f;   // 2^6: bound on a lane's partial row sum (hence on every P) in the steady state

template <typename T, int D, int BN, int WPE, int FEAT = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE)))
fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, float* __restrict__ o,
           float* __restrict__ aux_l, float* __restrict__ aux_m, int N, int nqb, int BH, Layout lay, int causal,
           int aux_mode, float tau) {
  using A = Atom<T>;
  typedef typename A::frag frag;
  constexpr bool HM = FEAT >= 1, HD = FEAT >= 2;   // key mask (staged as zeros when absent); dropout
  constexpr int KC = D / 16, KT = BN / 32, DT = D / 32;
  constexpr int TB = A::template tile_bytes<D>(BN);
  __shared__ __attribute__((aligned(16))) char smem_raw[4 * TB];
  __shared__ __attribute__((aligned(16))) float smask[HM ? 2 * BN : 4];   // key mask / tau of the two tiles in flight
  lds_char* smem = (lds_char*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches below
  int bh, qb;
  map_block(blockIdx.x, BH, nqb, bh, qb);
  if (causal) qb = nqb - 1 - qb;  // heaviest query blocks first
  const int q0 = qb * 128 + w * 32, qrow = q0 + r;
  const bool qvalid = qrow < N;
  const size_t base = head_base(lay, bh);
  const int ld = lay.ld;   // elements between consecutive rows
  const uint32_t mat_bytes = ((uint32_t)(N - 1) * ld + D) * (uint32_t)sizeof(T);
  const rsrc_t qrs = make_rsrc(q + base, mat_bytes);
  const rsrc_t krs = make_rsrc(k + base, mat_bytes);
  const rsrc_t vrs = make_rsrc(v + base, mat_bytes);
  const float c = tau * LOG2E;

  frag qf[KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc)
    qf[kc] = load_frag_buf<T>(qrs, (qrow * ld + 16 * kc + 8 * h) * (int)sizeof(T));

  f32x16 acc_o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc_o[dt] = zero16();
  float m_ref = 0.f, nmc = 0.f, m_true = -INFINITY, l_run = 0.f;   // raw score units; nmc = -m_ref * c

  const LaneAddr ra = A::template row_addr<D>(lane);
  const LaneAddr ta = A::template tr_addr<D>(lane);
  const int kmax = causal ? min(N, qb * 128 + 128) : N;
  const int nt = (kmax + BN - 1) / BN;
  TileStager<T, D, BN, 256> sk, sv;
  sk.init(tid, ld);
  sv.init(tid, ld);
  sk.load(krs, 0);
  sv.load(vrs, 0);
  sk.store(smem);
  sv.store(smem + 2 * TB);
  // additive key mask, staged per tile in raw score units (mask / tau) so that it enters S^T as the accumulator input
  const float* mrow = (HM && lay.kmask) ? lay.kmask + (size_t)(bh / lay.mask_heads) * N : nullptr;
  const uint32_t dbase = HD ? drop_base(lay, bh, qrow) : 0u;
  const float inv_tau = 1.0f / tau;
  float mreg = 0.f;
  auto mask_load = [&](int kb0) {
    if constexpr (HM) {
      if (tid < BN) mreg = (mrow != nullptr && kb0 + tid < N) ? mrow[kb0 + tid] * inv_tau : 0.f;
    }
  };
  auto mask_store = [&](int par) {
    if constexpr (HM) {
      if (tid < BN) smask[par * BN + tid] = mreg;
    }
  };
  mask_load(0);
  mask_store(0);
  __syncthreads();

  auto tile = [&](auto par, auto first_c, int t) {
    constexpr int PAR = decltype(par)::value;
    constexpr bool FIRST = decltype(first_c)::value != 0;
    const int kbase = t * BN;
    const bool more = t + 1 < nt;
    if (more) {
      sk.load(krs, kbase + BN);
      sv.load(vrs, kbase + BN);
      mask_load(kbase + BN);
    }
    lds_char* tk = smem + PAR * TB;
    lds_char* tv = smem + (2 + PAR) * TB;
    const bool active = !causal || kbase <= q0 + 31;  // wave-uniform
    if (active) {
      f32x16 s[KT];
      const bool need_mask = (kbase + BN > N) || (causal && kbase + BN - 1 > q0);  // wave-uniform
      auto scores = [&]() {   // S^T tile of this wave (raw units), masked
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          s[kt] = zero16();
          if constexpr (HM) {   // register i of lane half h is key 32*kt + acc_row(i, h): four aligned float4 reads
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 mk = *reinterpret_cast<const f32x4*>(&smask[PAR * BN + 32 * kt + 8 * g + 4 * h]);
#pragma unroll
              for (int j = 0; j < 4; ++j) s[kt][4 * g + j] = mk[j];
            }
          }
#pragma unroll
          for (int kc = 0; kc < KC; ++kc) A::mma(s[kt], A::template row_frag<D>(tk, ra, 32 * kt, kc), qf[kc]);
        }
        if (need_mask) {
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int key = kbase + 32 * kt + acc_row(i, h);
              if (key >= N || (causal && key > qrow)) s[kt][i] = -INFINITY;
            }
        }
      };
      auto tile_max = [&]() {   // row maximum of this tile (raw score units)
        float mx = s[0][0];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kt][i]);
        return xhalf_max(mx);
      };
      auto exps = [&]() {       // s <- P = exp2(c*s - c*m_ref); returns this lane's partial row sum
        float rowsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][i], c, nmc));
            s[kt][i] = p;
            rowsum += p;
          }
        return rowsum;
      };
      scores();
      float rowsum, alpha = 1.0f;
      if (FIRST) {                      // the first tile sets the reference to its row maximum
        m_ref = tile_max();
        m_true = m_ref;
        if (HM && m_ref == -INFINITY) m_ref = 0.f;   // every key of the first tile masked: any finite reference will do
        nmc = -m_ref * c;
        rowsum = exps();
      } else {
        if (aux_mode == AUX_FA1) m_true = fmaxf(m_true, tile_max());   // only FA-1 reports the true row maximum
        // Steady state: no maximum at all.  P is computed against the current reference; a lane whose partial row
        // sum stays under 2^MAX_DEFER cannot hold a P above it.  Otherwise (rare: some row outgrew its reference)
        // the tile is redone the classic way: scores again, true maximum, reference moved, O and l rescaled.
        rowsum = exps();
        if (__any(!(rowsum < MAX_DEFER_SUM))) {
          scores();
          const float delta = fmaxf(tile_max() - m_ref, 0.f);
          alpha = __builtin_amdgcn_exp2f(-delta * c);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_o[dt][i] *= alpha;
          m_ref += delta;
          nmc = -m_ref * c;
          rowsum = exps();
        }
      }
      l_run = l_run * alpha + rowsum;
      if constexpr (HD) {   // dropout acts on the normalised probabilities: after the row sum, before P.V
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            s[kt][i] = drop_keep(dbase, kbase + 32 * kt + acc_row(i, h), lay.drop_thr) ? s[kt][i] * lay.drop_scale : 0.f;
      }
      frag pf[KT][2];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        pf[kt][0] = A::pack(s[kt], 0);
        pf[kt][1] = A::pack(s[kt], 1);
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
            A::mma(acc_o[dt], A::template tr_frag<D>(tv, ta, 32 * kt + 16 * s2, dt), pf[kt][s2]);
    }
    if (more) {
      sk.store(smem + (PAR ^ 1) * TB);
      sv.store(smem + (2 + (PAR ^ 1)) * TB);
      mask_store(PAR ^ 1);
    }
    __syncthreads();
  };
  tile(ic<0>{}, ic<1>{}, 0);
  int t = 1;
  for (; t + 1 < nt; t += 2) {
    tile(ic<1>{}, ic<0>{}, t);
    tile(ic<0>{}, ic<0>{}, t + 1);
  }
  if (t < nt) tile(ic<1>{}, ic<0>{}, t);

  const float l_tot = xhalf_sum(l_run);   // sum of exp2(c*(s - m_ref))
  // a row whose every key is masked has l = 0: it returns O = 0 and L = -inf (and zero gradients in the backward)
  const float inv = (HM && !(l_tot > 0.f)) ? 0.f : 1.0f / l_tot;
  if (qvalid) {
    float* orow = o + base + (size_t)qrow * ld;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 val = {acc_o[dt][4 * g] * inv, acc_o[dt][4 * g + 1] * inv, acc_o[dt][4 * g + 2] * inv,
                     acc_o[dt][4 * g + 3] * inv};
        *reinterpret_cast<f32x4*>(orow + 32 * dt + 8 * g + 4 * h) = val;
      }
    if (h == 0) {
      const size_t ri = (size_t)bh * N + qrow;
      if (aux_mode == AUX_FA1) {   // l = sum exp(tau*s - m), m = tau * rowmax(s)
        aux_l[ri] = (HM && !(l_tot > 0.f)) ? 0.f : l_tot * __builtin_amdgcn_exp2f((m_ref - m_true) * c);
        aux_m[ri] = m_true * tau;
      } else {
        aux_l[ri] = m_ref * tau + __logf(l_tot);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Forward, slot-interleaved (bf16, d = 64 or 128, FA-2 side output): a workgroup = 8 waves = 256 query rows (two waves
// per SIMD), query on the lane as above.  K / V arrive by LDS-DMA in 16 KiB stages (128 keys at d = 64, 64 at d = 128;
// K and V rings of R slots, V above K) and are consumed as 32-key sub-tiles by a three-deep software pipeline of MFMA
// slots (see the dK/dV kernel):
//   period j = 2*KC slots (KC = d/16):   first KC slots  S^T(j+1) = K Q^T        last KC slots  O^T += V^T P^T(j-1)
// with the fma / exp / add / pack of sub-tile j spread over all of them.  At d = 64 the softmax is 36 issue cycles per
// slot against the 24 an MFMA leaves free (VALU-issue bound by construction, the slots make the MFMAs disappear under
// it); at d = 128 it is 18.  Reference handling as in fwd_kernel: the first sub-tile sets the per-row reference to its
// row maximum; afterwards P is computed against the reference with no maximum, and a lane whose partial row sum
// reaches 2^6 (rare) makes the wave redo that sub-tile the classic way (scores again from LDS, true maximum,
// reference moved, O and l rescaled) at the end of its period, before its P.V is issued.
// Stage hand-off: the barrier that publishes stage s+1 sits NSUBT-2 periods into stage s (the rows of a sub-tile are
// first requested two periods ahead).  With four sub-tiles per stage (d = 64) the DMA of stage s+1 is issued at the
// top of stage s into a three-slot ring; with two (d = 128) the barrier is at the top of the stage, the DMA of stage
// s+2 follows it, and the ring has four slots (a stage's V is still read one period into the next stage).
// ---------------------------------------------------------------------------------------------
// MASKS = false: the caller guarantees a non-causal launch with N a multiple of the stage (no sub-tile ever needs a mask),
// which removes the masked period variants and their register pressure at the joins (needed at d = 128).
template <typename T, int D, bool MASKS = true, int DIAG = 0>
__global__ void __launch_bounds__(512)
fwd_slot_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, float* __restrict__ o,
                float* __restrict__ aux_l, int N, int nqb, int BH, Layout lay, int causal, float tau) {
  static_assert((D == 64 || D == 128) && sizeof(T) == 2, "slot schedule is laid out for bf16, d = 64 / 128");
  using A = Atom<T>;
  typedef typename A::frag frag;
  constexpr int KC = D / 16, DT = D / 32, NS = 2 * KC, EPS = 16 / NS;   // slots per period, scores per slot
  constexpr int ST = 8192 / D;                        // keys per stage: 16 KiB of K and of V
  constexpr int NSUBT = ST / 32;                      // sub-tiles per stage: 4 (d = 64) or 2 (d = 128)
  constexpr int R = NSUBT == 2 ? 4 : 3;               // ring slots
  constexpr int TB = A::template tile_bytes<D>(ST);   // 16 KiB
  constexpr int VOFF = R * TB;
  constexpr int SUBB = (D / 32) * 512 * 4;            // bytes of one 32-key sub-tile inside a stage image
  static_assert(TB == 16384 && 2 * DT == KC, "stage geometry");
  __shared__ __attribute__((aligned(16))) char smem_raw[2 * R * TB];
  lds_char* smem = (lds_char*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bh, qb;
  map_block(blockIdx.x, BH, nqb, bh, qb);
  if (causal) qb = nqb - 1 - qb;
  const int q0 = qb * 256 + w * 32, qrow = q0 + r;
  const bool qvalid = qrow < N;
  const size_t base = head_base(lay, bh);
  const int ld = lay.ld;
  const uint32_t mat_bytes = ((uint32_t)(N - 1) * ld + D) * (uint32_t)sizeof(T);
  const rsrc_t qrs = make_rsrc(q + base, mat_bytes);
  const raw_rsrc_t kraw = make_raw_rsrc(k + base, mat_bytes), vraw = make_raw_rsrc(v + base, mat_bytes);
  const float c = tau * LOG2E;

  frag qf[KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) qf[kc] = load_frag_buf<T>(qrs, (qrow * ld + 16 * kc + 8 * h) * (int)sizeof(T));
  f32x16 acc_o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc_o[dt] = zero16();
  float m_ref = 0.f, nmc = 0.f, l_run = 0.f;

  const LaneAddr ra = A::template row_addr<D>(lane);
  const LaneAddr ta = A::template tr_addr<D>(lane);
  const int kmax = causal ? min(N, qb * 256 + 256) : N;
  const int nstage = (kmax + ST - 1) / ST;
  const uint32_t smem_addr = (uint32_t)(uintptr_t)smem;
  // LDS-DMA pieces of 1 KiB: d = 64: one 8-row group (piece = w, w + 8); d = 128: half of one (piece = 2 * group + half).
  // The image's chunk swizzle is applied to each lane's SOURCE address; a wave's pieces share its parity, hence one offset.
  constexpr int PPG = D / 64;   // pieces per 8-row group
  const int dma_row7 = (lane >> 2) & 7;
  const int dma_gpar = (PPG == 1) ? (w & 1) : ((w >> 1) & 1);
  const int dma_half = (PPG == 1) ? 0 : (w & 1);
  const int dma_voff = dma_row7 * ld * (int)sizeof(T) +
                       16 * (4 * (2 * dma_half + (lane >> 5)) + ((lane & 3) ^ ((2 * dma_gpar + (dma_row7 >> 2)) & 3)));
  auto stage_dma = [&](int row0, int slot_base) {
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
      const int piece = w + 8 * g2, g = piece / PPG;
      const int soff = (row0 + 8 * g) * ld * (int)sizeof(T);
      dma16(kraw, smem_addr + slot_base + 1024 * piece, dma_voff, soff);
      dma16(vraw, smem_addr + slot_base + VOFF + 1024 * piece, dma_voff, soff);
    }
  };
  auto slot_of = [&](int st) { return (st % R) * TB; };
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, k_t0 = 0, k_r0 = 0, t0 = 0, t1 = 0;
  if constexpr (DIAG) {
    k_t0 = stamp();
    k_r0 = __builtin_amdgcn_s_memrealtime();
  }
  if constexpr (MASKS) {   // ragged launches read stage rows past N: make sure they are zeros whatever an out-of-range
    // LDS-DMA lane does (0 * stale NaN bits would poison P.V); 96 / 128 KiB once per workgroup
#pragma unroll 4
    for (int off = tid * 16; off < 2 * R * TB; off += 512 * 16) *FA_LDS(u32x4, smem + off) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
  }
  stage_dma(0, 0);
  if (NSUBT == 2 && nstage > 1) stage_dma(ST, slot_of(1));
  dma_wait_all();
  __syncthreads();
  if constexpr (DIAG) { t0 = stamp(); ph[0] += t0 - k_t0; }

  f32x16 sA, sB;
  u32x4 pA0, pA1, pB0, pB1;   // packed P^T (bf16 pairs): chunks s2 = 0, 1 of the two sub-tiles in flight
  frag rk[4], tf[4];          // rings: K rows of the S^T chain, transposed V of the P.V chain (requested two slots ahead)
  auto SB = [&]() { __builtin_amdgcn_sched_barrier(0); };
  auto krow = [&](int b0, int b1, int sub, int kc) -> frag {
    return *FA_LDS(frag, smem + ((kc & 1) ? b1 : b0) + SUBB * sub + 512 * (kc >> 1));
  };
  auto vtr = [&](int b0, int b1, int sub, int s2, int dt) -> frag {
    const int kk = VOFF + SUBB * sub + (D / 32) * 512 * (2 * s2) + 512 * dt;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(FA_LDS(bf16x4, smem + b0 + kk));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(FA_LDS(bf16x4, smem + b1 + kk + (D / 32) * 512));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto cvt2 = [&](float a, float b) -> uint32_t {
    f32x2 pr = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, bf16x2));
  };
  auto mask_scores = [&](f32x16& x, int kcur) {   // raw scores of keys beyond N or (causal) beyond the query: -inf
    const int klim = causal ? (min(qrow, N - 1) - kcur) : (N - 1 - kcur);
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (acc_row(i, h) > klim) x[i] = -INFINITY;
  };
  auto tile_max = [&](const f32x16& x) {
    float mx = x[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, x[i]);
    return xhalf_max(mx);
  };
  // One period.  SUBN / rn*: sub-tile whose S^T is produced (its rows 2.. are requested here, rows 0, 1 were requested by
  // the period before); SUBP / tp*: sub-tile whose P.V is issued; SUB2 / r2*: the sub-tile two ahead (rows 0, 1 requested in
  // the last two slots); SUBC / rc* / kcur: the sub-tile in the softmax stream.
  auto period = [&](auto hn_c, auto hc_c, auto hp_c, auto mask_c, auto subn_c, auto subp_c, auto sub2_c, auto subc_c,
                    int rn0, int rn1, int tp0, int tp1, int r20, int r21, int rc0, int rc1, int kcur, f32x16& ns,
                    f32x16& cs, u32x4& pp0, u32x4& pp1, u32x4& pc0, u32x4& pc1) {
    constexpr bool HN = decltype(hn_c)::value != 0, HC = decltype(hc_c)::value != 0, HP = decltype(hp_c)::value != 0;
    constexpr bool MASK = decltype(mask_c)::value != 0;
    constexpr int SUBN = decltype(subn_c)::value, SUBP = decltype(subp_c)::value, SUB2 = decltype(sub2_c)::value;
    constexpr int SUBC = decltype(subc_c)::value;
    float rs = 0.f, cm = c;
    if constexpr (MASK) {
      asm volatile("" : "+v"(cm));   // keeps hipcc from hoisting the two variants' common fma out of the branch
      if constexpr (HC) mask_scores(cs, kcur);
    }
    auto fe = [&](int i) {
      const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(cs[i], cm, nmc));
      cs[i] = pv;
      rs += pv;
    };
    // softmax work of slot g: its EPS scores, then the bf16 pack of the pairs completed by the slot before (the last
    // slot also packs its own)
    auto valu = [&](int g) {
      if constexpr (HC) {
#pragma unroll
        for (int e = 0; e < EPS; ++e) fe(g * EPS + e);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int done_at = (2 * p + 1) / EPS;   // slot that finishes pair p
          if (done_at == g - 1 || (g == NS - 1 && done_at == g)) {
            const uint32_t pk = cvt2(cs[2 * p], cs[2 * p + 1]);
            if (p < 4) pc0[p] = pk;
            else pc1[p - 4] = pk;
          }
        }
      }
    };
#pragma unroll
    for (int kq = 0; kq < KC; ++kq) {   // S^T of the next sub-tile
      if constexpr (HN) {
        if (kq == 0) A::mma_c(ns, rk[0], qf[0], zero16());
        else A::mma(ns, rk[kq & 3], qf[kq]);
        SB();   // the MFMA opens its slot; the fillers follow in its shadow
        if (kq + 2 < KC) rk[(kq + 2) & 3] = krow(rn0, rn1, SUBN, kq + 2);
      }
      valu(kq);
      if constexpr (HP) {
        if (kq >= KC - 2) tf[kq - (KC - 2)] = vtr(tp0, tp1, SUBP, 0, kq - (KC - 2));
      }
      SB();
    }
#pragma unroll
    for (int t = 0; t < KC; ++t) {   // P.V of the previous sub-tile: chunk s2 = t / DT of its keys, columns 32 * (t % DT)
      if constexpr (HP) {
        A::mma(acc_o[t % DT], tf[t & 3], __builtin_bit_cast(frag, (t < DT) ? pp0 : pp1));
        SB();
        if (t + 2 < KC) tf[(t + 2) & 3] = vtr(tp0, tp1, SUBP, (t + 2) / DT, (t + 2) % DT);
      }
      valu(KC + t);
      if constexpr (HN) {
        if (t >= KC - 2) rk[t - (KC - 2)] = krow(r20, r21, SUB2, t - (KC - 2));
      }
      SB();
    }
    if constexpr (HC) {
      float alpha = 1.0f;
      if (__any(!(rs < MAX_DEFER_SUM))) {   // rare: some row outgrew its reference -> redo this sub-tile the classic way
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const frag kk = krow(rc0, rc1, SUBC, kc);
          if (kc == 0) A::mma_c(cs, kk, qf[0], zero16());
          else A::mma(cs, kk, qf[kc]);
        }
        if constexpr (MASK) mask_scores(cs, kcur);
        const float delta = fmaxf(tile_max(cs) - m_ref, 0.f);
        alpha = __builtin_amdgcn_exp2f(-delta * c);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc_o[dt][i] *= alpha;
        m_ref += delta;
        nmc = -m_ref * c;
        rs = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          cs[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(cs[i], c, nmc));
          rs += cs[i];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pc0[j] = cvt2(cs[2 * j], cs[2 * j + 1]);
          pc1[j] = cvt2(cs[8 + 2 * j], cs[9 + 2 * j]);
        }
      }
      l_run = l_run * alpha + rs;
    }
  };
  auto T1 = ic<1>{};
  auto T0 = ic<0>{};
  int cr0 = ra.b[0], cr1 = ra.b[1];   // row addresses of the current stage (slot 0)
  int ct0 = ta.b[0], ct1 = ta.b[1];   // transposed-read addresses of the current stage
  int pt0 = ct0, pt1 = ct1;           // ... of the previous stage (stage 0: any finite data, P = 0)
  // prologue: S^T of sub-tile 0, whose row maximum becomes the reference
  rk[0] = krow(cr0, cr1, 0, 0);
  rk[1] = krow(cr0, cr1, 0, 1);
  pB0 = pB1 = pA0 = pA1 = u32x4{0u, 0u, 0u, 0u};
  SB();
  period(T1, T0, T0, T0, ic<0>{}, ic<0>{}, ic<1>{}, ic<0>{}, cr0, cr1, ct0, ct1, cr0, cr1, cr0, cr1, 0, sA, sB, pB0, pB1, pA0, pA1);
  {
    const bool m0 = MASKS && ((31 >= N) || (causal && 31 > q0));
    if (m0) mask_scores(sA, 0);
    m_ref = tile_max(sA);      // key 0 is never masked, so the maximum is finite
    nmc = -m_ref * c;
  }
  for (int st = 0; st < nstage; ++st) {
    const int nb = slot_of(st + 1);
    const int nr0 = ra.b[0] + nb, nr1 = ra.b[1] + nb;
    const int kb = st * ST;
    // a sub-tile needs the mask when it crosses N or (causal) this wave's first query; wave-uniform
    auto need = [&](int sub) { return MASKS && ((kb + 32 * sub + 31 >= N) || (causal && kb + 32 * sub + 31 > q0)); };
    if constexpr (NSUBT == 4) {
      if (st + 1 < nstage) stage_dma((st + 1) * ST, nb);
      // period 4st+0: produce sub 1, softmax of sub 0, P.V of sub 3 of the previous stage
      if constexpr (MASKS) {
        if (need(0)) period(T1, T1, T1, T1, ic<1>{}, ic<3>{}, ic<2>{}, ic<0>{}, cr0, cr1, pt0, pt1, cr0, cr1, cr0, cr1, kb, sB, sA, pB0, pB1, pA0, pA1);
        else period(T1, T1, T1, T0, ic<1>{}, ic<3>{}, ic<2>{}, ic<0>{}, cr0, cr1, pt0, pt1, cr0, cr1, cr0, cr1, kb, sB, sA, pB0, pB1, pA0, pA1);
      } else {
        period(T1, T1, T1, T0, ic<1>{}, ic<3>{}, ic<2>{}, ic<0>{}, cr0, cr1, pt0, pt1, cr0, cr1, cr0, cr1, kb, sB, sA, pB0, pB1, pA0, pA1);
      }
      // period 4st+1: produce sub 2, softmax of sub 1, P.V of sub 0
      if constexpr (MASKS) {
        if (need(1)) period(T1, T1, T1, T1, ic<2>{}, ic<0>{}, ic<3>{}, ic<1>{}, cr0, cr1, ct0, ct1, cr0, cr1, cr0, cr1, kb + 32, sA, sB, pA0, pA1, pB0, pB1);
        else period(T1, T1, T1, T0, ic<2>{}, ic<0>{}, ic<3>{}, ic<1>{}, cr0, cr1, ct0, ct1, cr0, cr1, cr0, cr1, kb + 32, sA, sB, pA0, pA1, pB0, pB1);
      } else {
        period(T1, T1, T1, T0, ic<2>{}, ic<0>{}, ic<3>{}, ic<1>{}, cr0, cr1, ct0, ct1, cr0, cr1, cr0, cr1, kb + 32, sA, sB, pA0, pA1, pB0, pB1);
      }
      if constexpr (DIAG) { t1 = stamp(); ph[1] += t1 - t0; }
      dma_wait_all();   // this wave's pieces of the next stage have landed
      if constexpr (DIAG) { t0 = stamp(); ph[2] += t0 - t1; }
      __syncthreads();
      if constexpr (DIAG) { t1 = stamp(); ph[3] += t1 - t0; t0 = t1; }
      // period 4st+2: produce sub 3, softmax of sub 2, P.V of sub 1; rows two ahead = sub 0 of the next stage
      if constexpr (MASKS) {
        if (need(2)) period(T1, T1, T1, T1, ic<3>{}, ic<1>{}, ic<0>{}, ic<2>{}, cr0, cr1, ct0, ct1, nr0, nr1, cr0, cr1, kb + 64, sB, sA, pB0, pB1, pA0, pA1);
        else period(T1, T1, T1, T0, ic<3>{}, ic<1>{}, ic<0>{}, ic<2>{}, cr0, cr1, ct0, ct1, nr0, nr1, cr0, cr1, kb + 64, sB, sA, pB0, pB1, pA0, pA1);
      } else {
        period(T1, T1, T1, T0, ic<3>{}, ic<1>{}, ic<0>{}, ic<2>{}, cr0, cr1, ct0, ct1, nr0, nr1, cr0, cr1, kb + 64, sB, sA, pB0, pB1, pA0, pA1);
      }
      // period 4st+3: produce sub 0 of the next stage, softmax of sub 3, P.V of sub 2
      if constexpr (MASKS) {
        if (need(3)) period(T1, T1, T1, T1, ic<0>{}, ic<2>{}, ic<1>{}, ic<3>{}, nr0, nr1, ct0, ct1, nr0, nr1, cr0, cr1, kb + 96, sA, sB, pA0, pA1, pB0, pB1);
        else period(T1, T1, T1, T0, ic<0>{}, ic<2>{}, ic<1>{}, ic<3>{}, nr0, nr1, ct0, ct1, nr0, nr1, cr0, cr1, kb + 96, sA, sB, pA0, pA1, pB0, pB1);
      } else {
        period(T1, T1, T1, T0, ic<0>{}, ic<2>{}, ic<1>{}, ic<3>{}, nr0, nr1, ct0, ct1, nr0, nr1, cr0, cr1, kb + 96, sA, sB, pA0, pA1, pB0, pB1);
      }
    } else {
      // two sub-tiles per stage: stage st+1 (requested one stage ago) is published here, then stage st+2 is requested
      if (st > 0) {   // (stage 1 was waited for and published in the prologue)
        dma_wait_all();
        __syncthreads();
      }
      if (st + 2 < nstage) stage_dma((st + 2) * ST, slot_of(st + 2));
      // period 2st+0: produce sub 1, softmax of sub 0, P.V of sub 1 of the previous stage; rows two ahead: next stage, sub 0
      if constexpr (MASKS) {
        if (need(0)) period(T1, T1, T1, T1, ic<1>{}, ic<1>{}, ic<0>{}, ic<0>{}, cr0, cr1, pt0, pt1, nr0, nr1, cr0, cr1, kb, sB, sA, pB0, pB1, pA0, pA1);
        else period(T1, T1, T1, T0, ic<1>{}, ic<1>{}, ic<0>{}, ic<0>{}, cr0, cr1, pt0, pt1, nr0, nr1, cr0, cr1, kb, sB, sA, pB0, pB1, pA0, pA1);
      } else {
        period(T1, T1, T1, T0, ic<1>{}, ic<1>{}, ic<0>{}, ic<0>{}, cr0, cr1, pt0, pt1, nr0, nr1, cr0, cr1, kb, sB, sA, pB0, pB1, pA0, pA1);
      }
      // period 2st+1: produce sub 0 of the next stage, softmax of sub 1, P.V of sub 0; rows two ahead: next stage, sub 1
      if constexpr (MASKS) {
        if (need(1)) period(T1, T1, T1, T1, ic<0>{}, ic<0>{}, ic<1>{}, ic<1>{}, nr0, nr1, ct0, ct1, nr0, nr1, cr0, cr1, kb + 32, sA, sB, pA0, pA1, pB0, pB1);
        else period(T1, T1, T1, T0, ic<0>{}, ic<0>{}, ic<1>{}, ic<1>{}, nr0, nr1, ct0, ct1, nr0, nr1, cr0, cr1, kb + 32, sA, sB, pA0, pA1, pB0, pB1);
      } else {
        period(T1, T1, T1, T0, ic<0>{}, ic<0>{}, ic<1>{}, ic<1>{}, nr0, nr1, ct0, ct1, nr0, nr1, cr0, cr1, kb + 32, sA, sB, pA0, pA1, pB0, pB1);
      }
    }
    pt0 = ct0; pt1 = ct1;
    cr0 = nr0; cr1 = nr1;
    ct0 = ta.b[0] + nb; ct1 = ta.b[1] + nb;
  }
  // drain: P.V of the last sub-tile (the buffers alternate per sub-tile: an even count per stage ends on B)
  period(T0, T0, T1, T0, ic<0>{}, ic<NSUBT - 1>{}, ic<0>{}, ic<0>{}, cr0, cr1, pt0, pt1, cr0, cr1, cr0, cr1, 0, sB, sA, pB0, pB1, pA0, pA1);

  if constexpr (DIAG) { t1 = stamp(); ph[1] += t1 - t0; t0 = t1; }
  const float l_tot = xhalf_sum(l_run);
  const float inv = 1.0f / l_tot;
  if (qvalid) {
    float* orow = o + base + (size_t)qrow * ld;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 val = {acc_o[dt][4 * g] * inv, acc_o[dt][4 * g + 1] * inv, acc_o[dt][4 * g + 2] * inv,
                     acc_o[dt][4 * g + 3] * inv};
        *reinterpret_cast<f32x4*>(orow + 32 * dt + 8 * g + 4 * h) = val;
      }
    if (h == 0) aux_l[(size_t)bh * N + qrow] = m_ref * tau + __logf(l_tot);
  }
  if constexpr (DIAG) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long k_t1 = stamp(), k_r1 = __builtin_amdgcn_s_memrealtime();
    ph[4] += k_t1 - t0;   // epilogue: O / L stores
    const int slot = blockIdx.x * 8 + w;
    if (slot < 8192 && lane == 0) {
      for (int j = 0; j < 6; ++j) g_phase_cycles[slot * 8 + j] = ph[j];
      g_phase_cycles[slot * 8 + 6] = k_t1 - k_t0;
      g_phase_cycles[slot * 8 + 7] = k_r1 - k_r0;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward preprocess: ndelta = -rowsum(dO * O), nlc = -L / tau (raw score units) with L = m + log(l) (FA-1 side
// outputs) or L = l (FA-2), so that P = exp2(tau*log2e * ((q.k) + nlc)) and dS = P * (dO.V^T + ndelta): both row
// constants enter the main kernels as MFMA accumulator inputs (S' = Q.K^T + nlc, dP' = dO.V^T + ndelta).  The reference recomputes D_i per (i, j) tile
// (src/flash_attn_bw.cu:194-197); once per row gives the same value.
// ---------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ void __launch_bounds__(256)
bwd_prep_kernel(const float* __restrict__ o, const T* __restrict__ dout, const float* __restrict__ l,
                const float* __restrict__ m, float* __restrict__ nlc, float* __restrict__ ndelta, long rows, int N,
                Layout lay, int aux_mode, float inv_tau) {
  constexpr int LPR = D / 8;  // lanes per row, 8 elements each
  constexpr int RPB = 256 / LPR;
  const int tid = threadIdx.x;
  const long row = (long)blockIdx.x * RPB + tid / LPR;
  const int part = tid % LPR;
  float sum = 0.f;
  if (row < rows) {
    const size_t off = head_base(lay, (int)(row / N)) + (size_t)(row % N) * lay.ld + part * 8;
    const float* op = o + off;
    const T* dp = dout + off;
    f32x4 o0 = *reinterpret_cast<const f32x4*>(op), o1 = *reinterpret_cast<const f32x4*>(op + 4);
    typename Atom<T>::frag df = Atom<T>::load_global(dp);
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += o0[j] * (float)df[j] + o1[j] * (float)df[4 + j];
  }
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  if (row < rows && part == 0) {
    ndelta[row] = -sum;
    const float L = (aux_mode == AUX_FA1) ? (m[row] + __logf(l[row])) : l[row];
    nlc[row] = (L == -INFINITY) ? -INFINITY : -L * inv_tau;   // fully masked row: P = exp2(c * (S - inf)) = 0
  }
}

// ---------------------------------------------------------------------------------------------
// Backward dK / dV: a workgroup = NW waves = NW*KPW keys of one (batch*head); each wave keeps K, V fragments and
// the dK^T, dV^T accumulators of its KPW keys in registers while the workgroup sweeps 32-row query slices
// (Q, dO tiles + their nlc, delta staged in LDS, double buffered).
// ---------------------------------------------------------------------------------------------
template <typename T, int D, int KPW, int NW, int QS, int MODE = 0, bool HD = false, int MINW = 1>
__global__ void __launch_bounds__(NW * 64, MINW)   // MINW: minimum waves per SIMD the register allocation must allow
bwd_dkdv_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ dout,
                const float* __restrict__ nlc, const float* __restrict__ ndelta, float* __restrict__ dk,
                float* __restrict__ dv, int N, int nkb, int BH, Layout lay, int causal, float tau) {
  using A = Atom<T>;
  typedef typename A::frag frag;
  constexpr int KC = D / 16, KT = KPW / 32, DT = D / 32, BK = NW * KPW, NT = NW * 64, NSUB = QS / 32;
  constexpr int TB = A::template tile_bytes<D>(QS);
  constexpr int BUF = 2 * TB + 8 * QS;  // Q tile, dO tile, QS x nlc, QS x -delta
  __shared__ __attribute__((aligned(16))) char smem_raw[2 * BUF];
  lds_char* smem = (lds_char*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bh, kb;
  map_block(blockIdx.x, BH, nkb, bh, kb);
  const int kb0 = kb * BK, kw0 = kb0 + w * KPW;
  const size_t base = head_base(lay, bh);
  const int ld = lay.ld;   // elements between consecutive rows
  const uint32_t mat_bytes = ((uint32_t)(N - 1) * ld + D) * (uint32_t)sizeof(T);
  const rsrc_t qrs = make_rsrc(q + base, mat_bytes);
  const rsrc_t dors = make_rsrc(dout + base, mat_bytes);
  const rsrc_t krs = make_rsrc(k + base, mat_bytes);
  const rsrc_t vrs = make_rsrc(v + base, mat_bytes);
  const float* nlg = nlc + (size_t)bh * N;
  const float* deg = ndelta + (size_t)bh * N;
  const float c = tau * LOG2E;

  frag kf[KT][KC], vf[KT][KC];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int off = ((kw0 + 32 * kt + r) * ld + 16 * kc + 8 * h) * (int)sizeof(T);  // rows >= N read as zero
      kf[kt][kc] = load_frag_buf<T>(krs, off);
      vf[kt][kc] = load_frag_buf<T>(vrs, off);
    }
  // optional additive key mask: the key is on the lane, so it is one addend per lane and key tile, in log2 units
  // (P = exp2(c * S' + mask * log2e)); zero without a mask, where the fma costs what the multiply did
  float km[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int key = kw0 + 32 * kt + r;
    km[kt] = (lay.kmask != nullptr && key < N) ? lay.kmask[(size_t)(bh / lay.mask_heads) * N + key] * LOG2E : 0.f;
  }
  f32x16 acc_dk[DT][KT], acc_dv[DT][KT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      acc_dk[dt][kt] = zero16();
      acc_dv[dt][kt] = zero16();
    }

  const LaneAddr ra = A::template row_addr<D>(lane);
  const LaneAddr ta = A::template tr_addr<D>(lane);
  const int nqi = (N + QS - 1) / QS;
  const int qi_begin = causal ? (kb0 / QS) : 0;  // query slices entirely above the key block are fully masked
  // Stage copies of Q and dO.  bf16, d >= 64: LDS-DMA, 1 KiB pieces (half an 8-row group at d = 128), the image's chunk swizzle
  // applied to each lane's source address; wave w moves pieces w, w + NW, ... (same swizzle parity, one lane offset) -- no
  // staging registers, no ds_write pass.  Otherwise (fp32's padded image, d = 32): registers, written after the MFMA phase.
  constexpr bool DMA = sizeof(T) == 2 && D >= 64 && MODE != 9 && MODE != 13;   // MODE 13: slot path on register staging (A/B)
  constexpr int PPG = D >= 128 ? 2 : 1;                          // pieces per 8-row group
  constexpr int NP = QS * D * (int)sizeof(T) / 1024, NPW = DMA ? NP / NW : 0;
  static_assert(!DMA || (NP % NW == 0 && NW % 4 == 0), "every wave moves whole pieces of one swizzle parity");
  TileStager<T, D, QS, NT> sq, sdo;
  if constexpr (!DMA) {
    sq.init(tid, ld);
    sdo.init(tid, ld);
  }
  const raw_rsrc_t qraw = make_raw_rsrc(q + base, mat_bytes), doraw = make_raw_rsrc(dout + base, mat_bytes);
  const uint32_t smem_addr = (uint32_t)(uintptr_t)smem;
  const int dma_row7 = (lane >> 2) & 7;
  const int dma_gpar = (PPG == 1) ? (w & 1) : ((w >> 1) & 1);
  const int dma_half = (PPG == 1) ? 0 : (w & 1);
  const int dma_voff = dma_row7 * ld * (int)sizeof(T) +
                       16 * (4 * (2 * dma_half + (lane >> 5)) + ((lane & 3) ^ ((2 * dma_gpar + (dma_row7 >> 2)) & 3)));
  float st_nl = 0.f, st_de = 0.f;
  auto stage_load = [&](int qi, int dst /* LDS byte offset of the stage buffer */) {
    if constexpr (DMA) {
#pragma unroll
      for (int i = 0; i < NPW; ++i) {
        const int piece = w + NW * i, g = piece / PPG;
        const int soff = (qi * QS + 8 * g) * ld * (int)sizeof(T);
        dma16(qraw, smem_addr + dst + 1024 * piece, dma_voff, soff);
        dma16(doraw, smem_addr + dst + TB + 1024 * piece, dma_voff, soff);
      }
    } else {
      sq.load(qrs, qi * QS);
      sdo.load(dors, qi * QS);
    }
    if (tid < QS) {
      const int row = qi * QS + tid;
      st_nl = row < N ? nlg[row] : 0.f;
      st_de = row < N ? deg[row] : 0.f;
    }
  };
  auto stage_store = [&](lds_char* b) {
    if constexpr (DMA) {
      dma_wait_all();   // this wave's pieces have landed (the barrier that follows publishes them)
    } else {
      sq.store(b);
      sdo.store(b + TB);
    }
    if (tid < QS) {
      *FA_LDS(float, b + 2 * TB + 4 * tid) = st_nl;
      *FA_LDS(float, b + 2 * TB + 4 * QS + 4 * tid) = st_de;
    }
  };
  if (qi_begin < nqi) {
    stage_load(qi_begin, 0);
    stage_store(smem);
  }
  __syncthreads();

  constexpr bool DIAG = MODE == 9 || MODE == 93;
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long k_t0 = 0, k_r0 = 0;
  if constexpr (DIAG) {
    k_t0 = stamp();
    k_r0 = __builtin_amdgcn_s_memrealtime();
  }
  auto slice = [&](auto par, int qi) {
    constexpr int PAR = decltype(par)::value;
    const bool more = qi + 1 < nqi;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if constexpr (DIAG) t0 = stamp();
    if (more) stage_load(qi + 1, (PAR ^ 1) * BUF);
    if constexpr (DIAG) { t1 = stamp(); ph[0] += t1 - t0; }
    lds_char* buf = smem + PAR * BUF;
    lds_char* tq = buf;
    lds_char* tdo = buf + TB;
    // ---- slot-interleaved fast path (MODE 3; stage fully unmasked).  One wave's instruction stream is laid out as
    // MFMA "slots": each slot is one MFMA plus at most ~24 issue cycles of VALU (v_exp 8, others 4) plus the LDS reads
    // of later slots, pinned with sched_barrier(0).  On gfx950 an MFMA holds the SIMD's vector issue port for 8 of its
    // 32 cycles and a back-to-back MFMA waiting for the pipe blocks the port for every wave, so softmax VALU only hides
    // when it sits between a wave's OWN MFMAs (MI355X_MICROARCH.md, per-instruction constants).  A period is 16 slots:
    //   slots 0-7   S', dP' of sub-slice i+1 (row constants enter as accumulator inputs)   | exp of sub-slice i
    //   slots 8-15  dV^T += dO^T P, dK^T += Q^T dS of sub-slice i                           | mul / pack of sub-slice i
    // LDS fragments are requested four slots before the MFMA that consumes them.
    constexpr bool SLOT = !HD && (MODE == 3 || MODE == 93 || MODE == 13) && NSUB == 4 && D == 64 && KT == 1 && sizeof(T) == 2;
    if constexpr (SLOT) {
      const bool fast3 = (kw0 < N) && (!causal || qi * QS >= kw0 + KPW - 1);   // wave-uniform
      if (fast3) {
        f32x16 sA, dpA, sB, dpB, cS, cD;
        frag pf0, pf1, df0, df1, rq[4], rdo[4], tf[4];
        auto SB = [&]() { __builtin_amdgcn_sched_barrier(0); };
        auto ld_c = [&](f32x16& x, int off, int sub) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 a = *FA_LDS(f32x4, buf + 2 * TB + off + 128 * sub + 16 * h + 32 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) x[4 * g + j] = a[j];
          }
        };
        auto me = [&](f32x16& x, int i) { x[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], c, km[0])); };
        auto period = [&](auto subn_c, auto subc_c, f32x16& ns, f32x16& ndp, f32x16& cs, f32x16& cdp) {
          constexpr int SN = decltype(subn_c)::value, SC = decltype(subc_c)::value;
          constexpr bool HN = SN >= 0, HC = SC >= 0, HP = HN && SN + 1 < NSUB;
          constexpr int SNc = HN ? SN : 0, SCc = HC ? SC : 0;
          // slots 0-3: S' chain of the next sub-slice | exp of scores 0..7 | dO rows 1..3
#pragma unroll
          for (int kq = 0; kq < 4; ++kq) {
            if constexpr (HN) {
              if (kq == 0) A::mma_c(ns, rq[0], kf[0][0], cS);
              else A::mma(ns, rq[kq], kf[0][kq]);
              SB();   // the MFMA opens its slot; the fillers follow in its shadow
              if (kq < 3) rdo[kq + 1] = A::template row_frag<D>(tdo, ra, 32 * SNc, kq + 1);
            }
            if constexpr (HC) { me(cs, 2 * kq); me(cs, 2 * kq + 1); }
            SB();
          }
          // slot 4
          if constexpr (HN) { A::mma_c(ndp, rdo[0], vf[0][0], cD); SB(); }
          if constexpr (HC) {
            pf0 = A::pack(cs, 0);
            cdp[0] = cs[0] * cdp[0];
            tf[0] = A::template tr_frag<D>(tdo, ta, 32 * SCc, 0);
          }
          SB();
          // slots 5-7
#pragma unroll
          for (int kq = 1; kq < 4; ++kq) {
            if constexpr (HN) { A::mma(ndp, rdo[kq], vf[0][kq]); SB(); }
            if constexpr (HC) {
              me(cs, 6 + 2 * kq); me(cs, 7 + 2 * kq);
              tf[kq] = A::template tr_frag<D>(tdo, ta, 32 * SCc + 16 * (kq >> 1), kq & 1);
            }
            SB();
          }
          if constexpr (HC) {
            // slot 8
            A::mma(acc_dv[0][0], tf[0], pf0);
            SB();
            me(cs, 14); me(cs, 15);
            tf[0] = A::template tr_frag<D>(tq, ta, 32 * SCc, 0);
            SB();
            // slot 9
            A::mma(acc_dv[1][0], tf[1], pf0);
            SB();
            pf1 = A::pack(cs, 1);
            cdp[1] = cs[1] * cdp[1];
            tf[1] = A::template tr_frag<D>(tq, ta, 32 * SCc, 1);
            SB();
            // slot 10
            A::mma(acc_dv[0][0], tf[2], pf1);
            SB();
#pragma unroll
            for (int i = 2; i < 8; ++i) cdp[i] = cs[i] * cdp[i];
            tf[2] = A::template tr_frag<D>(tq, ta, 32 * SCc + 16, 0);
            SB();
            // slot 11
            A::mma(acc_dv[1][0], tf[3], pf1);
            SB();
            df0 = A::pack(cdp, 0);
            cdp[8] = cs[8] * cdp[8];
            tf[3] = A::template tr_frag<D>(tq, ta, 32 * SCc + 16, 1);
            SB();
            // slot 12
            A::mma(acc_dk[0][0], tf[0], df0);
            SB();
#pragma unroll
            for (int i = 9; i < 15; ++i) cdp[i] = cs[i] * cdp[i];
          }
          if constexpr (HP) {
            rq[0] = A::template row_frag<D>(tq, ra, 32 * (SNc + 1), 0);
            rq[1] = A::template row_frag<D>(tq, ra, 32 * (SNc + 1), 1);
          }
          SB();
          // slot 13
          if constexpr (HC) {
            A::mma(acc_dk[1][0], tf[1], df0);
            SB();
            cdp[15] = cs[15] * cdp[15];
            df1 = A::pack(cdp, 1);
          }
          if constexpr (HP) {
            rq[2] = A::template row_frag<D>(tq, ra, 32 * (SNc + 1), 2);
            rq[3] = A::template row_frag<D>(tq, ra, 32 * (SNc + 1), 3);
          }
          SB();
          // slot 14
          if constexpr (HC) { A::mma(acc_dk[0][0], tf[2], df1); SB(); }
          if constexpr (HP) ld_c(cS, 0, SNc + 1);
          SB();
          // slot 15
          if constexpr (HC) { A::mma(acc_dk[1][0], tf[3], df1); SB(); }
          if constexpr (HP) {
            ld_c(cD, 4 * QS, SNc + 1);
            rdo[0] = A::template row_frag<D>(tdo, ra, 32 * (SNc + 1), 0);
          }
          SB();
        };
        // operands of sub-slice 0
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) rq[kc] = A::template row_frag<D>(tq, ra, 0, kc);
        ld_c(cS, 0, 0);
        ld_c(cD, 4 * QS, 0);
        rdo[0] = A::template row_frag<D>(tdo, ra, 0, 0);
        SB();
        if constexpr (DIAG) t1 = stamp();
        period(ic<0>{}, ic<-1>{}, sA, dpA, sB, dpB);
        if constexpr (DIAG) { t2 = stamp(); ph[1] += t2 - t1; }
        period(ic<1>{}, ic<0>{}, sB, dpB, sA, dpA);
        period(ic<2>{}, ic<1>{}, sA, dpA, sB, dpB);
        period(ic<3>{}, ic<2>{}, sB, dpB, sA, dpA);
        if constexpr (DIAG) { t3 = stamp(); ph[2] += t3 - t2; }
        period(ic<-1>{}, ic<3>{}, sA, dpA, sB, dpB);
        if constexpr (DIAG) { t0 = stamp(); ph[3] += t0 - t3; }
      }
    }
    // ---- software-pipelined fast path (stage fully unmasked): S, dP of sub-slice i+1 are issued before the
    // exp / mul / pack work of sub-slice i, so one wave has independent MFMA and VALU streams to interleave.
    constexpr bool PIPE = !HD && MODE == 0 && NSUB == 4 && D <= 64;   // (needs ~250 VGPRs at d = 64; not for d = 128)
    const bool fast = PIPE && (kw0 < N) && (!causal || qi * QS >= kw0 + KPW - 1);   // wave-uniform
    const bool fast_slot = SLOT && (kw0 < N) && (!causal || qi * QS >= kw0 + KPW - 1);
    if (fast_slot) {
    } else if (fast) {
      auto mfma1 = [&](auto subc, f32x16(&s)[KT], f32x16(&dp)[KT]) {
        constexpr int sub = decltype(subc)::value;
        f32x16 nl16, nd16;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 a = *FA_LDS(f32x4, buf + 2 * TB + 128 * sub + 16 * h + 32 * g);
          const f32x4 b = *FA_LDS(f32x4, buf + 2 * TB + 4 * QS + 128 * sub + 16 * h + 32 * g);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            nl16[4 * g + j] = a[j];
            nd16[4 * g + j] = b[j];
          }
        }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const frag aq = A::template row_frag<D>(tq, ra, 32 * sub, kc);
          const frag ado = A::template row_frag<D>(tdo, ra, 32 * sub, kc);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            if (kc == 0) {
              A::mma_c(s[kt], aq, kf[kt][kc], nl16);
              A::mma_c(dp[kt], ado, vf[kt][kc], nd16);
            } else {
              A::mma(s[kt], aq, kf[kt][kc]);
              A::mma(dp[kt], ado, vf[kt][kc]);
            }
          }
        }
      };
      auto valu = [&](f32x16(&s)[KT], f32x16(&dp)[KT], frag(&pf)[KT][2], frag(&dsf)[KT][2]) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            s[kt][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][i], c, km[kt]));
            dp[kt][i] = s[kt][i] * dp[kt][i];
          }
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            pf[kt][s2] = A::pack(s[kt], s2);
            dsf[kt][s2] = A::pack(dp[kt], s2);
          }
        }
      };
      auto mfma2 = [&](auto subc, const frag(&pf)[KT][2], const frag(&dsf)[KT][2]) {
        constexpr int sub = decltype(subc)::value;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const frag adoT = A::template tr_frag<D>(tdo, ta, 32 * sub + 16 * s2, dt);
            const frag aqT = A::template tr_frag<D>(tq, ta, 32 * sub + 16 * s2, dt);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
              A::mma(acc_dv[dt][kt], adoT, pf[kt][s2]);
              A::mma(acc_dk[dt][kt], aqT, dsf[kt][s2]);
            }
          }
      };
      f32x16 sA[KT], dpA[KT], sB[KT], dpB[KT];
      frag pf[KT][2], dsf[KT][2];
      mfma1(ic<0>{}, sA, dpA);
      mfma1(ic<1>{}, sB, dpB);
      valu(sA, dpA, pf, dsf);
      mfma2(ic<0>{}, pf, dsf);
      mfma1(ic<2>{}, sA, dpA);
      valu(sB, dpB, pf, dsf);
      mfma2(ic<1>{}, pf, dsf);
      mfma1(ic<3>{}, sB, dpB);
      valu(sA, dpA, pf, dsf);
      mfma2(ic<2>{}, pf, dsf);
      valu(sB, dpB, pf, dsf);
      mfma2(ic<3>{}, pf, dsf);
    } else
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      const int qi0 = qi * QS + 32 * sub;
      const bool active = (kw0 < N) && (qi0 < N) && (!causal || qi0 + 31 >= kw0);  // wave-uniform
      if (active) {
        // register i of lane half h is query qi0 + acc_row(i, h): its nlc / -delta come from LDS (broadcast reads);
        // -delta enters the dP tile as the accumulator input of its first MFMA
        if constexpr (DIAG) t1 = stamp();
        f32x16 nl16, nd16;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 a = *FA_LDS(f32x4, buf + 2 * TB + 128 * sub + 16 * h + 32 * g);
          const f32x4 b = *FA_LDS(f32x4, buf + 2 * TB + 4 * QS + 128 * sub + 16 * h + 32 * g);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            nl16[4 * g + j] = a[j];
            nd16[4 * g + j] = b[j];
          }
        }
        f32x16 s[KT], dp[KT];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const frag aq = A::template row_frag<D>(tq, ra, 32 * sub, kc);
          const frag ado = A::template row_frag<D>(tdo, ra, 32 * sub, kc);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            if (kc == 0) {   // row constants ride in as accumulator inputs: S' = S - L/tau, dP' = dP - delta
              A::mma_c(s[kt], aq, kf[kt][kc], nl16);
              if constexpr (HD) A::mma_c(dp[kt], ado, vf[kt][kc], zero16());   // dropout scales dP before -delta is added
              else A::mma_c(dp[kt], ado, vf[kt][kc], nd16);
            } else {
              A::mma(s[kt], aq, kf[kt][kc]);
              A::mma(dp[kt], ado, vf[kt][kc]);
            }
          }
        }
        const bool need_mask = causal && (kw0 + KPW - 1 > qi0);  // wave-uniform
        if constexpr (DIAG) { t2 = stamp(); ph[1] += t2 - t1; }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i) s[kt][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][i], c, km[kt]));
        if (need_mask) {   // diagonal slices only (scalar branch)
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i)
              if (kw0 + 32 * kt + r > qi0 + acc_row(i, h)) s[kt][i] = 0.f;
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if constexpr (HD) {   // dS = P * (scale * M * dP - delta); the dV product takes scale * M * P
              const bool keep = drop_keep(drop_base(lay, bh, qi0 + acc_row(i, h)), kw0 + 32 * kt + r, lay.drop_thr);
              dp[kt][i] = s[kt][i] * ((keep ? dp[kt][i] * lay.drop_scale : 0.f) + nd16[i]);
              s[kt][i] = keep ? s[kt][i] * lay.drop_scale : 0.f;
            } else {
              dp[kt][i] = s[kt][i] * dp[kt][i];
            }
          }
        frag pf[KT][2], dsf[KT][2];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            pf[kt][s2] = A::pack(s[kt], s2);
            dsf[kt][s2] = A::pack(dp[kt], s2);
          }
        if constexpr (DIAG) {
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
              for (int j = 0; j < 4; ++j) {   // pin the VALU phase in front of the stamp
                asm volatile("" ::"v"(__builtin_bit_cast(u32x4, pf[kt][s2])[j]));
                asm volatile("" ::"v"(__builtin_bit_cast(u32x4, dsf[kt][s2])[j]));
              }
          t3 = stamp();
          ph[2] += t3 - t2;
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const frag adoT = A::template tr_frag<D>(tdo, ta, 32 * sub + 16 * s2, dt);
            const frag aqT = A::template tr_frag<D>(tq, ta, 32 * sub + 16 * s2, dt);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
              A::mma(acc_dv[dt][kt], adoT, pf[kt][s2]);
              A::mma(acc_dk[dt][kt], aqT, dsf[kt][s2]);
            }
          }
        if constexpr (DIAG) { t0 = stamp(); ph[3] += t0 - t3; }
      }
    }
    if constexpr (DIAG) t0 = stamp();
    if (more) stage_store(smem + (PAR ^ 1) * BUF);
    if constexpr (DIAG) { t1 = stamp(); ph[4] += t1 - t0; }
    __syncthreads();
    if constexpr (DIAG) { t2 = stamp(); ph[5] += t2 - t1; }
  };
  int qi = qi_begin;
  for (; qi + 1 < nqi; qi += 2) {
    slice(ic<0>{}, qi);
    slice(ic<1>{}, qi + 1);
  }
  if (qi < nqi) slice(ic<0>{}, qi);

  if constexpr (DIAG) {
    const int slot = blockIdx.x * NW + w;
    const unsigned long long k_t1 = stamp(), k_r1 = __builtin_amdgcn_s_memrealtime();
    if (slot < 8192 && lane == 0) {
      for (int j = 0; j < 6; ++j) g_phase_cycles[slot * 8 + j] = ph[j];
      g_phase_cycles[slot * 8 + 6] = k_t1 - k_t0;   // wave lifetime in shader cycles
      g_phase_cycles[slot * 8 + 7] = k_r1 - k_r0;   // the same in 100 MHz ticks
    }
  }
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int key = kw0 + 32 * kt + r;
    if (key < N) {
      float* dkrow = dk + base + (size_t)key * ld;
      float* dvrow = dv + base + (size_t)key * ld;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 a = {acc_dk[dt][kt][4 * g] * tau, acc_dk[dt][kt][4 * g + 1] * tau, acc_dk[dt][kt][4 * g + 2] * tau,
                     acc_dk[dt][kt][4 * g + 3] * tau};
          f32x4 b = {acc_dv[dt][kt][4 * g], acc_dv[dt][kt][4 * g + 1], acc_dv[dt][kt][4 * g + 2],
                     acc_dv[dt][kt][4 * g + 3]};
          *reinterpret_cast<f32x4*>(dkrow + 32 * dt + 8 * g + 4 * h) = a;
          *reinterpret_cast<f32x4*>(dvrow + 32 * dt + 8 * g + 4 * h) = b;
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward dK / dV, continuous slot pipeline (bf16, d = 64, NON-CAUSAL launches): the geometry and the 16-slot period of
// bwd_dkdv_kernel's MODE 3 (8 waves x 32 keys, 128-query stages of four 32-query sub-slices), but the pipeline never drains
// at a stage boundary: period c of a stage issues S', dP' of sub-slice c+1 (sub-slice 0 of the NEXT stage when c = 3) beside
// the exp / mul / pack and the dV^T, dK^T products of sub-slice c.  Stages (Q, dO tiles and the two row-constant vectors)
// arrive by LDS-DMA into a three-slot ring; the barrier that publishes stage s+1 sits between periods 1 and 2 of stage s
// (sub-slice 0 of stage s+1 is first requested in period 2), and its DMA is issued at the top of stage s into the slot of
// stage s-2, which every wave left before that barrier of stage s-1.  No compiler-tracked global load in the loop.
// ---------------------------------------------------------------------------------------------
template <typename T, int D, int DIAG = 0>
__global__ void __launch_bounds__(512)
bwd_dkdv_slot_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ dout,
                     const float* __restrict__ nlc, const float* __restrict__ ndelta, float* __restrict__ dk,
                     float* __restrict__ dv, int N, int nkb, int BH, Layout lay, float tau) {
  static_assert(D == 64 && sizeof(T) == 2, "slot schedule is laid out for bf16, d = 64");
  using A = Atom<T>;
  typedef typename A::frag frag;
  constexpr int KC = 4, QS = 128, NW = 8, KPW = 32, BK = NW * KPW;
  constexpr int TB = A::template tile_bytes<D>(QS);   // 16 KiB
  constexpr int BUF = 2 * TB + 8 * QS;                // Q tile, dO tile, QS x (-L/tau), QS x (-delta)
  constexpr int SUBB = (D / 32) * 512 * 4;            // bytes of one 32-row sub-slice inside a tile image
  __shared__ __attribute__((aligned(16))) char smem_raw[3 * BUF];
  lds_char* smem = (lds_char*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bh, kb;
  map_block(blockIdx.x, BH, nkb, bh, kb);
  const int kw0 = kb * BK + w * KPW;
  const bool active = kw0 < N;   // wave-uniform: a wave whose keys all lie past N only moves data and joins the barriers
  const size_t base = head_base(lay, bh);
  const int ld = lay.ld;
  const uint32_t mat_bytes = ((uint32_t)(N - 1) * ld + D) * (uint32_t)sizeof(T);
  const rsrc_t krs = make_rsrc(k + base, mat_bytes);
  const rsrc_t vrs = make_rsrc(v + base, mat_bytes);
  const raw_rsrc_t qraw = make_raw_rsrc(q + base, mat_bytes), doraw = make_raw_rsrc(dout + base, mat_bytes);
  const raw_rsrc_t nlraw = make_raw_rsrc(nlc + (size_t)bh * N, (uint32_t)N * 4u);
  const raw_rsrc_t ndraw = make_raw_rsrc(ndelta + (size_t)bh * N, (uint32_t)N * 4u);
  const float c = tau * LOG2E;

  frag kf[KC], vf[KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const int off = ((kw0 + r) * ld + 16 * kc + 8 * h) * (int)sizeof(T);   // rows >= N read as zero
    kf[kc] = load_frag_buf<T>(krs, off);
    vf[kc] = load_frag_buf<T>(vrs, off);
  }
  const int key = kw0 + r;
  const float km = (lay.kmask != nullptr && key < N) ? lay.kmask[(size_t)(bh / lay.mask_heads) * N + key] * LOG2E : 0.f;
  f32x16 acc_dk[2], acc_dv[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    acc_dk[dt] = zero16();
    acc_dv[dt] = zero16();
  }

  const LaneAddr ra = A::template row_addr<D>(lane);
  const LaneAddr ta = A::template tr_addr<D>(lane);
  const int nst = (N + QS - 1) / QS;
  const uint32_t smem_addr = (uint32_t)(uintptr_t)smem;
  // LDS-DMA: wave w moves pieces w and w + 8 (1 KiB = one 8-row group) of the Q and of the dO tile, waves 0-3 the row constants
  const int dma_row7 = (lane >> 2) & 7;
  const int dma_voff = dma_row7 * ld * (int)sizeof(T) + 16 * (4 * (lane >> 5) + ((lane & 3) ^ ((2 * (w & 1) + (dma_row7 >> 2)) & 3)));
  auto stage_dma = [&](int st, int dst) {
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
      const int g = w + 8 * g2;
      const int soff = (st * QS + 8 * g) * ld * (int)sizeof(T);
      dma16(qraw, smem_addr + dst + 1024 * g, dma_voff, soff);
      dma16(doraw, smem_addr + dst + TB + 1024 * g, dma_voff, soff);
    }
    if (w < 4) {   // rows past N read as zero: P = exp2(c * S') stays finite and meets dO = 0, Q = 0
      const int half = w & 1;
      dma4((w < 2) ? nlraw : ndraw, smem_addr + dst + 2 * TB + ((w < 2) ? 0 : 4 * QS) + 256 * half, 4 * lane,
           (st * QS + 64 * half) * 4);
    }
  };
  auto slot_of = [&](int st) { return (st % 3) * BUF; };
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, k_t0 = 0, k_r0 = 0, t0 = 0, t1 = 0;
  if constexpr (DIAG) {
    k_t0 = stamp();
    k_r0 = __builtin_amdgcn_s_memrealtime();
  }
  stage_dma(0, 0);
  dma_wait_all();
  __syncthreads();
  if constexpr (DIAG) { t0 = stamp(); ph[0] += t0 - k_t0; }

  f32x16 sA, dpA, sB, dpB, cS, cD;
  frag pf0, pf1, df0, df1, rq[4], rdo[4], tf[4];
  auto SB = [&]() { __builtin_amdgcn_sched_barrier(0); };
  // LDS readers: per-stage address registers (row / transposed, two swizzle phases each) + immediates
  auto rowf = [&](int b0, int b1, int tile_off, int sub, int kc) -> frag {
    return *FA_LDS(frag, smem + ((kc & 1) ? b1 : b0) + tile_off + SUBB * sub + 512 * (kc >> 1));
  };
  auto trf = [&](int b0, int b1, int tile_off, int sub, int s2, int dt) -> frag {
    const int kk = tile_off + SUBB * sub + (D / 32) * 512 * (2 * s2) + 512 * dt;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(FA_LDS(bf16x4, smem + b0 + kk));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(FA_LDS(bf16x4, smem + b1 + kk + (D / 32) * 512));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto ld_c = [&](f32x16& x, int hb /* stage base + 16 * h */, int off, int sub) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 a = *FA_LDS(f32x4, smem + hb + 2 * TB + off + 128 * sub + 32 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j) x[4 * g + j] = a[j];
    }
  };
  auto me = [&](f32x16& x, int i) { x[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], c, km)); };
  // One period.  SN: sub-slice whose S', dP' are produced, rows at (nr0, nr1) [its dO rows 1..3 are requested here]; SC:
  // sub-slice in the softmax / dV, dK stream, transposed reads at (ct0, ct1); SP: the sub-slice after SN, whose Q rows, row
  // constants and first dO row are requested in slots 12-15 at (pr0, pr1, ph16).
  auto period = [&](auto hn_c, auto hc_c, auto subn_c, auto subc_c, auto subp_c, int nr0, int nr1, int ct0, int ct1, int pr0,
                    int pr1, int ph16, f32x16& ns, f32x16& ndp, f32x16& cs, f32x16& cdp) {
    constexpr bool HN = decltype(hn_c)::value != 0, HC = decltype(hc_c)::value != 0;
    constexpr int SN = decltype(subn_c)::value, SC = decltype(subc_c)::value, SP = decltype(subp_c)::value;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {   // slots 0-3: S' chain | exp of scores 0..7 | dO rows 1..3
      if constexpr (HN) {
        if (kq == 0) A::mma_c(ns, rq[0], kf[0], cS);
        else A::mma(ns, rq[kq], kf[kq]);
        SB();
        if (kq < 3) rdo[kq + 1] = rowf(nr0, nr1, TB, SN, kq + 1);
      }
      if constexpr (HC) { me(cs, 2 * kq); me(cs, 2 * kq + 1); }
      SB();
    }
    // slot 4
    if constexpr (HN) { A::mma_c(ndp, rdo[0], vf[0], cD); SB(); }
    if constexpr (HC) {
      pf0 = A::pack(cs, 0);
      cdp[0] = cs[0] * cdp[0];
      tf[0] = trf(ct0, ct1, TB, SC, 0, 0);
    }
    SB();
#pragma unroll
    for (int kq = 1; kq < 4; ++kq) {   // slots 5-7
      if constexpr (HN) { A::mma(ndp, rdo[kq], vf[kq]); SB(); }
      if constexpr (HC) {
        me(cs, 6 + 2 * kq); me(cs, 7 + 2 * kq);
        tf[kq] = trf(ct0, ct1, TB, SC, kq >> 1, kq & 1);
      }
      SB();
    }
    if constexpr (HC) {
      A::mma(acc_dv[0], tf[0], pf0);   // slot 8
      SB();
      me(cs, 14); me(cs, 15);
      tf[0] = trf(ct0, ct1, 0, SC, 0, 0);
      SB();
      A::mma(acc_dv[1], tf[1], pf0);   // slot 9
      SB();
      pf1 = A::pack(cs, 1);
      cdp[1] = cs[1] * cdp[1];
      tf[1] = trf(ct0, ct1, 0, SC, 0, 1);
      SB();
      A::mma(acc_dv[0], tf[2], pf1);   // slot 10
      SB();
#pragma unroll
      for (int i = 2; i < 8; ++i) cdp[i] = cs[i] * cdp[i];
      tf[2] = trf(ct0, ct1, 0, SC, 1, 0);
      SB();
      A::mma(acc_dv[1], tf[3], pf1);   // slot 11
      SB();
      df0 = A::pack(cdp, 0);
      cdp[8] = cs[8] * cdp[8];
      tf[3] = trf(ct0, ct1, 0, SC, 1, 1);
      SB();
      A::mma(acc_dk[0], tf[0], df0);   // slot 12
      SB();
#pragma unroll
      for (int i = 9; i < 15; ++i) cdp[i] = cs[i] * cdp[i];
    }
    if constexpr (HN) {
      rq[0] = rowf(pr0, pr1, 0, SP, 0);
      rq[1] = rowf(pr0, pr1, 0, SP, 1);
    }
    SB();
    if constexpr (HC) {   // slot 13
      A::mma(acc_dk[1], tf[1], df0);
      SB();
      cdp[15] = cs[15] * cdp[15];
      df1 = A::pack(cdp, 1);
    }
    if constexpr (HN) {
      rq[2] = rowf(pr0, pr1, 0, SP, 2);
      rq[3] = rowf(pr0, pr1, 0, SP, 3);
    }
    SB();
    if constexpr (HC) { A::mma(acc_dk[0], tf[2], df1); SB(); }   // slot 14
    if constexpr (HN) ld_c(cS, ph16, 0, SP);
    SB();
    if constexpr (HC) { A::mma(acc_dk[1], tf[3], df1); SB(); }   // slot 15
    if constexpr (HN) {
      ld_c(cD, ph16, 4 * QS, SP);
      rdo[0] = rowf(pr0, pr1, TB, SP, 0);
    }
    SB();
  };
  auto T1 = ic<1>{};
  auto T0 = ic<0>{};
  int cr0 = ra.b[0], cr1 = ra.b[1], ct0 = ta.b[0], ct1 = ta.b[1], ch16 = 16 * h;   // addresses of the current stage (slot 0)
  if (active) {
    // operands of sub-slice 0, then its S', dP' alone (the pipeline fills)
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) rq[kc] = rowf(cr0, cr1, 0, 0, kc);
    ld_c(cS, ch16, 0, 0);
    ld_c(cD, ch16, 4 * QS, 0);
    rdo[0] = rowf(cr0, cr1, TB, 0, 0);
    SB();
    period(T1, T0, ic<0>{}, ic<0>{}, ic<1>{}, cr0, cr1, ct0, ct1, cr0, cr1, ch16, sA, dpA, sB, dpB);
  }
  for (int st = 0; st < nst; ++st) {
    const int nb = slot_of(st + 1);
    const int nr0 = ra.b[0] + nb, nr1 = ra.b[1] + nb, nh16 = 16 * h + nb;
    if (st + 1 < nst) stage_dma(st + 1, nb);
    if (active) {
      period(T1, T1, ic<1>{}, ic<0>{}, ic<2>{}, cr0, cr1, ct0, ct1, cr0, cr1, ch16, sB, dpB, sA, dpA);
      period(T1, T1, ic<2>{}, ic<1>{}, ic<3>{}, cr0, cr1, ct0, ct1, cr0, cr1, ch16, sA, dpA, sB, dpB);
    }
    if constexpr (DIAG) { t1 = stamp(); ph[1] += t1 - t0; }
    dma_wait_all();   // this wave's pieces of the next stage have landed
    if constexpr (DIAG) { t0 = stamp(); ph[2] += t0 - t1; }
    __syncthreads();
    if constexpr (DIAG) { t1 = stamp(); ph[3] += t1 - t0; t0 = t1; }
    if (active) {
      // sub-slice 0 of the next stage is requested from here on (after the last stage: stale data, results unused)
      period(T1, T1, ic<3>{}, ic<2>{}, ic<0>{}, cr0, cr1, ct0, ct1, nr0, nr1, nh16, sB, dpB, sA, dpA);
      period(T1, T1, ic<0>{}, ic<3>{}, ic<1>{}, nr0, nr1, ct0, ct1, nr0, nr1, nh16, sA, dpA, sB, dpB);
    }
    cr0 = nr0; cr1 = nr1; ch16 = nh16;
    ct0 = ta.b[0] + nb; ct1 = ta.b[1] + nb;
  }
  if constexpr (DIAG) {
    const unsigned long long k_t1 = stamp(), k_r1 = __builtin_amdgcn_s_memrealtime();
    ph[1] += k_t1 - t0;
    const int slot = blockIdx.x * 8 + w;
    if (slot < 8192 && lane == 0) {
      for (int j = 0; j < 6; ++j) g_phase_cycles[slot * 8 + j] = ph[j];
      g_phase_cycles[slot * 8 + 6] = k_t1 - k_t0;
      g_phase_cycles[slot * 8 + 7] = k_r1 - k_r0;
    }
  }
  if (key < N) {
    float* dkrow = dk + base + (size_t)key * ld;
    float* dvrow = dv + base + (size_t)key * ld;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 a = {acc_dk[dt][4 * g] * tau, acc_dk[dt][4 * g + 1] * tau, acc_dk[dt][4 * g + 2] * tau, acc_dk[dt][4 * g + 3] * tau};
        f32x4 b = {acc_dv[dt][4 * g], acc_dv[dt][4 * g + 1], acc_dv[dt][4 * g + 2], acc_dv[dt][4 * g + 3]};
        *reinterpret_cast<f32x4*>(dkrow + 32 * dt + 8 * g + 4 * h) = a;
        *reinterpret_cast<f32x4*>(dvrow + 32 * dt + 8 * g + 4 * h) = b;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward dQ: same shape as the forward (4 waves x 32 query rows, K/V tiles of BN keys through LDS).
// ---------------------------------------------------------------------------------------------
template <typename T, int D, int BN, int FEAT = 0>
__global__ void __launch_bounds__(256)
bwd_dq_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ dout,
              const float* __restrict__ nlc, const float* __restrict__ ndelta, float* __restrict__ dq, int N, int nqb,
              int BH, Layout lay, int causal, float tau) {
  using A = Atom<T>;
  typedef typename A::frag frag;
  constexpr bool HM = FEAT >= 1, HD = FEAT >= 2;   // key mask (staged as zeros when absent); dropout
  constexpr int KC = D / 16, KT = BN / 32, DT = D / 32;
  constexpr int TB = A::template tile_bytes<D>(BN);
  __shared__ __attribute__((aligned(16))) char smem_raw[4 * TB];
  __shared__ __attribute__((aligned(16))) float smask[HM ? 2 * BN : 4];   // key mask / tau of the two tiles in flight
  lds_char* smem = (lds_char*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bh, qb;
  map_block(blockIdx.x, BH, nqb, bh, qb);
  if (causal) qb = nqb - 1 - qb;
  const int q0 = qb * 128 + w * 32, qrow = q0 + r;
  const bool qvalid = qrow < N;
  const size_t base = head_base(lay, bh);
  const int ld = lay.ld;   // elements between consecutive rows
  const uint32_t mat_bytes = ((uint32_t)(N - 1) * ld + D) * (uint32_t)sizeof(T);
  const rsrc_t qrs = make_rsrc(q + base, mat_bytes);
  const rsrc_t dors = make_rsrc(dout + base, mat_bytes);
  const rsrc_t krs = make_rsrc(k + base, mat_bytes);
  const rsrc_t vrs = make_rsrc(v + base, mat_bytes);
  const float c = tau * LOG2E;

  frag qf[KC], dof[KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const int off = (qrow * ld + 16 * kc + 8 * h) * (int)sizeof(T);
    qf[kc] = load_frag_buf<T>(qrs, off);
    dof[kc] = load_frag_buf<T>(dors, off);
  }
  // this lane's row constants; -delta, in every register, is the accumulator input of the dP^T tiles
  const float nlq = qvalid ? nlc[(size_t)bh * N + qrow] * c : 0.f;   // -L * log2(e)
  const float ndq = qvalid ? ndelta[(size_t)bh * N + qrow] : 0.f;
  f32x16 nd16;
#pragma unroll
  for (int i = 0; i < 16; ++i) nd16[i] = ndq;

  f32x16 acc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc[dt] = zero16();

  const LaneAddr ra = A::template row_addr<D>(lane);
  const LaneAddr ta = A::template tr_addr<D>(lane);
  const int kmax = causal ? min(N, qb * 128 + 128) : N;
  const int nt = (kmax + BN - 1) / BN;
  TileStager<T, D, BN, 256> sk, sv;
  sk.init(tid, ld);
  sv.init(tid, ld);
  sk.load(krs, 0);
  sv.load(vrs, 0);
  sk.store(smem);
  sv.store(smem + 2 * TB);
  const float* mrow = (HM && lay.kmask) ? lay.kmask + (size_t)(bh / lay.mask_heads) * N : nullptr;
  const uint32_t dbase = HD ? drop_base(lay, bh, qrow) : 0u;
  const float inv_tau = 1.0f / tau;
  float mreg = 0.f;
  auto mask_load = [&](int kb0) {
    if constexpr (HM) {
      if (tid < BN) mreg = (mrow != nullptr && kb0 + tid < N) ? mrow[kb0 + tid] * inv_tau : 0.f;
    }
  };
  auto mask_store = [&](int par) {
    if constexpr (HM) {
      if (tid < BN) smask[par * BN + tid] = mreg;
    }
  };
  mask_load(0);
  mask_store(0);
  __syncthreads();

  auto tile = [&](auto par, int t) {
    constexpr int PAR = decltype(par)::value;
    const int kbase = t * BN;
    const bool more = t + 1 < nt;
    if (more) {
      sk.load(krs, kbase + BN);
      sv.load(vrs, kbase + BN);
      mask_load(kbase + BN);
    }
    lds_char* tk = smem + PAR * TB;
    lds_char* tv = smem + (2 + PAR) * TB;
    const bool active = !causal || kbase <= q0 + 31;
    if (active) {
      f32x16 s[KT], dp[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        f32x16 mk16 = zero16();
        if constexpr (HM) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 mk = *reinterpret_cast<const f32x4*>(&smask[PAR * BN + 32 * kt + 8 * g + 4 * h]);
#pragma unroll
            for (int j = 0; j < 4; ++j) mk16[4 * g + j] = mk[j];
          }
        }
        A::mma_c(s[kt], A::template row_frag<D>(tk, ra, 32 * kt, 0), qf[0], mk16);
        if constexpr (HD) A::mma_c(dp[kt], A::template row_frag<D>(tv, ra, 32 * kt, 0), dof[0], zero16());
        else A::mma_c(dp[kt], A::template row_frag<D>(tv, ra, 32 * kt, 0), dof[0], nd16);
#pragma unroll
        for (int kc = 1; kc < KC; ++kc) {
          A::mma(s[kt], A::template row_frag<D>(tk, ra, 32 * kt, kc), qf[kc]);
          A::mma(dp[kt], A::template row_frag<D>(tv, ra, 32 * kt, kc), dof[kc]);
        }
      }
      const bool need_mask = causal && (kbase + BN - 1 > q0);
      frag dsf[KT][2];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kt][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][i], c, nlq));
      if (need_mask) {   // diagonal tiles only (scalar branch)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (kbase + 32 * kt + acc_row(i, h) > qrow) s[kt][i] = 0.f;
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if constexpr (HD) {
            const bool keep = drop_keep(dbase, kbase + 32 * kt + acc_row(i, h), lay.drop_thr);
            dp[kt][i] = s[kt][i] * ((keep ? dp[kt][i] * lay.drop_scale : 0.f) + ndq);
          } else {
            dp[kt][i] = s[kt][i] * dp[kt][i];
          }
        }
        dsf[kt][0] = A::pack(dp[kt], 0);
        dsf[kt][1] = A::pack(dp[kt], 1);
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
            A::mma(acc[dt], A::template tr_frag<D>(tk, ta, 32 * kt + 16 * s2, dt), dsf[kt][s2]);
    }
    if (more) {
      sk.store(smem + (PAR ^ 1) * TB);
      sv.store(smem + (2 + (PAR ^ 1)) * TB);
      mask_store(PAR ^ 1);
    }
    __syncthreads();
  };
  int t = 0;
  for (; t + 1 < nt; t += 2) {
    tile(ic<0>{}, t);
    tile(ic<1>{}, t + 1);
  }
  if (t < nt) tile(ic<0>{}, t);

  if (qvalid) {
    float* row = dq + base + (size_t)qrow * ld;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 val = {acc[dt][4 * g] * tau, acc[dt][4 * g + 1] * tau, acc[dt][4 * g + 2] * tau,
                     acc[dt][4 * g + 3] * tau};
        *reinterpret_cast<f32x4*>(row + 32 * dt + 8 * g + 4 * h) = val;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward dQ, slot-interleaved (bf16, d = 64): a workgroup = 8 waves = 256 query rows (two waves per SIMD), the query
// on the lane as above.  Keys arrive in stages of 128 (K in a three-slot LDS ring, V in the matching slot 48 KiB
// higher, so V reads share K's address registers) and are consumed as 32-key sub-tiles by a three-deep software
// pipeline laid out in MFMA slots (see the dK/dV kernel): period j = 12 slots
//   slots 0-3   S^T(j+1) = K Q^T          slots 4-7   dP^T(j+1) = V dO^T - delta      slots 8-11  dQ^T += K^T dS^T(j-1)
// with the exp / mul / pack of sub-tile j spread over all twelve (24 issue cycles each) and every LDS fragment
// requested four slots ahead.  The pipeline never drains at a stage boundary: the barrier that publishes stage s+1
// sits in the middle of period 4s+2, and a stage's K slot is read (transposed, for dQ) two periods into the next
// stage, hence the third ring slot.  Whole stages are always processed; sub-tiles beyond the causal diagonal or
// N are masked (P = 0).
// ---------------------------------------------------------------------------------------------
// MASKS = false: non-causal launch with N a multiple of 128 (no sub-tile ever needs a mask): the masked period variants and
// their register pressure at the joins disappear.
template <typename T, int D, int DIAG = 0, bool MASKS = true>
__global__ void __launch_bounds__(512)
bwd_dq_slot_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ dout,
                   const float* __restrict__ nlc, const float* __restrict__ ndelta, float* __restrict__ dq, int N, int nqb,
                   int BH, Layout lay, int causal, float tau) {
  static_assert(D == 64 && sizeof(T) == 2, "slot schedule is laid out for bf16, d = 64");
  using A = Atom<T>;
  typedef typename A::frag frag;
  constexpr int KC = D / 16, ST = 128, NT = 512;
  constexpr int TB = A::template tile_bytes<D>(ST);   // 16 KiB
  constexpr int VOFF = 3 * TB;                        // V slot = K slot + 48 KiB
  __shared__ __attribute__((aligned(16))) char smem_raw[6 * TB];
  lds_char* smem = (lds_char*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bh, qb;
  map_block(blockIdx.x, BH, nqb, bh, qb);
  if (causal) qb = nqb - 1 - qb;
  const int q0 = qb * 256 + w * 32, qrow = q0 + r;
  const bool qvalid = qrow < N;
  const size_t base = head_base(lay, bh);
  const int ld = lay.ld;
  const uint32_t mat_bytes = ((uint32_t)(N - 1) * ld + D) * (uint32_t)sizeof(T);
  const rsrc_t qrs = make_rsrc(q + base, mat_bytes);
  const rsrc_t dors = make_rsrc(dout + base, mat_bytes);
  const rsrc_t krs = make_rsrc(k + base, mat_bytes);
  const rsrc_t vrs = make_rsrc(v + base, mat_bytes);
  const float c = tau * LOG2E;

  frag qf[KC], dof[KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const int off = (qrow * ld + 16 * kc + 8 * h) * (int)sizeof(T);
    qf[kc] = load_frag_buf<T>(qrs, off);
    dof[kc] = load_frag_buf<T>(dors, off);
  }
  const float nlq = qvalid ? nlc[(size_t)bh * N + qrow] * c : 0.f;   // -L * log2(e)
  const float ndq = qvalid ? ndelta[(size_t)bh * N + qrow] : 0.f;
  f32x16 nd16;
#pragma unroll
  for (int i = 0; i < 16; ++i) nd16[i] = ndq;
  f32x16 acc[2];
  acc[0] = zero16();
  acc[1] = zero16();

  const LaneAddr ra = A::template row_addr<D>(lane);
  const LaneAddr ta = A::template tr_addr<D>(lane);
  const int kmax = causal ? min(N, qb * 256 + 256) : N;
  const int nstage = (kmax + ST - 1) / ST;
  // Stage loads go global -> LDS directly (buffer_load ... lds, 1 KiB = 8 rows per wave-instruction, no staging
  // registers): LDS-DMA writes lane-linearly, so the image's chunk swizzle is applied to each lane's SOURCE address.
  // Wave w moves the 8-row groups w and w + 8 of K and of V (same parity, hence one lane offset).
  const raw_rsrc_t kraw = make_raw_rsrc(k + base, mat_bytes), vraw = make_raw_rsrc(v + base, mat_bytes);
  const uint32_t smem_addr = (uint32_t)(uintptr_t)smem;
  const int dma_row7 = (lane >> 2) & 7;
  const int dma_voff = dma_row7 * ld * (int)sizeof(T) +
                       16 * (4 * (lane >> 5) + ((lane & 3) ^ ((2 * (w & 1) + (dma_row7 >> 2)) & 3)));
  auto stage_dma = [&](int row0, int slot_base) {
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
      const int g = w + 8 * g2;
      const int soff = (row0 + 8 * g) * ld * (int)sizeof(T);
      dma16(kraw, smem_addr + slot_base + 1024 * g, dma_voff, soff);
      dma16(vraw, smem_addr + slot_base + VOFF + 1024 * g, dma_voff, soff);
    }
  };
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, k_t0 = 0, k_r0 = 0, t0 = 0, t1 = 0;
  if constexpr (DIAG == 1) {
    k_t0 = stamp();
    k_r0 = __builtin_amdgcn_s_memrealtime();
  }
  if constexpr (MASKS) {   // see fwd_slot_kernel: stage rows past N must read as zeros
#pragma unroll 4
    for (int off = tid * 16; off < 6 * TB; off += 512 * 16) *FA_LDS(u32x4, smem + off) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
  }
  stage_dma(0, 0);
  dma_wait_all();   // this wave's pieces have landed
  __syncthreads();
  if constexpr (DIAG == 1) { t0 = stamp(); ph[0] += t0 - k_t0; }

  // sub-tile state: A / B alternate between "being produced" and "being consumed"
  f32x16 sA, dpA, sB, dpB;
  frag dsA0, dsA1, dsB0, dsB1;   // packed dS^T of the sub-tile before the current one / of the current one
  frag rk[4], rv[4], tf[4];
  auto SB = [&]() { __builtin_amdgcn_sched_barrier(0); };
  // LDS readers on a per-stage address register + immediate
  auto krow = [&](int b0, int b1, int sub, int kc) -> frag {
    return *FA_LDS(frag, smem + ((kc & 1) ? b1 : b0) + (D / 32) * 512 * (4 * sub) + 512 * (kc >> 1));
  };
  auto ktr = [&](int b0, int b1, int sub, int s2, int dt) -> frag {
    const int kk = (D / 32) * 512 * (4 * sub + 2 * s2) + 512 * dt;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(FA_LDS(bf16x4, smem + b0 + kk));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(FA_LDS(bf16x4, smem + b1 + kk + (D / 32) * 512));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  // One period.  SUBN: sub-tile (0..3) whose S^T / dP^T are produced [row addresses rn*: its stage]; SUBP: the
  // sub-tile whose dQ product is issued [transposed addresses tp*: its stage]; SUB2: the sub-tile two ahead, whose K
  // rows are requested in slots 8-11 [row addresses r2*].  kcur: first key of the sub-tile in the exp / mul stream.
  auto period = [&](auto hn_c, auto hc_c, auto hp_c, auto mask_c, auto subn_c, auto subp_c, auto sub2_c, int rn0, int rn1,
                    int tp0, int tp1, int r20, int r21, int kcur, f32x16& ns, f32x16& ndp, f32x16& cs, f32x16& cdp,
                    frag& dp0, frag& dp1, frag& dc0, frag& dc1) {
    constexpr bool HN = decltype(hn_c)::value != 0, HC = decltype(hc_c)::value != 0, HP = decltype(hp_c)::value != 0;
    constexpr bool MASK = decltype(mask_c)::value != 0;
    constexpr int SUBN = decltype(subn_c)::value, SUBP = decltype(subp_c)::value, SUB2 = decltype(sub2_c)::value;
    const int qlim = min(qrow, N - 1) - kcur;   // keep key offset o iff o <= qlim (non-causal: only the N bound)
    const int klim = causal ? qlim : (N - 1 - kcur);
    float cm = c;
    if constexpr (MASK) asm volatile("" : "+v"(cm));   // keeps hipcc from hoisting the masked and unmasked variants' common fma
    auto fe = [&](int i) {
      float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(cs[i], cm, nlq));
      if constexpr (MASK) pv = (acc_row(i, h) > klim) ? 0.f : pv;
      cs[i] = pv;
    };
    auto md = [&](int i) { cdp[i] = cs[i] * cdp[i]; };
    auto vrow = [&](int kq) -> frag {
      return *FA_LDS(frag, smem + ((kq & 1) ? rn1 : rn0) + VOFF + (D / 32) * 512 * (4 * SUBN) + 512 * (kq >> 1));
    };
    // LDS fragments are requested LEAD slots before the MFMA that consumes them: 4 in the build without masked periods
    // (186 VGPRs), 2 where the masked variants' joins leave no registers for more
    constexpr int LEAD = MASKS ? 2 : 4;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {   // slots 0-3: S^T of the next sub-tile
      if constexpr (HN) {
        if (kq == 0) A::mma_c(ns, rk[0], qf[0], zero16());
        else A::mma(ns, rk[kq], qf[kq]);
        SB();   // the MFMA opens its slot; the fillers follow in its shadow
        if constexpr (LEAD == 4) rv[kq] = vrow(kq);
        else if (kq >= 2) rv[kq - 2] = vrow(kq - 2);
      }
      if constexpr (HC) { fe(2 * kq); fe(2 * kq + 1); }
      SB();
    }
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {   // slots 4-7: dP^T of the next sub-tile
      if constexpr (HN) {
        if (kq == 0) A::mma_c(ndp, rv[0], dof[0], nd16);
        else A::mma(ndp, rv[kq], dof[kq]);
        SB();
        if constexpr (LEAD == 2) {
          if (kq < 2) rv[kq + 2] = vrow(kq + 2);
        }
      }
      if constexpr (HC) {
        if (kq == 0) {
#pragma unroll
          for (int i = 0; i < 6; ++i) md(i);
        } else if (kq == 1) {
          md(6); md(7); dc0 = A::pack(cdp, 0);
        } else {
          fe(4 + 2 * kq); fe(5 + 2 * kq);
        }
      }
      if constexpr (HP) {
        if constexpr (LEAD == 4) tf[kq] = ktr(tp0, tp1, SUBP, kq >> 1, kq & 1);
        else if (kq >= 2) tf[kq - 2] = ktr(tp0, tp1, SUBP, 0, kq & 1);
      }
      SB();
    }
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {   // slots 8-11: dQ^T of the previous sub-tile
      if constexpr (HP) {
        A::mma(acc[kq & 1], tf[kq], (kq < 2) ? dp0 : dp1);
        SB();
        if constexpr (LEAD == 2) {
          if (kq < 2) tf[2 + kq] = ktr(tp0, tp1, SUBP, 1, kq);
        }
      }
      if constexpr (HC) {
        if (kq < 2) {
          fe(12 + 2 * kq); fe(13 + 2 * kq);
        } else if (kq == 2) {
#pragma unroll
          for (int i = 8; i < 14; ++i) md(i);
        } else {
          md(14); md(15); dc1 = A::pack(cdp, 1);
        }
      }
      if constexpr (HN) {
        if constexpr (LEAD == 4) {
          rk[kq] = krow(r20, r21, SUB2, kq);
        } else if (kq >= 2) {
          rk[2 * (kq - 2)] = krow(r20, r21, SUB2, 2 * (kq - 2));
          rk[2 * (kq - 2) + 1] = krow(r20, r21, SUB2, 2 * (kq - 2) + 1);
        }
      }
      SB();
    }
  };
  auto T1 = ic<1>{};
  auto T0 = ic<0>{};
  // per-stage address registers: K slot of stage s is s % 3
  auto slot_of = [&](int st) { return (st % 3) * TB; };
  int cr0 = ra.b[0], cr1 = ra.b[1];                 // rows of the current stage (slot 0)
  int ct0 = ta.b[0], ct1 = ta.b[1];                 // transposed reads of the current stage
  int pt0 = ct0, pt1 = ct1;                         // ... of the previous stage (stage 0: any finite data, dS = 0)
  // prologue: rows of sub-tile 0, then S^T(0), dP^T(0)
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) rk[kc] = krow(cr0, cr1, 0, kc);
  dsB0 = A::zero();
  dsB1 = A::zero();
  SB();
  period(T1, T0, T0, T0, ic<0>{}, ic<0>{}, ic<1>{}, cr0, cr1, ct0, ct1, cr0, cr1, 0, sA, dpA, sB, dpB, dsB0, dsB1, dsA0, dsA1);
  for (int st = 0; st < nstage; ++st) {
    const bool more = st + 1 < nstage;
    const int nb = slot_of(st + 1);
    const int nr0 = ra.b[0] + nb, nr1 = ra.b[1] + nb;   // rows of the next stage
    if (more) stage_dma((st + 1) * ST, nb);
    const int kb = st * ST;
    // a sub-tile needs the mask when it crosses N or (causal) this wave's first query; wave-uniform
    auto need = [&](int sub) { return MASKS && ((kb + 32 * sub + 31 >= N) || (causal && kb + 32 * sub + 31 > q0)); };
    // period 4st+0: produce sub 1 (this stage), consume sub 0, dQ of sub 3 of the previous stage
    if constexpr (MASKS) {
      if (need(0)) period(T1, T1, T1, T1, ic<1>{}, ic<3>{}, ic<2>{}, cr0, cr1, pt0, pt1, cr0, cr1, kb, sB, dpB, sA, dpA, dsB0, dsB1, dsA0, dsA1);
      else period(T1, T1, T1, T0, ic<1>{}, ic<3>{}, ic<2>{}, cr0, cr1, pt0, pt1, cr0, cr1, kb, sB, dpB, sA, dpA, dsB0, dsB1, dsA0, dsA1);
    } else {
      period(T1, T1, T1, T0, ic<1>{}, ic<3>{}, ic<2>{}, cr0, cr1, pt0, pt1, cr0, cr1, kb, sB, dpB, sA, dpA, dsB0, dsB1, dsA0, dsA1);
    }
    // period 4st+1: produce sub 2, consume sub 1, dQ of sub 0
    if constexpr (MASKS) {
      if (need(1)) period(T1, T1, T1, T1, ic<2>{}, ic<0>{}, ic<3>{}, cr0, cr1, ct0, ct1, cr0, cr1, kb + 32, sA, dpA, sB, dpB, dsA0, dsA1, dsB0, dsB1);
      else period(T1, T1, T1, T0, ic<2>{}, ic<0>{}, ic<3>{}, cr0, cr1, ct0, ct1, cr0, cr1, kb + 32, sA, dpA, sB, dpB, dsA0, dsA1, dsB0, dsB1);
    } else {
      period(T1, T1, T1, T0, ic<2>{}, ic<0>{}, ic<3>{}, cr0, cr1, ct0, ct1, cr0, cr1, kb + 32, sA, dpA, sB, dpB, dsA0, dsA1, dsB0, dsB1);
    }
    // the next stage goes to LDS and is published before the second half of period 4st+2 asks for its rows
    if constexpr (DIAG == 1) { t1 = stamp(); ph[1] += t1 - t0; }
    dma_wait_all();   // this wave's pieces of the next stage have landed
    if constexpr (DIAG == 1) { t0 = stamp(); ph[2] += t0 - t1; }
    if constexpr (DIAG != 2) __syncthreads();   // DIAG 2: timing ablation without the per-stage barrier (results are wrong)
    if constexpr (DIAG == 1) { t1 = stamp(); ph[3] += t1 - t0; t0 = t1; }
    // period 4st+2: produce sub 3, consume sub 2, dQ of sub 1; rows two ahead = sub 0 of the next stage
    if constexpr (MASKS) {
      if (need(2)) period(T1, T1, T1, T1, ic<3>{}, ic<1>{}, ic<0>{}, cr0, cr1, ct0, ct1, nr0, nr1, kb + 64, sB, dpB, sA, dpA, dsB0, dsB1, dsA0, dsA1);
      else period(T1, T1, T1, T0, ic<3>{}, ic<1>{}, ic<0>{}, cr0, cr1, ct0, ct1, nr0, nr1, kb + 64, sB, dpB, sA, dpA, dsB0, dsB1, dsA0, dsA1);
    } else {
      period(T1, T1, T1, T0, ic<3>{}, ic<1>{}, ic<0>{}, cr0, cr1, ct0, ct1, nr0, nr1, kb + 64, sB, dpB, sA, dpA, dsB0, dsB1, dsA0, dsA1);
    }
    // period 4st+3: produce sub 0 of the next stage, consume sub 3, dQ of sub 2
    if constexpr (MASKS) {
      if (need(3)) period(T1, T1, T1, T1, ic<0>{}, ic<2>{}, ic<1>{}, nr0, nr1, ct0, ct1, nr0, nr1, kb + 96, sA, dpA, sB, dpB, dsA0, dsA1, dsB0, dsB1);
      else period(T1, T1, T1, T0, ic<0>{}, ic<2>{}, ic<1>{}, nr0, nr1, ct0, ct1, nr0, nr1, kb + 96, sA, dpA, sB, dpB, dsA0, dsA1, dsB0, dsB1);
    } else {
      period(T1, T1, T1, T0, ic<0>{}, ic<2>{}, ic<1>{}, nr0, nr1, ct0, ct1, nr0, nr1, kb + 96, sA, dpA, sB, dpB, dsA0, dsA1, dsB0, dsB1);
    }
    pt0 = ct0; pt1 = ct1;
    cr0 = nr0; cr1 = nr1;
    ct0 = ta.b[0] + nb; ct1 = ta.b[1] + nb;
  }
  // drain: dQ of the last sub-tile (sub 3 of the last stage); the "produced" sub-tile of the last period is unused
  period(T0, T0, T1, T0, ic<0>{}, ic<3>{}, ic<0>{}, cr0, cr1, pt0, pt1, cr0, cr1, 0, sB, dpB, sA, dpA, dsB0, dsB1, dsA0, dsA1);

  if constexpr (DIAG == 1) {
    const unsigned long long k_t1 = stamp(), k_r1 = __builtin_amdgcn_s_memrealtime();
    ph[1] += k_t1 - t0;
    const int slot = blockIdx.x * 8 + w;
    if (slot < 8192 && lane == 0) {
      for (int j = 0; j < 6; ++j) g_phase_cycles[slot * 8 + j] = ph[j];
      g_phase_cycles[slot * 8 + 6] = k_t1 - k_t0;
      g_phase_cycles[slot * 8 + 7] = k_r1 - k_r0;
    }
  }
  if (qvalid) {
    float* row = dq + base + (size_t)qrow * ld;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 val = {acc[dt][4 * g] * tau, acc[dt][4 * g + 1] * tau, acc[dt][4 * g + 2] * tau,
                     acc[dt][4 * g + 3] * tau};
        *reinterpret_cast<f32x4*>(row + 32 * dt + 8 * g + 4 * h) = val;
      }
  }
}

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
