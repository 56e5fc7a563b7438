// FlashAttention forward kernels for MI355X (gfx950): phased (fwd_kernel) and MFMA-slot pipeline (fwd_slot_kernel).
// Part of the kernel set described in fa_kernels.h (included from there, inside its include order).
#pragma once
#include "fa_common.h"

namespace fa {

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

// CARE: the build with the split-operand path for rows with few admissible keys (see `careful` below); the launcher runs it for
// whole launches that need it everywhere (key mask, dropout, N < 64) and, behind a causal launch, for query block 0 alone.
template <typename T, int D, int BN, int WPE, int FEAT = 0, bool CARE = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE)))
fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, float* __restrict__ o,
           float* __restrict__ aux_l, float* __restrict__ aux_m, int N, int nqb, int BH, Layout lay, int causal,
           int aux_mode, float tau, int only_qb = -1) {
  if (guard_skip(lay)) return;   // guarded call: this launch is not the chosen one of its pair
  // only_qb >= 0: one workgroup per (batch*head) that handles just that 128-query block (the launcher re-runs block 0 behind a
  // slot kernel forced onto a causal launch: the slot kernels have no split-operand path for the rows with few keys)
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
  // Causal launches pair query block p with block nqb-1-p in one workgroup (heavy one first): every workgroup then sweeps the
  // same number of key tiles and the grid has no long tail (the launcher sizes the grid with fwd_blocks()).
  // ... or, with lay.rank_chunk set, one block per workgroup dispatched longest first across a chunk of heads (map_block_ranked)
  const bool ranked = only_qb < 0 && causal && lay.rank_chunk > 0;
  // (lay.twin_blocks > 1: the launch is the fp32-scaling twin of a guarded non-causal call -- it almost always returns at the guard
  // check above, and a quarter of the workgroups return in a quarter of the time; each takes that many consecutive query blocks)
  const int tb = (!causal && only_qb < 0 && lay.twin_blocks > 1) ? lay.twin_blocks : 1;
  const int nblk = only_qb >= 0 ? 1 : ((causal && !ranked) ? (nqb + 1) / 2 : nqb / tb);
  int bh, pblk;
  if (ranked) map_block_ranked(blockIdx.x, BH, nblk, lay.rank_chunk, bh, pblk);
  else map_block(blockIdx.x, BH, nblk, bh, pblk);
  const int npass = tb > 1 ? tb : ((only_qb < 0 && causal && !ranked && pblk != nqb - 1 - pblk) ? 2 : 1);
  for (int pass = 0; pass < npass; ++pass) {
  const int qb = only_qb >= 0 ? only_qb : (causal ? (pass == 0 ? nqb - 1 - pblk : pblk) : pblk * tb + pass);
  const int q0 = qb * 128 + w * 32, qrow = q0 + r;
  const bool qvalid = qrow < N;
  // wave-uniform: this wave's rows may see fewer than 64 admissible keys (or a mask / dropout thins them): operands that the
  // second product takes in bf16 are then split into two fragments (Atom::pack_lo).  bf16 only: the fp32 atom is exact.
  const bool careful = CARE && A::SPLITS && (HM || HD || (causal ? q0 < 64 : N < 64));
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
      if (!careful) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
              A::mma(acc_o[dt], A::template tr_frag<D>(tv, ta, 32 * kt + 16 * s2, dt), pf[kt][s2]);
      } else {   // rows with few admissible keys: P.V also takes what the bf16 rounding of P dropped (each V fragment feeds both)
        frag pl[KT][2];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          pl[kt][0] = A::pack_lo(s[kt], 0, pf[kt][0]);
          pl[kt][1] = A::pack_lo(s[kt], 1, pf[kt][1]);
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
              const frag vt = A::template tr_frag<D>(tv, ta, 32 * kt + 16 * s2, dt);
              A::mma(acc_o[dt], vt, pf[kt][s2]);
              A::mma(acc_o[dt], vt, pl[kt][s2]);
            }
      }
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
    const size_t orow = base + (size_t)qrow * ld;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 val = {acc_o[dt][4 * g] * inv, acc_o[dt][4 * g + 1] * inv, acc_o[dt][4 * g + 2] * inv,
                     acc_o[dt][4 * g + 3] * inv};
        store_out4(o, orow + 32 * dt + 8 * g + 4 * h, val, lay.out_bf16);
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
  }   // pass
}

// ---------------------------------------------------------------------------------------------
// Cold path of the reference-free slot builds (fwd_slot_kernel, MASKS = false): one wave redoes its 32 query rows the classic way
// (running maximum, O and l rescaled, exact fp32 scaling of the UNSCALED Q, P split into two bf16 fragments), everything straight from
// global memory: no LDS, no barrier, so the other waves of the workgroup are not involved.  Taken when a row's sum of exp2(S') left
// [2^-96, 2^96] (or is NaN): |tau q.k| beyond about 58 somewhere in the row.  Slow (scalar gathers of V^T), correct for any input the
// phased kernel is correct for.  On return acc_o / l_run are relative to m_ref, the row maximum in log2 units (tau*log2e * q.k).
// ---------------------------------------------------------------------------------------------
template <int D>
FA_DEV void fwd_redo_rows(rsrc_t qrs, rsrc_t krs, rsrc_t vrs, int qrow, int q0, int ld, int N, bool causal, float c, int r, int h,
                          f32x16 (&acc_o)[D / 32], float& m_ref, float& l_run) {
  using A = Atom<bf16_t>;
  typedef A::frag frag;
  constexpr int KC = D / 16, DT = D / 32;
  // every address below is formed from opaque copies of the lane's coordinates: hoisted to kernel entry (they are loop invariants
  // of the caller) they would be spilled around its pipeline
  asm volatile("" : "+v"(r), "+v"(h), "+v"(qrow));
  frag qf[KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) qf[kc] = load_frag_buf<bf16_t>(qrs, (qrow * ld + 16 * kc + 8 * h) * 2);
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc_o[dt] = zero16();
  m_ref = 0.f;
  l_run = 0.f;
  const int kend = causal ? min(N, q0 + 32) : N;   // wave-uniform
  for (int k0 = 0; k0 < kend; k0 += 32) {
    f32x16 s;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const frag kk = load_frag_buf<bf16_t>(krs, ((k0 + r) * ld + 16 * kc + 8 * h) * 2);   // rows >= N read as zero
      if (kc == 0) A::mma_c(s, kk, qf[0], zero16());
      else A::mma(s, kk, qf[kc]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = k0 + acc_row(i, h);
      if (key >= N || (causal && key > qrow)) s[i] = -INFINITY;
    }
    float mx = s[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
    mx = xhalf_max(mx) * c;   // log2 units; key 0 is admissible for every row, so the first sub-tile's maximum is finite
    float alpha = 1.0f;
    if (k0 == 0) {
      m_ref = mx;
    } else {
      const float delta = fmaxf(mx - m_ref, 0.f);
      if (__any(delta > 0.f)) {
        alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc_o[dt][i] *= alpha;
        m_ref += delta;
      }
    }
    const float nm = -m_ref;
    float rs = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], c, nm));
      rs += s[i];
    }
    l_run = l_run * alpha + rs;
    const frag pf0 = A::pack(s, 0), pf1 = A::pack(s, 1);
    const frag pl0 = A::pack_lo(s, 0, pf0), pl1 = A::pack_lo(s, 1, pf1);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
        u16x8 raw;
#pragma unroll
        for (int j = 0; j < 8; ++j)   // element j: V[k0 + 16*s2 + 8*(j>>2) + 4*h + (j&3)][32*dt + r]  (Atom::tr_frag's map)
          raw[j] = __builtin_amdgcn_raw_buffer_load_b16(vrs, ((k0 + 16 * s2 + 8 * (j >> 2) + 4 * h + (j & 3)) * ld + 32 * dt + r) * 2, 0, 0);
        const frag vt = __builtin_bit_cast(frag, raw);
        A::mma(acc_o[dt], vt, s2 ? pf1 : pf0);
        A::mma(acc_o[dt], vt, s2 ? pl1 : pl0);
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
// STK: keys per stage (default: 16 KiB of K and of V); MINW: waves per SIMD the register allocation must allow (STK = 64 at
// d = 64 halves the rings to 64 KiB, two workgroups per CU = four waves per SIMD at <= 128 VGPRs).
// CDIAG = true (MASKS = false, 64-key stages, N a multiple of 256): the causal launch.  The pipeline above sweeps the keys in
// front of the workgroup's first query (4 * qb full stages, no sub-tile of them needs a mask, so the unmasked build keeps its
// registers and its four waves per SIMD); the 256 keys of the workgroup's own diagonal block are the NEXT four stages of the same
// ring.  Wave w keeps sweeping through the first w / 2 of them (they lie wholly in front of its first query; every wave takes part
// in all stage hand-offs, without periods once its own sweep has ended) and takes what is left, sub-tiles 2 * (w / 2) .. w of the
// block, the classic way (true maximum, reference moved), masking the last one: at most two plain sub-tiles per wave instead of up
// to eight (round 3).  Rows 0..63 (fewer than 64 admissible keys) split P into two bf16 fragments there (Atom::pack_lo), which is
// what the phased CARE build does for them.
template <typename T, int D, bool MASKS = true, int DIAG = 0, int STK = 8192 / D, int MINW = 2, bool CDIAG = false>
__global__ void __launch_bounds__(512, MINW)
fwd_slot_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, float* __restrict__ o,
                float* __restrict__ aux_l, int N, int nqb, int BH, Layout lay, int causal, float tau) {
  if (guard_skip(lay)) return;   // guarded call: this launch is not the chosen one of its pair
  static_assert((D == 64 || D == 128) && sizeof(T) == 2, "slot schedule is laid out for bf16, d = 64 / 128");
  using A = Atom<T>;
  typedef typename A::frag frag;
  constexpr int KC = D / 16, DT = D / 32, NS = 2 * KC, EPS = 16 / NS;   // slots per period, scores per slot
  constexpr int ST = STK;                             // keys per stage
  constexpr int NSUBT = ST / 32;                      // sub-tiles per stage: 4 (d = 64) or 2 (d = 128)
  constexpr int R = NSUBT == 2 ? 4 : 3;               // ring slots
  constexpr int TB = A::template tile_bytes<D>(ST);   // 16 KiB (8 KiB with 64-key stages at d = 64)
  constexpr int PCS = TB / 8192;                      // 1 KiB DMA pieces per wave and tensor
  constexpr int VOFF = R * TB;
  constexpr int SUBB = (D / 32) * 512 * 4;            // bytes of one 32-key sub-tile inside a stage image
  static_assert((TB == 16384 || TB == 8192) && 2 * DT == KC && (NSUBT == 2 || NSUBT == 4), "stage geometry");
  static_assert(!CDIAG || (!MASKS && NSUBT == 2 && R * ST == 256), "causal build: the ring holds one 256-key diagonal block");
  // PRE (every build without masked periods): tau*log2(e) is folded into the Q fragments once per block (re-rounded to bf16), so
  // S' = K (cQ)^T leaves the MFMA chain in log2 units, and the sweep runs WITHOUT a reference: P = exp2(S'), one v_exp + one v_add
  // per score instead of fma + exp + add, no row maximum, no redo branch.  fp32 / bf16 carry exp2(S') at full relative precision
  // for |S'| up to ~100 (|tau q.k| up to ~58 natural units); a wave whose row sum left [2^-96, 2^96] redoes its rows in the cold
  // path above (fwd_redo_rows).  Rows with fewer than 64 admissible keys (causal build, query block 0) keep the unscaled Q and the
  // exact fp32 scaling: the rounding of cQ is common to all keys of a row and is only averaged out over many keys.
  constexpr bool PRE = !MASKS;
  __shared__ __attribute__((aligned(16))) char smem_raw[2 * R * TB];
  lds_char* smem = (lds_char*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, r0_ = lane & 31, h0_ = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // The causal build pairs query block p with block nqb-1-p in one workgroup (heavy one first): every workgroup sweeps the same
  // number of stages (the launcher sizes the grid with (nqb + 1) / 2 blocks per batch*head).
  const bool ranked = CDIAG && causal == 2;   // A/B: one block per workgroup, heaviest blocks of all heads first
  const int nblk = (CDIAG && !ranked) ? (nqb + 1) / 2 : nqb;
  int bh, pblk;
  if (ranked) map_block_ranked(blockIdx.x, BH, nblk, max(lay.rank_chunk, 1), bh, pblk);
  else map_block(blockIdx.x, BH, nblk, bh, pblk);
  const size_t base = head_base(lay, bh);
  const int ld = lay.ld;
  const uint32_t mat_bytes = ((uint32_t)(N - 1) * ld + D) * (uint32_t)sizeof(T);
  const rsrc_t qrs = make_rsrc(q + base, mat_bytes);
  const raw_rsrc_t kraw = make_raw_rsrc(k + base, mat_bytes), vraw = make_raw_rsrc(v + base, mat_bytes);
  const float c = tau * LOG2E;
  const int npass = (CDIAG && !ranked && pblk != nqb - 1 - pblk) ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
  const int qb = CDIAG ? (pass == 0 ? nqb - 1 - pblk : pblk) : (causal ? nqb - 1 - pblk : pblk);
  if (CDIAG && pass) __syncthreads();   // every wave is done with the diagonal image of the first block
  // (causal build: the lane's row constants are re-derived where they are needed -- here, behind the sweep, in the epilogue -- so
  // that the pipeline, which sits at its 128 registers, does not carry them)
  int r = r0_, h = h0_;
  if constexpr (CDIAG) {
    const int l2 = lane_fresh();
    r = l2 & 31;
    h = l2 >> 5;
  }
  const int q0 = qb * 256 + w * 32;
  int qrow = q0 + r;

  frag qf[KC];
  // This launch produces the call's scale guard (fa_common.h: guard_produce): the key rows first, on their own, then the query rows,
  // which stay.  Not the causal build: it sits at its 128 registers and spilled two to three around this in either order (causal
  // calls take the separate guard pass).
  const bool produce = PRE && !CDIAG && lay.guard_want == 2 && lay.guard != nullptr;
  frag kg[KC];   // (the key rows of the query rows' indices: requested here, summed behind the first stages' LDS-DMA, see below)
  if (produce) {
    const rsrc_t krs = make_rsrc(k + base, mat_bytes);
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) kg[kc] = load_frag_buf<T>(krs, (qrow * ld + 16 * kc + 8 * h) * (int)sizeof(T));
  }
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) qf[kc] = load_frag_buf<T>(qrs, (qrow * ld + 16 * kc + 8 * h) * (int)sizeof(T));
  const bool exactq = CDIAG && A::SPLITS && q0 < 64;   // wave-uniform: rows with fewer than 64 admissible keys (query block 0 only)
  const float cm = (PRE && !exactq) ? 1.0f : c;   // what a score of this wave's MFMA chain is multiplied by to reach log2 units
  f32x16 acc_o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc_o[dt] = zero16();
  float m_ref = 0.f, nmc = 0.f, l_run = 0.f;   // m_ref: raw score units; PRE builds: log2 units (0 throughout the sweep)

  const LaneAddr ra = A::template row_addr<D>(lane);
  const LaneAddr ta = A::template tr_addr<D>(lane);
  const int kmax = CDIAG ? qb * 256 : (causal ? min(N, qb * 256 + 256) : N);
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
    for (int g2 = 0; g2 < PCS; ++g2) {
      const int piece = w + 8 * g2, g = piece / PPG;
      const int soff = (row0 + 8 * g) * ld * (int)sizeof(T);
      dma16(kraw, smem_addr + slot_base + 1024 * piece, dma_voff, soff);
      dma16(vraw, smem_addr + slot_base + VOFF + 1024 * piece, dma_voff, soff);
    }
  };
  auto slot_of = [&](int st) { return (st % R) * TB; };
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, k_t0 = 0, k_r0 = 0, t0 = 0, t1 = 0;
  if constexpr (DIAG == 1) {
    k_t0 = stamp();
    k_r0 = __builtin_amdgcn_s_memrealtime();
  }
  if constexpr (MASKS) {   // ragged launches read stage rows past N: make sure they are zeros whatever an out-of-range
    // LDS-DMA lane does (0 * stale NaN bits would poison P.V); 96 / 128 KiB once per workgroup
#pragma unroll 4
    for (int off = tid * 16; off < 2 * R * TB; off += 512 * 16) *FA_LDS(u32x4, smem + off) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
  }
  // Causal build: the sweep also takes, wave by wave, the stages of the workgroup's own diagonal block that lie wholly in front of
  // the wave's first query (wave w: the first w / 2 of the block's four 64-key stages); every wave runs the same number of stage
  // hand-offs (DMA share, wait, barrier), the ones past its own sweep without the periods.
  const int nst_w = CDIAG ? nstage + (w >> 1) : nstage;   // stages this wave sweeps (wave-uniform)
  const int nst_all = CDIAG ? nstage + 3 : nstage;        // stage hand-offs every wave takes part in
  {
  stage_dma(0, 0);
  if (NSUBT == 2 && (CDIAG || nstage > 1)) stage_dma(ST, slot_of(1));
  if (produce) {   // the guard's row norms, while the first stages are in flight (scratch: the tail of the ring, written by stage R - 1 first)
    float qs = 0.f, ks = 0.f;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      qs += frag_sumsq(qf[kc]);
      ks += frag_sumsq(kg[kc]);
    }
    if (guard_produce(lay, qs, ks, smem + 2 * R * TB - 64)) {   // (workgroup-uniform)
      dma_wait_all();   // (no LDS-DMA may land after the workgroup has given its LDS back)
      return;
    }
  }
  if (PRE && !exactq) {
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) qf[kc] = A::scale(qf[kc], c);
  }
  dma_wait_all();
  __syncthreads();
  if constexpr (DIAG == 1) { t0 = stamp(); ph[0] += t0 - k_t0; }

  if (lay.young_prio && w >= 4) __builtin_amdgcn_s_setprio(1);   // the later-dispatched half loses VALU arbitration otherwise
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
      const float pv = PRE ? __builtin_amdgcn_exp2f(cs[i]) : __builtin_amdgcn_exp2f(__builtin_fmaf(cs[i], cm, nmc));
      cs[i] = pv;
      rs += pv;
    };
    // softmax work of slot g: its EPS scores, then the bf16 pack of the pairs completed by the slot before (the last
    // slot also packs its own).  (The row sum on v_pk_add_f32 -- 8 adds per sub-tile instead of 16 -- measured the same time, guarded
    // and unguarded, in same-process A/Bs: 0.2469 vs 0.2471 and 0.2383 vs 0.2381 ms; profiles/README.md, round 4.  Not kept.)
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
    if constexpr (HC && PRE) l_run += rs;   // no reference to move: the range check sits at the end of the block
    if constexpr (HC && !PRE) {
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
  if constexpr (!PRE) {
    const bool m0 = MASKS && ((31 >= N) || (causal && 31 > q0));
    if (m0) mask_scores(sA, 0);
    m_ref = tile_max(sA);      // key 0 is never masked, so the maximum is finite
    nmc = -m_ref * c;
  }
  for (int st = 0; st < nst_w; ++st) {
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
      if constexpr (DIAG == 1) { t1 = stamp(); ph[1] += t1 - t0; }
      dma_wait_all();   // this wave's pieces of the next stage have landed
      if constexpr (DIAG == 1) { t0 = stamp(); ph[2] += t0 - t1; }
      __syncthreads();
      if constexpr (DIAG == 1) { t1 = stamp(); ph[3] += t1 - t0; t0 = t1; }
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
        if constexpr (DIAG == 1) { t1 = stamp(); ph[1] += t1 - t0; }
        dma_wait_all();
        if constexpr (DIAG == 1) { t0 = stamp(); ph[2] += t0 - t1; }
        if constexpr (DIAG != 2) __syncthreads();   // DIAG 2: timing ablation without the per-stage barrier (results are wrong)
        if constexpr (DIAG == 1) { t1 = stamp(); ph[3] += t1 - t0; t0 = t1; }
      }
      if (CDIAG ? st + 2 < nstage + 4 : st + 2 < nstage) stage_dma((st + 2) * ST, slot_of(st + 2));   // (causal build: the diagonal block's four stages follow)
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
  if constexpr (CDIAG) {
    // (a wave with no stage to sweep -- the first two of query block 0 -- has just multiplied P = 0 into V rows beyond its
    // causal horizon: 0 * Inf there is NaN where the reference, which never touches those rows, is finite.  Its accumulators hold
    // nothing yet: cleared here.  Skipping the two periods instead cost the build, which sits at 128 registers, 152 B of scratch.)
    if (nst_w == 0) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) acc_o[dt] = zero16();
    }
  }
  if constexpr (CDIAG) {   // the stage hand-offs this wave has no periods for (same DMA share, wait and barrier as in the loop)
    for (int st = nst_w; st < nst_all; ++st) {
      if (st > 0) {
        dma_wait_all();
        __syncthreads();
      }
      if (st + 2 < nstage + 4) stage_dma((st + 2) * ST, slot_of(st + 2));
    }
  }
  }

  if constexpr (CDIAG) {
    // The diagonal block: keys kmax .. kmax + 255 = stages nstage .. nstage + 3 of the ring = slots 0 .. 3 (nstage % 4 == 0), i.e.
    // ONE 256-row image of K at smem and of V at smem + VOFF, all of it requested by the sweep's hand-offs.  What is left for the
    // classic per-sub-tile form: the sub-tiles from the wave's last swept stage up to its own (one or two of them).  (The last
    // hand-off above published the block's last stage: no further wait.)
    {
      const int l2 = lane_fresh();
      r = l2 & 31;
      h = l2 >> 5;
      qrow = q0 + r;
    }
    const bool careful = exactq;   // wave-uniform
    auto diag_tile = [&](int j, bool first) {
      f32x16 s;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const frag kk = A::template row_frag<D>(smem, ra, 32 * j, kc);
        if (kc == 0) A::mma_c(s, kk, qf[0], zero16());
        else A::mma(s, kk, qf[kc]);
      }
      if (j == w) {   // this wave's own 32 keys: key kmax + 32 * w + row against query q0 + r
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (acc_row(i, h) > r) s[i] = -INFINITY;
      }
      float mx = s[0];   // finite: the sub-tile's first key is admissible for every row
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
      mx = xhalf_max(mx) * cm;   // log2 units
      float alpha = 1.0f;
      if (first) {
        m_ref = mx;
      } else {   // (behind a sweep the reference is 0: it only ever moves up)
        const float delta = fmaxf(mx - m_ref, 0.f);
        if (__any(delta > 0.f)) {
          alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_o[dt][i] *= alpha;
          m_ref += delta;
        }
      }
      nmc = -m_ref;
      float rs = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], cm, nmc));
        rs += s[i];
      }
      l_run = l_run * alpha + rs;
      const frag pf0 = A::pack(s, 0), pf1 = A::pack(s, 1);
      if (!careful) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
            A::mma(acc_o[dt], A::template tr_frag<D>(smem + VOFF, ta, 32 * j + 16 * s2, dt), s2 ? pf1 : pf0);
      } else {   // rows with fewer than 64 admissible keys: P.V also takes what the bf16 rounding of P dropped
        const frag pl0 = A::pack_lo(s, 0, pf0), pl1 = A::pack_lo(s, 1, pf1);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const frag vt = A::template tr_frag<D>(smem + VOFF, ta, 32 * j + 16 * s2, dt);
            A::mma(acc_o[dt], vt, s2 ? pf1 : pf0);
            A::mma(acc_o[dt], vt, s2 ? pl1 : pl0);
          }
      }
    };
    for (int j = 2 * (w >> 1); j <= w; ++j) diag_tile(j, nst_w == 0 && j == 0);
  }

  if constexpr (DIAG == 1) { t1 = stamp(); ph[1] += t1 - t0; t0 = t1; }
  float l_tot = xhalf_sum(l_run);
  if constexpr (PRE) {   // a row sum outside [2^-96, 2^96] (or NaN): exp2(S') over- or underflowed somewhere in the wave's rows
    if (__any(!(l_tot >= 0x1p-96f && l_tot <= 0x1p96f))) {
      const uint32_t kv_bytes = ((uint32_t)(N - 1) * ld + D) * (uint32_t)sizeof(T);
      if constexpr (CDIAG) {
        const int l2 = lane_fresh();
        r = l2 & 31;
        h = l2 >> 5;
        qrow = q0 + r;
      }
      fwd_redo_rows<D>(qrs, make_rsrc(k + base, kv_bytes), make_rsrc(v + base, kv_bytes), qrow, q0, ld, N, CDIAG, c, r, h, acc_o,
                       m_ref, l_run);
      l_tot = xhalf_sum(l_run);
    }
  }
  const float inv = 1.0f / l_tot;
  // (the output addresses are formed HERE from opaque copies: computed before the loop, hipcc keeps the 64-bit row pointers live
  // through the pipeline, and the causal d = 64 build, which sits at its 128 registers, spills them around the loop)
  int qr = qrow, hh = h;
  if constexpr (CDIAG) {
    const int l2 = lane_fresh();
    qr = q0 + (l2 & 31);
    hh = l2 >> 5;
  } else {
    asm volatile("" : "+v"(qr), "+v"(hh));
  }
  if (qr < N) {
    const size_t orow = base + (size_t)qr * ld;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 val = {acc_o[dt][4 * g] * inv, acc_o[dt][4 * g + 1] * inv, acc_o[dt][4 * g + 2] * inv,
                     acc_o[dt][4 * g + 3] * inv};
        store_out4(o, orow + 32 * dt + 8 * g + 4 * hh, val, lay.out_bf16);
      }
    if (hh == 0)
      aux_l[(size_t)bh * N + qr] = PRE ? (m_ref + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f : m_ref * tau + __logf(l_tot);
  }
  if constexpr (DIAG == 1) {
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
  }   // pass
}


}  // namespace fa
