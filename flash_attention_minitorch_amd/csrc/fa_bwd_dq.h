// FlashAttention backward for MI355X (gfx950): the query-stationary dQ kernels.
// Part of the kernel set described in fa_kernels.h (included from there, inside its include order).
#pragma once
#include "fa_common.h"

namespace fa {

// pa.o != nullptr (both dQ kernels, plain builds; a run-time switch since round 3, one build instead of two): the launch also does the
// backward's preprocess (bwd_prep_kernel) for its own query rows: every wave forms
// -delta = -rowsum(dO * O) and -L / tau of its 32 rows from the forward's O and side outputs (its dO fragments are in registers
// anyway; the query is on the lane), uses them, and stores them to the workspace, from where the dK/dV kernel, launched AFTER this
// one, takes them as it always did.  One launch and one pass over dO less per backward (the preprocess kernel: 0.020 ms of the
// 1.14 ms step at the metric shape).
struct DqPrep {
  const float* o;      // forward output, fp32, same layout as q
  const float* l;      // FA-2: logsumexp; FA-1: sum exp(s - m)
  const float* m;      // FA-1: row maximum (else unused)
  float* nlc;          // workspace, written: -L / tau
  float* ndelta;       // workspace, written: -rowsum(dO * O)
  float* nl2;          // workspace, written: -L * log2(e)  (what the dK/dV slot kernel takes as its accumulator input)
  int aux_mode;
  float inv_tau;
};
// What bwd_prep_kernel does, for one wave's 32 query rows (the query is on the lane; lanes r and r + 32 hold the two halves of each
// 16-column chunk of the row, dof = this lane's dO fragments): forms -L/tau and -delta, stores them, returns the lane's constants.
template <typename T, int KC>
FA_DEV void dq_prep_rows(const DqPrep& pa, const typename Atom<T>::frag (&dof)[KC], size_t base, int bh, int N, int qrow, int ld, int h,
                         bool qvalid, float c, float& nlq, float& ndq) {
  float sum = 0.f, nl = 0.f;
  if (qvalid) {
    const float* orow = pa.o + base + (size_t)qrow * ld + 8 * h;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(orow + 16 * kc), b = *reinterpret_cast<const f32x4*>(orow + 16 * kc + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) sum += a[j] * (float)dof[kc][j] + b[j] * (float)dof[kc][4 + j];
    }
    const size_t ri = (size_t)bh * N + qrow;
    const float L = (pa.aux_mode == AUX_FA1) ? (pa.m[ri] + __logf(pa.l[ri])) : pa.l[ri];
    nl = (L == -INFINITY) ? -INFINITY : -L * pa.inv_tau;
  }
  sum = xhalf_sum(sum);
  if (qvalid && h == 0) {
    pa.nlc[(size_t)bh * N + qrow] = nl;
    pa.ndelta[(size_t)bh * N + qrow] = -sum;
    pa.nl2[(size_t)bh * N + qrow] = nl * c;
  }
  nlq = qvalid ? nl * c : 0.f;
  ndq = qvalid ? -sum : 0.f;
}


// ---------------------------------------------------------------------------------------------
// Backward dQ: same shape as the forward (NWQ waves x 32 query rows, K/V tiles of BN keys through LDS).  NWQ = 4 by
// default; the bf16 d = 128 launch uses 8 (one workgroup per CU sharing each staged tile between twice the waves).
// ---------------------------------------------------------------------------------------------
template <typename T, int D, int BN, int FEAT = 0, int NWQ = 4, bool CARE = false>   // CARE: as fwd_kernel's
__global__ void __launch_bounds__(64 * NWQ)
bwd_dq_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ dout,
              const float* __restrict__ nlc, const float* __restrict__ ndelta, float* __restrict__ dq, int N, int nqb,
              int BH, Layout lay, int causal, float tau, int only_qb = -1, DqPrep pa = DqPrep{}) {
  if (guard_skip(lay)) return;   // guarded call: this launch is not the chosen one of its pair
  constexpr bool CAN_PREP = FEAT == 0 && !CARE;   // the preprocess is folded into the plain main build only (the launcher knows)
  using A = Atom<T>;   // only_qb: as fwd_kernel's
  typedef typename A::frag frag;
  constexpr bool HM = FEAT >= 1, HD = FEAT >= 2;   // key mask (staged as zeros when absent); dropout
  constexpr int KC = D / 16, KT = BN / 32, DT = D / 32;
  constexpr int TB = A::template tile_bytes<D>(BN);
  __shared__ __attribute__((aligned(16))) char smem_raw[4 * TB];
  __shared__ __attribute__((aligned(16))) float smask[HM ? 2 * BN : 4];   // key mask / tau of the two tiles in flight
  lds_char* smem = (lds_char*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // causal launches pair query block p with block nqb-1-p in one workgroup (see fwd_kernel): uniform work per workgroup
  // ... or, with lay.rank_chunk set, one block per workgroup dispatched longest first across a chunk of heads (map_block_ranked)
  const bool ranked = only_qb < 0 && causal && lay.rank_chunk > 0;
  const int nblk = only_qb >= 0 ? 1 : ((causal && !ranked) ? (nqb + 1) / 2 : nqb);
  int bh, pblk;
  if (ranked) map_block_ranked(blockIdx.x, BH, nblk, lay.rank_chunk, bh, pblk);
  else map_block(blockIdx.x, BH, nblk, bh, pblk);
  const int npass = (only_qb < 0 && causal && !ranked && pblk != nqb - 1 - pblk) ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
  const int qb = only_qb >= 0 ? only_qb : (causal ? (pass == 0 ? nqb - 1 - pblk : pblk) : pblk);
  const int q0 = qb * (32 * NWQ) + w * 32, qrow = q0 + r;
  const bool qvalid = qrow < N;
  const bool careful = CARE && A::SPLITS && (HM || HD || (causal ? q0 < 64 : N < 64));   // wave-uniform: see fwd_kernel
  const size_t base = head_base(lay, bh);
  const int ld = lay.ld;   // elements between consecutive rows
  const uint32_t mat_bytes = ((uint32_t)(N - 1) * ld + D) * (uint32_t)sizeof(T);
  rsrc_t qrs = make_rsrc(q + base, mat_bytes);
  rsrc_t dors = make_rsrc(dout + base, mat_bytes);
  raw_rsrc_t kraw = make_raw_rsrc(k + base, mat_bytes), vraw = make_raw_rsrc(v + base, mat_bytes);
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
  float nlq, ndq;
  if (CAN_PREP && pa.o != nullptr) {   // (kernel argument: a scalar branch)
    dq_prep_rows<T, KC>(pa, dof, base, bh, N, qrow, ld, h, qvalid, c, nlq, ndq);
  } else {
    nlq = qvalid ? nlc[(size_t)bh * N + qrow] * c : 0.f;   // -L * log2(e)
    ndq = qvalid ? ndelta[(size_t)bh * N + qrow] : 0.f;
  }
  f32x16 nd16;
#pragma unroll
  for (int i = 0; i < 16; ++i) nd16[i] = ndq;

  f32x16 acc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc[dt] = zero16();

  const LaneAddr ra = A::template row_addr<D>(lane);
  const LaneAddr ta = A::template tr_addr<D>(lane);
  const int kmax = causal ? min(N, qb * (32 * NWQ) + 32 * NWQ) : N;
  const int nt = (kmax + BN - 1) / BN;
  TileStager<T, D, BN, 64 * NWQ> sk, sv;
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
      // keys past N read K = 0, so S = 0 and P = exp2(-L * log2e): finite garbage for ordinary rows, but +inf for a row whose
      // logsumexp lies below about -88 (then dS = inf * 0 = NaN poisons the row's dQ): the ragged tail is masked like the diagonal
      const bool need_mask = (kbase + BN > N) || (causal && (kbase + BN - 1 > q0));
      frag dsf[KT][2];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kt][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][i], c, nlq));
      if (need_mask) {   // diagonal tiles and the ragged last tile only (scalar branch)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = kbase + 32 * kt + acc_row(i, h);
            if (key >= N || (causal && key > qrow)) s[kt][i] = 0.f;
          }
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
      if (!careful) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
              A::mma(acc[dt], A::template tr_frag<D>(tk, ta, 32 * kt + 16 * s2, dt), dsf[kt][s2]);
      } else {   // rows with few admissible keys: K^T dS^T also takes what the bf16 rounding of dS dropped
        frag dl[KT][2];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          dl[kt][0] = A::pack_lo(dp[kt], 0, dsf[kt][0]);
          dl[kt][1] = A::pack_lo(dp[kt], 1, dsf[kt][1]);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              const frag kt_ = A::template tr_frag<D>(tk, ta, 32 * kt + 16 * s2, dt);
              A::mma(acc[dt], kt_, dsf[kt][s2]);
              A::mma(acc[dt], kt_, dl[kt][s2]);
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
  }   // pass
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
// CDIAG = true (MASKS = false, N a multiple of 256): the causal launch, as fwd_slot_kernel's: the pipeline sweeps the 2 * qb full
// stages in front of the workgroup's first query without a mask; the 256 keys of its own diagonal block are the next two stages of
// the ring.  Waves 4-7 sweep the first of them too (it lies wholly in front of their first query; round 3); what is left is taken
// wave by wave in plain per-sub-tile form: wave w multiplies sub-tiles 0..w (waves 0-3) or 4..w (waves 4-7) and masks the last one;
// rows 0..63 split dS into two bf16 fragments there (Atom::pack_lo).  Query blocks p and nqb-1-p share a workgroup (uniform work).
// TILED = true (MASKS = false, non-causal; round 3): a workgroup takes query block qb of lay.tiles CONSECUTIVE HEADS, one after the
// other, without leaving the pipeline's ring (as the tiled dK/dV build: the workgroups of all query blocks of a head group move from
// head to head together, so a head's K / V stream is still shared through the XCD's L2; consecutive query blocks of ONE head per
// workgroup keep every head's K / V live at once and measured 474 instead of 273 MB of HBM traffic per launch): the last stage
// iteration of a head requests key stage 0 of the NEXT head into the next ring slot (the ring simply goes on: stage j of the next head
// is ring stage nstage + j), so the next head starts at its prologue period with no DMA wait and no barrier, and its Q / dO / O loads
// run while the dQ stores of the finished one drain.
template <typename T, int D, int DIAG = 0, bool MASKS = true, bool CDIAG = false, bool TILED = false>
__global__ void __launch_bounds__(512)
bwd_dq_slot_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ dout,
                   const float* __restrict__ nlc, const float* __restrict__ ndelta, float* __restrict__ dq, int N, int nqb,
                   int BH, Layout lay, int causal, float tau, DqPrep pa = DqPrep{}) {
  if (guard_skip(lay)) return;   // guarded call: this launch is not the chosen one of its pair
  static_assert(D == 64 && sizeof(T) == 2, "slot schedule is laid out for bf16, d = 64");
  using A = Atom<T>;
  typedef typename A::frag frag;
  constexpr int KC = D / 16, ST = 128;
  constexpr int TB = A::template tile_bytes<D>(ST);   // 16 KiB
  constexpr int VOFF = 3 * TB;                        // V slot = K slot + 48 KiB
  __shared__ __attribute__((aligned(16))) char smem_raw[6 * TB];
  lds_char* smem = (lds_char*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  static_assert(!CDIAG || (!MASKS && DIAG == 0), "causal build: unmasked sweep + diagonal block");
  static_assert(!TILED || (!MASKS && !CDIAG && DIAG == 0), "tiled build: non-causal, unmasked, no stamps");
  const int tiles = TILED ? max(lay.tiles, 1) : 1;   // heads per workgroup (the launcher sizes the grid with BH / tiles head groups)
  const bool ranked = CDIAG && causal == 2;   // A/B: one block per workgroup, heaviest blocks of all heads first
  const int nblk = (CDIAG && !ranked) ? (nqb + 1) / 2 : nqb;
  int bh, pblk;
  if (ranked) map_block_ranked(blockIdx.x, BH, nblk, max(lay.rank_chunk, 1), bh, pblk);
  else map_block(blockIdx.x, BH / tiles, nblk, bh, pblk);
  bh *= tiles;
  size_t base = head_base(lay, bh);
  const int ld = lay.ld;
  const uint32_t mat_bytes = ((uint32_t)(N - 1) * ld + D) * (uint32_t)sizeof(T);
  rsrc_t qrs = make_rsrc(q + base, mat_bytes);
  rsrc_t dors = make_rsrc(dout + base, mat_bytes);
  raw_rsrc_t kraw = make_raw_rsrc(k + base, mat_bytes), vraw = make_raw_rsrc(v + base, mat_bytes);
  const float c = tau * LOG2E;
  const int npass = TILED ? tiles : ((CDIAG && !ranked && pblk != nqb - 1 - pblk) ? 2 : 1);
  int roff = 0;   // tiled build: ring position of the current block's stage 0
  for (int pass = 0; pass < npass; ++pass) {
  const int qb = CDIAG ? (pass == 0 ? nqb - 1 - pblk : pblk) : (causal ? nqb - 1 - pblk : pblk);
  if (CDIAG && pass) __syncthreads();   // every wave is done with the diagonal stages of the first block
  if (TILED && pass) {   // the next head (its K / V descriptors were set when its stage 0 was requested)
    ++bh;
    base = head_base(lay, bh);
    qrs = make_rsrc(q + base, mat_bytes);
    dors = make_rsrc(dout + base, mat_bytes);
  }
  const int q0 = qb * 256 + w * 32, qrow = q0 + r;
  const bool qvalid = qrow < N;

  frag qf[KC], dof[KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const int off = (qrow * ld + 16 * kc + 8 * h) * (int)sizeof(T);
    qf[kc] = load_frag_buf<T>(qrs, off);
    dof[kc] = load_frag_buf<T>(dors, off);
  }
  float nlq, ndq;
  if (!MASKS && pa.o != nullptr) {   // (kernel argument: a scalar branch; the masked build never folds the preprocess in)
    dq_prep_rows<T, KC>(pa, dof, base, bh, N, qrow, ld, h, qvalid, c, nlq, ndq);
  } else {
    nlq = qvalid ? nlc[(size_t)bh * N + qrow] * c : 0.f;   // -L * log2(e)
    ndq = qvalid ? ndelta[(size_t)bh * N + qrow] : 0.f;
  }
  // PRE (the builds without masked periods): tau*log2(e) is folded into the Q fragments once per block (re-rounded to bf16) and
  // -L*log2(e) enters S^T as the accumulator input of its MFMA chain, so P = exp2(S') is ONE instruction per score.  Rows with fewer
  // than 64 admissible keys (causal build, query block 0: no sweep, only the diagonal block) keep the unscaled Q and the fp32 fma.
  constexpr bool PRE = !MASKS;
  const bool exactq = CDIAG && A::SPLITS && q0 < 64;   // wave-uniform
  // Both scalings in one launch (Layout::scale_sel; the mask-free builds): exact = the Q fragments stay unscaled, -L/tau (raw score
  // units) enters S^T as the accumulator input and every score is multiplied in fp32, P = exp2(c * S').
  const bool exact = PRE && scale_exact(lay);   // wave-uniform
  if (PRE && !exactq && !exact) {
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) qf[kc] = A::scale(qf[kc], c);
  }
  // -delta enters dP^T as the accumulator input of its MFMA chain (sixteen registers holding one value).  The build with masked
  // periods has no registers for that (it spilled three around the loop): there the VALU adds it, dS = P * (dP + (-delta)).
  constexpr bool NDACC = !MASKS;
  f32x16 nd16, nl16;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    nd16[i] = NDACC ? ndq : 0.f;
    nl16[i] = PRE ? (exact ? nlq * (1.0f / c) : nlq) : 0.f;
  }
  f32x16 acc[2];
  acc[0] = zero16();
  acc[1] = zero16();

  const LaneAddr ra = A::template row_addr<D>(lane);
  const LaneAddr ta = A::template tr_addr<D>(lane);
  const int kmax = CDIAG ? qb * 256 : (causal ? min(N, qb * 256 + 256) : N);
  const int nstage = (kmax + ST - 1) / ST;
  // Stage loads go global -> LDS directly (buffer_load ... lds, 1 KiB = 8 rows per wave-instruction, no staging
  // registers): LDS-DMA writes lane-linearly, so the image's chunk swizzle is applied to each lane's SOURCE address.
  // Wave w moves the 8-row groups w and w + 8 of K and of V (same parity, hence one lane offset).
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
  // Causal build: waves 4-7 keep sweeping through the first stage of the workgroup's own diagonal block (its 128 keys lie wholly in
  // front of their first query); every wave takes part in all stage hand-offs (DMA share, wait, barrier), waves 0-3 without periods
  // in the last one.
  const int nst_w = CDIAG ? nstage + (w >> 2) : nstage;   // stages this wave sweeps (wave-uniform)
  const int nst_all = CDIAG ? nstage + 1 : nstage;        // stage hand-offs every wave takes part in
  {
  if (!(TILED && pass > 0)) {   // (tiled build: the last iteration of the block before requested and published this stage)
    stage_dma(0, 0);
    dma_wait_all();   // this wave's pieces have landed
    __syncthreads();
  }
  if constexpr (DIAG == 1) { t0 = stamp(); ph[0] += t0 - k_t0; }

  if (lay.young_prio && w >= 4) __builtin_amdgcn_s_setprio(1);   // the later-dispatched half loses VALU arbitration otherwise
  // (the mask-free builds carry two copies of the sweep, one per scaling: the fp32 multiply exists in the exact one's stream only)
  auto sweep = [&](auto ex_c) {
  constexpr bool EX = decltype(ex_c)::value != 0;
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
      float pv = PRE ? (EX ? __builtin_amdgcn_exp2f(cs[i] * c) : __builtin_amdgcn_exp2f(cs[i]))
                     : __builtin_amdgcn_exp2f(__builtin_fmaf(cs[i], cm, nlq));
      if constexpr (MASK) pv = (acc_row(i, h) > klim) ? 0.f : pv;
      cs[i] = pv;
    };
    auto md = [&](int i) { cdp[i] = NDACC ? cs[i] * cdp[i] : cs[i] * (cdp[i] + ndq); };
    auto vrow = [&](int kq) -> frag {
      return *FA_LDS(frag, smem + ((kq & 1) ? rn1 : rn0) + VOFF + (D / 32) * 512 * (4 * SUBN) + 512 * (kq >> 1));
    };
    // LDS fragments are requested LEAD slots before the MFMA that consumes them: 4 in the build without masked periods
    // (186 VGPRs), 2 where the masked variants' joins leave no registers for more
    constexpr int LEAD = MASKS ? 2 : 4;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {   // slots 0-3: S^T of the next sub-tile
      if constexpr (HN) {
        if (kq == 0) A::mma_c(ns, rk[0], qf[0], PRE ? nl16 : zero16());
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
        if (kq == 0) A::mma_c(ndp, rv[0], dof[0], NDACC ? nd16 : zero16());
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
  auto slot_of = [&](int st) { return ((st + roff) % 3) * TB; };
  const int b0_ = slot_of(0);
  int cr0 = ra.b[0] + b0_, cr1 = ra.b[1] + b0_;     // rows of the current stage (slot 0; tiled build: where the ring stands)
  int ct0 = ta.b[0] + b0_, ct1 = ta.b[1] + b0_;     // transposed reads of the current stage
  int pt0 = ct0, pt1 = ct1;                         // ... of the previous stage (stage 0: any finite data, dS = 0)
  // prologue: rows of sub-tile 0, then S^T(0), dP^T(0)
  // (causal build, ADVICE r3: a wave with no stage to sweep -- waves 0-3 of query block 0 -- runs neither the prologue nor the drain
  // period: the drain would multiply dS = 0 into K rows beyond the wave's causal horizon, and 0 * Inf there is NaN where the reference,
  // which never touches those rows, is finite)
  if (CDIAG && nst_w == 0) return;
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) rk[kc] = krow(cr0, cr1, 0, kc);
  dsB0 = A::zero();
  dsB1 = A::zero();
  SB();
  period(T1, T0, T0, T0, ic<0>{}, ic<0>{}, ic<1>{}, cr0, cr1, ct0, ct1, cr0, cr1, 0, sA, dpA, sB, dpB, dsB0, dsB1, dsA0, dsA1);
  for (int st = 0; st < nst_w; ++st) {
    const bool more = st + 1 < nstage;
    const int nb = slot_of(st + 1);
    const int nr0 = ra.b[0] + nb, nr1 = ra.b[1] + nb;   // rows of the next stage
    if (CDIAG || more) stage_dma((st + 1) * ST, nb);   // (causal build: the two stages of the diagonal block follow the sweep)
    else if (TILED && pass + 1 < npass) {   // the next head's sweep: its key stage 0 follows in the ring
      const size_t nbase = head_base(lay, bh + 1);
      kraw = make_raw_rsrc(k + nbase, mat_bytes);
      vraw = make_raw_rsrc(v + nbase, mat_bytes);
      stage_dma(0, nb);
    }
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
  };
  if constexpr (PRE) {
    if (exact) sweep(ic<1>{});
    else sweep(ic<0>{});
  } else {
    sweep(ic<0>{});
  }
  auto slot_of = [&](int st) { return ((st + roff) % 3) * TB; };   // (as inside the sweep)
  if constexpr (TILED) roff = (roff + nstage) % 3;
  if constexpr (CDIAG) {   // the stage hand-off this wave has no periods for (same DMA share, wait and barrier as in the loop)
    for (int st = nst_w; st < nst_all; ++st) {
      stage_dma((st + 1) * ST, slot_of(st + 1));
      dma_wait_all();
      __syncthreads();
    }
  }
  }

  if constexpr (CDIAG) {
    // The diagonal block: keys kmax .. kmax + 255 = stages nstage, nstage + 1 of the ring (slots nstage % 3, (nstage + 1) % 3), both
    // requested and published by the hand-offs above.  Left for the plain per-sub-tile form: sub-tiles 0..w (waves 0-3), 4..w
    // (waves 4-7, which swept the block's first stage): at most four per wave instead of up to eight.
    const bool careful = exactq;   // wave-uniform
    const float cmd = (exactq || exact) ? c : 1.0f;   // (the other waves' scores leave the MFMA chain in log2 units)
    for (int j = 4 * (w >> 2); j <= w; ++j) {
      const int sb = ((nstage + (j >> 2)) % 3) * TB;
      lds_char* tk = smem + sb;
      lds_char* tv = tk + VOFF;
      const int row32 = 32 * (j & 3);
      f32x16 s, dp;
      A::mma_c(s, A::template row_frag<D>(tk, ra, row32, 0), qf[0], zero16());
      A::mma_c(dp, A::template row_frag<D>(tv, ra, row32, 0), dof[0], nd16);
#pragma unroll
      for (int kc = 1; kc < KC; ++kc) {
        A::mma(s, A::template row_frag<D>(tk, ra, row32, kc), qf[kc]);
        A::mma(dp, A::template row_frag<D>(tv, ra, row32, kc), dof[kc]);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], cmd, nlq));
      if (j == w) {   // this wave's own 32 keys: key kmax + 32 * w + row against query q0 + r
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (acc_row(i, h) > r) s[i] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) dp[i] = s[i] * dp[i];
      const frag ds0 = A::pack(dp, 0), ds1 = A::pack(dp, 1);
      if (!careful) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
            A::mma(acc[dt], A::template tr_frag<D>(tk, ta, row32 + 16 * s2, dt), s2 ? ds1 : ds0);
      } else {   // rows with fewer than 64 admissible keys: K^T dS^T also takes what the bf16 rounding of dS dropped
        const frag dl0 = A::pack_lo(dp, 0, ds0), dl1 = A::pack_lo(dp, 1, ds1);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const frag kt_ = A::template tr_frag<D>(tk, ta, row32 + 16 * s2, dt);
            A::mma(acc[dt], kt_, s2 ? ds1 : ds0);
            A::mma(acc[dt], kt_, s2 ? dl1 : dl0);
          }
      }
    }
  }

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
  // (the output address is formed HERE from opaque copies: computed before the loop, hipcc keeps it in three registers the
  // masked build does not have and spills them around the loop)
  int qr = qrow, hh = h;
  asm volatile("" : "+v"(qr), "+v"(hh));
  if (qr < N) {
    float* row = dq + base + (size_t)qr * ld;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 val = {acc[dt][4 * g] * tau, acc[dt][4 * g + 1] * tau, acc[dt][4 * g + 2] * tau,
                     acc[dt][4 * g + 3] * tau};
        *reinterpret_cast<f32x4*>(row + 32 * dt + 8 * g + 4 * hh) = val;
      }
  }
  }   // pass
}


}  // namespace fa
