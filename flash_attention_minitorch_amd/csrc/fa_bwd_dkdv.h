// FlashAttention backward for MI355X (gfx950): preprocess and the key-stationary dK / dV kernels.
// Part of the kernel set described in fa_kernels.h (included from there, inside its include order).
#pragma once
#include "fa_common.h"

namespace fa {

// ---------------------------------------------------------------------------------------------
// Backward preprocess: ndelta = -rowsum(dO * O), nlc = -L / tau (raw score units) with L = m + log(l) (FA-1 side
// outputs) or L = l (FA-2), so that P = exp2(tau*log2e * ((q.k) + nlc)) and dS = P * (dO.V^T + ndelta): both row
// constants enter the main kernels as MFMA accumulator inputs (S' = Q.K^T + nlc, dP' = dO.V^T + ndelta).  The reference recomputes D_i per (i, j) tile
// (src/flash_attn_bw.cu:194-197); once per row gives the same value.  nl2 = -L * log2(e) is the same constant in log2 units, for
// the kernels whose key / query operand carries tau*log2(e) already (bwd_dkdv_slot_kernel: P = exp2(Q.(cK)^T + nl2)).
// ---------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ void __launch_bounds__(256)
bwd_prep_kernel(const float* __restrict__ o, const T* __restrict__ dout, const float* __restrict__ l,
                const float* __restrict__ m, float* __restrict__ nlc, float* __restrict__ ndelta, float* __restrict__ nl2,
                long rows, int N, Layout lay, int aux_mode, float inv_tau) {
  if (guard_skip(lay)) return;   // guarded call: this launch is not the chosen one of its pair
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
    nl2[row] = (L == -INFINITY) ? -INFINITY : -L * LOG2E;
  }
}

// ---------------------------------------------------------------------------------------------
// Backward dK / dV: a workgroup = NW waves = NW*KPW keys of one (batch*head); each wave keeps K, V fragments and
// the dK^T, dV^T accumulators of its KPW keys in registers while the workgroup sweeps 32-row query slices
// (Q, dO tiles + their nlc, delta staged in LDS, double buffered).
// ---------------------------------------------------------------------------------------------
// CARE: the build whose per-sub-slice path splits P and dS into two bf16 fragments on rows with few admissible keys (bf16 only).
// thin_mode (causal launches): 1 = this launch SKIPS the sub-slices of queries 0..63 (the rows with fewer than 64 keys); 2 = this
// launch handles ONLY those (key block 0, one workgroup per batch*head, CARE build) and ADDS its dK, dV to what mode 1 stored.
template <typename T, int D, int KPW, int NW, int QS, int MODE = 0, bool HD = false, int MINW = 1, bool CARE = false, bool PAIR = false>
__global__ void __launch_bounds__(NW * 64, MINW)   // MINW: minimum waves per SIMD the register allocation must allow
bwd_dkdv_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ dout,
                const float* __restrict__ nlc, const float* __restrict__ ndelta, float* __restrict__ dk,
                float* __restrict__ dv, int N, int nkb, int BH, Layout lay, int causal, float tau, int thin_mode = 0) {
  if (guard_skip(lay)) return;   // guarded call: this launch is not the chosen one of its pair
  // PAIR (builds for causal launches): one workgroup handles key block p and then key block nkb-1-p (heavy one first): under the causal mask
  // key block kb sweeps nqi - kb*BK/QS query stages, so paired workgroups all do the same work and the grid has no long tail
  // (the launcher sizes the grid with (nkb + 1) / 2 workgroups per batch*head); cf. the paired query blocks of fwd_kernel.
  using A = Atom<T>;
  typedef typename A::frag frag;
  constexpr int KC = D / 16, KT = KPW / 32, DT = D / 32, BK = NW * KPW, NT = NW * 64, NSUB = QS / 32;
  constexpr int TB = A::template tile_bytes<D>(QS);
  constexpr int BUF = 2 * TB + 8 * QS;  // Q tile, dO tile, QS x nlc, QS x -delta
  __shared__ __attribute__((aligned(16))) char smem_raw[2 * BUF];
  lds_char* smem = (lds_char*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bh, pblk;
  // (unpaired causal launches: key block 0 sweeps the most query stages; with lay.rank_chunk set the blocks are dispatched longest
  // first across a chunk of heads instead of head by head, cf. map_block_ranked)
  if (!PAIR && thin_mode != 2 && causal && lay.rank_chunk > 0) map_block_ranked(blockIdx.x, BH, nkb, lay.rank_chunk, bh, pblk);
  else map_block(blockIdx.x, BH, thin_mode == 2 ? 1 : (PAIR ? (nkb + 1) / 2 : nkb), bh, pblk);
  const int npass = (PAIR && pblk != nkb - 1 - pblk) ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
  const int kb = (PAIR && pass == 1) ? nkb - 1 - pblk : pblk;
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
  const int nqi = thin_mode == 2 ? 1 : (N + QS - 1) / QS;   // (mode 2: queries 0..63 lie in the first stage)
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
  // Stages whose query rows may see fewer than 64 admissible keys (the first rows under the causal mask, N < 64, a key mask or
  // dropout thinning the row) take the per-sub-slice path, where P and dS enter the dV / dK products as two bf16 fragments each
  // (Atom::pack_lo): on such rows their 2^-9 rounding is not averaged out (bf16 only; wave-uniform).
  const bool thin = CARE && A::SPLITS && (HD || lay.kmask != nullptr);
  // (causal: the stage of queries 0..63 always takes the per-sub-slice path, which is also where thin_mode skips / selects)
  auto careful_stage = [&](int qi_) { return A::SPLITS && (thin || (causal ? qi_ * QS < 64 : (CARE && N < 64))); };
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
      const bool fast3 = !careful_stage(qi) && (kw0 < N) && (!causal || qi * QS >= kw0 + KPW - 1);   // wave-uniform
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
    const bool fast = PIPE && !careful_stage(qi) && (kw0 < N) && (!causal || qi * QS >= kw0 + KPW - 1);   // wave-uniform
    const bool fast_slot = SLOT && !careful_stage(qi) && (kw0 < N) && (!causal || qi * QS >= kw0 + KPW - 1);
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
      const bool active = (kw0 < N) && (qi0 < N) && (!causal || qi0 + 31 >= kw0) &&
                          !(thin_mode == 1 && qi0 < 64) && !(thin_mode == 2 && qi0 >= 64);  // wave-uniform
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
        if (!(CARE && (thin || (A::SPLITS && (causal ? qi0 < 64 : N < 64))))) {
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
        } else {
          // rows with few admissible keys: both products also take what the bf16 rounding of P, dS dropped (the residual fragments
          // are formed first, so the fp32 tiles are dead before the MFMAs; each transposed fragment feeds both)
          frag pl[KT][2], dl[KT][2];
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              pl[kt][s2] = A::pack_lo(s[kt], s2, pf[kt][s2]);
              dl[kt][s2] = A::pack_lo(dp[kt], s2, dsf[kt][s2]);
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
                A::mma(acc_dv[dt][kt], adoT, pl[kt][s2]);
                A::mma(acc_dk[dt][kt], aqT, dsf[kt][s2]);
                A::mma(acc_dk[dt][kt], aqT, dl[kt][s2]);
              }
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
          f32x4* pk = reinterpret_cast<f32x4*>(dkrow + 32 * dt + 8 * g + 4 * h);
          f32x4* pv = reinterpret_cast<f32x4*>(dvrow + 32 * dt + 8 * g + 4 * h);
          if (thin_mode == 2) {   // the rows' other contributions were stored by the launch before this one (same stream)
            a += *pk;
            b += *pv;
          }
          *pk = a;
          *pv = b;
        }
    }
  }
  }   // pass
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
// CDIAG = true (N a multiple of 256): the causal launch.  Key block kb sweeps the query stages BELOW its diagonal block (stages
// 2 * (kb + 1) .. nst - 1, no sub-slice of them needs a mask) through the pipeline; the two stages of the diagonal block itself
// follow in the ring (requested by the last iteration) and are taken wave by wave in plain per-sub-slice form: wave w multiplies
// sub-slices w..7 of the block and masks the first one; queries 0..63 split P and dS into two bf16 fragments there
// (Atom::pack_lo).  Workgroups are ranked longest-first across all heads (map_block_ranked), so the grid ends level.
// TILED = true (non-causal, N a multiple of 256): a workgroup takes key block kb of lay.tiles CONSECUTIVE HEADS, one after the
// other, without leaving the pipeline's ring (the workgroups of all key blocks of a head group move from head to head together, so
// a head's Q / dO stream is still shared through the XCD's L2; consecutive key blocks of ONE head per workgroup lost that sharing
// and measured 9 % slower at B = 32): the last stage iteration of a head requests stage 0 of the next head into the next ring slot,
// the next head's K / V fragments are requested before the dK / dV stores of the finished one are issued, and nothing waits for
// those stores.  In-kernel stamps (tools/phase_cycles.py 393) put the un-overlapped head and tail of a
// one-block workgroup at 8 % of its life (1.6 k cycles of set-up, 5.5 k waiting for fragments and stage 0, 7 k until the stores
// have drained, of 200 k): with one workgroup per CU nothing else runs there meanwhile.
// Scaling: tau*log2(e) is folded into the K fragments once per key block (re-rounded to bf16) and the row constant arrives in log2
// units (nl2 = -L*log2(e), the workspace's third vector), so S' = Q (cK)^T + nl2 leaves the MFMA chain as the exp2 argument: ONE
// instruction per score.  The sub-slices of queries 0..63 in the causal build's diagonal block (fewer than 64 admissible keys) take
// the unscaled K and the fp32 fma instead.  The launcher never sends a key mask here.
template <typename T, int D, int DIAG = 0, bool CDIAG = false, bool TILED = false>
__global__ void __launch_bounds__(512)
bwd_dkdv_slot_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ dout,
                     const float* __restrict__ nl2, const float* __restrict__ ndelta, float* __restrict__ dk,
                     float* __restrict__ dv, int N, int nkb, int BH, Layout lay, float tau) {
  if (guard_skip(lay)) return;   // guarded call: this launch is not the chosen one of its pair
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
  static_assert(!CDIAG || DIAG == 0, "causal build: no stamps");
  unsigned long long k_t00 = 0;
  if constexpr (DIAG) k_t00 = stamp();   // first instruction of the wave
  static_assert(!TILED || DIAG == 0, "tiled builds: no stamps");
  // CT (round 4): the causal tiled build.  A workgroup takes key blocks p (heavy: it sweeps the most query stages) and nkb-1-p (light)
  // of lay.tiles consecutive heads, one UNIT after the other, without leaving the ring: every workgroup does the same work (a pair
  // sweeps nkb-1 blocks' worth of stages plus two diagonal blocks), so no ranking is needed; the first stage of the next unit is
  // requested while the diagonal block of the current one is worked off, its K / V fragments before the dK / dV stores are issued.
  constexpr bool CT = CDIAG && TILED;
  const int tiles = TILED ? max(lay.tiles, 1) : 1;   // heads per workgroup (the launcher sizes the grid with BH / tiles head groups)
  int bh, kb;   // (causal build: key block 0 sweeps the most query stages: the heaviest blocks of all heads are dispatched first)
  int pair = 0;
  if (CT) {
    map_block(blockIdx.x, BH / tiles, nkb / 2, bh, pair);   // (the launcher sends even nkb only)
    kb = pair;
  } else if (CDIAG) map_block_ranked(blockIdx.x, BH, nkb, max(lay.rank_chunk, 1), bh, kb);
  else map_block(blockIdx.x, BH / tiles, nkb, bh, kb);
  bh *= tiles;
  const int nunits = CT ? 2 * tiles : 1;
  size_t base = head_base(lay, bh);
  const int ld = lay.ld;
  const uint32_t mat_bytes = ((uint32_t)(N - 1) * ld + D) * (uint32_t)sizeof(T);
  rsrc_t krs = make_rsrc(k + base, mat_bytes);
  rsrc_t vrs = make_rsrc(v + base, mat_bytes);
  raw_rsrc_t qraw = make_raw_rsrc(q + base, mat_bytes), doraw = make_raw_rsrc(dout + base, mat_bytes);
  // Both scalings in one launch (Layout::scale_sel): exact = the K fragments stay unscaled, the row constant is -L/tau (raw score
  // units: the workspace's FIRST vector, two vectors in front of nl2) and every score is multiplied in fp32, P = exp2(c * S').
  const bool exact = scale_exact(lay);   // wave-uniform
  const float* nlv = exact ? nl2 - 2 * (size_t)BH * N : nl2;
  raw_rsrc_t nlraw = make_raw_rsrc(nlv + (size_t)bh * N, (uint32_t)N * 4u);
  raw_rsrc_t ndraw = make_raw_rsrc(ndelta + (size_t)bh * N, (uint32_t)N * 4u);
  const float c = tau * LOG2E;
  int kw0 = kb * BK + w * KPW;
  const bool active = CT || kw0 < N;   // wave-uniform: a wave whose keys all lie past N only moves data and joins the barriers
                                       // (tiled builds: N is a multiple of 256, every wave of every block is active)
  frag kf[KC], vf[KC];
  auto load_kv = [&](int k0) {
    int rr = r, hh = h;
    if constexpr (CDIAG && TILED) {   // (lane constants re-derived where the unit loop needs them instead of carried through its sweep)
      const int l2 = lane_fresh();
      rr = l2 & 31;
      hh = l2 >> 5;
    }
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int off = ((k0 + rr) * ld + 16 * kc + 8 * hh) * (int)sizeof(T);   // rows >= N read as zero
      kf[kc] = load_frag_buf<T>(krs, off);
      vf[kc] = load_frag_buf<T>(vrs, off);
    }
  };
  load_kv(kw0);
  auto scale_k = [&]() {   // (called where the fragments are first needed: scaling them forces the wait for their loads)
    if (exact) return;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) kf[kc] = A::scale(kf[kc], c);
  };
  int key = kw0 + r;
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
  int roff = 0;   // tiled build: ring position of the current block's stage 0
  auto slot_of = [&](int st) { return ((st + roff) % 3) * BUF; };
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, k_t0 = 0, k_r0 = 0, t0 = 0, t1 = 0;
  if constexpr (DIAG) {
    k_t0 = stamp();
    k_r0 = __builtin_amdgcn_s_memrealtime();
  }
  // MODE: which scaling's copy of the sweep the loop holds: 0 = folded, 1 = fp32, 2 = both behind a branch (every build but the causal
  // tiled one, whose unit loop carries its state around the sweep: with both copies inside it hipcc spilled 187 registers, so that
  // build takes the branch outside and holds two copies of the whole loop)
  auto units = [&](auto mode_c) {
  constexpr int MODE = decltype(mode_c)::value;
  for (int u = 0;; ++u) {   // units of the causal tiled build (every other build: one pass)
  int nbh = bh, nkbk = kb, nst0 = 0, nroff = 0;   // (causal tiled build: the unit after this one)
  bool nsweep = false;
  const int st0 = CDIAG ? 2 * (kb + 1) : 0;   // first stage of the sweep (causal build: the stage below the diagonal block)
  if (!CDIAG || st0 < nst) {
  if (!(CT && u > 0)) {   // (later units: requested during the previous unit's diagonal block, published by the barrier behind it)
    stage_dma(st0, slot_of(st0));
    dma_wait_all();
  }
  if constexpr (DIAG) { ph[4] = k_t0 - k_t00; ph[5] = stamp() - k_t0; }   // set-up + fragment-load issue; wait for fragments + stage 0
  // The K / V fragments are tracked loads whose first use sits behind `if (active)`: without an unconditional use HERE (where
  // everything has landed anyway) hipcc re-emits their s_waitcnt vmcnt(7..0) inside the stage loop, where they wait out the
  // LDS-DMA of the next stage that the loop has just issued.
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) asm volatile("" ::"v"(kf[kc]), "v"(vf[kc]));
  scale_k();
  if (!(CT && u > 0)) __syncthreads();
  if constexpr (DIAG) { t0 = stamp(); ph[0] += t0 - k_t0; }

  // (two copies of the sweep, one per scaling: the fp32 multiply of the exact one exists in its own instruction stream only)
  auto sweep = [&](auto ex_c) {
  constexpr bool EX = decltype(ex_c)::value != 0;
  if (lay.young_prio && w >= 4) __builtin_amdgcn_s_setprio(1);   // the later-dispatched half loses VALU arbitration otherwise
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
  auto me = [&](f32x16& x, int i) { x[i] = EX ? __builtin_amdgcn_exp2f(x[i] * c) : __builtin_amdgcn_exp2f(x[i]); };
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
  const int heads_here = CDIAG ? 1 : tiles;   // (the causal tiled build loops over its units outside the sweep)
  for (int t = 0; t < heads_here; ++t) {
  if (TILED && t) {   // the fragments of this head were requested before the previous head's stores (see below)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) asm volatile("" ::"v"(kf[kc]), "v"(vf[kc]));
    scale_k();
  }
  const int b0 = slot_of(st0);   // addresses of the current stage
  int cr0 = ra.b[0] + b0, cr1 = ra.b[1] + b0, ct0 = ta.b[0] + b0, ct1 = ta.b[1] + b0, ch16 = 16 * h + b0;
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
  for (int st = st0; st < nst; ++st) {
    const int nb = slot_of(st + 1);
    const int nr0 = ra.b[0] + nb, nr1 = ra.b[1] + nb, nh16 = 16 * h + nb;
    if (st + 1 < nst) stage_dma(st + 1, nb);
    else if (CDIAG) stage_dma(2 * kb, nb);   // the first stage of the diagonal block follows the sweep in the ring
    else if (TILED && t + 1 < heads_here) {   // the next head's sweep: its stage 0 follows in the ring
      const size_t nbase = head_base(lay, bh + 1);
      qraw = make_raw_rsrc(q + nbase, mat_bytes);
      doraw = make_raw_rsrc(dout + nbase, mat_bytes);
      nlraw = make_raw_rsrc(nlv + (size_t)(bh + 1) * N, (uint32_t)N * 4u);
      ndraw = make_raw_rsrc(ndelta + (size_t)(bh + 1) * N, (uint32_t)N * 4u);
      stage_dma(0, nb);
    }
    if (active) {
      period(T1, T1, ic<1>{}, ic<0>{}, ic<2>{}, cr0, cr1, ct0, ct1, cr0, cr1, ch16, sB, dpB, sA, dpA);
      period(T1, T1, ic<2>{}, ic<1>{}, ic<3>{}, cr0, cr1, ct0, ct1, cr0, cr1, ch16, sA, dpA, sB, dpB);
    }
    if constexpr (DIAG) { t1 = stamp(); ph[1] += t1 - t0; }
    dma_wait_all();   // this wave's pieces of the next stage have landed
    if constexpr (DIAG) { t0 = stamp(); ph[2] += t0 - t1; }
    __syncthreads();
    if constexpr (DIAG) { t1 = stamp(); ph[3] += t1 - t0; t0 = t1; }
    if (CDIAG && st + 1 == nst) stage_dma(2 * kb + 1, slot_of(nst + 1));   // second diagonal stage: the slot of stage nst-2 is free now
    if (active) {
      // sub-slice 0 of the next stage is requested from here on (after the last stage: stale data, results unused)
      period(T1, T1, ic<3>{}, ic<2>{}, ic<0>{}, cr0, cr1, ct0, ct1, nr0, nr1, nh16, sB, dpB, sA, dpA);
      period(T1, T1, ic<0>{}, ic<3>{}, ic<1>{}, nr0, nr1, ct0, ct1, nr0, nr1, nh16, sA, dpA, sB, dpB);
    }
    cr0 = nr0; cr1 = nr1; ch16 = nh16;
    ct0 = ta.b[0] + nb; ct1 = ta.b[1] + nb;
  }
  if constexpr (TILED && !CDIAG) {   // hand over to the next head: its fragments are requested before this head's stores are issued
    // (requesting them a stage ahead into spare registers, so that the pipeline never refills, spilled: 256 VGPRs + 128 B)
    float* dkrow = dk + base + (size_t)key * ld;
    float* dvrow = dv + base + (size_t)key * ld;
    if (t + 1 < heads_here) {
      ++bh;
      base = head_base(lay, bh);
      krs = make_rsrc(k + base, mat_bytes);
      vrs = make_rsrc(v + base, mat_bytes);
      load_kv(kw0);
      roff = (roff + nst) % 3;
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 a = {acc_dk[dt][4 * g] * tau, acc_dk[dt][4 * g + 1] * tau, acc_dk[dt][4 * g + 2] * tau, acc_dk[dt][4 * g + 3] * tau};
        f32x4 b = {acc_dv[dt][4 * g], acc_dv[dt][4 * g + 1], acc_dv[dt][4 * g + 2], acc_dv[dt][4 * g + 3]};
        *reinterpret_cast<f32x4*>(dkrow + 32 * dt + 8 * g + 4 * h) = a;
        *reinterpret_cast<f32x4*>(dvrow + 32 * dt + 8 * g + 4 * h) = b;
      }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      acc_dk[dt] = zero16();
      acc_dv[dt] = zero16();
    }
  }
  }   // heads of this workgroup
  };
  if constexpr (MODE == 2) {
    if (exact) sweep(ic<1>{});
    else sweep(ic<0>{});
  } else {
    sweep(ic<MODE>{});
  }
  if constexpr (TILED && !CDIAG) return;
  }
  if constexpr (CDIAG) {
    // The diagonal block: queries kb * 256 .. + 255 = stages 2 * kb, 2 * kb + 1, in the ring slots of stages nst, nst + 1.
    if (st0 >= nst) {   // (the last key block has no stage below its diagonal block)
      stage_dma(2 * kb, slot_of(nst));
      stage_dma(2 * kb + 1, slot_of(nst + 1));
      scale_k();        // (the sweep, which scales the K fragments at its top, did not run)
    }
    dma_wait_all();
    __syncthreads();
    // Causal tiled build: the next unit (the pair's light block, then the next head's heavy one).  The slot of the sweep's last stage
    // is free from here on (every wave is past the sweep): the next unit's first stage goes there while this block is worked off.
    if constexpr (CT) {
      if (u + 1 < nunits) {
        if (u & 1) { nbh = bh + 1; nkbk = pair; }
        else nkbk = nkb - 1 - pair;
        nst0 = 2 * (nkbk + 1);
        nsweep = nst0 < nst;
        const size_t nbase = head_base(lay, nbh);
        qraw = make_raw_rsrc(q + nbase, mat_bytes);      // (the diagonal block below reads LDS only)
        doraw = make_raw_rsrc(dout + nbase, mat_bytes);
        nlraw = make_raw_rsrc(nlv + (size_t)nbh * N, (uint32_t)N * 4u);
        ndraw = make_raw_rsrc(ndelta + (size_t)nbh * N, (uint32_t)N * 4u);
        if (nsweep) {
          const int free_slot = (nst + 2 + roff) % 3;                // = the slot of stage nst - 1
          nroff = ((free_slot - nst0) % 3 + 3) % 3;                  // slot_of(nst0) under the next unit's ring position
          stage_dma(nst0, free_slot * BUF);
        }
      }
    }
    for (int j = w; j < 8; ++j) {
      lds_char* tq = smem + slot_of(nst + (j >> 2));
      lds_char* tdo = tq + TB;
      const int sub = j & 3;
      const int qi0 = kb * BK + 32 * j;
      f32x16 nl16, nd16;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 a = *FA_LDS(f32x4, tq + 2 * TB + 128 * sub + 16 * h + 32 * g);
        const f32x4 b = *FA_LDS(f32x4, tq + 2 * TB + 4 * QS + 128 * sub + 16 * h + 32 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          nl16[4 * g + i] = a[i];
          nd16[4 * g + i] = b[i];
        }
      }
      f32x16 s, dp;
      const bool exactk = A::SPLITS && qi0 < 64;   // wave-uniform: queries with fewer than 64 admissible keys (key block 0 only)
      if (exactk) {   // the unscaled K and the fp32 fma (the rounding of cK is the same for every query and these rows average nothing out)
        frag ku[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) ku[kc] = load_frag_buf<T>(krs, ((kw0 + r) * ld + 16 * kc + 8 * h) * (int)sizeof(T));
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const frag aq = A::template row_frag<D>(tq, ra, 32 * sub, kc);
          const frag ado = A::template row_frag<D>(tdo, ra, 32 * sub, kc);
          if (kc == 0) {
            A::mma_c(s, aq, ku[kc], zero16());
            A::mma_c(dp, ado, vf[kc], nd16);
          } else {
            A::mma(s, aq, ku[kc]);
            A::mma(dp, ado, vf[kc]);
          }
        }
        if (exact) {   // (the staged row constant is -L/tau then)
#pragma unroll
          for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f((s[i] + nl16[i]) * c);
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], c, nl16[i]));
        }
      } else {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const frag aq = A::template row_frag<D>(tq, ra, 32 * sub, kc);
          const frag ado = A::template row_frag<D>(tdo, ra, 32 * sub, kc);
          if (kc == 0) {   // row constants ride in as accumulator inputs: S' = c S - L log2e, dP' = dP - delta
            A::mma_c(s, aq, kf[kc], nl16);
            A::mma_c(dp, ado, vf[kc], nd16);
          } else {
            A::mma(s, aq, kf[kc]);
            A::mma(dp, ado, vf[kc]);
          }
        }
        if (exact) {   // (unscaled K, row constant in raw units)
#pragma unroll
          for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i] * c);
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i]);
        }
      }
      if (j == w) {   // this wave's own 32 queries: key kw0 + r against query qi0 + row
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (r > acc_row(i, h)) s[i] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) dp[i] = s[i] * dp[i];
      const frag pf0 = A::pack(s, 0), pf1 = A::pack(s, 1), ds0 = A::pack(dp, 0), ds1 = A::pack(dp, 1);
      if (!(A::SPLITS && qi0 < 64)) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            A::mma(acc_dv[dt], A::template tr_frag<D>(tdo, ta, 32 * sub + 16 * s2, dt), s2 ? pf1 : pf0);
            A::mma(acc_dk[dt], A::template tr_frag<D>(tq, ta, 32 * sub + 16 * s2, dt), s2 ? ds1 : ds0);
          }
      } else {   // queries with fewer than 64 admissible keys: both products also take what the bf16 rounding of P, dS dropped
        const frag pl0 = A::pack_lo(s, 0, pf0), pl1 = A::pack_lo(s, 1, pf1);
        const frag dl0 = A::pack_lo(dp, 0, ds0), dl1 = A::pack_lo(dp, 1, ds1);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const frag adoT = A::template tr_frag<D>(tdo, ta, 32 * sub + 16 * s2, dt);
            const frag aqT = A::template tr_frag<D>(tq, ta, 32 * sub + 16 * s2, dt);
            A::mma(acc_dv[dt], adoT, s2 ? pf1 : pf0);
            A::mma(acc_dv[dt], adoT, s2 ? pl1 : pl0);
            A::mma(acc_dk[dt], aqT, s2 ? ds1 : ds0);
            A::mma(acc_dk[dt], aqT, s2 ? dl1 : dl0);
          }
      }
    }
  }
  if constexpr (CT) {
    if (u + 1 < nunits) {
      dma_wait_all();     // this wave's pieces of the next unit's first stage have landed
      __syncthreads();    // every wave is done with the diagonal block's stages; the prefetched stage is published
      const int l2 = lane_fresh();
      const int h2 = l2 >> 5;
      float* dkrow = dk + base + (size_t)(kw0 + (l2 & 31)) * ld;
      float* dvrow = dv + base + (size_t)(kw0 + (l2 & 31)) * ld;
      bh = nbh;
      kb = nkbk;
      base = head_base(lay, bh);
      krs = make_rsrc(k + base, mat_bytes);
      vrs = make_rsrc(v + base, mat_bytes);
      kw0 = kb * BK + w * KPW;
      load_kv(kw0);       // requested before the finished unit's stores are issued
      if (nsweep) roff = nroff;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 a = {acc_dk[dt][4 * g] * tau, acc_dk[dt][4 * g + 1] * tau, acc_dk[dt][4 * g + 2] * tau, acc_dk[dt][4 * g + 3] * tau};
          f32x4 b = {acc_dv[dt][4 * g], acc_dv[dt][4 * g + 1], acc_dv[dt][4 * g + 2], acc_dv[dt][4 * g + 3]};
          *reinterpret_cast<f32x4*>(dkrow + 32 * dt + 8 * g + 4 * h2) = a;
          *reinterpret_cast<f32x4*>(dvrow + 32 * dt + 8 * g + 4 * h2) = b;
        }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        acc_dk[dt] = zero16();
        acc_dv[dt] = zero16();
      }
      continue;
    }
  }
  break;
  }   // units
  };
  if constexpr (CT) {
    if (exact) units(ic<1>{});
    else units(ic<0>{});
  } else {
    units(ic<2>{});
  }
  if constexpr (TILED && !CDIAG) return;
  if constexpr (DIAG) ph[1] += stamp() - t0;
  if constexpr (CT) key = kw0 + (lane_fresh() & 31);
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
  if constexpr (DIAG) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left: upper bound of the epilogue
    const unsigned long long k_t2 = stamp(), k_r1 = __builtin_amdgcn_s_memrealtime();
    const int slot = blockIdx.x * 8 + w;
    if (slot < 8192 && lane == 0) {
      for (int j = 0; j < 6; ++j) g_phase_cycles[slot * 8 + j] = ph[j];
      g_phase_cycles[slot * 8 + 6] = k_t2 - k_t00;   // wave lifetime, first instruction to stores drained
      g_phase_cycles[slot * 8 + 7] = k_r1 - k_r0;
    }
  }
}


}  // namespace fa
