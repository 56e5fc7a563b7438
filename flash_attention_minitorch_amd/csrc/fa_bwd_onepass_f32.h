// FlashAttention backward for MI355X (gfx950), fp32 (the reference's own dtype), d = 64, N >= 256: ONE pass for
// dQ, dK and dV -- the five products of the reference's single-pass FA-2 backward (src/flash_attn2_bw.cu:94-247: S, dP, dV, dK, dQ),
// dQ summed over the key blocks with fp32 atomics as the reference does (:228).  Part of the kernel set described in fa_kernels.h.
//
// Why it pays HERE and not for bf16 (fa_bwd_chain.h, profiles/r04_chain_backward.txt): the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32 /
// 16x16x4_f32) runs at 1/16 of the bf16 rate, so the kernel is MFMA-bound by a wide margin and the two-kernel backward's seven
// products for five cost their full 40 %; the same dQ adds (N/256 per element) that exceed the chip's ~1.3 TB/s atomic rate beside
// bf16 MFMAs need a third of it beside fp32 ones (0.45 TB/s at any N: bytes and time both grow with N^2), spread evenly over the
// kernel (every 32-query stage adds its tile), and hide.
//
// Geometry: a workgroup = 8 waves x 32 keys = one 256-key block of one (batch*head); K, V fragments and the dK^T, dV^T accumulators
// of a wave's keys in registers (as bwd_dkdv_kernel).  It sweeps 32-query stages (Q, dO tiles and the rows' -L/tau, -delta, register
// staged into a double buffer).  Per stage every wave forms S' = Q K^T - L/tau and dP' = dO V^T - delta (row constants as MFMA
// accumulator inputs), P = exp2(c S'), dS = P o dP', dV^T += dO^T P, dK^T += Q^T dS from registers, and writes its 32 x 32 block of dS
// to an LDS image [query][key]; after a barrier wave w forms ONE 16 x 16 tile of dQ = dS K over all 256 keys (query block w >> 2, column
// block w & 3) with 64 v_mfma_f32_16x16x4_f32 -- dS rows and K columns (an LDS image of the block's 256 key rows) as one scalar LDS
// read per lane and MFMA -- and adds tau * tile to dq: four no-return atomics per wave and stage, each register four whole 64-byte row
// segments.  CAUSAL (the reference's causal kernels skip and mask the same way: src/flash_attn_causal_bw.cu, the j > i tiles): key
// block kb starts its sweep at its own queries; on the 8 diagonal stages the waves whose keys are all masked sit out, the wave on the
// diagonal masks above it, and dQ sums the live keys only; blocks are dispatched longest sweep first (map_block_ranked).  RAGGED: N not
// a multiple of 256 (its own builds).  The geometry keeps whole CUs busy only when the launch has about a workgroup per CU: the launcher
// (fa_api.hip: onepass_f32) sends smaller launches to the two kernels.  Measured (profiles/r04_onepass_f32.txt, r04_f32_pmc.json):
// 135 vs 98 TFLOP/s non-causal and 119 vs 87 causal at B=8 H=8 N=2048 against the two-kernel path, 138 at N = 4096 (88.7 % MFMA-busy
// at 2.39 GHz: fp32 MFMAs stay under the power limit; HBM traffic = the algorithmic bytes + the atomic adds, 0.22 TB/s).
// dq must be zero on entry (the launcher fills it; the reference's caller does the same, minitorch/cuda_kernel_ops.py:609-611).
#pragma once
#include "fa_common.h"

namespace fa {

constexpr int OP32_QS = 32, OP32_BK = 256;
constexpr int OP32_TB = (OP32_QS * (64 + 4)) * 4;                 // one 32 x 64 fp32 tile, rows padded by 16 B
constexpr int OP32_STG = 2 * OP32_TB + 8 * OP32_QS;               // Q tile, dO tile, 32 x (-L/tau), 32 x (-delta)
constexpr int OP32_KIMG = (OP32_BK * (64 + 4)) * 4;               // the block's 256 key rows
constexpr int OP32_DSROW = OP32_BK + 4;                           // floats per row of the dS image (pad: a lane pair's rows 4 apart)
constexpr int OP32_DS = OP32_QS * OP32_DSROW * 4;
constexpr int OP32_SMEM = 2 * OP32_STG + OP32_KIMG + OP32_DS;     // 138,240 B: one workgroup per CU, two waves per SIMD

template <int D, bool CAUSAL, bool RAGGED>
__global__ void __launch_bounds__(512, 2)
bwd_onepass_f32_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                       const float* __restrict__ dout, const float* __restrict__ nlc, const float* __restrict__ ndelta,
                       float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv, int N, int nkb, int BH, Layout lay,
                       float tau, int nsplit) {
  if (guard_skip(lay)) return;
  static_assert(D == 64, "laid out for d = 64");
  using A = Atom<float>;
  typedef typename A::frag frag;
  constexpr int KC = D / 16, DT = D / 32, QS = OP32_QS, TB = OP32_TB, STG = OP32_STG, KIMG = 2 * OP32_STG, DSB = KIMG + OP32_KIMG;
  __shared__ __attribute__((aligned(16))) char smem_raw[OP32_SMEM];
  lds_char* smem = (lds_char*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // nsplit > 1 (launches that would leave CUs idle): the query sweep of a key block is cut into nsplit consecutive parts, one workgroup
  // each; dK and dV are then sums over the parts as well: atomic adds into zero-filled dk, dv (two parts: still bitwise repeatable)
  const int sp = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % (unsigned)nsplit)), idb = (int)(blockIdx.x / (unsigned)nsplit);
  int bh, kb;
  if (CAUSAL) map_block_ranked(idb, BH, nkb, max(lay.rank_chunk, 1), bh, kb);   // key block 0 sweeps the most stages: first
  else map_block(idb, BH, nkb, bh, kb);
  const size_t base = head_base(lay, bh);
  const int ld = lay.ld;
  const uint32_t mat_bytes = ((uint32_t)(N - 1) * ld + D) * 4u;
  const rsrc_t qrs = make_rsrc(q + base, mat_bytes), dors = make_rsrc(dout + base, mat_bytes);
  const rsrc_t krs = make_rsrc(k + base, mat_bytes), vrs = make_rsrc(v + base, mat_bytes);
  const rsrc_t dqrs = make_rsrc(dq + base, mat_bytes);
  const float* nlg = nlc + (size_t)bh * N;
  const float* deg = ndelta + (size_t)bh * N;
  const float c = tau * LOG2E;
  const int kb0 = kb * OP32_BK, kw0 = kb0 + 32 * w;

  frag vf[KC];   // (the wave's K fragments are re-read from the key image every stage: 32 registers for 8 LDS reads)
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) vf[kc] = load_frag_buf<float>(vrs, ((kw0 + r) * ld + 16 * kc + 8 * h) * 4);
  {   // the block's key rows as an LDS image (the B operand of dQ = dS K by column reads)
    TileStager<float, D, OP32_BK, 512> sk;
    sk.init(tid, ld);
    sk.load(krs, kb0);
    if (RAGGED && kb0 + OP32_BK > N) {   // ragged N: key rows past N as exact zeros (dS is zero there; 0 * whatever lies behind the tensor must be too)
#pragma unroll
      for (int i = 0; i < sk.PER; ++i)
        if (kb0 + tid / sk.CPR + i * sk.RSTEP >= N) sk.regs[i] = u32x4{0u, 0u, 0u, 0u};
    }
    sk.store(smem + KIMG);
  }
  f32x16 acc_dk[DT], acc_dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    acc_dk[dt] = zero16();
    acc_dv[dt] = zero16();
  }
  const LaneAddr ra = A::template row_addr<D>(lane);
  const LaneAddr ta = A::template tr_addr<D>(lane);
  TileStager<float, D, QS, 512> sq, sdo;
  sq.init(tid, ld);
  sdo.init(tid, ld);
  float st_nl = 0.f, st_de = 0.f;
  auto stage_load = [&](int qi) {
    sq.load(qrs, qi * QS);
    sdo.load(dors, qi * QS);
    if (RAGGED && (qi + 1) * QS > N && qi * QS + tid / sq.CPR >= N) sq.regs[0] = sdo.regs[0] = u32x4{0u, 0u, 0u, 0u};   // (one chunk per thread)
    if (tid < QS) {   // (rows past a ragged N: P = exp2(c * (0 - 1e30)) = 0, so nothing of them reaches dK, dV)
      const bool in = !RAGGED || qi * QS + tid < N;
      st_nl = in ? nlg[qi * QS + tid] : -1e30f;
      st_de = in ? deg[qi * QS + tid] : 0.0f;
    }
  };
  auto stage_store = [&](lds_char* b) {
    sq.store(b);
    sdo.store(b + TB);
    if (tid < QS) {
      *FA_LDS(float, b + 2 * TB + 4 * tid) = st_nl;
      *FA_LDS(float, b + 2 * TB + 4 * QS + 4 * tid) = st_de;
    }
  };
  // dQ tile of this wave: query block qb (16 rows), column block db (16 columns).  v_mfma_f32_16x16x4_f32: lane (i16, g4) supplies
  // A[i16][g4] = dS[16 qb + i16][key k0 + g4] and B[g4][i16] = K[key k0 + g4][16 db + i16]; its four result registers are rows
  // 4 g4 + j of column i16.
  const int i16 = lane & 15, g4 = lane >> 4, qb = w >> 2, db = w & 3;
  // Bank conflicts: a K row is 68 words, so rows 4 apart sit 16 banks apart: the four keys of one MFMA are k, k + 4, k + 8, k + 12 (the B
  // read of a lane: key k + 4 g4, 16 consecutive columns: 64 distinct banks), and the dS image stores every 16 keys with their 4 x 4
  // index transposed (column 4 (key & 3) + ((key >> 2) & 3)), so that the A read stays four consecutive words per row (260-word rows:
  // 64 distinct banks).  With keys k .. k + 3 per MFMA the B reads were 4-way conflicts (SQ_LDS_BANK_CONFLICT 67 M cycles per launch).
  const int ds_rd = DSB + ((16 * qb + i16) * OP32_DSROW + g4) * 4;
  const int k_rd = KIMG + (4 * g4 * (D + 4) + 16 * db + i16) * 4;
  const int ds_wr = DSB + (32 * w + (r & 16) + 4 * (r & 3) + ((r >> 2) & 3)) * 4;   // this lane's key column of the dS image; row acc_row(i, h)
  const int dq_voff = ((16 * qb + 4 * g4) * ld + 16 * db + i16) * 4;

  // CAUSAL: the sweep starts at the block's own queries; in diagonal stage j = qi - 8 kb < 8 (queries 32 j .. 32 j + 31 of the block)
  // waves w > j see masked keys only and sit the stage out, wave j masks above its diagonal, and dQ sums the 32 (j + 1) live keys
  // RAGGED (N not a multiple of 256; its own build: the checks cost the aligned one 4-6 %): rows past N are staged as zeros (Q, dO, K; V by its resource's range check); the last key block clears P at its keys >= N,
  // the last stage skips the adds of its rows >= N, the epilogue the stores of its keys >= N.
  const int qi0 = CAUSAL ? kb * (OP32_BK / QS) : 0;   // (the diagonal stages are counted from here whatever part this workgroup sweeps)
  const int nq_all = (N + QS - 1) / QS, per = (nq_all - qi0 + nsplit - 1) / nsplit;
  const int q_begin = qi0 + sp * per, nqi = min(nq_all, q_begin + per);
  if (q_begin >= nqi) return;   // (more parts than stages: nothing to add; before any barrier)
  const bool ragged_keys = RAGGED && kw0 + 32 > N;   // (wave-uniform)
  stage_load(q_begin);
  stage_store(smem);
  __syncthreads();

  auto slice = [&](auto par, int qi) {
    constexpr int PAR = decltype(par)::value;
    const bool more = qi + 1 < nqi;
    if (more) stage_load(qi + 1);
    lds_char* tq = smem + PAR * STG;
    lds_char* tdo = tq + TB;
    const int j = CAUSAL ? qi - qi0 : 8;
    if (!CAUSAL || w <= j) {
      f32x16 nl16, nd16, s, dp;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 a = *FA_LDS(f32x4, tq + 2 * TB + 16 * h + 32 * g);
        const f32x4 b = *FA_LDS(f32x4, tq + 2 * TB + 4 * QS + 16 * h + 32 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          nl16[4 * g + i] = a[i];
          nd16[4 * g + i] = b[i];
        }
      }
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const frag aq = A::template row_frag<D>(tq, ra, 0, kc);
        const frag ado = A::template row_frag<D>(tdo, ra, 0, kc);
        const frag kfr = A::template row_frag<D>(smem + KIMG, ra, 32 * w, kc);
        if (kc == 0) {   // row constants ride in as accumulator inputs: S' = S - L/tau, dP' = dP - delta
          A::mma_c(s, aq, kfr, nl16);
          A::mma_c(dp, ado, vf[kc], nd16);
        } else {
          A::mma(s, aq, kfr);
          A::mma(dp, ado, vf[kc]);
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_exp2f(s[i] * c);
      if (CAUSAL && w == j) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = r <= acc_row(i, h) ? s[i] : 0.0f;
      }
      if (ragged_keys) {
        const bool keep = kw0 + r < N;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = keep ? s[i] : 0.0f;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) dp[i] = s[i] * dp[i];
#pragma unroll
      for (int i = 0; i < 16; ++i) *FA_LDS(float, smem + ds_wr + acc_row(i, h) * OP32_DSROW * 4) = dp[i];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          A::mma(acc_dv[dt], A::template tr_frag<D>(tdo, ta, 16 * s2, dt), A::pack(s, s2));
          A::mma(acc_dk[dt], A::template tr_frag<D>(tq, ta, 16 * s2, dt), A::pack(dp, s2));
        }
    }
    __syncthreads();   // the stage's dS image is complete
    f32x4 t4 = {0.f, 0.f, 0.f, 0.f};
    {
      // dQ tile: one MFMA per pinned slot, the operands of the step LA = 8 ahead requested in its shadow (left alone, hipcc requests all
      // 128 operands of the stage at once and spills; a rolled loop without look-ahead loses 8 %).  Step st of a 32-key group: keys
      // 16 (st >> 2) + (st & 3) + 4 g4 = image columns 4 st + g4.
      auto SB = [&]() { __builtin_amdgcn_sched_barrier(0); };
      // CAUSAL: only the 32-key groups at or below the stage's diagonal (a scalar branch per group of 8 steps)
      const int ng = CAUSAL ? min(8, j + 1) : 8;
      constexpr int LA = 8;
      float ra_[LA], rb_[LA];
      auto ld1 = [&](int st) {   // (two bases: the K image's 128-row offset does not fit the instruction's 16-bit immediate)
        const int half = st >> 5, s5 = st & 31;
        ra_[st % LA] = *FA_LDS(float, smem + ds_rd + half * 128 * 4 + 16 * s5);
        rb_[st % LA] = *FA_LDS(float, smem + k_rd + (half * 128 + 16 * (s5 >> 2) + (s5 & 3)) * (D + 4) * 4);
      };
#pragma unroll
      for (int st = 0; st < LA; ++st) ld1(st);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (CAUSAL && g >= ng) break;
        const bool more = !CAUSAL || g + 1 < ng;
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
          const int st = 8 * g + s8;
          SB();
          t4 = __builtin_amdgcn_mfma_f32_16x16x4f32(ra_[st % LA], rb_[st % LA], t4, 0, 0, 0);
          if (st + LA < 64 && more) ld1(st + LA);
        }
      }
      SB();
    }
    const int soff = qi * QS * ld * 4;
    if (!RAGGED || (qi + 1) * QS <= N) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(t4[jj] * tau, dqrs, dq_voff, soff + jj * ld * 4, 0);
    } else {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        if (qi * QS + 16 * qb + 4 * g4 + jj < N)
          __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(t4[jj] * tau, dqrs, dq_voff, soff + jj * ld * 4, 0);
    }
    if (more) stage_store(smem + (PAR ^ 1) * STG);
    __syncthreads();   // the next stage is published; every wave is done with this stage's dS image
  };
  int qi = q_begin;
  for (; qi + 1 < nqi; qi += 2) {
    slice(ic<0>{}, qi);
    slice(ic<1>{}, qi + 1);
  }
  if (qi < nqi) slice(ic<0>{}, qi);

  // (the lane's row and half re-derived from v_mbcnt: kept from kernel entry they are two registers across the sweep, spilled)
  const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int er = ln & 31, eh = ln >> 5;
  if (nsplit > 1) {
    // dK, dV of this part are ADDED (dk, dv zero-filled by the launcher).  The accumulators hold a key per lane (rows 256 B apart): every
    // wave transposes its two 32 x 64 tiles through LDS (all images are free after the sweep's last barrier; 65-word rows: conflict-free
    // both ways) so that one atomic instruction adds one whole 256-byte row (lane = column) instead of 64 scattered words.
    lds_char* tw = smem + w * (2 * 32 * 65 * 4);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int o = (er * 65 + 32 * dt + acc_row(i, eh)) * 4;
        *FA_LDS(float, tw + o) = acc_dk[dt][i] * tau;
        *FA_LDS(float, tw + 32 * 65 * 4 + o) = acc_dv[dt][i];
      }
    const rsrc_t dkrs = make_rsrc(dk + base, mat_bytes), dvrs = make_rsrc(dv + base, mat_bytes);
#pragma unroll 8
    for (int kk = 0; kk < 32; ++kk) {
      if (RAGGED && kw0 + kk >= N) break;
      const float x = *FA_LDS(float, tw + (kk * 65 + ln) * 4);
      const float y = *FA_LDS(float, tw + 32 * 65 * 4 + (kk * 65 + ln) * 4);
      __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(x, dkrs, ln * 4, (kw0 + kk) * ld * 4, 0);
      __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(y, dvrs, ln * 4, (kw0 + kk) * ld * 4, 0);
    }
    return;
  }
  if (RAGGED && kw0 + er >= N) return;
  float* dkrow = dk + base + (size_t)(kw0 + er) * ld;
  float* dvrow = dv + base + (size_t)(kw0 + er) * ld;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a = {acc_dk[dt][4 * g] * tau, acc_dk[dt][4 * g + 1] * tau, acc_dk[dt][4 * g + 2] * tau, acc_dk[dt][4 * g + 3] * tau};
      f32x4 b = {acc_dv[dt][4 * g], acc_dv[dt][4 * g + 1], acc_dv[dt][4 * g + 2], acc_dv[dt][4 * g + 3]};
      *reinterpret_cast<f32x4*>(dkrow + 32 * dt + 8 * g + 4 * eh) = a;
      *reinterpret_cast<f32x4*>(dvrow + 32 * dt + 8 * g + 4 * eh) = b;
    }
}

}  // namespace fa
