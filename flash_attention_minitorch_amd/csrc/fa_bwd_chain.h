// FlashAttention backward for MI355X (gfx950): ONE pass for dQ, dK and dV with the dQ sum CHAINED through the key blocks of a
// workgroup (bf16, d = 64, non-causal, N a multiple of 256).  Part of the kernel set described in fa_kernels.h.
//
// The five products of the reference's single-pass FA-2 backward (src/flash_attn2_bw.cu:94-247: S, dP, dV, dK, dQ) instead of the
// seven that the two-kernel backward (fa_bwd_dkdv.h + fa_bwd_dq.h) executes.  The reference sums dQ over its key blocks with
// atomicAdd (:228).  On this chip float atomics run at one chip-wide rate of about 1.3 TB/s of added bytes
// (MI355X_MICROARCH.md, "Global float atomics"): one add of every dQ element per 256-key block is 2.5 * 256 = 640 flop per
// added byte, i.e. at most 832 TFLOP/s for the whole backward -- less than the two-kernel path delivers.  512 keys per workgroup
// would halve the adds, but dK^T / dV^T accumulators (256 KiB) plus K / V fragments (128 KiB) of 512 keys leave no registers for
// the MFMA-slot pipeline at d = 64, and the K image + dS^T buffers of 512 keys no LDS for the Q / dO ring.  So the sum is cut
// the other way: a workgroup takes C CONSECUTIVE key blocks of one (batch*head), one after the other, and carries the running
// dQ tiles of its chain through memory with PLAIN loads and stores (about 6 TB/s, no ordering protocol: the wave that stored a
// tile is the wave that loads it a key block later); only the last key block of a chain ends in the cross-workgroup sum:
//
//   * nchains == 1 (a whole head per workgroup: enough heads to fill the chip): no atomics at all, the last key block scales the
//     sum by tau and stores it to dq.  Bitwise reproducible, no zero-fill.  (Until the end of round 4 dq ITSELF held the running sums
//     of this form, a lane's 16 bytes at their final place: a 128-byte line of dq is then written by the separate store instructions
//     of two waves and re-read with sc1 loads while the other half is in flight.  One run in about ten of tools/check_chain.py came
//     back with one wrong dq tile at B=32 H=8 N=1024 -- outside the hand-off forms MI355X_MICROARCH.md lists as measured valid, which
//     all write whole lines by one store instruction of one wave -- so this form now keeps its sums in the slab as well.)
//   * running sums (both forms) in a private slab per workgroup
//     (register-major: every store instruction writes 1 KiB contiguous);
//   * nchains > 1 (B*H < CUs; the metric shape: 4 chains of 4 key blocks): the last key block adds tau * sum to dq with
//     no-return fp32 atomics (dq zero-filled by the launcher): N / (256 * C) adds per element instead of N / 256, which at
//     C = 4 is a quarter of the atomic floor (0.21 ms at the metric shape) and hides under the chain's MFMA work.  The
//     dQ tile is formed as dQ[q][d] with d on the lane there (the MFMA's operands swapped), so one accumulator register of a
//     wave is four whole 64-byte row segments: the shape the atomic unit takes at full rate.
//
// Per key block the kernel is bwd_fused_kernel's pass (round 2) with round 3's scaling: the geometry and 16-slot MFMA period of
// bwd_dkdv_slot_kernel (8 waves x 32 keys; K, V fragments and the dK^T, dV^T accumulators in registers; tau*log2e folded into the
// K fragments, the row constant in log2 units: P = exp2(S') is one instruction per score), Q / dO in 64-query stages ("pairs" of
// 32-query sub-slices) by LDS-DMA into a three-slot ring; packed dS^T crosses LDS once (8-byte pieces, XOR-swizzled, conflict
// free for the ds_write_b64 and the transposed reads), and in the NEXT pair every wave forms two 16 x 16 tiles of dQ over all 256
// keys of the block with sixteen v_mfma_f32_16x16x32_bf16 (K from an LDS image of the block's 256 key rows).
// No flags, no polls, no persistent grid: a workgroup never waits for another one.
#pragma once
#include "fa_common.h"

namespace fa {

constexpr int CHAIN_HDR = 4096;   // head of the slab region: a 2-KiB dummy page (stores of tiles that do not exist yet land there)
constexpr int CHAIN_SMEM = 3 * (2 * 8192 + 512) + 32768 + 2 * 32768;   // 148,992 B

// ABL (timing ablations, diagnostic builds only; results are WRONG when non-zero): 1 = no running-tile traffic at all,
// 2 = no dQ MFMAs and operand reads, 4 = no dS^T writes, 64 = no running-tile stores / atomics, 128 = no running-tile loads,
// 256 = no atomics (the plain stores stay), 512 = nt stores, 1024 = nt loads, 2048 = every pair's tiles at ONE 16-KiB slab position
// (the traffic stays in L2: what the memory side costs).
template <typename T, int D, bool ATOMIC, int ABL = 0>
__global__ void __launch_bounds__(512)
bwd_chain_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ dout,
                 const float* __restrict__ nl2, const float* __restrict__ ndelta, float* __restrict__ dq,
                 float* __restrict__ dk, float* __restrict__ dv, float* __restrict__ slab, int N, int nkb, int BH,
                 int nchains, Layout lay, float tau) {
  static_assert(D == 64 && sizeof(T) == 2, "chained schedule is laid out for bf16, d = 64");
  using A = Atom<T>;
  typedef typename A::frag frag;
  constexpr int KC = 4, HS = 64, TB = 8192, STG = 2 * TB + 8 * HS, SUBB = 4096;
  constexpr int KIMG = 3 * STG, DSB = KIMG + 32768;
  static_assert(DSB + 65536 == CHAIN_SMEM, "LDS map");
  __shared__ __attribute__((aligned(16))) char smem_raw[CHAIN_SMEM];
  lds_char* smem = (lds_char*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bh, chain;   // (every chain of a head on one XCD: the head's Q / dO stream is shared through its L2)
  map_block(blockIdx.x, BH, nchains, bh, chain);
  const int C = nkb / nchains;        // key blocks of this chain: chain * C .. chain * C + C - 1
  const int npairs = N / HS;          // = 4 * nkb
  const int ld = lay.ld;
  const uint32_t mat_bytes = ((uint32_t)(N - 1) * ld + D) * (uint32_t)sizeof(T);
  const float c = tau * LOG2E;
  const uint32_t smem_addr = (uint32_t)(uintptr_t)smem;
  const size_t base = head_base(lay, bh);
  const rsrc_t krs = make_rsrc(k + base, mat_bytes);
  const rsrc_t vrs = make_rsrc(v + base, mat_bytes);
  const raw_rsrc_t qraw = make_raw_rsrc(q + base, mat_bytes), doraw = make_raw_rsrc(dout + base, mat_bytes);
  const raw_rsrc_t kraw = make_raw_rsrc(k + base, mat_bytes);
  // waves 0 / 1 move the stage's -L*log2e / -delta rows
  const raw_rsrc_t craw = make_raw_rsrc((w == 0 ? nl2 : ndelta) + (size_t)bh * N, (uint32_t)N * 4u);

  const LaneAddr ra = A::template row_addr<D>(lane);
  const LaneAddr ta = A::template tr_addr<D>(lane);
  // LDS-DMA source offsets (image chunk swizzle applied to the source address; pieces of one wave share a swizzle parity)
  const int dma_row7 = (lane >> 2) & 7;
  const int dma_voff = dma_row7 * ld * (int)sizeof(T) + 16 * (4 * (lane >> 5) + ((lane & 3) ^ ((2 * (w & 1) + (dma_row7 >> 2)) & 3)));
  // dS^T image of one sub-slice: 256 key rows x 32 queries, 64 B per row, 8-B unit uu (4 queries) of row at 64*row + 8*(uu ^ ((row>>1)&7))
  // write addresses of this lane's four 8-byte pieces (key row 32w + r, units 2g + h): 8 * ((2g + h) ^ s) = 8 * (h ^ (s & 1)) +
  // 16 * (g ^ (s >> 1)), and the rest of the address has bits 4-5 clear, so piece g sits at wa0 ^ (16 * g): one register, one v_xor
  const int wa0 = DSB + 64 * (32 * w + r) + 8 * (h ^ ((r >> 1) & 1)) + 16 * ((r >> 2) & 3);
  static_assert((DSB & 0x30) == 0, "dS^T buffers must start on a 64-byte boundary");
  // dQ tiles of this wave: d-block db (16 columns of K), sub-slice ss, query blocks 0 and 1.  v_mfma_f32_16x16x32_bf16: lane
  // (i16, g4) holds K[key = 32 ks + 8 g4 + j][d = 16 db + i16] and dS[q = 16 qb + i16][that key]: two ds_read_b64_tr_b16 each
  // (keys +0..3, +4..7), lane (qq, p4) of a 16-lane group supplying row qq, 8-byte piece p4 of a 4 x 16 block.
  const int g4 = lane >> 4, i16 = lane & 15, qq = i16 >> 2, p4 = i16 & 3, db = w & 3, ss = w >> 2;
  const int kch = 2 * (db & 1) + (p4 >> 1);   // 16-B chunk (mod 4) of K columns 16 db + 4 p4
  const int ka0 = KIMG + 1024 * g4 + 512 * (db >> 1) + 64 * qq + 16 * (kch ^ (2 * (g4 & 1))) + 8 * (p4 & 1);
  const int ka1 = KIMG + 1024 * g4 + 512 * (db >> 1) + 64 * (qq + 4) + 16 * (kch ^ (2 * (g4 & 1) + 1)) + 8 * (p4 & 1);
  int da0[2], da1[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    da0[qb] = DSB + 16384 * ss + 64 * (8 * g4 + qq) + 8 * ((4 * qb + p4) ^ (4 * (g4 & 1) + (qq >> 1)));
    da1[qb] = DSB + 16384 * ss + 64 * (8 * g4 + qq + 4) + 8 * ((4 * qb + p4) ^ (4 * (g4 & 1) + (qq >> 1) + 2));
  }

  auto SB = [&]() { __builtin_amdgcn_sched_barrier(0); };
  auto rowf = [&](int b0, int b1, int tile_off, int sub, int kc) -> frag {
    return *FA_LDS(frag, smem + ((kc & 1) ? b1 : b0) + tile_off + SUBB * sub + 512 * (kc >> 1));
  };
  auto trf = [&](int b0, int b1, int tile_off, int sub, int s2, int dt) -> frag {
    const int kk = tile_off + SUBB * sub + 1024 * (2 * s2) + 512 * dt;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(FA_LDS(bf16x4, smem + b0 + kk));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(FA_LDS(bf16x4, smem + b1 + kk + 1024));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto tr2 = [&](int a0, int a1, int off) -> frag {
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(FA_LDS(bf16x4, smem + a0 + off));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(FA_LDS(bf16x4, smem + a1 + off));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto ld_c = [&](f32x16& x, int hb /* stage base + 16 * h */, int off, int sub) {
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      const f32x4 a = *FA_LDS(f32x4, smem + hb + 2 * TB + off + 128 * sub + 32 * gg);
#pragma unroll
      for (int j = 0; j < 4; ++j) x[4 * gg + j] = a[j];
    }
  };
  auto me = [&](f32x16& x, int i) { x[i] = __builtin_amdgcn_exp2f(x[i]); };
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
  auto ds_put = [&](const frag& f, int g0, int off) {   // registers 8s..8s+3 -> piece g0, 8s+4..8s+7 -> piece g0 + 1
    if constexpr (ABL & 4) return;
    const u32x4 x = __builtin_bit_cast(u32x4, f);
    const u32x2 lo = {x[0], x[1]}, hi = {x[2], x[3]};
    *FA_LDS(u32x2, smem + (wa0 ^ (16 * g0)) + off) = lo;
    *FA_LDS(u32x2, smem + (wa0 ^ (16 * (g0 + 1))) + off) = hi;
  };

  f32x16 acc_dk[2], acc_dv[2], sA, dpA, sB, dpB, cS, cD;
  f32x4 dq0, dq1, pd0, pd1;   // this wave's two dQ tiles of the previous pair; the chain's running sums for them
  frag kf[KC], vf[KC], pf0, pf1, df0, df1, rq[4], rdo[4], tf[4], qa, qb0, qb1;

  // dQ work of one slot: key step ks (32 keys) of the previous pair's tiles: operands requested two slots before their MFMAs
  auto dq_load = [&](int ks, int rd) {
    if constexpr (ABL & 2) return;
    qa = tr2(ka0, ka1, 4096 * ks);
    qb0 = tr2(da0[0], da1[0], rd + 2048 * ks);
    qb1 = tr2(da0[1], da1[1], rd + 2048 * ks);
  };
  auto dq_mma = [&]() {
    if constexpr (ABL & 2) return;
    if constexpr (ATOMIC) {   // dQ[q][d]: register j of lane (i16, g4) = row 4 g4 + j, column i16 (64 contiguous bytes per row)
      dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qb0, qa, dq0, 0, 0, 0);
      SB();
      dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qb1, qa, dq1, 0, 0, 0);
    } else {                  // dQ^T[d][q]: register j = column (d) 4 g4 + j of row (query) i16: 16 contiguous bytes per lane
      dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, qb0, dq0, 0, 0, 0);
      SB();
      dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, qb1, dq1, 0, 0, 0);
    }
    SB();
  };

  // One period (as bwd_dkdv_slot_kernel's) plus: the dS^T pieces of the current sub-slice written to LDS at DSW, and four key
  // steps KS0..KS0+3 of the dQ tiles of the PREVIOUS pair (dS^T buffer at RD; RD < 0: none), requested in slots 1, 5, 9, 13 and
  // issued after the MFMAs of slots 3, 7, 11, 15.
  auto period = [&](auto hn_c, auto hc_c, auto subn_c, auto subc_c, auto subp_c, auto dsw_c, auto rd_c, auto ks0_c, int nr0, int nr1,
                    int ct0, int ct1, int pr0, int pr1, int ph16, f32x16& ns, f32x16& ndp, f32x16& cs, f32x16& cdp, auto&& vm) {
    // vm(ic<slot>): the pair's vector-memory work (running tiles, next stage's LDS-DMA), one piece per slot, behind the slot's MFMA
    constexpr bool HN = decltype(hn_c)::value != 0, HC = decltype(hc_c)::value != 0;
    constexpr int SN = decltype(subn_c)::value, SC = decltype(subc_c)::value, SP = decltype(subp_c)::value;
    constexpr int DSW = decltype(dsw_c)::value;   // byte offset of the dS^T image of sub-slice SC (relative to DSB)
    constexpr int RD = decltype(rd_c)::value, KS0 = decltype(ks0_c)::value;
    constexpr bool DQ = RD >= 0;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {   // slots 0-3: S' chain | exp of scores 0..7 | dO rows 1..3
      if constexpr (HN) {
        if (kq == 0) A::mma_c(ns, rq[0], kf[0], cS);
        else A::mma(ns, rq[kq], kf[kq]);
        SB();
        if (kq < 3) rdo[kq + 1] = rowf(nr0, nr1, TB, SN, kq + 1);
      }
      if constexpr (HC) { me(cs, 2 * kq); me(cs, 2 * kq + 1); }
      if constexpr (DQ) if (kq == 1) dq_load(KS0, RD);
      if (kq == 0) vm(ic<0>{});
      if (kq == 1) vm(ic<1>{});
      if (kq == 2) vm(ic<2>{});
      if (kq == 3) vm(ic<3>{});
      SB();
      if constexpr (DQ) if (kq == 3) dq_mma();
    }
    if constexpr (HN) { A::mma_c(ndp, rdo[0], vf[0], cD); SB(); }   // slot 4
    if constexpr (HC) {
      pf0 = A::pack(cs, 0);
      cdp[0] = cs[0] * cdp[0];
      tf[0] = trf(ct0, ct1, TB, SC, 0, 0);
    }
    vm(ic<4>{});
    SB();
#pragma unroll
    for (int kq = 1; kq < 4; ++kq) {   // slots 5-7
      if constexpr (HN) { A::mma(ndp, rdo[kq], vf[kq]); SB(); }
      if constexpr (HC) {
        me(cs, 6 + 2 * kq); me(cs, 7 + 2 * kq);
        tf[kq] = trf(ct0, ct1, TB, SC, kq >> 1, kq & 1);
      }
      if constexpr (DQ) if (kq == 1) dq_load(KS0 + 1, RD);
      if (kq == 1) vm(ic<5>{});
      if (kq == 2) vm(ic<6>{});
      if (kq == 3) vm(ic<7>{});
      SB();
      if constexpr (DQ) if (kq == 3) dq_mma();
    }
    if constexpr (HC) {
      A::mma(acc_dv[0], tf[0], pf0);   // slot 8
      SB();
      me(cs, 14); me(cs, 15);
      tf[0] = trf(ct0, ct1, 0, SC, 0, 0);
      vm(ic<8>{});
      SB();
      A::mma(acc_dv[1], tf[1], pf0);   // slot 9
      SB();
      pf1 = A::pack(cs, 1);
      cdp[1] = cs[1] * cdp[1];
      tf[1] = trf(ct0, ct1, 0, SC, 0, 1);
      if constexpr (DQ) dq_load(KS0 + 2, RD);
      vm(ic<9>{});
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
      if constexpr (DQ) dq_mma();
      A::mma(acc_dk[0], tf[0], df0);   // slot 12
      SB();
#pragma unroll
      for (int i = 9; i < 15; ++i) cdp[i] = cs[i] * cdp[i];
      ds_put(df0, 0, DSW);
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
      if constexpr (DQ) dq_load(KS0 + 3, RD);
    }
    if constexpr (HN) {
      rq[2] = rowf(pr0, pr1, 0, SP, 2);
      rq[3] = rowf(pr0, pr1, 0, SP, 3);
    }
    SB();
    if constexpr (HC) { A::mma(acc_dk[0], tf[2], df1); SB(); ds_put(df1, 2, DSW); }   // slot 14
    if constexpr (HN) ld_c(cS, ph16, 0, SP);
    SB();
    if constexpr (HC) { A::mma(acc_dk[1], tf[3], df1); SB(); }   // slot 15
    if constexpr (HN) {
      ld_c(cD, ph16, 4 * HS, SP);
      rdo[0] = rowf(pr0, pr1, TB, SP, 0);
    }
    SB();
    if constexpr (DQ && HC) dq_mma();
  };

  auto T1 = ic<1>{};
  auto T0 = ic<0>{};
  auto NO = ic<-1>{};

  // Running tiles of this wave: a private slab, register-major: pair u at slab_w + u * 16 KiB, tile qb at + qb * 1 KiB, a lane's
  // 16 bytes at lane * 16.  The last key block of the non-atomic form stores to dq: a lane's 16 bytes are 4 consecutive d of a row.
  const uint32_t dq_bytes = ((uint32_t)(N - 1) * ld + D) * 4u;
  const rsrc_t dqrs = make_rsrc(dq + base, dq_bytes);
  const uint32_t slab_bytes = (uint32_t)CHAIN_HDR + (uint32_t)N * 256u;
  const rsrc_t srs = make_rsrc(reinterpret_cast<char*>(slab) + (size_t)blockIdx.x * (size_t)N * 256u, slab_bytes);
  const rsrc_t dummy_rs = make_rsrc(slab, (uint32_t)CHAIN_HDR);
  const int t_voff = lane * 16;
  const int t_w = CHAIN_HDR + w * 2048;                        // this wave's tiles of pair 0
  const int t_pair = (ABL & 2048) ? 0 : 16384;                 // bytes from pair to pair
  const int t_d2 = 1024;                                       // ... from query block 0 to 1
  const int f_voff = ((32 * ss + i16) * ld + 16 * db + 4 * g4) * 4;   // the non-atomic form's final store
  // the atomic form's element map: register j of tile qb -> row 32 ss + 16 qb + 4 g4 + j, column 16 db + i16
  const int a_voff = ((32 * ss + 4 * g4) * ld + 16 * db + i16) * 4;

  for (int p = 0; p < C; ++p) {
    const int kb = chain * C + p;
    const bool first = p == 0, last = p == C - 1;
    const int kw0 = kb * 256 + w * 32;
    {   // (lane id recomputed per key block with v_mbcnt: hoisted to kernel entry these offsets are spilled around the sweep)
      const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      const int off0 = ((kw0 + (ln & 31)) * ld + 8 * (ln >> 5)) * (int)sizeof(T);
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        kf[kc] = load_frag_buf<T>(krs, off0 + 32 * kc);
        vf[kc] = load_frag_buf<T>(vrs, off0 + 32 * kc);
      }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      acc_dk[dt] = zero16();
      acc_dv[dt] = zero16();
    }
    auto stage_dma = [&](int j, int dst) {   // 64 queries of Q and dO (wave w: rows 8w..8w+7 of each), waves 0 / 1 the row constants
      const int soff = (HS * j + 8 * w) * ld * (int)sizeof(T);
      dma16(qraw, smem_addr + dst + 1024 * w, dma_voff, soff);
      dma16(doraw, smem_addr + dst + TB + 1024 * w, dma_voff, soff);
      if (w < 2) dma4(craw, smem_addr + dst + 2 * TB + 256 * w, 4 * lane, HS * j * 4);
    };
    // the block's 256 key rows as an LDS image (operand of dQ by transposed reads): wave w moves pieces w, w+8, w+16, w+24
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gp = w + 8 * i;
      dma16(kraw, smem_addr + KIMG + 1024 * gp, dma_voff, (kb * 256 + 8 * gp) * ld * (int)sizeof(T));
    }
    stage_dma(0, 0);
    stage_dma(1, STG);
    dma_wait_all();
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) kf[kc] = A::scale(kf[kc], c);   // tau*log2e rides in the K fragments (re-rounded to bf16)
    __syncthreads();

    int cs_ = 0;                                     // ring slot (byte offset) of the current stage
    int cr0 = ra.b[0], cr1 = ra.b[1], ct0 = ta.b[0], ct1 = ta.b[1], ch16 = 16 * h;
    // operands of sub-slice 0, then its S', dP' alone (the pipeline fills)
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) rq[kc] = rowf(cr0, cr1, 0, 0, kc);
    ld_c(cS, ch16, 0, 0);
    ld_c(cD, ch16, 4 * HS, 0);
    rdo[0] = rowf(cr0, cr1, TB, 0, 0);
    SB();
    auto novm0 = [&](auto) {};
    period(T1, T0, ic<0>{}, ic<0>{}, ic<1>{}, ic<0>{}, NO, ic<0>{}, cr0, cr1, ct0, ct1, cr0, cr1, ch16, sA, dpA, sB, dpB, novm0);

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    dq0 = dq1 = pd0 = pd1 = zero4;
    // ---- running tiles.  The tiles formed DURING local pair t belong to the queries of pair t-1 (whose dS^T was written in pair
    // t-1): their running sums are loaded early in pair t and added at its end (ho_finish); the result is stored (not the last key
    // block) / written or atomically added to dq scaled by tau (the last one) in the FIRST MFMA slots of pair t+1: a vector-memory
    // instruction blocks its wave for a few hundred cycles at issue, which the SIMD's other wave fills inside a period but nobody
    // fills in front of the pair's barrier (stores as the last operations of the pair cost 0.11 ms at the metric shape).
    int ld_soff = 0, st_soff = 0;
    bool ld_on = false, st_on = false;
    const float st_scale = last ? tau : 1.0f;
    auto prepare = [&](int t) {   // pair t loads the sums of tile t-1 and stores tile t-2
      const int u = t - 1, u2 = t - 2;
      ld_on = !first && u >= 0;
      st_on = u2 >= 0;
      ld_soff = t_w + ((ABL & 4096) ? 0 : u * t_pair);   // (4096: loads from one L2-resident spot, stores at full stride)
      st_soff = last ? HS * u2 * ld * 4 : t_w + ((ABL & 8192) ? 0 : u2 * t_pair);   // (8192: the reverse)
    };
    f32x4 s0 = zero4, s1 = zero4;
    auto ho_load = [&](int which) {
      if constexpr (ABL & (1 | 128)) return;
      if (ld_on) {
        const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, t_voff, ld_soff + which * t_d2, (ABL & 1024) ? 2 : 16 /* sc1: past the L1 */));
        if (which) pd1 = x; else pd0 = x;
      }
    };
    auto ho_finish = [&]() {   // dq0 / dq1: this key block's tiles; pd0 / pd1: the chain's running sums
      s0 = (pd0 + dq0) * st_scale;
      s1 = (pd1 + dq1) * st_scale;
      asm volatile("" : "+v"(s0), "+v"(s1)::"memory");   // formed HERE
      dq0 = dq1 = zero4;
      pd0 = pd1 = zero4;
    };
    auto ho_store = [&](int which) {
      if constexpr (ABL & (1 | 64)) return;
      const f32x4 sv = which ? s1 : s0;
      if constexpr (ATOMIC) {
        if (last) {
          if (st_on && !(ABL & 256)) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(sv[j], dqrs, a_voff, st_soff + (16 * which + j) * ld * 4, 0);
          }
          return;
        }
      } else {
        if (last) {
          if (st_on) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sv), dqrs, f_voff, st_soff + which * 16 * ld * 4, 0);
          return;
        }
      }
      // (pairs 0 and 1 have no finished tile yet: their stores land in the dummy page, so every pair runs the same stream)
      const rsrc_t rs = st_on ? srs : dummy_rs;
      const int so = st_on ? st_soff + which * t_d2 : 1024 * which, vo = st_on ? t_voff : lane * 16;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sv), rs, vo, so, (ABL & 512) ? 2 : 0);
    };

    auto pair_body = [&](auto par_c, int t) {
      constexpr int PAR = decltype(par_c)::value;
      constexpr int WR = PAR * 32768, RD = (PAR ^ 1) * 32768;   // dS^T buffer written in this pair / read for the dQ tiles
      __syncthreads();
      const int ns_ = cs_ == 2 * STG ? 0 : cs_ + STG;       // slot of the next stage
      const int n2_ = ns_ == 2 * STG ? 0 : ns_ + STG;       // ... and of the one after (last read in pair t-1)
      const int nr0 = ra.b[0] + ns_, nr1 = ra.b[1] + ns_, nh16 = 16 * h + ns_;
      bool more = false;
      int jn = 0, dsoff = 0;
      auto vmA = [&](auto slot_c) {
        constexpr int S = decltype(slot_c)::value;
        // (the scalar bookkeeping rides behind the first MFMAs: at the top of the pair it would run with both waves of a SIMD idle)
        if constexpr (S == 0) {
          prepare(t);
          more = t + 2 < npairs;
          jn = t + 2;
          dsoff = (HS * jn + 8 * w) * ld * (int)sizeof(T);
        }
        // the finished tiles of the previous pair leave; the running sums of this pair's tiles come from HBM / the memory-side cache
        // (written a key block ago): requested as early as the pair allows
        if constexpr (S == 0) ho_store(0);
        if constexpr (S == 1) { ho_store(1); ho_load(0); }
        if constexpr (S == 3) ho_load(1);
        // next-but-one stage by LDS-DMA, issued by waves 0-3 only (rows 8w..8w+7 and 8w+32..8w+39 of Q and of dO): the older half
        // of the workgroup wins the issue arbitration and idles at the pair's barrier
        if constexpr (S == 4) { if (more && w < 4) dma16(qraw, smem_addr + n2_ + 1024 * w, dma_voff, dsoff); }
        if constexpr (S == 5) { if (more && w < 4) dma16(doraw, smem_addr + n2_ + TB + 1024 * w, dma_voff, dsoff); }
        if constexpr (S == 6) { if (more && w < 4) dma16(qraw, smem_addr + n2_ + 1024 * (w + 4), dma_voff, dsoff + 32 * ld * (int)sizeof(T)); }
        if constexpr (S == 7) { if (more && w < 4) dma16(doraw, smem_addr + n2_ + TB + 1024 * (w + 4), dma_voff, dsoff + 32 * ld * (int)sizeof(T)); }
        if constexpr (S == 2) { if (more && w < 2) dma4(craw, smem_addr + n2_ + 2 * TB + 256 * w, 4 * lane, HS * jn * 4); }
      };
      auto vmB = [&](auto) {};
      // (in pair 0 the dQ products run on whatever the dS^T buffer holds; the pair's finish sends them to the dummy page)
      period(T1, T1, ic<1>{}, ic<0>{}, ic<0>{}, ic<WR>{}, ic<RD>{}, ic<0>{}, cr0, cr1, ct0, ct1, nr0, nr1, nh16, sB, dpB, sA, dpA, vmA);
      period(T1, T1, ic<0>{}, ic<1>{}, ic<1>{}, ic<WR + 16384>{}, ic<RD>{}, ic<4>{}, nr0, nr1, ct0, ct1, nr0, nr1, nh16, sA, dpA, sB, dpB, vmB);
      // the pair's tiles are complete: add the chain's sums (the compiler waits for their loads here); this wave's LDS-DMA pieces of
      // stage t+2 have landed
      ho_finish();
      dma_wait_all();
      cs_ = ns_;
      cr0 = nr0; cr1 = nr1; ch16 = nh16;
      ct0 = ta.b[0] + ns_; ct1 = ta.b[1] + ns_;
    };
    for (int t = 0; t < npairs; t += 2) {
      pair_body(T0, t);
      pair_body(T1, t + 1);
    }
    // drain: the last pair's dQ tiles (its dS^T sits in buffer 1: npairs is even)
    __syncthreads();
    prepare(npairs);
    ho_store(0);
    ho_store(1);
    ho_load(0);
    ho_load(1);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      dq_load(ks, 32768);
      SB();
      dq_mma();
    }
    ho_finish();
    prepare(npairs + 1);
    ho_store(0);
    ho_store(1);

    // (row pointers formed here from opaque copies: formed before the sweep they cost five registers across it, spilled)
    int key = kw0 + r, hh = h;
    asm volatile("" : "+v"(key), "+v"(hh));
    float* dkrow = dk + base + (size_t)key * ld;
    float* dvrow = dv + base + (size_t)key * ld;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        f32x4 a = {acc_dk[dt][4 * gg] * tau, acc_dk[dt][4 * gg + 1] * tau, acc_dk[dt][4 * gg + 2] * tau, acc_dk[dt][4 * gg + 3] * tau};
        f32x4 b = {acc_dv[dt][4 * gg], acc_dv[dt][4 * gg + 1], acc_dv[dt][4 * gg + 2], acc_dv[dt][4 * gg + 3]};
        *reinterpret_cast<f32x4*>(dkrow + 32 * dt + 8 * gg + 4 * hh) = a;
        *reinterpret_cast<f32x4*>(dvrow + 32 * dt + 8 * gg + 4 * hh) = b;
      }
    // the next key block overwrites the K image, the ring and the running tiles this one has just stored
    dma_wait_all();
    __syncthreads();
  }
}

}  // namespace fa
