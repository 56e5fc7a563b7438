// FlashAttention backward for MI355X (gfx950): ONE pass for dQ, dK and dV (bf16, d = 64, non-causal, N a multiple of 256).
// Part of the kernel set described in fa_kernels.h.
//
// The five products of the reference's single-pass FA-2 backward (src/flash_attn2_bw.cu:94-247: S, dP, dV, dK, dQ) with no
// atomics (the reference adds dQ with atomicAdd, :228) and no recomputation (the two-kernel backward of fa_bwd_dkdv.h +
// fa_bwd_dq.h executes 7 products for these 5):
//
//   * geometry and MFMA-slot period of bwd_dkdv_slot_kernel: a workgroup = 8 waves x 32 keys of one (batch*head); K, V fragments
//     and the dK^T, dV^T accumulators live in registers; the workgroup sweeps 32-query sub-slices, S' / dP' of sub-slice i+1
//     issued beside the exp / mul / pack and the dV^T, dK^T products of sub-slice i.  Q / dO stages are 64 queries (a "pair" of
//     sub-slices), moved by LDS-DMA into a three-slot ring, one barrier per pair.
//   * dS crosses LDS once: every wave writes its packed dS^T (key on the row, 8-byte pieces of 4 queries, XOR-swizzled so that
//     both the ds_write_b64 and the transposed reads are bank-conflict free) into a per-pair buffer; in the NEXT pair every
//     wave forms two 16(d) x 16(query) tiles of dQ^T = K^T dS^T over all 256 keys of the workgroup with sixteen
//     v_mfma_f32_16x16x32_bf16 (the wave's d-block is w & 3, its sub-slice w >> 2, the two tiles its two query blocks: the K
//     operand is shared), both operands by transposed LDS reads (K from an image of the workgroup's 256 key rows that stays
//     in LDS).  Every wave runs the same stream: two small MFMAs ride in every fourth slot of a period.
//   * dQ is summed ACROSS the nkb = N/256 key-block workgroups of a head by an ordered hand-off in a fixed order per query
//     pair (bitwise reproducible): workgroup kb starts its sweep at query pair 4*kb and wraps around, so for every pair the
//     nkb workgroups arrive 4 pairs apart: position p = (local pair index) >> 2.  Position 0 stores its own tiles, position p > 0
//     adds position p-1's running fp32 tiles to its own (loaded a pair before they are needed), the last position scales by
//     tau and writes dq.  Running tiles live in a per-group slab of N x 64 floats (register-major: every store instruction writes
//     1 KiB contiguous), written with write-through (sc1) 16-byte stores, drained by every storing wave (s_waitcnt vmcnt(0))
//     before the pair's barrier, then ONE lane stores the pair's flag (sc1); the consumer wave polls that flag with an sc1 load
//     and reads the tile with sc1 16-byte loads (cdna_hip_programming.md Guideline 16, recipe R1 with sc1 loads in place of
//     the acquire: MI355X_MICROARCH.md "Valid forms", first table row).  The load is issued one period before its first use and
//     the flag it depends on was stored two pairs earlier when the workgroups run in step, so the chain is a pipeline.
//   * persistent grid: nkb workgroups (one per CU: 146 KiB of LDS) form a group that walks heads g, g + ngroups, ...; every
//     member of a chain is therefore resident for the whole launch (the launcher sizes the grid to the CU count), flags count
//     up across the heads of a group (value = head iteration * nkb + position) so a slab is never overwritten before its last
//     reader is done, spins are bounded (error word in the workspace, fa_mi355x_bwd_status).
#pragma once
#include "fa_common.h"

namespace fa {

typedef __attribute__((address_space(1))) unsigned gu32;

FA_DEV unsigned flag_load(unsigned* p) {
  return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_load_dword ... sc1
}
// The same load issued as inline asm: hipcc does not track its result, so no compiler-placed s_waitcnt follows it (a tracked
// load whose value is consumed behind a branchy region makes hipcc wait vmcnt(0) there, i.e. for the stores just issued).
// The ISSUER waits: the value is valid after the wave's next "s_waitcnt vmcnt(0)" (dma_wait_all at the end of the pair).
FA_DEV unsigned flag_load_untracked(raw_rsrc_t rs, int soff) {
  unsigned x;
  asm volatile("buffer_load_dword %0, off, %1, %2 sc1" : "=v"(x) : "s"(rs), "s"(soff) : "memory");
  return x;
}
// ... and the poll of the (rare) spin: load and wait in ONE statement, so the value is valid when it ends and hipcc has no pending
// load to account for at the loop's exit (a tracked load there cost an s_waitcnt vmcnt(0) on the no-spin path as well)
FA_DEV unsigned flag_load_waited(raw_rsrc_t rs, int soff) {
  unsigned x;
  asm volatile("buffer_load_dword %0, off, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "s"(rs), "s"(soff) : "memory");
  return x;
}
FA_DEV void flag_store(unsigned* p, unsigned x) {
  __hip_atomic_store((gu32*)p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_store_dword ... sc1
}

// Hand-off region of the workspace (one buffer resource over all of it, byte offsets):
//   [0, 256)          control words: 0 = error, 8 = dummy flag (stores for tiles that do not exist yet land there)
//   [256, 256 + 16 K) flags, one per (group, query pair)
//   then              a zero page (2 KiB, never written in the launch) and a dummy page (2 KiB, never read)
//   then              the running-tile slabs, N * 256 B per group
constexpr int FUSED_FLAG0 = 256, FUSED_FLAG_BYTES = 16384, FUSED_PAGE0 = FUSED_FLAG0 + FUSED_FLAG_BYTES, FUSED_PAGES = 4096;
constexpr int FUSED_SLAB0 = FUSED_PAGE0 + FUSED_PAGES;
constexpr int FUSED_SMEM = 3 * (2 * 8192 + 512) + 32768 + 2 * 32768;   // 148,992 B

// ABL (timing ablations, diagnostic builds only; results are WRONG when non-zero): 1 = no hand-off memory traffic (no flag, no
// running-tile load / store), 2 = no dQ^T MFMAs and operand reads, 4 = no dS^T writes, 8 = no early poll, 64 = no running-tile stores, 128 = no running-tile loads, 256 = no flags (no signal, poll or wait),
// 32 = phase stamps into g_phase_cycles (read the shares, never the run time).
template <typename T, int D, int ABL = 0>
__global__ void __launch_bounds__(512)
bwd_fused_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ dout,
                 const float* __restrict__ nlc, const float* __restrict__ ndelta, float* __restrict__ dq,
                 float* __restrict__ dk, float* __restrict__ dv, unsigned* hand, int N,
                 int nkb, int BH, int ngroups, int xcdmap, Layout lay, float tau) {
  static_assert(D == 64 && sizeof(T) == 2, "fused schedule is laid out for bf16, d = 64");
  using A = Atom<T>;
  typedef typename A::frag frag;
  constexpr int KC = 4, HS = 64, TB = 8192, STG = 2 * TB + 8 * HS, SUBB = 4096;
  constexpr int KIMG = 3 * STG, DSB = KIMG + 32768;
  static_assert(DSB + 65536 == FUSED_SMEM, "LDS map");
  __shared__ __attribute__((aligned(16))) char smem_raw[FUSED_SMEM];
  lds_char* smem = (lds_char*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (group, key block): members of a group share an XCD under round-robin dispatch (speed only)
  int g, kb;
  if (xcdmap) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    g = (slot / nkb) * 8 + xcd;
    kb = slot % nkb;
  } else {
    g = blockIdx.x / nkb;
    kb = blockIdx.x % nkb;
  }
  if (g >= ngroups) return;
  const int npairs = N / HS;          // = 4 * nkb
  const int j0 = 4 * kb;              // first query pair of this workgroup's sweep
  const int kw0 = kb * 256 + w * 32;
  const int ld = lay.ld;
  const uint32_t mat_bytes = ((uint32_t)(N - 1) * ld + D) * (uint32_t)sizeof(T);
  const float c = tau * LOG2E;
  const uint32_t smem_addr = (uint32_t)(uintptr_t)smem;
  const int gflag0 = FUSED_FLAG0 + g * npairs * 4;   // byte offset of this group's flags
  // running dQ tiles, register-major: [zero page][dummy page] then per group a slab: pair j, wave w, query block qb at
  // ((j * 8 + w) * 2 + qb) * 1024 B.  One resource over all of it (the launcher keeps it under 2 GiB)
  const uint32_t hand_bytes = (uint32_t)FUSED_SLAB0 + (uint32_t)ngroups * (uint32_t)N * 256u;
  const rsrc_t prs = make_rsrc(hand, hand_bytes);
  const raw_rsrc_t praw = make_raw_rsrc(hand, hand_bytes);
  const raw_rsrc_t craw = make_raw_rsrc(nlc, 0xfffffffcu);   // both row-constant vectors (the launcher keeps 2 * rows * 4 < 4 GiB)

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
  // dQ^T tiles of this wave: d-block db (16 columns of K), sub-slice ss, query blocks 0 and 1.  v_mfma_f32_16x16x32_bf16 lane
  // (i16, g4) holds A[d = 16 db + i16][key = 32 ks + 8 g4 + j] and B[that key][q = 16 qb + i16]: two ds_read_b64_tr_b16 each
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
  auto me = [&](f32x16& x, int i) { x[i] = __builtin_amdgcn_exp2f(x[i] * c); };
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
  auto ds_put = [&](const frag& f, int g0, int off) {   // registers 8s..8s+3 -> piece g0, 8s+4..8s+7 -> piece g0 + 1
    if constexpr (ABL & 4) return;
    const u32x4 x = __builtin_bit_cast(u32x4, f);
    const u32x2 lo = {x[0], x[1]}, hi = {x[2], x[3]};
    *FA_LDS(u32x2, smem + (wa0 ^ (16 * g0)) + off) = lo;
    *FA_LDS(u32x2, smem + (wa0 ^ (16 * (g0 + 1))) + off) = hi;
  };

  f32x16 acc_dk[2], acc_dv[2], sA, dpA, sB, dpB, cS, cD;
  f32x4 dq0, dq1, pd0, pd1;   // this wave's two dQ^T tiles of the previous pair; the previous chain position's running tiles
  frag kf[KC], vf[KC], pf0, pf1, df0, df1, rq[4], rdo[4], tf[4], qa, qb0, qb1;

  // dQ^T work of one slot: key step ks (32 keys) of the previous pair's tiles: operands requested two slots before their MFMAs
  auto dq_load = [&](int ks, int rd) {
    if constexpr (ABL & 2) return;
    qa = tr2(ka0, ka1, 4096 * ks);
    qb0 = tr2(da0[0], da1[0], rd + 2048 * ks);
    qb1 = tr2(da0[1], da1[1], rd + 2048 * ks);
  };
  auto dq_mma = [&]() {
    if constexpr (ABL & 2) return;
    dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, qb0, dq0, 0, 0, 0);
    SB();
    dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, qb1, dq1, 0, 0, 0);
    SB();
  };

  // One period (as bwd_dkdv_slot_kernel's) plus: the dS^T pieces of the current sub-slice written to LDS at DSW, and four key
  // steps KS0..KS0+3 of the dQ^T tiles of the PREVIOUS pair (dS^T buffer at RD; RD < 0: none), requested in slots 1, 5, 9, 13 and
  // issued after the MFMAs of slots 3, 7, 11, 15.
  auto period = [&](auto hn_c, auto hc_c, auto subn_c, auto subc_c, auto subp_c, auto dsw_c, auto rd_c, auto ks0_c, int nr0, int nr1,
                    int ct0, int ct1, int pr0, int pr1, int ph16, f32x16& ns, f32x16& ndp, f32x16& cs, f32x16& cdp, auto&& vm) {
    // vm(ic<slot>): the pair's vector-memory work (hand-off, next stage's LDS-DMA), one piece per slot, behind the slot's MFMA: a
    // VMEM issue blocks the wave for 60-180 cycles, which only hides while the partner wave and the wave's own MFMA keep the pipe busy
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
  constexpr bool STAMPS = (ABL & 32) != 0;
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, k_t0 = 0, k_r0 = 0, ts = 0;
  if constexpr (STAMPS) {
    k_t0 = stamp();
    k_r0 = __builtin_amdgcn_s_memrealtime();
  }
  auto lap = [&](int i) {   // cycles since the previous lap -> phase i
    if constexpr (STAMPS) {
      const unsigned long long t = stamp();
      ph[i] += t - ts;
      ts = t;
    }
  };
  int it = 0;
  for (int bh = g; bh < BH; bh += ngroups, ++it) {
    const size_t base = head_base(lay, bh);
    const rsrc_t krs = make_rsrc(k + base, mat_bytes);
    const rsrc_t vrs = make_rsrc(v + base, mat_bytes);
    const raw_rsrc_t qraw = make_raw_rsrc(q + base, mat_bytes), doraw = make_raw_rsrc(dout + base, mat_bytes);
    const raw_rsrc_t kraw = make_raw_rsrc(k + base, mat_bytes);
    const int crow = (w == 0 ? 0 : (int)(ndelta - nlc)) + bh * N;   // waves 0 / 1 move -L/tau / -delta (one resource: ndelta = nlc + rows)
    const unsigned want0 = (unsigned)it * (unsigned)nkb;   // flag value a pair carries before position 0 of this head has added

    {   // (lane id recomputed per head with v_mbcnt: hoisted to kernel entry these offsets are spilled around the sweep)
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
    auto pair_of = [&](int t) { int j = j0 + t; return j >= npairs ? j - npairs : j; };   // t < npairs
    auto stage_dma = [&](int j, int dst) {   // 64 queries of Q and dO (wave w: rows 8w..8w+7 of each), waves 0 / 1 the row constants
      const int soff = (HS * j + 8 * w) * ld * (int)sizeof(T);
      dma16(qraw, smem_addr + dst + 1024 * w, dma_voff, soff);
      dma16(doraw, smem_addr + dst + TB + 1024 * w, dma_voff, soff);
      if (w < 2) dma4(craw, smem_addr + dst + 2 * TB + 256 * w, 4 * lane, (crow + HS * j) * 4);
    };
    // the workgroup's 256 key rows as an LDS image (A operand of dQ^T by transposed reads): wave w moves pieces w, w+8, w+16, w+24
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gp = w + 8 * i;
      dma16(kraw, smem_addr + KIMG + 1024 * gp, dma_voff, (kb * 256 + 8 * gp) * ld * (int)sizeof(T));
    }
    stage_dma(pair_of(0), 0);
    stage_dma(pair_of(1), STG);
    dma_wait_all();
    __syncthreads();
    if (lay.young_prio && w >= 4) __builtin_amdgcn_s_setprio(1);

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

    unsigned pre = 0;   // early poll of the flag this wave checks at the next pair boundary
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    dq0 = dq1 = pd0 = pd1 = zero4;
    // ---- hand-off, branch-free: every pair runs the same vector-memory stream; what differs between chain positions (and in the
    // first / last pairs of a sweep) is chosen by scalar selects: a position-0 "load" reads a zero page, a not-yet-existing tile is
    // "stored" to a dummy page, a not-yet-due flag goes to a dummy word, the last position stores tau * sum through dq's resource.
    // Indexing: the tiles formed DURING local pair s belong to the queries of local pair s-1 (whose dS^T was written in pair s-1).
    const rsrc_t dqrs = make_rsrc(dq + base, ((uint32_t)(N - 1) * ld + D) * 4u);
    const int dq_voff = ((32 * ss + i16) * ld + 16 * db + 4 * g4) * 4;   // this lane's 16 B of the wave's first tile inside a pair of dq
    const int sl_voff = lane * 16;                                      // ... inside the wave's 2 KiB of a slab pair (register-major)
    const int slab_w = FUSED_SLAB0 + g * N * 256 + w * 2048;            // byte offset of this wave's tiles of pair 0 in this group's slab
    constexpr int ZERO_PAGE = FUSED_PAGE0, DUMMY_PAGE = FUSED_PAGE0 + 2048, DUMMY_FLAG = 32;
    int sig_off = DUMMY_FLAG, poll_off = gflag0;
    unsigned sig_val = 0, want = 0;
    int st_soff = DUMMY_PAGE, st_d2 = 1024, st_voff = sl_voff, ld_soff = ZERO_PAGE;
    float st_scale = 1.0f;
    rsrc_t st_rs = prs;
    auto wrap = [&](int j) { return j < 0 ? j + npairs : (j >= npairs ? j - npairs : j); };
    // scalar state of the hand-off steps of local pair t, whose query pair is jt (t up to npairs + 2: the drain)
    auto prepare = [&](int t, int jt) {
      const int t3 = t - 3, t1 = t - 1;
      const bool v3 = t3 >= 0 && t3 < npairs, v1 = t1 >= 0 && t1 < npairs;
      const int j3 = wrap(jt - 3), j1 = wrap(jt - 1);
      sig_off = v3 ? gflag0 + 4 * j3 : DUMMY_FLAG;   // tiles of pair t-3: stored at the end of pair t-2, drained by the end of pair t-1
      sig_val = want0 + (unsigned)(t3 >> 2) + 1u;
      const int p1 = t1 >> 2;                        // tiles of pair t-1: formed during THIS pair, stored at its end
      const bool last = v1 && p1 == nkb - 1;
      const int sl = slab_w + j1 * 16384;
      st_rs = last ? dqrs : prs;
      st_voff = last ? dq_voff : sl_voff;
      st_soff = last ? HS * j1 * ld * 4 : (v1 ? sl : DUMMY_PAGE);
      st_d2 = last ? 16 * ld * 4 : 1024;
      st_scale = last ? tau : 1.0f;
      ld_soff = (v1 && p1 > 0) ? sl : ZERO_PAGE;
      want = v1 ? want0 + (unsigned)p1 : 0u;
      poll_off = gflag0 + 4 * j1;
    };
    auto check_flag = [&]() {   // the previous position has published the tiles this pair loads (normally seen by the early poll)
      if constexpr (ABL & (1 | 256)) return;
      unsigned seen = (unsigned)__builtin_amdgcn_readfirstlane((int)pre);
      int spins = 0;
      while (seen < want) {
        __builtin_amdgcn_s_sleep(4);
        seen = (unsigned)__builtin_amdgcn_readfirstlane((int)flag_load_waited(praw, poll_off));
        if constexpr (STAMPS) ph[5] += 1;
        if (++spins > (1 << 22)) {   // a chain member is not running: give up loudly (results are then wrong)
          if (lane == 0) flag_store(hand, 1u + (unsigned)blockIdx.x);
          break;
        }
      }
      asm volatile("" ::: "memory");   // the loads below stay behind the poll
    };
    f32x4 s0 = zero4, s1 = zero4;
    auto ho_signal = [&]() {
      if constexpr (ABL & (1 | 256)) return;
      if (w == 0 && lane == 0) __builtin_amdgcn_raw_buffer_store_b32(sig_val, prs, 0, sig_off, 16 /* sc1 */);
    };
    auto ho_finish = [&]() {   // dq0 / dq1: this workgroup's tiles; pd0 / pd1: the previous position's running sums
      s0 = (pd0 + dq0) * st_scale;
      s1 = (pd1 + dq1) * st_scale;
      asm volatile("" : "+v"(s0), "+v"(s1)::"memory");   // formed HERE (hipcc otherwise sinks the sums behind the flag store and waits for it)
      dq0 = dq1 = zero4;
    };
    auto ho_store = [&](int which) {
      if constexpr (ABL & (1 | 64)) return;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, which ? s1 : s0), st_rs, st_voff, st_soff + (which ? st_d2 : 0), 16 /* sc1 */);
    };
    auto ho_load = [&](int which) {
      if constexpr (ABL & (1 | 128)) return;
      const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, sl_voff, ld_soff + 1024 * which, 16 /* sc1 */));
      if (which) pd1 = x; else pd0 = x;
    };

    int jt_ = j0;   // query pair of the current local pair
    auto pair_body = [&](auto par_c, int t) {
      constexpr int PAR = decltype(par_c)::value;
      constexpr int WR = PAR * 32768, RD = (PAR ^ 1) * 32768;   // dS^T buffer written in this pair / read for the dQ^T tiles
      if constexpr (STAMPS) ts = stamp();
      __syncthreads();
      lap(4);
      const int ns_ = cs_ == 2 * STG ? 0 : cs_ + STG;       // slot of the next stage
      const int n2_ = ns_ == 2 * STG ? 0 : ns_ + STG;       // ... and of the one after (last read in pair t-1)
      const int nr0 = ra.b[0] + ns_, nr1 = ra.b[1] + ns_, nh16 = 16 * h + ns_;
      const int jt = jt_;
      bool more = false;
      int jn = 0, dsoff = 0;
      auto vmA = [&](auto slot_c) {
        constexpr int S = decltype(slot_c)::value;
        // (the scalar bookkeeping rides behind the first MFMAs: at the top of the pair it would run with both waves of a SIMD idle)
        if constexpr (S == 0) {
          prepare(t, jt);
          check_flag();
          more = t + 2 < npairs;
          jn = wrap(jt + 2);
          dsoff = (HS * jn + 8 * w) * ld * (int)sizeof(T);
        }
        if constexpr (S == 1) ho_signal();
        // next-but-one stage by LDS-DMA, issued by waves 0-3 only (rows 8w..8w+7 and 8w+32..8w+39 of Q and of dO): the older half
        // of the workgroup wins the issue arbitration and idles at the pair's barrier, so the 60-180 cycles a VMEM issue blocks a
        // wave are free there and come off the younger half's critical path
        if constexpr (S == 4) { if (more && w < 4) dma16(qraw, smem_addr + n2_ + 1024 * w, dma_voff, dsoff); }
        if constexpr (S == 5) { if (more && w < 4) dma16(doraw, smem_addr + n2_ + TB + 1024 * w, dma_voff, dsoff); }
        if constexpr (S == 6) { if (more && w < 4) dma16(qraw, smem_addr + n2_ + 1024 * (w + 4), dma_voff, dsoff + 32 * ld * (int)sizeof(T)); }
        if constexpr (S == 7) { if (more && w < 4) dma16(doraw, smem_addr + n2_ + TB + 1024 * (w + 4), dma_voff, dsoff + 32 * ld * (int)sizeof(T)); }
        if constexpr (S == 2) { if (more && w < 2) dma4(craw, smem_addr + n2_ + 2 * TB + 256 * w, 4 * lane, (crow + HS * jn) * 4); }
        if constexpr (S == 8) ho_load(0);
        if constexpr (S == 9) ho_load(1);
      };
      auto vmB = [&](auto slot_c) {
        constexpr int S = decltype(slot_c)::value;
        if constexpr (S == 0 && !(ABL & (9 | 256))) pre = flag_load_untracked(praw, gflag0 + 4 * jt);   // early poll for the next pair's check
      };
      // (in pair 0 the dQ^T products run on whatever the dS^T buffer holds; the next pair's finish stores them to the dummy page)
      lap(0);
      period(T1, T1, ic<1>{}, ic<0>{}, ic<0>{}, ic<WR>{}, ic<RD>{}, ic<0>{}, cr0, cr1, ct0, ct1, nr0, nr1, nh16, sB, dpB, sA, dpA, vmA);
      lap(1);
      period(T1, T1, ic<0>{}, ic<1>{}, ic<1>{}, ic<WR + 16384>{}, ic<RD>{}, ic<4>{}, nr0, nr1, ct0, ct1, nr0, nr1, nh16, sA, dpA, sB, dpB, vmB);
      // the pair's tiles are complete: add the previous position's sums and store (the LAST two vector-memory operations of the pair)
      ho_finish();
      ho_store(0);
      ho_store(1);
      lap(2);
      // Everything but those two stores: this wave's LDS-DMA pieces of stage t+2 have landed, the previous pair's tile stores are
      // written through (a write-through store stays counted for microseconds: waiting for the pair's own stores here cost 0.15 ms)
      if constexpr (ABL & (1 | 64)) dma_wait_all();
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      lap(3);
      cs_ = ns_;
      cr0 = nr0; cr1 = nr1; ch16 = nh16;
      ct0 = ta.b[0] + ns_; ct1 = ta.b[1] + ns_;
      jt_ = jt + 1 == npairs ? 0 : jt + 1;
    };
    for (int t = 0; t < npairs; t += 2) {
      pair_body(T0, t);
      pair_body(T1, t + 1);
    }
    // drain: the last pair's dQ^T tiles (its dS^T sits in buffer 1: npairs is even), then the remaining flags
    __syncthreads();
    prepare(npairs, j0);
    check_flag();
    ho_signal();
    ho_load(0);
    ho_load(1);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      dq_load(ks, 32768);
      SB();
      dq_mma();
    }
    ho_finish();
    ho_store(0);
    ho_store(1);
    dma_wait_all();
    __syncthreads();
    prepare(npairs + 1, wrap(j0 + 1));
    ho_signal();
    prepare(npairs + 2, wrap(j0 + 2));
    ho_signal();

    if constexpr (STAMPS) {
      const int slot = blockIdx.x * 8 + w;
      const unsigned long long k_t1 = stamp(), k_r1 = __builtin_amdgcn_s_memrealtime();
      if (slot < 8192 && lane == 0) {
        for (int j = 0; j < 6; ++j) g_phase_cycles[slot * 8 + j] = ph[j];
        g_phase_cycles[slot * 8 + 6] = k_t1 - k_t0;   // (to the end of this head's sweep; the last head's values stay)
        g_phase_cycles[slot * 8 + 7] = k_r1 - k_r0;
      }
    }
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
  }
}

}  // namespace fa
