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

template <typename T, int D, int BN, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE)))
fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, float* __restrict__ o,
           float* __restrict__ aux_l, float* __restrict__ aux_m, int N, int nqb, int BH, Layout lay, int causal,
           int aux_mode, float tau) {
  using A = Atom<T>;
  typedef typename A::frag frag;
  constexpr int KC = D / 16, KT = BN / 32, DT = D / 32;
  constexpr int TB = A::template tile_bytes<D>(BN);
  __shared__ __attribute__((aligned(16))) char smem_raw[4 * TB];
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
  __syncthreads();

  auto tile = [&](auto par, auto first_c, int t) {
    constexpr int PAR = decltype(par)::value;
    constexpr bool FIRST = decltype(first_c)::value != 0;
    const int kbase = t * BN;
    const bool more = t + 1 < nt;
    if (more) {
      sk.load(krs, kbase + BN);
      sv.load(vrs, kbase + BN);
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
    if (h == 0) {
      const size_t ri = (size_t)bh * N + qrow;
      if (aux_mode == AUX_FA1) {   // l = sum exp(tau*s - m), m = tau * rowmax(s)
        aux_l[ri] = l_tot * __builtin_amdgcn_exp2f((m_ref - m_true) * c);
        aux_m[ri] = m_true * tau;
      } else {
        aux_l[ri] = m_ref * tau + __logf(l_tot);
      }
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
    nlc[row] = -L * inv_tau;
  }
}

// ---------------------------------------------------------------------------------------------
// Backward dK / dV: a workgroup = NW waves = NW*KPW keys of one (batch*head); each wave keeps K, V fragments and
// the dK^T, dV^T accumulators of its KPW keys in registers while the workgroup sweeps 32-row query slices
// (Q, dO tiles + their nlc, delta staged in LDS, double buffered).
// ---------------------------------------------------------------------------------------------
template <typename T, int D, int KPW, int NW, int QS, int MODE = 0>
__global__ void __launch_bounds__(NW * 64)
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
  TileStager<T, D, QS, NT> sq, sdo;
  sq.init(tid, ld);
  sdo.init(tid, ld);
  float st_nl = 0.f, st_de = 0.f;
  auto stage_load = [&](int qi) {
    sq.load(qrs, qi * QS);
    sdo.load(dors, qi * QS);
    if (tid < QS) {
      const int row = qi * QS + tid;
      st_nl = row < N ? nlg[row] : 0.f;
      st_de = row < N ? deg[row] : 0.f;
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
  if (qi_begin < nqi) {
    stage_load(qi_begin);
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
    if (more) stage_load(qi + 1);
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
    constexpr bool SLOT = (MODE == 3 || MODE == 93) && NSUB == 4 && D == 64 && KT == 1 && sizeof(T) == 2;
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
        auto me = [&](f32x16& x, int i) { x[i] = __builtin_amdgcn_exp2f(x[i] * c); };
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
              if (kq < 3) rdo[kq + 1] = A::template row_frag<D>(tdo, ra, 32 * SNc, kq + 1);
            }
            if constexpr (HC) { me(cs, 2 * kq); me(cs, 2 * kq + 1); }
            SB();
          }
          // slot 4
          if constexpr (HN) A::mma_c(ndp, rdo[0], vf[0][0], cD);
          if constexpr (HC) {
            pf0 = A::pack(cs, 0);
            cdp[0] = cs[0] * cdp[0];
            tf[0] = A::template tr_frag<D>(tdo, ta, 32 * SCc, 0);
          }
          SB();
          // slots 5-7
#pragma unroll
          for (int kq = 1; kq < 4; ++kq) {
            if constexpr (HN) A::mma(ndp, rdo[kq], vf[0][kq]);
            if constexpr (HC) {
              me(cs, 6 + 2 * kq); me(cs, 7 + 2 * kq);
              tf[kq] = A::template tr_frag<D>(tdo, ta, 32 * SCc + 16 * (kq >> 1), kq & 1);
            }
            SB();
          }
          if constexpr (HC) {
            // slot 8
            A::mma(acc_dv[0][0], tf[0], pf0);
            me(cs, 14); me(cs, 15);
            tf[0] = A::template tr_frag<D>(tq, ta, 32 * SCc, 0);
            SB();
            // slot 9
            A::mma(acc_dv[1][0], tf[1], pf0);
            pf1 = A::pack(cs, 1);
            cdp[1] = cs[1] * cdp[1];
            tf[1] = A::template tr_frag<D>(tq, ta, 32 * SCc, 1);
            SB();
            // slot 10
            A::mma(acc_dv[0][0], tf[2], pf1);
#pragma unroll
            for (int i = 2; i < 8; ++i) cdp[i] = cs[i] * cdp[i];
            tf[2] = A::template tr_frag<D>(tq, ta, 32 * SCc + 16, 0);
            SB();
            // slot 11
            A::mma(acc_dv[1][0], tf[3], pf1);
            df0 = A::pack(cdp, 0);
            cdp[8] = cs[8] * cdp[8];
            tf[3] = A::template tr_frag<D>(tq, ta, 32 * SCc + 16, 1);
            SB();
            // slot 12
            A::mma(acc_dk[0][0], tf[0], df0);
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
            cdp[15] = cs[15] * cdp[15];
            df1 = A::pack(cdp, 1);
          }
          if constexpr (HP) {
            rq[2] = A::template row_frag<D>(tq, ra, 32 * (SNc + 1), 2);
            rq[3] = A::template row_frag<D>(tq, ra, 32 * (SNc + 1), 3);
          }
          SB();
          // slot 14
          if constexpr (HC) A::mma(acc_dk[0][0], tf[2], df1);
          if constexpr (HP) ld_c(cS, 0, SNc + 1);
          SB();
          // slot 15
          if constexpr (HC) A::mma(acc_dk[1][0], tf[3], df1);
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
    constexpr bool PIPE = MODE == 0 && NSUB == 4 && D <= 64;   // (needs ~250 VGPRs at d = 64; not for d = 128)
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
            s[kt][i] = __builtin_amdgcn_exp2f(s[kt][i] * c);
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
              A::mma_c(dp[kt], ado, vf[kt][kc], nd16);
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
          for (int i = 0; i < 16; ++i) s[kt][i] = __builtin_amdgcn_exp2f(s[kt][i] * c);
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
          for (int i = 0; i < 16; ++i) dp[kt][i] = s[kt][i] * dp[kt][i];
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
// Backward dQ: same shape as the forward (4 waves x 32 query rows, K/V tiles of BN keys through LDS).
// ---------------------------------------------------------------------------------------------
template <typename T, int D, int BN>
__global__ void __launch_bounds__(256)
bwd_dq_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ dout,
              const float* __restrict__ nlc, const float* __restrict__ ndelta, float* __restrict__ dq, int N, int nqb,
              int BH, Layout lay, int causal, float tau) {
  using A = Atom<T>;
  typedef typename A::frag frag;
  constexpr int KC = D / 16, KT = BN / 32, DT = D / 32;
  constexpr int TB = A::template tile_bytes<D>(BN);
  __shared__ __attribute__((aligned(16))) char smem_raw[4 * TB];
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
  __syncthreads();

  auto tile = [&](auto par, int t) {
    constexpr int PAR = decltype(par)::value;
    const int kbase = t * BN;
    const bool more = t + 1 < nt;
    if (more) {
      sk.load(krs, kbase + BN);
      sv.load(vrs, kbase + BN);
    }
    lds_char* tk = smem + PAR * TB;
    lds_char* tv = smem + (2 + PAR) * TB;
    const bool active = !causal || kbase <= q0 + 31;
    if (active) {
      f32x16 s[KT], dp[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        A::mma_c(s[kt], A::template row_frag<D>(tk, ra, 32 * kt, 0), qf[0], zero16());
        A::mma_c(dp[kt], A::template row_frag<D>(tv, ra, 32 * kt, 0), dof[0], nd16);
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
        for (int i = 0; i < 16; ++i) dp[kt][i] = s[kt][i] * dp[kt][i];
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
