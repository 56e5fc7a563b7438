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
// Backward = preprocess (delta = rowsum(dO*O), -L/tau) + a key-stationary dK/dV kernel (S, dP with the KEY on the
// lane; P and dS feed dV^T += dO^T P and dK^T += Q^T dS from registers) + a query-stationary dQ kernel
// (S^T, dP^T with the query on the lane; dQ^T += K^T dS^T).  No atomics: results are bitwise reproducible
// (the reference's FA-2 backward uses atomicAdd for dQ, src/flash_attn2_bw.cu:228).
#pragma once
#include "fa_atoms.h"

namespace fa {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int AUX_FA1 = 1;  // l = sum exp(s - m), m = row max            (src/flash_attn_fw.cu:259-276)
constexpr int AUX_FA2 = 2;  // l = logsumexp, m untouched                  (src/flash_attn2_fw.cu:279-294)

// ---------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------
template <typename T, int D, int BN>
__global__ void __launch_bounds__(256)
fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, float* __restrict__ o,
           float* __restrict__ aux_l, float* __restrict__ aux_m, int N, int nqb, int BH, int causal, int aux_mode,
           float tau) {
  using A = Atom<T>;
  typedef typename A::frag frag;
  constexpr int KC = D / 16, KT = BN / 32, DT = D / 32;
  constexpr int TB = A::template tile_bytes<D>(BN);
  __shared__ __attribute__((aligned(16))) char smem_raw[4 * TB];
  lds_char* smem = (lds_char*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  int bh, qb;
  map_block(blockIdx.x, BH, nqb, bh, qb);
  if (causal) qb = nqb - 1 - qb;  // heaviest query blocks first
  const int q0 = qb * 128 + w * 32, qrow = q0 + r;
  const bool qvalid = qrow < N;
  const size_t base = (size_t)bh * N * D;
  const T* kg = k + base;
  const T* vg = v + base;
  const float c = tau * LOG2E;

  frag qf[KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc)
    qf[kc] = qvalid ? A::load_global(q + base + (size_t)qrow * D + 16 * kc + 8 * h) : A::zero();

  f32x16 acc_o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc_o[dt][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int kmax = causal ? min(N, qb * 128 + 128) : N;
  const int nt = (kmax + BN - 1) / BN;
  TileStager<T, D, BN, 256> sk, sv;
  sk.load(kg, 0, N, tid);
  sv.load(vg, 0, N, tid);
  sk.store(smem, tid);
  sv.store(smem + 2 * TB, tid);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int kbase = t * BN;
    if (t + 1 < nt) {
      sk.load(kg, kbase + BN, N, tid);
      sv.load(vg, kbase + BN, N, tid);
    }
    lds_char* tk = smem + (t & 1) * TB;
    lds_char* tv = smem + (2 + (t & 1)) * TB;
    const bool active = !causal || kbase <= q0 + 31;  // wave-uniform
    if (active) {
      f32x16 s[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kt][i] = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) A::mma(s[kt], A::template row_frag<D>(tk, 32 * kt + r, kc, h), qf[kc]);
      }
      const bool need_mask = (kbase + BN > N) || (causal && kbase + BN - 1 > q0);  // wave-uniform
      if (need_mask) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = kbase + 32 * kt + acc_row(i, h);
            if (key >= N || (causal && key > qrow)) s[kt][i] = -INFINITY;
          }
      }
      float mx = s[0][0];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kt][i]);
      mx = xhalf_max(mx);
      const float m_new = fmaxf(m_run, mx);
      const float mc = m_new * c;
      const float alpha = __builtin_amdgcn_exp2f(m_run * c - mc);
      float rowsum = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][i], c, -mc));
          s[kt][i] = p;
          rowsum += p;
        }
      l_run = l_run * alpha + rowsum;
      if (!__all(m_new == m_run)) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc_o[dt][i] *= alpha;
      }
      m_run = m_new;
      frag pf[KT][2];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        pf[kt][0] = A::pack(s[kt], 0);
        pf[kt][1] = A::pack(s[kt], 1);
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
            A::mma(acc_o[dt], A::template tr_frag<D>(tv, 32 * kt + 16 * s2, dt, lane), pf[kt][s2]);
    }
    if (t + 1 < nt) {
      sk.store(smem + ((t + 1) & 1) * TB, tid);
      sv.store(smem + (2 + ((t + 1) & 1)) * TB, tid);
    }
    __syncthreads();
  }

  const float l_tot = xhalf_sum(l_run);
  const float inv = 1.0f / l_tot;
  if (qvalid) {
    float* orow = o + base + (size_t)qrow * D;
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
      if (aux_mode == AUX_FA1) {
        aux_l[ri] = l_tot;
        aux_m[ri] = m_run * tau;
      } else {
        aux_l[ri] = m_run * tau + __logf(l_tot);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward preprocess: nd = -rowsum(dO * O), nl = -L / tau with L = m + log(l) (FA-1 side outputs) or L = l (FA-2).
// The reference recomputes D_i per (i, j) tile (src/flash_attn_bw.cu:194-197); once per row gives the same value.
// ---------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ void __launch_bounds__(256)
bwd_prep_kernel(const float* __restrict__ o, const T* __restrict__ dout, const float* __restrict__ l,
                const float* __restrict__ m, float* __restrict__ nl, float* __restrict__ nd, long rows, int aux_mode,
                float inv_tau) {
  constexpr int LPR = D / 8;  // lanes per row, 8 elements each
  constexpr int RPB = 256 / LPR;
  const int tid = threadIdx.x;
  const long row = (long)blockIdx.x * RPB + tid / LPR;
  const int part = tid % LPR;
  float sum = 0.f;
  if (row < rows) {
    const float* op = o + row * D + part * 8;
    const T* dp = dout + row * D + part * 8;
    f32x4 o0 = *reinterpret_cast<const f32x4*>(op), o1 = *reinterpret_cast<const f32x4*>(op + 4);
    typename Atom<T>::frag df = Atom<T>::load_global(dp);
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += o0[j] * (float)df[j] + o1[j] * (float)df[4 + j];
  }
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  if (row < rows && part == 0) {
    nd[row] = -sum;
    const float L = (aux_mode == AUX_FA1) ? (m[row] + __logf(l[row])) : l[row];
    nl[row] = -L * inv_tau;
  }
}

// ---------------------------------------------------------------------------------------------
// Backward dK / dV: a workgroup = 4 waves = 4*KPW keys of one (batch*head); each wave keeps K, V fragments and
// the dK^T, dV^T accumulators of its KPW keys in registers while the workgroup sweeps 32-row query slices
// (Q, dO tiles + their -L/tau, -delta staged in LDS, double buffered).
// ---------------------------------------------------------------------------------------------
template <typename T, int D, int KPW>
__global__ void __launch_bounds__(256)
bwd_dkdv_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ dout,
                const float* __restrict__ nl, const float* __restrict__ nd, float* __restrict__ dk,
                float* __restrict__ dv, int N, int nkb, int BH, int causal, float tau) {
  using A = Atom<T>;
  typedef typename A::frag frag;
  constexpr int KC = D / 16, KT = KPW / 32, DT = D / 32, BK = 4 * KPW;
  constexpr int TB = A::template tile_bytes<D>(32);
  constexpr int BUF = 2 * TB + 256;  // Q tile, dO tile, 32 x nl, 32 x nd
  __shared__ __attribute__((aligned(16))) char smem_raw[2 * BUF];
  lds_char* smem = (lds_char*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  int bh, kb;
  map_block(blockIdx.x, BH, nkb, bh, kb);
  const int kb0 = kb * BK, kw0 = kb0 + w * KPW;
  const size_t base = (size_t)bh * N * D;
  const T* qg = q + base;
  const T* dog = dout + base;
  const float* nlg = nl + (size_t)bh * N;
  const float* ndg = nd + (size_t)bh * N;
  const float c = tau * LOG2E;

  frag kf[KT][KC], vf[KT][KC];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int key = kw0 + 32 * kt + r;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const size_t off = base + (size_t)key * D + 16 * kc + 8 * h;
      kf[kt][kc] = key < N ? A::load_global(k + off) : A::zero();
      vf[kt][kc] = key < N ? A::load_global(v + off) : A::zero();
    }
  }
  f32x16 acc_dk[DT][KT], acc_dv[DT][KT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc_dk[dt][kt][i] = 0.f;
        acc_dv[dt][kt][i] = 0.f;
      }

  const int nqi = (N + 31) / 32;
  const int qi_begin = causal ? (kb0 / 32) : 0;  // query slices entirely above the key block are fully masked
  TileStager<T, D, 32, 256> sq, sdo;
  float st_nl = 0.f, st_nd = 0.f;
  auto stage_load = [&](int qi) {
    sq.load(qg, qi * 32, N, tid);
    sdo.load(dog, qi * 32, N, tid);
    if (tid < 32) {
      const int row = qi * 32 + tid;
      st_nl = row < N ? nlg[row] : 0.f;
      st_nd = row < N ? ndg[row] : 0.f;
    }
  };
  auto stage_store = [&](int buf) {
    lds_char* b = smem + buf * BUF;
    sq.store(b, tid);
    sdo.store(b + TB, tid);
    if (tid < 32) {
      *FA_LDS(float, b + 2 * TB + 4 * tid) = st_nl;
      *FA_LDS(float, b + 2 * TB + 128 + 4 * tid) = st_nd;
    }
  };
  if (qi_begin < nqi) {
    stage_load(qi_begin);
    stage_store(0);
  }
  __syncthreads();

  for (int qi = qi_begin; qi < nqi; ++qi) {
    const int it = qi - qi_begin;
    if (qi + 1 < nqi) stage_load(qi + 1);
    lds_char* buf = smem + (it & 1) * BUF;
    lds_char* tq = buf;
    lds_char* tdo = buf + TB;
    const int qi0 = qi * 32;
    const bool active = (kw0 < N) && (!causal || qi0 + 31 >= kw0);  // wave-uniform
    if (active) {
      f32x16 s[KT], dp[KT];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 a = *FA_LDS(f32x4, buf + 2 * TB + 4 * (8 * g + 4 * h));
        const f32x4 b = *FA_LDS(f32x4, buf + 2 * TB + 128 + 4 * (8 * g + 4 * h));
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            s[kt][4 * g + j] = a[j];
            dp[kt][4 * g + j] = b[j];
          }
      }
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const frag aq = A::template row_frag<D>(tq, r, kc, h);
        const frag ado = A::template row_frag<D>(tdo, r, kc, h);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          A::mma(s[kt], aq, kf[kt][kc]);
          A::mma(dp[kt], ado, vf[kt][kc]);
        }
      }
      const bool need_mask = causal && (kw0 + KPW - 1 > qi0);  // wave-uniform
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float p = __builtin_amdgcn_exp2f(s[kt][i] * c);
          if (need_mask && (kw0 + 32 * kt + r > qi0 + acc_row(i, h))) p = 0.f;
          s[kt][i] = p;
          dp[kt][i] = p * dp[kt][i];
        }
      frag pf[KT][2], dsf[KT][2];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          pf[kt][s2] = A::pack(s[kt], s2);
          dsf[kt][s2] = A::pack(dp[kt], s2);
        }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const frag adoT = A::template tr_frag<D>(tdo, 16 * s2, dt, lane);
          const frag aqT = A::template tr_frag<D>(tq, 16 * s2, dt, lane);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            A::mma(acc_dv[dt][kt], adoT, pf[kt][s2]);
            A::mma(acc_dk[dt][kt], aqT, dsf[kt][s2]);
          }
        }
    }
    if (qi + 1 < nqi) stage_store((it + 1) & 1);
    __syncthreads();
  }

#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int key = kw0 + 32 * kt + r;
    if (key < N) {
      float* dkrow = dk + base + (size_t)key * D;
      float* dvrow = dv + base + (size_t)key * D;
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
              const float* __restrict__ nl, const float* __restrict__ nd, float* __restrict__ dq, int N, int nqb,
              int BH, int causal, float tau) {
  using A = Atom<T>;
  typedef typename A::frag frag;
  constexpr int KC = D / 16, KT = BN / 32, DT = D / 32;
  constexpr int TB = A::template tile_bytes<D>(BN);
  __shared__ __attribute__((aligned(16))) char smem_raw[4 * TB];
  lds_char* smem = (lds_char*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  int bh, qb;
  map_block(blockIdx.x, BH, nqb, bh, qb);
  if (causal) qb = nqb - 1 - qb;
  const int q0 = qb * 128 + w * 32, qrow = q0 + r;
  const bool qvalid = qrow < N;
  const size_t base = (size_t)bh * N * D;
  const T* kg = k + base;
  const T* vg = v + base;
  const float c = tau * LOG2E;

  frag qf[KC], dof[KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const size_t off = base + (size_t)qrow * D + 16 * kc + 8 * h;
    qf[kc] = qvalid ? A::load_global(q + off) : A::zero();
    dof[kc] = qvalid ? A::load_global(dout + off) : A::zero();
  }
  const float nlq = qvalid ? nl[(size_t)bh * N + qrow] : 0.f;
  const float ndq = qvalid ? nd[(size_t)bh * N + qrow] : 0.f;

  f32x16 acc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[dt][i] = 0.f;

  const int kmax = causal ? min(N, qb * 128 + 128) : N;
  const int nt = (kmax + BN - 1) / BN;
  TileStager<T, D, BN, 256> sk, sv;
  sk.load(kg, 0, N, tid);
  sv.load(vg, 0, N, tid);
  sk.store(smem, tid);
  sv.store(smem + 2 * TB, tid);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int kbase = t * BN;
    if (t + 1 < nt) {
      sk.load(kg, kbase + BN, N, tid);
      sv.load(vg, kbase + BN, N, tid);
    }
    lds_char* tk = smem + (t & 1) * TB;
    lds_char* tv = smem + (2 + (t & 1)) * TB;
    const bool active = !causal || kbase <= q0 + 31;
    if (active) {
      f32x16 s[KT], dp[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s[kt][i] = nlq;
          dp[kt][i] = ndq;
        }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          A::mma(s[kt], A::template row_frag<D>(tk, 32 * kt + r, kc, h), qf[kc]);
          A::mma(dp[kt], A::template row_frag<D>(tv, 32 * kt + r, kc, h), dof[kc]);
        }
      }
      const bool need_mask = causal && (kbase + BN - 1 > q0);
      frag dsf[KT][2];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float p = __builtin_amdgcn_exp2f(s[kt][i] * c);
          if (need_mask && (kbase + 32 * kt + acc_row(i, h) > qrow)) p = 0.f;
          dp[kt][i] = p * dp[kt][i];
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
            A::mma(acc[dt], A::template tr_frag<D>(tk, 32 * kt + 16 * s2, dt, lane), dsf[kt][s2]);
    }
    if (t + 1 < nt) {
      sk.store(smem + ((t + 1) & 1) * TB, tid);
      sv.store(smem + (2 + ((t + 1) & 1)) * TB, tid);
    }
    __syncthreads();
  }

  if (qvalid) {
    float* row = dq + base + (size_t)qrow * D;
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
// ---------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ void __launch_bounds__(64)
probe_kernel(const T* __restrict__ tile_in /*[64][D]*/, const T* __restrict__ b_in /*[32][D]*/,
             float* __restrict__ row_out /*[D/16][64][8]*/, float* __restrict__ tr_out /*[D/32][4][64][8]*/,
             float* __restrict__ mma_out /*[2][64][16]*/, float* __restrict__ swap_out /*[2][64]*/) {
  using A = Atom<T>;
  typedef typename A::frag frag;
  __shared__ __attribute__((aligned(16))) char smem_raw[A::template tile_bytes<D>(64)];
  lds_char* smem = (lds_char*)smem_raw;
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  TileStager<T, D, 64, 64> st;
  st.load(tile_in, 0, 64, lane);
  st.store(smem, lane);
  __syncthreads();
  for (int kc = 0; kc < D / 16; ++kc) {
    frag f = A::template row_frag<D>(smem, 32 + r, kc, h);  // rows 32..63
    for (int j = 0; j < 8; ++j) row_out[(kc * 64 + lane) * 8 + j] = (float)f[j];
  }
  for (int ct = 0; ct < D / 32; ++ct)
    for (int s = 0; s < 4; ++s) {
      frag f = A::template tr_frag<D>(smem, 16 * s, ct, lane);
      for (int j = 0; j < 8; ++j) tr_out[((ct * 4 + s) * 64 + lane) * 8 + j] = (float)f[j];
    }
  // X = tile[0:32] . b^T  (32 x 32, sum over D); then Y = tile[0:32, 0:32]^T-style product through pack/tr_frag:
  // Y[c][n] = sum_m tile[m][c] * X[m][n]  for c < 32  (A operand = tr_frag of the tile, B operand = pack(X)).
  f32x16 x;
  for (int i = 0; i < 16; ++i) x[i] = 0.f;
  for (int kc = 0; kc < D / 16; ++kc)
    A::mma(x, A::template row_frag<D>(smem, r, kc, h), A::load_global(b_in + (size_t)r * D + 16 * kc + 8 * h));
  f32x16 y;
  for (int i = 0; i < 16; ++i) y[i] = 0.f;
  for (int s = 0; s < 2; ++s) A::mma(y, A::template tr_frag<D>(smem, 16 * s, 0, lane), A::pack(x, s));
  for (int i = 0; i < 16; ++i) {
    mma_out[lane * 16 + i] = x[i];
    mma_out[(64 + lane) * 16 + i] = y[i];
  }
  swap_out[lane] = xhalf_max((float)lane);
  swap_out[64 + lane] = xhalf_sum((float)lane);
}

}  // namespace fa
