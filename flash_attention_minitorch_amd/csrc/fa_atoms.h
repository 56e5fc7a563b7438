// MFMA "atoms" for gfx950 (MI355X, CDNA4): the primitives every attention kernel here is written in terms
// of, for bf16 (v_mfma_f32_32x32x16_bf16) and exact fp32 (v_mfma_f32_32x32x2_f32).
//
// Conventions (wave64, lane l: r = l & 31, h = l >> 5):
//   * A "fragment" is 8 consecutive k-elements of one operand row: k = 16*kc + 8*h + j, j = 0..7.
//     For a 32x32 output tile the A operand lane holds A[row r][k], the B operand lane holds B[k][col r].
//     bf16: one MFMA consumes a fragment pair.  fp32: eight 32x32x2 MFMAs, MFMA j taking element j of
//     both lane halves (k = 8*h + j, h = 0,1) -- the sum over k may run in any order as long as A and B agree.
//   * A 32x32 fp32 accumulator tile X has its column on the lane and its rows in the 16 registers:
//     register i of lane (r, h) is X[(i & 3) + 8*(i >> 2) + 4*h][r].
//   * pack(X, s) turns registers 8s..8s+7 into the fragment of k-chunk s for a following MFMA that sums
//     over X's ROW index; the matching "transposed" fragment of the other operand (tr_frag) holds, in
//     element j, row 16*s + 8*(j >> 2) + 4*h + (j & 3) of an LDS tile at column r.
//   * LDS tile images are addressed in 16-byte chunks: off(row, ch).  Every read address splits into a
//     per-lane part computed ONCE per kernel (RowAddr / TrAddr) and a compile-time constant that lands in
//     the ds_read instruction's offset field: no address arithmetic in the tile loops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

typedef __attribute__((address_space(3))) char lds_char;
#define FA_LDS(T, p) ((__attribute__((address_space(3))) T*)(p))
#define FA_DEV __device__ __forceinline__

// Row of accumulator register i for lane half h (see header comment).
FA_DEV constexpr int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// Both lane halves' values combined (wave64 has the two halves of a 32x32 tile's rows in lanes l, l+32).
// v_permlane32_swap exchanges vdst[32..63] with src[0..31]; called with the same value twice it returns
// {lower half broadcast, upper half broadcast}.
FA_DEV float xhalf_max(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
FA_DEV float xhalf_sum(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

struct LaneAddr { int b[2]; };  // per-lane byte offsets inside a tile image (two XOR phases for bf16)

template <typename T> struct Atom;

// ---------------------------------------------------------------------------------------------
// bf16: one LDS image serves row reads (ds_read_b128) and transposed reads (ds_read_b64_tr_b16):
// 8-row x 32-column subtiles of 512 B, 64-B rows inside a subtile, 16-B chunks XORed with (row>>2)&3.
// Both kinds of read are bank-conflict free (a half-wave's transposed read covers one 256-B bank row;
// a ds_read_b128 lane group {0-3,12-15,20-27} hits 16 distinct 16-B slots).
// ---------------------------------------------------------------------------------------------
template <> struct Atom<bf16_t> {
  typedef bf16x8 frag;
  static constexpr int ESZ = 2;
  template <int D> static constexpr int tile_bytes(int rows) { return rows * D * 2; }
  template <int D> static FA_DEV constexpr int off(int row, int ch) {
    return (D / 32) * 512 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
  }
  static FA_DEV frag zero() { frag z; for (int j = 0; j < 8; ++j) z[j] = (bf16_t)0.0f; return z; }
  static FA_DEV frag load_global(const bf16_t* p) { return *reinterpret_cast<const frag*>(p); }

  // row_frag(tile, a, row32, kc): 8 elements of row (row32 + r) at columns 16*kc + 8*h ..   (row32 % 32 == 0)
  template <int D> static FA_DEV LaneAddr row_addr(int lane) {
    const int r = lane & 31, h = lane >> 5;
    LaneAddr a;
    const int rowpart = (D / 32) * 512 * (r >> 3) + 64 * (r & 7);
    a.b[0] = rowpart + 16 * ((0 + h) ^ ((r >> 2) & 3));   // kc even: chunk & 3 = h
    a.b[1] = rowpart + 16 * ((2 + h) ^ ((r >> 2) & 3));   // kc odd:  chunk & 3 = 2 + h
    return a;
  }
  template <int D> static FA_DEV frag row_frag(lds_char* tile, const LaneAddr& a, int row32, int kc) {
    return *FA_LDS(frag, tile + a.b[kc & 1] + (D / 32) * 512 * (row32 >> 3) + 512 * (kc >> 1));
  }
  // tr_frag(tile, a, rowbase, ct): element j <- tile[rowbase + 8*(j>>2) + 4*h + (j&3)][32*ct + r]  (rowbase % 16 == 0)
  template <int D> static FA_DEV LaneAddr tr_addr(int lane) {
    const int i = lane & 15, qq = i >> 2, p = i & 3, dsel = (lane >> 4) & 1, h = lane >> 5;
    LaneAddr a;
    const int c = 2 * dsel + (p >> 1);
    a.b[0] = 64 * (4 * h + qq) + 16 * (c ^ h) + 8 * (p & 1);         // rows rowbase + 4h + qq      ((row>>2)&3 = h)
    a.b[1] = 64 * (4 * h + qq) + 16 * (c ^ (h + 2)) + 8 * (p & 1);   // rows rowbase + 8 + 4h + qq  ((row>>2)&3 = h+2)
    return a;
  }
  template <int D> static FA_DEV frag tr_frag(lds_char* tile, const LaneAddr& a, int rowbase, int ct) {
    const int k = (D / 32) * 512 * (rowbase >> 3) + 512 * ct;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(FA_LDS(bf16x4, tile + a.b[0] + k));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(FA_LDS(bf16x4, tile + a.b[1] + k + (D / 32) * 512));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  static FA_DEV frag pack(const f32x16& x, int s) {   // four v_cvt_pk_bf16_f32, no repacking
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
    u4 u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x2 pr = {x[8 * s + 2 * j], x[8 * s + 2 * j + 1]};
      u[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, bf16x2));
    }
    return __builtin_bit_cast(frag, u);
  }
  // What pack() rounded away: registers 8s..8s+7 minus their bf16 images, as a second fragment.  A product issued once with
  // pack() and once with pack_lo() carries the operand at 16 significant bits: rows with few admissible keys (the first rows
  // under the causal mask, N < 64, most keys dropped by a mask) hold P, dS of order 1 whose 2^-9 rounding is not averaged out.
  static FA_DEV frag pack_lo(const f32x16& x, int s, const frag& hi) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
    const u4 hu = __builtin_bit_cast(u4, hi);
    u4 u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float h0 = __uint_as_float(hu[j] << 16), h1 = __uint_as_float(hu[j] & 0xffff0000u);
      f32x2 pr = {x[8 * s + 2 * j] - h0, x[8 * s + 2 * j + 1] - h1};
      u[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, bf16x2));
    }
    return __builtin_bit_cast(frag, u);
  }
  static constexpr bool SPLITS = true;
  // the fragment times a scalar, re-rounded to bf16: the slot kernels fold tau*log2(e) into their lane-stationary operand once
  // per block (Q in the forward / dQ kernels, K in the dK/dV kernel), so that P = exp2(S') needs no multiply per score
  static FA_DEV frag scale(const frag& a, float c) {
    frag o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)a[j] * c);
    return o;
  }
  static FA_DEV void mma(f32x16& acc, const frag& a, const frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
  // d = a.b + c0 with c0 left intact (row constants ride in as the accumulator input)
  static FA_DEV void mma_c(f32x16& d, const frag& a, const frag& b, const f32x16& c0) {
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
  }
};

// ---------------------------------------------------------------------------------------------
// fp32 (exact, v_mfma_f32_32x32x2_f32): row-major LDS tile padded by 16 B per row -- ds_read_b128 row
// reads and ds_read_b32 column reads are both conflict free.
// ---------------------------------------------------------------------------------------------
template <> struct Atom<float> {
  typedef f32x8 frag;
  static constexpr int ESZ = 4;
  template <int D> static constexpr int tile_bytes(int rows) { return rows * (D + 4) * 4; }
  template <int D> static FA_DEV constexpr int off(int row, int ch) { return row * (D + 4) * 4 + ch * 16; }
  static FA_DEV frag zero() { frag z; for (int j = 0; j < 8; ++j) z[j] = 0.0f; return z; }
  static FA_DEV frag load_global(const float* p) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  template <int D> static FA_DEV LaneAddr row_addr(int lane) {
    const int r = lane & 31, h = lane >> 5;
    LaneAddr a;
    a.b[0] = a.b[1] = (r * (D + 4) + 8 * h) * 4;
    return a;
  }
  template <int D> static FA_DEV frag row_frag(lds_char* tile, const LaneAddr& a, int row32, int kc) {
    const int k = (row32 * (D + 4) + 16 * kc) * 4;
    f32x4 x = *FA_LDS(f32x4, tile + a.b[0] + k);
    f32x4 y = *FA_LDS(f32x4, tile + a.b[0] + k + 16);
    return __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  template <int D> static FA_DEV LaneAddr tr_addr(int lane) {
    const int r = lane & 31, h = lane >> 5;
    LaneAddr a;
    a.b[0] = a.b[1] = (4 * h * (D + 4) + r) * 4;
    return a;
  }
  template <int D> static FA_DEV frag tr_frag(lds_char* tile, const LaneAddr& a, int rowbase, int ct) {
    frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      f[j] = *FA_LDS(float, tile + a.b[0] + ((rowbase + 8 * (j >> 2) + (j & 3)) * (D + 4) + 32 * ct) * 4);
    return f;
  }
  static FA_DEV frag pack(const f32x16& x, int s) {
    frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = x[8 * s + j];
    return f;
  }
  static FA_DEV frag pack_lo(const f32x16&, int, const frag&) { return zero(); }   // pack() is exact
  static constexpr bool SPLITS = false;
  static FA_DEV frag scale(const frag& a, float c) {
    frag o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = a[j] * c;
    return o;
  }
  static FA_DEV void mma(f32x16& acc, const frag& a, const frag& b) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
  }
  static FA_DEV void mma_c(f32x16& d, const frag& a, const frag& b, const f32x16& c0) {
    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c0, 0, 0, 0);
#pragma unroll
    for (int j = 1; j < 8; ++j) d = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], d, 0, 0, 0);
  }
};

// ---------------------------------------------------------------------------------------------
// Buffer resource over one (batch*head) matrix: loads past its end return zero, so ragged tails need no
// guards, and the tile's row offset rides in the scalar soffset operand (no per-tile address VALU).
// Built from kernel arguments and blockIdx-derived scalars only, so it stays in SGPRs.
// ---------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
FA_DEV rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// LDS-DMA: 64 lanes x 16 B of global memory land lane-linearly at LDS byte address lds_dst (wave-uniform) with no
// VGPR destination.  Issued as inline asm on purpose: hipcc counts a builtin LDS-DMA as a pending LDS write and then
// waits vmcnt(0) in front of the next transposed LDS read, which would serialise the copy with the tile loop.  The
// asm form is invisible to that bookkeeping, so the ISSUER must wait (dma_wait_all) before the barrier that
// publishes the data.  M0 (the destination base) is saved and restored inside the statement.
typedef u32x4 raw_rsrc_t;
FA_DEV raw_rsrc_t make_raw_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  raw_rsrc_t r = {(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
  return r;
}
FA_DEV void dma16(raw_rsrc_t rs, uint32_t lds_dst, int voff, int soff) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff)
      : "memory");
}
// the same for one dword per lane (256 B per wave-instruction): row constants
FA_DEV void dma4(raw_rsrc_t rs, uint32_t lds_dst, int voff, int soff) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff)
      : "memory");
}
FA_DEV void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Register-staged global -> LDS tile copy of ROWS x D elements by NT threads, split into an early issue (load)
// and a late LDS write (store) so the HBM/L2 latency hides under the MFMA phase in between.
template <typename T, int D, int ROWS, int NT> struct TileStager {
  static constexpr int CPR = D * (int)sizeof(T) / 16;   // 16-B chunks per row
  static constexpr int NCH = ROWS * CPR;
  static constexpr int PER = (NCH + NT - 1) / NT;
  static constexpr int RSTEP = NT / CPR;                // rows between a thread's consecutive chunks
  static_assert(NT % CPR == 0, "threads must tile whole rows");
  static_assert(sizeof(T) == 4 || RSTEP % 16 == 0 || PER == 1, "row step must keep the swizzle phase");
  u32x4 regs[PER];
  int voff;       // byte offset of this thread's first chunk inside the global tile
  int lds_off;    // byte offset of this thread's first chunk inside the LDS image
  int ldb;        // bytes between consecutive rows in global memory (row stride: D elements, or H*D for [B][N][H][d])
  bool live;      // this thread has a chunk at all (NCH < NT)
  FA_DEV void init(int tid, int ld) {
    ldb = ld * (int)sizeof(T);
    int row = tid / CPR, ch = tid % CPR;
    if constexpr (sizeof(T) == 2 && CPR >= 8) {
      // bf16 image: a ds_write_b128 lane group (8 consecutive lanes) must cover 128 distinct bytes mod 128:
      // two adjacent rows x four chunks of one 32-column subtile (rows are 64 B apart inside a subtile).
      const int g = tid >> 3;
      row = 2 * (g / (CPR / 4)) + ((tid >> 2) & 1);
      ch = 4 * (g % (CPR / 4)) + (tid & 3);
    }
    voff = row * ldb + ch * 16;
    lds_off = Atom<T>::template off<D>(row, ch);
    live = (NCH >= NT) || tid < NCH;
  }
  // row0 (wave-uniform): first row of the tile inside the matrix the resource covers
  FA_DEV void load(rsrc_t rs, int row0) {
    const int soff = row0 * ldb;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (NCH % NT == 0 || live)
        v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                          rs, voff + i * RSTEP * ldb, soff, 0));
      regs[i] = v;
    }
  }
  FA_DEV void store(lds_char* tile) const {
#pragma unroll
    for (int i = 0; i < PER; ++i)
      if (NCH % NT == 0 || live)
        *FA_LDS(u32x4, tile + lds_off + Atom<T>::template off<D>(i * RSTEP, 0)) = regs[i];
  }
};

// Where one (batch*head) matrix lives: element (n, :) of head bh is at head_base(bh) + n * ld.
// [BH][N][d] (the reference layout): H = 1, ld = d, bstride = N*d, hstride = 0.
// [B][N][H][d] (what the projection writes before minitorch's permute + contiguous, modules_transfomer.py:67-89):
// H = heads, ld = H*d, bstride = N*H*d, hstride = d.
struct Layout {
  int H;
  int ld;
  long bstride;
  long hstride;
  // optional additive key mask [mask_batch][N] (fp32, in units of the SCALED score tau*q.k: 0 keeps a key, -inf drops
  // it; the reference's fused softmax takes the same [batch, to_len] mask, src/softmax_kernel.cu:27-34,77-90);
  // batch*head bh reads row bh / mask_heads.  nullptr: no mask.
  const float* kmask;
  int mask_heads;
  // optional dropout on the attention probabilities (see drop_keep): drop_thr = 0 disables it
  uint32_t drop_thr;     // a position is kept iff its 24-bit uniform r24 >= drop_thr  (drop_thr = floor(rate * 2^24))
  float drop_scale;      // kept probabilities are multiplied by this (1 = minitorch's nn.dropout, 1/(1-rate) = inverted)
  uint32_t drop_seed;
  int young_prio;        // slot kernels: waves 4-7 of a workgroup run at s_setprio 1 (0 = off); scheduling only
  int rank_chunk;        // causal slot builds: heads per XCD whose blocks are dispatched together, longest first (map_block_ranked)
  int tiles;             // tiled dK/dV build: consecutive heads per workgroup
  // Scale guard (round 4; fa_common.h: guard_skip): partial maxima of the squared row norms of q (first GUARD_SLOTS floats) and
  // k (next GUARD_SLOTS), written by scale_guard_kernel.  A guarded call launches the kernels that carry tau*log2(e) in a bf16
  // operand with guard_want = 0 and their fp32-scaling twins with guard_want = 1; every workgroup evaluates the same predicate on
  // entry and the ones of the launch that was not chosen return at once.  nullptr: no guard (the launch always runs).
  const float* guard;
  float guard_coef;      // the operand-rounding estimate in units of its budget is guard_coef * sqrt(max |q|^2 * max |k|^2)
  int guard_want;
  int out_bf16;          // forward kernels: `o` points to bf16 elements (same shape and row stride), rounded once from the fp32 result
  // Backward MFMA-slot kernels (bwd_dq_slot_kernel's mask-free builds, bwd_dkdv_slot_kernel): both scalings live in ONE launch, chosen
  // per launch by a wave-uniform branch around two copies of the sweep: 0 = tau*log2(e) folded into the bf16 operand (P = exp2(S')),
  // 1 = the operand stays unscaled and every score is multiplied in fp32 (P = exp2(c * S'), the reference's arithmetic), 2 = by the
  // scale guard (fa_common.h: scale_exact).  The forward keeps the pair-of-launches form (its fp32-scaling twin is another kernel).
  int scale_sel;
  int twin_blocks;       // phased forward launched as the fp32-scaling twin of a guarded call: consecutive query blocks per workgroup
};

// Counter-based dropout bit of attention position (batch*head bh, query q, key k): a 32-bit finaliser (two
// multiply / xor-shift rounds) of a seed-offset linear index -- stateless, so the forward, the dK/dV kernel and the dQ
// kernel regenerate the same mask from their own (register, lane) -> (q, k) maps, and oracle/attention_ref.py restates
// it in NumPy bit for bit.  minitorch's dropout keeps a position iff rate < r, r uniform in [0,1) (minitorch/nn.py:168-186).
FA_DEV uint32_t drop_base(const Layout& L, int bh, int q) {
  return L.drop_seed + (uint32_t)bh * 0xC2B2AE3Du + (uint32_t)q * 0x9E3779B1u;
}
FA_DEV bool drop_keep(uint32_t base_bh_q, int k, uint32_t thr) {
  uint32_t a = base_bh_q + (uint32_t)k * 0x85EBCA77u;
  a ^= a >> 16;
  a *= 0x7FEB352Du;
  a ^= a >> 15;
  a *= 0x846CA68Bu;
  a ^= a >> 16;
  return (a >> 8) >= thr;
}
FA_DEV size_t head_base(const Layout& L, int bh) {
  return (size_t)(bh / L.H) * (size_t)L.bstride + (size_t)(bh % L.H) * (size_t)L.hstride;
}

// Workgroup id -> (batch*head, block) with every block of one (batch*head) on the same XCD
// (blocks b and b+8 share an XCD's L2 under round-robin dispatch; speed only, never correctness).
FA_DEV void map_block(int id, int BH, int nb, int& bh, int& b) {
  if ((BH & 7) == 0) {
    const int xcd = id & 7, slot = id >> 3, per = BH >> 3;
    bh = xcd * per + slot / nb;
    b = slot % nb;
  } else {
    bh = id / nb;
    b = id % nb;
  }
}

// Causal launches: block rank b = 0 is the heaviest block of every (batch*head) (the caller maps rank -> block).  The heads of an
// XCD are taken in chunks of C; within a chunk all blocks of rank 0 are dispatched first, then rank 1, ...: longest first, so the
// dynamic assignment of workgroups to CUs ends level (the work per rank falls linearly: a chunk of two rounds of the chip pairs
// rank r with rank nb-1-r by itself), while only C heads per XCD stream their K / V (Q / dO) through its L2 at a time.
FA_DEV void map_block_ranked(int id, int BH, int nb, int C, int& bh, int& b) {
  if ((BH & 7) == 0) {
    const int xcd = id & 7, slot = id >> 3, per = BH >> 3;
    const int chunk = slot / (C * nb), within = slot - chunk * (C * nb);
    const int cper = min(C, per - chunk * C);   // (the last chunk of an XCD may be smaller)
    bh = xcd * per + chunk * C + within % cper;
    b = within / cper;
  } else {
    bh = id % BH;
    b = id / BH;
  }
}

}  // namespace fa
