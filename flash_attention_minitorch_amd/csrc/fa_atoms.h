// MFMA "atoms" for gfx950 (MI355X, CDNA4): the four primitives every attention kernel here is
// written in terms of, for bf16 (v_mfma_f32_32x32x16_bf16) and exact fp32 (v_mfma_f32_32x32x2_f32).
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
//   * LDS tile images are addressed in 16-byte chunks: off(row, ch).
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
  static constexpr int CH_ELEMS = 8;  // elements per 16-byte chunk
  template <int D> static constexpr int tile_bytes(int rows) { return rows * D * 2; }
  template <int D> static FA_DEV int off(int row, int ch) {
    return (D / 32) * 512 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
  }
  static FA_DEV frag zero() { frag z; for (int j = 0; j < 8; ++j) z[j] = (bf16_t)0.0f; return z; }
  static FA_DEV frag load_global(const bf16_t* p) { return *reinterpret_cast<const frag*>(p); }
  template <int D> static FA_DEV frag row_frag(lds_char* tile, int row, int kc, int h) {
    return *FA_LDS(frag, tile + off<D>(row, 2 * kc + h));
  }
  // element j <- tile[rowbase + 8*(j>>2) + 4*h + (j&3)][32*ct + r]
  template <int D> static FA_DEV frag tr_frag(lds_char* tile, int rowbase, int ct, int lane) {
    const int i = lane & 15, qq = i >> 2, p = i & 3, dsel = (lane >> 4) & 1, h = lane >> 5;
    const int c = 4 * ct + 2 * dsel + (p >> 1);
    const int r0 = rowbase + 4 * h + qq;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(FA_LDS(bf16x4, tile + off<D>(r0, c) + 8 * (p & 1)));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(FA_LDS(bf16x4, tile + off<D>(r0 + 8, c) + 8 * (p & 1)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  static FA_DEV frag pack(const f32x16& x, int s) {
    frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (bf16_t)x[8 * s + j];
    return f;
  }
  static FA_DEV void mma(f32x16& acc, const frag& a, const frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
};

// ---------------------------------------------------------------------------------------------
// fp32 (exact, v_mfma_f32_32x32x2_f32): row-major LDS tile padded by 16 B per row -- ds_read_b128 row
// reads and ds_read_b32 column reads are both conflict free.
// ---------------------------------------------------------------------------------------------
template <> struct Atom<float> {
  typedef f32x8 frag;
  static constexpr int ESZ = 4;
  static constexpr int CH_ELEMS = 4;
  template <int D> static constexpr int tile_bytes(int rows) { return rows * (D + 4) * 4; }
  template <int D> static FA_DEV int off(int row, int ch) { return row * (D + 4) * 4 + ch * 16; }
  static FA_DEV frag zero() { frag z; for (int j = 0; j < 8; ++j) z[j] = 0.0f; return z; }
  static FA_DEV frag load_global(const float* p) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  template <int D> static FA_DEV frag row_frag(lds_char* tile, int row, int kc, int h) {
    f32x4 a = *FA_LDS(f32x4, tile + off<D>(row, 4 * kc + 2 * h));
    f32x4 b = *FA_LDS(f32x4, tile + off<D>(row, 4 * kc + 2 * h + 1));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  template <int D> static FA_DEV frag tr_frag(lds_char* tile, int rowbase, int ct, int lane) {
    const int r = lane & 31, h = lane >> 5;
    frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = rowbase + 8 * (j >> 2) + 4 * h + (j & 3);
      f[j] = *FA_LDS(float, tile + row * (D + 4) * 4 + (32 * ct + r) * 4);
    }
    return f;
  }
  static FA_DEV frag pack(const f32x16& x, int s) {
    frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = x[8 * s + j];
    return f;
  }
  static FA_DEV void mma(f32x16& acc, const frag& a, const frag& b) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
  }
};

// ---------------------------------------------------------------------------------------------
// Register-staged global -> LDS tile copy, split into an early issue (load) and a late LDS write
// (store) so the HBM/L2 latency hides under the MFMA phase in between.  Rows >= nrows read as zero.
// ---------------------------------------------------------------------------------------------
template <typename T, int D, int ROWS, int NT> struct TileStager {
  static constexpr int CPR = D * (int)sizeof(T) / 16;   // 16-B chunks per row
  static constexpr int NCH = ROWS * CPR;
  static constexpr int PER = (NCH + NT - 1) / NT;
  u32x4 regs[PER];
  // g: pointer to row 0 of this (batch*head) matrix; row0: first row of the tile.
  FA_DEV void load(const T* g, int row0, int nrows, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * NT;
      const int row = c / CPR, ch = c % CPR;
      u32x4 v = {0u, 0u, 0u, 0u};
      if ((NCH % NT == 0 || c < NCH) && row0 + row < nrows)
        v = *reinterpret_cast<const u32x4*>(g + (size_t)(row0 + row) * D + ch * (16 / (int)sizeof(T)));
      regs[i] = v;
    }
  }
  FA_DEV void store(lds_char* tile, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * NT;
      const int row = c / CPR, ch = c % CPR;
      if (NCH % NT == 0 || c < NCH) *FA_LDS(u32x4, tile + Atom<T>::template off<D>(row, ch)) = regs[i];
    }
  }
};

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

}  // namespace fa
