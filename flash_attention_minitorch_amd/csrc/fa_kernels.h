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
// Scaling (round 3): the MFMA-slot kernels fold tau*log2(e) into their lane-stationary bf16 operand once per block (Q in the forward
// and dQ kernels, K in the dK/dV kernel), so P = exp2(S') costs one VALU instruction per score; rows with fewer than 64 admissible
// keys and every phased kernel scale in fp32.  The slot forward sweeps without a softmax reference and redoes a wave's rows in a
// wave-local cold path when exp2 left its range (fwd_redo_rows).
//
// Backward = preprocess (delta = rowsum(dO*O), -L/tau, -L*log2e) + a key-stationary dK/dV kernel (S, dP with the KEY on
// the lane; P and dS feed dV^T += dO^T P and dK^T += Q^T dS from registers) + a query-stationary dQ kernel
// (S^T, dP^T with the query on the lane; dQ^T += K^T dS^T).  No atomics: results are bitwise reproducible
// (the reference's FA-2 backward uses atomicAdd for dQ, src/flash_attn2_bw.cu:228).
//
// Tile loops are unrolled by two so the LDS double-buffer index is a compile-time constant: every LDS address is
// a per-lane register computed once plus an instruction immediate, and every global tile load is a buffer load
// whose tile offset is a scalar operand (out-of-range rows read as zero) -- no address VALU inside the loops.
//
// Files: fa_atoms.h (MFMA / LDS / LDS-DMA primitives), fa_common.h (constants, phase stamps, small helpers), fa_fwd.h (forward),
// fa_bwd_dkdv.h (preprocess, dK / dV), fa_bwd_dq.h (dQ), fa_aux.h (sustained-peak loop, layout probes).
#pragma once
#include "fa_common.h"
#include "fa_fwd.h"
#include "fa_fwd_splitk_f32.h"
#include "fa_bwd_dkdv.h"
#include "fa_bwd_dq.h"
#include "fa_bwd_onepass_f32.h"
#ifdef FA_DIAG
#include "fa_bwd_fused.h"   // the one-pass backward: diagnostic build only (tools/check_fused.py)
#include "fa_bwd_chain.h"   // its round-4 form (chained key blocks, fp32 atomics from the last one): tools/check_chain.py
#endif
#include "fa_aux.h"
