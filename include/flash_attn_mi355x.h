/*
 * flash_attn_mi355x.h -- C ABI of the MI355X (gfx950) FlashAttention forward/backward library.
 *
 * Two groups of entry points:
 *
 *  (1) The reference's own FFI, unchanged.  Each of the six shared objects the reference opens with
 *      ctypes.CDLL at import time (minitorch/cuda_kernel_ops.py:30-35)
 *          flash_attn_fw.so        flash_attn_bw.so          FA-1            (Makefile:28-34)
 *          flash_attn_causal_fw.so flash_attn_causal_bw.so   FA-1 + causal block skipping (Makefile:36-42)
 *          flash_attn2_fw.so       flash_attn2_bw.so         FA-2            (Makefile:44-50)
 *      exports ONE unmangled symbol, launch_flash_attn_fw or launch_flash_attn_bw, with the reference's
 *      signature (src/flash_attn_fw.cu:300-312, src/flash_attn_bw.cu:275-291; identical in the FA-2
 *      files src/flash_attn2_fw.cu:310-322, src/flash_attn2_bw.cu:277-293).  Host fp32 pointers,
 *      synchronous, no return code, message on stderr + exit(EXIT_FAILURE) on a GPU error
 *      (src/flash_attn_fw.cu:343-349).  The shims forward to fa_mi355x_launch_fw_host / _bw_host below.
 *
 *  (2) Additive device-pointer entry points (libflash_attn_mi355x.so): asynchronous on the caller's
 *      stream, int status, no allocation inside the call.  These are what bench.py times and what the
 *      multi-GPU shard uses; the reference has no counterpart (its launchers malloc/copy/free per call,
 *      src/flash_attn_fw.cu:314-357).
 *
 * Layout everywhere: row-major contiguous [batch][N][d] for q, k, v, out, out_grad, q_grad, k_grad,
 * v_grad and [batch][N] for l, m, where batch = B*H (minitorch/cuda_kernel_ops.py:542-547,570).
 */
#ifndef FLASH_ATTN_MI355X_H
#define FLASH_ATTN_MI355X_H

#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- (1) reference FFI: one of each per variant library ------------------------------------ */

/* Replaces launch_flash_attn_fw of src/flash_attn_fw.cu:302-312 (FA-1; l = sum exp(s-m), m = row max)
 * and of src/flash_attn2_fw.cu:312-322 (FA-2; l = logsumexp, m left as passed in).
 * q,k,v,out: host float[batch*N*d]; l,m: host float[batch*N].  The caller pre-initialises out=0, l=0,
 * m=-FLT_MAX (minitorch/cuda_kernel_ops.py:537-539); the values are not read.  stream: hipStream_t
 * (torch.cuda.current_stream().cuda_stream on PyTorch-ROCm) or NULL. */
void launch_flash_attn_fw(float* q, float* k, float* v, float* out, float* l, float* m,
                          int batch, int N, int d, bool causal_mask, void* stream);

/* Replaces launch_flash_attn_bw of src/flash_attn_bw.cu:277-291 / src/flash_attn2_bw.cu:279-293.
 * q_grad, k_grad, v_grad are overwritten (the caller passes zeros, minitorch/cuda_kernel_ops.py:609-611).
 * l, m: the side outputs of the SAME variant's forward (minitorch/tensor_functions.py:462-497). */
void launch_flash_attn_bw(float* q, float* k, float* v, float* out, float* out_grad,
                          float* q_grad, float* k_grad, float* v_grad, float* l, float* m,
                          int batch, int N, int d, bool causal_mask, void* stream);

/* ---- (2) core library ------------------------------------------------------------------------ */

/* Side-output convention of a variant library. */
#define FA_VARIANT_FA1 1        /* flash_attn_{fw,bw}.so, flash_attn_causal_{fw,bw}.so */
#define FA_VARIANT_FA2 2        /* flash_attn2_{fw,bw}.so */

/* Element type of q, k, v, out_grad on the device-pointer path (outputs are always fp32). */
#define FA_DTYPE_F32  0         /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32) */
#define FA_DTYPE_BF16 1         /* bf16 MFMA (v_mfma_f32_32x32x16_bf16), fp32 accumulate / softmax state */

/* Tensor layouts of the *_layout / *_scaled / *_guarded / *_masked / *_dropout entry points (see fa_mi355x_fwd_layout). */
#define FA_LAYOUT_BHND 0        /* [B][H][N][d]: the reference's contiguous (batch*head, N, d) */
#define FA_LAYOUT_BNHD 1        /* [B][N][H][d]: what minitorch's projection writes before its permute + contiguous */

/* Status codes of the int-returning entry points. */
#define FA_OK 0
#define FA_ERR_BAD_ARG 1        /* null pointer, non-positive size, unknown variant/dtype */
#define FA_ERR_UNSUPPORTED_D 2  /* row length not in {32, 64, 128}: other d <= 128 go through fa_mi355x_*_padded (device) or the host launchers */
#define FA_ERR_HIP 3            /* a HIP call failed: see fa_mi355x_last_error() */

/* Host-pointer launchers behind the six shims (same argument meaning as the reference FFI + variant). */
void fa_mi355x_launch_fw_host(int variant, float* q, float* k, float* v, float* out, float* l, float* m,
                              int batch, int N, int d, bool causal_mask, void* stream);
void fa_mi355x_launch_bw_host(int variant, float* q, float* k, float* v, float* out, float* out_grad,
                              float* q_grad, float* k_grad, float* v_grad, float* l, float* m,
                              int batch, int N, int d, bool causal_mask, void* stream);

/* The host launchers copy the caller's arrays as pageable memory (the reference's cudaMemcpy calls do the same).  With FA_MI355X_HOST_PIN=1
 * in the environment they pin them in place for the duration of a call instead (page-aligned, merged ranges of at least 4 MiB;
 * hipHostRegister): faster only for backward calls of hundreds of MiB, see fa_api.hip (PinSet).  Cumulative, process-wide: how many such
 * ranges were registered, and how many could not be (already registered by the caller, or not lockable) and were copied pageable
 * instead -- same results.  Both stay 0 without the opt-in.  Either pointer may be NULL. */
void fa_mi355x_host_pin_stats(unsigned long long* pinned_ranges, unsigned long long* pageable_ranges);

/* Forward on device pointers.  q,k,v: dtype elements [batch][N][d]; out: float [batch][N][d];
 * l, m: float [batch][N] (m may be NULL for FA_VARIANT_FA2).  Asynchronous on `stream`. */
int fa_mi355x_fwd(const void* q, const void* k, const void* v, float* out, float* l, float* m,
                  int batch, int N, int d, int causal, int variant, int dtype, void* stream);

/* Bytes of scratch fa_mi355x_bwd needs: 3 * batch * N floats (-L/tau, -rowsum(dO*O), -L*log2(e)).  Every backward entry point
 * below expects a workspace of at least this size.  (The diagnostic library adds the hand-off region of its round-2 one-pass
 * backward for d = 64, N a multiple of 256; the product library has no kernel that needs more.) */
size_t fa_mi355x_bwd_workspace_bytes(int batch, int N, int d);

/* The same for a backward call with per-call options (fa_mi355x_bwd_ex).  Product library: fa_mi355x_bwd_workspace_bytes.  Diagnostic
 * library: opts[4] = 3 (the chained one-pass backward) adds a 4-KiB header and, when a chain is more than one key block (nchains <
 * N / 256), one slab of N * 64 floats per workgroup: batch * nchains * N * 256 bytes with nchains = the smallest divisor of N / 256
 * that gives batch * nchains >= CUs (256 MiB at batch 64, N 4096 on 256 CUs). */
size_t fa_mi355x_bwd_workspace_bytes_ex(int batch, int N, int d, const int* opts, int nopts);

/* Product library: a no-op that sets *status = 0 (no kernel of it waits for another workgroup, and backward calls may run
 * concurrently on different streams).  Diagnostic library: the error word of the round-2 one-pass backward (a persistent grid with a
 * bounded-spin hand-off, opts[4] = 2) in a workspace the last backward call used; FA_ERR_HIP and a message when a wait timed out. */
int fa_mi355x_bwd_status(const void* workspace, int batch, int N, int d, int* status);

/* Backward on device pointers.  out: float (the forward's output); out_grad: dtype elements;
 * q_grad,k_grad,v_grad: float, overwritten; workspace: device scratch of the size above. */
int fa_mi355x_bwd(const void* q, const void* k, const void* v, const float* out, const void* out_grad,
                  float* q_grad, float* k_grad, float* v_grad, const float* l, const float* m,
                  void* workspace, int batch, int N, int d, int causal, int variant, int dtype, void* stream);

/* The same, restricted to some of its stages (bench.py times them one by one).  dK/dV and dQ need the workspace contents that the
 * PREP stage leaves (-L/tau and -rowsum(dO*O) per row).  A call that asks for PREP and DQ together runs them as ONE launch wherever a
 * plain dQ kernel is selected (no key mask / dropout, bf16 with N >= 64, ...): the dQ kernel preprocesses its own rows, writes the
 * workspace and runs first, dK/dV (if asked for) after it; otherwise the order is prep kernel -> dK/dV -> dQ.  The workspace
 * contents are the same either way (up to the summation order of delta). */
#define FA_BWD_STAGE_PREP 1     /* workspace <- -L/tau, -rowsum(dO*O) */
#define FA_BWD_STAGE_DKDV 2     /* k_grad, v_grad */
#define FA_BWD_STAGE_DQ   4     /* q_grad */
#define FA_BWD_STAGE_ALL  7
int fa_mi355x_bwd_stages(const void* q, const void* k, const void* v, const float* out, const void* out_grad,
                         float* q_grad, float* k_grad, float* v_grad, const float* l, const float* m,
                         void* workspace, int batch, int N, int d, int causal, int variant, int dtype, int stages,
                         void* stream);

/* Forward / backward with per-call kernel options (no process-wide state): opts[0..nopts-1], nopts <= 10, 0 = default.
 *   opts[0]  dK/dV kernel: 3 = (d = 64) the phased kernel with the slot path on unmasked stages (what causal launches that do not
 *            fill the chip run anyway); 4 = the compiler-interleaved phased kernel (fp32 scaling: OPTS_EXACT_SCALE); 5 = at d = 64, causal,
 *            N % 256 == 0: the causal build of the continuous slot pipeline whatever the launch size
 *   opts[1]  forward kernel: 2 = phased (fp32 scaling), 3 = slot kernel also under the causal mask (whatever the launch size);
 *            fp32, d = 64: 4 = the split-key forward whatever the launch size (a workgroup = one 32-query block, its four waves a quarter
 *            of the keys each, partial (O, l, m) combined through LDS: the default of launches that would leave most of the chip idle,
 *            up to 128 workgroups of the phased kernel, 256 under the causal mask), 2 = never
 *   opts[2]  dQ kernel: 2 = phased with 32-key tiles (fp32 scaling), 3 = slot kernel also under the causal mask
 *   opts[3]  (diagnostic library only)
 *   opts[4]  1 = keep the separate preprocess kernel (default: the dQ launch preprocesses its own rows, writes the workspace and runs
 *            BEFORE the dK/dV launch; same results up to summation order of delta);
 *            4 / 5 = fp32, d = 64: two kernels always / the ONE-PASS backward whatever the launch size.  By default an fp32, d = 64
 *            backward with N >= 256 and no key mask / dropout, asked for dQ and dK/dV together, runs bwd_onepass_f32_kernel when its
 *            launch -- batch * ceil(N / 256) workgroups, one per CU; below that the query sweep of every key block is cut into 2, 4
 *            or 8 workgroups, which then add their dK, dV as well -- runs in rounds that are at least 80 % full: the five products of
 *            src/flash_attn2_bw.cu:94-247 in one key-stationary pass, dQ added to the (library zero-filled) q_grad with fp32 atomics as
 *            the reference does at :228 -- 131 vs 98.5 TFLOP/s at BASELINE configs[2] because the exact-fp32 MFMA bounds it, not the
 *            atomics (B = 1, H = 8, N = 1024: 0.085 vs 0.240 ms in 8 parts; launches that stay under 80 % keep the two kernels).  dq then
 *            differs from run to run in the last bits (order of the N/256 adds per element; dk, dv are bitwise stable unless the sweeps
 *            are cut into 4 or 8 parts); 4 restores
 *            the bitwise repeatable two-kernel path (FA_MI355X_DETERMINISTIC=1 in the environment does the same for every call
 *            that does not ask for 5: the reference's launch_flash_attn_bw has no options argument);
 *   opts[5]  1 = the non-causal d = 64 dK/dV kernel takes one head per workgroup (default: key block kb of several consecutive heads
 *            per workgroup when the launch still covers every CU), and so does the non-causal d = 64 dQ kernel (default: query block
 *            qb of several consecutive heads, same condition); bitwise the same results
 *   opts[6]  (diagnostic library only)
 *   opts[7]  block order of causal launches: 1 = query blocks p and nqb-1-p paired in one workgroup (slot and phased forward / dQ
 *            kernels) and head-by-head order for the unpaired dK/dV launches (fp32 d = 64, bf16 d = 128); 2 = one block per
 *            workgroup dispatched longest first across a chunk of heads, everywhere; 0 = per kernel what measured faster (slot
 *            builds: ranked below 8 rounds of the chip; phased forward / dQ: paired; unpaired dK/dV: ranked)
 *   opts[8]  where the softmax scale is applied (bf16, d = 64 / 128; "Softmax scale and the scale guard" below): 0 = fp32 scaling of
 *            every score, as the reference does, unless the call carries a scale guard (fa_mi355x_*_guarded); 1 = the caller vouches
 *            that q and k are of the north star's U(-1, 1) magnitude: the MFMA-slot kernels that fold tau*log2(e) into a bf16
 *            operand run unguarded (8-10 % faster; at x2 inputs their error on O doubles to 1.4e-3); 2 = fp32 scaling whatever a
 *            guard says; 3 = fa_mi355x_plan only: plan a guarded call (both launches of every pair)
 *   opts[9]  forward only: 1 = `out` points to BF16 elements of the same shape and row stride (one rounding of the fp32 result, 2^-9
 *            relative: |error| <= 2e-3 |o|, i.e. above the 1e-3 parity bound once |o| > 0.5; fp32 stays the default and the parity
 *            path).  For consumers that take a bf16 activation, e.g. the sharded gather of BASELINE configs[4] at half the bytes.  The
 *            backward needs the fp32 `out` of a default forward.
 * Values that lost their A/B (opts[0] = 1 / 2, opts[1] = 6, opts[2] = 1 / 4, opts[3] = 1, opts[4] = 2 / 3 = the one-pass
 * backwards, opts[6] = 1) exist in the diagnostic library only; the product library answers them with FA_ERR_BAD_ARG.
 * Every value selects kernels with the same results within the stated tolerances; stamp / ablation builds are not in this
 * library (FA_ERR_BAD_ARG).  `stages` as fa_mi355x_bwd_stages. */
int fa_mi355x_fwd_ex(const void* q, const void* k, const void* v, float* out, float* l, float* m, int batch, int N, int d,
                     int causal, int variant, int dtype, const int* opts, int nopts, void* stream);
int fa_mi355x_bwd_ex(const void* q, const void* k, const void* v, const float* out, const void* out_grad, float* q_grad,
                     float* k_grad, float* v_grad, const float* l, const float* m, void* workspace, int batch, int N, int d,
                     int causal, int variant, int dtype, int stages, const int* opts, int nopts, void* stream);

/* ---- Softmax scale and the scale guard (round 4) ----
 * The reference multiplies every score by tau = sqrt(1/d) in fp32 (src/flash_attn2_fw.cu:152-167).  The fastest bf16 kernels here
 * (MFMA-slot pipelines, d = 64 / 128) instead carry tau*log2(e) inside one bf16 MFMA operand, re-rounded once: exp2 then needs no
 * multiply per score (8-10 % of the step), at the price of one more 2^-9 relative rounding of q (or k).  That is invisible for inputs of
 * the reference tests' U(-1, 1) magnitude and grows with the square of the input magnitude, so those kernels run only on evidence:
 *   fa_mi355x_scale_guard   one pass over q and k (HBM-bound: 12 us at B=8, H=8, N=4096, d=64): the largest squared row norms, as
 *                           fa_mi355x_guard_bytes() bytes of device memory; rows = the number of rows of row_elems contiguous
 *                           elements (B*H*N rows of d in either layout; padded rows: dp).  fp32 or other row lengths: zero-filled.
 *   fa_mi355x_*_guarded     launch the selected kernels AND their fp32-scaling twins; every workgroup evaluates the guard on entry
 *                           (estimate 2^-9/sqrt(3) * tau*log2(e) * max|q| * max|k| against a budget of 1e-2 in log2 units: U(-1, 1)
 *                           gives 5.7e-3 at d = 64, inputs 1.3x larger and up take fp32 scaling) and the launch on the wrong side
 *                           returns at once: no host synchronisation, about 4 us per skipped launch.  guard = NULL, or any other
 *                           entry point of this header: fp32 scaling (opts[8] = 1 overrides).  The forward and the backward of one
 *                           q / k pair take the same guard (compute it once).  softmax_scale = 0: sqrt(1/d).
 *   produce_guard = 1       (forward only, opts[8] = 0) the call FILLS `guard` instead of reading it, for the backward of the same
 *                           (q, k): a forward that folds forms the row norms inside its own launch (every wave holds its 32 query
 *                           rows anyway and loads the 32 key rows of the same indices; atomic maxima into the zero-filled guard) and
 *                           runs optimistically; its fp32-scaling twin, launched behind it, redoes the call if the finished guard says
 *                           so.  No separate pass over q and k: the forward costs one 2-KiB memset and the twin's empty launch.  A
 *                           forward that folds nothing runs fa_mi355x_scale_guard itself (the backward of the call may fold).
 * A caller that folds log2(e)/sqrt(d) into its query projection and passes softmax_scale = ln(2) needs no guard: the factor is 1. */
size_t fa_mi355x_guard_bytes(void);
int fa_mi355x_scale_guard(const void* q, const void* k, long rows, int row_elems, int dtype, void* guard, void* stream);
int fa_mi355x_fwd_guarded(const void* q, const void* k, const void* v, float* out, float* l, float* m, int B, int H, int N, int d,
                          int layout, float softmax_scale, int causal, int variant, int dtype, const int* opts, int nopts,
                          void* guard, int produce_guard, void* stream);
int fa_mi355x_bwd_guarded(const void* q, const void* k, const void* v, const float* out, const void* out_grad, float* q_grad,
                          float* k_grad, float* v_grad, const float* l, const float* m, void* workspace, int B, int H, int N, int d,
                          int layout, float softmax_scale, int causal, int variant, int dtype, int stages, const int* opts,
                          int nopts, const void* guard, void* stream);

/* The same two operations with the caller's softmax scale instead of sqrt(1/d): P = softmax_k(softmax_scale * q.k).  The reference's
 * operator has no such argument (tau = sqrt(1/d) is fixed, src/flash_attn_fw.cu:37); it is here for callers that fold the scale into
 * their query projection, the usual arrangement in fused-attention stacks: with q' = (log2(e)/sqrt(d)) * q formed in fp32 BEFORE the
 * rounding to bf16 (i.e. folded into the projection's weights) and softmax_scale = ln(2), the bf16 d = 64 / 128 default kernels' folded
 * scale tau*log2(e) is exactly 1 and costs no rounding at all (include/flash_attn_mi355x.h "softmax scale").
 * q_grad is the gradient with respect to the q that was passed in.  Layouts and the other arguments as fa_mi355x_*_layout. */
int fa_mi355x_fwd_scaled(const void* q, const void* k, const void* v, float* out, float* l, float* m, int B, int H, int N, int d,
                         int layout, float softmax_scale, int causal, int variant, int dtype, void* stream);
int fa_mi355x_bwd_scaled(const void* q, const void* k, const void* v, const float* out, const void* out_grad, float* q_grad,
                         float* k_grad, float* v_grad, const float* l, const float* m, void* workspace, int B, int H, int N, int d,
                         int layout, float softmax_scale, int causal, int variant, int dtype, void* stream);

/* Any head dim d <= 128 on the device path (the reference operator accepts any d up to its assert, src/flash_attn_fw.cu:43; the host
 * launchers above pad on the fly): q, k, v, out_grad, out and the gradients are [batch][N][dp] with dp in {32, 64, 128}, dp >= d, and
 * columns d .. dp-1 of q, k, v, out_grad ZERO (zero columns of q / k add nothing to the scores, zero columns of v / out_grad give zero
 * columns of out / the gradients); tau = 1/sqrt(d) uses the caller's d.  device_ops pads and slices with torch. */
int fa_mi355x_fwd_padded(const void* q, const void* k, const void* v, float* out, float* l, float* m, int batch, int N, int d,
                         int dp, int causal, int variant, int dtype, void* stream);
int fa_mi355x_bwd_padded(const void* q, const void* k, const void* v, const float* out, const void* out_grad, float* q_grad,
                         float* k_grad, float* v_grad, const float* l, const float* m, void* workspace, int batch, int N, int d,
                         int dp, int causal, int variant, int dtype, void* stream);

/* Which kernels would a call launch, in order?  Runs the library's own dispatch code with the launches skipped (no HIP call, works
 * without a GPU except for launch-size rules that read the CU count: 256 is assumed then) and writes the kernel names, separated by
 * ';', to out[0..n-1] (NUL terminated), e.g. "bwd_dq_slot_kernel;bwd_dkdv_slot_kernel".  stages = 0: the forward (fa_mi355x_fwd_ex);
 * otherwise the backward stage mask of fa_mi355x_bwd_ex.  A backward plan without "bwd_prep_kernel" means the dQ launch does the
 * preprocess (and therefore runs first).  bench.py labels its per-kernel timings and its roofline from this. */
int fa_mi355x_plan(int batch, int N, int d, int causal, int variant, int dtype, int stages, const int* opts, int nopts, char* out,
                   size_t n);

/* The same two operations on the layout the projection writes: SURVEY.md row f1.  minitorch's MultiHeadAttention
 * produces q, k, v as (B, N, H, d) views and then pays permute(0,2,1,3).contiguous() for each of them, and the
 * inverse for the output (minitorch/modules_transfomer.py:67-89,137-139): four full-tensor copies per layer.
 * FA_LAYOUT_BNHD reads and writes [B][N][H][d] directly (element (b,n,h,:) at ((b*N + n)*H + h)*d);
 * l, m and the workspace stay [B][H][N]. */
int fa_mi355x_fwd_layout(const void* q, const void* k, const void* v, float* out, float* l, float* m, int B, int H,
                         int N, int d, int layout, int causal, int variant, int dtype, void* stream);
int fa_mi355x_bwd_layout(const void* q, const void* k, const void* v, const float* out, const void* out_grad,
                         float* q_grad, float* k_grad, float* v_grad, const float* l, const float* m,
                         void* workspace, int B, int H, int N, int d, int layout, int causal, int variant, int dtype,
                         void* stream);

/* The same two operations with an additive KEY mask: SURVEY.md row f4.  The reference's flash path has no mask
 * argument (its tests multiply by an all-ones matrix, kernel_tests/test_flashattn_fw.py:66,71); its fused softmax
 * does: attn_mask [batch, to_len], 0 for tokens and -inf for padding, added to the scaled score before the row maximum
 * (src/softmax_kernel.cu:27-34,77-90; minitorch/modules_transfomer.py:131-136).  key_mask: device float [B][N] in those
 * units (any finite value or -inf), shared by the H heads of a batch element, combined with `causal`; NULL = no mask.
 * P = softmax_k(tau * q.k + key_mask[b][k]).  A row whose every admissible key is masked returns out = 0, L = -inf
 * (FA-1: l = 0, m = -inf) and contributes zero gradients.  Layout and the other arguments as fa_mi355x_*_layout. */
int fa_mi355x_fwd_masked(const void* q, const void* k, const void* v, float* out, float* l, float* m,
                         const float* key_mask, int B, int H, int N, int d, int layout, int causal, int variant,
                         int dtype, void* stream);
int fa_mi355x_bwd_masked(const void* q, const void* k, const void* v, const float* out, const void* out_grad,
                         float* q_grad, float* k_grad, float* v_grad, const float* l, const float* m,
                         const float* key_mask, void* workspace, int B, int H, int N, int d, int layout, int causal,
                         int variant, int dtype, void* stream);

/* ... and with dropout on the attention probabilities (the other half of SURVEY.md row f4):
 *   out = scale * (M o softmax(tau*q.k + key_mask)) v,   M[b,h,q,k] = 1 iff r24(seed, b*H+h, q, k) >= floor(rate * 2^24)
 * minitorch's dropout keeps a position iff rate < r and does NOT rescale (minitorch/nn.py:168-186): scale = 1 is that
 * convention, scale = 1/(1-rate) the inverted one; the reference's own vanilla-attention test multiplies the
 * probabilities by such a 0/1 matrix (kernel_tests/test_flashattn_fw.py:66,71).  r24 is the top 24 bits of a stateless
 * 32-bit hash of (seed, batch*head, query, key) (csrc/fa_atoms.h drop_keep; oracle/attention_ref.py restates it), so the
 * backward regenerates the same mask: call it with the rate, scale and seed of the forward.  l (and m) are the softmax
 * statistics BEFORE dropout.  rate = 0 is fa_mi355x_*_masked.  key_mask may be NULL. */
int fa_mi355x_fwd_dropout(const void* q, const void* k, const void* v, float* out, float* l, float* m,
                          const float* key_mask, float rate, float scale, unsigned seed, int B, int H, int N, int d,
                          int layout, int causal, int variant, int dtype, void* stream);
int fa_mi355x_bwd_dropout(const void* q, const void* k, const void* v, const float* out, const void* out_grad,
                          float* q_grad, float* k_grad, float* v_grad, const float* l, const float* m,
                          const float* key_mask, float rate, float scale, unsigned seed, void* workspace, int B, int H,
                          int N, int d, int layout, int causal, int variant, int dtype, void* stream);

/* Message of the last FA_ERR_* on this thread ("" if none). */
const char* fa_mi355x_last_error(void);

/* Library version, e.g. "flash_attn_mi355x 0.1 gfx950". */
const char* fa_mi355x_version(void);

/* ---- diagnostic build only (libflash_attn_mi355x_diag.so, compiled with -DFA_DIAG; used by tools/, never by the product
 * path or the tests) ----
 * fa_mi355x_set_tuning: process-wide defaults for the option slots of fa_mi355x_*_ex, plus the values the product library
 * rejects: stamp builds (opts[0] = 9 / 93 / 193, opts[1] = 93, opts[2] = 93), the register-staging A/B build (opts[0] = 13), the
 * barrier-less dQ timing ablation (opts[2] = 94, WRONG results) and opts[5] = timing ablations / stamps of the one-pass backward.
 * fa_mi355x_debug_phase_cycles: copies the first n per-wave phase counters (8 per wave slot) a stamp build wrote. */
#ifdef FA_DIAG
int fa_mi355x_set_tuning(int key, int value);
int fa_mi355x_debug_phase_cycles(unsigned long long* host_out, int n);
#endif

/* Measurement aid: runs a bare v_mfma_f32_32x32x16_bf16 loop on pseudo-random operands on every CU (two waves per SIMD) for
 * at least min_ms and reports what the device SUSTAINS under power: dense bf16 TFLOP/s and the in-kernel clock (GHz).  bench.py
 * prints it next to the nominal 2.5 PFLOP/s.  Synchronous. */
int fa_mi355x_measure_mfma_peak(double min_ms, double* tflops, double* clock_ghz, void* stream);

/* Test hook: dumps what the MFMA operand readers see for a [64][d] tile (tests/test_gpu_layout.py).
 * All pointers are device pointers; returns a status code. */
int fa_mi355x_probe(const void* tile, const void* b, float* row_out, float* tr_out, float* mma_out,
                    float* swap_out, int d, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLASH_ATTN_MI355X_H */
