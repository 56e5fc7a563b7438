// One translation unit, built six times (compile_cuda.sh): each build is one of the shared objects the
// reference opens with ctypes.CDLL (minitorch/cuda_kernel_ops.py:30-35) and exports exactly the symbol
// that library exported there -- launch_flash_attn_fw (src/flash_attn_fw.cu:302, src/flash_attn2_fw.cu:312)
// or launch_flash_attn_bw (src/flash_attn_bw.cu:277, src/flash_attn2_bw.cu:279).  The variant is chosen by
// WHICH library is loaded, as in the reference (Makefile:28-50); the kernels live in
// libflash_attn_mi355x.so next to it (found through $ORIGIN).
#include "../../include/flash_attn_mi355x.h"

#if !defined(FA_SHIM_VARIANT) || !(defined(FA_SHIM_FW) || defined(FA_SHIM_BW))
#error "build with -DFA_SHIM_VARIANT=1|2 and -DFA_SHIM_FW or -DFA_SHIM_BW"
#endif

extern "C" {
#ifdef FA_SHIM_FW
void launch_flash_attn_fw(float* q, float* k, float* v, float* out, float* l, float* m, int batch, int N, int d,
                          bool causal_mask, void* stream) {
  fa_mi355x_launch_fw_host(FA_SHIM_VARIANT, q, k, v, out, l, m, batch, N, d, causal_mask, stream);
}
#endif
#ifdef FA_SHIM_BW
void launch_flash_attn_bw(float* q, float* k, float* v, float* out, float* out_grad, float* q_grad, float* k_grad,
                          float* v_grad, float* l, float* m, int batch, int N, int d, bool causal_mask, void* stream) {
  fa_mi355x_launch_bw_host(FA_SHIM_VARIANT, q, k, v, out, out_grad, q_grad, k_grad, v_grad, l, m, batch, N, d,
                           causal_mask, stream);
}
#endif
}
