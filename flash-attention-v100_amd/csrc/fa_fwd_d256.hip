// fa_fwd_d256.hip - the head-dim-256 (and 129 .. 192) instantiations of fa_fwd_kernel as their own translation unit.
//
// Same source as fa_fwd.hip (reference: kernel/fused_mha_forward.cu:421-428 is uniform over the head dim too); what differs is ONE
// compiler switch, given to this file only by build.py (EXTRA_FLAGS): -mllvm -amdgpu-mfma-vgpr-form.  A D = 256 workgroup runs one
// wave per SIMD with the whole 512-register file; by default hipcc then selects the accumulator form for every MFMA, which parks
// the score tile S in AGPRs - the online softmax reads each of its values back through v_accvgpr_read, the rare O rescale turns
// the loop-carried accumulators into VGPR-class values - and the kernels spilled 38 (plain) to 156 (paged) registers.  With the
// VGPR form nothing spills (csrc/spill_budget.json) and the other head dims keep their code.
#define FA_FWD_TU_D256 1
#include "fa_fwd.hip"
