// gemm_kernels.hip -- K3: batched queries, f32 MFMA Q x C^T with fused candidate
// selection.  (placeholder until the MFMA kernel lands: reports UNSUPPORTED so
// callers route batches through the K2 scan, which is exact for any batch.)
#include "common.h"

namespace smt {

int launch_gemm_topk(smt_ctx *ctx, const ScanArgs &a)
{
    (void)ctx; (void)a;
    set_error("batched MFMA path not built");
    return SMT_E_UNSUPPORTED;
}

}  // namespace smt
