// stubs.hip -- entry points declared in include/vcp.h that this build does not run on the GPU yet.
// They fail loudly (never fall back to a CPU path).
#include "vcp_ctx.hpp"

#define VCP_STUB(name) return vcp_fail(ctx, VCP_ERR_UNSUPPORTED, name " is not built yet")

extern "C" {
int vcp_dbscan_blocks(vcp_ctx* ctx, const double*, int64_t, double, int, int, int, int32_t*, int32_t*, int64_t*,
                      int64_t*, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, int64_t*) { VCP_STUB("vcp_dbscan_blocks"); }
}
