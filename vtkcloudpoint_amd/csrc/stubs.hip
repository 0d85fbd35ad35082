// stubs.hip -- entry points declared in include/vcp.h that this build does not run on the GPU yet.
// They fail loudly (never fall back to a CPU path).
#include "vcp_ctx.hpp"

#define VCP_STUB(name) return vcp_fail(ctx, VCP_ERR_UNSUPPORTED, name " is not built yet")

extern "C" {
int vcp_dbscan_blocks(vcp_ctx* ctx, const double*, int64_t, double, int, int, int, int32_t*, int32_t*, int64_t*,
                      int64_t*, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, int64_t*) { VCP_STUB("vcp_dbscan_blocks"); }
int vcp_centroids(vcp_ctx* ctx, const double*, const double*, const int32_t*, int64_t, int32_t, double*, double*,
                  int64_t*) { VCP_STUB("vcp_centroids"); }
int vcp_centroids_dev(vcp_ctx* ctx, const double*, const double*, const int32_t*, int64_t, int32_t, double*, double*,
                      int64_t*) { VCP_STUB("vcp_centroids_dev"); }
int vcp_merge_centroids(vcp_ctx* ctx, const double*, const int32_t*, int32_t, double, int32_t*, int32_t*) { VCP_STUB("vcp_merge_centroids"); }
int vcp_refresh_by_dictionary(vcp_ctx* ctx, const double*, const double*, int32_t*, int64_t, int32_t, const int32_t*,
                              int32_t*, double*, double*, int64_t*) { VCP_STUB("vcp_refresh_by_dictionary"); }
int vcp_match(vcp_ctx* ctx, const double*, int32_t, const double*, int32_t, const double*, double, double*, uint8_t*,
              int32_t*, double*, int32_t*) { VCP_STUB("vcp_match"); }
}
