// rolling.hip -- sliding / expanding window regressions (pl_rolling_lr, pl_recursive_lr).
#include "common.hpp"

namespace pds {

template <typename T>
int launch_rolling(pds_ctx* ctx, const DeviceCols<T>& dc, int n_feat, int64_t n_rows, int add_bias, int64_t window,
                   int64_t min_size, double lambda, bool expanding, T* d_coeffs, T* d_pred, uint8_t* d_valid) {
    return fail(PDS_ERR_UNSUPPORTED, "rolling / recursive kernels: not built yet");
}
template int launch_rolling<double>(pds_ctx*, const DeviceCols<double>&, int, int64_t, int, int64_t, int64_t, double,
                                    bool, double*, double*, uint8_t*);
template int launch_rolling<float>(pds_ctx*, const DeviceCols<float>&, int, int64_t, int, int64_t, int64_t, double, bool,
                                   float*, float*, uint8_t*);
}  // namespace pds
