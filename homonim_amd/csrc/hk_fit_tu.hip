// hk_fit_tu.hip -- one translation unit of the fused fit(+apply) kernel: every build (kernel width, ring mode, nodata
// specialisation, certificate-only, lock-step, batched) of ONE model with or without the R2 quantity set.  homonim_amd/build.py
// compiles this file six times (-DHK_TU_MODEL=0|1|2 -DHK_TU_R2=0|1) side by side; hk_kernels.hip dispatches to them.
#include "hk_fit_kernel.h"

#if !defined(HK_TU_MODEL) || !defined(HK_TU_R2)
#error "compile with -DHK_TU_MODEL=0|1|2 -DHK_TU_R2=0|1 (homonim_amd/build.py)"
#endif

namespace hk {

#define HK_TU_CAT2(p, m, r) p##m##_r##r
#define HK_TU_CAT(p, m, r) HK_TU_CAT2(p, m, r)

hipError_t HK_TU_CAT(launch_fit_m, HK_TU_MODEL, HK_TU_R2)(const FitArgs& a, hipStream_t stream) {
    return launch_dense<HK_TU_MODEL, HK_TU_R2 != 0>(a, stream);
}
hipError_t HK_TU_CAT(read_stamps_m, HK_TU_MODEL, HK_TU_R2)(unsigned long long* acc16, bool reset) { return read_stamps_tu(acc16, reset); }

}  // namespace hk
