/* A plain C99 consumer of include/homonim_hk.h: what a cgo / JNI / FFI binding would see.  Built and run by
 * tests/test_abi_cpu.py with gcc; it loads the library with dlopen (no link-time dependency on HIP) and calls the entry
 * points that need no GPU.  Exit code 0 = everything as documented. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "homonim_hk.h"

/* POSIX dlsym idiom (ISO C has no object-pointer -> function-pointer conversion) */
#define LOAD(name)                                                \
    name##_fn name##_p;                                           \
    *(void**)(&name##_p) = dlsym(lib, #name);                     \
    if (!name##_p) {                                              \
        fprintf(stderr, "missing symbol %s\n", #name);            \
        return 2;                                                 \
    }

typedef int (*hk_abi_version_fn)(void);
typedef const char* (*hk_backend_name_fn)(void);
typedef const char* (*hk_last_error_fn)(void);
typedef int (*hk_device_count_fn)(int*);
typedef int (*hk_ctx_create_fn)(int, int, hk_ctx**);
typedef int (*hk_ctx_destroy_fn)(hk_ctx*);
typedef int (*hk_fit_apply_fn)(hk_ctx*, const hk_fit_desc*, const float*, int64_t, const float*, int64_t, int32_t, int32_t,
                               const double*, float*, int32_t, float*, double*, uint64_t*);
typedef int (*hk_compare_sums_fn)(hk_ctx*, const float*, int64_t, int32_t, float, const float*, int64_t, int32_t, float,
                                  int32_t, int32_t, double*);

int main(int argc, char** argv) {
    if (argc < 2) return 64;
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 1;
    }
    LOAD(hk_abi_version)
    if (hk_abi_version_p() != HK_ABI_VERSION) { /* struct layouts differ between versions and carry no size field */
        fprintf(stderr, "library ABI %d, header %d\n", hk_abi_version_p(), HK_ABI_VERSION);
        return 5;
    }
    LOAD(hk_backend_name) LOAD(hk_last_error) LOAD(hk_device_count) LOAD(hk_ctx_create) LOAD(hk_ctx_destroy)
    LOAD(hk_fit_apply) LOAD(hk_compare_sums)
    if (strcmp(hk_backend_name_p(), "hip-gfx950") != 0) return 3;

    /* the structs have the documented C layout */
    if (sizeof(hk_fit_desc) != 40) return 4;
    if (sizeof(hk_dev_job) != 8 * 8 + 3 * 4 + 4 + 2 * 8 + 2 * 4 + 4 * 4 + 2 * 8 || sizeof(hk_out_window) != 40) return 14;
    hk_fit_desc desc;
    memset(&desc, 0, sizeof desc);
    desc.model = HK_MODEL_GAIN_OFFSET, desc.kh = 5, desc.kw = 5, desc.has_r2_thresh = 1, desc.r2_thresh = 0.25f;
    desc.src_nodata_mode = HK_NODATA_NAN, desc.ref_nodata_mode = HK_NODATA_NONE;

    /* NULL context: a clean error code + message, never a crash */
    float px[4] = {1.f, 2.f, 3.f, 4.f}, out[4];
    double sums[7];
    uint64_t fails = 0;
    if (hk_fit_apply_p(NULL, &desc, px, 2, px, 2, 2, 2, NULL, NULL, 0, out, NULL, &fails) != HK_ERR_ARG) return 5;
    if (strlen(hk_last_error_p()) == 0) return 6;
    if (hk_compare_sums_p(NULL, px, 2, HK_NODATA_NONE, 0.f, px, 2, HK_NODATA_NONE, 0.f, 2, 2, sums) != HK_ERR_ARG) return 7;

    int n = -1;
    int rc = hk_device_count_p(&n);
    hk_ctx* ctx = NULL;
    if (rc != HK_OK || n <= 0) {
        /* no GPU on this host: creating a context must fail loudly (there is no CPU path behind this ABI) */
        if (hk_ctx_create_p(0, 1, &ctx) == HK_OK) return 8;
        printf("abi_consumer: ok (no GPU: context creation refused: %s)\n", hk_last_error_p());
        return 0;
    }
    if (hk_ctx_create_p(0, 1, &ctx) != HK_OK) return 9;
    /* with a GPU: identical rasters compare with zero residual */
    if (hk_compare_sums_p(ctx, px, 2, HK_NODATA_NONE, 0.f, px, 2, HK_NODATA_NONE, 0.f, 2, 2, sums) != HK_OK) return 10;
    if (sums[6] != 4.0 || sums[5] != 0.0 || sums[0] != 10.0) return 11;
    hk_ctx_destroy_p(ctx);
    printf("abi_consumer: ok (GPU present)\n");
    return 0;
}
