/* Plain C99 translation unit against include/rfgpu.h: proves the header is C (no C++ / torch
 * types) and that a C host links and calls the library.  Compute entries need a GPU; this
 * program only touches host-side entries and checks that rf_ctx_create fails cleanly when
 * no device is usable (exit 0 either way, prints what happened). */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "rfgpu.h"
#include "rfgpu_ext.h"   /* the extensions header is plain C too */

int main(void)
{
    enum { NSMP = 64 };
    double *r = (double *)malloc(sizeof(double) * NSMP * NSMP);
    int32_t rank = -1;
    if (rf_abi_version() != RFGPU_ABI_VERSION) return 2;
    if (rf_compute_r_inv(NSMP, 4.0, 0.05, r, &rank, NULL) != 0) return 3;
    /* symmetric to rounding, finite, positive rank */
    double asym = 0.0;
    for (int i = 0; i < NSMP; ++i)
        for (int j = 0; j < NSMP; ++j) {
            if (!isfinite(r[i + NSMP * j])) return 4;
            asym = fmax(asym, fabs(r[i + NSMP * j] - r[j + NSMP * i]));
        }
    printf("abi %d rank %d asym %.3e\n", rf_abi_version(), (int)rank, asym);

    double rayps[1] = {0.06}, a_gus[1] = {4.0}, obs[NSMP] = {0};
    int32_t ipha[1] = {1};
    rf_config cfg;
    cfg.nfft = 256; cfg.ntrc = 1; cfg.nsmp = NSMP; cfg.deconv_mode = 0;
    cfg.delta = 0.05; cfg.t_start = 0.0; cfg.sdep = 0.0;
    cfg.rayps = rayps; cfg.a_gus = a_gus; cfg.ipha = ipha; cfg.obs = obs; cfg.ldobs = NSMP;
    cfg.r_inv = r; cfg.max_walkers = 2; cfg.nlay_max = 8; cfg.device = 0;
    rf_ctx *ctx = NULL;
    if (rf_ctx_create(&cfg, &ctx) != 0) {
        printf("no context: %s\n", rf_last_error());
    } else {
        double alpha[2] = {5.0, 6.0}, beta[2] = {2.9, 3.5}, rho[2] = {2.5, 2.8}, h[2] = {3.0, 999.0};
        double *rft = (double *)malloc(sizeof(double) * 256);
        int rc = rf_calc_rf(ctx, 2, alpha, beta, rho, h, rft);
        printf("calc_rf rc %d rft[0] %.6f\n", rc, rft[0]);
        free(rft);
        rf_ctx_destroy(ctx);
    }
    free(r);
    return 0;
}
