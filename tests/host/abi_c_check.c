/* tests/host/abi_c_check.c -- TEST INFRASTRUCTURE: include/turbo_metrics_hip.h and turbo_metrics_hip_debug.h must be plain C (a Rust `extern "C"` block,
 * cgo or ctypes bind the same declarations).  Compiled with gcc -std=c99 -pedantic and linked against the library; runs
 * only the host-side entry points (no GPU needed).  Built twice: against the ship library with the facade header alone, and with
 * -DTM_ABI_CHECK_LAB against the laboratory build with both headers. */
#include <stdio.h>
#include <string.h>

#include "../../include/turbo_metrics_hip.h"
#ifdef TM_ABI_CHECK_LAB
#include "../../include/turbo_metrics_hip_debug.h"
#endif

int main(void)
{
    double sums[108], ssum[30];
    float g[11];
    tm_frame_scores fs;
    tm_engine *e = NULL;
    int i, rc;
    memset(sums, 0, sizeof sums);
    memset(ssum, 0, sizeof ssum);
    memset(&fs, 0, sizeof fs);
    if (strncmp(tm_version(), "turbo-metrics-hip", 17) != 0) return 1;
    if (strcmp(tm_strerror(TM_OK), "ok") != 0) return 2;
    if (tm_ssimulacra2_score_from_sums(sums, 64, 64) != 100.0) return 3;
#ifdef TM_ABI_CHECK_LAB
    tm_ssim_window(g);
    if (!(g[5] > g[4] && g[4] > g[0] && g[0] == g[10])) return 4;
#else
    (void)g;
    if (tm_p10_row_bytes(385) != 1024) return 4;
#endif
    for (i = 0; i < 30; ++i) ssum[i] = 54.0 * 54.0;
    if (tm_ssim_from_sums(ssum, 64, 64) != 1.0) return 5;
    rc = tm_engine_create(&e, 0, 0, TM_METRIC_SSIMULACRA2, 1); /* argument check happens before any device call */
    if (rc != TM_ERR_INVALID_ARG || e != NULL) return 6;
    if (tm_psnr_from_sse(0, 100) <= 1e30) return 7;             /* identical frames: +inf */
    if (sizeof(fs) != 4 * sizeof(double) + 8) return 8;         /* layout the Rust #[repr(C)] struct mirrors */
    printf("abi ok: %s\n", tm_version());
    return 0;
}
