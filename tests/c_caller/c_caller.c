/* A plain C caller of the drop-in boundary (include/fnft_amd.h), the calling pattern of the reference's
 * examples/fnft_nsev_example.c:34-85: the options struct RETURNED BY VALUE, fnft_nsev with caller-owned buffers, and
 * fnft__poly_chirpz with its two `double _Complex` arguments BY VALUE.  Prints numbers for tests/test_c_caller.py. */
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "fnft_amd.h"

int main(void)
{
    enum { D = 256, M = 8 };
    static FNFT_COMPLEX q[D], contspec[3 * M], p[4] = {1.0 + 2.0 * I, -0.5, 0.25 * I, 3.0}, cz[5];
    FNFT_REAL T[2] = {-25.0, 25.0}, XI[2] = {-1.4, 1.6};
    for (int i = 0; i < D; i++) q[i] = 3.2 * I / cosh(T[0] + i * (T[1] - T[0]) / (D - 1));
    fnft_nsev_opts_t opts = fnft_nsev_default_opts();          /* struct by value */
    printf("opts %d %d %zu %zu %d %d %d %d %zu\n", (int)opts.bound_state_filtering, (int)opts.bound_state_localization,
           (size_t)opts.niter, (size_t)opts.Dsub, (int)opts.discspec_type, (int)opts.contspec_type,
           (int)opts.normalization_flag, (int)opts.discretization, (size_t)opts.richardson_extrapolation_flag);
    fnft_kdvv_opts_t ko = fnft_kdvv_default_opts();
    fnft_nsep_opts_t po = fnft_nsep_default_opts();
    printf("opts2 %d %d %zu %g\n", (int)ko.discretization, (int)po.localization, (size_t)po.max_evals, po.tol);
    opts.discretization = fnft_nse_discretization_2SPLIT2_MODAL;
    opts.contspec_type = fnft_nsev_cstype_BOTH;
    FNFT_INT rc = fnft_nsev(D, q, T, M, contspec, XI, NULL, NULL, NULL, +1, &opts);
    printf("nsev %d\n", (int)rc);
    if (rc == FNFT_SUCCESS)
        for (int i = 0; i < 3 * M; i++) printf("cs %.17g %.17g\n", creal(contspec[i]), cimag(contspec[i]));
    const FNFT_COMPLEX A = 0.9 * cexp(0.3 * I), W = cexp(-0.2 * I);
    rc = fnft__poly_chirpz(3, p, A, W, 5, cz);                 /* complex arguments by value */
    printf("chirpz %d\n", (int)rc);
    if (rc == FNFT_SUCCESS)
        for (int i = 0; i < 5; i++) printf("cz %.17g %.17g\n", creal(cz[i]), cimag(cz[i]));
    return 0;
}
