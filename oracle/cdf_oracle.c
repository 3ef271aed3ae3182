/*
 * cdf_oracle.c -- CPU restatement of the CDFTOOLS hot-path loops.  TEST INFRASTRUCTURE ONLY:
 * see cdf_oracle.h for who may use it and for the pinning status of each function.
 *
 * Build: gcc -O2 -ffp-contract=off [-fopenmp] (oracle/Makefile).  x86-64 baseline, no FMA
 * contraction, like the reference's plain -O/-O2 builds (Macrolib/macro.gfortran:26).
 */
#include "cdf_oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Fortran (ji,jj[,jk]) -> C offset, 0-based indices */
#define IJ(i, j) ((size_t)(j) * npi + (i))
#define IJK(i, j, k) (((size_t)(k) * npj + (j)) * npi + (i))

/*
 * src/cdfmoc.f90:352      dmoc(:,:,:) = 0.d0
 * src/cdfmoc.f90:355-379  DO jk = 1, npk-1 ; DO jbasin ; !$OMP PARALLEL DO ; DO jj ; DO ji
 *                            dmoc(jbasin,jj,jk) = dmoc(jbasin,jj,jk) -
 *                                 e1v(ji,jj)*e3v(ji,jj,jk)*ibmask(jbasin,ji,jj)*zv(ji,jj)*1.d0
 *   (r4 product evaluated left to right, the INTEGER*2 mask converted to r4, then *1.d0 widens)
 * src/cdfmoc.f90:382-388  DO jj ; DO jk = npk-1, 1, -1
 *                            dmoc(:,jj,jk) = dmoc(:,jj,jk+1) + dmoc(:,jj,jk)/1.d6
 */
void oracle_moc(int npi, int npj, int npk, int nb, const float *e1v, const float *e3v,
                const int16_t *ibmask, const float *zv, double *dmoc) {
    memset(dmoc, 0, sizeof(double) * (size_t)nb * npj * npk);
    for (int jk = 0; jk < npk - 1; ++jk) {
        for (int jb = 0; jb < nb; ++jb) {
#pragma omp parallel for schedule(runtime)
            for (int jj = 0; jj < npj; ++jj) {
                double d = dmoc[((size_t)jk * npj + jj) * nb + jb];
                for (int ji = 0; ji < npi; ++ji) {
                    float p = e1v[IJ(ji, jj)] * e3v[IJK(ji, jj, jk)];
                    p = p * (float)ibmask[IJ(ji, jj) * nb + jb];
                    p = p * zv[IJK(ji, jj, jk)];
                    d = d - (double)p * 1.0;
                }
                dmoc[((size_t)jk * npj + jj) * nb + jb] = d;
            }
        }
    }
#pragma omp parallel for schedule(runtime)
    for (int jj = 0; jj < npj; ++jj)
        for (int jk = npk - 2; jk >= 0; --jk)
            for (int jb = 0; jb < nb; ++jb)
                dmoc[((size_t)jk * npj + jj) * nb + jb] =
                    dmoc[((size_t)(jk + 1) * npj + jj) * nb + jb] +
                    dmoc[((size_t)jk * npj + jj) * nb + jb] / 1.0e6;
}
