/*
 * oracle/shim/fftw3.h -- TEST INFRASTRUCTURE ONLY (never part of the product).
 *
 * Minimal stand-in for the FFTW3 single-precision API, so that the reference's
 * src/fsk.c can be compiled UNMODIFIED from /root/reference into oracle/_ref/.
 * FFTW3f is a third-party dependency of the reference (configure.ac:16,
 * "deps_packages=fftw3f", version unpinned) that is not vendored under
 * /root/reference and is not installed in this image.
 *
 * Only the five entry points fsk.c uses are declared:
 *   fftwf_malloc / fftwf_free            src/fsk.c:73,75,86,87,100,101
 *   fftwf_plan_many_dft_r2c              src/fsk.c:78-82 (rank 1, howmany 1)
 *   fftwf_execute                        src/fsk.c:157,552
 *   fftwf_destroy_plan                   src/fsk.c:102
 *
 * The transform itself (fftw3_shim.c) is the textbook DFT
 *     X[k] = sum_{n<N} x[n] * exp(-2*pi*i*k*n/N),   k = 0 .. N/2
 * evaluated with a mixed-radix FFT in double precision and rounded to float
 * on output.  It is therefore MORE accurate than FFTW's f32 codelets; bin
 * values differ from a real FFTW build by ~1e-7 relative (see DESIGN.md,
 * "oracle fidelity").
 */
#ifndef ORACLE_SHIM_FFTW3_H
#define ORACLE_SHIM_FFTW3_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef float fftwf_complex[2];
typedef struct oracle_fft_plan *fftwf_plan;

#define FFTW_MEASURE  (0U)
#define FFTW_ESTIMATE (1U << 6)

void *fftwf_malloc(size_t n);
void  fftwf_free(void *p);

fftwf_plan fftwf_plan_many_dft_r2c(int rank, const int *n, int howmany,
				   float *in, const int *inembed,
				   int istride, int idist,
				   fftwf_complex *out, const int *onembed,
				   int ostride, int odist,
				   unsigned flags);

void fftwf_execute(const fftwf_plan p);
void fftwf_destroy_plan(fftwf_plan p);

/* shim-only: number of fftwf_execute() calls so far (profiling aid) */
unsigned long long oracle_fft_execute_count(void);

#ifdef __cplusplus
}
#endif

#endif
