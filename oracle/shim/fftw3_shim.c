/*
 * oracle/shim/fftw3_shim.c -- TEST INFRASTRUCTURE ONLY (never shipped, never
 * linked into libmifsk.so).
 *
 * Implements the five FFTW3f entry points declared in oracle/shim/fftw3.h so
 * that the reference's src/fsk.c links unmodified.  Real-input forward DFT,
 * any N, mixed radix (recursive decimation in time, generic O(p^2) butterfly
 * for each prime factor p), all arithmetic in double, output rounded to float.
 * Returns all N/2+1 bins (fsk_detect_carrier scans every bin,
 * /root/reference/src/fsk.c:568-576).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "fftw3.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

struct oracle_fft_plan {
    int		n;		/* real transform length */
    int		nc;		/* complex FFT length: n/2 if n even, else n */
    float	*in;
    fftwf_complex *out;
    double	*tw_re, *tw_im;	/* exp(-2 pi i k / nc), k < nc */
    double	*rw_re, *rw_im;	/* exp(-2 pi i k / n),  k <= n/2 (even n only) */
    double	*a_re, *a_im;	/* input of the complex FFT */
    double	*b_re, *b_im;	/* output of the complex FFT */
};

static unsigned long long execute_count;

unsigned long long oracle_fft_execute_count(void) { return execute_count; }

void *fftwf_malloc(size_t n)
{
    void *p = NULL;
    if ( posix_memalign(&p, 64, n ? n : 64) != 0 )
	return NULL;
    return p;
}

void fftwf_free(void *p) { free(p); }

static int smallest_factor(int n)
{
    if ( n % 4 == 0 ) return 4;
    if ( n % 2 == 0 ) return 2;
    for ( int p = 3; p * p <= n; p += 2 )
	if ( n % p == 0 ) return p;
    return n;
}

/*
 * out[0..n) = DFT_n of in[0], in[is], in[2 is], ...   (forward, e^{-i...})
 * tws = NC / n  (stride into the plan's twiddle table)
 */
static void fft_rec(const struct oracle_fft_plan *pl, int n,
	const double *in_re, const double *in_im, int is,
	double *out_re, double *out_im, int tws)
{
    if ( n == 1 ) {
	out_re[0] = in_re[0];
	out_im[0] = in_im[0];
	return;
    }
    const int p = smallest_factor(n);
    const int m = n / p;
    const int NC = pl->nc;

    for ( int r = 0; r < p; r++ )
	fft_rec(pl, m, in_re + (size_t)r * is, in_im + (size_t)r * is, is * p,
		out_re + (size_t)r * m, out_im + (size_t)r * m, tws * p);

    double yr[64], yi[64];
    double *tr = yr, *ti = yi;
    if ( p > 64 ) {
	tr = malloc(sizeof(double) * 2 * (size_t)p);
	ti = tr + p;
    }
    const int pstep = NC / p;		/* W_p^1 = W_NC^{NC/p} */

    for ( int k = 0; k < m; k++ ) {
	/* twiddle the sub-transform outputs: Y_r[k] * W_n^{r k} */
	for ( int r = 0; r < p; r++ ) {
	    double ar = out_re[(size_t)r * m + k], ai = out_im[(size_t)r * m + k];
	    long idx = ((long)r * k * tws) % NC;
	    double wr = pl->tw_re[idx], wi = pl->tw_im[idx];
	    tr[r] = ar * wr - ai * wi;
	    ti[r] = ar * wi + ai * wr;
	}
	if ( p == 2 ) {
	    out_re[k]     = tr[0] + tr[1];  out_im[k]     = ti[0] + ti[1];
	    out_re[k + m] = tr[0] - tr[1];  out_im[k + m] = ti[0] - ti[1];
	} else if ( p == 4 ) {
	    double s0r = tr[0] + tr[2], s0i = ti[0] + ti[2];
	    double s1r = tr[0] - tr[2], s1i = ti[0] - ti[2];
	    double s2r = tr[1] + tr[3], s2i = ti[1] + ti[3];
	    double s3r = tr[1] - tr[3], s3i = ti[1] - ti[3];
	    out_re[k]         = s0r + s2r;  out_im[k]         = s0i + s2i;
	    out_re[k + m]     = s1r + s3i;  out_im[k + m]     = s1i - s3r;
	    out_re[k + 2 * m] = s0r - s2r;  out_im[k + 2 * m] = s0i - s2i;
	    out_re[k + 3 * m] = s1r - s3i;  out_im[k + 3 * m] = s1i + s3r;
	} else {
	    for ( int q = 0; q < p; q++ ) {
		double sr = tr[0], si = ti[0];
		for ( int r = 1; r < p; r++ ) {
		    long idx = ((long)r * q % p) * pstep;
		    double wr = pl->tw_re[idx], wi = pl->tw_im[idx];
		    sr += tr[r] * wr - ti[r] * wi;
		    si += tr[r] * wi + ti[r] * wr;
		}
		/* cannot write in place yet: stash in the upper half of scratch */
		out_re[(size_t)q * m + k] = sr;	/* safe: column k of every */
		out_im[(size_t)q * m + k] = si;	/* block was consumed above */
	    }
	}
    }
    if ( tr != yr )
	free(tr);
}

fftwf_plan fftwf_plan_many_dft_r2c(int rank, const int *n, int howmany,
				   float *in, const int *inembed,
				   int istride, int idist,
				   fftwf_complex *out, const int *onembed,
				   int ostride, int odist,
				   unsigned flags)
{
    (void)inembed; (void)onembed; (void)idist; (void)odist; (void)flags;
    /* only the shape fsk.c asks for is supported */
    if ( rank != 1 || howmany != 1 || istride != 1 || ostride != 1 || n[0] < 1 )
	return NULL;

    struct oracle_fft_plan *pl = calloc(1, sizeof(*pl));
    if ( !pl )
	return NULL;
    pl->n = n[0];
    pl->in = in;
    pl->out = out;
    pl->nc = ( pl->n % 2 == 0 && pl->n >= 2 ) ? pl->n / 2 : pl->n;

    size_t nc = (size_t)pl->nc;
    pl->tw_re = malloc(sizeof(double) * 2 * nc);
    pl->a_re  = malloc(sizeof(double) * 4 * nc);
    pl->rw_re = malloc(sizeof(double) * 2 * ((size_t)pl->n / 2 + 1));
    if ( !pl->tw_re || !pl->a_re || !pl->rw_re ) {
	fftwf_destroy_plan(pl);
	return NULL;
    }
    pl->tw_im = pl->tw_re + nc;
    pl->a_im = pl->a_re + nc;
    pl->b_re = pl->a_re + 2 * nc;
    pl->b_im = pl->a_re + 3 * nc;
    pl->rw_im = pl->rw_re + ((size_t)pl->n / 2 + 1);

    for ( size_t k = 0; k < nc; k++ ) {
	double ang = -2.0 * M_PI * (double)k / (double)nc;
	pl->tw_re[k] = cos(ang);
	pl->tw_im[k] = sin(ang);
    }
    for ( size_t k = 0; k <= (size_t)pl->n / 2; k++ ) {
	double ang = -2.0 * M_PI * (double)k / (double)pl->n;
	pl->rw_re[k] = cos(ang);
	pl->rw_im[k] = sin(ang);
    }
    return pl;
}

void fftwf_execute(const fftwf_plan pl)
{
    execute_count++;
    const int n = pl->n, nc = pl->nc;
    const float *x = pl->in;

    if ( nc == n ) {
	/* odd (or length-1) transform: plain complex FFT of the real input */
	for ( int i = 0; i < n; i++ ) {
	    pl->a_re[i] = x[i];
	    pl->a_im[i] = 0.0;
	}
	fft_rec(pl, n, pl->a_re, pl->a_im, 1, pl->b_re, pl->b_im, 1);
	for ( int k = 0; k <= n / 2; k++ ) {
	    pl->out[k][0] = (float)pl->b_re[k];
	    pl->out[k][1] = (float)pl->b_im[k];
	}
	return;
    }

    /* even n: pack z[j] = x[2j] + i x[2j+1], FFT of length n/2, then split */
    for ( int j = 0; j < nc; j++ ) {
	pl->a_re[j] = x[2 * j];
	pl->a_im[j] = x[2 * j + 1];
    }
    fft_rec(pl, nc, pl->a_re, pl->a_im, 1, pl->b_re, pl->b_im, 1);

    for ( int k = 0; k <= nc; k++ ) {
	int k1 = k % nc;		/* Z[k]        */
	int k2 = (nc - k) % nc;		/* Z[nc-k]     */
	double zr = pl->b_re[k1], zi = pl->b_im[k1];
	double cr = pl->b_re[k2], ci = -pl->b_im[k2];	/* conj(Z[nc-k]) */
	/* even part E = (Z + conj(Z'))/2, odd part O = (Z - conj(Z'))/(2i) */
	double er = 0.5 * (zr + cr), ei = 0.5 * (zi + ci);
	double dr = 0.5 * (zr - cr), di = 0.5 * (zi - ci);
	double or_ = di, oi = -dr;			/* d / i */
	double wr = pl->rw_re[k], wi = pl->rw_im[k];
	pl->out[k][0] = (float)(er + or_ * wr - oi * wi);
	pl->out[k][1] = (float)(ei + or_ * wi + oi * wr);
    }
}

void fftwf_destroy_plan(fftwf_plan pl)
{
    if ( !pl )
	return;
    free(pl->tw_re);
    free(pl->a_re);
    free(pl->rw_re);
    free(pl);
}
