/*
 * oracle/fsk_oracle.h -- CPU restatement of the reference's FSK receive path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product (libmifsk.so and the
 * minimodem_amd package) never links, imports or executes anything under
 * oracle/.
 *
 * What it restates (all citations into /root/reference/):
 *   ofsk_plan_new / _destroy        src/fsk.c:33-104
 *   ofsk_bit_analyze                src/fsk.c:107-174
 *   ofsk_frame_analyze              src/fsk.c:178-446 (CONFIDENCE_ALGO 6)
 *   ofsk_find_frame                 src/fsk.c:449-538
 *   ofsk_detect_carrier             src/fsk.c:543-581
 *   ofsk_set_tones_by_bandshift     src/fsk.c:584-598
 *   ofsk_rx_config_init             src/minimodem.c:819-965,1037-1131
 *   ofsk_rx_stream                  src/minimodem.c:1137-1463 (+ :253-291 stats)
 *
 * Arithmetic: the reference reads two bins of an FFTW r2c transform of the
 * zero-padded bit window; by definition that is
 *     X[b] = sum_{n < bit_nsamples} x[n] * exp(-2 pi i b n / fftsize).
 * FFTW3f is third-party, unpinned (configure.ac:16) and absent here, so the
 * oracle evaluates this sum directly: double twiddles (glibc cos/sin of the
 * exactly reduced angle), double fma accumulation in index order, result
 * rounded to float; everything after that (hypotf, scaling, confidence) is
 * the reference's own f32 expression sequence.  Pinned against: the
 * reference built from its own sources (oracle/_ref, shim FFT) on the
 * reference's tests 01-15, 21, 40, 41, 60, 80, 81 -- see tests/test_oracle_*.py
 * and tests/golden/.
 */
#ifndef FSK_ORACLE_H
#define FSK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/mifsk.h"	/* POD types shared across the boundary */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ofsk_plan {
    float		sample_rate;
    float		f_mark;
    float		f_space;
    float		filter_bw;
    int			fftsize;
    unsigned int	nbands;
    float		band_width;
    unsigned int	b_mark;
    unsigned int	b_space;
    /* twiddle cache for (b_mark, b_space) at the current window length */
    unsigned int	tw_bit_nsamples;
    unsigned int	tw_b_mark, tw_b_space;
    double		*tw;	/* [bit_nsamples][4]: cos_m, -sin_m, cos_s, -sin_s */
} ofsk_plan;

ofsk_plan *ofsk_plan_new( float sample_rate, float f_mark, float f_space, float filter_bw );
void ofsk_plan_destroy( ofsk_plan *p );

void ofsk_bit_analyze( ofsk_plan *p, const float *samples, unsigned int bit_nsamples,
	unsigned int *bit_outp, float *bit_signal_mag_outp, float *bit_noise_mag_outp );

float ofsk_frame_analyze( ofsk_plan *p, const float *samples, float samples_per_bit,
	int n_bits, const char *expect_bits_string,
	unsigned long long *bits_outp, float *ampl_outp );

float ofsk_find_frame( ofsk_plan *p, const float *samples, unsigned int frame_nsamples,
	unsigned int try_first_sample, unsigned int try_max_nsamples,
	unsigned int try_step_nsamples, float try_confidence_search_limit,
	const char *expect_bits_string,
	unsigned long long *bits_outp, float *ampl_outp, unsigned int *frame_start_outp );

int ofsk_detect_carrier( ofsk_plan *p, const float *samples, unsigned int nsamples,
	float min_mag_threshold );

void ofsk_set_tones_by_bandshift( ofsk_plan *p, unsigned int b_mark, int b_shift );

/* number of frame positions analysed by the last ofsk_find_frame() call */
unsigned int ofsk_last_n_positions( void );

/* twiddle of bin b at sample n: w[0] = cos(2 pi b n / N), w[1] = -sin(...) */
void ofsk_twiddle( unsigned int b, unsigned int n, unsigned int fftsize, double w[2] );

/* the raw 2-bin correlation of one bit window (before hypotf / scaling) */
void ofsk_bit_dft_f64( ofsk_plan *p, const float *samples, unsigned int bit_nsamples,
	double out[4] );	/* the same sums before rounding to float */
void ofsk_bit_dft( ofsk_plan *p, const float *samples, unsigned int bit_nsamples,
	float re_im_out[4] /* mark re, mark im, space re, space im */ );

void ofsk_modem_args_default( mifsk_modem_args *args );
int  ofsk_rx_config_init( mifsk_rx_config *cfg, const mifsk_modem_args *args );

/*
 * The receive loop over one in-memory stream.
 *   ring_mode 1: the reference's own buffering, cell for cell -- samplebuf of
 *                cfg->samplebuf_size floats (zero-initialised), memmove by
 *                `advance`, half-buffer refills (minimodem.c:1144-1174);
 *                searches may read stale cells past samples_nvalid exactly as
 *                the reference does.
 *   ring_mode 0: "flat" semantics, the contract of mifsk_demod_batch: the
 *                whole stream is addressable and reads past its end see 0.0.
 * Outputs go to caller arrays (capacity in *_cap; counts are returned even
 * when they exceed the capacity).  Returns 0, or -errno.
 */
typedef struct ofsk_rx_result {
    mifsk_frame		*frames;	size_t frames_cap;	size_t nframes;
    mifsk_episode	*episodes;	size_t episodes_cap;	size_t nepisodes;
    uint8_t		*bytes;		size_t bytes_cap;	size_t nbytes;
    /* work counters */
    /* --auto-carrier (minimodem.c:1179-1220): the band the mark tone was found in
     * (-1: never / not asked for), the space band derived from it, windows scanned */
    int			carrier_band;
    unsigned int	carrier_b_space;
    unsigned long long	n_scan_windows;
    unsigned long long	n_iterations;
    unsigned long long	n_find_frame;
    unsigned long long	n_positions;
} ofsk_rx_result;

int ofsk_rx_stream( const mifsk_rx_config *cfg, const float *samples, size_t nsamples,
	int ring_mode, ofsk_rx_result *res );

#ifdef __cplusplus
}
#endif

#endif
