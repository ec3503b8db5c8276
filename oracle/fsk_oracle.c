/*
 * oracle/fsk_oracle.c -- CPU restatement of the reference's FSK receive path.
 *
 * TEST INFRASTRUCTURE ONLY (see fsk_oracle.h).  Parity status: PINNED against
 * the reference built from its own sources (oracle/_ref) on the reference's
 * own test inputs; see tests/test_oracle_vs_reference.py and tests/golden/.
 *
 * Build: gcc -O2 -mfma -ffp-contract=off (oracle/Makefile).  fma() below must
 * be the hardware instruction and nothing else may be contracted, because the
 * HIP kernels perform the same operation sequence and are compared bit for
 * bit.
 */
#include <errno.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <ctype.h>

#include "fsk_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ */
/* plan: reference src/fsk.c:33-104                                    */
/* ------------------------------------------------------------------ */

ofsk_plan *
ofsk_plan_new( float sample_rate, float f_mark, float f_space, float filter_bw )
{
    ofsk_plan *p = calloc(1, sizeof(*p));
    if ( !p )
	return NULL;
    p->sample_rate = sample_rate;
    p->f_mark = f_mark;
    p->f_space = f_space;
    /* filter_bw is left unset by the reference (fsk.c:45-50); keep 0 here */
    p->band_width = filter_bw;

    /* fsk.c:52-57 -- all in float, truncating conversions */
    float half_bw = p->band_width / 2.0f;
    p->fftsize = (sample_rate + half_bw) / p->band_width;
    p->nbands = p->fftsize / 2 + 1;
    p->b_mark = (f_mark + half_bw) / p->band_width;
    p->b_space = (f_space + half_bw) / p->band_width;

    if ( p->b_mark >= p->nbands || p->b_space >= p->nbands ) {	/* fsk.c:58-64 */
	fprintf(stderr, "b_mark=%u or b_space=%u is invalid (nbands=%u)\n",
		p->b_mark, p->b_space, p->nbands);
	free(p);
	errno = EINVAL;
	return NULL;
    }
    return p;
}

void
ofsk_plan_destroy( ofsk_plan *p )
{
    if ( !p )
	return;
    free(p->tw);
    free(p);
}

void
ofsk_twiddle( unsigned int b, unsigned int n, unsigned int fftsize, double w[2] )
{
    /* reduce the angle exactly in integers before going to floating point */
    unsigned long long k = ((unsigned long long)b * n) % fftsize;
    double ang = 2.0 * M_PI * (double)k / (double)fftsize;
    w[0] = cos(ang);
    w[1] = -sin(ang);
}

static void
ensure_twiddles( ofsk_plan *p, unsigned int bit_nsamples )
{
    if ( p->tw && p->tw_bit_nsamples == bit_nsamples
	    && p->tw_b_mark == p->b_mark && p->tw_b_space == p->b_space )
	return;
    free(p->tw);
    p->tw = malloc(sizeof(double) * 4 * (bit_nsamples ? bit_nsamples : 1));
    for ( unsigned int n = 0; n < bit_nsamples; n++ ) {
	ofsk_twiddle(p->b_mark, n, p->fftsize, &p->tw[4 * n]);
	ofsk_twiddle(p->b_space, n, p->fftsize, &p->tw[4 * n + 2]);
    }
    p->tw_bit_nsamples = bit_nsamples;
    p->tw_b_mark = p->b_mark;
    p->tw_b_space = p->b_space;
}

/* ------------------------------------------------------------------ */
/* per-bit two-band correlator: reference src/fsk.c:107-174            */
/* ------------------------------------------------------------------ */

/* the two bins before they are rounded to the FFT's output type (what the tests
 * that build inputs ON a float rounding boundary need: tests/test_gpu_guard.py) */
void
ofsk_bit_dft_f64( ofsk_plan *p, const float *samples, unsigned int bit_nsamples,
	double out[4] )
{
    ensure_twiddles(p, bit_nsamples);
    /* the window is the first bit_nsamples inputs of a zero-padded length-
     * fftsize transform (fsk.c:124-130): the padded tail adds nothing */
    double mr = 0.0, mi = 0.0, sr = 0.0, si = 0.0;
    const double *tw = p->tw;
    for ( unsigned int n = 0; n < bit_nsamples; n++, tw += 4 ) {
	double x = (double)samples[n];
	mr = fma(x, tw[0], mr);
	mi = fma(x, tw[1], mi);
	sr = fma(x, tw[2], sr);
	si = fma(x, tw[3], si);
    }
    out[0] = mr;
    out[1] = mi;
    out[2] = sr;
    out[3] = si;
}

void
ofsk_bit_dft( ofsk_plan *p, const float *samples, unsigned int bit_nsamples,
	float out[4] )
{
    double X[4];
    ofsk_bit_dft_f64(p, samples, bit_nsamples, X);
    /* fftout is an array of float pairs (fftwf_complex) */
    out[0] = (float)X[0];
    out[1] = (float)X[1];
    out[2] = (float)X[2];
    out[3] = (float)X[3];
}

void
ofsk_bit_analyze( ofsk_plan *p, const float *samples, unsigned int bit_nsamples,
	unsigned int *bit_outp, float *bit_signal_mag_outp, float *bit_noise_mag_outp )
{
    float X[4];
    ofsk_bit_dft(p, samples, bit_nsamples, X);

    float magscalar = 2.0f / (float)bit_nsamples;		/* fsk.c:132 */
    float mag_mark  = hypotf(X[0], X[1]) * magscalar;		/* fsk.c:107-114,158 */
    float mag_space = hypotf(X[2], X[3]) * magscalar;		/* fsk.c:159 */

    if ( mag_mark > mag_space ) {				/* fsk.c:161-169 */
	*bit_outp = 1;
	*bit_signal_mag_outp = mag_mark;
	*bit_noise_mag_outp = mag_space;
    } else {
	*bit_outp = 0;
	*bit_signal_mag_outp = mag_space;
	*bit_noise_mag_outp = mag_mark;
    }
}

/* ------------------------------------------------------------------ */
/* frame analysis / confidence: reference src/fsk.c:178-446 (ALGO 6)   */
/* ------------------------------------------------------------------ */

float
ofsk_frame_analyze( ofsk_plan *p, const float *samples, float samples_per_bit,
	int n_bits, const char *expect, unsigned long long *bits_outp, float *ampl_outp )
{
    unsigned int bit_nsamples = (float)(samples_per_bit + 0.5f);	/* fsk.c:183 */
    unsigned int value[MIFSK_MAX_FRAME_BITS];
    float sig[MIFSK_MAX_FRAME_BITS], noise[MIFSK_MAX_FRAME_BITS];

    /* required bits first; the first mismatch rejects the frame with
     * confidence 0 and leaves the out-params untouched (fsk.c:199-212) */
    for ( int k = 0; k < n_bits; k++ ) {
	if ( expect[k] == 'd' )
	    continue;
	unsigned int begin = (float)(samples_per_bit * k + 0.5f);	/* fsk.c:204 */
	ofsk_bit_analyze(p, samples + begin, bit_nsamples, &value[k], &sig[k], &noise[k]);
	if ( (unsigned int)(expect[k] - '0') != value[k] )
	    return 0.0f;
    }
    /* then the don't-care bits (fsk.c:246-254) */
    for ( int k = 0; k < n_bits; k++ ) {
	if ( expect[k] != 'd' )
	    continue;
	unsigned int begin = (float)(samples_per_bit * k + 0.5f);	/* fsk.c:249 */
	ofsk_bit_analyze(p, samples + begin, bit_nsamples, &value[k], &sig[k], &noise[k]);
    }

    /* fsk.c:271-289: f32 running sums, in bit order */
    float total_sig = 0.0f, total_noise = 0.0f;
    float mark_sig = 0.0f, space_sig = 0.0f;
    unsigned int n_mark = 0, n_space = 0;
    for ( int k = 0; k < n_bits; k++ ) {
	total_sig += sig[k];
	if ( noise[k] > FLT_EPSILON )
	    total_noise += noise[k];
	if ( value[k] == 1 ) {
	    mark_sig += sig[k];
	    n_mark++;
	} else {
	    space_sig += sig[k];
	    n_space++;
	}
    }
    float snr = total_sig / total_noise;			/* fsk.c:292 */
    float avg_sig = total_sig / n_bits;				/* fsk.c:295 */
    if ( n_mark )
	mark_sig /= n_mark;					/* fsk.c:298-301 */
    if ( n_space )
	space_sig /= n_space;

    float divergence = 0.0f;					/* fsk.c:305-313 */
    for ( int k = 0; k < n_bits; k++ ) {
	float cls = value[k] ? mark_sig : space_sig;
	divergence += fabsf(sig[k] - cls) / cls;
    }
    divergence *= 2;
    divergence /= n_bits;

    float confidence = snr * (1.0f - divergence);		/* fsk.c:336 */
    *ampl_outp = avg_sig;					/* fsk.c:342 */

    unsigned long long bits = 0;				/* fsk.c:439-441 */
    for ( int k = 0; k < n_bits; k++ )
	bits |= (unsigned long long)value[k] << k;
    *bits_outp = bits;
    return confidence;
}

/* ------------------------------------------------------------------ */
/* sliding search: reference src/fsk.c:449-538                         */
/* ------------------------------------------------------------------ */

/* (thread-local: the whole-batch parity checks run ofsk_rx_stream on every host core) */
static __thread unsigned int last_n_positions;

unsigned int ofsk_last_n_positions( void ) { return last_n_positions; }

float
ofsk_find_frame( ofsk_plan *p, const float *samples, unsigned int frame_nsamples,
	unsigned int try_first_sample, unsigned int try_max_nsamples,
	unsigned int try_step_nsamples, float limit, const char *expect,
	unsigned long long *bits_outp, float *ampl_outp, unsigned int *frame_start_outp )
{
    int n_bits = (int)strlen(expect);
    if ( n_bits > MIFSK_MAX_FRAME_BITS ) {	/* assert in the reference (fsk.c:463) */
	*bits_outp = 0; *ampl_outp = 0; *frame_start_outp = 0;
	return 0.0f;
    }
    float samples_per_bit = (float)frame_nsamples / n_bits;	/* fsk.c:465 */

    unsigned int best_t = 0;
    float best_c = 0.0f, best_a = 0.0f;
    unsigned long long best_bits = 0;
    last_n_positions = 0;

    /* zig-zag: first, first+step, first-step, first+2 step, ... (fsk.c:477-502) */
    for ( int j = 0; ; j++ ) {
	int up = ( j % 2 ) ? 1 : -1;
	int t = try_first_sample + up * ((j + 1) / 2) * try_step_nsamples;
	if ( t >= (int)try_max_nsamples )
	    break;			/* ends the whole scan */
	if ( t < 0 )
	    continue;
	float a = 0.0f;
	unsigned long long b = 0;
	last_n_positions++;
	float c = ofsk_frame_analyze(p, samples + t, samples_per_bit, n_bits, expect, &b, &a);
	if ( best_c < c ) {		/* strict: first tried wins ties; NaN never wins */
	    best_t = t;
	    best_c = c;
	    best_a = a;
	    best_bits = b;
	    if ( best_c >= limit )
		break;
	}
    }
    *bits_outp = best_bits;		/* always written (fsk.c:504-506) */
    *ampl_outp = best_a;
    *frame_start_outp = best_t;
    return best_c;
}

/* ------------------------------------------------------------------ */
/* carrier autodetect: reference src/fsk.c:543-598                     */
/* ------------------------------------------------------------------ */

int
ofsk_detect_carrier( ofsk_plan *p, const float *samples, unsigned int nsamples,
	float min_mag_threshold )
{
    if ( nsamples > (unsigned int)p->fftsize )	/* assert in the reference (fsk.c:547) */
	return -1;
    const unsigned int N = p->fftsize;
    /* table of the last fftsize, one per thread: ofsk_rx_stream is called from many threads at
     * once (tests/_oracle.py oracle_batch_mismatches, tools/soak.py), and a table shared between
     * them was freed and rebuilt by two threads at the same time at the first --auto-carrier
     * mode of a soak run (round 5: one run in 108 died of a corrupted heap in the CHECKER) */
    static __thread double *cs;
    static __thread unsigned int cs_n;
    if ( cs_n != N ) {
	free(cs);
	cs = malloc(sizeof(double) * 2 * N);
	for ( unsigned int k = 0; k < N; k++ ) {
	    double ang = 2.0 * M_PI * (double)k / (double)N;
	    cs[2 * k] = cos(ang);
	    cs[2 * k + 1] = -sin(ang);
	}
	cs_n = N;
    }
    float magscalar = 1.0f / ((float)nsamples / 2.0f);		/* fsk.c:553 */
    float max_mag = 0.0f;
    int max_band = -1;
    for ( unsigned int b = 1; b < p->nbands; b++ ) {		/* fsk.c:556,568 */
	double re = 0.0, im = 0.0;
	for ( unsigned int n = 0; n < nsamples; n++ ) {
	    unsigned long long k = ((unsigned long long)b * n) % N;
	    double x = (double)samples[n];
	    re = fma(x, cs[2 * k], re);
	    im = fma(x, cs[2 * k + 1], im);
	}
	float mag = hypotf((float)re, (float)im) * magscalar;
	if ( mag < min_mag_threshold )
	    continue;
	if ( max_mag < mag ) {
	    max_mag = mag;
	    max_band = (int)b;
	}
    }
    return max_band;
}

void
ofsk_set_tones_by_bandshift( ofsk_plan *p, unsigned int b_mark, int b_shift )
{
    int b_space = (int)b_mark + b_shift;
    p->b_mark = b_mark;
    p->b_space = (unsigned int)b_space;
    p->f_mark = b_mark * p->band_width;
    p->f_space = b_space * p->band_width;
}

/* ------------------------------------------------------------------ */
/* option defaults + derived receive configuration                     */
/* reference src/minimodem.c:492-553 (defaults), :819-965 (presets),   */
/* :1037-1131 (derived), :442-487 (expect strings)                     */
/* ------------------------------------------------------------------ */

void
ofsk_modem_args_default( mifsk_modem_args *a )
{
    memset(a, 0, sizeof(*a));
    a->baudmode = "1200";
    a->nstartbits = -1;
    a->nstopbits = -1.0f;
    a->confidence_threshold = -1.0f;
    a->search_limit = -1.0f;
    a->sync_byte = -1;
}

static int
expect_string( char *out, int nstart, int ndata, float nstop, int invert,
	int use_bits, unsigned long long bits )
{
    char start_v = invert ? '1' : '0', stop_v = invert ? '0' : '1';
    int j = 0;
    if ( nstop != 0.0f )
	out[j++] = stop_v;		/* the previous frame's stop bit */
    for ( int i = 0; i < nstart; i++ )
	out[j++] = start_v;
    for ( int i = 0; i < ndata; i++ )
	out[j++] = use_bits ? (char)('0' + ((bits >> i) & 1)) : 'd';
    if ( nstop != 0.0f )
	out[j++] = stop_v;
    out[j] = 0;
    return j;
}

int
ofsk_rx_config_init( mifsk_rx_config *c, const mifsk_modem_args *a )
{
    memset(c, 0, sizeof(*c));
    const char *mode = a->baudmode ? a->baudmode : "";

    float band_width = a->band_width;
    float mark = a->mark_f, space = a->space_f;
    int nstart = a->nstartbits < 0 ? -1 : a->nstartbits;
    float nstop = a->nstopbits < 0 ? -1.0f : a->nstopbits;
    unsigned int do_sync = a->have_sync_byte ? 1 : 0;
    unsigned long long sync_byte = a->have_sync_byte ? (unsigned long long)a->sync_byte
						      : (unsigned long long)-1;
    unsigned int ndata = a->n_data_bits > 0 ? (unsigned int)a->n_data_bits : 0;
    int decoder = a->baudot ? MIFSK_DECODE_BAUDOT : MIFSK_DECODE_ASCII8;
    float rate = 0.0f;
    const char *fixed_expect = NULL;

    if ( strncasecmp(mode, "rtty", 5) == 0 ) {			/* :819-826 */
	decoder = MIFSK_DECODE_BAUDOT;
	rate = 45.45;
	if ( ndata == 0 ) ndata = 5;
	if ( nstop < 0 ) nstop = 1.5;
    } else if ( strncasecmp(mode, "tdd", 4) == 0 ) {		/* :827-836 */
	decoder = MIFSK_DECODE_BAUDOT;
	rate = 45.45;
	if ( ndata == 0 ) ndata = 5;
	if ( nstop < 0 ) nstop = 2.0;
	mark = 1400;
	space = 1800;
    } else if ( strncasecmp(mode, "same", 5) == 0 ) {		/* :837-848 */
	rate = 520.0 + 5 / 6.0;
	ndata = 8;
	nstart = 0;
	nstop = 0;
	do_sync = 1;
	sync_byte = 0xAB;
	mark = 2083.0 + 1 / 3.0;
	space = 1562.5;
	band_width = rate;
    } else if ( strncasecmp(mode, "caller", 6) == 0 ) {		/* :849-858 */
	decoder = MIFSK_DECODE_CALLERID;
	rate = 1200;
	ndata = 8;
    } else if ( strncasecmp(mode, "uic", 3) == 0 ) {		/* :859-876 */
	decoder = ( strlen(mode) > 4 && tolower((unsigned char)mode[4]) == 't' )
		? MIFSK_DECODE_UIC_TRAIN : MIFSK_DECODE_UIC_GROUND;
	rate = 600;
	ndata = 39;
	mark = 1300;
	space = 1700;
	nstart = 8;
	nstop = 0;
	fixed_expect = "11110010ddddddddddddddddddddddddddddddddddddddd";
    } else if ( strncasecmp(mode, "V.21", 4) == 0 ) {		/* :877-881 */
	rate = 300;
	mark = 980;
	space = 1180;
	ndata = 8;
    } else {							/* :882-886 */
	rate = atof(mode);
	if ( ndata == 0 ) ndata = 8;
    }
    if ( rate == 0.0f )
	return -EINVAL;

    if ( a->binary_output || a->binary_raw_nbits )		/* :891-898 */
	decoder = MIFSK_DECODE_BINARY;
    if ( a->binary_raw_nbits ) {
	nstart = 0;
	nstop = 0;
	ndata = a->binary_raw_nbits;
    }

    int autodetect_shift;
    if ( rate >= 400 ) {					/* :900-910 */
	autodetect_shift = -( rate * 5 / 6 );
	if ( mark == 0 ) mark = rate / 2 + 600;
	if ( space == 0 ) space = mark - autodetect_shift;
	if ( band_width == 0 ) band_width = 200;
    } else if ( rate >= 100 ) {					/* :911-921 */
	autodetect_shift = 200;
	if ( mark == 0 ) mark = 1270;
	if ( space == 0 ) space = mark - autodetect_shift;
	if ( band_width == 0 ) band_width = 50;
    } else {							/* :922-934 */
	autodetect_shift = 170;
	if ( mark == 0 ) mark = 1585;
	if ( space == 0 ) space = mark - autodetect_shift;
	if ( band_width == 0 ) band_width = 10;
    }
    if ( nstart < 0 ) nstart = 1;				/* :937-940 */
    if ( nstop < 0 ) nstop = 1.0;

    unsigned int frame_n_bits = ndata + nstart + nstop;		/* :943 (float -> unsigned) */
    if ( frame_n_bits > 64 )
	return -EINVAL;
    if ( a->inverted_freqs ) {					/* :953-957 */
	float t = mark; mark = space; space = t;
    }
    if ( band_width > rate )					/* :960-961 */
	band_width = rate;

    float thr = a->confidence_threshold < 0 ? 1.5f : a->confidence_threshold;
    float lim = a->search_limit < 0 ? 2.3f : a->search_limit;
    if ( lim < thr )						/* :964-965 */
	lim = thr;

    unsigned int sample_rate = a->sample_rate ? a->sample_rate : 48000;

    c->sample_rate = sample_rate;
    c->data_rate = rate;
    c->mark_f = mark;
    c->space_f = space;
    c->band_width = band_width;
    c->n_data_bits = ndata;
    c->nstartbits = nstart;
    c->nstopbits = nstop;
    c->invert_start_stop = a->invert_start_stop;
    c->msb_first = a->msb_first;
    c->do_rx_sync = do_sync;
    c->sync_byte = sync_byte;
    c->decoder = decoder;
    c->rx_one = a->rx_one;
    c->confidence_threshold = thr;
    c->search_limit = lim;
    c->auto_carrier_threshold = a->auto_carrier_threshold;
    c->autodetect_shift = autodetect_shift;
    c->inverted_freqs = a->inverted_freqs;

    /* plan bins (fsk.c:52-57) */
    ofsk_plan *p = ofsk_plan_new((float)sample_rate, mark, space, band_width);
    if ( !p )
	return -EINVAL;
    c->fftsize = p->fftsize;
    c->nbands = p->nbands;
    c->b_mark = p->b_mark;
    c->b_space = p->b_space;
    ofsk_plan_destroy(p);

    /* minimodem.c:1037 */
    float spb = sample_rate / rate;
    c->nsamples_per_bit = spb;
    c->frame_n_bits = frame_n_bits;

    /* :1056-1070 sample buffer */
    unsigned int nbits = 1 + nstart + ndata + 1;
    size_t bufsize = ceilf(spb) * (nbits + 1);
    bufsize *= 2;
    if ( bufsize < sample_rate / 12 )
	bufsize = sample_rate / 12;
    c->samplebuf_size = (unsigned int)bufsize;

    /* :1091,1105-1113 */
    float overscan_frac = 0.5;
    unsigned int overscan = spb * overscan_frac + 0.5f;
    if ( overscan_frac > 0.0f && overscan == 0 )
	overscan = 1;
    c->nsamples_overscan = overscan;
    float frame_bits_f = frame_n_bits;
    c->frame_nsamples = spb * frame_bits_f + 0.5f;

    /* :1115-1131 */
    unsigned int expect_n;
    if ( fixed_expect ) {
	strcpy(c->expect_data, fixed_expect);
	expect_n = 47;
    } else {
	expect_n = expect_string(c->expect_data, nstart, ndata, nstop,
				 a->invert_start_stop, 0, 0);
    }
    if ( do_sync && (long long)sync_byte >= 0 )
	expect_string(c->expect_sync, nstart, ndata, nstop, a->invert_start_stop, 1, sync_byte);
    else
	strcpy(c->expect_sync, c->expect_data);
    c->expect_n_bits = expect_n;
    c->expect_nsamples = spb * expect_n;

    /* search grid, :1236-1263 and :1366 */
    for ( int carrier = 0; carrier < 2; carrier++ ) {
	unsigned int try_max;
	if ( carrier )
	    try_max = spb * 0.75f + 0.5f;
	else
	    try_max = spb;
	try_max += overscan;
	unsigned int step = try_max / 3;
	if ( step == 0 ) step = 1;
	unsigned int fine = try_max / 8;
	if ( fine == 0 ) fine = 1;
	c->try_max[carrier] = try_max;
	c->try_step[carrier] = step;
	c->try_step_fine[carrier] = fine;
	c->try_first[carrier] = carrier ? overscan : 0;
    }

    /* bit windows, fsk.c:183,204,465 */
    float fspb = (float)c->expect_nsamples / (int)expect_n;
    c->find_samples_per_bit = fspb;
    c->bit_nsamples = (float)(fspb + 0.5f);
    for ( unsigned int k = 0; k < expect_n && k < MIFSK_MAX_FRAME_BITS; k++ )
	c->bit_offset[k] = (float)(fspb * (int)k + 0.5f);
    return 0;
}

/* ------------------------------------------------------------------ */
/* the receive loop: reference src/minimodem.c:1137-1463               */
/* ------------------------------------------------------------------ */

/* databits.h:21-46 */
static unsigned long long
reverse_bits( unsigned long long value, unsigned int bits )
{
    unsigned int out = 0;	/* 32-bit on purpose: that is what the reference does */
    while ( bits-- ) {
	out = (out << 1) | (value & 1);
	value >>= 1;
    }
    return out;
}

static unsigned long long
window_bits( unsigned long long value, unsigned int offset, unsigned int bits )
{
    unsigned long long mask = (1ULL << (bits & 63)) - 1;
    if ( bits >= 64 || mask == 0 )
	return value >> offset;
    return (value >> offset) & mask;
}

static void
push_episode( ofsk_rx_result *res, size_t first_frame, unsigned int nframes,
	size_t carrier_nsamples, float conf_total, float ampl_total, unsigned int reason,
	unsigned int b_mark )
{
    if ( res->nepisodes < res->episodes_cap ) {
	mifsk_episode *e = &res->episodes[res->nepisodes];
	memset(e, 0, sizeof(*e));
	e->carrier_nsamples = carrier_nsamples;
	e->first_frame = (uint32_t)first_frame;
	e->nframes = nframes;
	e->confidence_total = conf_total;
	e->amplitude_total = ampl_total;
	e->end_reason = reason;
	e->b_mark = b_mark;
    }
    res->nepisodes++;
}

/* the space band that goes with an autodetected mark band (minimodem.c:1203-1215);
 * returns 0 when the pair is rejected */
static int
auto_space_band( const mifsk_rx_config *cfg, const ofsk_plan *p, int carrier_band, int *b_shift_out )
{
    int b_shift = - (float)(cfg->autodetect_shift + p->band_width / 2.0f) / p->band_width;
    if ( cfg->inverted_freqs )
	b_shift *= -1;
    int b_space = carrier_band + b_shift;
    *b_shift_out = b_shift;
    return !( b_space < 1 || b_space >= (int)p->nbands );
}

int
ofsk_rx_stream( const mifsk_rx_config *cfg, const float *samples, size_t nsamples,
	int ring_mode, ofsk_rx_result *res )
{

    ofsk_plan *p = ofsk_plan_new((float)cfg->sample_rate, cfg->mark_f, cfg->space_f,
				 cfg->band_width);
    if ( !p )
	return -EINVAL;

    const float spb = cfg->nsamples_per_bit;
    const unsigned int overscan = cfg->nsamples_overscan;
    const unsigned int frame_nsamples = cfg->frame_nsamples;
    const unsigned int expect_nsamples = cfg->expect_nsamples;
    const size_t bufsize = cfg->samplebuf_size;

    /*
     * One buffer arithmetic for both modes -- (base, nvalid) = absolute index of
     * samplebuf[0] and samples_nvalid, evolving exactly as minimodem.c:1144-1174
     * makes them evolve (the file position is always base + nvalid).  The modes
     * differ only in what a search sees in cells at or beyond nvalid:
     *   ring: whatever the reference's buffer holds there (stale samples that
     *         memmove left behind; zero where nothing was ever written),
     *   flat: the stream itself, 0.0 beyond its end.
     */
    size_t pad = 2 * (size_t)expect_nsamples + 4 * (size_t)ceilf(spb) + 64;
    float *buf;
    size_t nvalid = 0;		/* samples_nvalid */
    size_t base = 0;		/* absolute index of samplebuf[0] */
    if ( ring_mode ) {
	buf = calloc(bufsize + pad, sizeof(float));
    } else {
	buf = calloc(nsamples + pad, sizeof(float));
	if ( buf )
	    memcpy(buf, samples, nsamples * sizeof(float));
    }
    if ( !buf ) {
	ofsk_plan_destroy(p);
	return -ENOMEM;
    }

    res->nframes = res->nepisodes = res->nbytes = 0;
    res->n_iterations = res->n_find_frame = res->n_positions = 0;
    res->carrier_band = -1;
    res->carrier_b_space = 0;
    res->n_scan_windows = 0;
    const int autodetect = cfg->auto_carrier_threshold > 0.0f;
    int carrier_band = -1;		/* `static int carrier_band = -1` (minimodem.c:1180) */

    int carrier = 0;
    float confidence_total = 0, amplitude_total = 0;
    unsigned int nframes_decoded = 0;
    size_t carrier_nsamples = 0;
    unsigned int noconfidence = 0;
    unsigned int advance = 0;
    float track_amplitude = 0.0f, peak_confidence = 0.0f;
    size_t episode_first_frame = 0;
    unsigned int episode_b_mark = 0;

    for (;;) {
	float *win;
	/* minimodem.c:1144-1177 */
	if ( advance == bufsize ) {
	    nvalid = 0;
	    base += advance;
	    advance = 0;
	}
	if ( advance ) {
	    if ( advance > nvalid )
		break;
	    if ( ring_mode )
		memmove(buf, buf + advance, (bufsize - advance) * sizeof(float));
	    nvalid -= advance;
	    base += advance;
	}
	if ( nvalid < bufsize / 2 ) {
	    size_t want = bufsize / 2;
	    size_t rp = base + nvalid;		/* the file position */
	    size_t r = nsamples - rp < want ? nsamples - rp : want;
	    if ( ring_mode )
		memcpy(buf + nvalid, samples + rp, r * sizeof(float));
	    nvalid += r;
	}
	win = ring_mode ? buf : buf + base;
	if ( nvalid == 0 )
	    break;
	if ( autodetect && carrier_band < 0 ) {			/* :1179-1220 */
	    unsigned int i;
	    float nps = spb;
	    if ( nps > p->fftsize )
		nps = p->fftsize;
	    for ( i = 0; i + nps <= nvalid; i += nps ) {
		carrier_band = ofsk_detect_carrier(p, win + i, nps, cfg->auto_carrier_threshold);
		res->n_scan_windows++;
		if ( carrier_band >= 0 )
		    break;
	    }
	    advance = i + nps;
	    if ( advance > nvalid )
		advance = nvalid;
	    if ( carrier_band < 0 )
		continue;
	    int b_shift;
	    if ( !auto_space_band(cfg, p, carrier_band, &b_shift) ) {
		carrier_band = -1;
		continue;
	    }
	    ofsk_set_tones_by_bandshift(p, (unsigned int)carrier_band, b_shift);
	    if ( res->carrier_band < 0 ) {		/* the first pair found */
		res->carrier_band = carrier_band;
		res->carrier_b_space = p->b_space;
	    }
	}
	if ( nvalid < expect_nsamples )				/* :1229 */
	    break;
	res->n_iterations++;

	const int ci = carrier ? 1 : 0;
	unsigned int try_max = cfg->try_max[ci];		/* :1236-1241 */
	unsigned int try_step = cfg->try_step[ci];		/* :1249-1251 */
	unsigned int try_first = cfg->try_first[ci];		/* :1263 */
	float limit = cfg->search_limit;

	float confidence, amplitude = 0;
	unsigned long long bits = 0;
	unsigned int frame_start = 0;

	confidence = ofsk_find_frame(p, win, expect_nsamples, try_first, try_max, try_step,
		limit, carrier ? cfg->expect_data : cfg->expect_sync,
		&bits, &amplitude, &frame_start);		/* :1265-1274 */
	res->n_find_frame++;
	res->n_positions += ofsk_last_n_positions();

	int refine = 0;
	if ( confidence < peak_confidence * 0.75f ) {		/* :1278-1282 */
	    refine = 1;
	    peak_confidence = 0;
	}
	if ( amplitude < track_amplitude * 0.25f )		/* :1286-1288 */
	    confidence = 0;

	if ( confidence <= cfg->confidence_threshold ) {	/* :1292-1321 */
	    if ( ++noconfidence > 20 ) {
		carrier_band = -1;				/* :1297: look for the tone again */
		if ( carrier ) {
		    push_episode(res, episode_first_frame, nframes_decoded, carrier_nsamples,
			    confidence_total, amplitude_total, 1, episode_b_mark);
		    carrier = 0;
		    carrier_nsamples = 0;
		    confidence_total = 0;
		    amplitude_total = 0;
		    nframes_decoded = 0;
		    track_amplitude = 0.0f;
		    if ( cfg->rx_one )
			break;
		}
	    }
	    advance = try_max;
	    continue;
	}

	carrier_nsamples += frame_nsamples;			/* :1324 */
	uint32_t flags = 0;
	if ( carrier ) {
	    carrier_nsamples += frame_start;			/* :1329-1330 */
	    carrier_nsamples -= overscan;
	} else {
	    carrier = 1;					/* :1350-1353 */
	    refine = 1;
	    flags |= MIFSK_FRAME_ACQUIRE;
	    episode_first_frame = res->nframes;
	    episode_b_mark = p->b_mark;				/* :1340,1344: "@ %.1f Hz" */
	}

	if ( refine ) {						/* :1357-1389 */
	    if ( confidence < INFINITY && try_step > 1 ) {
		unsigned int fine = cfg->try_step_fine[ci];
		float c2, a2 = 0;
		unsigned long long b2 = 0;
		unsigned int s2 = 0;
		/* note: `carrier` is already 1 here, so a just-acquired frame is
		 * re-searched with the data string over the no-carrier range */
		c2 = ofsk_find_frame(p, win, expect_nsamples, try_first, try_max, fine,
			INFINITY, carrier ? cfg->expect_data : cfg->expect_sync,
			&b2, &a2, &s2);
		res->n_find_frame++;
		res->n_positions += ofsk_last_n_positions();
		flags |= MIFSK_FRAME_REFINED;
		if ( c2 > confidence ) {
		    bits = b2;
		    amplitude = a2;
		    frame_start = s2;
		}
	    }
	}

	track_amplitude = ( track_amplitude + amplitude ) / 2;	/* :1391-1400 */
	if ( peak_confidence < confidence )
	    peak_confidence = confidence;
	confidence_total += confidence;
	amplitude_total += amplitude;
	nframes_decoded++;
	noconfidence = 0;

	advance = frame_start + frame_nsamples - overscan;	/* :1407 */

	if ( cfg->nstopbits != 0.0f )				/* :1415-1428 */
	    bits = bits >> 1;
	bits = window_bits(bits, cfg->nstartbits, cfg->n_data_bits);
	if ( cfg->msb_first )
	    bits = reverse_bits(bits, cfg->n_data_bits);

	int suppressed = cfg->do_rx_sync && bits == cfg->sync_byte;	/* :1436-1439 */
	if ( suppressed )
	    flags |= MIFSK_FRAME_SYNC;

	if ( res->nframes < res->frames_cap ) {
	    mifsk_frame *f = &res->frames[res->nframes];
	    f->bits = bits;
	    f->start = base + frame_start;
	    f->confidence = confidence;
	    f->amplitude = amplitude;
	    f->flags = flags;
	    f->reserved = 0;
	}
	res->nframes++;
	if ( !suppressed ) {
	    /* what databits_decode_ascii8 would write (databits_ascii.c:127-136) */
	    if ( res->nbytes < res->bytes_cap )
		res->bytes[res->nbytes] = (uint8_t)(bits & 0xFF);
	    res->nbytes++;
	}
    }

    if ( carrier )						/* :1469-1474 */
	push_episode(res, episode_first_frame, nframes_decoded, carrier_nsamples,
		confidence_total, amplitude_total, 2, episode_b_mark);

    free(buf);
    ofsk_plan_destroy(p);
    return 0;
}
