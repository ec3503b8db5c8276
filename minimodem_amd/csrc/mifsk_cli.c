/*
 * mifsk_cli.c -- `minimodem --rx --file` / `--tx --file` over libmifsk.so's BATCH entry.
 *
 * The reference's main() (src/minimodem.c:455-1481) parses the command line,
 * then for --rx runs its receive loop (:1137-1463) around fsk_find_frame() one
 * search at a time.  This program keeps the command line and replaces the loop:
 *
 *     read the whole file -> mifsk_demod_batch_host() (a batch of one stream: the
 *     receive loop runs on the MI355X) -> mifsk_stream_text() -> stdout / stderr
 *
 * i.e. it is the binding a maintainer would put behind `--rx --file`, as code.
 * --tx --file is served by the host transmitter (mifsk_tx_synthesize), so the
 * reference's own tests/self-test script runs against this binary unchanged
 * (MINIMODEM=.../minimodem_mifsk_batch; tests/test_gpu_cli.py).
 *
 * Plain C over include/mifsk.h; no audio back-ends (ALSA / Pulse / sndio):
 * --file is required.  Options follow src/minimodem.c:591-972; those that only
 * concern live audio are rejected.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <float.h>
#include <getopt.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mifsk.h"

enum {
    OPT_MSBFIRST = 256, OPT_STARTBITS, OPT_STOPBITS, OPT_INVERT_START_STOP, OPT_SYNC_BYTE, OPT_LUT,
    OPT_FLOAT_SAMPLES, OPT_RX_ONE, OPT_BINARY_OUTPUT, OPT_BINARY_RAW, OPT_PRINT_FILTER, OPT_XRXNOISE,
    OPT_RING_EXACT
};

static void usage( void )
{
    fprintf(stderr,
	"usage: minimodem_mifsk_batch --rx|--tx --file FILE [options] {baudmode}\n"
	"       (options of minimodem 0.24 that apply to audio files; see `minimodem --help`)\n");
    exit(1);
}

static void *read_whole_file( const char *path, size_t *len )
{
    FILE *f = fopen(path, "rb");
    if ( !f )
	return NULL;
    size_t cap = 1 << 20, n = 0;
    unsigned char *buf = malloc(cap);
    for (;;) {
	if ( n == cap ) {
	    cap *= 2;
	    buf = realloc(buf, cap);
	}
	if ( !buf )
	    break;
	size_t r = fread(buf + n, 1, cap - n, f);
	n += r;
	if ( r == 0 )
	    break;
    }
    fclose(f);
    *len = n;
    return buf;
}

static void put_le( unsigned char *p, unsigned long v, int nbytes )
{
    for ( int i = 0; i < nbytes; i++ )
	p[i] = (unsigned char)( v >> ( 8 * i ) );
}

/* the header libsndfile writes for a mono WAV (simpleaudio-sndfile.c:113-160);
 * float files carry the 'fact' and 'PEAK' chunks in libsndfile, which readers skip --
 * the sample data is what the tests compare */
static int write_wav( const char *path, const float *x, size_t n, unsigned rate, int is_float )
{
    FILE *f = fopen(path, "wb");
    if ( !f )
	return -1;
    const unsigned bytes = is_float ? 4 : 2;
    unsigned char h[44];
    memcpy(h, "RIFF", 4);
    put_le(h + 4, 36 + n * bytes, 4);
    memcpy(h + 8, "WAVEfmt ", 8);
    put_le(h + 16, 16, 4);
    put_le(h + 20, is_float ? 3 : 1, 2);
    put_le(h + 22, 1, 2);
    put_le(h + 24, rate, 4);
    put_le(h + 28, rate * bytes, 4);
    put_le(h + 32, bytes, 2);
    put_le(h + 34, bytes * 8, 2);
    memcpy(h + 36, "data", 4);
    put_le(h + 40, n * bytes, 4);
    fwrite(h, 1, 44, f);
    if ( is_float ) {
	fwrite(x, 4, n, f);
    } else {
	for ( size_t i = 0; i < n; i++ ) {
	    /* mifsk_tx_synthesize(as_s16) returns value / 32768: exact */
	    short s = (short)lrintf(x[i] * 32768.0f);
	    unsigned char b[2];
	    put_le(b, (unsigned short)s, 2);
	    fwrite(b, 1, 2, f);
	}
    }
    return fclose(f);
}

int main( int argc, char *argv[] )
{
    int tx_mode = -1;
    const char *filename = NULL;
    mifsk_modem_args a;
    mifsk_modem_args_default(&a);
    float tx_amplitude = 1.0f;
    unsigned lut = 4096;
    int float_samples = 0, quiet = 0, print_filter = 0, ring_exact = 0;
    float rxnoise = 0.0f;

    static struct option long_options[] = {
	{ "tx", 0, 0, 't' }, { "transmit", 0, 0, 't' }, { "write", 0, 0, 't' },
	{ "rx", 0, 0, 'r' }, { "receive", 0, 0, 'r' }, { "read", 0, 0, 'r' },
	{ "confidence", 1, 0, 'c' }, { "limit", 1, 0, 'l' }, { "auto-carrier", 0, 0, 'a' },
	{ "inverted", 0, 0, 'i' }, { "ascii", 0, 0, '8' }, { "baudot", 0, 0, '5' },
	{ "msb-first", 0, 0, OPT_MSBFIRST }, { "file", 1, 0, 'f' }, { "bandwidth", 1, 0, 'b' },
	{ "volume", 1, 0, 'v' }, { "mark", 1, 0, 'M' }, { "space", 1, 0, 'S' },
	{ "startbits", 1, 0, OPT_STARTBITS }, { "stopbits", 1, 0, OPT_STOPBITS },
	{ "invert-start-stop", 0, 0, OPT_INVERT_START_STOP }, { "sync-byte", 1, 0, OPT_SYNC_BYTE },
	{ "quiet", 0, 0, 'q' }, { "samplerate", 1, 0, 'R' }, { "lut", 1, 0, OPT_LUT },
	{ "float-samples", 0, 0, OPT_FLOAT_SAMPLES }, { "rx-one", 0, 0, OPT_RX_ONE },
	{ "binary-output", 0, 0, OPT_BINARY_OUTPUT }, { "binary-raw", 1, 0, OPT_BINARY_RAW },
	{ "print-filter", 0, 0, OPT_PRINT_FILTER }, { "Xrxnoise", 1, 0, OPT_XRXNOISE },
	{ "ring-exact", 0, 0, OPT_RING_EXACT },
	{ 0 }
    };
    int c;
    while ( ( c = getopt_long(argc, argv, "trc:l:ai875f:b:v:M:S:qR:", long_options, NULL) ) != -1 ) {
	switch ( c ) {
	case 't': tx_mode = 1; break;
	case 'r': tx_mode = 0; break;
	case 'c': a.confidence_threshold = (float)atof(optarg); break;
	case 'l': a.search_limit = (float)atof(optarg); break;
	case 'a': a.auto_carrier_threshold = 0.001f; break;
	case 'i': a.inverted_freqs = 1; break;
	case 'f': filename = optarg; break;
	case '8': a.n_data_bits = 8; break;
	case '7': a.n_data_bits = 7; break;
	case '5': a.n_data_bits = 5; a.baudot = 1; break;
	case OPT_MSBFIRST: a.msb_first = 1; break;
	case 'b': a.band_width = (float)atof(optarg); break;
	case 'v': tx_amplitude = optarg[0] == 'E' ? FLT_EPSILON : (float)atof(optarg); break;
	case 'M': a.mark_f = (float)atof(optarg); break;
	case 'S': a.space_f = (float)atof(optarg); break;
	case OPT_STARTBITS: a.nstartbits = atoi(optarg); break;
	case OPT_STOPBITS: a.nstopbits = (float)atof(optarg); break;
	case OPT_INVERT_START_STOP: a.invert_start_stop = 1; break;
	case OPT_SYNC_BYTE: a.have_sync_byte = 1; a.sync_byte = strtol(optarg, NULL, 0); break;
	case 'q': quiet = 1; break;
	case 'R': a.sample_rate = (unsigned)atoi(optarg); break;
	case OPT_LUT: lut = (unsigned)atoi(optarg); break;
	case OPT_FLOAT_SAMPLES: float_samples = 1; break;
	case OPT_RX_ONE: a.rx_one = 1; break;
	case OPT_BINARY_OUTPUT: a.binary_output = 1; break;
	case OPT_BINARY_RAW: a.binary_raw_nbits = atoi(optarg); break;
	case OPT_PRINT_FILTER: print_filter = 1; break;
	case OPT_XRXNOISE: rxnoise = (float)atof(optarg); break;
	case OPT_RING_EXACT: ring_exact = 1; break;
	default: usage();
	}
    }
    if ( tx_mode < 0 || !filename || optind + 1 != argc )
	usage();
    a.baudmode = argv[optind];

    mifsk_rx_config cfg;
    int rc = mifsk_rx_config_init(&cfg, &a);
    if ( rc ) {
	fprintf(stderr, "E: invalid modem configuration (%d)\n", rc);
	return 1;
    }

    if ( tx_mode ) {
	/* fsk_transmit_stdin (minimodem.c:114-250): characters -> data words */
	mifsk_databits *enc = NULL;
	if ( mifsk_databits_create(&enc, cfg.decoder) )
	    return 1;
	size_t cap = 4096, nwords = 0;
	uint8_t *words = malloc(cap);
	int ch;
	while ( ( ch = getchar() ) != EOF ) {
	    unsigned w[2];
	    unsigned n = mifsk_databits_encode(enc, w, (char)ch);
	    if ( n == 0 && cfg.decoder == MIFSK_DECODE_BAUDOT )
		fprintf(stderr, "W: baudot skipping non-encodable character '%c' 0x%02x\n", ch, ch);
	    for ( unsigned j = 0; j < n; j++ ) {
		if ( nwords == cap )
		    words = realloc(words, cap *= 2);
		words[nwords++] = (uint8_t)w[j];
	    }
	}
	mifsk_databits_destroy(enc);
	long n = mifsk_tx_synthesize(&cfg, words, nwords, lut, tx_amplitude, 0, !float_samples, NULL, 0);
	if ( n < 0 ) {
	    fprintf(stderr, "E: transmit failed (%ld)\n", n);
	    return 1;
	}
	float *x = malloc(( n ? (size_t)n : 1 ) * sizeof(float));
	mifsk_tx_synthesize(&cfg, words, nwords, lut, tx_amplitude, 0, !float_samples, x, (size_t)n);
	rc = nwords ? write_wav(filename, x, (size_t)n, cfg.sample_rate, float_samples)
		    : write_wav(filename, x, 0, cfg.sample_rate, float_samples);
	return rc ? 1 : 0;
    }

    /* ---- receive: the whole file as a batch of one ------------------------------ */
    size_t len = 0;
    unsigned char *file = read_whole_file(filename, &len);
    if ( !file ) {
	perror(filename);
	return 1;
    }
    mifsk_wav_info wi;
    rc = mifsk_wav_parse(file, len, &wi);
    if ( rc ) {
	fprintf(stderr, "E: %s: not a mono PCM16 / float32 WAV file (%d)\n", filename, rc);
	return 1;
    }
    if ( wi.sample_rate != cfg.sample_rate ) {
	/* the reference takes the rate from the file (minimodem.c:1021-1032) */
	a.sample_rate = wi.sample_rate;
	if ( mifsk_rx_config_init(&cfg, &a) )
	    return 1;
    }
    const size_t n = wi.nframes;
    float *x = malloc(( n ? n : 1 ) * sizeof(float));
    const unsigned char *d = file + wi.data_offset;
    if ( wi.is_float ) {
	memcpy(x, d, n * sizeof(float));
    } else {
	for ( size_t i = 0; i < n; i++ ) {		/* sf_readf_float on 16-bit input: / 32768 */
	    short s = (short)( d[2 * i] | ( d[2 * i + 1] << 8 ) );
	    x[i] = (float)s / 32768.0f;
	}
    }
    if ( rxnoise != 0.0f ) {				/* simpleaudio-sndfile.c:64-69 */
	const float f = rxnoise * 2;
	for ( size_t i = 0; i < n; i++ )
	    x[i] += ( 0 - 0.5f ) * f;			/* rand()/RAND_MAX is an integer division */
    }

    mifsk_ctx *ctx = NULL;
    rc = mifsk_ctx_create(&ctx, -1);
    if ( rc ) {
	fprintf(stderr, "E: no MI355X available (%d); this program has no CPU receive path\n", rc);
	return 1;
    }
    const size_t fcap = mifsk_max_frames(&cfg, n) + 8, ecap = fcap / 8 + 64;
    uint64_t *bits = calloc(fcap, sizeof(uint64_t));
    mifsk_episode *eps = calloc(ecap, sizeof(mifsk_episode));
    uint32_t nframes = 0, neps = 0, status = 0;
    mifsk_demod_io io;
    memset(&io, 0, sizeof(io));
    io.d_samples = x;
    io.stream_stride = n;
    io.nsamples = (uint32_t)n;
    io.nstreams = 1;
    io.d_bits = bits;
    io.d_nframes = &nframes;
    io.frames_cap = fcap;
    io.d_episodes = eps;
    io.d_nepisodes = &neps;
    io.episodes_cap = ecap;
    io.d_status = &status;
    io.flags = ring_exact ? MIFSK_IO_RING_EXACT : 0;
    rc = mifsk_demod_batch_host(ctx, &cfg, &io);
    if ( rc ) {
	fprintf(stderr, "E: mifsk_demod_batch_host failed (%d)\n", rc);
	return 1;
    }
    if ( nframes > fcap ) nframes = (uint32_t)fcap;
    if ( neps > ecap ) neps = (uint32_t)ecap;

    const unsigned tflags = ( print_filter ? MIFSK_TEXT_PRINT_FILTER : 0 ) | ( quiet ? MIFSK_TEXT_QUIET : 0 );
    size_t out_len = 0, err_len = 0;
    size_t out_cap = 64 + 320 * (size_t)( nframes ? nframes : 1 ), err_cap = 256 + 512 * (size_t)( neps ? neps : 1 );
    char *out = malloc(out_cap), *err = malloc(err_cap);
    rc = mifsk_stream_text(&cfg, bits, nframes, eps, neps, tflags, out, out_cap, &out_len,
			   err, err_cap, &err_len);
    if ( rc && rc != -ENOSPC )
	return 1;
    fwrite(err, 1, err_len < err_cap ? err_len : err_cap, stderr);
    fwrite(out, 1, out_len < out_cap ? out_len : out_cap, stdout);
    mifsk_ctx_destroy(ctx);
    return 0;
}
