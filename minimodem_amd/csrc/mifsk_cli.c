/*
 * mifsk_cli.c -- `minimodem --rx --file` / `--tx --file` over libmifsk.so's BATCH entry.
 *
 * The reference's main() (src/minimodem.c:455-1481) parses the command line,
 * then for --rx runs its receive loop (:1137-1463) around fsk_find_frame() one
 * search at a time.  This program keeps the command line and replaces the loop:
 *
 *     every --file of the command line -> mifsk_demod_files() (headers parsed, raw
 *     samples through pinned memory to the MI355X, PCM16 -> float and --Xrxnoise on the
 *     device, the receive loop over the batch) -> mifsk_stream_text() per file ->
 *     stdout / stderr
 *
 * i.e. it is the binding a maintainer would put behind `--rx --file`, as code;
 * `--file a.wav --file b.wav ...` decodes a list of files as one batch.
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
#include <time.h>

#include "mifsk.h"

enum {
    OPT_MSBFIRST = 256, OPT_STARTBITS, OPT_STOPBITS, OPT_INVERT_START_STOP, OPT_SYNC_BYTE, OPT_LUT,
    OPT_FLOAT_SAMPLES, OPT_RX_ONE, OPT_BINARY_OUTPUT, OPT_BINARY_RAW, OPT_PRINT_FILTER, OPT_XRXNOISE,
    OPT_RING_EXACT, OPT_FLAT, OPT_STATS
};

#ifndef MIFSK_CLI_NO_MAIN
static void usage( void )
{
    fprintf(stderr,
	"usage: minimodem_mifsk_batch --rx --file FILE [--file FILE ...] [options] {baudmode}\n"
	"       minimodem_mifsk_batch --tx --file FILE [options] {baudmode}\n"
	"       --flat         flat instead of ring-exact addressing past the end of the stream\n"
	"       --batch-stats  print what the host pipeline moved\n"
	"       (options of minimodem 0.24 that apply to audio files; see `minimodem --help`)\n");
    exit(1);
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

#endif /* MIFSK_CLI_NO_MAIN */

/* Everything `minimodem --rx --file F [--file G ...]` prints, through the batch entry: the
 * files as ONE batch on the MI355X, then each file's text (stdout) and CARRIER / NOCARRIER
 * lines (stderr) in command-line order.  `a` is the command line as data (what was given;
 * zero / negative = not given, as the options leave minimodem's own variables).  Returns the
 * process's exit status.  Also what integration/minimodem-rx-batch.patch calls from inside the
 * reference's own main() (there with one file: the reference takes one --file). */
enum { MIFSK_CLI_FLAT = 1, MIFSK_CLI_QUIET = 2, MIFSK_CLI_PRINT_FILTER = 4, MIFSK_CLI_STATS = 8,
       /* a file that is not a mono PCM16 / float32 WAV is NOT an error of this function: it
	* returns MIFSK_CLI_NOT_HANDLED before anything is printed or any device is touched, and the
	* caller -- the reference's main(), which reads FLAC, AIFF, AU, 24-bit PCM ... through
	* libsndfile -- goes on into its own receive loop */
       MIFSK_CLI_FALLTHROUGH = 16 };
#define MIFSK_CLI_NOT_HANDLED	(-2)

static double cli_now( void )
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* 0: `path` starts with a RIFF/WAVE header the batch path takes; otherwise mifsk_wav_parse's
 * verdict (-EINVAL / -ENOTSUP) or -errno */
static int probe_wav( const char *path )
{
    unsigned char head[4096];
    FILE *f = fopen(path, "rb");
    if ( !f )
	return -errno;
    const size_t n = fread(head, 1, sizeof(head), f);
    fclose(f);
    mifsk_wav_info info;
    return mifsk_wav_parse(head, n, &info);
}

int mifsk_cli_rx_files( const mifsk_modem_args *a, const char *const *files, int nfiles, float rxnoise,
	unsigned flags )
{
    /* ---- receive: every --file of the command line as ONE batch --------------------
     * (headers parsed, raw samples pread() into pinned memory, copied and converted on
     * the device, demodulated chunk by chunk: mifsk_demod_files) */
    const int flat = ( flags & MIFSK_CLI_FLAT ) != 0, quiet = ( flags & MIFSK_CLI_QUIET ) != 0;
    const int print_filter = ( flags & MIFSK_CLI_PRINT_FILTER ) != 0, stats = ( flags & MIFSK_CLI_STATS ) != 0;
    if ( flags & MIFSK_CLI_FALLTHROUGH ) {
	for ( int i = 0; i < nfiles; i++ ) {
	    const int v = probe_wav(files[i]);
	    if ( v == -EINVAL || v == -ENOTSUP )
		return MIFSK_CLI_NOT_HANDLED;
	}
    }
    /* MIFSK_CLI_TIMING=1: where a call's wall time goes, on stderr */
    const int timing = getenv("MIFSK_CLI_TIMING") != NULL;
    const double t_0 = cli_now();
    mifsk_ctx *ctx = NULL;
    int rc = mifsk_ctx_create(&ctx, -1);
    if ( rc ) {
	fprintf(stderr, "E: no MI355X available (%d); this program has no CPU receive path\n", rc);
	return 1;
    }
    const double t_ctx = cli_now();
    mifsk_files *res = NULL;
    rc = mifsk_demod_files(ctx, a, files, nfiles, rxnoise,
			   flat ? 0u : MIFSK_IO_RING_EXACT, &res);
    const double t_demod = cli_now();
    if ( rc ) {
	fprintf(stderr, "E: mifsk_demod_files failed (%d)\n", rc);
	mifsk_ctx_destroy(ctx);
	return 1;
    }
    const unsigned tflags = ( print_filter ? MIFSK_TEXT_PRINT_FILTER : 0 ) | ( quiet ? MIFSK_TEXT_QUIET : 0 );
    int failed = 0;
    for ( int i = 0; i < nfiles; i++ ) {
	const mifsk_file_result *fr = mifsk_files_get(res, i);
	if ( nfiles > 1 && !quiet )
	    fprintf(stderr, "### FILE %s\n", files[i]);
	if ( fr->error ) {
	    if ( fr->error == -EINVAL || fr->error == -ENOTSUP )
		fprintf(stderr, "E: %s: not a mono PCM16 / float32 WAV file (%d)\n", files[i], fr->error);
	    else
		fprintf(stderr, "E: %s: %s\n", files[i], strerror(-fr->error));
	    failed = 1;
	    continue;
	}
	if ( fr->status & ( MIFSK_STREAM_ABORTED | MIFSK_STREAM_FRAMES_TRUNCATED | MIFSK_STREAM_EPISODES_TRUNCATED ) ) {
	    fprintf(stderr, "E: %s: receive loop %s (status %u)\n", files[i],
		    fr->status & MIFSK_STREAM_ABORTED ? "aborted" : "ran out of output room", fr->status);
	    failed = 1;
	}
	size_t out_len = 0, err_len = 0;
	const size_t out_cap = 64 + 320 * (size_t)( fr->nframes ? fr->nframes : 1 );
	const size_t err_cap = 256 + 512 * (size_t)( fr->nepisodes ? fr->nepisodes : 1 );
	char *out = malloc(out_cap), *err = malloc(err_cap);
	if ( !out || !err ) {
	    fprintf(stderr, "E: %s: out of memory\n", files[i]);
	    free(out);
	    free(err);
	    failed = 1;
	    continue;
	}
	rc = mifsk_stream_text(fr->cfg, fr->bits, fr->nframes, fr->episodes, fr->nepisodes, tflags,
			       out, out_cap, &out_len, err, err_cap, &err_len);
	if ( rc && rc != -ENOSPC ) {
	    failed = 1;
	} else {
	    fwrite(err, 1, err_len < err_cap ? err_len : err_cap, stderr);
	    fwrite(out, 1, out_len < out_cap ? out_len : out_cap, stdout);
	    fflush(stdout);
	}
	free(out);
	free(err);
    }
    if ( stats && !quiet ) {
	const mifsk_host_stats *st = mifsk_files_stats(res);
	fprintf(stderr, "### BATCH files=%u chunks=%u h2d=%.1f MB in %.3f s (%.2f GB/s, staging %.3f s)\n",
		st->streams, st->chunks, st->bytes_h2d / 1e6, st->seconds_total,
		st->seconds_total > 0 ? st->bytes_h2d / st->seconds_total / 1e9 : 0.0, st->seconds_staging);
    }
    const double t_text = cli_now();
    mifsk_files_free(res);
    mifsk_ctx_destroy(ctx);
    if ( timing )
	fprintf(stderr, "### TIMING context (HIP start, device) %.1f ms, mifsk_demod_files %.1f ms, text %.1f ms, "
			"teardown %.1f ms\n", 1e3 * ( t_ctx - t_0 ), 1e3 * ( t_demod - t_ctx ),
		1e3 * ( t_text - t_demod ), 1e3 * ( cli_now() - t_text ));
    return failed;
}

#ifndef MIFSK_CLI_NO_MAIN
int main( int argc, char *argv[] )
{
    int tx_mode = -1;
    const char *filename = NULL;
    const char **files = NULL;
    int nfiles = 0;
    mifsk_modem_args a;
    mifsk_modem_args_default(&a);
    float tx_amplitude = 1.0f;
    unsigned lut = 4096;
    /* --rx reads past-the-end samples the way the reference's ring buffer does (RING
     * addressing) unless --flat asks for the faster flat addressing */
    int float_samples = 0, quiet = 0, print_filter = 0, flat = 0, stats = 0;
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
	{ "ring-exact", 0, 0, OPT_RING_EXACT }, { "flat", 0, 0, OPT_FLAT }, { "batch-stats", 0, 0, OPT_STATS },
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
	case 'f':
	    filename = optarg;
	    files = realloc(files, (size_t)( nfiles + 1 ) * sizeof(*files));
	    files[nfiles++] = optarg;
	    break;
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
	case OPT_RING_EXACT: flat = 0; break;
	case OPT_FLAT: flat = 1; break;
	case OPT_STATS: stats = 1; break;
	default: usage();
	}
    }
    if ( tx_mode < 0 || !filename || optind + 1 != argc || ( tx_mode == 1 && nfiles != 1 ) )
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

    return mifsk_cli_rx_files(&a, (const char *const *)files, nfiles, rxnoise,
			      ( flat ? MIFSK_CLI_FLAT : 0 ) | ( quiet ? MIFSK_CLI_QUIET : 0 )
			      | ( print_filter ? MIFSK_CLI_PRINT_FILTER : 0 ) | ( stats ? MIFSK_CLI_STATS : 0 ));
}
#endif /* MIFSK_CLI_NO_MAIN */
