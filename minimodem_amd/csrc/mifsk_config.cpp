// mifsk_config.cpp -- host-side derivation of the receive configuration.
//
// Everything the reference's main() computes between option parsing and the
// receive loop, as one pure function (include/mifsk.h: mifsk_rx_config_init).
// All arithmetic is done in C `float` with the reference's truncating
// conversions, because the device consumes the resulting INTEGERS (window
// lengths, offsets, search grid) and they must match the reference's exactly:
//
//   baud-mode presets, tone/bandwidth defaults   src/minimodem.c:819-934
//   start/stop defaults, frame_n_bits truncation src/minimodem.c:936-947
//   bandwidth clamp, search-limit sanitising     src/minimodem.c:959-965
//   plan bins                                    src/fsk.c:52-64
//   samples-per-bit, buffer, overscan, frame     src/minimodem.c:1037-1113
//   expect strings                               src/minimodem.c:442-487,1115-1131
//   search grid                                  src/minimodem.c:1236-1263,1366
//   bit windows inside fsk_find_frame            src/fsk.c:183,204,465
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <strings.h>

#include "mifsk.h"

namespace {

struct Preset {
    const char	*prefix;	// matched case-insensitively over `len` chars
    size_t	len;
    float	rate;
    int		n_data_bits;	// 0: keep the caller's (or 8)
    int		only_if_unset;	// n_data_bits applies only when the caller gave none
    float	nstopbits;	// <0: leave
    int		stop_only_if_unset;
    int		nstartbits;	// <0: leave
    float	mark, space;	// 0: leave
    int		decoder;	// <0: leave
};

// src/minimodem.c:819-881.  Rates are float conversions of the same double
// literals the reference uses.
const Preset kPresets[] = {
    { "rtty",   5, (float)45.45,             5, 1,  1.5f, 1, -1, 0.f, 0.f, MIFSK_DECODE_BAUDOT },
    { "tdd",    4, (float)45.45,             5, 1,  2.0f, 1, -1, 1400.f, 1800.f, MIFSK_DECODE_BAUDOT },
    { "same",   5, (float)(520.0 + 5 / 6.0), 8, 0,  0.0f, 0,  0, (float)(2083.0 + 1 / 3.0), 1562.5f, -1 },
    { "caller", 6, 1200.f,                   8, 0, -1.0f, 0, -1, 0.f, 0.f, MIFSK_DECODE_CALLERID },
    { "uic",    3, 600.f,                   39, 0,  0.0f, 0,  8, 1300.f, 1700.f, MIFSK_DECODE_UIC_GROUND },
    { "V.21",   4, 300.f,                    8, 0, -1.0f, 0, -1, 980.f, 1180.f, -1 },
};

const char kUicExpect[] = "11110010ddddddddddddddddddddddddddddddddddddddd";

// "isddddddddp": previous stop, start bits, data bits, stop (minimodem.c:442-487)
unsigned build_expect(char *dst, const mifsk_rx_config &c, bool with_value,
		      unsigned long long value)
{
    const char start_c = c.invert_start_stop ? '1' : '0';
    const char stop_c  = c.invert_start_stop ? '0' : '1';
    unsigned n = 0;
    const bool framed = c.nstopbits != 0.0f;
    if ( framed )
	dst[n++] = stop_c;
    for ( int i = 0; i < c.nstartbits; i++ )
	dst[n++] = start_c;
    for ( unsigned i = 0; i < c.n_data_bits; i++ )
	dst[n++] = with_value ? char('0' + ((value >> i) & 1ULL)) : 'd';
    if ( framed )
	dst[n++] = stop_c;
    dst[n] = '\0';
    return n;
}

} // namespace

extern "C" void mifsk_modem_args_default( mifsk_modem_args *a )
{
    std::memset(a, 0, sizeof(*a));
    a->baudmode = "1200";
    a->nstartbits = -1;
    a->nstopbits = -1.0f;
    a->sync_byte = -1;
    a->confidence_threshold = -1.0f;	// -> 1.5   (minimodem.c:519)
    a->search_limit = -1.0f;		// -> 2.3   (minimodem.c:528)
}

extern "C" int mifsk_rx_config_init( mifsk_rx_config *cfg, const mifsk_modem_args *a )
{
    if ( !cfg || !a )
	return -EINVAL;
    mifsk_rx_config c;
    std::memset(&c, 0, sizeof(c));

    const char *mode = a->baudmode ? a->baudmode : "";
    float rate = 0.0f;
    float mark = a->mark_f, space = a->space_f, bw = a->band_width;
    int n_data = a->n_data_bits > 0 ? a->n_data_bits : 0;
    int nstart = a->nstartbits < 0 ? -1 : a->nstartbits;
    float nstop = a->nstopbits < 0.0f ? -1.0f : a->nstopbits;
    int decoder = a->baudot ? MIFSK_DECODE_BAUDOT : MIFSK_DECODE_ASCII8;
    bool sync = a->have_sync_byte != 0;
    unsigned long long sync_byte = sync ? (unsigned long long)a->sync_byte : ~0ULL;
    bool uic = false;

    const Preset *hit = nullptr;
    for ( const Preset &p : kPresets )
	if ( strncasecmp(mode, p.prefix, p.len) == 0 ) {
	    hit = &p;
	    break;
	}
    if ( hit ) {
	rate = hit->rate;
	if ( !hit->only_if_unset || n_data == 0 )
	    n_data = hit->n_data_bits;
	if ( hit->nstopbits >= 0.0f && ( !hit->stop_only_if_unset || nstop < 0.0f ) )
	    nstop = hit->nstopbits;
	if ( hit->nstartbits >= 0 )
	    nstart = hit->nstartbits;
	if ( hit->mark != 0.f ) {
	    mark = hit->mark;
	    space = hit->space;
	}
	if ( hit->decoder >= 0 )
	    decoder = hit->decoder;
	if ( hit->prefix[0] == 's' ) {		// SAME (minimodem.c:837-848)
	    sync = true;
	    sync_byte = 0xAB;
	    bw = rate;
	}
	if ( hit->prefix[0] == 'u' ) {		// UIC-751-3 (minimodem.c:859-876)
	    uic = true;
	    if ( std::strlen(mode) > 4 && ( mode[4] == 't' || mode[4] == 'T' ) )
		decoder = MIFSK_DECODE_UIC_TRAIN;
	}
    } else {
	rate = (float)std::atof(mode);		// minimodem.c:883-885
	if ( n_data == 0 )
	    n_data = 8;
    }
    if ( rate == 0.0f )
	return -EINVAL;

    if ( a->binary_output || a->binary_raw_nbits )	// minimodem.c:891-898
	decoder = MIFSK_DECODE_BINARY;
    if ( a->binary_raw_nbits ) {
	nstart = 0;
	nstop = 0.0f;
	n_data = a->binary_raw_nbits;
    }

    // tone and bandwidth defaults by rate class (minimodem.c:900-934)
    int shift;
    if ( rate >= 400.f ) {
	shift = -(int)( rate * 5 / 6 );
	if ( mark == 0.f ) mark = rate / 2 + 600;
	if ( space == 0.f ) space = mark - shift;
	if ( bw == 0.f ) bw = 200.f;
    } else if ( rate >= 100.f ) {
	shift = 200;
	if ( mark == 0.f ) mark = 1270.f;
	if ( space == 0.f ) space = mark - shift;
	if ( bw == 0.f ) bw = 50.f;
    } else {
	shift = 170;
	if ( mark == 0.f ) mark = 1585.f;
	if ( space == 0.f ) space = mark - shift;
	if ( bw == 0.f ) bw = 10.f;
    }
    if ( nstart < 0 ) nstart = 1;
    if ( nstop < 0.0f ) nstop = 1.0f;

    // unsigned = unsigned + int + float: the float sum is truncated, so
    // 1.5 stop bits count as 1 (minimodem.c:943; why RTTY reads "5.4% slow")
    const unsigned frame_n_bits = (unsigned)( (float)( (unsigned)n_data + nstart ) + nstop );
    if ( frame_n_bits > MIFSK_MAX_FRAME_BITS || n_data <= 0 )
	return -EINVAL;

    if ( a->inverted_freqs ) {
	const float t = mark;
	mark = space;
	space = t;
    }
    if ( bw > rate )
	bw = rate;

    float thr = a->confidence_threshold < 0.0f ? 1.5f : a->confidence_threshold;
    float lim = a->search_limit < 0.0f ? 2.3f : a->search_limit;
    if ( lim < thr )
	lim = thr;

    c.sample_rate = a->sample_rate ? a->sample_rate : 48000u;
    c.data_rate = rate;
    c.mark_f = mark;
    c.space_f = space;
    c.band_width = bw;
    c.n_data_bits = (unsigned)n_data;
    c.nstartbits = nstart;
    c.nstopbits = nstop;
    c.invert_start_stop = a->invert_start_stop;
    c.msb_first = a->msb_first;
    c.do_rx_sync = sync ? 1 : 0;
    c.sync_byte = sync_byte;
    c.decoder = decoder;
    c.rx_one = a->rx_one;
    c.confidence_threshold = thr;
    c.search_limit = lim;
    c.auto_carrier_threshold = a->auto_carrier_threshold;
    c.autodetect_shift = shift;
    c.inverted_freqs = a->inverted_freqs;
    c.frame_n_bits = frame_n_bits;

    // plan: src/fsk.c:52-64
    {
	const float sr = (float)c.sample_rate;
	const float half = bw / 2.0f;
	c.fftsize = (int)( (sr + half) / bw );
	if ( c.fftsize < 2 )
	    return -EINVAL;
	c.nbands = (unsigned)( c.fftsize / 2 + 1 );
	c.b_mark = (unsigned)( (mark + half) / bw );
	c.b_space = (unsigned)( (space + half) / bw );
	if ( c.b_mark >= c.nbands || c.b_space >= c.nbands )
	    return -EINVAL;
    }

    const float spb = c.sample_rate / rate;			// minimodem.c:1037
    c.nsamples_per_bit = spb;

    {	// minimodem.c:1056-1070
	const unsigned nbits = 1u + (unsigned)nstart + (unsigned)n_data + 1u;
	size_t sz = (size_t)( std::ceil(spb) * (float)( nbits + 1 ) );
	sz *= 2;
	if ( sz < c.sample_rate / 12 )
	    sz = c.sample_rate / 12;
	c.samplebuf_size = (unsigned)sz;
    }

    {	// minimodem.c:1091-1113
	const float overscan = 0.5f;
	unsigned os = (unsigned)( spb * overscan + 0.5f );
	if ( os == 0 )
	    os = 1;
	c.nsamples_overscan = os;
	c.frame_nsamples = (unsigned)( spb * (float)frame_n_bits + 0.5f );
    }

    if ( uic ) {
	std::strcpy(c.expect_data, kUicExpect);
	c.expect_n_bits = 47;
    } else {
	c.expect_n_bits = build_expect(c.expect_data, c, false, 0);
    }
    if ( c.do_rx_sync && (long long)c.sync_byte >= 0 )
	build_expect(c.expect_sync, c, true, c.sync_byte);
    else
	std::strcpy(c.expect_sync, c.expect_data);
    if ( c.expect_n_bits == 0 || c.expect_n_bits > MIFSK_MAX_FRAME_BITS )
	return -EINVAL;
    c.expect_nsamples = (unsigned)( spb * (float)c.expect_n_bits );	// minimodem.c:1131

    for ( int lock = 0; lock < 2; lock++ ) {		// minimodem.c:1236-1263,1366
	unsigned reach = lock ? (unsigned)( spb * 0.75f + 0.5f ) : (unsigned)spb;
	reach += c.nsamples_overscan;
	c.try_max[lock] = reach;
	c.try_first[lock] = lock ? c.nsamples_overscan : 0u;
	c.try_step[lock] = reach / 3 ? reach / 3 : 1u;
	c.try_step_fine[lock] = reach / 8 ? reach / 8 : 1u;
    }

    // fsk_find_frame is handed expect_nsamples as its frame length
    // (minimodem.c:1265) and re-derives samples-per-bit from it (fsk.c:465)
    const float fspb = (float)c.expect_nsamples / (float)(int)c.expect_n_bits;
    c.find_samples_per_bit = fspb;
    c.bit_nsamples = (unsigned)( fspb + 0.5f );
    for ( unsigned k = 0; k < c.expect_n_bits; k++ )
	c.bit_offset[k] = (unsigned)( fspb * (float)(int)k + 0.5f );

    *cfg = c;
    return 0;
}

extern "C" size_t mifsk_max_frames( const mifsk_rx_config *cfg, size_t nsamples )
{
    // every decoded frame advances the cursor by at least
    // frame_nsamples - overscan samples (minimodem.c:1407 with frame_start 0)
    size_t adv = cfg->frame_nsamples > cfg->nsamples_overscan
		? cfg->frame_nsamples - cfg->nsamples_overscan : 1;
    return nsamples / adv + 2;
}

extern "C" size_t mifsk_stream_padding( const mifsk_rx_config *cfg )
{
    // furthest sample a search can touch past the cursor: last candidate
    // position + last bit window (fsk.c:480,204,206)
    size_t reach = cfg->try_max[0] > cfg->try_max[1] ? cfg->try_max[0] : cfg->try_max[1];
    size_t last = cfg->bit_offset[cfg->expect_n_bits - 1] + cfg->bit_nsamples;
    return ( reach + last + 3 ) & ~(size_t)3;
}
