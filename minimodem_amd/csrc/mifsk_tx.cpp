// mifsk_tx.cpp -- host-side synthetic-input generator (NOT on the hot path).
//
// The benchmark and the parity tests need FSK audio that is sample-for-sample
// what `minimodem --tx --file` would have written, so that the receive path is
// exercised on the reference's own signal shape.  This restates, for in-memory
// buffers:
//   framing        src/minimodem.c:81-112 (fsk_transmit_frame), :114-250
//                  (leader / sync preamble / data / trailer, no flush for files)
//   tone synthesis src/simple-tone-generator.c:35-175 (phase-continuous sine,
//                  optional lookup table, f32 phase accumulator)
// It runs on the host because it is setup work (SURVEY 8(f4) lists a device
// version as "next"); everything it produces is checked bit-for-bit against
// the reference program's WAV output in tests/test_tx_synth.py.
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "mifsk.h"

namespace {

struct ToneGen {
    unsigned		table_len;
    float		mag;
    std::vector<float>	tab_f;
    std::vector<short>	tab_s;
    unsigned short	mag_s;
    float		cphase;
    unsigned		sample_rate;
    int			s16;		// emit what an S16 file reads back as

    void init( unsigned len, float m, unsigned sr, int as_s16 )
    {
	table_len = len;
	mag = m;
	sample_rate = sr;
	s16 = as_s16;
	cphase = 0.0f;
	// simple-tone-generator.c:50-60
	mag_s = (unsigned short)( 32767.0f * mag + 0.5f );
	if ( mag > 1.0f )
	    mag_s = 32767;
	if ( mag_s < 1 )
	    mag_s = 1;
	tab_f.resize(len);
	tab_s.resize(len);
	for ( unsigned i = 0; i < len; i++ ) {
	    const float ang = (float)M_PI * 2 * i / len;
	    tab_s[i] = (short)lroundf(mag_s * sinf(ang));
	    tab_f[i] = mag * sinf(ang);
	}
    }

    // simple-tone-generator.c:106-175 ; returns samples appended
    size_t tone( float freq, size_t nsamples, float *out, size_t cap, size_t pos )
    {
	if ( freq != 0.0f ) {
	    const float wave_nsamples = sample_rate / freq;
	    for ( size_t i = 0; i < nsamples; i++ ) {
		const float turns = (float)i / wave_nsamples + cphase;
		float v;
		if ( table_len ) {
		    int t = (int)( (float)table_len * turns + 0.5f );
		    t %= (int)table_len;
		    v = s16 ? (float)tab_s[t] / 32768.0f : tab_f[t];
		} else {
		    const float rad = (float)M_PI * 2 * turns;
		    v = s16 ? (float)(short)lroundf(mag_s * sinf(rad)) / 32768.0f
			    : mag * sinf(rad);
		}
		if ( out && pos + i < cap )
		    out[pos + i] = v;
	    }
	    cphase = fmodf(cphase + (float)nsamples / wave_nsamples, 1.0f);
	} else {
	    for ( size_t i = 0; i < nsamples; i++ )
		if ( out && pos + i < cap )
		    out[pos + i] = 0.0f;
	    cphase = 0.0f;
	}
	return nsamples;
    }
};

} // namespace

// Synthesize one stream.  `words` are the data words (what the databits
// encoder would have produced: bytes for ascii, 5-bit codes for baudot).
// Returns the number of samples the stream has (also when out == NULL or
// out_cap is too small: nothing past out_cap is written), or -errno.
extern "C" long mifsk_tx_synthesize( const mifsk_rx_config *cfg, const uint8_t *words,
	size_t nwords, unsigned sin_table_len, float amplitude,
	unsigned leading_silence, int as_s16, float *out, size_t out_cap )
{
    if ( !cfg || ( nwords && !words ) || !( amplitude > 0.0f ) )
	return -EINVAL;
    ToneGen g;
    g.init(sin_table_len, amplitude, cfg->sample_rate, as_s16);

    const float mark = cfg->mark_f, space = cfg->space_f;
    const int inv = cfg->invert_start_stop;
    // minimodem.c:131-132
    const size_t sample_rate = cfg->sample_rate;
    const size_t bit_nsamples = (size_t)( sample_rate / cfg->data_rate + 0.5f );
    const float nstart = (float)cfg->nstartbits;
    const float nstop = cfg->nstopbits;
    const unsigned ndata = cfg->n_data_bits;

    size_t pos = 0;
    if ( leading_silence )
	pos += g.tone(0.0f, leading_silence, out, out_cap, pos);

    auto frame = [&]( unsigned bits, int msb_first ) {	// minimodem.c:81-112
	if ( nstart > 0 )
	    pos += g.tone(inv ? mark : space, (size_t)( bit_nsamples * nstart ), out, out_cap, pos);
	for ( unsigned i = 0; i < ndata; i++ ) {
	    const unsigned bit = msb_first ? ( bits >> ( ndata - i - 1 ) ) & 1u : ( bits >> i ) & 1u;
	    pos += g.tone(bit ? mark : space, bit_nsamples, out, out_cap, pos);
	}
	if ( nstop > 0 )
	    pos += g.tone(inv ? space : mark, (size_t)( bit_nsamples * nstop ), out, out_cap, pos);
    };

    if ( nwords ) {
	// leader: two mark bits unless the mode has no start bit (minimodem.c:207-213,950-951)
	const int leader = cfg->nstartbits == 0 ? 0 : 2;
	for ( int j = 0; j < leader; j++ )
	    pos += g.tone(inv ? space : mark, bit_nsamples, out, out_cap, pos);
	// sync preamble: 16 frames of the sync byte (minimodem.c:214-222,718,844)
	if ( cfg->do_rx_sync )
	    for ( int j = 0; j < 16; j++ )
		frame((unsigned)cfg->sync_byte, 0);
	for ( size_t w = 0; w < nwords; w++ )
	    frame(words[w], cfg->msb_first);
	// trailer: two mark bits (minimodem.c:59-74,249); no flush for files (:137-140)
	for ( int j = 0; j < 2; j++ )
	    pos += g.tone(mark, bit_nsamples, out, out_cap, pos);
    }
    return (long)pos;
}
