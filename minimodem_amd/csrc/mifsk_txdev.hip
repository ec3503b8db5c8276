// mifsk_txdev.hip -- the step on the other side of the path: the
// reference's transmitter for a whole batch of streams on the device -- framing
// (src/minimodem.c:81-250: leader, sync preamble, start / data / stop tones,
// trailer) and the phase-continuous table-lookup tone generator
// (src/simple-tone-generator.c:106-175).  Bit-identical to csrc/mifsk_tx.cpp,
// which is pinned to the WAV files the reference writes.  The sine TABLE is built
// on the host with the host's sinf; --lut=0 (a sinf per sample,
// simple-tone-generator.c:134,155) uses the restatement of glibc's sinf in
// mifsk_sinf.h, pinned to the C library over every non-negative float.
// gfx950 only.
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cmath>
#include <cstdint>
#include <vector>

#include "mifsk.h"
#include "mifsk_device.h"
#include "mifsk_sinf.h"

namespace mifsk {

constexpr int TX_BLOCK = 256;		// threads per stream; also tones per chunk

struct TxArgs {
    const uint8_t	*d_words;	size_t words_stride;
    const uint32_t	*d_nwords;	uint32_t nwords;
    const uint32_t	*d_lead;	uint32_t lead;
    const float		*tab_f;		// [table_len] mag * sinf
    const short		*tab_s;		// [table_len] what an S16 file holds
    uint32_t		table_len;	// 0: --lut=0, sinf per sample
    float		mag;		// tone_mag
    float		mag_s;		// (float)(unsigned short) 32767 * tone_mag, clamped (:150-154)
    int			as_s16;
    float		*d_out;		size_t out_stride;
    uint32_t		*d_nsamples;
    // framing, derived on the host from mifsk_rx_config
    float		mark, space;
    uint32_t		sample_rate;
    uint32_t		bit_nsamples, start_nsamples, stop_nsamples;
    uint32_t		ndata;
    int			inv, msb_first, leader, nsync;
    uint32_t		sync_byte;
};

// tone q of a stream: frequency and length (0 length = past the end)
struct Tone { float freq; uint32_t n; };

__device__ Tone tone_at( const TxArgs &a, const uint8_t *words, uint32_t nwords, uint32_t q )
{
    Tone t;
    t.freq = 0.0f;
    t.n = 0;
    if ( q < (uint32_t)a.leader ) {				// minimodem.c:207-213
	t.freq = a.inv ? a.space : a.mark;
	t.n = a.bit_nsamples;
	return t;
    }
    q -= (uint32_t)a.leader;
    const uint32_t has_start = a.start_nsamples ? 1u : 0u, has_stop = a.stop_nsamples ? 1u : 0u;
    const uint32_t per_frame = has_start + a.ndata + has_stop;
    const uint32_t nframes = (uint32_t)a.nsync + nwords;
    const uint32_t fi = q / per_frame, k = q - fi * per_frame;
    if ( fi < nframes ) {					// minimodem.c:81-112
	const bool sync = fi < (uint32_t)a.nsync;
	const uint32_t bits = sync ? a.sync_byte : words[fi - (uint32_t)a.nsync];
	const bool msb = !sync && a.msb_first;
	if ( has_start && k == 0 ) {
	    t.freq = a.inv ? a.mark : a.space;
	    t.n = a.start_nsamples;
	} else if ( k < has_start + a.ndata ) {
	    const uint32_t i = k - has_start;
	    const uint32_t bit = msb ? ( bits >> ( a.ndata - i - 1u ) ) & 1u : ( bits >> i ) & 1u;
	    t.freq = bit ? a.mark : a.space;
	    t.n = a.bit_nsamples;
	} else {
	    t.freq = a.inv ? a.space : a.mark;
	    t.n = a.stop_nsamples;
	}
	return t;
    }
    if ( q - nframes * per_frame < 2u ) {			// trailer, minimodem.c:59-74,249
	t.freq = a.mark;
	t.n = a.bit_nsamples;
    }
    return t;
}

__global__ __launch_bounds__(TX_BLOCK)
void tx_synth_kernel( TxArgs a )
{
    __shared__ float s_freq[TX_BLOCK];
    __shared__ float s_phase[TX_BLOCK];		// cphase at the start of each tone
    __shared__ uint32_t s_off[TX_BLOCK + 1];	// first sample of each tone within the chunk
    __shared__ float s_carry;

    const uint32_t s = blockIdx.x;
    const uint8_t *words = a.d_words + (size_t)s * a.words_stride;
    const uint32_t nwords = a.d_nwords ? a.d_nwords[s] : a.nwords;
    const uint32_t lead = a.d_lead ? a.d_lead[s] : a.lead;
    float *out = a.d_out + (size_t)s * a.out_stride;
    const size_t cap = a.out_stride;

    // leading silence: tone(0): zeros, phase reset (simple-tone-generator.c:168-172)
    for ( size_t j = threadIdx.x; j < lead && j < cap; j += TX_BLOCK )
	out[j] = 0.0f;
    size_t pos = lead;
    if ( threadIdx.x == 0 )
	s_carry = 0.0f;
    __syncthreads();

    const uint32_t per_frame = ( a.start_nsamples ? 1u : 0u ) + a.ndata + ( a.stop_nsamples ? 1u : 0u );
    const uint32_t ntones = nwords ? (uint32_t)a.leader + ( (uint32_t)a.nsync + nwords ) * per_frame + 2u
				   : 0u;
    for ( uint32_t q0 = 0; q0 < ntones; q0 += TX_BLOCK ) {
	const uint32_t q = q0 + threadIdx.x;
	const Tone t = q < ntones ? tone_at(a, words, nwords, q) : Tone{ 0.0f, 0u };
	s_freq[threadIdx.x] = t.freq;
	s_off[threadIdx.x + 1] = t.n;
	__syncthreads();
	if ( threadIdx.x == 0 ) {
	    // the phase accumulator is a sequential f32 recurrence over the tones
	    // (simple-tone-generator.c:164-166); lengths -> prefix offsets
	    float cphase = s_carry;
	    uint32_t off = 0;
	    s_off[0] = 0;
	    for ( int k = 0; k < TX_BLOCK; k++ ) {
		const uint32_t n = s_off[k + 1];
		s_phase[k] = cphase;
		if ( n ) {
		    const float wave_nsamples = (float)a.sample_rate / s_freq[k];
		    cphase = cphase + (float)n / wave_nsamples;
		    cphase = cphase - truncf(cphase);		// fmodf(x, 1.0f), x >= 0: exact
		}
		off += n;
		s_off[k + 1] = off;
	    }
	    s_carry = cphase;
	}
	__syncthreads();
	const uint32_t total = s_off[TX_BLOCK];
	for ( uint32_t j = threadIdx.x; j < total; j += TX_BLOCK ) {
	    // tone of sample j: last k with s_off[k] <= j
	    uint32_t lo = 0, hi = TX_BLOCK;
	    while ( hi - lo > 1 ) {
		const uint32_t mid = ( lo + hi ) >> 1;
		if ( s_off[mid] <= j ) lo = mid; else hi = mid;
	    }
	    const uint32_t i = j - s_off[lo];
	    const float wave_nsamples = (float)a.sample_rate / s_freq[lo];
	    const float turns = (float)i / wave_nsamples + s_phase[lo];	// :118
	    float v;
	    if ( a.table_len ) {
		int ti = (int)( (float)a.table_len * turns + 0.5f );		// :120-121
		ti %= (int)a.table_len;
		v = a.as_s16 ? (float)a.tab_s[ti] / 32768.0f : a.tab_f[ti];
	    } else {
		const float rad = (float)M_PI * 2 * turns;			// :116,134,155
		const float sn = mifsk_glibc_sinf(rad);
		v = a.as_s16 ? (float)(short)lroundf(a.mag_s * sn) / 32768.0f : a.mag * sn;
	    }
	    if ( pos + j < cap )
		out[pos + j] = v;
	}
	pos += total;
	__syncthreads();
    }
    // the rest of the row is defined (zero), and the stream's length is reported
    for ( size_t j = pos + threadIdx.x; j < cap; j += TX_BLOCK )
	out[j] = 0.0f;
    if ( threadIdx.x == 0 && a.d_nsamples )
	a.d_nsamples[s] = (uint32_t)pos;
}

} // namespace mifsk

namespace {

// the sine table of simple-tone-generator.c:35-89 for (len, mag), on the device
struct TxTable {
    unsigned len; float mag;
    float *d_f; short *d_s;
};
std::vector<TxTable> g_tables;		// per process; tiny (one per distinct --lut / --volume)

int get_table( unsigned len, float mag, const float **d_f, const short **d_s )
{
    for ( const TxTable &t : g_tables )
	if ( t.len == len && t.mag == mag ) {
	    *d_f = t.d_f; *d_s = t.d_s;
	    return 0;
	}
    unsigned short mag_s = (unsigned short)( 32767.0f * mag + 0.5f );
    if ( mag > 1.0f )
	mag_s = 32767;
    if ( mag_s < 1 )
	mag_s = 1;
    std::vector<float> f(len);
    std::vector<short> sh(len);
    for ( unsigned i = 0; i < len; i++ ) {
	const float ang = (float)M_PI * 2 * i / len;
	sh[i] = (short)lroundf(mag_s * sinf(ang));
	f[i] = mag * sinf(ang);
    }
    TxTable t = { len, mag, nullptr, nullptr };
    if ( hipMalloc(&t.d_f, len * sizeof(float)) != hipSuccess
	    || hipMalloc(&t.d_s, len * sizeof(short)) != hipSuccess
	    || hipMemcpy(t.d_f, f.data(), len * sizeof(float), hipMemcpyHostToDevice) != hipSuccess
	    || hipMemcpy(t.d_s, sh.data(), len * sizeof(short), hipMemcpyHostToDevice) != hipSuccess )
	return -ENOMEM;
    g_tables.push_back(t);
    *d_f = t.d_f; *d_s = t.d_s;
    return 0;
}

} // namespace

extern "C" int mifsk_tx_synthesize_batch( mifsk_ctx *ctx, const mifsk_rx_config *cfg,
	const uint8_t *d_words, size_t words_stride, const uint32_t *d_nwords, uint32_t nwords,
	int nstreams, unsigned sin_table_len, float amplitude,
	const uint32_t *d_leading_silence, uint32_t leading_silence, int as_s16,
	float *d_out, size_t out_stride, uint32_t *d_nsamples_out, void *stream )
{
    if ( !ctx || !cfg || !d_out || nstreams < 0 || !( amplitude > 0.0f ) || cfg->n_data_bits > 8
	    || ( !d_words && ( d_nwords || nwords ) ) )
	return -EINVAL;
    if ( nstreams == 0 )
	return 0;
    if ( hipSetDevice(mifsk::ctx_device(ctx)) != hipSuccess )
	return -EIO;
    mifsk::TxArgs a;
    a.tab_f = nullptr;
    a.tab_s = nullptr;
    if ( sin_table_len ) {
	int rc = get_table(sin_table_len, amplitude, &a.tab_f, &a.tab_s);
	if ( rc )
	    return rc;
    }
    a.mag = amplitude;
    {
	// simple-tone-generator.c:150-154
	unsigned short mag_s = (unsigned short)( 32767.0f * amplitude + 0.5f );
	if ( amplitude > 1.0f )
	    mag_s = 32767;
	if ( mag_s < 1 )
	    mag_s = 1;
	a.mag_s = (float)mag_s;
    }
    a.d_words = d_words;	a.words_stride = words_stride;
    a.d_nwords = d_nwords;	a.nwords = nwords;
    a.d_lead = d_leading_silence;	a.lead = leading_silence;
    a.table_len = sin_table_len;
    a.as_s16 = as_s16;
    a.d_out = d_out;		a.out_stride = out_stride;
    a.d_nsamples = d_nsamples_out;
    a.mark = cfg->mark_f;	a.space = cfg->space_f;
    a.sample_rate = cfg->sample_rate;
    // minimodem.c:131-132 and :86-110, in the host generator's arithmetic
    const size_t bit_nsamples = (size_t)( (size_t)cfg->sample_rate / cfg->data_rate + 0.5f );
    a.bit_nsamples = (uint32_t)bit_nsamples;
    a.start_nsamples = cfg->nstartbits > 0 ? (uint32_t)(size_t)( bit_nsamples * (float)cfg->nstartbits ) : 0u;
    a.stop_nsamples = cfg->nstopbits > 0 ? (uint32_t)(size_t)( bit_nsamples * cfg->nstopbits ) : 0u;
    a.ndata = cfg->n_data_bits;
    a.inv = cfg->invert_start_stop;
    a.msb_first = cfg->msb_first;
    a.leader = cfg->nstartbits == 0 ? 0 : 2;
    a.nsync = cfg->do_rx_sync ? 16 : 0;
    a.sync_byte = (uint32_t)cfg->sync_byte;
    hipLaunchKernelGGL(mifsk::tx_synth_kernel, dim3((unsigned)nstreams), dim3(mifsk::TX_BLOCK), 0,
		       (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -EIO;
}
