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
constexpr int TX_QUOT = 1024;		// longest bit (samples) with a quotient table

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

// One workgroup per stream, chunks of TX_BLOCK tones.  Wave 0 is the phase
// accumulator -- a sequential f32 recurrence over the tones
// (simple-tone-generator.c:164-166), run by its lane 0 one chunk ahead -- and
// waves 1-3 write the samples of the chunk before.  What the recurrence adds per
// tone and the tone's samples per wave (the two divisions of :117,164) are
// computed one tone per thread when the chunk is fetched.
__global__ __launch_bounds__(TX_BLOCK)
void tx_synth_kernel( TxArgs a )
{
    __shared__ float s_wn[2][TX_BLOCK];		// samples per wave of each tone (:117)
    __shared__ __attribute__((aligned(16))) float s_dq[2][TX_BLOCK];	// what the tone adds to the phase (:164)
    __shared__ __attribute__((aligned(16))) float s_phase[2][TX_BLOCK];	// cphase at the start of each tone
    __shared__ __attribute__((aligned(16))) uint32_t s_n[2][TX_BLOCK];	// samples of each tone
    __shared__ __attribute__((aligned(16))) uint32_t s_off[2][TX_BLOCK + 4];	// first sample of each tone within the chunk
    // (float)i / wave_nsamples for the two tones, where every tone is as long as a
    // data bit and short enough: a sample then looks its quotient up instead of
    // dividing (the same division, made once per i)
    __shared__ float s_quot[2][TX_QUOT];

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

    const uint32_t per_frame = ( a.start_nsamples ? 1u : 0u ) + a.ndata + ( a.stop_nsamples ? 1u : 0u );
    const uint32_t ntones = nwords ? (uint32_t)a.leader + ( (uint32_t)a.nsync + nwords ) * per_frame + 2u
				   : 0u;
    const uint32_t nchunks = ( ntones + TX_BLOCK - 1u ) / TX_BLOCK;
    // every tone of the stream as long as a data bit (no 1.5 stop bits ...): the
    // tone of a sample is a division by a constant instead of a search
    const bool uniform = ( !a.start_nsamples || a.start_nsamples == a.bit_nsamples )
		      && ( !a.stop_nsamples || a.stop_nsamples == a.bit_nsamples ) && a.bit_nsamples > 1u;
    const uint32_t bit_magic = uniform ? (uint32_t)( 0x100000000ULL / a.bit_nsamples ) : 0u;
    const bool pow2 = a.table_len && ( a.table_len & ( a.table_len - 1u ) ) == 0u;
    float cphase = 0.0f;			// (lane 0 of wave 0)
    const bool quot = uniform && a.bit_nsamples <= (uint32_t)TX_QUOT;
    const bool vec4 = quot && a.table_len && ( a.bit_nsamples & 3u ) == 0u;
    if ( quot ) {
	const float wn_mark = (float)a.sample_rate / a.mark, wn_space = (float)a.sample_rate / a.space;
	for ( uint32_t i = threadIdx.x; i < a.bit_nsamples; i += TX_BLOCK ) {
	    s_quot[0][i] = (float)i / wn_mark;
	    s_quot[1][i] = (float)i / wn_space;
	}
    }

    auto fetch = [&]( uint32_t c, uint32_t b ) {
	const uint32_t q = c * TX_BLOCK + threadIdx.x;
	const Tone t = q < ntones ? tone_at(a, words, nwords, q) : Tone{ 0.0f, 0u };
	const float wn = t.n ? (float)a.sample_rate / t.freq : 1.0f;
	// (with the quotient table: which of the two tones, as a float)
	s_wn[b][threadIdx.x] = quot ? ( t.freq == a.mark ? 0.0f : 1.0f ) : wn;
	s_dq[b][threadIdx.x] = t.n ? (float)t.n / wn : 0.0f;
	s_n[b][threadIdx.x] = t.n;
    };
    // lengths -> prefix offsets, phases.  Four tones per step: the LDS reads do not
    // sit inside the dependent chain.  (A tone of no samples -- past the end --
    // adds 0.0f to a phase in [0, 1): nothing.)
    auto scan = [&]( uint32_t b ) {
	uint32_t off = 0;
	for ( int k = 0; k < TX_BLOCK; k += 4 ) {
	    const float4 d = *reinterpret_cast<const float4 *>(&s_dq[b][k]);
	    const uint4 n = *reinterpret_cast<const uint4 *>(&s_n[b][k]);
	    float4 ph;
	    uint4 st;
	    ph.x = cphase;  st.x = off;
	    cphase = cphase + d.x;  cphase = cphase - truncf(cphase);	// fmodf(x, 1.0f), x >= 0: exact
	    off += n.x;
	    ph.y = cphase;  st.y = off;
	    cphase = cphase + d.y;  cphase = cphase - truncf(cphase);
	    off += n.y;
	    ph.z = cphase;  st.z = off;
	    cphase = cphase + d.z;  cphase = cphase - truncf(cphase);
	    off += n.z;
	    ph.w = cphase;  st.w = off;
	    cphase = cphase + d.w;  cphase = cphase - truncf(cphase);
	    off += n.w;
	    *reinterpret_cast<float4 *>(&s_phase[b][k]) = ph;
	    *reinterpret_cast<uint4 *>(&s_off[b][k]) = st;
	}
	s_off[b][TX_BLOCK] = off;
    };

    if ( nchunks ) {
	fetch(0, 0);
	__syncthreads();
	if ( threadIdx.x == 0 )
	    scan(0);
	__syncthreads();
    }
    for ( uint32_t c = 0; c < nchunks; c++ ) {
	const uint32_t b = c & 1u;
	if ( c + 1u < nchunks )
	    fetch(c + 1u, b ^ 1u);
	__syncthreads();
	const uint32_t total = s_off[b][TX_BLOCK];
	if ( threadIdx.x < 64u ) {
	    if ( threadIdx.x == 0 && c + 1u < nchunks )
		scan(b ^ 1u);
	} else if ( vec4 ) {
	    // four consecutive samples per thread: they share their tone (bits are
	    // a multiple of four samples long), hence its offset, phase and table
	    // row, and leave in one 16-byte store -- a third of the instructions
	    for ( uint32_t j = 4u * ( threadIdx.x - 64u ); j < total; j += 4u * ( TX_BLOCK - 64u ) ) {
		uint32_t lo = __umulhi(j, bit_magic);			// floor(j / B) or one less
		if ( j - lo * a.bit_nsamples >= a.bit_nsamples )
		    lo++;
		const uint32_t i = j - s_off[b][lo];
		const float *qrow = s_quot[s_wn[b][lo] != 0.0f] + i;
		const float ph = s_phase[b][lo];
		float v[4];
#pragma unroll
		for ( int u = 0; u < 4; u++ ) {
		    const float turns = qrow[u] + ph;				// :117-118
		    int ti = (int)( (float)a.table_len * turns + 0.5f );	// :120-121
		    if ( pow2 )
			ti &= (int)( a.table_len - 1u );			// (ti >= 0)
		    else
			ti %= (int)a.table_len;
		    v[u] = a.as_s16 ? (float)a.tab_s[ti] / 32768.0f : a.tab_f[ti];
		}
		// (total is a multiple of four here: all four samples exist)
		if ( pos + j + 3u < cap ) {
		    typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));
		    float4_u o; o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
		    *reinterpret_cast<float4_u *>(out + pos + j) = o;
		} else {
#pragma unroll
		    for ( int u = 0; u < 4; u++ )
			if ( pos + j + (uint32_t)u < cap )
			    out[pos + j + (uint32_t)u] = v[u];
		}
	    }
	} else {
	    for ( uint32_t j = threadIdx.x - 64u; j < total; j += TX_BLOCK - 64u ) {
		// tone of sample j: last k with s_off[k] <= j
		uint32_t lo;
		if ( uniform ) {
		    lo = __umulhi(j, bit_magic);			// floor(j / B) or one less
		    if ( j - lo * a.bit_nsamples >= a.bit_nsamples )
			lo++;
		} else {
		    lo = 0;
		    uint32_t hi = TX_BLOCK;
		    while ( hi - lo > 1 ) {
			const uint32_t mid = ( lo + hi ) >> 1;
			if ( s_off[b][mid] <= j ) lo = mid; else hi = mid;
		    }
		}
		const uint32_t i = j - s_off[b][lo];
		const float w = s_wn[b][lo];
		const float turns = ( quot ? s_quot[w != 0.0f][i] : (float)i / w ) + s_phase[b][lo];	// :117-118
		float v;
		if ( a.table_len ) {
		    int ti = (int)( (float)a.table_len * turns + 0.5f );		// :120-121
		    if ( pow2 )
			ti &= (int)( a.table_len - 1u );		// (ti >= 0)
		    else
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
