// mifsk_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X / CDNA4).
//
// The reference's FSK receive path (src/fsk.c:107-538 driven by the loop in
// src/minimodem.c:1137-1463) restructured for a 64-wide, LDS-centred machine:
//
//  * The reference runs a full r2c FFT per bit and reads two bins.  Here a bit
//    window is what it mathematically is: two complex dot products
//        X[b] = sum_n x[n] * exp(-2 pi i b n / fftsize),  b in {mark, space}
//    One LANE owns one bit window; the twiddle for sample n is the same for
//    every lane, so it is fetched once per wave through the scalar cache
//    (s_load) and each sample costs one LDS read and four f64 FMAs.
//  * f64 accumulation in index order, exactly the operation sequence of the
//    oracle, so results are bit-identical to it (the "-P" reference tests need
//    the off-tone bin below FLT_EPSILON, which f32 accumulation cannot
//    guarantee).
//  * Audio is staged once, coalesced (16 B per lane), from HBM into an LDS
//    slab.  The slab is row-skewed (one pad word per bit length) so that 64
//    lanes striding by one bit length hit 64 different banks.
//  * The receive loop is a serial, data-dependent cursor.  A workgroup owns a
//    stream; whenever it must evaluate a search it also evaluates, with the
//    otherwise idle lanes, the first-try position of the next M frames (where
//    the cursor will land if every frame locks at its expected offset -- the
//    normal case once carrier is acquired).  Those results sit in an LDS cache
//    keyed by absolute position; later iterations hit the cache and replay the
//    reference's decision logic without touching the samples again.  A miss
//    simply recomputes: results never depend on what was predicted.
//
// No MFMA: the path is HBM-bound (4 B read per sample, ~5 f64 FMA per sample).
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdint>

#include "mifsk_device.h"

namespace mifsk {

constexpr int BLOCK = 256;	// threads per stream workgroup (4 waves)
constexpr int P_CAP = 64;	// candidate positions per batch (= one wave of lanes)
constexpr int W_CAP = 512;	// bit windows per batch (LDS scratch)

// ---------------------------------------------------------------------------
// arithmetic shared by every kernel
// ---------------------------------------------------------------------------

// |X[b]| * scalar, as the reference computes it (fsk.c:107-114): the FFT output
// is a pair of floats; hypotf in glibc 2.35 is exactly
// (float)sqrt((double)re*re + (double)im*im) (verified exhaustively on the
// host, tests/test_host_math.py); f64 sqrt on gfx950 is correctly rounded.
__device__ __forceinline__ float band_mag( double re, double im, float scalar )
{
    const float fr = (float)re, fi = (float)im;
    const double s = (double)fr * (double)fr + (double)fi * (double)fi;
    return (float)sqrt(s) * scalar;
}

struct FrameOut {
    float	conf;
    float	ampl;
    uint64_t	bits;
};

// fsk_frame_analyze after the per-bit magnitudes are known (fsk.c:199-212,
// 271-342, 439-441).  `mags[k]` = (mark, space) magnitude of bit k.  One lane
// runs this for one candidate position; every operation is f32, in the
// reference's order (contraction is disabled for this file).
__device__ __forceinline__ FrameOut
frame_confidence( const float2 *mags, const uint8_t *expect, uint32_t n_bits )
{
    FrameOut out;
    out.conf = 0.0f;
    out.ampl = 0.0f;
    out.bits = 0;

    uint64_t bits = 0;
    float total_sig = 0.0f, total_noise = 0.0f;
    float mark_sig = 0.0f, space_sig = 0.0f;
    uint32_t n_mark = 0, n_space = 0;
    bool mismatch = false;
    for ( uint32_t k = 0; k < n_bits; k++ ) {
	const float2 m = mags[k];
	const bool one = m.x > m.y;			// fsk.c:161 (strict)
	const float sig = one ? m.x : m.y;
	const float noise = one ? m.y : m.x;
	const uint32_t e = expect[k];
	if ( e != 2u && e != (one ? 1u : 0u) )
	    mismatch = true;				// fsk.c:211-212
	bits |= (uint64_t)(one ? 1u : 0u) << k;
	total_sig += sig;				// fsk.c:278
	if ( noise > FLT_EPSILON )			// fsk.c:279
	    total_noise += noise;
	if ( one ) {
	    mark_sig += sig;
	    n_mark++;
	} else {
	    space_sig += sig;
	    n_space++;
	}
    }
    if ( mismatch )
	return out;		// confidence 0, bits/ampl stay 0 (fsk.c:486-487)

    const float snr = total_sig / total_noise;		// fsk.c:292
    const float avg_sig = total_sig / (float)(int)n_bits;	// fsk.c:295 (int n_bits)
    if ( n_mark )
	mark_sig /= (float)n_mark;			// fsk.c:298-301
    if ( n_space )
	space_sig /= (float)n_space;

    float divergence = 0.0f;				// fsk.c:305-313
    for ( uint32_t k = 0; k < n_bits; k++ ) {
	const float2 m = mags[k];
	const bool one = m.x > m.y;
	const float sig = one ? m.x : m.y;
	const float cls = one ? mark_sig : space_sig;
	divergence += fabsf(sig - cls) / cls;
    }
    divergence *= 2.0f;
    divergence /= (float)(int)n_bits;

    out.conf = snr * (1.0f - divergence);		// fsk.c:336
    out.ampl = avg_sig;					// fsk.c:342
    out.bits = bits;					// fsk.c:439-441
    return out;
}

// The reference's zig-zag scan order (fsk.c:477-484): first, first+s, first-s,
// first+2s, first-2s, ...; an up-step reaching try_max ends the scan, a
// down-step below 0 is skipped.  Closed form: U up-positions (u = 0..U-1),
// D valid down-positions (u = 1..D), J = U + D candidates in total.
struct ZigZag {
    uint32_t first, step, U, D, J;
    __device__ __forceinline__ ZigZag( uint32_t f, uint32_t mx, uint32_t s )
    {
	first = f;
	step = s;
	if ( (int)f >= (int)mx || s == 0 ) {
	    U = D = J = 0;
	} else {
	    U = ( mx - f - 1 ) / s + 1;
	    const uint32_t dmax = f / s;
	    D = U - 1 < dmax ? U - 1 : dmax;
	    J = U + D;
	}
    }
    // i-th candidate (0-based, scan order)
    __device__ __forceinline__ uint32_t at( uint32_t i ) const
    {
	if ( i == 0 )
	    return first;
	if ( i <= 2 * D ) {
	    const uint32_t u = ( i + 1 ) >> 1;
	    return ( i & 1u ) ? first + u * step : first - u * step;
	}
	return first + ( i - D ) * step;
    }
};

// ---------------------------------------------------------------------------
// kernel 1: N independent fsk_find_frame problems (one 64-lane workgroup each)
// Samples are read straight from global memory with a bounds check; this is
// the compatibility / known-answer path, not the throughput path.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(64)
void find_frame_kernel( DevCfg cfg, const double *__restrict__ tw,
	const float *__restrict__ samples, const mifsk_search *__restrict__ problems,
	mifsk_search_result *__restrict__ results )
{
    __shared__ float2 s_mags[W_CAP];
    __shared__ float s_conf[P_CAP];
    __shared__ float s_ampl[P_CAP];
    __shared__ uint64_t s_bits[P_CAP];

    const mifsk_search pr = problems[blockIdx.x];
    const float *x = samples + pr.sample_offset;
    const uint32_t navail = pr.navail;
    const uint32_t n_bits = cfg.n_bits;
    const uint32_t B = cfg.bit_nsamples;
    const uint8_t *expect = cfg.expect[pr.use_sync_string ? 1 : 0];
    const uint32_t lane = threadIdx.x;

    const ZigZag zz(pr.try_first, pr.try_max, pr.try_step);
    uint32_t qmax = W_CAP / n_bits;
    if ( qmax > P_CAP ) qmax = P_CAP;

    float best_c = 0.0f, best_a = 0.0f;
    uint64_t best_bits = 0;
    uint32_t best_t = 0, n_tried = 0;
    bool done = false;

    for ( uint32_t c0 = 0; c0 < zz.J && !done; c0 += qmax ) {
	const uint32_t Q = zz.J - c0 < qmax ? zz.J - c0 : qmax;
	const uint32_t nwin = Q * n_bits;
	for ( uint32_t w0 = 0; w0 < nwin; w0 += 64 ) {
	    const uint32_t w = w0 + lane;
	    const bool active = w < nwin;
	    const uint32_t q = active ? w / n_bits : 0;
	    const uint32_t k = active ? w - q * n_bits : 0;
	    const uint32_t a = zz.at(c0 + q) + cfg.bit_offset[k];
	    double mr = 0.0, mi = 0.0, sr = 0.0, si = 0.0;
	    for ( uint32_t n = 0; n < B; n++ ) {
		const uint32_t idx = a + n;
		const float xv = ( active && idx < navail ) ? x[idx] : 0.0f;
		const double xd = (double)xv;
		const double *t = tw + 4 * (size_t)n;
		mr = fma(xd, t[0], mr);
		mi = fma(xd, t[1], mi);
		sr = fma(xd, t[2], sr);
		si = fma(xd, t[3], si);
	    }
	    if ( active )
		s_mags[w] = make_float2(band_mag(mr, mi, cfg.magscalar),
					band_mag(sr, si, cfg.magscalar));
	}
	__syncthreads();
	if ( lane < Q ) {
	    const FrameOut f = frame_confidence(&s_mags[lane * n_bits], expect, n_bits);
	    s_conf[lane] = f.conf;
	    s_ampl[lane] = f.ampl;
	    s_bits[lane] = f.bits;
	}
	__syncthreads();
	for ( uint32_t i = 0; i < Q; i++ ) {		// fsk.c:492-501
	    const float c = s_conf[i];
	    n_tried++;
	    if ( best_c < c ) {
		best_c = c;
		best_a = s_ampl[i];
		best_bits = s_bits[i];
		best_t = zz.at(c0 + i);
		if ( best_c >= pr.search_limit ) {
		    done = true;
		    break;
		}
	    }
	}
	__syncthreads();
    }
    if ( lane == 0 ) {
	mifsk_search_result r;
	r.bits = best_bits;
	r.confidence = best_c;
	r.amplitude = best_a;
	r.frame_start = best_t;
	r.n_positions = n_tried;
	results[blockIdx.x] = r;
    }
}

// ---------------------------------------------------------------------------
// kernel 2: the receive loop, one workgroup per stream
// ---------------------------------------------------------------------------

struct StreamLds {
    float2	mags[W_CAP];
    uint64_t	c_bits[P_CAP];
    float	c_conf[P_CAP];
    float	c_ampl[P_CAP];
    uint32_t	c_pos[P_CAP];
    uint32_t	c_n;
    uint32_t	c_kind;
    uint32_t	pad[2];
    float	slab[1];	// really slab_floats long (dynamic LDS)
};

struct ScanResult {
    float	conf;
    float	ampl;
    uint64_t	bits;
    uint32_t	start;
    uint32_t	computed;	// 1 when a batch was evaluated (cache was rewritten)
};

template <bool USE_SLAB>
struct StreamCtx {
    const DevCfg	&cfg;
    const double	*tw;
    const float		*x;		// this stream's samples
    uint32_t		N;		// valid samples; reads beyond see 0.0
    StreamLds		*lds;
    uint32_t		slab_cap;	// samples the slab can hold
    uint32_t		slab_lo, slab_hi;	// absolute range currently staged
    uint32_t		npredict;	// frames to run ahead (0 = none)

    __device__ __forceinline__ StreamCtx( const DevCfg &c, const double *t, const float *xs,
	    uint32_t n, StreamLds *l, uint32_t cap, uint32_t np )
	: cfg(c), tw(t), x(xs), N(n), lds(l), slab_cap(cap), slab_lo(0), slab_hi(0),
	  npredict(np) {}

    // LDS word of slab-relative sample a: rows of B samples, `skew` pad words
    // between rows, so lanes one bit length apart land on different banks.
    __device__ __forceinline__ uint32_t slab_index( uint32_t a ) const
    {
	return a + ( a / cfg.bit_nsamples ) * cfg.skew;
    }

    // stage [lo, lo + slab_cap) (lo rounded down to 16 B) into the slab
    __device__ void stage( uint32_t lo )
    {
	__syncthreads();		// everyone is done reading the old contents
	const uint32_t org = lo & ~3u;
	float *slab = lds->slab;
	const uint32_t nvec = slab_cap >> 2;
	for ( uint32_t v = threadIdx.x; v < nvec; v += BLOCK ) {
	    const uint32_t rel = v << 2;
	    const uint32_t a = org + rel;
	    float4 s;
	    if ( a + 3 < N && a + 3 >= a ) {
		s = *reinterpret_cast<const float4 *>(x + a);	// coalesced 16 B / lane
	    } else {
		s.x = a < N ? x[a] : 0.0f;
		s.y = ( a + 1 < N && a + 1 > a ) ? x[a + 1] : 0.0f;
		s.z = ( a + 2 < N && a + 2 > a ) ? x[a + 2] : 0.0f;
		s.w = ( a + 3 < N && a + 3 > a ) ? x[a + 3] : 0.0f;
	    }
	    slab[slab_index(rel)] = s.x;
	    slab[slab_index(rel + 1)] = s.y;
	    slab[slab_index(rel + 2)] = s.z;
	    slab[slab_index(rel + 3)] = s.w;
	}
	slab_lo = org;
	slab_hi = org + slab_cap;
	__syncthreads();
    }

    // Evaluate candidate positions c_pos[0..nq) (already in LDS): bit windows
    // -> magnitudes -> per-position confidence, results into the cache arrays.
    __device__ void evaluate( uint32_t nq, uint32_t kind, uint32_t lo, uint32_t hi )
    {
	const uint32_t n_bits = cfg.n_bits;
	const uint32_t B = cfg.bit_nsamples;
	if ( USE_SLAB ) {
	    if ( lo < slab_lo || hi > slab_hi )
		stage(lo);
	}
	const uint32_t nwin = nq * n_bits;
	for ( uint32_t w0 = 0; w0 < nwin; w0 += BLOCK ) {
	    const uint32_t w = w0 + threadIdx.x;
	    const bool active = w < nwin;
	    const uint32_t q = active ? w / n_bits : 0;
	    const uint32_t k = active ? w - q * n_bits : 0;
	    const uint32_t a = lds->c_pos[q] + cfg.bit_offset[k];
	    double mr = 0.0, mi = 0.0, sr = 0.0, si = 0.0;
	    if ( USE_SLAB ) {
		const uint32_t rel = a - slab_lo;
		const uint32_t row = rel / B;
		const uint32_t col = rel - row * B;
		const float *p = lds->slab + rel + row * cfg.skew;
		const uint32_t wrap = B - col;	// first n that falls into the next row
		const uint32_t skew = cfg.skew;
		for ( uint32_t n = 0; n < B; n++ ) {
		    const float xv = p[n + ( n >= wrap ? skew : 0u )];
		    const double xd = (double)xv;
		    const double *t = tw + 4 * (size_t)n;
		    mr = fma(xd, t[0], mr);
		    mi = fma(xd, t[1], mi);
		    sr = fma(xd, t[2], sr);
		    si = fma(xd, t[3], si);
		}
	    } else {
		for ( uint32_t n = 0; n < B; n++ ) {
		    const uint32_t idx = a + n;
		    const float xv = ( idx < N && idx >= a ) ? x[idx] : 0.0f;
		    const double xd = (double)xv;
		    const double *t = tw + 4 * (size_t)n;
		    mr = fma(xd, t[0], mr);
		    mi = fma(xd, t[1], mi);
		    sr = fma(xd, t[2], sr);
		    si = fma(xd, t[3], si);
		}
	    }
	    if ( active )
		lds->mags[w] = make_float2(band_mag(mr, mi, cfg.magscalar),
					   band_mag(sr, si, cfg.magscalar));
	}
	__syncthreads();
	if ( threadIdx.x < nq ) {
	    const FrameOut f = frame_confidence(&lds->mags[threadIdx.x * n_bits],
						cfg.expect[kind], n_bits);
	    lds->c_conf[threadIdx.x] = f.conf;
	    lds->c_ampl[threadIdx.x] = f.ampl;
	    lds->c_bits[threadIdx.x] = f.bits;
	}
	__syncthreads();
    }

    // fsk_find_frame at cursor `base` (absolute), with run-ahead (see file header)
    __device__ ScanResult scan( uint32_t base, uint32_t first, uint32_t tmax, uint32_t step,
	    float limit, uint32_t kind, bool may_predict )
    {
	ScanResult r;
	r.conf = 0.0f; r.ampl = 0.0f; r.bits = 0; r.start = 0; r.computed = 0;
	const ZigZag zz(first, tmax, step);
	if ( zz.J == 0 )
	    return r;

	// cache probe for the first candidate
	{
	    const uint32_t lane = threadIdx.x & 63u;
	    const uint32_t p0 = base + first;
	    const bool m = lane < lds->c_n && lds->c_kind == kind && lds->c_pos[lane] == p0;
	    const unsigned long long b = __ballot(m);
	    if ( b ) {
		const int hit = __ffsll((long long)b) - 1;
		const float c = lds->c_conf[hit];
		if ( c > 0.0f && c >= limit ) {		// fsk.c:492,499
		    r.conf = c;
		    r.ampl = lds->c_ampl[hit];
		    r.bits = lds->c_bits[hit];
		    r.start = first;
		    return r;
		}
	    }
	}

	uint32_t qmax = W_CAP / cfg.n_bits;
	if ( qmax > P_CAP ) qmax = P_CAP;
	bool done = false;
	for ( uint32_t c0 = 0; c0 < zz.J && !done; c0 += qmax ) {
	    const uint32_t Q = zz.J - c0 < qmax ? zz.J - c0 : qmax;
	    // extent of this chunk's candidates
	    uint32_t tlo = 0xFFFFFFFFu, thi = 0;
	    for ( uint32_t i = 0; i < Q; i++ ) {
		const uint32_t t = zz.at(c0 + i);
		tlo = t < tlo ? t : tlo;
		thi = t > thi ? t : thi;
	    }
	    const uint32_t lo = base + tlo;
	    uint32_t hi = base + thi + cfg.last_reach;
	    // run-ahead: the next frames' first-try positions
	    uint32_t M = 0;
	    if ( may_predict && c0 == 0 && zz.J <= qmax ) {
		M = qmax - Q < npredict ? qmax - Q : npredict;
		const uint32_t p0 = base + first;
		if ( USE_SLAB ) {
		    const uint32_t org = lo & ~3u;
		    const uint32_t used = ( p0 - org ) + cfg.last_reach;
		    const uint32_t fit = used < slab_cap ? ( slab_cap - used ) / cfg.frame_nsamples : 0;
		    M = M < fit ? M : fit;
		}
		// nothing to gain past the end of the stream
		const uint32_t left = p0 < N ? ( N - p0 ) / cfg.frame_nsamples : 0;
		M = M < left ? M : left;
		if ( M ) {
		    const uint32_t ph = p0 + M * cfg.frame_nsamples + cfg.last_reach;
		    hi = ph > hi ? ph : hi;
		}
	    }
	    __syncthreads();		// all waves are past their cache probe / replay
	    if ( threadIdx.x < Q )
		lds->c_pos[threadIdx.x] = base + zz.at(c0 + threadIdx.x);
	    else if ( threadIdx.x < Q + M )
		lds->c_pos[threadIdx.x] = base + first + ( threadIdx.x - Q + 1 ) * cfg.frame_nsamples;
	    if ( threadIdx.x == 0 ) {
		lds->c_n = Q + M;
		lds->c_kind = kind;
	    }
	    __syncthreads();
	    evaluate(Q + M, kind, lo, hi);
	    r.computed = 1;
	    for ( uint32_t i = 0; i < Q; i++ ) {	// fsk.c:492-501
		const float c = lds->c_conf[i];
		if ( r.conf < c ) {
		    r.conf = c;
		    r.ampl = lds->c_ampl[i];
		    r.bits = lds->c_bits[i];
		    r.start = zz.at(c0 + i);
		    if ( r.conf >= limit ) {
			done = true;
			break;
		    }
		}
	    }
	}
	return r;
    }
};

// databits.h:21-46
__device__ __forceinline__ uint64_t bit_window( uint64_t v, uint32_t offset, uint32_t bits )
{
    if ( bits >= 64 )
	return v >> offset;
    const uint64_t mask = ( 1ULL << bits ) - 1ULL;
    return ( v >> offset ) & mask;
}

__device__ __forceinline__ uint64_t bit_reverse( uint64_t v, uint32_t bits )
{
    uint32_t out = 0;		// the reference accumulates in 32 bits
    while ( bits-- ) {
	out = ( out << 1 ) | (uint32_t)( v & 1ULL );
	v >>= 1;
    }
    return out;
}

template <bool USE_SLAB>
__global__ __launch_bounds__(BLOCK)
void demod_kernel( DevCfg cfg, const double *__restrict__ tw, mifsk_demod_io io,
	uint32_t slab_cap, uint32_t npredict )
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    StreamLds *lds = reinterpret_cast<StreamLds *>(smem);

    const uint32_t s = blockIdx.x;
    const float *x = io.d_samples + (size_t)s * io.stream_stride;
    const uint32_t N = io.d_nsamples ? io.d_nsamples[s] : io.nsamples;
    const size_t fcap = io.frames_cap, ecap = io.episodes_cap;
    uint8_t *o_bytes = io.d_bytes ? io.d_bytes + (size_t)s * fcap : nullptr;
    uint64_t *o_bits = io.d_bits ? io.d_bits + (size_t)s * fcap : nullptr;
    mifsk_frame *o_frames = io.d_frames ? io.d_frames + (size_t)s * fcap : nullptr;
    mifsk_episode *o_eps = io.d_episodes ? io.d_episodes + (size_t)s * ecap : nullptr;
    const bool t0 = threadIdx.x == 0;

    if ( t0 )
	lds->c_n = 0;
    __syncthreads();

    StreamCtx<USE_SLAB> ctx(cfg, tw, x, N, lds, slab_cap, npredict);

    // reference loop state (minimodem.c:1079-1088,1132-1133)
    bool carrier = false;
    float confidence_total = 0.0f, amplitude_total = 0.0f;
    uint32_t nframes_decoded = 0;
    uint64_t carrier_nsamples = 0;
    uint32_t noconfidence = 0;
    uint32_t advance = 0;
    float track_amplitude = 0.0f, peak_confidence = 0.0f;

    uint32_t base = 0;			// absolute index of samplebuf[0]
    uint32_t n_out_frames = 0, n_out_bytes = 0, n_out_eps = 0, ep_first = 0;
    uint32_t status = 0;

    for (;;) {
	// minimodem.c:1150-1156,1176,1229 under flat addressing
	if ( advance ) {
	    if ( advance > N - base )
		break;
	    base += advance;
	}
	const uint32_t avail = N - base;
	if ( avail == 0 || avail < cfg.expect_nsamples )
	    break;

	const uint32_t ci = carrier ? 1u : 0u;
	const uint32_t try_max = cfg.try_max[ci];
	const uint32_t try_step = cfg.try_step[ci];
	const uint32_t try_first = cfg.try_first[ci];

	ScanResult sr = ctx.scan(base, try_first, try_max, try_step, cfg.search_limit,
				 carrier ? 0u : 1u, carrier);	// minimodem.c:1265-1274
	float confidence = sr.conf;
	float amplitude = sr.ampl;
	uint64_t bits = sr.bits;
	uint32_t frame_start = sr.start;

	bool refine = false;
	if ( confidence < peak_confidence * 0.75f ) {		// minimodem.c:1278-1282
	    refine = true;
	    peak_confidence = 0.0f;
	}
	if ( amplitude < track_amplitude * 0.25f )		// minimodem.c:1286-1288
	    confidence = 0.0f;

	if ( confidence <= cfg.conf_threshold ) {		// minimodem.c:1292-1321
	    if ( ++noconfidence > 20u ) {
		if ( carrier ) {
		    if ( t0 && o_eps && n_out_eps < ecap ) {
			mifsk_episode e;
			e.carrier_nsamples = carrier_nsamples;
			e.first_frame = ep_first;
			e.nframes = nframes_decoded;
			e.confidence_total = confidence_total;
			e.amplitude_total = amplitude_total;
			e.end_reason = 1;
			e.reserved = 0;
			o_eps[n_out_eps] = e;
		    }
		    n_out_eps++;
		    carrier = false;
		    carrier_nsamples = 0;
		    confidence_total = 0.0f;
		    amplitude_total = 0.0f;
		    nframes_decoded = 0;
		    track_amplitude = 0.0f;
		    if ( cfg.rx_one )
			break;
		}
	    }
	    advance = try_max;
	    continue;
	}

	carrier_nsamples += cfg.frame_nsamples;			// minimodem.c:1324
	uint32_t flags = 0;
	if ( carrier ) {
	    carrier_nsamples += frame_start;			// minimodem.c:1329-1330
	    carrier_nsamples -= cfg.overscan;
	} else {
	    carrier = true;					// minimodem.c:1350-1353
	    refine = true;
	    flags |= MIFSK_FRAME_ACQUIRE;
	    ep_first = n_out_frames;
	}

	if ( refine && confidence < INFINITY && try_step > 1u ) {	// minimodem.c:1357-1389
	    // `carrier` is already set: an acquiring frame is re-searched with
	    // the data string over the no-carrier range (minimodem.c:1378)
	    ScanResult s2 = ctx.scan(base, try_first, try_max, cfg.try_step_fine[ci],
				     INFINITY, 0u, false);
	    flags |= MIFSK_FRAME_REFINED;
	    if ( s2.conf > confidence ) {
		bits = s2.bits;
		amplitude = s2.ampl;
		frame_start = s2.start;
	    }
	}

	track_amplitude = ( track_amplitude + amplitude ) / 2.0f;	// minimodem.c:1391-1400
	if ( peak_confidence < confidence )
	    peak_confidence = confidence;
	confidence_total += confidence;
	amplitude_total += amplitude;
	nframes_decoded++;
	noconfidence = 0;

	advance = frame_start + cfg.frame_nsamples - cfg.overscan;	// minimodem.c:1407

	if ( cfg.has_stopbits )						// minimodem.c:1415-1428
	    bits >>= 1;
	bits = bit_window(bits, cfg.nstartbits, cfg.n_data_bits);
	if ( cfg.msb_first )
	    bits = bit_reverse(bits, cfg.n_data_bits);

	const bool suppressed = cfg.do_rx_sync && bits == cfg.sync_byte;	// minimodem.c:1436-1439
	if ( suppressed )
	    flags |= MIFSK_FRAME_SYNC;

	if ( t0 ) {
	    if ( n_out_frames < fcap ) {
		if ( o_bits )
		    o_bits[n_out_frames] = bits;
		if ( o_frames ) {
		    mifsk_frame f;
		    f.bits = bits;
		    f.start = (uint64_t)base + frame_start;
		    f.confidence = confidence;
		    f.amplitude = amplitude;
		    f.flags = flags;
		    f.reserved = 0;
		    o_frames[n_out_frames] = f;
		}
	    }
	    if ( !suppressed && o_bytes && n_out_bytes < fcap )
		o_bytes[n_out_bytes] = (uint8_t)( bits & 0xFFu );
	}
	n_out_frames++;
	if ( !suppressed )
	    n_out_bytes++;
    }

    if ( carrier ) {						// minimodem.c:1469-1474
	if ( t0 && o_eps && n_out_eps < ecap ) {
	    mifsk_episode e;
	    e.carrier_nsamples = carrier_nsamples;
	    e.first_frame = ep_first;
	    e.nframes = nframes_decoded;
	    e.confidence_total = confidence_total;
	    e.amplitude_total = amplitude_total;
	    e.end_reason = 2;
	    e.reserved = 0;
	    o_eps[n_out_eps] = e;
	}
	n_out_eps++;
    }
    if ( t0 ) {
	if ( n_out_frames > fcap && ( o_bits || o_frames || o_bytes ) )
	    status |= MIFSK_STREAM_FRAMES_TRUNCATED;
	if ( n_out_eps > ecap && o_eps )
	    status |= MIFSK_STREAM_EPISODES_TRUNCATED;
	if ( io.d_nframes ) io.d_nframes[s] = n_out_frames;
	if ( io.d_nbytes ) io.d_nbytes[s] = n_out_bytes;
	if ( io.d_nepisodes ) io.d_nepisodes[s] = n_out_eps;
	if ( io.d_status ) io.d_status[s] = status;
    }
}

// ---------------------------------------------------------------------------
// kernel 3: full-spectrum magnitudes for fsk_detect_carrier (fsk.c:543-581)
// one thread per band; window of <= fftsize samples
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256)
void spectrum_kernel( const float *__restrict__ x, uint32_t nsamples,
	const double *__restrict__ cs, uint32_t fftsize, uint32_t nbands,
	float *__restrict__ mags )
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if ( b >= nbands )
	return;
    double re = 0.0, im = 0.0;
    uint32_t k = 0;				// (b * n) mod fftsize, incrementally
    for ( uint32_t n = 0; n < nsamples; n++ ) {
	const double xd = (double)x[n];
	re = fma(xd, cs[2 * (size_t)k], re);
	im = fma(xd, cs[2 * (size_t)k + 1], im);
	k += b;
	if ( k >= fftsize )
	    k -= fftsize;
    }
    const float magscalar = 1.0f / ( (float)nsamples / 2.0f );	// fsk.c:553
    mags[b] = band_mag(re, im, magscalar);
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------

static inline int hip_rc( hipError_t e )
{
    return e == hipSuccess ? 0 : -5 /* -EIO */;
}

int launch_find_frame_batch( const DevCfg &cfg, const double *d_tw,
	const float *d_samples, const mifsk_search *d_problems,
	mifsk_search_result *d_results, int nproblems, void *stream )
{
    if ( nproblems <= 0 )
	return 0;
    hipLaunchKernelGGL(find_frame_kernel, dim3((unsigned)nproblems), dim3(64), 0,
		       (hipStream_t)stream, cfg, d_tw, d_samples, d_problems, d_results);
    return hip_rc(hipGetLastError());
}

static constexpr size_t kLdsHeader = offsetof(StreamLds, slab);
static constexpr size_t kLdsPerCu = 160 * 1024;

int launch_demod_batch( const DevCfg &cfg, const double *d_tw,
	const mifsk_demod_io &io, void *stream )
{
    if ( io.nstreams <= 0 )
	return 0;
    const uint32_t B = cfg.bit_nsamples;
    // samples one search must see at once
    const uint32_t reach = ( cfg.try_max[0] > cfg.try_max[1] ? cfg.try_max[0] : cfg.try_max[1] )
			 + cfg.last_reach + 8;
    uint32_t qmax = W_CAP / cfg.n_bits;
    if ( qmax > P_CAP ) qmax = P_CAP;
    // run-ahead depth: fill one pass of BLOCK lanes with bit windows
    uint32_t coarse_j = 3;	// typical carrier coarse scan (first, +step, -step)
    uint32_t want = BLOCK / cfg.n_bits > coarse_j ? BLOCK / cfg.n_bits - coarse_j : 0;
    if ( want > qmax - coarse_j ) want = qmax > coarse_j ? qmax - coarse_j : 0;

    // LDS budget: 4 workgroups per CU when the stream count can use them
    const size_t budget_small = kLdsPerCu / 4 - 64;
    auto floats_for = [&]( uint32_t nsamp ) -> size_t {
	return (size_t)nsamp + (size_t)( nsamp / B + 1 ) * cfg.skew + 8;
    };
    auto samples_in = [&]( size_t bytes ) -> uint32_t {
	if ( bytes <= kLdsHeader + 64 ) return 0;
	size_t fl = ( bytes - kLdsHeader ) / 4;
	size_t ns = fl * B / ( B + cfg.skew );
	ns = ns > 16 ? ns - 16 : 0;
	return (uint32_t)( ns & ~(size_t)3 );
    };

    uint32_t slab_cap = 0, npredict = 0;
    bool use_slab = true;
    const uint32_t ideal = reach + want * cfg.frame_nsamples;
    if ( kLdsHeader + floats_for(ideal) * 4 <= budget_small ) {
	slab_cap = ( ideal + 3 ) & ~3u;
	npredict = want;
    } else if ( samples_in(budget_small) >= reach + cfg.frame_nsamples ) {
	slab_cap = samples_in(budget_small);
	npredict = ( slab_cap - reach ) / cfg.frame_nsamples;
	if ( npredict > want ) npredict = want;
    } else {
	// long bit windows: take as much LDS as one search needs (fewer WGs/CU)
	const size_t need = kLdsHeader + floats_for(reach + 4) * 4;
	if ( need <= kLdsPerCu - 1024 ) {
	    slab_cap = ( reach + 4 + 3 ) & ~3u;
	    npredict = 0;
	} else {
	    use_slab = false;	// e.g. 0.5 baud: windows of 96000 samples
	}
    }

    hipStream_t st = (hipStream_t)stream;
    if ( use_slab ) {
	const size_t lds_bytes = kLdsHeader + floats_for(slab_cap) * 4;
	hipError_t e = hipFuncSetAttribute(
		reinterpret_cast<const void *>(&demod_kernel<true>),
		hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
	if ( e != hipSuccess )
	    return hip_rc(e);
	hipLaunchKernelGGL(demod_kernel<true>, dim3((unsigned)io.nstreams), dim3(BLOCK),
			   lds_bytes, st, cfg, d_tw, io, slab_cap, npredict);
    } else {
	hipLaunchKernelGGL(demod_kernel<false>, dim3((unsigned)io.nstreams), dim3(BLOCK),
			   kLdsHeader + 16, st, cfg, d_tw, io, 0u, 0u);
    }
    return hip_rc(hipGetLastError());
}

int launch_detect_carrier( const float *d_samples, unsigned nsamples,
	const double *d_cs, unsigned fftsize, unsigned nbands, float *d_mags, void *stream )
{
    const unsigned blocks = ( nbands + 255 ) / 256;
    hipLaunchKernelGGL(spectrum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
		       d_samples, nsamples, d_cs, fftsize, nbands, d_mags);
    return hip_rc(hipGetLastError());
}

} // namespace mifsk
