// mifsk_ingest.hip -- the step before the path: what the reference
// does between the file and samplebuf[] -- libsndfile's S16 -> float conversion
// (simpleaudio-sndfile.c:42-56, sf_readf_float on a PCM16 file) and the
// --Xrxnoise term (simpleaudio-sndfile.c:64-69) -- on the device, over a whole
// batch of streams, so that 16-bit recordings cross PCIe and are read from HBM
// at 2 bytes per sample; plus the RIFF/WAVE header parse for the batched file
// loader.  gfx950 only.
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cstdint>
#include <cstring>

#include "mifsk.h"
#include "mifsk_device.h"

namespace mifsk {

// One thread converts 8 consecutive samples: one 16-byte load, two 16-byte
// stores.  HBM-bound: 2 B read + 4 B written per sample.  Samples at or beyond
// the stream's length are written as 0.0 up to the row's stride, so the whole
// row is defined afterwards.
//   value / 32768 : libsndfile's normalisation of PCM16 (a power of two: exact)
//   + dc          : what --Xrxnoise really adds -- `rand()/RAND_MAX` is an
//                   integer division, i.e. 0, so the "noise" is the constant
//                   (0 - 0.5f) * (2 * factor)
__global__ __launch_bounds__(256)
void ingest_s16_kernel( const int16_t *__restrict__ pcm, size_t pcm_stride,
	float *__restrict__ out, size_t out_stride,
	const uint32_t *__restrict__ nsamples_v, uint32_t nsamples_u, float dc, int vec_ok )
{
    const uint32_t s = blockIdx.y;
    const uint32_t n = nsamples_v ? nsamples_v[s] : nsamples_u;
    const int16_t *row = pcm + (size_t)s * pcm_stride;
    float *dst = out + (size_t)s * out_stride;
    const size_t i0 = ( (size_t)blockIdx.x * blockDim.x + threadIdx.x ) * 8u;
    if ( i0 >= out_stride )
	return;
    float v[8];
    if ( vec_ok && i0 + 8 <= n ) {
	const int4 raw = *reinterpret_cast<const int4 *>(row + i0);	// 8 x int16, coalesced
	const int w[4] = { raw.x, raw.y, raw.z, raw.w };
#pragma unroll
	for ( int k = 0; k < 4; k++ ) {
	    v[2 * k] = (float)(int16_t)( w[k] & 0xFFFF ) / 32768.0f + dc;
	    v[2 * k + 1] = (float)(int16_t)( (uint32_t)w[k] >> 16 ) / 32768.0f + dc;
	}
    } else {
#pragma unroll
	for ( int k = 0; k < 8; k++ )
	    v[k] = i0 + k < n ? (float)row[i0 + k] / 32768.0f + dc : 0.0f;
    }
    if ( vec_ok && i0 + 8 <= out_stride ) {
	*reinterpret_cast<float4 *>(dst + i0) = make_float4(v[0], v[1], v[2], v[3]);
	*reinterpret_cast<float4 *>(dst + i0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
	for ( int k = 0; k < 8; k++ )
	    if ( i0 + k < out_stride )
		dst[i0 + k] = v[k];
    }
}

// --Xrxnoise on input that is already float: x += dc over the valid samples
__global__ __launch_bounds__(256)
void offset_f32_kernel( float *__restrict__ x, size_t stride,
	const uint32_t *__restrict__ nsamples_v, uint32_t nsamples_u, float dc )
{
    const uint32_t s = blockIdx.y;
    const uint32_t n = nsamples_v ? nsamples_v[s] : nsamples_u;
    float *row = x + (size_t)s * stride;
    const size_t i0 = ( (size_t)blockIdx.x * blockDim.x + threadIdx.x ) * 4u;
    for ( int k = 0; k < 4; k++ )
	if ( i0 + k < n )
	    row[i0 + k] += dc;
}

} // namespace mifsk

// (0 - 0.5f) * (factor * 2), in float as at simpleaudio-sndfile.c:67-69
static float rxnoise_term( float factor )
{
    if ( factor == 0.0f )
	return 0.0f;
    const float f = factor * 2;
    return ( 0 - 0.5f ) * f;
}

extern "C" int mifsk_ingest_s16( mifsk_ctx *ctx, const int16_t *d_pcm, size_t pcm_stride,
	float *d_samples, size_t stream_stride, const uint32_t *d_nsamples, uint32_t nsamples,
	int nstreams, float rxnoise, void *stream )
{
    if ( !ctx || !d_pcm || !d_samples || nstreams < 0 )
	return -EINVAL;
    if ( nstreams == 0 || stream_stride == 0 )
	return 0;
    if ( !d_nsamples && ( nsamples > pcm_stride || nsamples > stream_stride ) )
	return -EINVAL;
    if ( hipSetDevice(mifsk::ctx_device(ctx)) != hipSuccess )
	return -EIO;
    const int vec_ok = pcm_stride % 8 == 0 && stream_stride % 4 == 0
		     && (uintptr_t)d_pcm % 16 == 0 && (uintptr_t)d_samples % 16 == 0;
    const size_t per_block = 256 * 8;
    dim3 grid((unsigned)( ( stream_stride + per_block - 1 ) / per_block ), (unsigned)nstreams);
    hipLaunchKernelGGL(mifsk::ingest_s16_kernel, grid, dim3(256), 0, (hipStream_t)stream,
		       d_pcm, pcm_stride, d_samples, stream_stride, d_nsamples, nsamples,
		       rxnoise_term(rxnoise), vec_ok);
    return hipGetLastError() == hipSuccess ? 0 : -EIO;
}

extern "C" int mifsk_ingest_rxnoise_f32( mifsk_ctx *ctx, float *d_samples, size_t stream_stride,
	const uint32_t *d_nsamples, uint32_t nsamples, int nstreams, float rxnoise, void *stream )
{
    if ( !ctx || !d_samples || nstreams < 0 )
	return -EINVAL;
    if ( nstreams == 0 || stream_stride == 0 || rxnoise == 0.0f )
	return 0;
    if ( hipSetDevice(mifsk::ctx_device(ctx)) != hipSuccess )
	return -EIO;
    const size_t per_block = 256 * 4;
    dim3 grid((unsigned)( ( stream_stride + per_block - 1 ) / per_block ), (unsigned)nstreams);
    hipLaunchKernelGGL(mifsk::offset_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream,
		       d_samples, stream_stride, d_nsamples, nsamples, rxnoise_term(rxnoise));
    return hipGetLastError() == hipSuccess ? 0 : -EIO;
}

// ---- RIFF/WAVE header (host) -------------------------------------------------
static uint32_t rd32( const unsigned char *p ) { return p[0] | p[1] << 8 | p[2] << 16 | (uint32_t)p[3] << 24; }
static uint32_t rd16( const unsigned char *p ) { return p[0] | p[1] << 8; }

// `have` bytes of the file are in memory at `file`, the file itself is `len` bytes long
// (the batched loader reads headers only)
namespace mifsk { int wav_parse_sized( const void *file, size_t have, size_t len, mifsk_wav_info *info ); }

extern "C" int mifsk_wav_parse( const void *file, size_t len, mifsk_wav_info *info )
{
    return mifsk::wav_parse_sized(file, len, len, info);
}

int mifsk::wav_parse_sized( const void *file, size_t have, size_t len, mifsk_wav_info *info )
{
    if ( !file || !info || have > len )
	return -EINVAL;
    memset(info, 0, sizeof *info);
    const unsigned char *p = (const unsigned char *)file;
    if ( have < 12 || memcmp(p, "RIFF", 4) != 0 || memcmp(p + 8, "WAVE", 4) != 0 )
	return -EINVAL;
    bool have_fmt = false;
    unsigned block_align = 0;
    size_t pos = 12;
    while ( pos + 8 <= have ) {
	const uint32_t sz = rd32(p + pos + 4);
	const unsigned char *body = p + pos + 8;
	const size_t avail = len - ( pos + 8 );		// of the file
	const size_t here = have - ( pos + 8 );		// of it, in memory
	if ( memcmp(p + pos, "fmt ", 4) == 0 ) {
	    if ( sz < 16 || here < 16 )
		return -EINVAL;
	    unsigned tag = rd16(body);
	    info->channels = rd16(body + 2);
	    info->sample_rate = rd32(body + 4);
	    block_align = rd16(body + 12);
	    info->bits_per_sample = rd16(body + 14);
	    if ( tag == 0xFFFE && sz >= 26 && here >= 26 )	// WAVE_FORMAT_EXTENSIBLE: sub-format GUID
		tag = rd16(body + 24);
	    if ( tag == 1 && info->bits_per_sample == 16 )
		info->is_float = 0;
	    else if ( tag == 3 && info->bits_per_sample == 32 )
		info->is_float = 1;
	    else
		return -ENOTSUP;				// the reference's tests use PCM16 and float32 only
	    have_fmt = true;
	} else if ( memcmp(p + pos, "data", 4) == 0 ) {
	    if ( !have_fmt || block_align == 0 )
		return -EINVAL;
	    info->data_offset = pos + 8;
	    const size_t bytes = sz <= avail ? sz : avail;	// a truncated file yields what is there
	    info->nframes = bytes / block_align;
	    return info->channels == 1 ? 0 : -ENOTSUP;		// minimodem opens its input mono
	}
	pos += 8 + (size_t)sz + ( sz & 1u );
    }
    return -EINVAL;
}
