// Micro-benchmark (gfx950): how a wavefront should get at the samples of LONG bit
// windows (RTTY: 1056 samples per bit, 8 bit windows per candidate, 4 coarse /
// 8 fine candidates 528 / 198 samples apart -- one window per lane, 64 windows
// advancing in lockstep, so one load instruction touches 64 different cache lines
// and the candidates re-read the same 40 kB span 4-8 times).  Variants:
//   G<BURST>  every lane streams its own window from global memory, BURST samples
//             (BURST/4 dwordx4 loads) per step, the next step in flight
//   S         the span is staged once per frame into LDS (coalesced), windows are
//             read from there (ds_read_b32, one pad word per bit row)
// The arithmetic is the demod kernel's: 4 f64 FMAs per sample with the factor
// broadcast from a lane of the row (v_fmac_f64_dpp row_newbcast), the table group
// of the next 16 samples in flight.  Result: ms for a 4096-stream x 181-frame batch.
//   hipcc --offload-arch=gfx950 -O3 -o longwin longwin.hip && ./longwin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if ( e != hipSuccess ) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int B = 1056;			// samples per bit
constexpr int NB = 8;			// bit windows per candidate
constexpr int ADV = 7920;		// cursor advance per frame
constexpr int SPAN = 1584 + NB * B;	// what one frame's searches read

typedef float __attribute__((ext_vector_type(4), aligned(4))) f4u;

struct Tw { double v[4]; };

template <int J> __device__ __forceinline__ void fb( double &acc, const double &w, double xd )
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
	: "+v"(acc) : "v"(w), "v"(xd), "i"(J));
}
template <int J> __device__ __forceinline__ void fma4( double (&a)[4], const Tw &t, float x )
{
    const double xd = (double)x;
    fb<J>(a[0], t.v[0], xd); fb<J>(a[1], t.v[1], xd); fb<J>(a[2], t.v[2], xd); fb<J>(a[3], t.v[3], xd);
}
__device__ __forceinline__ void group16( double (&a)[4], const Tw &t, const float (&x)[16] )
{
    asm volatile("s_nop 4");
    fma4<0>(a, t, x[0]); fma4<1>(a, t, x[1]); fma4<2>(a, t, x[2]); fma4<3>(a, t, x[3]);
    fma4<4>(a, t, x[4]); fma4<5>(a, t, x[5]); fma4<6>(a, t, x[6]); fma4<7>(a, t, x[7]);
    fma4<8>(a, t, x[8]); fma4<9>(a, t, x[9]); fma4<10>(a, t, x[10]); fma4<11>(a, t, x[11]);
    fma4<12>(a, t, x[12]); fma4<13>(a, t, x[13]); fma4<14>(a, t, x[14]); fma4<15>(a, t, x[15]);
}
__device__ __forceinline__ Tw tw_load( const double *tw, int g, int lane )
{
    const double *p = tw + 4 * ( 16 * g + ( lane & 15 ) );
    Tw t; t.v[0] = p[0]; t.v[1] = p[1]; t.v[2] = p[2]; t.v[3] = p[3];
    return t;
}

// window start of a lane, relative to the cursor: candidate lane/8, bit lane%8
__device__ int g_bit_stride = B;	// (32 with candidate steps of 256: every window 128 bytes after its neighbour -- the
					//  same 64-lines-per-load pattern, but everything stays in L1/L2)
__device__ __forceinline__ int win_rel( int lane, int step ) { return ( lane >> 3 ) * step + ( lane & 7 ) * g_bit_stride; }

template <int BURST>
__device__ __forceinline__ void pass_global( const float *xa, const double *tw, int lane, double (&acc)[4] )
{
    constexpr int NG = B / BURST, G16 = BURST / 16;
    float cur[BURST], nxt[BURST];
#pragma unroll
    for ( int i = 0; i < BURST / 4; i++ ) {
	const f4u v = *reinterpret_cast<const f4u *>(xa + 4 * i);
	cur[4 * i] = v.x; cur[4 * i + 1] = v.y; cur[4 * i + 2] = v.z; cur[4 * i + 3] = v.w;
    }
    Tw T = tw_load(tw, 0, lane);
    for ( int g = 0; g < NG; g++ ) {
	const int gn = g + 1 < NG ? g + 1 : g;
#pragma unroll
	for ( int i = 0; i < BURST / 4; i++ ) {
	    const f4u v = *reinterpret_cast<const f4u *>(xa + BURST * gn + 4 * i);
	    nxt[4 * i] = v.x; nxt[4 * i + 1] = v.y; nxt[4 * i + 2] = v.z; nxt[4 * i + 3] = v.w;
	}
#pragma unroll
	for ( int h = 0; h < G16; h++ ) {
	    const int gi = g * G16 + h + 1;
	    const Tw Tn = tw_load(tw, gi < B / 16 ? gi : B / 16 - 1, lane);
	    float x16[16];
#pragma unroll
	    for ( int j = 0; j < 16; j++ ) x16[j] = cur[16 * h + j];
	    group16(acc, T, x16);
	    T = Tn;
	}
#pragma unroll
	for ( int j = 0; j < BURST; j++ ) cur[j] = nxt[j];
    }
}

template <int BURST>
__global__ __launch_bounds__(64) void k_global( const float *x, size_t stride, const double *tw, int nframes,
						float *out, int cstep, int fstep )
{
    const int lane = threadIdx.x;
    const float *xs = x + (size_t)blockIdx.x * stride;
    double sum = 0.0;
    for ( int f = 0; f < nframes; f++ ) {
	const float *cur = xs + (size_t)f * ADV + ( blockIdx.x & 31 );
	{   // coarse: 4 candidates x 8 bits = 32 lanes busy
	    double acc[4] = { 0, 0, 0, 0 };
	    const int l = lane < 32 ? lane : 0;	// (idle lanes sit on window 0)
	    pass_global<BURST>(cur + win_rel(l, cstep), tw, lane, acc);
	    sum += acc[0] + acc[1] + acc[2] + acc[3];
	}
	if ( ( f + blockIdx.x ) % 9 < 4 ) {		// fine: 8 candidates x 8 bits
	    double acc[4] = { 0, 0, 0, 0 };
	    pass_global<BURST>(cur + win_rel(lane, fstep), tw, lane, acc);
	    sum += acc[0] + acc[1] + acc[2] + acc[3];
	}
    }
    out[blockIdx.x * 64 + lane] = (float)sum;
}

// LDS variant: row r of the span = samples [r*B, (r+1)*B) at words r*(B+1) ...
__device__ __forceinline__ int slab_word( int s ) { return s + s / B; }

template <int SV>
__device__ __forceinline__ void stage( float *slab, const float *src, int lane )
{
    // src 16-byte aligned here (the real kernel aligns down and skips the head)
    for ( int v0 = 0; v0 < SPAN / 4; v0 += 64 * SV ) {
	float4 buf[SV];
#pragma unroll
	for ( int i = 0; i < SV; i++ ) {
	    const int v = v0 + 64 * i + lane;
	    if ( v < SPAN / 4 ) buf[i] = *reinterpret_cast<const float4 *>(src + 4 * v);
	}
#pragma unroll
	for ( int i = 0; i < SV; i++ ) {
	    const int v = v0 + 64 * i + lane;
	    if ( v < SPAN / 4 ) {
		const int w = slab_word(4 * v);		// (B % 4 == 0: a float4 never straddles rows)
		slab[w] = buf[i].x; slab[w + 1] = buf[i].y; slab[w + 2] = buf[i].z; slab[w + 3] = buf[i].w;
	    }
	}
    }
}

__device__ __forceinline__ void pass_slab( const float *slab, int rel, const double *tw, int lane, double (&acc)[4] )
{
    const int row = rel / B, col = rel - row * B;
    const float *p = slab + rel + row;
    const int wrap = B - col;
    float cur[16], nxt[16];
#pragma unroll
    for ( int j = 0; j < 16; j++ ) cur[j] = p[j + ( j >= wrap ? 1 : 0 )];
    Tw T = tw_load(tw, 0, lane);
    Tw T1 = tw_load(tw, 1, lane);
    for ( int g = 0; g < B / 16; g++ ) {
	const int gn = g + 1 < B / 16 ? g + 1 : g;
	const int g2 = g + 2 < B / 16 ? g + 2 : B / 16 - 1;
	const Tw T2 = tw_load(tw, g2, lane);
#pragma unroll
	for ( int j = 0; j < 16; j++ ) {
	    const int n = 16 * gn + j;
	    nxt[j] = p[n + ( n >= wrap ? 1 : 0 )];
	}
	group16(acc, T, cur);
	T = T1; T1 = T2;
#pragma unroll
	for ( int j = 0; j < 16; j++ ) cur[j] = nxt[j];
    }
}

template <int SV>
__global__ __launch_bounds__(64) void k_slab( const float *x, size_t stride, const double *tw, int nframes,
					      float *out )
{
    extern __shared__ float slab[];
    const int lane = threadIdx.x;
    const float *xs = x + (size_t)blockIdx.x * stride;
    double sum = 0.0;
    for ( int f = 0; f < nframes; f++ ) {
	const float *cur = xs + (size_t)f * ADV;
	stage<SV>(slab, cur, lane);
	const int head = blockIdx.x & 31;
	{
	    double acc[4] = { 0, 0, 0, 0 };
	    pass_slab(slab, head + win_rel(lane < 32 ? lane : 0, 528), tw, lane, acc);
	    sum += acc[0] + acc[1] + acc[2] + acc[3];
	}
	if ( ( f + blockIdx.x ) % 9 < 4 ) {
	    double acc[4] = { 0, 0, 0, 0 };
	    pass_slab(slab, head + win_rel(lane, 198), tw, lane, acc);
	    sum += acc[0] + acc[1] + acc[2] + acc[3];
	}
    }
    out[blockIdx.x * 64 + lane] = (float)sum;
}

// Tiled variant: the wave fetches 16 samples of each of its 64 windows with FOUR
// loads whose quads read 64 contiguous bytes of one window (16 cycles of the
// address path each instead of ~80), passes them through a 5 kB LDS tile
// ([window][16 samples], 80-byte rows) and every lane reads its own row back
// with four aligned ds_read_b128.  One tile, no barrier: the wave's LDS
// operations execute in order.
constexpr int TROW = 20;
__device__ __forceinline__ void tile_write( float *tile, int lane, const float4 (&L)[4], int nq )
{
#pragma unroll
    for ( int i = 0; i < 4; i++ )
	if ( i < nq )
	    *reinterpret_cast<float4 *>(tile + ( 16 * i + ( lane >> 2 ) ) * TROW + 4 * ( lane & 3 )) = L[i];
}
__device__ __forceinline__ void tile_read( const float *tile, int lane, float (&x)[16] )
{
#pragma unroll
    for ( int j = 0; j < 4; j++ ) {
	const float4 v = *reinterpret_cast<const float4 *>(tile + lane * TROW + 4 * j);
	x[4 * j] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
    }
}
__device__ __forceinline__ void tile_fetch( const float *const (&src)[4], int g, float4 (&L)[4], int nq )
{
#pragma unroll
    for ( int i = 0; i < 4; i++ )
	if ( i < nq ) {
	    const f4u v = *reinterpret_cast<const f4u *>(src[i] + 16 * g);
	    L[i] = make_float4(v.x, v.y, v.z, v.w);
	}
}

// nq = quarters of the wave that hold windows (coarse: 2, fine: 4); DEPTH = groups
// of global loads in flight beyond the one in the tile
template <int DEPTH>
__device__ __forceinline__ void pass_tiled( const float *cur, int step, int nq, const double *tw, float *tile,
					    int lane, double (&acc)[4] )
{
    constexpr int NG = B / 16;
    const float *src[4];
#pragma unroll
    for ( int i = 0; i < 4; i++ )
	src[i] = cur + win_rel(16 * i + ( lane >> 2 ), step) + 4 * ( lane & 3 );
    float4 L[DEPTH][4];
    float X[2][16];
    tile_fetch(src, 0, L[0], nq);
    tile_write(tile, lane, L[0], nq);
#pragma unroll
    for ( int d = 0; d < DEPTH; d++ )
	tile_fetch(src, 1 + d, L[d], nq);		// groups 1 .. DEPTH
    tile_read(tile, lane, X[0]);
    Tw T = tw_load(tw, 0, lane);
    static_assert(NG % ( 2 * DEPTH ) == 0 || DEPTH == 1 || true, "");
    for ( int g0 = 0; g0 < NG; g0 += 2 * DEPTH ) {
#pragma unroll
	for ( int u = 0; u < 2 * DEPTH; u++ ) {
	    const int g = g0 + u;
	    if ( g < NG ) {
		const Tw Tn = tw_load(tw, g + 1 < NG ? g + 1 : g, lane);
		tile_write(tile, lane, L[u % DEPTH], nq);			// group g+1
		tile_read(tile, lane, X[( u + 1 ) & 1]);
		tile_fetch(src, g + 1 + DEPTH < NG ? g + 1 + DEPTH : g, L[u % DEPTH], nq);
		group16(acc, T, X[u & 1]);
		T = Tn;
	    }
	}
    }
}

template <int DEPTH>
__global__ __launch_bounds__(64) void k_tiled( const float *x, size_t stride, const double *tw, int nframes,
					       float *out, int cstep, int fstep )
{
    extern __shared__ float slab[];
    const int lane = threadIdx.x;
    const float *xs = x + (size_t)blockIdx.x * stride;
    double sum = 0.0;
    for ( int f = 0; f < nframes; f++ ) {
	const float *cur = xs + (size_t)f * ADV + ( blockIdx.x & 31 );
	{
	    double acc[4] = { 0, 0, 0, 0 };
	    pass_tiled<DEPTH>(cur, cstep, 2, tw, slab, lane, acc);
	    sum += acc[0] + acc[1] + acc[2] + acc[3];
	}
	if ( ( f + blockIdx.x ) % 9 < 4 ) {
	    double acc[4] = { 0, 0, 0, 0 };
	    pass_tiled<DEPTH>(cur, fstep, 4, tw, slab, lane, acc);
	    sum += acc[0] + acc[1] + acc[2] + acc[3];
	}
    }
    out[blockIdx.x * 64 + lane] = (float)sum;
}

// arithmetic only: the samples never change (what the f64 pipe alone allows)
template <bool CVT>
__global__ __launch_bounds__(64) void k_compute( const double *tw, int npasses, float *out )
{
    const int lane = threadIdx.x;
    double sum = 0.0;
    float x[16];
#pragma unroll
    for ( int j = 0; j < 16; j++ ) x[j] = 1.0f + lane + j;
    for ( int f = 0; f < npasses; f++ ) {
	double acc[4] = { 0, 0, 0, 0 };
	Tw T = tw_load(tw, 0, lane);
	for ( int g = 0; g < B / 16; g++ ) {
	    const Tw Tn = tw_load(tw, g + 1 < B / 16 ? g + 1 : g, lane);
	    if ( CVT ) {
#pragma unroll
		for ( int j = 0; j < 16; j++ ) asm volatile("" : "+v"(x[j]));
		group16(acc, T, x);
	    } else {
		double xd[16];
#pragma unroll
		for ( int j = 0; j < 16; j++ ) { xd[j] = x[j]; asm volatile("" : "+v"(xd[j])); }
		asm volatile("s_nop 4");
#pragma unroll
		for ( int j = 0; j < 16; j++ ) {
		    asm volatile("v_fmac_f64_dpp %0, %4, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
				 "v_fmac_f64_dpp %1, %5, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
				 "v_fmac_f64_dpp %2, %6, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
				 "v_fmac_f64_dpp %3, %7, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf"
			: "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
			: "v"(T.v[0]), "v"(T.v[1]), "v"(T.v[2]), "v"(T.v[3]), "v"(xd[j]));
		}
	    }
	    T = Tn;
	}
	sum += acc[0] + acc[1] + acc[2] + acc[3];
    }
    out[blockIdx.x * 64 + lane] = (float)sum;
}

// what each ingredient of the tiled pass adds to the arithmetic: MODE 1 = + the
// global loads (landing in registers nobody reads), 2 = + tile write / read
// (results dropped), 3 = the FMAs take the samples read from the tile
template <int MODE>
__global__ __launch_bounds__(64) void k_mix( const float *x, size_t stride, const double *tw, int nframes,
					     float *out, int cstep, int fstep )
{
    extern __shared__ float slab[];
    const int lane = threadIdx.x;
    const float *xs = x + (size_t)blockIdx.x * stride;
    constexpr int NG = B / 16;
    double sum = 0.0;
    float xc[16];
#pragma unroll
    for ( int j = 0; j < 16; j++ ) xc[j] = 1.0f + lane + j;
    for ( int f = 0; f < 2 * nframes; f++ ) {
	const bool fine = f & 1;
	if ( fine && ( ( f >> 1 ) + blockIdx.x ) % 9 >= 4 )
	    continue;
	const float *cur = xs + (size_t)( f >> 1 ) * ADV + ( blockIdx.x & 31 );
	const int nq = fine ? 4 : 2, step = fine ? fstep : cstep;
	const float *src[4];
#pragma unroll
	for ( int i = 0; i < 4; i++ )
	    src[i] = cur + win_rel(16 * i + ( lane >> 2 ), step) + 4 * ( lane & 3 );
	double acc[4] = { 0, 0, 0, 0 };
	float4 L[4];
	float X[16];
	tile_fetch(src, 0, L, nq);
	Tw T = tw_load(tw, 0, lane);
	for ( int g = 0; g < NG; g++ ) {
	    const Tw Tn = tw_load(tw, g + 1 < NG ? g + 1 : g, lane);
	    if ( MODE >= 2 ) {
		tile_write(slab, lane, L, nq);
		tile_read(slab, lane, X);
	    } else {
#pragma unroll
		for ( int i = 0; i < 4; i++ ) asm volatile("" :: "v"(L[i].x), "v"(L[i].y), "v"(L[i].z), "v"(L[i].w));
	    }
	    tile_fetch(src, g + 1 < NG ? g + 1 : g, L, nq);
	    if ( MODE >= 3 ) {
		group16(acc, T, X);
	    } else {
		if ( MODE == 2 ) {
#pragma unroll
		    for ( int j = 0; j < 16; j++ ) asm volatile("" :: "v"(X[j]));
		}
#pragma unroll
		for ( int j = 0; j < 16; j++ ) asm volatile("" : "+v"(xc[j]));
		group16(acc, T, xc);
	    }
	    T = Tn;
	}
	sum += acc[0] + acc[1] + acc[2] + acc[3];
    }
    out[blockIdx.x * 64 + lane] = (float)sum;
}

// Tiled, K samples per window per step (K/4 lanes read K*4 contiguous bytes of one
// window): fewer, longer runs per cache line fetched.  Tile rows of K+4 floats.
template <int K>
__device__ __forceinline__ void tileK_fetch( const float *cur, const uint32_t (&off)[K / 4], int s, float4 (&L)[K / 4], int nld )
{
#pragma unroll
    for ( int i = 0; i < K / 4; i++ )
	if ( i < nld ) {
	    const f4u v = *reinterpret_cast<const f4u *>(cur + off[i] + K * s);
	    L[i] = make_float4(v.x, v.y, v.z, v.w);
	}
}
template <int J> __device__ __forceinline__ void fma2( double (&a)[4], const Tw &t, float x )
{
    const double xd = (double)x;
    fb<J>(a[0], t.v[0], xd); fb<J>(a[1], t.v[1], xd);
}
__device__ __forceinline__ void group16_2( double (&a)[4], const Tw &t, const float (&x)[16] )
{
    asm volatile("s_nop 4");
    fma2<0>(a, t, x[0]); fma2<1>(a, t, x[1]); fma2<2>(a, t, x[2]); fma2<3>(a, t, x[3]);
    fma2<4>(a, t, x[4]); fma2<5>(a, t, x[5]); fma2<6>(a, t, x[6]); fma2<7>(a, t, x[7]);
    fma2<8>(a, t, x[8]); fma2<9>(a, t, x[9]); fma2<10>(a, t, x[10]); fma2<11>(a, t, x[11]);
    fma2<12>(a, t, x[12]); fma2<13>(a, t, x[13]); fma2<14>(a, t, x[14]); fma2<15>(a, t, x[15]);
}
__device__ __forceinline__ Tw tw_load2( const double *tw, int g, int lane )
{
    const double *p = tw + 4 * ( 16 * g + ( lane & 15 ) ) + ( lane < 32 ? 0 : 2 );
    Tw t; t.v[0] = p[0]; t.v[1] = p[1]; t.v[2] = 0; t.v[3] = 0;
    return t;
}

// SPLIT: at most 32 windows -- lanes 32..63 take the second tone of windows 0..31
template <int K, bool SPLIT = false>
__device__ __forceinline__ void pass_tiledK( const float *cur, int step, int nwin, const double *tw, float *tile,
					     int lane, double (&acc)[4] )
{
    constexpr int LPW = K / 4;			// lanes per window in a load
    constexpr int WPL = 64 / LPW;		// windows per load
    constexpr int NLD = 64 / WPL;		// loads per step (= K / 4)
    constexpr int TR = K + 4;
    constexpr int NS = B / K;			// steps
    const int nld = ( nwin + WPL - 1 ) / WPL;
    uint32_t off[NLD];
#pragma unroll
    for ( int i = 0; i < NLD; i++ )
	off[i] = win_rel(WPL * i + lane / LPW, step) + 4 * ( lane % LPW );
    float4 L[NLD];
    float X[2][16];
    tileK_fetch<K>(cur, off, 0, L, nld);
    Tw T = SPLIT ? tw_load2(tw, 0, lane) : tw_load(tw, 0, lane);
    const int row = SPLIT ? ( lane & 31 ) : lane;
    for ( int s = 0; s < NS; s++ ) {
#pragma unroll
	for ( int i = 0; i < NLD; i++ )
	    if ( i < nld )
		*reinterpret_cast<float4 *>(tile + ( WPL * i + lane / LPW ) * TR + 4 * ( lane % LPW )) = L[i];
	tileK_fetch<K>(cur, off, s + 1 < NS ? s + 1 : s, L, nld);
	{
#pragma unroll
	    for ( int j = 0; j < 4; j++ ) {
		const float4 v = *reinterpret_cast<const float4 *>(tile + row * TR + 4 * j);
		X[0][4 * j] = v.x; X[0][4 * j + 1] = v.y; X[0][4 * j + 2] = v.z; X[0][4 * j + 3] = v.w;
	    }
	}
#pragma unroll
	for ( int h = 0; h < K / 16; h++ ) {
	    const int g = s * ( K / 16 ) + h;
	    const Tw Tn = SPLIT ? tw_load2(tw, g + 1 < B / 16 ? g + 1 : g, lane) : tw_load(tw, g + 1 < B / 16 ? g + 1 : g, lane);
	    if ( h + 1 < K / 16 ) {
#pragma unroll
		for ( int j = 0; j < 4; j++ ) {
		    const float4 v = *reinterpret_cast<const float4 *>(tile + row * TR + 16 * ( h + 1 ) + 4 * j);
		    X[( h + 1 ) & 1][4 * j] = v.x; X[( h + 1 ) & 1][4 * j + 1] = v.y;
		    X[( h + 1 ) & 1][4 * j + 2] = v.z; X[( h + 1 ) & 1][4 * j + 3] = v.w;
		}
	    }
	    if ( SPLIT ) group16_2(acc, T, X[h & 1]);
	    else group16(acc, T, X[h & 1]);
	    T = Tn;
	}
    }
}

template <int K, bool SPLIT>
__global__ __launch_bounds__(64) void k_tiledK( const float *x, size_t stride, const double *tw, int nframes,
						float *out, int cstep, int fstep )
{
    extern __shared__ float slab[];
    const int lane = threadIdx.x;
    const float *xs = x + (size_t)blockIdx.x * stride;
    double sum = 0.0;
    for ( int f = 0; f < nframes; f++ ) {
	const float *cur = xs + (size_t)f * ADV + ( blockIdx.x & 31 );
	{
	    double acc[4] = { 0, 0, 0, 0 };
	    pass_tiledK<K, SPLIT>(cur, cstep, 32, tw, slab, lane, acc);
	    sum += acc[0] + acc[1] + acc[2] + acc[3];
	}
	if ( ( f + blockIdx.x ) % 9 < 4 ) {
	    double acc[4] = { 0, 0, 0, 0 };
	    pass_tiledK<K, false>(cur, fstep, 64, tw, slab, lane, acc);
	    sum += acc[0] + acc[1] + acc[2] + acc[3];
	}
    }
    out[blockIdx.x * 64 + lane] = (float)sum;
}

int main( int argc, char **argv )
{
    const int nstreams = argc > 1 ? atoi(argv[1]) : 4096;
    const int nframes = 181;
    const bool dense = argc > 2 && atoi(argv[2]) == 1;
    const int cstep = dense ? 256 : 528, fstep = dense ? 256 : 198;
    if ( dense ) {
	const int bs = 32;
	CK(hipMemcpyToSymbol(HIP_SYMBOL(g_bit_stride), &bs, sizeof bs));
	printf("dense windows (cache-resident pattern)\n");
    }
    const size_t stride = (size_t)nframes * ADV + SPAN + 64;
    float *x, *out; double *tw;
    CK(hipMalloc(&x, (size_t)nstreams * stride * 4));
    CK(hipMemset(x, 0, (size_t)nstreams * stride * 4));
    CK(hipMalloc(&out, (size_t)nstreams * 64 * 4));
    std::vector<double> htw(4 * ( B + 32 ), 0.5);
    CK(hipMalloc(&tw, htw.size() * 8));
    CK(hipMemcpy(tw, htw.data(), htw.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&]( const char *name, auto launch ) {
	launch();
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0));
	launch();
	CK(hipEventRecord(e1));
	CK(hipDeviceSynchronize());
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	printf("%-40s %8.2f ms\n", name, ms);
	fflush(stdout);
    };
    for ( size_t l : { (size_t)10240, (size_t)18448, (size_t)19968, (size_t)20480, (size_t)24576 } ) {
	int nb = 0;
	CK(hipFuncSetAttribute((const void *)k_tiledK<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
	CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_tiledK<64, false>, 64, l));
	printf("occupancy query: k_tiledK<64> with %zu bytes of LDS: %d workgroups per CU\n", l, nb);
    }
    run("arithmetic only (261 passes), with cvt", [&] { hipLaunchKernelGGL(k_compute<true>, dim3(nstreams), dim3(64), 0, 0, tw, 261, out); });
    run("arithmetic only (261 passes), cvt hoisted", [&] { hipLaunchKernelGGL(k_compute<false>, dim3(nstreams), dim3(64), 0, 0, tw, 261, out); });
    run("arithmetic + global loads (unused)", [&] { hipLaunchKernelGGL(k_mix<1>, dim3(nstreams), dim3(64), 8192, 0, x, stride, tw, nframes, out, cstep, fstep); });
    run("arithmetic + loads + tile write/read (unused)", [&] { hipLaunchKernelGGL(k_mix<2>, dim3(nstreams), dim3(64), 8192, 0, x, stride, tw, nframes, out, cstep, fstep); });
    run("arithmetic on the tile's samples", [&] { hipLaunchKernelGGL(k_mix<3>, dim3(nstreams), dim3(64), 8192, 0, x, stride, tw, nframes, out, cstep, fstep); });
    // occupancy is set through the dynamic LDS size (160 kB per CU)
    const int wpcs[] = { 16, 8, 4 };
    for ( int wpc : wpcs ) {
	const size_t lds = 160 * 1024 / wpc - 512;
	char nm[96];
	CK(hipFuncSetAttribute((const void *)k_global<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
	CK(hipFuncSetAttribute((const void *)k_global<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
	CK(hipFuncSetAttribute((const void *)k_global<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
	snprintf(nm, sizeof nm, "global, 64 B per lane per step, %d/CU", wpc);
	run(nm, [&] { hipLaunchKernelGGL(k_global<16>, dim3(nstreams), dim3(64), lds, 0, x, stride, tw, nframes, out, cstep, fstep); });
	snprintf(nm, sizeof nm, "global, 128 B per lane per step, %d/CU", wpc);
	run(nm, [&] { hipLaunchKernelGGL(k_global<32>, dim3(nstreams), dim3(64), lds, 0, x, stride, tw, nframes, out, cstep, fstep); });
	snprintf(nm, sizeof nm, "global, 256 B per lane per step, %d/CU", wpc);
	run(nm, [&] { hipLaunchKernelGGL(k_global<64>, dim3(nstreams), dim3(64), lds, 0, x, stride, tw, nframes, out, cstep, fstep); });
    }
    for ( int wpc : { 16, 8 } ) {
	const size_t lds = 160 * 1024 / wpc - 512;
	char nm[96];
	CK(hipFuncSetAttribute((const void *)k_tiledK<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
	CK(hipFuncSetAttribute((const void *)k_tiledK<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
	CK(hipFuncSetAttribute((const void *)k_tiledK<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
	snprintf(nm, sizeof nm, "tiled, 16 samples per step, %d/CU", wpc);
	run(nm, [&] { hipLaunchKernelGGL((k_tiledK<16, false>), dim3(nstreams), dim3(64), lds, 0, x, stride, tw, nframes, out, cstep, fstep); });
	snprintf(nm, sizeof nm, "tiled, 32 samples per step, %d/CU", wpc);
	run(nm, [&] { hipLaunchKernelGGL((k_tiledK<32, false>), dim3(nstreams), dim3(64), lds, 0, x, stride, tw, nframes, out, cstep, fstep); });
	if ( wpc <= 8 ) {
	    CK(hipFuncSetAttribute((const void *)k_tiledK<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
	    snprintf(nm, sizeof nm, "tiled, 64 per step, coarse passes split by tone, %d/CU", wpc);
	    run(nm, [&] { hipLaunchKernelGGL((k_tiledK<64, true>), dim3(nstreams), dim3(64), lds, 0, x, stride, tw, nframes, out, cstep, fstep); });
	    snprintf(nm, sizeof nm, "tiled, 64 samples per step, %d/CU", wpc);
	    run(nm, [&] { hipLaunchKernelGGL((k_tiledK<64, false>), dim3(nstreams), dim3(64), lds, 0, x, stride, tw, nframes, out, cstep, fstep); });
	}
    }
    for ( int wpc : wpcs ) {
	const size_t lds = 160 * 1024 / wpc - 512;
	char nm[96];
	CK(hipFuncSetAttribute((const void *)k_tiled<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
	CK(hipFuncSetAttribute((const void *)k_tiled<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
	CK(hipFuncSetAttribute((const void *)k_tiled<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
	snprintf(nm, sizeof nm, "tiled through LDS, 1 group ahead, %d/CU", wpc);
	run(nm, [&] { hipLaunchKernelGGL(k_tiled<1>, dim3(nstreams), dim3(64), lds, 0, x, stride, tw, nframes, out, cstep, fstep); });
	snprintf(nm, sizeof nm, "tiled through LDS, 2 groups ahead, %d/CU", wpc);
	run(nm, [&] { hipLaunchKernelGGL(k_tiled<2>, dim3(nstreams), dim3(64), lds, 0, x, stride, tw, nframes, out, cstep, fstep); });
	snprintf(nm, sizeof nm, "tiled through LDS, 3 groups ahead, %d/CU", wpc);
	run(nm, [&] { hipLaunchKernelGGL(k_tiled<3>, dim3(nstreams), dim3(64), lds, 0, x, stride, tw, nframes, out, cstep, fstep); });
    }
    {
	const size_t lds = (size_t)( SPAN + 32 + SPAN / B + 8 ) * 4;
	CK(hipFuncSetAttribute((const void *)k_slab<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
	CK(hipFuncSetAttribute((const void *)k_slab<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
	printf("slab: %zu bytes of LDS per wave\n", lds);
	run("LDS slab, staged 4 deep", [&] { hipLaunchKernelGGL(k_slab<4>, dim3(nstreams), dim3(64), lds, 0, x, stride, tw, nframes, out); });
	run("LDS slab, staged 10 deep", [&] { hipLaunchKernelGGL(k_slab<10>, dim3(nstreams), dim3(64), lds, 0, x, stride, tw, nframes, out); });
    }
    return 0;
}
