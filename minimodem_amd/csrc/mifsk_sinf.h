/*
 * mifsk_sinf.h -- glibc's sinf(), restated, for the device transmitter's --lut=0
 * mode (reference src/simple-tone-generator.c:134,155: `mag * sinf(radians)` per
 * sample; tests 07, 11 and 13 transmit that way).
 *
 * The reference calls the C library; bit-identical output needs the library's
 * algorithm.  glibc >= 2.28 (this image: 2.35) computes sinf in double precision
 * with the ARM "optimized routines" scheme (sysdeps/ieee754/flt-32/s_sinf.c,
 * sincosf.h, s_sincosf_data.c; the source is not in this image): |x| < pi/4
 * polynomial; |x| < 120 reduction by n = round(x * 2/pi) with hpi_inv scaled by
 * 2^24; larger arguments by a 192-bit fixed-point product with 4/pi; then a
 * degree-7 sine or degree-8 cosine polynomial in double, rounded to float.  On
 * x86-64 CPUs with FMA the library dispatches to its FMA build, in which every
 * a * b + c below is one fused operation -- written out here as explicit fma()
 * so that host compilers (-ffp-contract=off) and hipcc produce the same
 * sequence.  PINNED: tools/sinf_check.c compares this file against the running
 * C library on ALL 2 139 095 040 non-negative finite floats (0 mismatches here;
 * without the fusions 6 in the first 1.2e9 differ), tests/test_sinf.py runs a
 * strided sweep of it in the CPU suite, and the device output for --lut=0 is
 * compared with the reference's own WAV samples (goldens t07, t11, t13) and
 * with the host transmitter (which calls libm) in tests/test_gpu_txdev.py.
 * Negative arguments never occur in the tone generator and are not handled.
 */
#ifndef MIFSK_SINF_H
#define MIFSK_SINF_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
#define MIFSK_HD __host__ __device__ __forceinline__
#else
#define MIFSK_HD static inline
#endif

MIFSK_HD uint32_t mifsk_f32_bits( float f )
{
    uint32_t u;
    memcpy(&u, &f, sizeof(u));
    return u;
}

/* sinf_poly (sincosf.h): x2 = x * x; n odd -> cosine polynomial.  `neg` selects
 * the table entry with negated cosine coefficients (quadrants 2 and 3). */
MIFSK_HD float mifsk_sinf_poly( double x, double x2, int neg, int n )
{
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double sg = neg ? -1.0 : 1.0;
    const double c0 = sg * 0x1p0, c1 = sg * -0x1.ffffffd0c621cp-2, c2 = sg * 0x1.55553e1068f19p-5,
		 c3 = sg * -0x1.6c087e89a359dp-10, c4 = sg * 0x1.99343027bf8c3p-16;
    if ( ( n & 1 ) == 0 ) {
	const double x3 = x * x2;
	const double t1 = fma(x2, s3, s2);
	const double x7 = x3 * x2;
	const double s = fma(x3, s1, x);
	return (float)fma(x7, t1, s);
    } else {
	const double x4 = x2 * x2;
	const double t2 = fma(x2, c4, c3);
	const double t1 = fma(x2, c1, c0);
	const double x6 = x4 * x2;
	const double c = fma(x4, c2, t1);
	return (float)fma(x6, t2, c);
    }
}

/* sinf for y >= 0 (finite) */
MIFSK_HD float mifsk_glibc_sinf( float y )
{
    const uint32_t top = ( mifsk_f32_bits(y) >> 20 ) & 0x7ffu;		/* abstop12 */
    double x = (double)y;
    const double sign[4] = { 1.0, -1.0, -1.0, 1.0 };
    if ( top < 0x3f4u ) {						/* |y| < pi/4 */
	if ( top < 0x398u )						/* |y| < 2^-12 */
	    return y;
	return mifsk_sinf_poly(x, x * x, 0, 0);
    }
    int n;
    if ( top < 0x42fu ) {						/* |y| < 120: reduce_fast */
	const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
	const double r = x * hpi_inv;
	n = ( (int32_t)r + 0x800000 ) >> 24;
	x = fma(-(double)n, hpi, x);
    } else {								/* reduce_large */
	const uint32_t inv_pio4[24] = {
	    0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
	    0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd,
	    0xf534ddc0, 0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43,
	    0x993c4390, 0x3c439041 };				/* the bits of 4/pi */
	uint32_t xi = mifsk_f32_bits(y);
	const uint32_t *arr = &inv_pio4[( xi >> 26 ) & 15u];
	const int shift = (int)( ( xi >> 23 ) & 7u );
	xi = ( xi & 0xffffffu ) | 0x800000u;
	xi <<= shift;
	uint64_t res0 = (uint64_t)( xi * arr[0] );			/* 32-bit product, as there */
	const uint64_t res1 = (uint64_t)xi * arr[4];
	const uint64_t res2 = (uint64_t)xi * arr[8];
	res0 = ( res2 >> 32 ) | ( res0 << 32 );
	res0 += res1;
	const uint64_t nn = ( res0 + ( 1ULL << 61 ) ) >> 62;
	res0 -= nn << 62;
	n = (int)nn;
	x = (double)(int64_t)res0 * 0x1.921FB54442D18p-62;
    }
    const double s = sign[n & 3];
    return mifsk_sinf_poly(x * s, x * x, ( n & 2 ) != 0, n);
}

#endif /* MIFSK_SINF_H */
