/* tools/hypotf_check.c -- pins the identity band_mag() relies on (minimodem_amd/csrc/mifsk_devmath.h):
 * the C library's hypotf(x, y) -- what the reference's band_mag calls, src/fsk.c:107-114 -- is
 * (float)sqrt((double)x * x + (double)y * y) bit for bit (the squares of floats are exact in
 * double, their sum is rounded once, sqrt is correctly rounded, then one more rounding to float)
 * -- with C's one special rule, hypot(+-inf, NaN) = +inf, which band_mag() restates too.
 * The pair space is 2^62, so this is a sweep, not a proof: `npairs` pseudo-random finite pairs
 * (every exponent combination equally likely), the same number of pairs of nearby magnitude
 * (where the sum's rounding matters), and a grid of special values against each other.
 *     gcc -O2 -ffp-contract=off -o /tmp/hypotf_check tools/hypotf_check.c -lm
 *     /tmp/hypotf_check [log2 npairs = 24]     exit status 0 iff no pair differs             */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint32_t bits_of( float f ) { uint32_t b; memcpy(&b, &f, 4); return b; }
static float float_of( uint32_t b ) { float f; memcpy(&f, &b, 4); return f; }

static float restated( float x, float y )
{
    const double s = (double)x * (double)x + (double)y * (double)y;
    if ( s != s && ( isinf(x) || isinf(y) ) )		/* C11 F.10.4.3: an infinity beats a NaN */
	return INFINITY;
    return (float)sqrt(s);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rng( void )
{
    uint64_t z = ( rng_state += 0x9E3779B97F4A7C15ull );
    z = ( z ^ ( z >> 30 ) ) * 0xBF58476D1CE4E5B9ull;
    z = ( z ^ ( z >> 27 ) ) * 0x94D049BB133111EBull;
    return z ^ ( z >> 31 );
}

static unsigned long bad = 0, cnt = 0;
static void check( float x, float y )
{
    const float lib = hypotf(x, y), mine = restated(x, y);
    cnt++;
    /* (a NaN is a NaN: payloads are not part of the contract) */
    if ( bits_of(lib) != bits_of(mine) && !( lib != lib && mine != mine ) ) {
	if ( bad < 10 )
	    printf("x=%a y=%a  hypotf=%a  restated=%a\n", x, y, lib, mine);
	bad++;
    }
}

int main( int argc, char **argv )
{
    const unsigned lg = argc > 1 ? (unsigned)strtoul(argv[1], NULL, 0) : 24;
    const unsigned long n = 1ul << ( lg > 34 ? 34 : lg );
    /* uniformly random bit patterns of finite floats, signs included */
    for ( unsigned long i = 0; i < n; i++ ) {
	const uint64_t r = rng();
	uint32_t a = (uint32_t)r, b = (uint32_t)( r >> 32 );
	if ( ( a & 0x7f800000u ) == 0x7f800000u ) a ^= 0x00800000u;
	if ( ( b & 0x7f800000u ) == 0x7f800000u ) b ^= 0x00800000u;
	check(float_of(a), float_of(b));
    }
    /* pairs within a factor of 2^4 of each other: both squares contribute to the rounded sum */
    for ( unsigned long i = 0; i < n; i++ ) {
	const uint64_t r = rng();
	uint32_t a = (uint32_t)r & 0x7fffffffu;
	if ( ( a & 0x7f800000u ) == 0x7f800000u ) a ^= 0x00800000u;
	const int ea = (int)( a >> 23 ), de = (int)( ( r >> 32 ) & 7 ) - 4;
	int eb = ea + de;
	if ( eb < 0 ) eb = 0;
	if ( eb > 254 ) eb = 254;
	const uint32_t b = ( (uint32_t)eb << 23 ) | ( (uint32_t)( r >> 40 ) & 0x007fffffu );
	check(float_of(a), float_of(b));
    }
    /* special values against each other and against a sweep */
    {
	static const uint32_t sp[] = { 0x00000000u, 0x80000000u, 0x00000001u, 0x007fffffu, 0x00800000u,
	    0x00800001u, 0x3f800000u, 0x3f7fffffu, 0x3f800001u, 0x40400000u, 0x40800000u, 0x5f000000u,
	    0x5f3504f3u, 0x7f000000u, 0x7f7fffffu, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0x1f800000u,
	    0x1fb504f3u, 0x20000000u, 0x33800000u, 0x34000000u };
	const unsigned ns = sizeof(sp) / sizeof(sp[0]);
	for ( unsigned i = 0; i < ns; i++ ) {
	    for ( unsigned j = 0; j < ns; j++ )
		check(float_of(sp[i]), float_of(sp[j]));
	    for ( uint64_t u = 0; u <= 0x7f800000ull; u += 4099 )
		check(float_of(sp[i]), float_of((uint32_t)u));
	}
    }
    printf("%lu pairs compared, %lu differ\n", cnt, bad);
    return bad != 0;
}
