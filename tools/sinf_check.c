/* tools/sinf_check.c -- pins minimodem_amd/csrc/mifsk_sinf.h (the sinf the device transmitter
 * uses for --lut=0) to the C library it restates: bit-for-bit comparison with libm's sinf over
 * every `stride`-th non-negative finite float (stride 1 = all 2 139 095 040 of them, ~15 s).
 *     gcc -O2 -mfma -ffp-contract=off -I minimodem_amd/csrc -o /tmp/sinf_check tools/sinf_check.c -lm
 *     /tmp/sinf_check [stride]           exit status 0 iff no value differs                    */
#include <stdio.h>
#include <stdlib.h>
#include "mifsk_sinf.h"

int main( int argc, char **argv )
{
    const unsigned long stride = argc > 1 ? strtoul(argv[1], NULL, 0) : 1;
    unsigned long bad = 0, cnt = 0;
    for ( unsigned long u = 0; u <= 0x7f7fffffUL; u += stride ) {
	const uint32_t b = (uint32_t)u;
	float f;
	memcpy(&f, &b, 4);
	const float mine = mifsk_glibc_sinf(f), lib = sinf(f);
	if ( mifsk_f32_bits(mine) != mifsk_f32_bits(lib) ) {
	    if ( bad < 10 )
		printf("x=%a  restated=%a  libm=%a\n", f, mine, lib);
	    bad++;
	}
	cnt++;
    }
    printf("%lu values compared, %lu differ\n", cnt, bad);
    return bad != 0;
}
