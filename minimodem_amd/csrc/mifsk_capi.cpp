// mifsk_capi.cpp -- the C ABI of libmifsk.so (include/fsk.h, include/mifsk.h).
//
// Host-side glue only: contexts, twiddle tables, argument marshalling, and the
// reference-compatible five-function API layered over the same kernels as the
// batch entry points.  There is no CPU implementation of the signal path in
// this library: every entry point that produces a result launches a HIP kernel,
// and context creation fails with -ENODEV when no gfx950 device is available.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "fsk.h"
#include "mifsk.h"
#include "mifsk_device.h"
#include "mifsk_ctx.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace mifsk {

// The shared-segment plan of one zig-zag scan (SegPlan in mifsk_device.h): cut the span
// the scan's windows cover at every window edge, drop pieces no window covers (bit
// offsets are rounded, consecutive windows may leave a sample between them), split the
// longest pieces until the lanes of ceil(n / 64) passes are full, hand the pieces to
// the passes longest first.
// candidates of one zig-zag scan in scan order (fsk.c:477-484), appended to `out`
static void zigzag_candidates( std::vector<unsigned> &out, unsigned first, unsigned mx, unsigned step )
{
    if ( (int)first >= (int)mx || step == 0 )
	return;
    const unsigned U = ( mx - first - 1 ) / step + 1;
    const unsigned D = U - 1 < first / step ? U - 1 : first / step;
    for ( unsigned i = 0; i < U + D; i++ ) {
	if ( i == 0 ) out.push_back(first);
	else if ( i <= 2 * D ) out.push_back(( i & 1u ) ? first + ( ( i + 1 ) / 2 ) * step : first - ( ( i + 1 ) / 2 ) * step);
	else out.push_back(first + ( i - D ) * step);
    }
}

// `cand`: the candidates whose windows the plan covers, window w = candidate w / n_bits, bit w % n_bits
static void plan_segments( SegPlan &sp, const mifsk_rx_config &c, const std::vector<unsigned> &cand )
{
    std::memset(&sp, 0, sizeof(sp));
    const unsigned nb = c.expect_n_bits, B = c.bit_nsamples;
    const unsigned J = (unsigned)cand.size();
    if ( J == 0 || nb == 0 || J * nb > (unsigned)SEGW_MAX )
	return;
    auto at = [&]( unsigned i ) -> unsigned { return cand[i]; };
    std::vector<unsigned> wstart(J * nb);
    std::vector<unsigned> cuts;
    for ( unsigned j = 0; j < J; j++ )
	for ( unsigned k = 0; k < nb; k++ ) {
	    const unsigned a = at(j) + c.bit_offset[k];
	    wstart[j * nb + k] = a;
	    cuts.push_back(a);
	    cuts.push_back(a + B);
	}
    std::sort(cuts.begin(), cuts.end());
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    struct Seg { unsigned rel, len; };
    std::vector<Seg> segs;
    for ( size_t i = 0; i + 1 < cuts.size(); i++ ) {
	const unsigned lo = cuts[i], hi = cuts[i + 1];
	bool covered = false;
	for ( unsigned a : wstart )
	    covered = covered || ( a <= lo && hi <= a + B );
	if ( covered )
	    segs.push_back(Seg{lo, hi - lo});
    }
    if ( segs.empty() || segs.size() > (size_t)SEG_MAX )
	return;
    // Balance.  Every piece may be cut further; the parts go to (at most two) passes of 64 lanes,
    // longest first.  What a pass costs is decided by its longest part: whole groups of 16 samples
    // (a group of the sums: ~86 instructions, ~118 where some lane's part ends inside it and the
    // samples are masked) in whole tile steps of 32 (stage, read back, fetch: ~60).  Two ways of
    // cutting are tried and the cheapest plan is taken:
    //  * equal parts: with a target length T piece i gets ceil(len_i / T) parts (all T);
    //  * caps (round 6): pass 0 takes parts of at most A0 samples, pass 1 of at most A1 <= A0, and
    //    a piece is cut UNEQUALLY into n0 parts for the one and n1 for the other -- which (n0, n1)
    //    per piece is a small dynamic program over the 64 lanes of each pass.  RTTY's carrier-held
    //    plan (pieces of 165, 110, 66, 55, 44 samples) went from 83 + 82 | 110 whole -- 7 + 6 groups
    //    in 4 + 3 steps -- to 101 + 64 | 110 whole: 7 + 4 groups in 4 + 2 steps.
    {
	const std::vector<Seg> pieces = segs;
	// (the assembly: ~16 instructions per segment of the longest window, once per 64 windows --
	// per 32 where two lanes share a window, Wave::seg_correlate)
	const unsigned asm_units = J * nb <= 32u ? 1u : 2u * ( ( J * nb + 63u ) / 64u );
	auto cost_of = [&]( const std::vector<std::vector<unsigned>> &parts ) -> unsigned {
	    std::vector<unsigned> lens, at;
	    for ( size_t i = 0; i < parts.size(); i++ ) {
		unsigned a = pieces[i].rel;
		for ( unsigned l : parts[i] ) {
		    lens.push_back(l);
		    at.push_back(a);
		    a += l;
		}
	    }
	    if ( lens.empty() || lens.size() > (size_t)SEG_MAX )
		return 0xFFFFFFFFu;
	    unsigned cmax = 0;
	    for ( unsigned a : wstart ) {
		unsigned n = 0;
		for ( size_t i = 0; i < lens.size(); i++ )
		    n += ( at[i] >= a && at[i] + lens[i] <= a + B ) ? 1u : 0u;
		cmax = std::max(cmax, n);
	    }
	    std::sort(lens.begin(), lens.end(), [](unsigned x, unsigned y) { return x > y; });
	    unsigned cost = 8u * cmax * asm_units;
	    for ( size_t p0 = 0; p0 < lens.size(); p0 += 64 ) {
		const size_t p1 = std::min(lens.size(), p0 + 64);
		const unsigned lmax = lens[p0], lmin = lens[p1 - 1];
		const unsigned g = ( lmax + 15 ) / 16, st = ( g + 1 ) / 2, full = lmin / 16;
		cost += 86u * g + 60u * st + 32u * ( g - std::min(g, full) );
	    }
	    return cost;
	};
	unsigned best_cost = 0xFFFFFFFFu;
	std::vector<std::vector<unsigned>> best;		// the parts of every piece, in position order
	// equal parts
	{
	    std::vector<unsigned> cand;
	    for ( const Seg &s : pieces )
		for ( unsigned k = 1; k <= 16u && s.len / k >= 16u; k++ )
		    cand.push_back(( s.len + k - 1 ) / k);
	    std::sort(cand.begin(), cand.end());
	    cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
	    for ( unsigned T : cand ) {
		std::vector<std::vector<unsigned>> parts;
		size_t nparts = 0;
		for ( const Seg &s : pieces ) {
		    const unsigned k = ( s.len + T - 1 ) / T;
		    parts.emplace_back();
		    for ( unsigned q = 0; q < k; q++ )
			parts.back().push_back(s.len / k + ( q < s.len % k ? 1u : 0u ));
		    nparts += k;
		}
		if ( nparts > (size_t)SEG_MAX )
		    continue;
		const unsigned c = cost_of(parts);
		if ( c < best_cost ) {
		    best_cost = c;
		    best = parts;
		}
	    }
	}
	// caps
	{
	    unsigned lnat = 0, ltot = 0;
	    for ( const Seg &s : pieces ) {
		lnat = std::max(lnat, s.len);
		ltot += s.len;
	    }
	    const unsigned amax = ( lnat + 15u ) & ~15u;
	    constexpr unsigned INF = 0xFFFFu;
	    std::vector<uint16_t> choice(pieces.size() * 65u);
	    std::vector<unsigned> dp(65), nx(65);
	    std::vector<std::vector<unsigned>> parts(pieces.size());
	    // (caps in whole groups; for very long windows in coarser steps: at most 32 values)
	    const unsigned astep = std::max(16u, ( amax / 32u + 15u ) & ~15u);
	    for ( unsigned A0 = astep; A0 <= amax + astep - 1u; A0 += astep )
		for ( unsigned A1 = astep; A1 <= A0; A1 += astep ) {
		    if ( 64u * ( A0 + A1 ) < ltot )
			continue;			// (the lanes of two passes cannot hold the span)
		    // (a plan whose longest parts are a whole group below its caps is found under the
		    // smaller caps: these caps cost at least their own groups and steps)
		    if ( 86u * ( A0 / 16u + A1 / 16u ) + 60u * ( ( A0 + 16u ) / 32u + ( A1 + 16u ) / 32u ) >= best_cost )
			continue;
		    // dp[j]: fewest pass-1 parts with j pass-0 parts over the pieces so far
		    std::fill(dp.begin(), dp.end(), INF);
		    dp[0] = 0;
		    for ( size_t i = 0; i < pieces.size(); i++ ) {
			const unsigned L = pieces[i].len;
			std::fill(nx.begin(), nx.end(), INF);
			for ( unsigned j = 0; j <= 64; j++ ) {
			    if ( dp[j] == INF )
				continue;
			    for ( unsigned n0 = 0; n0 <= ( L + A0 - 1 ) / A0 && j + n0 <= 64; n0++ ) {
				const unsigned rest = L > n0 * A0 ? L - n0 * A0 : 0u;
				const unsigned n1 = ( rest + A1 - 1 ) / A1;
				if ( n0 + n1 == 0 || n0 + n1 > L )
				    continue;
				if ( dp[j] + n1 < nx[j + n0] ) {
				    nx[j + n0] = dp[j] + n1;
				    choice[i * 65u + j + n0] = (uint16_t)n0;
				}
			    }
			}
			dp.swap(nx);
		    }
		    unsigned jbest = 65;
		    for ( unsigned j = 0; j <= 64; j++ )
			if ( dp[j] <= 64 && ( jbest == 65 || dp[j] + j < dp[jbest] + jbest ) )
			    jbest = j;
		    if ( jbest == 65 )
			continue;
		    // walk back: n0 of every piece; its n1 follows
		    unsigned j = jbest;
		    for ( size_t i = pieces.size(); i-- > 0; ) {
			const unsigned L = pieces[i].len, n0 = choice[i * 65u + j];
			const unsigned rest = L > n0 * A0 ? L - n0 * A0 : 0u;
			const unsigned n1 = ( rest + A1 - 1 ) / A1;
			// pass 1's parts as long as they may be, pass 0's share the rest equally
			unsigned t1 = n1 ? std::min(n1 * A1, L - n0) : 0u;
			if ( n0 == 0 )
			    t1 = L;
			const unsigned t0 = L - t1;
			parts[i].clear();
			for ( unsigned q = 0; q < n0; q++ )
			    parts[i].push_back(t0 / n0 + ( q < t0 % n0 ? 1u : 0u ));
			for ( unsigned q = 0; q < n1; q++ )
			    parts[i].push_back(t1 / n1 + ( q < t1 % n1 ? 1u : 0u ));
			j -= n0;
		    }
		    bool sound = true;
		    for ( const std::vector<unsigned> &pp : parts )
			for ( unsigned l : pp )
			    sound = sound && l >= 1u;
		    const unsigned c = sound ? cost_of(parts) : 0xFFFFFFFFu;
		    if ( c < best_cost ) {
			best_cost = c;
			best = parts;
		    }
		}
	}
	if ( best.empty() )
	    return;
	segs.clear();
	for ( size_t i = 0; i < pieces.size(); i++ ) {
	    unsigned at = pieces[i].rel, total = 0;
	    for ( unsigned l : best[i] ) {
		segs.push_back(Seg{at, l});
		at += l;
		total += l;
	    }
	    if ( total != pieces[i].len )
		return;				// (cannot happen; leaves valid = 0)
	}
	if ( segs.size() > (size_t)SEG_MAX )
	    return;
    }
    const unsigned npass = (unsigned)( ( segs.size() + 63 ) / 64 );
    // passes: longest pieces first, position order inside a pass
    std::vector<unsigned> order(segs.size());
    for ( size_t i = 0; i < order.size(); i++ ) order[i] = (unsigned)i;
    std::stable_sort(order.begin(), order.end(), [&]( unsigned a, unsigned b ) { return segs[a].len > segs[b].len; });
    sp.nseg = (unsigned)segs.size();
    sp.npass = npass;
    sp.nwin = J * nb;
    for ( unsigned i = 0; i < (unsigned)SEG_MAX; i++ )
	sp.slot_seg[i] = 0xFFFFu;
    unsigned lmax = 0;
    for ( unsigned pss = 0; pss < npass; pss++ ) {
	std::vector<unsigned> mine(order.begin() + 64 * pss,
				   order.begin() + (long)std::min<size_t>(order.size(), 64 * ( pss + 1 )));
	std::sort(mine.begin(), mine.end());
	sp.pass_len[pss] = 0;
	sp.pass_min[pss] = 0xFFFFFFFFu;
	for ( size_t l = 0; l < mine.size(); l++ ) {
	    sp.slot_seg[64 * pss + l] = (uint16_t)mine[l];
	    sp.pass_len[pss] = std::max(sp.pass_len[pss], segs[mine[l]].len);
	    sp.pass_min[pss] = std::min(sp.pass_min[pss], segs[mine[l]].len);
	}
	lmax = std::max(lmax, sp.pass_len[pss]);
    }
    for ( size_t i = 0; i < segs.size(); i++ ) {
	sp.seg_rel[i] = segs[i].rel;
	sp.seg_len[i] = (uint16_t)segs[i].len;
	sp.span_hi = std::max(sp.span_hi, segs[i].rel + segs[i].len);
    }
    unsigned cmax = 0;
    for ( unsigned w = 0; w < sp.nwin; w++ ) {
	unsigned f = 0, n = 0;
	bool in = false;
	for ( unsigned i = 0; i < sp.nseg; i++ ) {
	    const bool inside = segs[i].rel >= wstart[w] && segs[i].rel + segs[i].len <= wstart[w] + B;
	    if ( inside && !in ) { f = i; in = true; }
	    if ( inside ) n++;
	}
	sp.win_first[w] = (uint16_t)f;
	sp.win_count[w] = (uint16_t)n;
	cmax = std::max(cmax, n);
	// (its pieces are consecutive and tile it but for the uncovered samples, which no
	// window contains: those lie between windows, never inside one)
	unsigned total = 0;
	for ( unsigned i = f; i < f + n; i++ ) total += segs[i].len;
	if ( total != B )
	    return;				// (cannot happen; leaves valid = 0)
    }
    // DESIGN.md "shared segments": index-order rounding (B - 1) + segment sums
    // sqrt(2) (L - 1) + assembly 2 n + table entries' own rounding 85, in units of
    // 2^-53 * sum |x|; rounded up generously
    // (+ 2: a short scan's windows are assembled as two half sums and one more addition)
    sp.bound_c = (float)( B + 1.5 * lmax + 2.0 * cmax + 2.0 + 128.0 );
    // (a pass loads table group ceil(L / 16) + 3 at most; the table has ceil(B / 16) + 1)
    if ( ( lmax + 15 ) / 16 + 3 > ( B + 15 ) / 16 )
	return;
    // packed copies; a plan whose numbers do not fit the fields is not used
    if ( lmax >= 4096u || sp.span_hi >= ( 1u << 20 ) || sp.nseg > 255u )
	return;
    for ( unsigned i = 0; i < (unsigned)SEG_MAX; i++ ) {
	const unsigned s = sp.slot_seg[i];
	sp.p_slot_seg[i] = s == 0xFFFFu ? 0xFFu : (uint8_t)s;
	sp.p_slot[i] = s == 0xFFFFu ? 0u : ( sp.seg_rel[s] | ( (uint32_t)sp.seg_len[s] << 20 ) );
    }
    for ( unsigned w = 0; w < sp.nwin; w++ ) {
	if ( wstart[w] >= 65536u || sp.win_count[w] > 255u )
	    return;
	sp.p_win[w] = sp.win_first[w] | ( (uint32_t)sp.win_count[w] << 8 ) | ( wstart[w] << 16 );
    }
    sp.valid = 1;
}

void fill_devcfg( DevCfg &d, const mifsk_rx_config &c )
{
    std::memset(&d, 0, sizeof(d));
    d.n_bits = c.expect_n_bits;
    d.bit_nsamples = c.bit_nsamples;
    d.last_reach = c.bit_offset[c.expect_n_bits - 1] + c.bit_nsamples;
    d.magscalar = 2.0f / (float)c.bit_nsamples;		// fsk.c:132
    d.frame_nsamples = c.frame_nsamples;
    d.expect_nsamples = c.expect_nsamples;
    d.overscan = c.nsamples_overscan;
    for ( int i = 0; i < 2; i++ ) {
	d.try_first[i] = c.try_first[i];
	d.try_max[i] = c.try_max[i];
	d.try_step[i] = c.try_step[i];
	d.try_step_fine[i] = c.try_step_fine[i];
    }
    d.conf_threshold = c.confidence_threshold;
    d.search_limit = c.search_limit;
    d.n_data_bits = c.n_data_bits;
    d.nstartbits = (uint32_t)c.nstartbits;
    d.has_stopbits = c.nstopbits != 0.0f ? 1u : 0u;
    d.msb_first = c.msb_first ? 1u : 0u;
    d.do_rx_sync = c.do_rx_sync ? 1u : 0u;
    d.rx_one = c.rx_one ? 1u : 0u;
    d.sync_byte = c.sync_byte;
    d.b_mark = c.b_mark;
    d.b_space = c.b_space;
    d.fftsize = (uint32_t)c.fftsize;

    // One pad word per bit row of a SCAN slab where that spreads the lanes of a
    // search over more LDS banks: the first 64 windows of the carrier-held fine
    // search (lane = candidate * n_bits + bit, ds_read_b32, 32 lanes per LDS
    // cycle, bank = word mod 32) are laid out both ways and the cheaper pitch
    // wins; unpadded rows on a tie (no per-sample row test in the correlator).
    {
	auto cost = [&]( unsigned skew ) -> unsigned {
	    const unsigned f = c.try_first[1], mx = c.try_max[1], st = c.try_step_fine[1];
	    const unsigned nb = c.expect_n_bits ? c.expect_n_bits : 1u, B = c.bit_nsamples ? c.bit_nsamples : 1u;
	    unsigned U = 0, D = 0;
	    if ( f < mx && st ) {
		U = ( mx - f - 1 ) / st + 1;
		D = U - 1 < f / st ? U - 1 : f / st;
	    }
	    const unsigned J = U + D;
	    unsigned word[64], n = 0;
	    for ( unsigned i = 0; i < J && n < 64; i++ ) {
		unsigned t = f;
		if ( i && i <= 2 * D )
		    t = ( i & 1u ) ? f + ( ( i + 1 ) / 2 ) * st : f - ( ( i + 1 ) / 2 ) * st;
		else if ( i )
		    t = f + ( i - D ) * st;
		for ( unsigned k = 0; k < nb && n < 64; k++ ) {
		    const unsigned rel = t + c.bit_offset[k];
		    word[n++] = rel + ( rel / B ) * skew;
		}
	    }
	    unsigned total = 0;
	    for ( unsigned h = 0; h < n; h += 32 ) {
		unsigned worst = 0;
		for ( unsigned bank = 0; bank < 32; bank++ ) {
		    unsigned distinct = 0;
		    for ( unsigned i = h; i < n && i < h + 32; i++ ) {
			if ( word[i] % 32 != bank )
			    continue;
			bool seen = false;
			for ( unsigned j = h; j < i; j++ )
			    seen = seen || word[j] == word[i];
			distinct += seen ? 0u : 1u;
		    }
		    worst = distinct > worst ? distinct : worst;
		}
		total += worst;
	    }
	    return total;
	};
	d.skew = cost(1) < cost(0) ? 1u : 0u;
    }
    for ( int i = 0; i < 4; i++ ) {
	const unsigned f = c.try_first[i & 1], mx = c.try_max[i & 1];
	const unsigned st = ( i & 2 ) ? c.try_step_fine[i & 1] : c.try_step[i & 1];
	if ( (int)f < (int)mx && st ) {
	    d.zz_up[i] = ( mx - f - 1 ) / st + 1;
	    d.zz_down[i] = d.zz_up[i] - 1 < f / st ? d.zz_up[i] - 1 : f / st;
	}
    }
    // long windows (what the wavefront engine reads through its LDS tile): the scans share
    // their segments' partial sums
    if ( c.bit_nsamples >= 256u && c.bit_nsamples <= 65535u )
	for ( int i = 0; i < 4; i++ ) {
	    std::vector<unsigned> cand;
	    zigzag_candidates(cand, c.try_first[i & 1], c.try_max[i & 1],
			      ( i & 2 ) ? c.try_step_fine[i & 1] : c.try_step[i & 1]);
	    plan_segments(d.seg[i], c, cand);
	}
    if ( d.seg[1].valid && d.seg[3].valid ) {
	std::vector<unsigned> cand;
	zigzag_candidates(cand, c.try_first[1], c.try_max[1], c.try_step[1]);
	d.seg_union_first_fine = (uint32_t)cand.size() * c.expect_n_bits;
	zigzag_candidates(cand, c.try_first[1], c.try_max[1], c.try_step_fine[1]);
	plan_segments(d.seg[4], c, cand);
    }
    d.div_magic = c.bit_nsamples > 1 ? (uint32_t)( 0x100000000ULL / c.bit_nsamples ) : 0xFFFFFFFFu;
    // minimodem.c:1407 with frame_start == try_first (carrier)
    d.lock_advance = c.try_first[1] + c.frame_nsamples - c.nsamples_overscan;
    d.la_magic = d.lock_advance > 1 ? (uint32_t)( 0x100000000ULL / d.lock_advance ) : 0xFFFFFFFFu;
    d.nbits_magic = c.expect_n_bits > 1 ? (uint32_t)( 0x100000000ULL / c.expect_n_bits ) : 0xFFFFFFFFu;
    {
	// lowest candidate of the carrier coarse scan relative to its first try
	// (fsk.c:477-484), rounded up to whole bit lengths
	const unsigned f = c.try_first[1], mx = c.try_max[1], st = c.try_step[1];
	unsigned down = 0;
	if ( f < mx && st ) {
	    const unsigned U = ( mx - f - 1 ) / st + 1;
	    const unsigned D = U - 1 < f / st ? U - 1 : f / st;
	    down = D * st;
	}
	d.lock_back = ( down + c.bit_nsamples - 1 ) / c.bit_nsamples * c.bit_nsamples;
    }
    // every lattice window starts a multiple of 4 samples after the first one
    // when the bit length, all bit offsets and the frame step are multiples of
    // 4: then an unskewed region read with 16-byte LDS loads is conflict-light
    d.lat_linear = ( c.bit_nsamples % 4 == 0 && d.lock_advance % 4 == 0 ) ? 1u : 0u;
    for ( unsigned k = 0; k < c.expect_n_bits; k++ )
	if ( c.bit_offset[k] % 4 != 0 )
	    d.lat_linear = 0;
    d.lat_grid = ( d.lat_linear && c.expect_n_bits >= 2
		   && d.lock_advance == ( c.expect_n_bits - 1 ) * c.bit_nsamples ) ? 1u : 0u;
    for ( unsigned k = 0; k < c.expect_n_bits; k++ )
	if ( c.bit_offset[k] != k * c.bit_nsamples )
	    d.lat_grid = 0;
    for ( unsigned k = 0; k < c.expect_n_bits; k++ ) {
	d.bit_offset[k] = c.bit_offset[k];
	for ( int s = 0; s < 2; s++ ) {
	    const char ch = ( s ? c.expect_sync : c.expect_data )[k];
	    if ( ch != 'd' ) {
		d.req_mask[s] |= 1ULL << k;
		if ( ch == '1' )
		    d.req_val[s] |= 1ULL << k;
	    }
	}
    }
}

} // namespace mifsk

using mifsk::DevCfg;

static_assert(sizeof(mifsk_stream_state) == 96, "mifsk_stream_state is part of the ABI");



// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------

int mifsk::ctx_device( const mifsk_ctx *ctx ) { return ctx->device; }

extern "C" int mifsk_abi_version( void ) { return MIFSK_ABI_VERSION; }

extern "C" size_t mifsk_abi_sizeof( const char *name )
{
    if ( !name )
	return 0;
#define MIFSK_SIZEOF(T) if ( std::strcmp(name, #T) == 0 ) return sizeof(T)
    MIFSK_SIZEOF(mifsk_modem_args);
    MIFSK_SIZEOF(mifsk_rx_config);
    MIFSK_SIZEOF(mifsk_search);
    MIFSK_SIZEOF(mifsk_search_result);
    MIFSK_SIZEOF(mifsk_frame);
    MIFSK_SIZEOF(mifsk_episode);
    MIFSK_SIZEOF(mifsk_demod_io);
    MIFSK_SIZEOF(mifsk_launch_info);
    MIFSK_SIZEOF(mifsk_pipeline_info);
    MIFSK_SIZEOF(mifsk_gather_info);
    MIFSK_SIZEOF(mifsk_session_result);
    MIFSK_SIZEOF(mifsk_scan_plan);
    MIFSK_SIZEOF(mifsk_host_stats);
    MIFSK_SIZEOF(mifsk_stream_state);
    MIFSK_SIZEOF(mifsk_wav_info);
    MIFSK_SIZEOF(mifsk_file_result);
    MIFSK_SIZEOF(fsk_plan);
#undef MIFSK_SIZEOF
    return 0;
}

extern "C" int mifsk_selftest_sqrt( mifsk_ctx *ctx, uint64_t seed, uint64_t nvalues, uint64_t counts[4] )
{
    if ( !ctx || !counts )
	return -EINVAL;
    HIP_OK(hipSetDevice(ctx->device));
    unsigned long long *d = nullptr;
    HIP_OK(hipMalloc(&d, 4 * sizeof(unsigned long long)));
    int rc = hipMemset(d, 0, 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -EIO;
    const uint32_t per_thread = 4096;
    uint64_t blocks = ( nvalues + 256ull * per_thread - 1 ) / ( 256ull * per_thread );
    if ( blocks < 1 ) blocks = 1;
    if ( blocks > 0x7FFFFFFFull ) blocks = 0x7FFFFFFFull;
    if ( rc == 0 )
	rc = mifsk::launch_selftest_sqrt(seed, (uint32_t)blocks, per_thread, d, nullptr);
    unsigned long long h[4] = { 0, 0, 0, 0 };
    if ( rc == 0 && hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess )
	rc = -EIO;
    (void)hipFree(d);
    for ( int i = 0; i < 4; i++ )
	counts[i] = h[i];
    return rc;
}

extern "C" int mifsk_ctx_create( mifsk_ctx **out, int device )
{
    if ( !out )
	return -EINVAL;
    *out = nullptr;
    int ndev = 0;
    if ( hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 ) {
	fprintf(stderr, "mifsk: no HIP device available (this library has no CPU path)\n");
	return -ENODEV;
    }
    if ( device >= 0 ) {
	if ( device >= ndev )
	    return -ENODEV;
	HIP_OK(hipSetDevice(device));
    } else {
	HIP_OK(hipGetDevice(&device));
    }
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, device));
    if ( std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 ) {
	fprintf(stderr, "mifsk: device %d is %s; the kernels are built for gfx950 only\n",
		device, prop.gcnArchName);
	return -ENODEV;
    }
    mifsk_ctx *ctx = new (std::nothrow) mifsk_ctx();
    if ( !ctx )
	return -ENOMEM;
    ctx->device = device;
    ctx->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    std::snprintf(ctx->name, sizeof(ctx->name), "%s (%s)", prop.name, prop.gcnArchName);
    *out = ctx;
    return 0;
}

extern "C" void mifsk_ctx_destroy( mifsk_ctx *ctx )
{
    if ( !ctx )
	return;
    for ( TwEntry &e : ctx->tables )
	(void)hipFree(e.d_tw);
    for ( CfgEntry &e : ctx->configs ) {
	(void)hipFree(e.dev);
	for ( double *r : e.d_rot )
	    if ( r ) (void)hipFree(r);
    }
    if ( ctx->host )
	mifsk::host_work_destroy(ctx->host);
    {
	for ( void *st : ctx->chain.streams )
	    if ( st ) {
		(void)hipStreamSynchronize((hipStream_t)st);
		(void)hipStreamDestroy((hipStream_t)st);
	    }
	for ( void *e : ctx->chain.ev_done )
	    if ( e ) (void)hipEventDestroy((hipEvent_t)e);
	if ( ctx->chain.ev_fork ) (void)hipEventDestroy((hipEvent_t)ctx->chain.ev_fork);
	if ( ctx->chain.d_state ) (void)hipFree(ctx->chain.d_state);
    }
    delete ctx;
}

extern "C" const char *mifsk_ctx_device_name( const mifsk_ctx *ctx )
{
    return ctx ? ctx->name : "";
}

// exp(-2 pi i b n / N): the angle is reduced exactly in integers first
static inline void twiddle( unsigned b, unsigned n, unsigned fftsize, double w[2] )
{
    const unsigned long long k = ( (unsigned long long)b * n ) % fftsize;
    const double ang = 2.0 * M_PI * (double)k / (double)fftsize;
    w[0] = std::cos(ang);
    w[1] = -std::sin(ang);
}

// The caches are bounded: the legacy API makes a DevCfg per window shape and a twiddle
// table per tone pair it is re-tuned to (fsk_set_tones_by_bandshift).  Called at the top
// of every entry point that launches, BEFORE it takes the gate: when a cache has outgrown
// its bound, wait until no call is between lookup and launch (exclusive gate), let the
// device finish what is running, and drop everything -- the next lookups rebuild what is
// still in use.
static void cache_gc( mifsk_ctx *ctx )
{
    constexpr size_t kMaxConfigs = 64, kMaxTables = 64, kMaxTableBytes = 64u << 20;
    {
	std::lock_guard<std::mutex> g(ctx->lock);
	if ( ctx->configs.size() < kMaxConfigs && ctx->tables.size() < kMaxTables
		&& ctx->table_bytes < kMaxTableBytes )
	    return;
    }
    std::unique_lock<std::shared_mutex> x(ctx->gate);
    std::lock_guard<std::mutex> g(ctx->lock);
    // (another caller may have flushed while this one waited for the gate)
    if ( ctx->configs.size() < kMaxConfigs && ctx->tables.size() < kMaxTables
	    && ctx->table_bytes < kMaxTableBytes )
	return;
    (void)hipDeviceSynchronize();
    for ( CfgEntry &e : ctx->configs ) {
	(void)hipFree(e.dev);
	for ( double *r : e.d_rot )
	    if ( r ) (void)hipFree(r);
    }
    ctx->configs.clear();
    for ( TwEntry &e : ctx->tables )
	(void)hipFree(e.d_tw);
    ctx->tables.clear();
    ctx->table_bytes = 0;
}

static int get_twiddles( mifsk_ctx *ctx, const TwKey &key, const double **d_out )
{
    std::lock_guard<std::mutex> g(ctx->lock);
    for ( const TwEntry &e : ctx->tables )
	if ( e.key == key ) {
	    *d_out = e.d_tw;
	    return 0;
	}
    // zero-padded: the workgroup kernel consumes the table in chunks of 8 (and
    // fetches one half chunk ahead), the wavefront kernel in groups of 16 (one
    // group ahead, and keeps groups 0..2 resident): mifsk_tw_entries()
    const size_t n = mifsk::tw_entries(key.bit_nsamples);
    std::vector<double> h(4 * ( n ? n : 8 ), 0.0);
    for ( unsigned i = 0; i < key.bit_nsamples; i++ ) {
	twiddle(key.b_mark, i, key.fftsize, &h[4 * (size_t)i]);
	twiddle(key.b_space, i, key.fftsize, &h[4 * (size_t)i + 2]);
    }
    double *d = nullptr;
    HIP_OK(hipMalloc(&d, h.size() * sizeof(double)));
    if ( hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ) {
	(void)hipFree(d);
	return -EIO;
    }
    ctx->tables.push_back(TwEntry{key, d});
    ctx->table_bytes += h.size() * sizeof(double);
    *d_out = d;
    return 0;
}

static int get_devcfg( mifsk_ctx *ctx, const DevCfg &d, const DevCfg **d_out, CfgEntry *entry_out = nullptr )
{
    std::lock_guard<std::mutex> g(ctx->lock);
    for ( const CfgEntry &e : ctx->configs )
	if ( std::memcmp(&e.host, &d, sizeof(DevCfg)) == 0 ) {
	    *d_out = e.dev;
	    if ( entry_out ) *entry_out = e;		// (a copy: the vector may grow under another caller)
	    return 0;
	}
    DevCfg *dev = nullptr;
    HIP_OK(hipMalloc(&dev, sizeof(DevCfg)));
    if ( hipMemcpy(dev, &d, sizeof(DevCfg), hipMemcpyHostToDevice) != hipSuccess ) {
	(void)hipFree(dev);
	return -EIO;
    }
    CfgEntry ce;
    ce.host = d;
    ce.dev = dev;
    size_t rot_bytes = 0;
    for ( int k = 0; k < 5; k++ )
	ce.d_rot[k] = nullptr;
    // shared segments: every window's rotation factors, [segment of the window][window]
    for ( int k = 0; k < 5; k++ ) {
	ce.d_rot[k] = nullptr;
	ce.rot_stride[k] = 0;
	const mifsk::SegPlan &sp = d.seg[k];
	if ( !sp.valid )
	    continue;
	unsigned cmax = 0;
	for ( unsigned w = 0; w < sp.nwin; w++ )
	    cmax = sp.win_count[w] > cmax ? sp.win_count[w] : cmax;
	const unsigned stride = ( sp.nwin + 63u ) & ~63u;
	// (zero rows beyond the longest window's: Wave::seg_correlate walks the rows with a running
	// pointer, four segments a turn, and where two lanes share a window -- every other row each --
	// up to 2 x 3 + 1 rows beyond the last)
	std::vector<double> h((size_t)( ( cmax + 12u + 3u ) & ~3u ) * stride * 4, 0.0);
	for ( unsigned w = 0; w < sp.nwin; w++ )
	    for ( unsigned i = 0; i < sp.win_count[w]; i++ ) {
		const unsigned off = sp.seg_rel[sp.win_first[w] + i] - ( sp.p_win[w] >> 16 );
		double *t = &h[( (size_t)i * stride + w ) * 4];
		twiddle(d.b_mark, off, d.fftsize, t);
		twiddle(d.b_space, off, d.fftsize, t + 2);
	    }
	if ( hipMalloc(&ce.d_rot[k], h.size() * sizeof(double)) != hipSuccess
		|| hipMemcpy(ce.d_rot[k], h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ) {
	    // nothing of a half-made entry stays behind
	    for ( int j = 0; j <= k; j++ )
		if ( ce.d_rot[j] ) (void)hipFree(ce.d_rot[j]);
	    (void)hipFree(dev);
	    return -ENOMEM;
	}
	ce.rot_stride[k] = stride;
	rot_bytes += h.size() * sizeof(double);
    }
    ctx->table_bytes += rot_bytes;
    ctx->configs.push_back(ce);
    *d_out = dev;
    if ( entry_out ) *entry_out = ce;
    return 0;
}

// cos / -sin of 2 pi k / N for k < N (the spectrum table of fsk_detect_carrier), one per
// FFT size seen; like the other tables it is only ever freed by cache_gc()
static int get_cs( mifsk_ctx *ctx, unsigned N, const double **d_out )
{
    std::lock_guard<std::mutex> g(ctx->lock);
    for ( const TwEntry &e : ctx->tables )
	if ( e.key == TwKey{N, 0u, 0u, 0u} ) {		// (bit_nsamples 0: never a bit table's key)
	    *d_out = e.d_tw;
	    return 0;
	}
    std::vector<double> h(2 * (size_t)N);
    for ( unsigned k = 0; k < N; k++ ) {
	const double ang = 2.0 * M_PI * (double)k / (double)N;
	h[2 * (size_t)k] = std::cos(ang);
	h[2 * (size_t)k + 1] = -std::sin(ang);
    }
    double *d = nullptr;
    if ( hipMalloc(&d, h.size() * sizeof(double)) != hipSuccess )
	return -ENOMEM;
    if ( hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ) {
	(void)hipFree(d);
	return -ENOMEM;
    }
    ctx->tables.push_back(TwEntry{TwKey{N, 0u, 0u, 0u}, d});
    ctx->table_bytes += h.size() * sizeof(double);
    *d_out = d;
    return 0;
}

// fill_devcfg through the context's small cache (keyed by the configuration's bytes: two equal
// byte strings are equal configurations; unequal padding only costs a miss)
static void derive_cfg( mifsk_ctx *ctx, const mifsk_rx_config &cfg, DevCfg &d )
{
    {
	std::lock_guard<std::mutex> g(ctx->lock);
	for ( const mifsk_ctx::DerivedCfg &e : ctx->derived )
	    if ( std::memcmp(&e.key, &cfg, sizeof(cfg)) == 0 ) {
		d = e.d;
		return;
	    }
    }
    mifsk::fill_devcfg(d, cfg);
    std::lock_guard<std::mutex> g(ctx->lock);
    // (another caller with the same configuration may have filled it in while this one derived:
    // keep one entry per key)
    for ( const mifsk_ctx::DerivedCfg &e : ctx->derived )
	if ( std::memcmp(&e.key, &cfg, sizeof(cfg)) == 0 )
	    return;
    constexpr size_t kKeep = 8;
    if ( ctx->derived.size() < kKeep ) {
	ctx->derived.push_back(mifsk_ctx::DerivedCfg{cfg, d});
    } else {
	ctx->derived[ctx->derived_next % kKeep] = mifsk_ctx::DerivedCfg{cfg, d};
	ctx->derived_next++;
    }
}

int mifsk_check_cfg( const mifsk_rx_config *cfg )
{
    if ( !cfg || cfg->expect_n_bits == 0 || cfg->expect_n_bits > MIFSK_MAX_FRAME_BITS
	    || cfg->bit_nsamples == 0 || cfg->fftsize < 2 )
	return -EINVAL;
    return 0;
}

// ---------------------------------------------------------------------------
// batch entry points
// ---------------------------------------------------------------------------

extern "C" int mifsk_find_frame_batch( mifsk_ctx *ctx, const mifsk_rx_config *cfg,
	const float *d_samples, const mifsk_search *d_problems,
	mifsk_search_result *d_results, int nproblems, void *stream )
{
    if ( !ctx || mifsk_check_cfg(cfg) || ( nproblems > 0 && ( !d_samples || !d_problems || !d_results ) ) )
	return -EINVAL;
    HIP_OK(hipSetDevice(ctx->device));
    cache_gc(ctx);
    std::shared_lock<std::shared_mutex> gate(ctx->gate);	// lookup .. enqueue
    const double *d_tw = nullptr;
    int rc = get_twiddles(ctx, TwKey{(unsigned)cfg->fftsize, cfg->b_mark, cfg->b_space,
				     cfg->bit_nsamples}, &d_tw);
    if ( rc )
	return rc;
    DevCfg d;
    derive_cfg(ctx, *cfg, d);
    const DevCfg *d_cfg = nullptr;
    rc = get_devcfg(ctx, d, &d_cfg);
    if ( rc )
	return rc;
    return mifsk::launch_find_frame_batch(d, d_cfg, d_tw, d_samples, d_problems, d_results,
					  nproblems, stream);
}

// Engine.  One wavefront per stream is the general engine (every option,
// every mode).  Where bit windows are staged through LDS and long enough
// that correlation, not the per-frame decisions, is the work (linear
// LATTICE, >= 16 samples per bit: Bell-202, 2400 baud, ...), the workgroup
// engine's master / worker pipeline overlaps the two and wins at every
// batch size measured (0.32 vs 0.52 ms at 512 streams, 0.50 vs 0.59 at
// 1024, 1.64 vs 2.19 at 4096); at 12000 baud (4 samples per bit) the
// wavefront engine is 4 x faster.  With longer windows and a clean signal it
// still wins (tools/gpu/eng50.py, 2048 streams: 50 baud 3.0 vs 4.4 ms, 150 baud
// 1.3 vs 4.1 ms).  Identical results either way.
static bool use_workgroup_engine( const mifsk_rx_config *cfg, const DevCfg &d, unsigned flags )
{
    const bool plain = !( flags & MIFSK_IO_RING_EXACT ) && !( cfg->auto_carrier_threshold > 0.0f );
    bool workgroup = plain && !( flags & MIFSK_IO_ENGINE_WAVE )
		  && ( ( flags & MIFSK_IO_ENGINE_WORKGROUP )
		       || ( d.lat_linear && d.bit_nsamples >= 16u ) );
    if ( const char *e = mifsk::experiment_env("MIFSK_ENGINE") )	// diagnostic override: "workgroup" / "wave"
	if ( !( flags & ( MIFSK_IO_ENGINE_WORKGROUP | MIFSK_IO_ENGINE_WAVE ) ) )
	    workgroup = plain && e[0] == 'w' && e[1] == 'o';
    return workgroup;
}

// What a chained launch needs (mifsk_device.h WaveChain), with room for `ns` streams' states.
// Called with ctx->chain_lock held.
static int chain_prepare( mifsk_ctx *ctx, size_t ns )
{
    mifsk::WaveChain &ch = ctx->chain;
    if ( !ctx->chain_made ) {
	// (whatever an earlier, failed attempt made is kept and used: nothing is created twice)
	for ( void *&st : ch.streams )
	    if ( !st ) {
		hipStream_t h = nullptr;
		HIP_OK(hipStreamCreateWithFlags(&h, hipStreamNonBlocking));
		st = h;
	    }
	if ( !ch.ev_fork ) {
	    hipEvent_t e = nullptr;
	    HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	    ch.ev_fork = e;
	}
	for ( void *&d : ch.ev_done )
	    if ( !d ) {
		hipEvent_t e = nullptr;
		HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		d = e;
	    }
	ctx->chain_made = true;
    }
    if ( ns > ch.state_cap ) {
	// (the chain before may still be running on the old array)
	for ( void *st : ch.streams )
	    HIP_OK(hipStreamSynchronize((hipStream_t)st));
	if ( ch.d_state )
	    (void)hipFree(ch.d_state);
	ch.d_state = nullptr;
	ch.state_cap = 0;
	const size_t cap = ( ns + 1023 ) & ~(size_t)1023;
	if ( hipMalloc((void **)&ch.d_state, cap * sizeof(mifsk_stream_state)) != hipSuccess )
	    return -ENOMEM;
	ch.state_cap = cap;
    }
    return 0;
}

// One wavefront per stream (mifsk_wave.hip): --auto-carrier and RING addressing
// need per-call device scratch; it is allocated and freed in stream order, so
// concurrent calls on different streams never share it.
static int demod_batch_wave( mifsk_ctx *ctx, const mifsk_rx_config *cfg, const DevCfg &d,
	const DevCfg *d_cfg, const double *d_tw, const mifsk_demod_io *io, void *stream,
	mifsk::LaunchInfo *plan_only = nullptr, mifsk_stream_state *d_state = nullptr,
	const uint64_t *d_origin = nullptr, bool final = true, const CfgEntry *tables = nullptr,
	float *d_ring_persistent = nullptr )
{
    hipStream_t st = (hipStream_t)stream;
    const size_t ns = (size_t)io->nstreams;
    mifsk::WaveHostArgs ha;
    std::memset(&ha, 0, sizeof(ha));
    ha.ncu = ctx->ncu;
    ha.samplebuf_size = cfg->samplebuf_size;
    ha.fftsize = (uint32_t)cfg->fftsize;
    ha.nbands = cfg->nbands;
    ha.tw_entries = (uint32_t)mifsk::tw_entries(cfg->bit_nsamples);
    ha.d_state = d_state;
    ha.d_origin = d_origin;
    ha.final = final;
    // a whole-stream call over a plain batch may be cut into chained launches (the launcher
    // decides by the batch's shape)
    ha.chain_ok = !d_state && !( io->flags & MIFSK_IO_RING_EXACT ) && !io->d_counters;
    // (--auto-carrier retunes per stream: its rotation factors come from the stream's own table)
    if ( tables && !( cfg->auto_carrier_threshold > 0.0f ) )
	for ( int k = 0; k < 5; k++ ) {
	    ha.d_rot[k] = tables->d_rot[k];
	    ha.rot_stride[k] = tables->rot_stride[k];
	}
    if ( plan_only ) {
	ha.ring_exact = ( io->flags & MIFSK_IO_RING_EXACT ) != 0;
	ha.autodetect = cfg->auto_carrier_threshold > 0.0f;
	return mifsk::launch_demod_wave(d, d_cfg, d_tw, *io, ha, stream, plan_only);
    }
    void *scratch_tw = nullptr, *scratch_ring = nullptr;
    if ( cfg->auto_carrier_threshold > 0.0f ) {
	// default negative shift, in the reference's float arithmetic (minimodem.c:1203-1206)
	int b_shift = - (float)( cfg->autodetect_shift + cfg->band_width / 2.0f ) / cfg->band_width;
	if ( cfg->inverted_freqs )
	    b_shift *= -1;
	if ( b_shift == 0 )
	    return -EINVAL;			// assert in fsk_set_tones_by_bandshift (fsk.c:587)
	if ( (unsigned long long)cfg->nbands * cfg->bit_nsamples > 0xFFFFFFFFull )
	    return -EINVAL;			// (band * n is reduced in 32 bits on the device)
	const double *d_cs = nullptr;
	int rc = get_cs(ctx, (unsigned)cfg->fftsize, &d_cs);
	if ( rc )
	    return rc;
	ha.autodetect = true;
	ha.auto_threshold = cfg->auto_carrier_threshold;
	ha.nps = cfg->nsamples_per_bit > (float)cfg->fftsize ? (float)cfg->fftsize
							      : cfg->nsamples_per_bit;
	ha.b_shift = b_shift;
	ha.d_cs = d_cs;
	if ( hipMallocAsync(&scratch_tw, ns * ha.tw_entries * 4 * sizeof(double), st) != hipSuccess )
	    return -ENOMEM;
	ha.d_tw_scratch = (double *)scratch_tw;
    }
    if ( io->flags & MIFSK_IO_RING_EXACT ) {
	// samplebuf plus what a search at the top of it may touch beyond
	const size_t reach = (size_t)( cfg->try_max[0] > cfg->try_max[1] ? cfg->try_max[0] : cfg->try_max[1] )
			   + d.last_reach + 64;
	ha.ring_exact = true;
	ha.ring_stride = (uint32_t)( ( (size_t)cfg->samplebuf_size + reach + 3 ) & ~(size_t)3 );
	if ( d_ring_persistent ) {
	    // mifsk_demod_slab_ring: the caller's buffer IS the reference's samplebuf between calls
	    ha.d_ring = d_ring_persistent;
	} else {
	    if ( hipMallocAsync(&scratch_ring, ns * ha.ring_stride * sizeof(float), st) != hipSuccess
		    || hipMemsetAsync(scratch_ring, 0, ns * ha.ring_stride * sizeof(float), st) != hipSuccess ) {
		if ( scratch_tw ) (void)hipFreeAsync(scratch_tw, st);
		return -ENOMEM;
	    }
	    ha.d_ring = (float *)scratch_ring;
	}
    }
    int rc;
    if ( ha.chain_ok ) {
	mifsk::LaunchInfo li;
	std::memset(&li, 0, sizeof(li));
	rc = mifsk::launch_demod_wave(d, d_cfg, d_tw, *io, ha, stream, &li);
	if ( rc == 0 && li.chain_groups ) {
	    std::lock_guard<std::mutex> one(ctx->chain_lock);
	    rc = chain_prepare(ctx, ns);
	    if ( rc ) {
		if ( scratch_tw ) (void)hipFreeAsync(scratch_tw, st);
		if ( scratch_ring ) (void)hipFreeAsync(scratch_ring, st);
		return rc;
	    }
	    ha.chain = &ctx->chain;
	    rc = mifsk::launch_demod_wave(d, d_cfg, d_tw, *io, ha, stream);
	    // (the caller's stream has joined the groups' by now: freed in stream order behind them)
	    if ( scratch_tw ) (void)hipFreeAsync(scratch_tw, st);
	    return rc;
	}
    }
    rc = mifsk::launch_demod_wave(d, d_cfg, d_tw, *io, ha, stream);
    if ( scratch_tw ) (void)hipFreeAsync(scratch_tw, st);
    if ( scratch_ring ) (void)hipFreeAsync(scratch_ring, st);
    return rc;
}

// The workgroup engine: one call over whole streams (maybe cut into chained launches of its
// resumable instantiation: the launcher decides by the batch's shape) or, with d_state, one
// slab of streams that arrive in pieces.
static int demod_batch_workgroup( mifsk_ctx *ctx, const mifsk_rx_config *cfg, const DevCfg &d,
	const DevCfg *d_cfg, const double *d_tw, const mifsk_demod_io *io, void *stream,
	mifsk::LaunchInfo *plan_only = nullptr, mifsk_stream_state *d_state = nullptr,
	const uint64_t *d_origin = nullptr, bool final = true )
{
    mifsk::WgHostArgs wh;
    std::memset(&wh, 0, sizeof(wh));
    wh.ncu = ctx->ncu;
    wh.samplebuf_size = cfg->samplebuf_size;
    wh.d_state = d_state;
    wh.d_origin = d_origin;
    wh.final = final;
    wh.chain_ok = !d_state && !io->d_counters;
    if ( plan_only )
	return mifsk::launch_demod_batch(d, d_cfg, d_tw, *io, stream, plan_only, &wh);
    if ( wh.chain_ok && mifsk::experiment_env("MIFSK_CHAIN") ) {	// (this engine's default is one launch)
	mifsk::LaunchInfo li;
	std::memset(&li, 0, sizeof(li));
	int rc = mifsk::launch_demod_batch(d, d_cfg, d_tw, *io, stream, &li, &wh);
	if ( rc == 0 && li.chain_groups ) {
	    std::lock_guard<std::mutex> one(ctx->chain_lock);
	    rc = chain_prepare(ctx, (size_t)io->nstreams);
	    if ( rc )
		return rc;
	    wh.chain = &ctx->chain;
	    return mifsk::launch_demod_batch(d, d_cfg, d_tw, *io, stream, nullptr, &wh);
	}
    }
    return mifsk::launch_demod_batch(d, d_cfg, d_tw, *io, stream, nullptr, &wh);
}

extern "C" int mifsk_demod_batch( mifsk_ctx *ctx, const mifsk_rx_config *cfg,
	const mifsk_demod_io *io, void *stream )
{
    if ( !ctx || !io || mifsk_check_cfg(cfg) )
	return -EINVAL;
    if ( io->nstreams < 0 || ( io->nstreams > 0 && !io->d_samples ) )
	return -EINVAL;
    if ( io->stream_stride % 4 != 0 || ( (uintptr_t)io->d_samples & 15u ) )
	return -EINVAL;		// rows must be 16-byte aligned (coalesced float4 staging)
    if ( ( io->d_bytes || io->d_bits || io->d_frames ) && io->frames_cap == 0 )
	return -EINVAL;
    if ( io->flags & ~( MIFSK_IO_RING_EXACT | MIFSK_IO_ENGINE_WORKGROUP | MIFSK_IO_ENGINE_WAVE ) )
	return -EINVAL;
    if ( ( io->flags & MIFSK_IO_ENGINE_WORKGROUP ) && ( io->flags & MIFSK_IO_ENGINE_WAVE ) )
	return -EINVAL;
    // the workgroup engine has neither RING addressing nor the in-loop --auto-carrier
    if ( ( io->flags & MIFSK_IO_ENGINE_WORKGROUP )
	    && ( ( io->flags & MIFSK_IO_RING_EXACT ) || cfg->auto_carrier_threshold > 0.0f ) )
	return -EINVAL;
    HIP_OK(hipSetDevice(ctx->device));
    cache_gc(ctx);
    std::shared_lock<std::shared_mutex> gate(ctx->gate);	// lookup .. enqueue
    const double *d_tw = nullptr;
    int rc = get_twiddles(ctx, TwKey{(unsigned)cfg->fftsize, cfg->b_mark, cfg->b_space,
				     cfg->bit_nsamples}, &d_tw);
    if ( rc )
	return rc;
    DevCfg d;
    derive_cfg(ctx, *cfg, d);
    const DevCfg *d_cfg = nullptr;
    CfgEntry tables;
    rc = get_devcfg(ctx, d, &d_cfg, &tables);
    if ( rc )
	return rc;
    if ( io->nstreams == 0 )
	return 0;
    const bool workgroup = use_workgroup_engine(cfg, d, io->flags);
    if ( !workgroup )
	return demod_batch_wave(ctx, cfg, d, d_cfg, d_tw, io, stream, nullptr, nullptr, nullptr, true, &tables);
    return demod_batch_workgroup(ctx, cfg, d, d_cfg, d_tw, io, stream);
}

static_assert(sizeof(mifsk_scan_plan) == sizeof(mifsk::SegPlan), "mifsk_scan_plan mirrors SegPlan");

extern "C" int mifsk_scan_plan_get( const mifsk_rx_config *cfg, int kind, mifsk_scan_plan *out )
{
    if ( mifsk_check_cfg(cfg) || !out || kind < 0 || kind > 4 )
	return -EINVAL;
    DevCfg *d = new (std::nothrow) DevCfg();
    if ( !d )
	return -ENOMEM;
    mifsk::fill_devcfg(*d, *cfg);
    std::memcpy(out, &d->seg[kind], sizeof(*out));
    delete d;
    return 0;
}

// the same loop for streams that arrive in pieces: state in, state out (either engine: the
// same choice as mifsk_demod_batch makes, the same state record)
extern "C" int mifsk_demod_slab( mifsk_ctx *ctx, const mifsk_rx_config *cfg, const mifsk_demod_io *io,
	mifsk_stream_state *d_state, const uint64_t *d_origin, int final, void *stream )
{
    if ( !ctx || !io || !d_state || mifsk_check_cfg(cfg) )
	return -EINVAL;
    if ( cfg->samplebuf_size < 2u )		// (the loop refills half a buffer at a time)
	return -EINVAL;
    if ( io->nstreams < 0 || ( io->nstreams > 0 && !io->d_samples ) )
	return -EINVAL;
    if ( io->stream_stride % 4 != 0 || ( (uintptr_t)io->d_samples & 15u ) )
	return -EINVAL;
    if ( ( io->d_bytes || io->d_bits || io->d_frames ) && io->frames_cap == 0 )
	return -EINVAL;
    if ( io->flags & ~( MIFSK_IO_ENGINE_WAVE | MIFSK_IO_ENGINE_WORKGROUP ) )	// flat addressing
	return -EINVAL;
    if ( ( io->flags & MIFSK_IO_ENGINE_WORKGROUP )
	    && ( ( io->flags & MIFSK_IO_ENGINE_WAVE ) || cfg->auto_carrier_threshold > 0.0f ) )
	return -EINVAL;
    HIP_OK(hipSetDevice(ctx->device));
    cache_gc(ctx);
    std::shared_lock<std::shared_mutex> gate(ctx->gate);	// lookup .. enqueue
    const double *d_tw = nullptr;
    int rc = get_twiddles(ctx, TwKey{(unsigned)cfg->fftsize, cfg->b_mark, cfg->b_space,
				     cfg->bit_nsamples}, &d_tw);
    if ( rc )
	return rc;
    DevCfg d;
    derive_cfg(ctx, *cfg, d);
    const DevCfg *d_cfg = nullptr;
    CfgEntry tables;
    rc = get_devcfg(ctx, d, &d_cfg, &tables);
    if ( rc )
	return rc;
    if ( io->nstreams == 0 )
	return 0;
    if ( use_workgroup_engine(cfg, d, io->flags) )
	return demod_batch_workgroup(ctx, cfg, d, d_cfg, d_tw, io, stream, nullptr, d_state, d_origin, final != 0);
    return demod_batch_wave(ctx, cfg, d, d_cfg, d_tw, io, stream, nullptr, d_state, d_origin, final != 0, &tables);
}

// floats per stream of the buffer mifsk_demod_slab_ring keeps the reference's samplebuf in
extern "C" size_t mifsk_ring_floats( const mifsk_rx_config *cfg )
{
    if ( mifsk_check_cfg(cfg) )
	return 0;
    DevCfg *d = new (std::nothrow) DevCfg();
    if ( !d )
	return 0;
    mifsk::fill_devcfg(*d, *cfg);
    const size_t reach = (size_t)( cfg->try_max[0] > cfg->try_max[1] ? cfg->try_max[0] : cfg->try_max[1] )
		       + d->last_reach + 64;
    delete d;
    return ( (size_t)cfg->samplebuf_size + reach + 3 ) & ~(size_t)3;
}

// mifsk_demod_slab with the reference's buffer semantics (MIFSK_IO_RING_EXACT) for streams fed
// in pieces: the samplebuf cells -- stale ones behind samples_nvalid included -- persist in
// d_ring between the calls (minimodem.c:1150-1156)
extern "C" int mifsk_demod_slab_ring( mifsk_ctx *ctx, const mifsk_rx_config *cfg, const mifsk_demod_io *io,
	mifsk_stream_state *d_state, const uint64_t *d_origin, float *d_ring, int final, void *stream )
{
    if ( !ctx || !io || !d_state || !d_ring || mifsk_check_cfg(cfg) || cfg->samplebuf_size < 2u )
	return -EINVAL;
    if ( io->nstreams < 0 || ( io->nstreams > 0 && !io->d_samples ) )
	return -EINVAL;
    if ( io->stream_stride % 4 != 0 || ( (uintptr_t)io->d_samples & 15u ) )
	return -EINVAL;
    if ( ( io->d_bytes || io->d_bits || io->d_frames ) && io->frames_cap == 0 )
	return -EINVAL;
    if ( io->flags & ~( MIFSK_IO_ENGINE_WAVE | MIFSK_IO_RING_EXACT ) )	// (RING addressing is the wavefront engine's)
	return -EINVAL;
    HIP_OK(hipSetDevice(ctx->device));
    cache_gc(ctx);
    std::shared_lock<std::shared_mutex> gate(ctx->gate);	// lookup .. enqueue
    const double *d_tw = nullptr;
    int rc = get_twiddles(ctx, TwKey{(unsigned)cfg->fftsize, cfg->b_mark, cfg->b_space,
				     cfg->bit_nsamples}, &d_tw);
    if ( rc )
	return rc;
    DevCfg d;
    derive_cfg(ctx, *cfg, d);
    const DevCfg *d_cfg = nullptr;
    CfgEntry tables;
    rc = get_devcfg(ctx, d, &d_cfg, &tables);
    if ( rc )
	return rc;
    if ( io->nstreams == 0 )
	return 0;
    mifsk_demod_io rio = *io;
    rio.flags |= MIFSK_IO_RING_EXACT;
    return demod_batch_wave(ctx, cfg, d, d_cfg, d_tw, &rio, stream, nullptr, d_state, d_origin, final != 0,
			    &tables, d_ring);
}

// what mifsk_demod_batch would launch for this configuration and batch size
extern "C" int mifsk_demod_plan( mifsk_ctx *ctx, const mifsk_rx_config *cfg, int nstreams,
	unsigned flags, mifsk_launch_info *out )
{
    return mifsk_demod_plan_ex(ctx, cfg, nstreams, 0xFFFFFFFFu, flags, out);
}

extern "C" int mifsk_demod_plan_ex( mifsk_ctx *ctx, const mifsk_rx_config *cfg, int nstreams,
	uint32_t nsamples, unsigned flags, mifsk_launch_info *out )
{
    if ( !ctx || !out || mifsk_check_cfg(cfg) || nstreams < 0 )
	return -EINVAL;
    DevCfg d;
    derive_cfg(ctx, *cfg, d);
    mifsk_demod_io io;
    std::memset(&io, 0, sizeof(io));
    io.nstreams = nstreams;
    io.nsamples = nsamples;
    io.flags = flags;
    mifsk::LaunchInfo li;
    std::memset(&li, 0, sizeof(li));
    const bool workgroup = use_workgroup_engine(cfg, d, flags);
    int rc = workgroup ? demod_batch_workgroup(ctx, cfg, d, nullptr, nullptr, &io, nullptr, &li)
		       : demod_batch_wave(ctx, cfg, d, nullptr, nullptr, &io, nullptr, &li);
    if ( rc )
	return rc;
    std::memset(out, 0, sizeof(*out));
    std::snprintf(out->kernel, sizeof(out->kernel), "%s", li.kernel ? li.kernel : "");
    out->engine = workgroup ? MIFSK_IO_ENGINE_WORKGROUP : MIFSK_IO_ENGINE_WAVE;
    out->workgroup_size = li.workgroup_size;
    out->lds_bytes_per_workgroup = li.lds_bytes;
    const unsigned by_lds = li.lds_bytes ? (unsigned)( 160u * 1024u / li.lds_bytes ) : 32u;
    // (waves per SIMD the instantiation's VGPR budget allows x 4 SIMDs)
    const unsigned by_waves = ( li.waves_per_simd ? li.waves_per_simd : 8u ) * 4u * 64u
			    / ( li.workgroup_size ? li.workgroup_size : 64u );
    out->workgroups_per_cu = by_lds < by_waves ? by_lds : by_waves;
    out->lattice_mode = li.lattice_mode;
    out->frames_per_block = li.frames_per_block;
    out->compute_units = (uint32_t)ctx->ncu;
    out->chain_groups = li.chain_groups;
    out->chain_chunks = li.chain_chunks;
    return 0;
}

// ---------------------------------------------------------------------------
// several devices, one process: streams are independent, so context k (one per
// GPU) takes the contiguous range mifsk_shard_range(nstreams, k, nctx) and the
// "gather" is each device's copy back into the caller's arrays -- no collective.
// (One process per GPU with an RCCL gather of the decoded bytes is the other
// arrangement: minimodem_amd.ByteGatherer, bench.py --gpus N.)
// ---------------------------------------------------------------------------

extern "C" void mifsk_shard_range( int nstreams, int rank, int world, int *lo, int *hi )
{
    if ( world <= 0 ) world = 1;
    if ( nstreams < 0 ) nstreams = 0;
    const int q = nstreams / world, r = nstreams % world;
    const int a = rank * q + ( rank < r ? rank : r );
    if ( lo ) *lo = a;
    if ( hi ) *hi = a + q + ( rank < r ? 1 : 0 );
}

extern "C" int mifsk_demod_batch_host_multi( mifsk_ctx *const *ctxs, int nctx,
	const mifsk_rx_config *cfg, const mifsk_demod_io *hio )
{
    if ( !ctxs || nctx <= 0 || !hio || hio->nstreams < 0 )
	return -EINVAL;
    for ( int k = 0; k < nctx; k++ )
	if ( !ctxs[k] )
	    return -EINVAL;
    std::vector<int> rcs((size_t)nctx, 0);
    std::vector<std::thread> workers;
    for ( int k = 0; k < nctx; k++ ) {
	int lo = 0, hi = 0;
	mifsk_shard_range(hio->nstreams, k, nctx, &lo, &hi);
	if ( hi == lo )
	    continue;
	mifsk_demod_io io = *hio;
	const size_t o = (size_t)lo, fc = hio->frames_cap, ec = hio->episodes_cap;
	io.nstreams = hi - lo;
	io.d_samples = (const float *)( (const char *)hio->d_samples
					+ o * hio->stream_stride * ( ( hio->flags & MIFSK_IO_HOST_S16 ) ? 2u : 4u ) );
	if ( io.d_nsamples )	 io.d_nsamples += o;
	if ( io.d_bytes )	 io.d_bytes += o * fc;
	if ( io.d_nbytes )	 io.d_nbytes += o;
	if ( io.d_bits )	 io.d_bits += o * fc;
	if ( io.d_frames )	 io.d_frames += o * fc;
	if ( io.d_nframes )	 io.d_nframes += o;
	if ( io.d_episodes )	 io.d_episodes += o * ec;
	if ( io.d_nepisodes )	 io.d_nepisodes += o;
	if ( io.d_status )	 io.d_status += o;
	if ( io.d_counters )	 io.d_counters += o * MIFSK_NCOUNTERS;
	if ( io.d_carrier_band ) io.d_carrier_band += o;
	mifsk_ctx *ctx = ctxs[k];
	int *rcp = &rcs[(size_t)k];
	// one host thread per device: the HIP "current device" is per thread, so
	// the per-device copies, launch and synchronisation proceed side by side
	workers.emplace_back([ctx, cfg, io, rcp]() { *rcp = mifsk_demod_batch_host(ctx, cfg, &io); });
    }
    for ( std::thread &t : workers )
	t.join();
    for ( int rc : rcs )
	if ( rc )
	    return rc;
    return 0;
}

// ---------------------------------------------------------------------------
// the reference's five-function API (include/fsk.h) over the same kernels
// ---------------------------------------------------------------------------

namespace {

struct LegacyPlan {
    mifsk_ctx		*ctx;
    float		*d_samples;	size_t cap_samples;
    mifsk_search	*d_problem;
    mifsk_search_result	*d_result;
    float		*d_mags;	size_t cap_mags;
};

int legacy_reserve( LegacyPlan *lp, size_t nsamples )
{
    if ( nsamples <= lp->cap_samples )
	return 0;
    if ( lp->d_samples )
	(void)hipFree(lp->d_samples);
    lp->d_samples = nullptr;
    lp->cap_samples = 0;
    const size_t cap = ( nsamples + 4095 ) & ~(size_t)4095;
    if ( hipMalloc(&lp->d_samples, cap * sizeof(float)) != hipSuccess )
	return -ENOMEM;
    lp->cap_samples = cap;
    return 0;
}

} // namespace

extern "C" fsk_plan *fsk_plan_new( float sample_rate, float f_mark, float f_space,
	float filter_bw )
{
    fsk_plan *p = (fsk_plan *)std::calloc(1, sizeof(fsk_plan));
    if ( !p ) {
	errno = ENOMEM;
	return nullptr;
    }
    p->sample_rate = sample_rate;
    p->f_mark = f_mark;
    p->f_space = f_space;
    p->band_width = filter_bw;
    const float half = p->band_width / 2.0f;			// fsk.c:52-57
    p->fftsize = (int)( (sample_rate + half) / p->band_width );
    p->nbands = (unsigned)( p->fftsize / 2 + 1 );
    p->b_mark = (unsigned)( (f_mark + half) / p->band_width );
    p->b_space = (unsigned)( (f_space + half) / p->band_width );
    if ( p->b_mark >= p->nbands || p->b_space >= p->nbands ) {	// fsk.c:58-64
	fprintf(stderr, "b_mark=%u or b_space=%u is invalid (nbands=%u)\n",
		p->b_mark, p->b_space, p->nbands);
	std::free(p);
	errno = EINVAL;
	return nullptr;
    }
    LegacyPlan *lp = new (std::nothrow) LegacyPlan();
    if ( !lp ) {
	std::free(p);
	errno = ENOMEM;
	return nullptr;
    }
    std::memset(lp, 0, sizeof(*lp));
    int rc = mifsk_ctx_create(&lp->ctx, -1);
    if ( rc == 0 && ( hipMalloc(&lp->d_problem, sizeof(mifsk_search)) != hipSuccess
		   || hipMalloc(&lp->d_result, sizeof(mifsk_search_result)) != hipSuccess ) )
	rc = -ENOMEM;
    if ( rc ) {
	if ( lp->ctx ) mifsk_ctx_destroy(lp->ctx);
	delete lp;
	std::free(p);
	errno = -rc;
	return nullptr;
    }
    p->fftplan = lp;
    p->fftin = nullptr;
    p->fftout = nullptr;
    return p;
}

extern "C" void fsk_plan_destroy( fsk_plan *p )
{
    if ( !p )
	return;
    LegacyPlan *lp = (LegacyPlan *)p->fftplan;
    if ( lp ) {
	if ( lp->d_samples ) (void)hipFree(lp->d_samples);
	if ( lp->d_problem ) (void)hipFree(lp->d_problem);
	if ( lp->d_result ) (void)hipFree(lp->d_result);
	if ( lp->d_mags ) (void)hipFree(lp->d_mags);
	mifsk_ctx_destroy(lp->ctx);
	delete lp;
    }
    std::free(p);
}

extern "C" float fsk_find_frame( fsk_plan *p, float *samples, unsigned int frame_nsamples,
	unsigned int try_first_sample, unsigned int try_max_nsamples,
	unsigned int try_step_nsamples, float try_confidence_search_limit,
	const char *expect_bits_string, unsigned long long *bits_outp,
	float *ampl_outp, unsigned int *frame_start_outp )
{
    *bits_outp = 0;
    *ampl_outp = 0.0f;
    *frame_start_outp = 0;
    LegacyPlan *lp = p ? (LegacyPlan *)p->fftplan : nullptr;
    const size_t n_bits = expect_bits_string ? std::strlen(expect_bits_string) : 0;
    if ( !lp || n_bits == 0 || n_bits > MIFSK_MAX_FRAME_BITS	// assert in fsk.c:463
	    || (int)try_first_sample >= (int)try_max_nsamples || try_step_nsamples == 0 )
	return 0.0f;

    // the slice of the configuration that fsk_find_frame itself derives
    // (fsk.c:465,183,204); everything else in DevCfg is unused by this kernel
    mifsk_rx_config c;
    std::memset(&c, 0, sizeof(c));
    c.expect_n_bits = (unsigned)n_bits;
    const float spb = (float)frame_nsamples / (float)(int)n_bits;
    c.find_samples_per_bit = spb;
    c.bit_nsamples = (unsigned)( spb + 0.5f );
    if ( c.bit_nsamples == 0 )
	return 0.0f;
    for ( unsigned k = 0; k < n_bits; k++ ) {
	c.bit_offset[k] = (unsigned)( spb * (float)(int)k + 0.5f );
	c.expect_data[k] = expect_bits_string[k];
	c.expect_sync[k] = expect_bits_string[k];
    }
    c.fftsize = p->fftsize;
    c.b_mark = p->b_mark;
    c.b_space = p->b_space;

    // what the reference can touch (fsk.c:204-206,477-484): the highest candidate
    // it may try, first + k * step < try_max, plus the last bit window -- not
    // try_max - 1: a caller whose buffer ends at the reference's true extent
    // must not be over-read
    const unsigned up_steps = ( try_max_nsamples - try_first_sample - 1u ) / try_step_nsamples;
    const size_t t_last = (size_t)try_first_sample + (size_t)up_steps * try_step_nsamples;
    const size_t reach = t_last + c.bit_offset[n_bits - 1] + c.bit_nsamples;
    if ( hipSetDevice(lp->ctx->device) != hipSuccess || legacy_reserve(lp, reach) )
	return 0.0f;
    mifsk_search pr;
    pr.sample_offset = 0;
    pr.navail = (uint32_t)reach;
    pr.try_first = try_first_sample;
    pr.try_max = try_max_nsamples;
    pr.try_step = try_step_nsamples;
    pr.search_limit = try_confidence_search_limit;
    pr.use_sync_string = 0;
    mifsk_search_result res;
    if ( hipMemcpy(lp->d_samples, samples, reach * sizeof(float), hipMemcpyHostToDevice) != hipSuccess
	    || hipMemcpy(lp->d_problem, &pr, sizeof(pr), hipMemcpyHostToDevice) != hipSuccess )
	return 0.0f;
    if ( mifsk_find_frame_batch(lp->ctx, &c, lp->d_samples, lp->d_problem, lp->d_result, 1, nullptr) )
	return 0.0f;
    if ( hipMemcpy(&res, lp->d_result, sizeof(res), hipMemcpyDeviceToHost) != hipSuccess )
	return 0.0f;
    *bits_outp = res.bits;
    *ampl_outp = res.amplitude;
    *frame_start_outp = res.frame_start;
    return res.confidence;
}

extern "C" int fsk_detect_carrier( fsk_plan *p, float *samples, unsigned int nsamples,
	float min_mag_threshold )
{
    LegacyPlan *lp = p ? (LegacyPlan *)p->fftplan : nullptr;
    if ( !lp || nsamples == 0 || nsamples > (unsigned)p->fftsize )	// assert in fsk.c:547
	return -1;
    mifsk_ctx *ctx = lp->ctx;
    if ( hipSetDevice(ctx->device) != hipSuccess )
	return -1;
    const unsigned N = (unsigned)p->fftsize;
    cache_gc(ctx);
    std::shared_lock<std::shared_mutex> gate(ctx->gate);	// lookup .. enqueue
    const double *d_cs = nullptr;
    if ( get_cs(ctx, N, &d_cs) )
	return -1;
    if ( legacy_reserve(lp, nsamples) )
	return -1;
    if ( lp->cap_mags < p->nbands ) {
	if ( lp->d_mags ) (void)hipFree(lp->d_mags);
	lp->d_mags = nullptr;
	lp->cap_mags = 0;
	if ( hipMalloc(&lp->d_mags, p->nbands * sizeof(float)) != hipSuccess )
	    return -1;
	lp->cap_mags = p->nbands;
    }
    if ( hipMemcpy(lp->d_samples, samples, nsamples * sizeof(float), hipMemcpyHostToDevice) != hipSuccess )
	return -1;
    if ( mifsk::launch_detect_carrier(lp->d_samples, nsamples, d_cs, N, p->nbands,
				      lp->d_mags, nullptr) )
	return -1;
    std::vector<float> mags(p->nbands);
    if ( hipMemcpy(mags.data(), lp->d_mags, p->nbands * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess )
	return -1;
    // peak pick over the device-computed magnitudes (fsk.c:554-580)
    float max_mag = 0.0f;
    int max_band = -1;
    for ( unsigned i = 1; i < p->nbands; i++ ) {
	const float mag = mags[i];
	if ( mag < min_mag_threshold )
	    continue;
	if ( max_mag < mag ) {
	    max_mag = mag;
	    max_band = (int)i;
	}
    }
    return max_band;
}

extern "C" void fsk_set_tones_by_bandshift( fsk_plan *p, unsigned int b_mark, int b_shift )
{
    if ( !p || b_shift == 0 || b_mark >= p->nbands )		// asserts in fsk.c:587-592
	return;
    const int b_space = (int)b_mark + b_shift;
    if ( b_space < 0 || b_space >= (int)p->nbands )
	return;
    p->b_mark = b_mark;
    p->b_space = (unsigned)b_space;
    p->f_mark = b_mark * p->band_width;
    p->f_space = b_space * p->band_width;
}
