// mifsk_databits.cpp -- host post-pass: frame data bits -> the text minimodem
// prints.  Plain host C++ behind the C ABI of include/mifsk.h;
// no device code: these decoders are O(1) per frame, stateful and byte-serial.
//
// Behaviour follows the reference's decoders, quirks included (each function
// cites the lines it mirrors); tests/test_databits.py checks them against
// vectors produced by the reference's own objects (tests/golden/).
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "mifsk.h"

namespace {

// ---- Baudot (ITA2, U.S. figures) -- baudot.c:32-70,217-243 -----------------
// index = 5-bit code; column 0 letters, column 1 U.S. figures.  '_' / '^' for
// NUL and '%' for the shift codes are the reference's debugging marks (the
// shift codes never print).
const char kLetters[33] = "_E\nA SIU\rDRJNFCKTZLWHYPQOBG%MXV%";
const char kFigures[33] = "^3\n- \a87\r$4',!:(5\")2#6019?&%./;%";
enum { kLtrs = 0x1F, kFigs = 0x1B, kSpace = 0x04 };

struct Baudot {
    unsigned charset = 0;	// 0 unknown, 1 letters, 2 figures (baudot.c:204-209)
    void reset() { charset = 1; }			// baudot.c:217-221
    unsigned decode( char *out, unsigned bits )		// baudot.c:229-255
    {
	bits &= 0x1Fu;
	if ( bits == kFigs ) { charset = 2; return 0; }
	if ( bits == kLtrs ) { charset = 1; return 0; }
	if ( bits == kSpace )
	    charset = 1;				// un-shift on space (always on in RX)
	*out = charset == 1 ? kLetters[bits] : kFigures[bits];
	return 1;
    }

    // baudot_encode (baudot.c:249-311): the 5-bit word(s) for one character -- a
    // shift code first when the character is not in the current set.  The
    // reference's encode table is the inverse of the decode tables above, plus
    // NUL (code 0, either set) and '+' (sent as 0x12 in figures).  Returns 0
    // for a character that cannot be sent.
    unsigned encode( unsigned *out, char ch )
    {
	int c = (unsigned char)ch;
	if ( c >= 'a' && c <= 'z' )
	    c -= 32;					// toupper, "C" locale
	if ( c >= 0x60 || ch < 0 )
	    return 0;
	unsigned code = 0, mask = 0;
	for ( unsigned k = 0; k < 32; k++ ) {
	    if ( k == kLtrs || k == kFigs )
		continue;
	    if ( k != 0 && (unsigned char)kLetters[k] == c ) { mask |= 1u; code = k; }
	    if ( k != 0 && (unsigned char)kFigures[k] == c ) { mask |= 2u; code = k; }
	}
	if ( c == 0 ) { mask = 3; code = 0; }
	if ( c == '+' ) { mask = 2; code = 0x12; }
	unsigned n = 0;
	if ( ( charset & mask ) == 0 ) {
	    if ( mask == 0 )
		return 0;
	    if ( charset == 0 )
		charset = 1;
	    if ( mask != 3 )
		charset = mask;
	    out[n++] = charset == 1 ? kLtrs : kFigs;
	}
	out[n++] = code;
	if ( c == ' ' )
	    charset = 1;				// TX un-shift on space
	return n;
    }
};

// ---- Caller-ID SDMF / MDMF -- databits_callerid.c ---------------------------
struct CallerId {
    int			msgtype = 0;
    unsigned		ndata = 0;
    // 256 message bytes as in the reference; the tail padding stands in for
    // the zero-initialised statics its over-reads land in
    unsigned char	buf[256 + 16] = { 0 };

    void reset() { msgtype = 0; ndata = 0; }		// :150-156 (the buffer keeps its bytes)

    // printf("%.*s"): up to prec chars, stopping at a NUL; prec < 0 = no limit
    void put( std::string &o, const unsigned char *p, int prec ) const
    {
	const unsigned char *end = buf + sizeof(buf);
	for ( int i = 0; ( prec < 0 || i < prec ) && p + i < end && p[i]; i++ )
	    o.push_back((char)p[i]);
    }
    void label( std::string &o, unsigned type ) const	// "%-6s " of cid_datatype_names[]
    {
	static const char *names[] = { "unknown0:", "Time:", "Phone:", "unknown3:", "Phone:",
				       "unknown5:", "unknown6:", "Name:", "Name:" };
	std::string n = names[type];
	while ( n.size() < 6 ) n.push_back(' ');
	o += n;
	o.push_back(' ');
    }
    void datetime( std::string &o, const unsigned char *m ) const	// "%.2s/%.2s %.2s:%.2s\n"
    {
	put(o, m, 2); o.push_back('/'); put(o, m + 2, 2); o.push_back(' ');
	put(o, m + 4, 2); o.push_back(':'); put(o, m + 6, 2); o.push_back('\n');
    }
    void phone10( std::string &o, const unsigned char *m ) const	// "%.3s-%.3s-%.4s\n"
    {
	put(o, m, 3); o.push_back('-'); put(o, m + 3, 3); o.push_back('-');
	put(o, m + 6, 4); o.push_back('\n');
    }

    std::string mdmf() const				// :50-122
    {
	std::string o;
	const unsigned msglen = buf[1];
	const unsigned char *m = buf + 2;
	unsigned i = 0;
	while ( i < msglen ) {
	    const unsigned type = *m++;
	    if ( type > 8 )
		return std::string();			// bad stream: nothing of the body is printed
	    const unsigned len = *m++;
	    if ( m + 2 + len >= buf + 256 )
		return std::string();
	    label(o, type);
	    const unsigned char *pr = nullptr;
	    const char *fixed = nullptr;
	    int prlen = 0;
	    switch ( type ) {
	    case 1: datetime(o, m); break;
	    case 2:
		if ( len == 10 ) { phone10(o, m); break; }
		/* fallthrough: a number that is not 10 digits prints like a name */
	    case 7: pr = m; prlen = (int)len; break;
	    case 4:
	    case 8:
		if ( len == 1 && *m == 'O' ) fixed = "[N/A]";
		else if ( len == 1 && *m == 'P' ) fixed = "[blocked]";
		break;
	    default: break;				// label only, no newline
	    }
	    if ( pr ) { put(o, pr, prlen); o.push_back('\n'); }
	    if ( fixed ) { o += fixed; o.push_back('\n'); }
	    m += len;
	    i += len + 2;
	}
	return o;
    }
    std::string sdmf() const				// :125-148
    {
	std::string o;
	const unsigned msglen = buf[1];
	const unsigned char *m = buf + 2;
	label(o, 1); datetime(o, m);
	m += 8;
	label(o, 2);
	const unsigned len = msglen - 8u;		// unsigned, as in the reference
	if ( len == 10 ) phone10(o, m);
	else { put(o, m, (int)len); o.push_back('\n'); }
	return o;
    }
    std::string decode( unsigned long long bits )	// :160-210
    {
	if ( msgtype == 0 ) {
	    if ( bits == 0x80 ) msgtype = 0x80;		// MDMF
	    else if ( bits == 0x04 ) msgtype = 0x04;	// SDMF
	    else return std::string();
	    buf[ndata++] = (unsigned char)bits;
	    return std::string();
	}
	if ( ndata >= 256 ) { reset(); return std::string(); }
	buf[ndata++] = (unsigned char)bits;
	// complete once type + length + `length` bytes are in (the checksum
	// byte is not waited for, nor checked)
	if ( ndata < (unsigned)buf[1] + 2u )
	    return std::string();
	std::string o = "CALLER-ID\n";
	o += msgtype == 0x80 ? mdmf() : sdmf();
	reset();
	return o;
    }
};

// ---- UIC-751-3 -- databits_uic.c:30-53, uic_codes.c:24-68 -------------------
struct UicCode { int code; const char *meaning; };
const UicCode kGroundToTrain[] = {
    { 0x00, "Test" }, { 0x02, "Run slower" }, { 0x03, "Extension of telegram" },
    { 0x04, "Run faster" }, { 0x06, "Written order" }, { 0x08, "Speech" },
    { 0x09, "Emergency stop" }, { 0x0C, "Announcem. by loudspeaker" }, { 0x55, "Idle" },
    { -1, nullptr } };
const UicCode kTrainToGround[] = {
    { 0x08, "Communic. desired" }, { 0x0A, "Acknowl. of order" }, { 0x06, "Advice" },
    { 0x00, "Test" }, { 0x09, "Train staff wish to comm." }, { 0x0C, "Telephone link desired" },
    { 0x03, "Extension of telegram" }, { -1, nullptr } };

unsigned long long window( unsigned long long v, unsigned off, unsigned n )	// databits.h:36-46
{
    return ( v >> off ) & ( ( 1ULL << n ) - 1ULL );
}

std::string uic( unsigned long long in, bool ground )
{
    unsigned code = 0;					// the message byte, bit-reversed
    for ( unsigned j = 0, v = (unsigned)window(in, 24, 8); j < 8; j++, v >>= 1 )
	code = ( code << 1 ) | ( v & 1u );
    const char *meaning = "Unknown";
    for ( const UicCode *t = ground ? kGroundToTrain : kTrainToGround; t->code != -1; t++ )
	if ( (unsigned)t->code == code ) { meaning = t->meaning; break; }
    char b[160];
    snprintf(b, sizeof b, "Train ID: %X%X%X%X%X%X - Message: %02X (%s)\n",
	     (unsigned)window(in, 0, 4), (unsigned)window(in, 4, 4), (unsigned)window(in, 8, 4),
	     (unsigned)window(in, 12, 4), (unsigned)window(in, 16, 4), (unsigned)window(in, 20, 4),
	     code, meaning);
    return b;
}

} // namespace

struct mifsk_databits {
    int		decoder;
    Baudot	baudot;
    CallerId	cid;
};

extern "C" int mifsk_databits_create( mifsk_databits **out, int decoder )
{
    if ( !out || decoder < MIFSK_DECODE_ASCII8 || decoder > MIFSK_DECODE_UIC_TRAIN )
	return -EINVAL;
    mifsk_databits *d = new (std::nothrow) mifsk_databits();
    if ( !d )
	return -ENOMEM;
    d->decoder = decoder;
    *out = d;
    return 0;
}

extern "C" void mifsk_databits_destroy( mifsk_databits *d ) { delete d; }

extern "C" void mifsk_databits_reset( mifsk_databits *d )
{
    if ( !d )
	return;
    if ( d->decoder == MIFSK_DECODE_BAUDOT ) d->baudot.reset();	// databits_baudot.c:33-36
    if ( d->decoder == MIFSK_DECODE_CALLERID ) d->cid.reset();		// databits_callerid.c:164-165
}

// bfsk_databits_encode (databits.h:49-92): the data word(s) one input character
// is transmitted as -- the byte itself (databits_ascii.c:118-124,
// databits_binary.c), or Baudot with shift codes.  The caller-ID and UIC
// decoders have no encoder in the reference either.
extern "C" unsigned mifsk_databits_encode( mifsk_databits *d, unsigned *words_out, char c )
{
    if ( !d || !words_out )
	return 0;
    if ( d->decoder == MIFSK_DECODE_BAUDOT )
	return d->baudot.encode(words_out, c);
    if ( d->decoder == MIFSK_DECODE_ASCII8 || d->decoder == MIFSK_DECODE_BINARY ) {
	words_out[0] = (unsigned char)c;
	return 1;
    }
    return 0;
}

extern "C" unsigned mifsk_databits_decode( mifsk_databits *d, char *out, unsigned out_size,
	unsigned long long bits, unsigned n_databits )
{
    if ( !d || !out ) {			// the reference's "reset" calling convention
	mifsk_databits_reset(d);
	return 0;
    }
    std::string o;
    switch ( d->decoder ) {
    case MIFSK_DECODE_ASCII8:		// databits_ascii.c:38-44
	o.push_back((char)( bits & 0xFF ));
	break;
    case MIFSK_DECODE_BAUDOT: {		// databits_baudot.c:30-39
	char c;
	if ( d->baudot.decode(&c, (unsigned)bits) )
	    o.push_back(c);
	break;
    }
    case MIFSK_DECODE_BINARY:		// databits_binary.c:30-41
	for ( unsigned j = 0; j < n_databits && j < 64; j++ )
	    o.push_back((char)( '0' + ( ( bits >> j ) & 1ULL ) ));
	o.push_back('\n');
	break;
    case MIFSK_DECODE_CALLERID:
	o = d->cid.decode(bits);
	break;
    case MIFSK_DECODE_UIC_GROUND: o = uic(bits, true); break;
    case MIFSK_DECODE_UIC_TRAIN: o = uic(bits, false); break;
    default: break;
    }
    const size_t n = o.size() < out_size ? o.size() : out_size;
    memcpy(out, o.data(), n);
    return (unsigned)n;
}

namespace {

struct Sink {
    char *p; size_t cap, len;
    void add( const char *s, size_t n )
    {
	for ( size_t i = 0; i < n; i++, len++ )
	    if ( len < cap )
		p[len] = s[i];
    }
    void add( const std::string &s ) { add(s.data(), s.size()); }
};

// "### CARRIER 1200 @ 1200.0 Hz ###\n" -- minimodem.c:1336-1348
// (fskp->b_mark at the time of the acquisition: the episode carries it, because
// --auto-carrier can move the tones between episodes, minimodem.c:1297,1219)
std::string carrier_line( const mifsk_rx_config *cfg, const mifsk_episode &e )
{
    char b[128];
    const double hz = (double)( (float)e.b_mark * cfg->band_width );
    if ( cfg->data_rate >= 100 )
	snprintf(b, sizeof b, "### CARRIER %u @ %.1f Hz ###\n",
		 (unsigned)( cfg->data_rate + 0.5f ), hz);
    else
	snprintf(b, sizeof b, "### CARRIER %.2f @ %.1f Hz ###\n", (double)cfg->data_rate, hz);
    return b;
}

// report_no_carrier -- minimodem.c:253-291 (all arithmetic in float, as there)
std::string nocarrier_line( const mifsk_rx_config *cfg, const mifsk_episode &e )
{
    const float frame_n_bits = (float)cfg->frame_n_bits;
    const float sample_rate = (float)cfg->sample_rate;
    const float nbits = (float)e.nframes * frame_n_bits;
    const float rate = nbits * sample_rate / (float)e.carrier_nsamples;
    char b[256];
    int n = snprintf(b, sizeof b, "\n### NOCARRIER ndata=%u confidence=%.3f ampl=%.3f bps=%.2f",
		     e.nframes, (double)( e.confidence_total / (float)e.nframes ),
		     (double)( e.amplitude_total / (float)e.nframes ), (double)rate);
    std::string s(b, (size_t)n);
    if ( (unsigned long long)( nbits * sample_rate + 0.5f )
	    == (unsigned long long)( cfg->data_rate * (float)e.carrier_nsamples ) ) {
	s += " (rate perfect) ###\n";
    } else {
	const float skew = ( rate - cfg->data_rate ) / cfg->data_rate;
	snprintf(b, sizeof b, " (%.1f%% %s) ###\n", (double)( fabsf(skew) * 100.0f ),
		 std::signbit(skew) ? "slow" : "fast");
	s += b;
    }
    return s;
}

} // namespace

extern "C" int mifsk_stream_text( const mifsk_rx_config *cfg,
	const uint64_t *bits, uint32_t nframes,
	const mifsk_episode *episodes, uint32_t nepisodes, unsigned flags,
	char *out, size_t out_cap, size_t *out_len,
	char *err, size_t err_cap, size_t *err_len )
{
    if ( !cfg || ( nframes && !bits ) || ( nepisodes && !episodes )
	    || ( out_cap && !out ) || ( err_cap && !err ) )
	return -EINVAL;
    mifsk_databits dec;
    dec.decoder = cfg->decoder;
    if ( dec.decoder < MIFSK_DECODE_ASCII8 || dec.decoder > MIFSK_DECODE_UIC_TRAIN )
	return -EINVAL;
    Sink so = { out, out_cap, 0 }, se = { err, err_cap, 0 };
    const bool quiet = ( flags & MIFSK_TEXT_QUIET ) != 0;
    const bool filter = ( flags & MIFSK_TEXT_PRINT_FILTER ) != 0;

    uint32_t e = 0;			// next episode to open
    char buf[4096];			// dataoutbuf, minimodem.c:1430-1431
    for ( uint32_t i = 0; i < nframes; i++ ) {
	while ( e < nepisodes && episodes[e].first_frame == i ) {
	    // carrier acquired here: the previous episode has ended before it
	    if ( e > 0 && !quiet )
		se.add(nocarrier_line(cfg, episodes[e - 1]));
	    if ( !quiet )
		se.add(carrier_line(cfg, episodes[e]));
	    mifsk_databits_reset(&dec);					// minimodem.c:1351
	    e++;
	}
	const unsigned long long b = bits[i];
	if ( cfg->do_rx_sync && b == cfg->sync_byte )			// minimodem.c:1435-1439
	    continue;
	const unsigned n = mifsk_databits_decode(&dec, buf, sizeof buf, b, cfg->n_data_bits);
	if ( !filter ) {
	    so.add(buf, n);
	} else {							// minimodem.c:1454-1460, "C" locale
	    for ( unsigned k = 0; k < n; k++ ) {
		const unsigned char c = (unsigned char)buf[k];
		const bool keep = ( c >= 0x20 && c < 0x7F ) || ( c >= 0x09 && c <= 0x0D );
		const char pc = keep ? (char)c : '.';
		so.add(&pc, 1);
	    }
	}
    }
    if ( !quiet ) {
	// episodes that were opened and not yet closed (and any the frame list
	// was too short to reach: truncated outputs)
	for ( uint32_t k = e ? e - 1 : 0; k < nepisodes; k++ ) {
	    if ( k >= e )
		se.add(carrier_line(cfg, episodes[k]));
	    se.add(nocarrier_line(cfg, episodes[k]));
	}
    }
    if ( out_len ) *out_len = so.len;
    if ( err_len ) *err_len = se.len;
    return ( so.len > so.cap || se.len > se.cap ) ? -ENOSPC : 0;
}
