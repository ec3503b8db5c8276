/*
 * oracle/shim/sndfile_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * RIFF/WAVE (PCM16 / IEEE float32) reader+writer behind the handful of
 * libsndfile calls made by /root/reference/src/simpleaudio-sndfile.c.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "sndfile.h"

struct oracle_sndfile {
    FILE	*fp;
    int		writing;
    int		subformat;	/* SF_FORMAT_PCM_16 or SF_FORMAT_FLOAT */
    int		channels;
    int		samplerate;
    long	data_offset;	/* file offset of the first sample byte */
    uint64_t	data_bytes;	/* reading: size of data chunk; writing: so far */
    uint64_t	read_bytes;
};

static const char *last_error = "No Error.";

static void put_u32(unsigned char *p, uint32_t v)
{ p[0] = v & 0xFF; p[1] = (v >> 8) & 0xFF; p[2] = (v >> 16) & 0xFF; p[3] = (v >> 24) & 0xFF; }
static void put_u16(unsigned char *p, uint16_t v)
{ p[0] = v & 0xFF; p[1] = (v >> 8) & 0xFF; }
static uint32_t get_u32(const unsigned char *p)
{ return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t get_u16(const unsigned char *p)
{ return (uint16_t)(p[0] | (p[1] << 8)); }

static int write_header(struct oracle_sndfile *s)
{
    unsigned char h[44];
    int bytes_per_sample = s->subformat == SF_FORMAT_FLOAT ? 4 : 2;
    uint32_t data = (uint32_t)s->data_bytes;
    memcpy(h, "RIFF", 4);
    put_u32(h + 4, 36 + data);
    memcpy(h + 8, "WAVEfmt ", 8);
    put_u32(h + 16, 16);
    put_u16(h + 20, s->subformat == SF_FORMAT_FLOAT ? 3 : 1);
    put_u16(h + 22, (uint16_t)s->channels);
    put_u32(h + 24, (uint32_t)s->samplerate);
    put_u32(h + 28, (uint32_t)(s->samplerate * s->channels * bytes_per_sample));
    put_u16(h + 32, (uint16_t)(s->channels * bytes_per_sample));
    put_u16(h + 34, (uint16_t)(8 * bytes_per_sample));
    memcpy(h + 36, "data", 4);
    put_u32(h + 40, data);
    if ( fseek(s->fp, 0, SEEK_SET) != 0 )
	return -1;
    if ( fwrite(h, 1, sizeof(h), s->fp) != sizeof(h) )
	return -1;
    return 0;
}

static int parse_header(struct oracle_sndfile *s, SF_INFO *info)
{
    unsigned char b[12];
    if ( fread(b, 1, 12, s->fp) != 12 || memcmp(b, "RIFF", 4) || memcmp(b + 8, "WAVE", 4) ) {
	last_error = "File contains data in an unknown format.";
	return -1;
    }
    int have_fmt = 0;
    unsigned bits = 0, tag = 0;
    for (;;) {
	unsigned char ch[8];
	if ( fread(ch, 1, 8, s->fp) != 8 ) {
	    last_error = "WAV file has no data chunk.";
	    return -1;
	}
	uint32_t len = get_u32(ch + 4);
	if ( !memcmp(ch, "fmt ", 4) ) {
	    unsigned char f[40];
	    uint32_t want = len < sizeof(f) ? len : (uint32_t)sizeof(f);
	    if ( len < 16 || fread(f, 1, want, s->fp) != want ) {
		last_error = "Bad WAV fmt chunk.";
		return -1;
	    }
	    tag = get_u16(f);
	    s->channels = get_u16(f + 2);
	    s->samplerate = (int)get_u32(f + 4);
	    bits = get_u16(f + 14);
	    if ( tag == 0xFFFE && len >= 26 )	/* WAVE_FORMAT_EXTENSIBLE */
		tag = get_u16(f + 24);
	    if ( len > want )
		fseek(s->fp, (long)(len - want), SEEK_CUR);
	    if ( len & 1 )
		fseek(s->fp, 1, SEEK_CUR);
	    have_fmt = 1;
	} else if ( !memcmp(ch, "data", 4) ) {
	    if ( !have_fmt ) {
		last_error = "WAV data chunk before fmt chunk.";
		return -1;
	    }
	    s->data_offset = ftell(s->fp);
	    /* tolerate a bogus length (streamed files): clamp to file size */
	    long cur = s->data_offset;
	    fseek(s->fp, 0, SEEK_END);
	    long end = ftell(s->fp);
	    fseek(s->fp, cur, SEEK_SET);
	    uint64_t avail = (uint64_t)(end - cur);
	    s->data_bytes = len <= avail ? len : avail;
	    break;
	} else {
	    fseek(s->fp, (long)(len + (len & 1)), SEEK_CUR);
	}
    }
    if ( tag == 1 && bits == 16 )
	s->subformat = SF_FORMAT_PCM_16;
    else if ( tag == 3 && bits == 32 )
	s->subformat = SF_FORMAT_FLOAT;
    else {
	last_error = "Unsupported WAV sample encoding (shim handles PCM16 and float32 only).";
	return -1;
    }
    int bps = s->subformat == SF_FORMAT_FLOAT ? 4 : 2;
    info->samplerate = s->samplerate;
    info->channels = s->channels;
    info->format = SF_FORMAT_WAV | s->subformat;
    info->frames = (sf_count_t)(s->data_bytes / (uint64_t)(bps * (s->channels ? s->channels : 1)));
    info->sections = 1;
    info->seekable = 1;
    return 0;
}

SNDFILE *sf_open(const char *path, int mode, SF_INFO *sfinfo)
{
    struct oracle_sndfile *s = calloc(1, sizeof(*s));
    if ( !s ) {
	last_error = "Out of memory.";
	return NULL;
    }
    if ( mode == SFM_WRITE ) {
	int sub = sfinfo->format & SF_FORMAT_SUBMASK;
	if ( sub != SF_FORMAT_PCM_16 && sub != SF_FORMAT_FLOAT ) {
	    last_error = "Format not recognised.";
	    free(s);
	    return NULL;
	}
	s->fp = fopen(path, "wb");
	if ( !s->fp ) {
	    last_error = "System error : cannot open file for writing.";
	    free(s);
	    return NULL;
	}
	s->writing = 1;
	s->subformat = sub;
	s->channels = sfinfo->channels;
	s->samplerate = sfinfo->samplerate;
	s->data_offset = 44;
	if ( write_header(s) != 0 ) {
	    last_error = "System error : write failed.";
	    fclose(s->fp);
	    free(s);
	    return NULL;
	}
	return s;
    }
    s->fp = fopen(path, "rb");
    if ( !s->fp ) {
	last_error = "System error : No such file or directory.";
	free(s);
	return NULL;
    }
    if ( parse_header(s, sfinfo) != 0 ) {
	fclose(s->fp);
	free(s);
	return NULL;
    }
    return s;
}

int sf_close(SNDFILE *s)
{
    if ( !s )
	return -1;
    int rc = 0;
    if ( s->writing )
	rc = write_header(s);
    if ( fclose(s->fp) != 0 )
	rc = -1;
    free(s);
    return rc;
}

int sf_command(SNDFILE *s, int cmd, void *data, int datasize)
{
    (void)s; (void)cmd; (void)data; (void)datasize;
    return 0;	/* the shim never writes a PEAK chunk */
}

const char *sf_strerror(SNDFILE *s) { (void)s; return last_error; }

int sf_perror(SNDFILE *s)
{
    fprintf(stderr, "%s\n", sf_strerror(s));
    return 0;
}

static sf_count_t frames_left(struct oracle_sndfile *s, sf_count_t frames, int bps)
{
    uint64_t left = (s->data_bytes - s->read_bytes) / (uint64_t)(bps * s->channels);
    return (uint64_t)frames < left ? frames : (sf_count_t)left;
}

sf_count_t sf_readf_float(SNDFILE *s, float *ptr, sf_count_t frames)
{
    if ( s->writing || frames <= 0 )
	return 0;
    if ( s->subformat == SF_FORMAT_FLOAT ) {
	sf_count_t n = frames_left(s, frames, 4);
	size_t got = fread(ptr, 4 * (size_t)s->channels, (size_t)n, s->fp);
	s->read_bytes += (uint64_t)got * 4 * (uint64_t)s->channels;
	return (sf_count_t)got;
    }
    sf_count_t n = frames_left(s, frames, 2);
    size_t nsamp = (size_t)n * (size_t)s->channels;
    short *tmp = malloc(sizeof(short) * (nsamp ? nsamp : 1));
    size_t got = fread(tmp, 2 * (size_t)s->channels, (size_t)n, s->fp);
    for ( size_t i = 0; i < got * (size_t)s->channels; i++ )
	ptr[i] = (float)tmp[i] / 32768.0f;	/* libsndfile default normalisation */
    free(tmp);
    s->read_bytes += (uint64_t)got * 2 * (uint64_t)s->channels;
    return (sf_count_t)got;
}

sf_count_t sf_readf_short(SNDFILE *s, short *ptr, sf_count_t frames)
{
    if ( s->writing || frames <= 0 )
	return 0;
    if ( s->subformat == SF_FORMAT_PCM_16 ) {
	sf_count_t n = frames_left(s, frames, 2);
	size_t got = fread(ptr, 2 * (size_t)s->channels, (size_t)n, s->fp);
	s->read_bytes += (uint64_t)got * 2 * (uint64_t)s->channels;
	return (sf_count_t)got;
    }
    sf_count_t n = frames_left(s, frames, 4);
    size_t nsamp = (size_t)n * (size_t)s->channels;
    float *tmp = malloc(sizeof(float) * (nsamp ? nsamp : 1));
    size_t got = fread(tmp, 4 * (size_t)s->channels, (size_t)n, s->fp);
    for ( size_t i = 0; i < got * (size_t)s->channels; i++ ) {
	float v = tmp[i] * 32767.0f;
	if ( v > 32767.0f ) v = 32767.0f;
	if ( v < -32768.0f ) v = -32768.0f;
	ptr[i] = (short)lrintf(v);
    }
    free(tmp);
    s->read_bytes += (uint64_t)got * 4 * (uint64_t)s->channels;
    return (sf_count_t)got;
}

sf_count_t sf_writef_float(SNDFILE *s, const float *ptr, sf_count_t frames)
{
    if ( !s->writing || frames <= 0 )
	return 0;
    size_t nsamp = (size_t)frames * (size_t)s->channels;
    if ( s->subformat == SF_FORMAT_FLOAT ) {
	if ( fwrite(ptr, 4, nsamp, s->fp) != nsamp )
	    return -1;
	s->data_bytes += 4 * (uint64_t)nsamp;
	return frames;
    }
    for ( size_t i = 0; i < nsamp; i++ ) {
	float v = ptr[i] * 32767.0f;
	if ( v > 32767.0f ) v = 32767.0f;
	if ( v < -32768.0f ) v = -32768.0f;
	short q = (short)lrintf(v);
	if ( fwrite(&q, 2, 1, s->fp) != 1 )
	    return -1;
    }
    s->data_bytes += 2 * (uint64_t)nsamp;
    return frames;
}

sf_count_t sf_writef_short(SNDFILE *s, const short *ptr, sf_count_t frames)
{
    if ( !s->writing || frames <= 0 )
	return 0;
    size_t nsamp = (size_t)frames * (size_t)s->channels;
    if ( s->subformat == SF_FORMAT_PCM_16 ) {
	if ( fwrite(ptr, 2, nsamp, s->fp) != nsamp )
	    return -1;
	s->data_bytes += 2 * (uint64_t)nsamp;
	return frames;
    }
    for ( size_t i = 0; i < nsamp; i++ ) {
	float v = (float)ptr[i] / 32768.0f;
	if ( fwrite(&v, 4, 1, s->fp) != 1 )
	    return -1;
    }
    s->data_bytes += 4 * (uint64_t)nsamp;
    return frames;
}
