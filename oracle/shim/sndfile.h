/*
 * oracle/shim/sndfile.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Minimal stand-in for libsndfile (third-party, unpinned, not under
 * /root/reference, not installed here) so that the reference's
 * src/simpleaudio-sndfile.c compiles unmodified into oracle/_ref/.
 * Only RIFF/WAVE with PCM16 or IEEE float32 samples is supported -- the two
 * formats the reference's --file path writes and reads
 * (/root/reference/src/simpleaudio-sndfile.c:170-190).
 *
 * Assumption about the un-vendored library that matters for parity:
 * reading a PCM16 file with sf_readf_float() yields sample/32768.0f
 * (libsndfile's default float normalisation).
 */
#ifndef ORACLE_SHIM_SNDFILE_H
#define ORACLE_SHIM_SNDFILE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t sf_count_t;
typedef struct oracle_sndfile SNDFILE;

typedef struct SF_INFO {
    sf_count_t	frames;
    int		samplerate;
    int		channels;
    int		format;
    int		sections;
    int		seekable;
} SF_INFO;

enum {
    SF_FORMAT_WAV	= 0x010000,
    SF_FORMAT_AIFF	= 0x020000,
    SF_FORMAT_AU	= 0x030000,
    SF_FORMAT_RAW	= 0x040000,
    SF_FORMAT_PAF	= 0x050000,
    SF_FORMAT_SVX	= 0x060000,
    SF_FORMAT_NIST	= 0x070000,
    SF_FORMAT_VOC	= 0x080000,
    SF_FORMAT_IRCAM	= 0x0A0000,
    SF_FORMAT_W64	= 0x0B0000,
    SF_FORMAT_MAT4	= 0x0C0000,
    SF_FORMAT_MAT5	= 0x0D0000,
    SF_FORMAT_PVF	= 0x0E0000,
    SF_FORMAT_XI	= 0x0F0000,
    SF_FORMAT_HTK	= 0x100000,
    SF_FORMAT_SDS	= 0x110000,
    SF_FORMAT_AVR	= 0x120000,
    SF_FORMAT_WAVEX	= 0x130000,
    SF_FORMAT_SD2	= 0x160000,
    SF_FORMAT_FLAC	= 0x170000,
    SF_FORMAT_CAF	= 0x180000,
    SF_FORMAT_WVE	= 0x190000,
    SF_FORMAT_OGG	= 0x200000,
    SF_FORMAT_MPC2K	= 0x210000,
    SF_FORMAT_RF64	= 0x220000,

    SF_FORMAT_PCM_16	= 0x0002,
    SF_FORMAT_FLOAT	= 0x0006,

    SF_FORMAT_SUBMASK	= 0x0000FFFF,
    SF_FORMAT_TYPEMASK	= 0x0FFF0000
};

enum { SFM_READ = 0x10, SFM_WRITE = 0x20 };
enum { SF_FALSE = 0, SF_TRUE = 1 };
enum { SFC_SET_ADD_PEAK_CHUNK = 0x1050 };

SNDFILE    *sf_open(const char *path, int mode, SF_INFO *sfinfo);
int	    sf_close(SNDFILE *s);
int	    sf_command(SNDFILE *s, int cmd, void *data, int datasize);
int	    sf_perror(SNDFILE *s);
const char *sf_strerror(SNDFILE *s);

sf_count_t  sf_readf_float(SNDFILE *s, float *ptr, sf_count_t frames);
sf_count_t  sf_readf_short(SNDFILE *s, short *ptr, sf_count_t frames);
sf_count_t  sf_writef_float(SNDFILE *s, const float *ptr, sf_count_t frames);
sf_count_t  sf_writef_short(SNDFILE *s, const short *ptr, sf_count_t frames);

#ifdef __cplusplus
}
#endif

#endif
