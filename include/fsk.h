/*
 * include/fsk.h -- drop-in C ABI of the MI355X-native FSK demodulator.
 *
 * These are exactly the five entry points (and the one public struct) that
 * the reference exports from src/fsk.h and that its only caller, main() in
 * src/minimodem.c, binds to.  libmifsk.so exports them with the same names,
 * argument meaning, ownership and error behaviour, so the reference's
 * minimodem.c links against it unchanged.
 *
 *   this header                  replaces (reference)          called from
 *   fsk_plan_new                 src/fsk.h:49-55,  fsk.c:33    minimodem.c:1045
 *   fsk_plan_destroy             src/fsk.h:57-58,  fsk.c:97    minimodem.c:1478
 *   fsk_find_frame               src/fsk.h:60-71,  fsk.c:449   minimodem.c:1265,1373
 *   fsk_detect_carrier           src/fsk.h:73-75,  fsk.c:543   minimodem.c:1188
 *   fsk_set_tones_by_bandshift   src/fsk.h:77-78,  fsk.c:584   minimodem.c:1219
 *
 * Every call runs on the GPU (hand-written HIP kernels for gfx950); there is
 * no CPU fallback: fsk_plan_new() fails (NULL, errno = ENODEV) when no HIP
 * device is usable.
 *
 * struct fsk_plan keeps the reference's field names, types and offsets
 * (src/fsk.h:30-46; x86-64: 64 bytes) because main() reads ->fftsize,
 * ->nbands, ->band_width and ->b_mark directly (minimodem.c:1184,1203,1210,
 * 1217,1340,1344).  The three pointer slots that hold FFTW objects in the
 * reference (fftplan / fftin / fftout, private to fsk.c) are reused here as
 * opaque handles owned by the library.
 */
#ifndef MIFSK_FSK_H
#define MIFSK_FSK_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fsk_plan fsk_plan;

struct fsk_plan {
    float		sample_rate;	/* @0  */
    float		f_mark;		/* @4  */
    float		f_space;	/* @8  */
    float		filter_bw;	/* @12  never initialised by the reference either */

    int			fftsize;	/* @16  (int)((sample_rate + bw/2) / bw) */
    unsigned int	nbands;		/* @20  fftsize/2 + 1 */
    float		band_width;	/* @24 */
    unsigned int	b_mark;		/* @28  DFT bin of the mark tone  */
    unsigned int	b_space;	/* @32  DFT bin of the space tone */
    /* 4 bytes padding */
    void		*fftplan;	/* @40  opaque: library context (struct mifsk_ctx *) */
    float		*fftin;		/* @48  opaque: reserved, always NULL */
    void		*fftout;	/* @56  opaque: reserved, always NULL */
};

/*
 * Returns NULL with errno = EINVAL (and a message on stderr) when a tone's
 * bin falls outside the spectrum, exactly as the reference does; NULL with
 * errno = ENOMEM on allocation failure; NULL with errno = ENODEV when no
 * gfx950-class HIP device can be opened.
 */
fsk_plan *
fsk_plan_new(
	float		sample_rate,
	float		f_mark,
	float		f_space,
	float		filter_bw
	);

void
fsk_plan_destroy( fsk_plan *fskp );

/*
 * Search `samples` for the best frame.  `samples` is a caller-owned HOST
 * buffer (mono f32); it is never written.  Exactly the floats the reference
 * can touch are read: up to the highest candidate it may try,
 * try_first + k * try_step < try_max, plus the last bit window
 * (src/fsk.c:204-206,477-484).  `expect_bits_string` is a borrowed
 * NUL-terminated string over {'0','1','d'} of at most 64 characters.
 * Returns the best confidence (0.0 when no candidate matched) and ALWAYS
 * writes *bits_outp, *ampl_outp and *frame_start_outp (zeros when nothing
 * matched).  Same scan order, tie-breaking and early exit as
 * reference src/fsk.c:477-511.
 */
float
fsk_find_frame( fsk_plan *fskp, float *samples, unsigned int frame_nsamples,
	unsigned int try_first_sample,
	unsigned int try_max_nsamples,
	unsigned int try_step_nsamples,
	float try_confidence_search_limit,
	const char *expect_bits_string,
	unsigned long long *bits_outp,
	float *ampl_outp,
	unsigned int *frame_start_outp
	);

/* Returns the index of the strongest band above the threshold, or -1. */
int
fsk_detect_carrier(fsk_plan *fskp, float *samples, unsigned int nsamples,
	float min_mag_threshold );

void
fsk_set_tones_by_bandshift( fsk_plan *fskp, unsigned int b_mark, int b_shift );

#ifdef __cplusplus
}
#endif

#endif /* MIFSK_FSK_H */
