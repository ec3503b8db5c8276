/*
 * include/mifsk.h -- batch C ABI of the MI355X-native FSK demodulator.
 *
 * include/fsk.h keeps the reference's five-function API.  One fsk_find_frame()
 * call covers <= ~10 kB of audio, so offloading it call by call is launch-
 * and PCIe-bound.  This header is the extension that makes the path worth
 * running on a GPU: whole streams resident in HBM, and the reference's
 * receive loop (main() in src/minimodem.c:1137-1463 -- the only caller of
 * fsk_find_frame) executed on the device, one workgroup per stream.
 *
 * Everything here is plain C: PODs, pointers and sizes.  Pointers whose name
 * starts with d_ are DEVICE pointers (hipMalloc / torch.cuda tensors); all
 * others are host pointers.  `stream` arguments are a hipStream_t passed as
 * void* (NULL = the default stream).  Functions return 0 on success or a
 * negative errno value; nothing aborts across this boundary.
 *
 *   entry point                   stands in for (reference file:line)
 *   mifsk_modem_args_default      option defaults        minimodem.c:492-553
 *   mifsk_rx_config_init          preset + derived state minimodem.c:819-965,1037-1131
 *                                 and plan bins          fsk.c:52-57
 *   mifsk_ctx_create/destroy      fsk_plan_new/destroy   fsk.c:33-104
 *   mifsk_find_frame_batch        fsk_find_frame         fsk.c:449-538 (N problems)
 *   mifsk_demod_batch             the --rx main loop     minimodem.c:1137-1463
 *   mifsk_pipeline_*              ... several batches in flight (lanes of context + stream)
 *   mifsk_gather_*                (none: one process per GPU) decoded bytes to rank 0, RCCL
 *   mifsk_demod_slab[_ring]       the loop over a stream that arrives in pieces  minimodem.c:1144-1174
 *   mifsk_session_*               ... fed from host memory, bookkeeping included
 *   mifsk_demod_batch_host[_ex]   same, host buffers     (chunked H2D | demod | D2H, overlapped)
 *   mifsk_demod_files             --rx --file, N files   simpleaudio-sndfile.c:42-74,
 *                                                        minimodem.c:1014-1032
 */
#ifndef MIFSK_H
#define MIFSK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIFSK_MAX_FRAME_BITS	64	/* fsk.c:185-187,463 */
#define MIFSK_ABI_VERSION	8

/* which databits decoder main() would have selected (minimodem.c:549-553,
 * 675,820,856,866,892).  Decoding frame bits to text is O(1)/frame host work
 * on the gathered frames; the device emits frame bits. */
enum mifsk_decoder {
    MIFSK_DECODE_ASCII8	= 0,
    MIFSK_DECODE_BAUDOT	= 1,
    MIFSK_DECODE_BINARY	= 2,
    MIFSK_DECODE_CALLERID = 3,
    MIFSK_DECODE_UIC_GROUND = 4,
    MIFSK_DECODE_UIC_TRAIN = 5
};

/* The --rx command line, as data.  Zero / negative means "not given". */
typedef struct mifsk_modem_args {
    const char	*baudmode;	/* "1200" "300" "rtty" "tdd" "same" "callerid"
				   "uic-train" "uic-ground" "V.21" or a number */
    unsigned	sample_rate;	/* -R            (0 -> 48000)              */
    float	mark_f;		/* -M            (0 -> mode default)       */
    float	space_f;	/* -S            (0 -> mode default)       */
    float	band_width;	/* -b            (0 -> mode default)       */
    int		n_data_bits;	/* -8 / -7 / -5  (0 -> mode default)       */
    int		baudot;		/* -5 also selects the baudot decoder      */
    int		nstartbits;	/* --startbits   (<0 -> 1)                 */
    float	nstopbits;	/* --stopbits    (<0 -> 1.0)               */
    int		invert_start_stop;	/* --invert-start-stop             */
    int		inverted_freqs;	/* -i                                      */
    int		msb_first;	/* --msb-first                             */
    int		have_sync_byte;	/* --sync-byte given                       */
    long long	sync_byte;
    float	confidence_threshold;	/* -c (<0 -> 1.5)                  */
    float	search_limit;	/* -l            (<0 -> 2.3)               */
    int		binary_output;	/* --binary-output                         */
    int		binary_raw_nbits;	/* --binary-raw N                  */
    int		rx_one;		/* --rx-one                                */
    float	auto_carrier_threshold;	/* -a -> 0.001, -A x -> x: find the mark
					   tone per stream before the first
					   search (minimodem.c:1179-1220)  */
} mifsk_modem_args;

/* Everything main() derives before entering the receive loop, computed on the
 * host in C `float` with the reference's own expressions and truncations. */
typedef struct mifsk_rx_config {
    /* modem */
    unsigned	sample_rate;
    float	data_rate;
    float	mark_f, space_f, band_width;
    unsigned	n_data_bits;
    int		nstartbits;
    float	nstopbits;
    int		invert_start_stop;
    int		msb_first;
    int		do_rx_sync;
    unsigned long long sync_byte;
    int		decoder;		/* enum mifsk_decoder */
    int		rx_one;
    float	confidence_threshold;
    float	search_limit;
    float	auto_carrier_threshold;
    int		autodetect_shift;
    int		inverted_freqs;

    /* plan (fsk.c:52-57) */
    int		fftsize;
    unsigned	nbands;
    unsigned	b_mark, b_space;

    /* framing (minimodem.c:943,1037,1105-1131) */
    unsigned	frame_n_bits;		/* unsigned = n_data + nstart + nstop(float) */
    float	nsamples_per_bit;
    unsigned	nsamples_overscan;
    unsigned	frame_nsamples;
    unsigned	expect_n_bits;
    unsigned	expect_nsamples;
    char	expect_data[MIFSK_MAX_FRAME_BITS + 4];
    char	expect_sync[MIFSK_MAX_FRAME_BITS + 4];
    unsigned	samplebuf_size;		/* minimodem.c:1063-1070 */

    /* search grid; index 0 = no carrier, 1 = carrier (minimodem.c:1236-1263,1366) */
    unsigned	try_first[2];
    unsigned	try_max[2];
    unsigned	try_step[2];
    unsigned	try_step_fine[2];

    /* bit windows inside fsk_find_frame (fsk.c:183,204,465) */
    float	find_samples_per_bit;	/* (float)expect_nsamples / expect_n_bits */
    unsigned	bit_nsamples;
    unsigned	bit_offset[MIFSK_MAX_FRAME_BITS];
} mifsk_rx_config;

void mifsk_modem_args_default( mifsk_modem_args *args );
int  mifsk_rx_config_init( mifsk_rx_config *cfg, const mifsk_modem_args *args );

/* upper bound on frames one stream of nsamples can yield */
size_t mifsk_max_frames( const mifsk_rx_config *cfg, size_t nsamples );
/* How far past a stream's last sample a search may logically look (last
 * candidate position + last bit window, fsk.c:204-206,480).  Informational:
 * those samples read as 0.0 whatever the memory holds, so no padding is
 * required.  (The kernels may load -- and ignore -- floats between a row's
 * nsamples[s] and the end of the batch: all nstreams * stream_stride floats
 * must be readable, nothing beyond them is touched.  The reference reads
 * stale ring-buffer memory there -- DESIGN.md "past-the-end reads".) */
size_t mifsk_stream_padding( const mifsk_rx_config *cfg );

/* ---- device context --------------------------------------------------- */

typedef struct mifsk_ctx mifsk_ctx;

/* device < 0: use the current HIP device.  -ENODEV when none is usable. */
int  mifsk_ctx_create( mifsk_ctx **ctx_out, int device );
void mifsk_ctx_destroy( mifsk_ctx *ctx );
const char *mifsk_ctx_device_name( const mifsk_ctx *ctx );
int  mifsk_abi_version( void );
/* sizeof() of the public struct `name` ("mifsk_demod_io", ...) as this library was built, 0
 * for a name it does not know: lets a foreign-function binding check its mirror of the
 * layouts before it passes one across (tests/test_config.py does, for the ctypes mirror). */
size_t mifsk_abi_sizeof( const char *name );

/* Diagnostic: the kernels take (float)sqrt(fr^2 + fi^2) of a bit window's sums (the reference's
 * hypotf, fsk.c:107-114) by a short sequence wherever that is provably the float the correctly
 * rounded one gives, and by the exact sequence otherwise (csrc/mifsk_devmath.h).  This runs both
 * on `nvalues` sums of squares (half of them placed at float rounding boundaries) and counts:
 * [0] short results accepted as safe that differ from the exact ones -- must be 0, [1] values
 * the guard sent to the exact sequence, [2] values whose unguarded short result differs,
 * [3] values evaluated.  Synchronous. */
int mifsk_selftest_sqrt( mifsk_ctx *ctx, uint64_t seed, uint64_t nvalues, uint64_t counts[4] );

/* ---- N independent fsk_find_frame() problems --------------------------- */

typedef struct mifsk_search {
    uint64_t	sample_offset;	/* window start, in floats, into d_samples  */
    uint32_t	navail;		/* readable floats from there; beyond = 0.0 */
    uint32_t	try_first;
    uint32_t	try_max;
    uint32_t	try_step;
    float	search_limit;
    uint32_t	use_sync_string;	/* 0: cfg->expect_data, 1: cfg->expect_sync */
} mifsk_search;

typedef struct mifsk_search_result {
    uint64_t	bits;
    float	confidence;
    float	amplitude;
    uint32_t	frame_start;
    uint32_t	n_positions;	/* positions the reference would have analysed */
} mifsk_search_result;

int mifsk_find_frame_batch( mifsk_ctx *ctx, const mifsk_rx_config *cfg,
	const float *d_samples,
	const mifsk_search *d_problems, mifsk_search_result *d_results,
	int nproblems, void *stream );

/* ---- whole-stream receive loop ----------------------------------------- */

/* one decoded frame, in loop order (optional detailed output) */
typedef struct mifsk_frame {
    uint64_t	bits;		/* data bits handed to the databits decoder
				   (after >>1, bit_window, bit_reverse)     */
    uint64_t	start;		/* absolute sample index of the frame start */
    float	confidence;
    float	amplitude;
    uint32_t	flags;		/* MIFSK_FRAME_* */
    uint32_t	reserved;
} mifsk_frame;

#define MIFSK_FRAME_ACQUIRE	1u	/* carrier was acquired on this frame   */
#define MIFSK_FRAME_SYNC	2u	/* == sync byte: suppressed from output */
#define MIFSK_FRAME_REFINED	4u	/* a fine rescan was run for this frame */

/* one carrier episode = what "### CARRIER" ... "### NOCARRIER" brackets */
typedef struct mifsk_episode {
    uint64_t	carrier_nsamples;
    uint32_t	first_frame;	/* index of its first frame in loop order   */
    uint32_t	nframes;	/* nframes_decoded                          */
    float	confidence_total;
    float	amplitude_total;
    uint32_t	end_reason;	/* 1: carrier lost, 2: end of stream        */
    uint32_t	b_mark;		/* the plan's mark band when the carrier was
				   acquired ("### CARRIER ... @ f Hz" is
				   b_mark * band_width; with --auto-carrier it
				   can change from episode to episode)      */
} mifsk_episode;

#define MIFSK_STREAM_FRAMES_TRUNCATED	1u
#define MIFSK_STREAM_EPISODES_TRUNCATED	2u
#define MIFSK_STREAM_ABORTED		4u	/* internal error: the loop was cut short */

typedef struct mifsk_demod_io {
    /* inputs */
    const float		*d_samples;	/* nstreams rows, stream-major       */
    size_t		stream_stride;	/* floats between rows (16 B aligned)*/
    const uint32_t	*d_nsamples;	/* per stream; NULL -> all = nsamples*/
    uint32_t		nsamples;
    int			nstreams;
    /* outputs (any of bytes/bits/frames may be NULL) */
    uint8_t		*d_bytes;	/* [nstreams][frames_cap]: low 8 data
					   bits of every unsuppressed frame  */
    uint32_t		*d_nbytes;	/* [nstreams]                        */
    uint64_t		*d_bits;	/* [nstreams][frames_cap] data bits of
					   every frame incl. suppressed ones */
    mifsk_frame		*d_frames;	/* [nstreams][frames_cap]            */
    uint32_t		*d_nframes;	/* [nstreams] frames in loop order   */
    size_t		frames_cap;
    mifsk_episode	*d_episodes;	/* [nstreams][episodes_cap]          */
    uint32_t		*d_nepisodes;	/* [nstreams]                        */
    size_t		episodes_cap;
    uint32_t		*d_status;	/* [nstreams] MIFSK_STREAM_* or NULL */
    uint64_t		*d_counters;	/* [nstreams][MIFSK_NCOUNTERS] work
					   counters (MIFSK_CNT_*) or NULL    */
    int32_t		*d_carrier_band;/* [nstreams] or NULL; --auto-carrier
					   only: the band the mark tone was
					   FIRST detected in, -1 = never (the
					   stream then yields nothing); later
					   re-detections (minimodem.c:1297)
					   show in mifsk_episode.b_mark      */
    uint32_t		flags;		/* MIFSK_IO_*                        */
    uint32_t		reserved;	/* 0                                 */
} mifsk_demod_io;

/* Buffer addressing of a search that reads past samples_nvalid (minimodem.c:
 * 1153,1229; fsk.c:204-206).  Default ("flat"): it sees the stream itself and
 * 0.0 beyond the stream's end.  MIFSK_IO_RING_EXACT: it sees what the
 * reference's samplebuf holds there -- the stale cells memmove left behind,
 * 0.0 where nothing was ever written (the reference: uninitialised heap) -- the
 * reference's buffer is kept cell for cell in device memory and every frame
 * goes through the general path: exact, and several times slower. */
#define MIFSK_IO_RING_EXACT	1u
/* Run the receive loop with one workgroup per stream (192 or 256 threads: master wave +
 * worker waves; the round-1 kernel) instead of one wavefront per stream.
 * Flat addressing only; --auto-carrier looks for the tone once per stream. */
#define MIFSK_IO_ENGINE_WORKGROUP 2u
/* ... or with one wavefront per stream whatever the mode.  With neither flag the
 * library chooses (the workgroup engine in the modes whose bit windows it
 * stages through LDS and that have at least 16 samples per bit). */
#define MIFSK_IO_ENGINE_WAVE	4u

/* per-stream work counters (diagnostics; cycle counts are s_memtime ticks) */
#define MIFSK_NCOUNTERS		32
#define MIFSK_CNT_ITERATIONS	0	/* passes through the general loop body  */
#define MIFSK_CNT_BATCHES	1	/* candidate batches evaluated           */
#define MIFSK_CNT_STAGES	2	/* LDS slab (re)loads                    */
#define MIFSK_CNT_BULK_FRAMES	3	/* frames accepted from the run-ahead    */
#define MIFSK_CNT_REFINES	4	/* fine rescans                          */
#define MIFSK_CNT_CACHE_HITS	5	/* searches answered from the cache      */
#define MIFSK_CNT_POSITIONS	6	/* candidate positions evaluated (SCAN)  */
#define MIFSK_CNT_LATTICE_BATCHES 7	/* pipelined lattice batches scored      */
#define MIFSK_CNT_CYC_TOTAL	8
#define MIFSK_CNT_CYC_PARALLEL	9	/* SCAN: stage + correlate + barriers    */
#define MIFSK_CNT_CYC_WAIT	10	/* LATTICE: master waiting for workers   */
#define MIFSK_CNT_CYC_CONFIDENCE 11
#define MIFSK_CNT_CYC_BULK	12
/* (13 .. 23: cycle totals of the profile build; 20 .. 22 double as event counts of the
 * shared-segment scans and of --auto-carrier) */
#define MIFSK_CNT_CONF_FALLBACKS 24	/* confidence passes that took the divisions proper (a
					   zero / non-finite class mean or a subnormal quotient
					   somewhere in the wave: frame_confidence_fixed).  A LOWER
					   bound: lane 0 does the counting, and a pass whose lane 0
					   had left already (its candidate's required bits mismatch)
					   is not counted -- counting it through a ballot moved the
					   register allocation of the loop kernels and cost 1-2 %
					   (profiles/r06_history.md)                               */
#define MIFSK_CNT_SEG_SECOND_LOOKS 25	/* shared-segment scans in which the second look (the running
					   error bound, csrc/mifsk_wave.hip) settled a window the
					   a-priori bound had not                                  */

/* Asynchronous on `stream`: the outputs are complete when `stream` reaches the point
 * behind the call.  (A large wavefront-engine batch is run as several launches on
 * streams of the context's own, forked from and joined back into `stream` with events --
 * mifsk_launch_info.chain_groups below; nothing changes for the caller.)  Calls on ONE
 * context are ordered by the caller (a context owns its launch scratch); to keep several
 * batches in flight -- a launch ends well after its mean stream, so the next one fills the
 * chip -- use a mifsk_pipeline (below), which gives each pass a context and a stream. */
int mifsk_demod_batch( mifsk_ctx *ctx, const mifsk_rx_config *cfg,
	const mifsk_demod_io *io, void *stream );

/* ---- several batches in flight ---------------------------------------------- */

/* A launch ends well after its mean stream (the streams of a batch are serial chains of unequal
 * length), so the way to keep the chip full is to have the next batch running under the tail of
 * this one.  A pipeline owns what that needs: `depth` lanes -- each a context, a HIP stream and a
 * completion event of its own -- and, if asked, `depth` sets of output arrays.  Pass number t
 * (its "ticket", counting from 0) runs on lane t % depth, behind whatever ran on that lane
 * before.  For the reference's call site -- one file after another through the batch entry,
 * src/minimodem.c:1265,1373 via integration/minimodem-rx-batch.patch -- this is "decode the
 * next batch of files while the last streams of this one finish".
 *
 * HIP runs a process's streams on GPU_MAX_HW_QUEUES hardware queues (4 unless the environment
 * said otherwise when the runtime started) and the null stream takes one: lanes beyond that
 * would share a queue and run one behind the other, which measures slower than fewer lanes.
 * The depth is therefore clamped to the queues there are; mifsk_pipeline_info_get says what
 * was asked for, what is in effect and how many queues the runtime has.  (Export
 * GPU_MAX_HW_QUEUES=8 before the first HIP call for a depth above 3.)
 *
 * Results are identical to mifsk_demod_batch's, pass for pass: a lane is an ordinary context
 * and stream.  The calls are serialised per pipeline; submit never blocks on the device. */
typedef struct mifsk_pipeline mifsk_pipeline;
#define MIFSK_PIPELINE_MAX_DEPTH	8
#define MIFSK_PIPELINE_NO_PRODUCER	((void *)(intptr_t)-1)

typedef struct mifsk_pipeline_info {
    uint32_t	depth_requested;
    uint32_t	depth;		/* lanes in effect: min(requested, hw_queues - 1), at least 1 */
    uint32_t	hw_queues;	/* GPU_MAX_HW_QUEUES as this process's environment has it (4: unset) */
    uint32_t	output_sets;	/* `depth` once mifsk_pipeline_outputs_alloc has run, else 0 */
} mifsk_pipeline_info;

#define MIFSK_WANT_BYTES	1u
#define MIFSK_WANT_BITS		2u
#define MIFSK_WANT_FRAMES	4u
#define MIFSK_WANT_EPISODES	8u

/* device < 0: the current HIP device.  -ENODEV without a usable device. */
int  mifsk_pipeline_create( mifsk_pipeline **out, int device, int depth );
void mifsk_pipeline_destroy( mifsk_pipeline *p );	/* waits for what is in flight */
int  mifsk_pipeline_info_get( const mifsk_pipeline *p, mifsk_pipeline_info *info );

/* One set of output arrays per lane, for batches of up to `nstreams` streams: counts and status
 * always, the arrays `want` names (MIFSK_WANT_*).  Replaces sets made before (after waiting for
 * what is in flight).  mifsk_pipeline_outputs_get fills the output fields of *io (pointers and
 * capacities, nothing else) with the set pass `ticket` writes or wrote. */
int  mifsk_pipeline_outputs_alloc( mifsk_pipeline *p, int nstreams, size_t frames_cap,
	size_t episodes_cap, unsigned want );
int  mifsk_pipeline_outputs_get( mifsk_pipeline *p, uint64_t ticket, mifsk_demod_io *io );

/* mifsk_demod_batch(cfg, io) as the pipeline's next pass; *ticket (may be NULL) receives its
 * number (= mifsk_pipeline_next_ticket before the call).  `after`: the hipStream_t the batch
 * was produced on -- the lane waits for the point that stream has reached now (NULL is the
 * null stream) -- or MIFSK_PIPELINE_NO_PRODUCER.  An io without any output pointer is given the
 * lane's own set (mifsk_pipeline_outputs_alloc).  The set a pass writes is the set pass
 * ticket - depth wrote: whatever reads that one must be ordered before this submit -- on the
 * lane's stream (mifsk_pipeline_stream: what a gather of the results is queued on), or by a
 * mifsk_pipeline_wait. */
int  mifsk_pipeline_submit( mifsk_pipeline *p, const mifsk_rx_config *cfg,
	const mifsk_demod_io *io, void *after, uint64_t *ticket );
uint64_t mifsk_pipeline_next_ticket( const mifsk_pipeline *p );
/* the host waits until pass `ticket` is complete (and with it everything submitted on its lane
 * up to now); -EINVAL for a ticket not yet issued */
int  mifsk_pipeline_wait( mifsk_pipeline *p, uint64_t ticket );
/* ... or `stream` does, on the device */
int  mifsk_pipeline_join( mifsk_pipeline *p, uint64_t ticket, void *stream );
/* everything submitted so far */
int  mifsk_pipeline_drain( mifsk_pipeline *p );
/* the lane of pass `ticket`: its hipStream_t and its context */
void *mifsk_pipeline_stream( mifsk_pipeline *p, uint64_t ticket );
mifsk_ctx *mifsk_pipeline_ctx( mifsk_pipeline *p, uint64_t ticket );

/* ---- one process per GPU: decoded bytes to one rank -------------------------- */

/* Streams are independent: rank r of `world` demodulates mifsk_shard_range(nstreams, r, world)
 * and nothing is exchanged on the way.  Where ONE rank must end up holding every stream's bytes
 * (a job launched as one process per GPU; the reference is one process and has no counterpart,
 * src/minimodem.c:1014-1032), the results are gathered to rank 0 over RCCL: every peer sends on
 * its own xGMI link, the root posts all its receives as one group.  RCCL is opened at run time
 * (dlopen of librccl.so.1, the copy the process already has if there is one) and only by these
 * calls: -ENOSYS when it cannot be.
 *
 * Rendezvous: rank 0 makes an id (mifsk_gather_unique_id) and the job's launcher carries its
 * MIFSK_GATHER_ID_BYTES bytes to every rank (MPI, a socket, torch.distributed, a file); every
 * rank then calls mifsk_gather_create -- a collective call, like ncclCommInitRank.
 *
 * mifsk_gather_start enqueues ONE gather on `stream` behind whatever produced the arrays there
 * (the lane's stream of a pipeline pass: mifsk_pipeline_stream) and returns its ticket; it is
 * complete when `stream` reaches the point behind the call.  Only the first `cols` columns of
 * every [row_pitch]-wide row travel (a stream's bytes cannot exceed mifsk_max_frames of its
 * length), through a dense staging copy made on `stream`.  rows[world]: streams per rank (NULL:
 * every rank holds `nstreams`).  The root keeps `slots` receive sets and the k-th gather fills
 * set k % slots, the senders as many staging copies: with up to `slots` gathers in flight none
 * overwrites what an older one delivered.  mifsk_gather_received (rank 0) names what peer
 * `peer` >= 1 sent in gather `ticket`: rows x cols bytes, dense, and the counts.
 *
 * MIFSK_GATHER_LOOPBACK (world == 1 only): the one rank sends to ITSELF through the same calls
 * and receives it as peer 0 -- the whole transport on one GPU (tests/test_gpu_gather.py).
 * Without it a world of one makes no communicator and start() has nothing to do. */
typedef struct mifsk_gather mifsk_gather;
#define MIFSK_GATHER_ID_BYTES	128
#define MIFSK_GATHER_LOOPBACK	1u

typedef struct mifsk_gather_info {
    int		rank, world, device;
    uint32_t	slots;
    uint32_t	loopback;
    uint32_t	communicator;	/* 1: an RCCL communicator was made */
} mifsk_gather_info;

int  mifsk_gather_unique_id( void *id /* [MIFSK_GATHER_ID_BYTES] */ );
/* device < 0: the current HIP device; slots clamped to 2 .. 16 */
int  mifsk_gather_create( mifsk_gather **out, const void *id, int rank, int world, int device,
	int slots, unsigned flags );
void mifsk_gather_destroy( mifsk_gather *g );	/* synchronises the device first */
int  mifsk_gather_info_get( const mifsk_gather *g, mifsk_gather_info *info );
int  mifsk_gather_start( mifsk_gather *g, const uint8_t *d_bytes, size_t row_pitch,
	const int32_t *d_nbytes, int nstreams, int cols, const int *rows, void *stream,
	uint64_t *ticket );
int  mifsk_gather_received( mifsk_gather *g, uint64_t ticket, int peer,
	const uint8_t **d_bytes, const int32_t **d_nbytes, int *rows, int *cols );

/* What mifsk_demod_batch would launch for `cfg`, a batch of `nstreams` streams and
 * `flags` (MIFSK_IO_*): the kernel instantiation, its launch geometry and the
 * occupancy its LDS allows -- what rocprofv3 does not report for dynamic LDS. */
typedef struct mifsk_launch_info {
    char	kernel[64];
    uint32_t	engine;			/* MIFSK_IO_ENGINE_WAVE or _WORKGROUP   */
    uint32_t	workgroup_size;		/* threads (64: a workgroup is a wave)  */
    uint32_t	lds_bytes_per_workgroup;/* dynamic LDS                          */
    uint32_t	workgroups_per_cu;	/* min(160 KiB / LDS, 32 waves / size): VGPRs
					   may lower it (4 waves per SIMD at 128) */
    uint32_t	lattice_mode;		/* 0 none, 1 LDS-staged rounds, 2 streamed */
    uint32_t	frames_per_block;	/* LATTICE frames scored at once, at most */
    uint32_t	compute_units;
    /* Chained launches: a flat-addressed wavefront-engine batch of more streams than the
     * chip holds at once is cut into chain_groups groups of streams x chain_chunks time
     * chunks, every (group, chunk) one launch of the resumable kernel, each group's chunks
     * in order on a HIP stream of the context's -- so that the slots one group's
     * stragglers leave are filled with another group's next chunk instead of staying
     * empty until the round ends.  Same frames, bit for bit (mifsk_demod_slab's
     * guarantee).  0 / 0: one launch. */
    uint32_t	chain_groups, chain_chunks;
} mifsk_launch_info;

int mifsk_demod_plan( mifsk_ctx *ctx, const mifsk_rx_config *cfg, int nstreams,
	unsigned flags, mifsk_launch_info *out );
/* ... for rows of `nsamples` samples (mifsk_demod_io.nsamples): whether a batch is cut in
 * time depends on how long its streams are (mifsk_demod_plan assumes long ones) */
int mifsk_demod_plan_ex( mifsk_ctx *ctx, const mifsk_rx_config *cfg, int nstreams,
	uint32_t nsamples, unsigned flags, mifsk_launch_info *out );

/* ---- streams that start in host memory (SURVEY 8 d "H2D-inclusive") -------- */

/* How a zig-zag scan with long bit windows is evaluated (diagnostic; DESIGN.md "shared
 * segments"): kind = 2 * fine + carrier.  valid == 0: every window is correlated by
 * itself.  Otherwise the span the scan's windows cover is cut into nseg segments, each
 * summed once in one of npass passes of lanes (lock-step length pass_len[p]); window w
 * (= candidate in scan order * expect_n_bits + bit) is the sum of segments
 * win_first[w] .. win_first[w] + win_count[w] - 1, rotated. */
typedef struct mifsk_scan_plan {
    uint32_t	valid, nseg, npass, nwin, span_hi;
    uint32_t	pass_len[2], pass_min[2];
    float	bound_c;
    uint32_t	seg_rel[128];
    uint16_t	seg_len[128], slot_seg[128], win_first[128], win_count[128];
    uint32_t	p_slot[128], p_win[128];	/* the same, packed as the kernel reads it */
    uint8_t	p_slot_seg[128];
} mifsk_scan_plan;

int mifsk_scan_plan_get( const mifsk_rx_config *cfg, int kind, mifsk_scan_plan *out );

/* Same, for HOST pointers in `io` (all fields, d_ prefix notwithstanding).  The
 * batch is cut into chunks of whole streams (~64 MB of input); chunk k+1 crosses
 * PCIe on a copy stream while chunk k is demodulated and chunk k-1's results are
 * copied back, so the call runs at the speed of the bus.  Input rows in
 * page-locked memory (mifsk_host_alloc, hipHostMalloc, hipHostRegister) are
 * copied by DMA from where they are; anything else is first moved into the
 * context's own pinned staging buffers by worker threads.  Synchronous: results
 * are in place on return.  One host call at a time per context (calls on one
 * context serialise; use one context per device and per concurrent caller). */
int mifsk_demod_batch_host( mifsk_ctx *ctx, const mifsk_rx_config *cfg,
	const mifsk_demod_io *io );

/* `io->d_samples` points to int16_t rows (`stream_stride` counts int16 elements):
 * PCM16 as a WAV file holds it.  It crosses the bus as 16-bit and is converted on
 * the device the way libsndfile converts it for the reference (value / 32768,
 * simpleaudio-sndfile.c:42-56).  Host entry points only. */
#define MIFSK_IO_HOST_S16	0x100u

typedef struct mifsk_host_stats {
    double	seconds_total;		/* wall time of the call                          */
    double	seconds_staging;	/* of which: worker threads filling pinned memory */
    uint64_t	bytes_h2d, bytes_d2h;
    uint32_t	chunks, streams;
    uint32_t	source_pinned;		/* 1: input rows were DMA'd from the caller's memory */
    uint32_t	reserved;
} mifsk_host_stats;

/* ... with the --Xrxnoise term (0 = off; see mifsk_ingest_s16) applied on the
 * device, and what the pipeline did reported in *stats (may be NULL). */
int mifsk_demod_batch_host_ex( mifsk_ctx *ctx, const mifsk_rx_config *cfg,
	const mifsk_demod_io *io, float rxnoise, mifsk_host_stats *stats );

/* page-locked host memory for the inputs of the host entry points */
void *mifsk_host_alloc( size_t bytes );
void  mifsk_host_free( void *p );

/* upper bound on carrier episodes one stream of nsamples can yield */
size_t mifsk_max_episodes( const mifsk_rx_config *cfg, size_t nsamples );

/* ---- streams that arrive in pieces (minimodem.c:1144-1174; SURVEY 8 e) ------- */

/* The receive loop's state between two calls, one per stream, in DEVICE memory;
 * all zero = a stream that has not started.  Everything the reference carries
 * from one pass of its loop to the next (minimodem.c:1079-1088,1132-1133,
 * 1144-1174): the buffer arithmetic, the carrier episode's running totals, the
 * tracker's amplitude and peak confidence, the --auto-carrier band. */
typedef struct mifsk_stream_state {
    uint64_t	base;		/* stream index of samplebuf[0]: the caller may drop
				   every sample before it                         */
    uint64_t	rp;		/* the reference's file position                  */
    uint64_t	carrier_nsamples;
    uint64_t	nframes_total;	/* frames emitted by all calls so far             */
    uint32_t	advance;
    uint32_t	flags;		/* MIFSK_STATE_*                                  */
    float	confidence_total, amplitude_total;
    uint32_t	nframes_decoded;
    uint32_t	noconfidence;
    float	track_amplitude, peak_confidence;
    int32_t	carrier_band, first_band;
    uint32_t	b_mark, ep_b_mark;
    uint32_t	ep_first;
    uint32_t	nbytes_total;	/* bytes, episodes emitted by all calls so far    */
    uint32_t	nepisodes_total;
    uint32_t	status;		/* MIFSK_STREAM_* bits of all calls so far        */
} mifsk_stream_state;

#define MIFSK_STATE_STARTED	1u
#define MIFSK_STATE_CARRIER	2u
#define MIFSK_STATE_FINISHED	4u	/* the final slab has been processed */

/* mifsk_demod_batch for streams that are longer than what is resident, or that
 * arrive in pieces.  Row s of io->d_samples holds the samples of stream s from
 * stream index d_origin[s] on (NULL: every row starts at index 0 -- only right
 * for the first slab), io->d_nsamples[s] of them; d_origin[s] must not exceed
 * d_state[s].base, i.e. the caller keeps what the loop has not passed yet and
 * appends the new samples behind it.  With final == 0 the loop stops, and saves
 * its state, at the first pass that could read beyond the row (it needs a whole
 * samplebuf -- cfg->samplebuf_size samples -- beyond the cursor to be sure of
 * seeing exactly what one call over the whole stream sees); with final != 0 the
 * row's end is the stream's end.  Per call the outputs start at index 0 of their
 * arrays (frames in loop order, episodes as they END); mifsk_frame.start and
 * mifsk_episode.first_frame count from the start of the stream.  Any cut of a
 * stream into slabs gives the frames and episodes of the single call, bit for
 * bit.  Flat addressing; io->flags may force an engine (MIFSK_IO_ENGINE_WAVE /
 * _WORKGROUP), otherwise the library chooses as mifsk_demod_batch does -- the
 * state record is the same for both, a stream may even change engines between
 * slabs.  Asynchronous on `stream`. */
int mifsk_demod_slab( mifsk_ctx *ctx, const mifsk_rx_config *cfg, const mifsk_demod_io *io,
	mifsk_stream_state *d_state, const uint64_t *d_origin, int final, void *stream );

/* The same with the reference's buffer semantics (MIFSK_IO_RING_EXACT): what a search
 * reads behind samples_nvalid is what the reference's samplebuf holds there -- stale
 * samples that memmove left behind (minimodem.c:1150-1156) -- also across calls.
 * d_ring is [nstreams][mifsk_ring_floats(cfg)] floats of device memory that the
 * caller zeroes before a stream's first slab and keeps with d_state; a pass of the
 * loop needs only the refill the reference's fread would deliver (half a samplebuf)
 * to be in the row, not a whole samplebuf beyond the cursor.  Any cut gives the
 * one-shot MIFSK_IO_RING_EXACT result, i.e. `minimodem --rx --file`'s, digit for
 * digit.  Wavefront engine. */
size_t mifsk_ring_floats( const mifsk_rx_config *cfg );
int mifsk_demod_slab_ring( mifsk_ctx *ctx, const mifsk_rx_config *cfg, const mifsk_demod_io *io,
	mifsk_stream_state *d_state, const uint64_t *d_origin, float *d_ring, int final,
	void *stream );

/* ---- streams fed in pieces from host memory ---------------------------------- */

/* mifsk_demod_slab with the bookkeeping done (reference: the loop that reads its stream half a
 * samplebuf at a time, src/minimodem.c:1144-1174 -- a recording longer than memory, live audio).
 * A session holds `nstreams` streams: per stream the samples the loop has not passed yet (host
 * memory), where they start in the stream, the loop state and (RING) the samplebuf cells on the
 * device, and arrays sized for every feed.  mifsk_session_feed takes each stream's NEW samples
 * -- samples[s] / nsamples[s], any amount, none (NULL / 0) included -- runs the loop as far as
 * the data allows (`final`: no more will come, the rows' ends are the streams' ends) and waits
 * for the results: mifsk_session_get(s, i) is what THIS feed made of stream i -- frames in loop
 * order, episodes as they end, frame starts and episode frame indices counted from the start of
 * the stream -- in host memory the session owns, valid until the next feed.  Any cut gives,
 * concatenated, the results of one call over the whole streams, bit for bit.
 * flags: MIFSK_IO_RING_EXACT (the reference's buffer semantics across the feeds: `minimodem
 * --rx --file`'s frames digit for digit), MIFSK_IO_ENGINE_*, MIFSK_SESSION_WANT_FRAMES.
 * One feed at a time per session; sessions on one context are ordered by the caller. */
typedef struct mifsk_session mifsk_session;
#define MIFSK_SESSION_WANT_FRAMES	0x1000u	/* the per-frame records too */

typedef struct mifsk_session_result {
    uint32_t		nframes, nbytes, nepisodes;	/* of this feed */
    uint32_t		status;		/* MIFSK_STREAM_* of this feed                 */
    int32_t		carrier_band;	/* --auto-carrier (see mifsk_demod_io)         */
    uint32_t		finished;	/* 1: the final piece has been decoded         */
    const uint64_t	*bits;		/* [nframes] data bits of every frame          */
    const uint8_t	*bytes;		/* [nbytes]                                    */
    const mifsk_frame	*frames;	/* [nframes], or NULL without WANT_FRAMES      */
    const mifsk_episode	*episodes;	/* [nepisodes]                                 */
    uint64_t		consumed;	/* stream index the loop has passed for good   */
} mifsk_session_result;

int  mifsk_session_create( mifsk_session **out, mifsk_ctx *ctx, const mifsk_rx_config *cfg,
	int nstreams, unsigned flags );
void mifsk_session_destroy( mifsk_session *s );
int  mifsk_session_feed( mifsk_session *s, const float *const *samples, const uint32_t *nsamples,
	int final );
const mifsk_session_result *mifsk_session_get( const mifsk_session *s, int stream );
/* samples of stream `stream` the session still holds (fed, not yet passed by the loop) */
size_t mifsk_session_pending( const mifsk_session *s, int stream );

/* ---- several GPUs (SURVEY 8 e) -------------------------------------------- */

/* Streams are independent: device k of `world` owns the contiguous, balanced
 * range [*lo, *hi) of the batch.  The same arithmetic shards a multi-process
 * job (one process per GPU; the decoded bytes are then gathered over RCCL). */
void mifsk_shard_range( int nstreams, int rank, int world, int *lo, int *hi );

/* mifsk_demod_batch_host over several devices of ONE process: ctxs[k] (one
 * context per GPU, mifsk_ctx_create(&c, k)) takes range k of the batch; inputs
 * are copied to, and every output array is filled from, the device that owns the
 * stream -- no collective, the gather is the copy back.  Returns 0 or the first
 * failing device's error. */
int mifsk_demod_batch_host_multi( mifsk_ctx *const *ctxs, int nctx,
	const mifsk_rx_config *cfg, const mifsk_demod_io *io );

/* ---- host post-pass: frame bits -> text -------------------- */

/* The databits decoders main() plugs in behind the search (databits.h:49-92:
 * databits_decode_ascii8 / _baudot / _binary / _callerid / _uic_ground /
 * _uic_train).  O(1) per frame, stateful (Baudot shift, caller-ID message
 * buffer), byte-serial: host work over the frame bits the device gathered.
 * One object = the decoder's static state in the reference. */
typedef struct mifsk_databits mifsk_databits;

int  mifsk_databits_create( mifsk_databits **out, int decoder /* enum mifsk_decoder */ );
void mifsk_databits_destroy( mifsk_databits *d );
/* bfsk_databits_decode(0, 0, 0, 0): minimodem.c:1351 */
void mifsk_databits_reset( mifsk_databits *d );
/* bfsk_databits_decode(out, out_size, bits, n_databits): returns the bytes
 * produced (never more than out_size; the reference's 4096-byte buffer is the
 * usual size) */
unsigned mifsk_databits_decode( mifsk_databits *d, char *out, unsigned out_size,
	unsigned long long bits, unsigned n_databits );

/* bfsk_databits_encode(words, c): the 1 or 2 data words character c is sent as
 * (Baudot: a shift code first when needed; 0 = cannot be sent).  Transmit-side
 * state (the Baudot shift) lives in the same object. */
unsigned mifsk_databits_encode( mifsk_databits *d, unsigned *words_out /* [2] */, char c );

#define MIFSK_TEXT_PRINT_FILTER	1u	/* -p, --print-filter (minimodem.c:1451-1460) */
#define MIFSK_TEXT_QUIET	2u	/* -q, --quiet: no CARRIER / NOCARRIER lines  */

/* Everything the reference writes for one stream: stdout (decoded text) and
 * stderr ("### CARRIER ..." at each acquisition, "\n### NOCARRIER ..." with
 * the episode statistics, minimodem.c:253-291,1336-1348) from the frame data
 * bits in loop order (mifsk_demod_io.d_bits) and the episodes.  *out_len /
 * *err_len receive the full lengths; at most the capacities are written.
 * Returns 0, -ENOSPC when something was cut, -EINVAL. */
int mifsk_stream_text( const mifsk_rx_config *cfg,
	const uint64_t *bits, uint32_t nframes,
	const mifsk_episode *episodes, uint32_t nepisodes, unsigned flags,
	char *out, size_t out_cap, size_t *out_len,
	char *err, size_t err_cap, size_t *err_len );

/* ---- input side: the step before the path ------------------- */

/* RIFF/WAVE header of a `--rx --file` input: PCM16 or IEEE float32, mono
 * (what the reference reads through libsndfile in its tests;
 * simpleaudio-sndfile.c:113-160).  -EINVAL: not a WAV file; -ENOTSUP: a
 * format or channel count the receive path does not take. */
typedef struct mifsk_wav_info {
    unsigned	sample_rate;
    unsigned	channels;
    unsigned	bits_per_sample;
    int		is_float;
    size_t	data_offset;	/* bytes from the start of the file */
    size_t	nframes;
} mifsk_wav_info;

int mifsk_wav_parse( const void *file, size_t len, mifsk_wav_info *info );

/* sf_readf_float() on 16-bit input, for a whole batch on the device:
 * d_samples[s][i] = d_pcm[s][i] / 32768 (+ the --Xrxnoise term) for
 * i < nsamples[s], 0.0 up to stream_stride.  `rxnoise` is the option's factor
 * (0 = off): the reference adds (rand()/RAND_MAX - 0.5f) * 2 * factor with an
 * INTEGER division, i.e. the constant -factor (simpleaudio-sndfile.c:64-69;
 * tests/40-noise.test sweeps it as a DC offset).  Strides in elements;
 * fastest when pcm_stride % 8 == 0, stream_stride % 4 == 0 and both bases are
 * 16-byte aligned.  Asynchronous on `stream`. */
int mifsk_ingest_s16( mifsk_ctx *ctx, const int16_t *d_pcm, size_t pcm_stride,
	float *d_samples, size_t stream_stride, const uint32_t *d_nsamples, uint32_t nsamples,
	int nstreams, float rxnoise, void *stream );
/* the --Xrxnoise term alone, in place, for input that is already float */
int mifsk_ingest_rxnoise_f32( mifsk_ctx *ctx, float *d_samples, size_t stream_stride,
	const uint32_t *d_nsamples, uint32_t nsamples, int nstreams, float rxnoise, void *stream );

/* ---- a list of audio files -> one batch ----------------------- */

/* `minimodem --rx --file F <mode>` for N files at once (reference:
 * simpleaudio-sndfile.c:113-160 open, :42-74 read, minimodem.c:1014-1032): every
 * file's RIFF/WAVE header is parsed (PCM16 or float32, mono), the files are
 * grouped by (sample rate, sample format) -- the reference takes the sample rate
 * from the file and derives the whole configuration from it, so each group gets
 * its own mifsk_rx_config from `args` -- and each group runs through the host
 * pipeline above: worker threads pread() the RAW samples into pinned memory, they
 * cross PCIe as they are in the file, PCM16 is converted and --Xrxnoise added on
 * the device, and the receive loop runs over the chunk while the next one is read
 * and copied.  Results are owned by the returned object. */
typedef struct mifsk_file_result {
    int			error;		/* 0, or -errno: open/read failed, not a WAV
					   (-EINVAL), unsupported format (-ENOTSUP)  */
    mifsk_wav_info	info;
    const mifsk_rx_config *cfg;		/* the configuration this file was decoded
					   with (its sample rate); NULL on error     */
    uint32_t		nframes, nbytes, nepisodes;
    uint32_t		status;		/* MIFSK_STREAM_*                            */
    int32_t		carrier_band;	/* --auto-carrier (see mifsk_demod_io)       */
    uint32_t		reserved;
    const uint64_t	*bits;		/* [nframes] data bits of every frame        */
    const uint8_t	*bytes;		/* [nbytes]                                  */
    const mifsk_frame	*frames;	/* [nframes], or NULL without WANT_FRAMES    */
    const mifsk_episode	*episodes;	/* [nepisodes]                               */
} mifsk_file_result;

typedef struct mifsk_files mifsk_files;

#define MIFSK_FILES_WANT_FRAMES	0x1000u	/* keep the per-frame records too */

/* flags: MIFSK_IO_RING_EXACT / _ENGINE_* / MIFSK_FILES_*.  Returns 0 when the
 * batch ran (per-file failures are in mifsk_file_result.error), -errno when it
 * could not. */
int mifsk_demod_files( mifsk_ctx *ctx, const mifsk_modem_args *args,
	const char *const *paths, int nfiles, float rxnoise, unsigned flags, mifsk_files **out );
int mifsk_files_count( const mifsk_files *f );
const mifsk_file_result *mifsk_files_get( const mifsk_files *f, int i );
const mifsk_host_stats *mifsk_files_stats( const mifsk_files *f );
void mifsk_files_free( mifsk_files *f );

/* ---- transmit side (test and benchmark input generator) ------------------ */

/* (simpleaudio_tone_init, simple-tone-generator.c:60-89, has no counterpart: the generator's
 * state is per call -- mifsk_tx_synthesize takes the table length and the magnitude itself.) */
/* fsk_transmit_stdin + simpleaudio_tone for one stream of data words
 * (minimodem.c:81-250, simple-tone-generator.c:106-175): returns the stream's
 * length in samples (also when out is NULL or too small), or -errno */
long mifsk_tx_synthesize( const mifsk_rx_config *cfg, const uint8_t *words, size_t nwords,
	unsigned sin_table_len, float amplitude, unsigned leading_silence, int as_s16,
	float *out, size_t out_cap );

/* The same generator for a whole batch on the device: stream s is
 * d_words[s][0 .. nwords[s]) (uniform `nwords` when d_nwords is NULL) behind
 * leading_silence[s] zero samples, written to d_out[s][..out_stride) with the rest
 * of the row zeroed; d_nsamples_out[s] receives its length (which may exceed
 * out_stride: the row is then cut).  sin_table_len == 0 is --lut=0 (a sinf per
 * sample: glibc's algorithm restated on the device, csrc/mifsk_sinf.h).
 * Bit-identical to mifsk_tx_synthesize.  Asynchronous on `stream`. */
int mifsk_tx_synthesize_batch( mifsk_ctx *ctx, const mifsk_rx_config *cfg,
	const uint8_t *d_words, size_t words_stride, const uint32_t *d_nwords, uint32_t nwords,
	int nstreams, unsigned sin_table_len, float amplitude,
	const uint32_t *d_leading_silence, uint32_t leading_silence, int as_s16,
	float *d_out, size_t out_stride, uint32_t *d_nsamples_out, void *stream );

#ifdef __cplusplus
}
#endif

#endif /* MIFSK_H */
