// mifsk_gather.cpp -- one process per GPU: every rank's decoded bytes to the root, behind the C
// ABI (include/mifsk.h "decoded bytes to one rank").
//
// The receive path has no exchange step: streams are independent and rank r demodulates
// mifsk_shard_range(nstreams, r, world).  What crosses devices is the RESULT -- 1 byte per 400
// input samples at 1200 baud -- when the deployment has one process per GPU and one of them
// must end up holding everything (the reference has no counterpart: it is one process, one
// file, src/minimodem.c:1014-1032).  That exchange is a gather to a root, and on xGMI (point to
// point, one link per peer) the cheapest form of it is what this file enqueues: every peer
// sends on its own link, the root posts all its receives in one group.
//
// RCCL is opened at run time (dlopen), only when a gather object is made: a program that
// decodes files on one GPU -- the reference's own main() over this library -- neither links
// nor loads a communication library (what one file costs: INTEGRATION.md 1b).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cerrno>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "mifsk.h"

static_assert(MIFSK_GATHER_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is RCCL's ncclUniqueId");

namespace {

struct Rccl {
    void		*handle = nullptr;
    ncclResult_t	(*GetUniqueId)( ncclUniqueId * ) = nullptr;
    ncclResult_t	(*CommInitRank)( ncclComm_t *, int, ncclUniqueId, int ) = nullptr;
    ncclResult_t	(*CommDestroy)( ncclComm_t ) = nullptr;
    ncclResult_t	(*GroupStart)() = nullptr;
    ncclResult_t	(*GroupEnd)() = nullptr;
    ncclResult_t	(*Send)( const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t ) = nullptr;
    ncclResult_t	(*Recv)( void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t ) = nullptr;
    const char		*(*GetErrorString)( ncclResult_t ) = nullptr;
    int			rc = -ENOSYS;
};

// the copy of RCCL this process already has (a torch.distributed job brings its own), else the
// ROCm installation's
const Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
	const char *names[] = { "librccl.so.1", "librccl.so" };
	for ( const char *n : names )
	    if ( !r.handle )
		r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
	for ( const char *n : names )
	    if ( !r.handle )
		r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
	if ( !r.handle ) {
	    fprintf(stderr, "mifsk: librccl.so.1 cannot be loaded: %s\n", dlerror());
	    return;
	}
	bool ok = true;
	auto sym = [&]( const char *name ) -> void * {
	    void *p = dlsym(r.handle, name);
	    ok = ok && p != nullptr;
	    return p;
	};
	r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
	r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
	r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
	r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
	r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
	r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
	r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
	r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
	r.rc = ok ? 0 : -ENOSYS;
    });
    return r;
}

// one receive set of the root (or the loopback rank): what every peer sent in one gather
struct RxSet {
    std::vector<uint8_t *>	bytes;		// [peer] rows[peer] x cols, dense
    std::vector<int32_t *>	counts;		// [peer] rows[peer]
    std::vector<int>		rows;
    int				cols = 0;
    bool			filled = false;
};

struct TxSet {
    uint8_t	*bytes = nullptr;		// the narrow staging copy
    size_t	cap = 0;
};

} // namespace

struct mifsk_gather {
    int			rank = 0, world = 1, device = 0;
    bool		loopback = false;
    uint32_t		slots = 2;
    ncclComm_t		comm = nullptr;
    std::vector<RxSet>	rx;
    std::vector<TxSet>	tx;
    uint64_t		next = 0;
    std::mutex		lock;
};

#define RCCL_OK(call)	do { ncclResult_t e_ = (call); if ( e_ != ncclSuccess ) { \
	fprintf(stderr, "mifsk: %s failed: %s\n", #call, R.GetErrorString(e_)); \
	return -EIO; } } while (0)
#define HIP_OK_(call)	do { hipError_t e_ = (call); if ( e_ != hipSuccess ) { \
	fprintf(stderr, "mifsk: %s failed: %s\n", #call, hipGetErrorString(e_)); \
	return -EIO; } } while (0)

extern "C" int mifsk_gather_unique_id( void *id )
{
    if ( !id )
	return -EINVAL;
    const Rccl &R = rccl();
    if ( R.rc != 0 )
	return R.rc;
    ncclUniqueId u;
    RCCL_OK(R.GetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return 0;
}

static void free_sets( mifsk_gather *g )
{
    for ( RxSet &s : g->rx ) {
	for ( uint8_t *p : s.bytes ) if ( p ) (void)hipFree(p);
	for ( int32_t *p : s.counts ) if ( p ) (void)hipFree(p);
	s = RxSet();
    }
    for ( TxSet &t : g->tx ) {
	if ( t.bytes ) (void)hipFree(t.bytes);
	t = TxSet();
    }
}

extern "C" void mifsk_gather_destroy( mifsk_gather *g )
{
    if ( !g )
	return;
    (void)hipSetDevice(g->device);
    (void)hipDeviceSynchronize();		// (a set may still be the target of a receive)
    free_sets(g);
    if ( g->comm )
	(void)rccl().CommDestroy(g->comm);
    delete g;
}

extern "C" int mifsk_gather_create( mifsk_gather **out, const void *id, int rank, int world,
	int device, int slots, unsigned flags )
{
    if ( !out )
	return -EINVAL;
    *out = nullptr;
    if ( world < 1 || rank < 0 || rank >= world || ( flags & ~MIFSK_GATHER_LOOPBACK ) )
	return -EINVAL;
    if ( ( flags & MIFSK_GATHER_LOOPBACK ) && world != 1 )
	return -EINVAL;
    const bool comm_needed = world > 1 || ( flags & MIFSK_GATHER_LOOPBACK );
    if ( comm_needed && !id )
	return -EINVAL;
    int ndev = 0;
    if ( hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 )
	return -ENODEV;
    if ( device < 0 && hipGetDevice(&device) != hipSuccess )
	return -ENODEV;
    if ( device >= ndev )
	return -ENODEV;
    mifsk_gather *g = new (std::nothrow) mifsk_gather;
    if ( !g )
	return -ENOMEM;
    g->rank = rank;
    g->world = world;
    g->device = device;
    g->loopback = ( flags & MIFSK_GATHER_LOOPBACK ) != 0;
    g->slots = (uint32_t)( slots < 2 ? 2 : ( slots > 16 ? 16 : slots ) );
    g->rx.resize(g->slots);
    g->tx.resize(g->slots);
    if ( comm_needed ) {
	const Rccl &R = rccl();
	int rc = R.rc;
	if ( rc == 0 && hipSetDevice(device) != hipSuccess )
	    rc = -EIO;
	if ( rc == 0 ) {
	    ncclUniqueId u;
	    std::memcpy(&u, id, sizeof(u));
	    const ncclResult_t e = R.CommInitRank(&g->comm, world, u, rank);
	    if ( e != ncclSuccess ) {
		fprintf(stderr, "mifsk: ncclCommInitRank(rank %d of %d) failed: %s\n", rank, world,
			R.GetErrorString(e));
		g->comm = nullptr;
		rc = -EIO;
	    }
	}
	if ( rc != 0 ) {
	    delete g;
	    return rc;
	}
    }
    *out = g;
    return 0;
}

// room for `rows[p] x cols` from every peer in receive set `s`
static int fit_rx( mifsk_gather *g, RxSet &s, const int *rows, int rows_all, int cols )
{
    const int npeers = g->loopback ? 1 : g->world;
    bool same = s.cols == cols && (int)s.rows.size() == npeers;
    for ( int p = 0; same && p < npeers; p++ )
	same = s.rows[p] == ( rows ? rows[p] : rows_all );
    if ( same )
	return 0;
    // (a set being replaced may still be the target of a receive in flight)
    HIP_OK_(hipDeviceSynchronize());
    for ( uint8_t *p : s.bytes ) if ( p ) (void)hipFree(p);
    for ( int32_t *p : s.counts ) if ( p ) (void)hipFree(p);
    s = RxSet();
    s.bytes.assign(npeers, nullptr);
    s.counts.assign(npeers, nullptr);
    s.rows.assign(npeers, 0);
    s.cols = cols;
    for ( int p = 0; p < npeers; p++ ) {
	const int r = rows ? rows[p] : rows_all;
	s.rows[p] = r;
	if ( !g->loopback && p == g->rank )
	    continue;				// (the root's own rows stay where they are)
	const size_t nb = (size_t)r * (size_t)cols;
	if ( hipMalloc((void **)&s.bytes[p], nb ? nb : 1) != hipSuccess
		|| hipMalloc((void **)&s.counts[p], ( r ? (size_t)r : 1 ) * sizeof(int32_t)) != hipSuccess )
	    return -ENOMEM;
    }
    return 0;
}

extern "C" int mifsk_gather_start( mifsk_gather *g, const uint8_t *d_bytes, size_t row_pitch,
	const int32_t *d_nbytes, int nstreams, int cols, const int *rows, void *stream,
	uint64_t *ticket )
{
    if ( !g || nstreams < 0 || cols < 0 || (size_t)cols > row_pitch )
	return -EINVAL;
    if ( nstreams > 0 && ( !d_bytes || !d_nbytes ) )
	return -EINVAL;
    if ( rows && rows[g->rank] != nstreams )
	return -EINVAL;
    std::lock_guard<std::mutex> hold(g->lock);
    const uint64_t t = g->next++;
    if ( ticket )
	*ticket = t;
    if ( g->world == 1 && !g->loopback )
	return 0;				// nothing to exchange: the one rank holds everything
    const Rccl &R = rccl();
    hipStream_t st = (hipStream_t)stream;
    HIP_OK_(hipSetDevice(g->device));
    const uint32_t slot = (uint32_t)( t % g->slots );
    const bool root = g->rank == 0;
    const bool sends = !root || g->loopback;
    const uint8_t *src = d_bytes;
    const size_t nb = (size_t)nstreams * (size_t)cols;
    if ( sends && nb && (size_t)cols != row_pitch ) {
	// only the columns that can hold data travel: a dense copy made on the caller's stream
	TxSet &tx = g->tx[slot];
	if ( tx.cap < nb ) {
	    HIP_OK_(hipDeviceSynchronize());
	    if ( tx.bytes ) (void)hipFree(tx.bytes);
	    tx = TxSet();
	    if ( hipMalloc((void **)&tx.bytes, nb) != hipSuccess )
		return -ENOMEM;
	    tx.cap = nb;
	}
	HIP_OK_(hipMemcpy2DAsync(tx.bytes, (size_t)cols, d_bytes, row_pitch, (size_t)cols,
				 (size_t)nstreams, hipMemcpyDeviceToDevice, st));
	src = tx.bytes;
    }
    if ( root ) {
	const int rc = fit_rx(g, g->rx[slot], g->loopback ? nullptr : rows, nstreams, cols);
	if ( rc != 0 )
	    return rc;
    }
    RCCL_OK(R.GroupStart());
    ncclResult_t e = ncclSuccess;
    if ( root ) {
	RxSet &s = g->rx[slot];
	const int npeers = g->loopback ? 1 : g->world;
	for ( int p = 0; p < npeers && e == ncclSuccess; p++ ) {
	    if ( !g->loopback && p == 0 )
		continue;
	    const size_t n = (size_t)s.rows[p] * (size_t)cols;
	    if ( n )
		e = R.Recv(s.bytes[p], n, ncclUint8, p, g->comm, st);
	    if ( e == ncclSuccess && s.rows[p] )
		e = R.Recv(s.counts[p], (size_t)s.rows[p], ncclInt32, p, g->comm, st);
	}
	s.filled = true;
    }
    if ( sends && e == ncclSuccess ) {
	if ( nb )
	    e = R.Send(src, nb, ncclUint8, 0, g->comm, st);
	if ( e == ncclSuccess && nstreams )
	    e = R.Send(d_nbytes, (size_t)nstreams, ncclInt32, 0, g->comm, st);
    }
    const ncclResult_t e2 = R.GroupEnd();
    if ( e != ncclSuccess || e2 != ncclSuccess ) {
	fprintf(stderr, "mifsk: gather %llu failed: %s\n", (unsigned long long)t,
		R.GetErrorString(e != ncclSuccess ? e : e2));
	return -EIO;
    }
    return 0;
}

extern "C" int mifsk_gather_received( mifsk_gather *g, uint64_t ticket, int peer,
	const uint8_t **d_bytes, const int32_t **d_nbytes, int *rows, int *cols )
{
    if ( !g )
	return -EINVAL;
    std::lock_guard<std::mutex> hold(g->lock);
    if ( ticket >= g->next || g->next - ticket > g->slots )
	return -EINVAL;				// not started yet, or its set has been reused
    if ( g->rank != 0 )
	return -EINVAL;
    const RxSet &s = g->rx[ticket % g->slots];
    const int npeers = (int)s.rows.size();
    if ( !s.filled || peer < 0 || peer >= npeers || ( !g->loopback && peer == 0 ) )
	return -EINVAL;
    if ( d_bytes ) *d_bytes = s.bytes[peer];
    if ( d_nbytes ) *d_nbytes = s.counts[peer];
    if ( rows ) *rows = s.rows[peer];
    if ( cols ) *cols = s.cols;
    return 0;
}

extern "C" int mifsk_gather_info_get( const mifsk_gather *g, mifsk_gather_info *info )
{
    if ( !g || !info )
	return -EINVAL;
    std::memset(info, 0, sizeof(*info));
    info->rank = g->rank;
    info->world = g->world;
    info->device = g->device;
    info->slots = g->slots;
    info->loopback = g->loopback ? 1u : 0u;
    info->communicator = g->comm ? 1u : 0u;
    return 0;
}
