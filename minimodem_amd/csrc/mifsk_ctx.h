// mifsk_ctx.h -- the device context as the host-side sources of libmifsk.so share it
// (mifsk_capi.cpp owns it; mifsk_hostpipe.cpp adds the host-memory pipeline's state).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>
#include <mutex>
#include <shared_mutex>
#include <vector>

#include "mifsk.h"
#include "mifsk_device.h"

namespace mifsk {
struct HostWork;
void host_work_destroy( HostWork *w );		// (mifsk_hostpipe.cpp)
}

using mifsk::DevCfg;

struct TwKey {
    unsigned fftsize, b_mark, b_space, bit_nsamples;
    bool operator==( const TwKey &o ) const
    {
	return fftsize == o.fftsize && b_mark == o.b_mark && b_space == o.b_space
	    && bit_nsamples == o.bit_nsamples;
    }
};

struct TwEntry {
    TwKey	key;
    double	*d_tw;
};

// device-resident copies of the kernel configuration, one per distinct config
struct CfgEntry {
    DevCfg	host;
    DevCfg	*dev;
    // shared segments: rotation tables of the four scans (mifsk_device.h WaveAuto::d_rot)
    double	*d_rot[5];
    uint32_t	rot_stride[5];
};

struct mifsk_ctx {
    int			device;
    int			ncu;		// compute units (occupancy planning)
    char		name[256];
    std::mutex		lock;
    // Cached device tables (twiddles, DevCfg copies, the spectrum table) are shared by every
    // call on the context.  A call holds `gate` shared from the moment it looks a table up
    // until its kernels are enqueued; the collector (cache_gc, mifsk_capi.cpp) takes it
    // exclusively and synchronises the device before it frees anything -- so nothing is freed
    // between a lookup and a launch, or under a kernel that is still running.
    std::shared_mutex	gate;
    size_t		table_bytes = 0;	// twiddle tables held
    std::vector<TwEntry>	tables;
    std::vector<CfgEntry>	configs;
    // the derived kernel configuration of the configurations seen last (fill_devcfg plans the
    // shared segments of every scan: 0.3 ms for RTTY -- per call, on the launch path, it was
    // 3 % of that kernel's time): a few entries, replaced round-robin
    struct DerivedCfg { mifsk_rx_config key; DevCfg d; };
    std::vector<DerivedCfg>	derived;
    size_t		derived_next = 0;
    // (the spectrum table of fsk_detect_carrier is a TwEntry with bit_nsamples == 0)
    // the host-memory pipeline's streams, events and pinned staging (mifsk_hostpipe.cpp)
    mifsk::HostWork	*host = nullptr;
    // chained launches (mifsk_device.h WaveChain): the groups' streams and events and the state
    // array, made at the first batch that is cut; one chain is enqueued at a time (chain_lock),
    // and each waits for the one before on the device (ev_done)
    std::mutex		chain_lock;
    mifsk::WaveChain	chain = {};
    bool		chain_made = false;
};

#define HIP_OK(call)	do { hipError_t e_ = (call); if ( e_ != hipSuccess ) { \
	fprintf(stderr, "mifsk: %s failed: %s\n", #call, hipGetErrorString(e_)); \
	return -EIO; } } while (0)


// -EINVAL for a configuration the kernels cannot take (mifsk_capi.cpp)
int mifsk_check_cfg( const mifsk_rx_config *cfg );
