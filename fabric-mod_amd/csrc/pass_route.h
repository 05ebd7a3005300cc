// Which device of the provider's pool a block pass runs on (bccsp_host.h GPUCSP::RouteBlock).
//
// The reference hands ONE process-global BCCSP to every channel's validator (bccsp/factory/factory.go:41-55,
// core/peer/peer.go:337-355) and validates channels side by side (core/committer/txvalidator/v20/validator.go:194-210), so the
// provider - not its callers - decides where a pass goes.  Rule: the device with the fewest passes in flight; among equals the first
// one round the ring from  block_seq mod G  (block_seq = MemoSeq(channel, number), go/extensions/gossip/state/preverify_on_arrival.go:
// a hash, so the channels of a peer and the consecutive blocks of a channel start at different points of the ring).  Pure function of
// its arguments: the CPU tests drive it through libfabgpu_hosttest.so.
#pragma once
#include <stdint.h>

namespace fab {

inline int route_block(uint64_t block_seq, const uint32_t* in_flight, int n_devices) {
    if (n_devices <= 1) return 0;
    const int start = (int)(block_seq % (uint64_t)n_devices);
    int best = start;
    uint32_t least = in_flight[start];
    for (int k = 1; k < n_devices && least != 0; k++) {
        const int g = (start + k) % n_devices;
        if (in_flight[g] < least) {
            least = in_flight[g];
            best = g;
        }
    }
    return best;
}

}  // namespace fab
