// Partition of one flattened batch over G devices - pure host arithmetic, shared by the product's multi-device
// dispatcher (fabgpu_multi.hip) and by the fake backend of the CPU tests (hosttest.cpp: G host threads + a memcpy "all-gather").
//
// Signatures are independent, so a shard is a contiguous range whose start is a multiple of 64: each device then writes whole
// u64 verdict words, every device contributes the SAME number of words (the all-gather wants equal counts; tail devices pad
// with zero words), and the merged bitmap is simply the concatenation.
//   count mode   rank g gets [g * per, (g + 1) * per) with per = 64 * ceil(ceil(n / 64) / G)            (verify-only batches)
//   bytes mode   boundaries still at multiples of 64 tuples, but chosen so that every shard hashes about the same number of
//                message BYTES (fused batches: "in hash mode balance by total message bytes"); a shard is at most words_per_rank
//                words long, so the equal-count all-gather still holds.
// The reference has no analogue: its fan-out is one goroutine per transaction (core/committer/txvalidator/v20/validator.go:198-208).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace fab {

struct ShardPlan {
    uint32_t G = 1;
    size_t n = 0;
    size_t words_per_rank = 0;            // u64 words every rank contributes to the all-gather
    std::vector<size_t> lo, hi;           // tuples [lo[g], hi[g]); lo % 64 == 0; empty shards have lo == hi
    std::vector<size_t> word_at;          // bytes mode: index of the shard's first word inside the merged bitmap (lo / 64)
};

inline ShardPlan plan_by_count(size_t n, uint32_t G) {
    ShardPlan p;
    p.G = G ? G : 1;
    p.n = n;
    const size_t words = (n + 63) / 64;
    p.words_per_rank = (words + p.G - 1) / p.G;
    const size_t per = p.words_per_rank * 64;
    for (uint32_t g = 0; g < p.G; g++) {
        size_t lo = (size_t)g * per;
        if (lo > n) lo = n;
        size_t hi = lo + per;
        if (hi > n) hi = n;
        p.lo.push_back(lo);
        p.hi.push_back(hi);
        p.word_at.push_back((size_t)g * p.words_per_rank);
    }
    return p;
}

// off: n + 1 running byte offsets (off[i+1] >= off[i]).  Greedy cut at 64-tuple granularity: shard g ends at the first word
// boundary where the bytes so far reach (g + 1) / G of the total, but never holds more than 2x the even share of WORDS (a bound
// on the per-device workspace) and always leaves the remaining shards enough room for the remaining words.
inline ShardPlan plan_by_bytes(size_t n, const uint32_t* off, uint32_t G) {
    ShardPlan p;
    p.G = G ? G : 1;
    p.n = n;
    const size_t words = (n + 63) / 64;
    const size_t even = (words + p.G - 1) / p.G;
    const size_t cap = even * 2 > words ? words : even * 2;
    p.words_per_rank = cap ? cap : 0;
    const uint64_t total = n ? (uint64_t)off[n] - off[0] : 0;
    size_t w = 0;                         // next unassigned word
    for (uint32_t g = 0; g < p.G; g++) {
        const size_t left_ranks = p.G - g - 1;
        size_t end = w;
        const uint64_t target = total / p.G * (g + 1) + (g + 1 == p.G ? total % p.G : 0);
        while (end < words && end - w < cap) {
            const size_t t = (end + 1) * 64 < n ? (end + 1) * 64 : n;
            const uint64_t bytes_to = (uint64_t)off[t] - off[0];
            if (left_ranks && bytes_to > target && end > w) break;             // the next word would overshoot this shard's share
            end++;
            if (left_ranks && bytes_to >= target) break;
        }
        // the remaining ranks must be able to take what is left
        while (words - end > left_ranks * cap) end++;
        if (g + 1 == p.G) end = words;
        const size_t lo = w * 64 < n ? w * 64 : n, hi = end * 64 < n ? end * 64 : n;
        p.lo.push_back(lo);
        p.hi.push_back(hi);
        p.word_at.push_back(w);
        w = end;
    }
    return p;
}

}  // namespace fab
