// store.hpp -- the packed sequence store shared by the host ingest and the device launchers.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>

struct kmers_store {
    uint32_t *bases = nullptr;  // 2-bit codes, 16 positions per word, big-endian inside the word
    uint32_t *valid = nullptr;  // 1 bit per position, little-endian inside the word
    size_t bases_words = 0, valid_words = 0;
    bool pinned = false;
    std::vector<uint64_t> rec_start;  // n_rec + 1 entries, local to this shard
    uint64_t n_rec_total = 0;         // records over all shards
    uint64_t first_rec = 0;           // global index of this shard's first record
};

namespace kmers {

// Threads for host-side work: KMER_NUM_THREADS (the reference's knob, src/libkmers.cpp:50-52) or all cores.
int host_threads();

// Run fn(i) for i in [0, n) on up to host_threads() threads (dynamic chunks of `grain`).
void parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t)> &fn);

}  // namespace kmers
