// kernels.hpp -- host-callable launchers of kernels.hip (internal to libkmers.so).
#pragma once
#include <cstddef>
#include <cstdint>
#include <hip/hip_runtime_api.h>

#include <string>
#include <vector>

namespace kk {

struct ProfEntry {
    std::string name;
    double ms;
    uint64_t calls;
};
void profile_enable(bool on);
void profile_reset();
std::vector<ProfEntry> profile_snapshot();
bool profile_on();
void profile_add(const char *name, double ms);  // one launch of `name` took ms (model.hip's own event pairs)
int read_stamps(unsigned long long *out16, bool reset);  // diagnostic builds (-DKK_STAMPS) only

// Path switches for tests and same-run A/B measurements.  The environment is read ONCE, at the first use in the process
// (a product call never calls getenv again, so a change of the environment cannot silently switch the kernels of a
// concurrent caller); after that only kmers_hip_debug_knob() changes them.
struct Knobs {
    int y_atomics = 0;                    // KMERS_Y_ATOMICS: y for K >= 9 with plain global atomics
    unsigned long long y_cap = 0;         // KMERS_Y_CAP: entries a slice's workgroup keeps before the overflow queue (0: default)
    int force_wide = 0;                   // KMERS_FORCE_WIDE: 8-byte partition entries for K <= 8 too
    int x_path = 0;                       // KMERS_X_PATH: 0 default, 1 "tables" (per-isoform LDS tables), 2 "rle" (no fused emit)
    int old_pass0 = 0;                    // KMERS_X_OLD_PASS0: the position-ordered pass 0 of rounds 1-2
    unsigned long long x_chunk_pos = 0;   // KMERS_X_CHUNK_POS: positions per device call of the file-level X calls (0: the limit)
    int verbose = 0;                      // KMERS_VERBOSE
    int wide14 = 0;                       // KMERS_X_WIDE14: K = 14 in three passes of 10 + 9 + 9 bits (measured: no faster)
};
const Knobs &knobs();
int set_knob(const char *name, long long value);   // "reset" re-reads the environment; 0 or -12

size_t x_workspace_bytes(uint64_t n_pos, uint64_t n_rec, int K);

size_t y_workspace_bytes(uint64_t n_pos, int K);  // 0 for K <= 8

// Both return 0 or a KMERS_ERR_* code (include/kmers_hip.h).  count_y needs a workspace for K >= 9 (any size from one
// tile's worth up: a smaller one means more chunks).
int count_y(const uint32_t *d_bases, size_t bases_words, const uint32_t *d_valid, size_t valid_words, uint64_t n_pos, int K,
            int32_t *d_counts, void *d_ws, size_t ws_bytes, hipStream_t s, bool may_profile = true);

// The library-owned scratch of kmers_hip_count_y_static: a file-scope __device__ array in kernels.hip.  Returns its
// address on the CURRENT device (looked up with hipGetSymbolAddress on the first call per device) and its size.
constexpr size_t STATIC_SCRATCH_BYTES = (size_t)32 << 20;
constexpr uint64_t STATIC_MAX_POS = 1ull << 28;   // positions per call of the no-allocation entry point
int static_scratch(void **ptr, size_t *bytes);

int build_x(const uint32_t *d_bases, size_t bases_words, const uint32_t *d_valid, size_t valid_words, const uint64_t *d_rec_start,
            uint64_t n_pos, uint64_t n_rec, uint64_t first_col, int K, float *d_data, uint64_t *d_col_ind, uint64_t *d_row_ind,
            uint64_t max_nz, void *d_ws, size_t ws_bytes, uint64_t *nnz_out, hipStream_t s);

// d_out[i] = (uint32_t)d_in[i] (asynchronous on s): the file-level calls send 32-bit indices over PCIe
int narrow64(const uint64_t *d_in, uint32_t *d_out, uint64_t n, hipStream_t s);

}  // namespace kk
