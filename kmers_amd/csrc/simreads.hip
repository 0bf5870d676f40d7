// simreads.hip -- SURVEY.md section 8(f) row 3: simulated reads written straight into a packed store in HBM, so that
// BASELINE.json configs[3] needs no multi-gigabyte reads FASTA on disk.
//
// Reference: kmerexpr/simulate_reads.py:54-122.  Its random choices come from numpy's legacy generator
// (np.random.seed(42), :55): psi ~ dirichlet (:96), the isoform of every read from np.random.choice(T, N, p=theta_true)
// (:105) and every read's start from np.random.choice(len - L + 1) (:116).  The host side here
// (kmers_amd/simulate_reads.py) makes the SAME numpy calls for psi and the isoforms, and continues the SAME MT19937
// stream for the starts: numpy's scalar choice(n) is randint(0, n), which masks successive 32-bit outputs with the
// smallest 2^k - 1 >= n - 1 and rejects values above n - 1 (numpy/random/src/distributions/distributions.c,
// buffered_bounded_masked_uint32; n == 1 draws nothing).  kmers_sim_starts replays exactly that over a block of raw
// 32-bit outputs, so (isoform, start) of every read are the reference's, bit for bit, and only the copying of the L
// bases -- the part that is O(N L) -- runs on the GPU.
#include "kmers_hip.h"

#include <hip/hip_runtime.h>

#include <cstdio>

namespace {

using u32 = uint32_t;
using u64 = uint64_t;

// One thread per 32 output positions (two words of bases, one of validity).  Read n occupies positions
// [n (L + 1), n (L + 1) + L) and one separator; everything from n_pos_out up to the end of the arrays is zero.
__global__ __launch_bounds__(256) void k_sim_reads(const u32 *__restrict__ bases, const u32 *__restrict__ valid,
                                                   const u64 *__restrict__ rec_start, u64 n_rec,
                                                   const u32 *__restrict__ iso, const u32 *__restrict__ start, u64 n_reads,
                                                   u32 L, u32 *__restrict__ out_bases, u64 out_bases_words,
                                                   u32 *__restrict__ out_valid, u64 out_valid_words, u32 *__restrict__ err) {
    const u64 g = (u64)blockIdx.x * 256 + threadIdx.x;  // index of the validity word
    if (g >= out_valid_words && 2 * g >= out_bases_words) return;
    const u64 n_pos_out = n_reads * (L + 1);
    u64 p = g * 32;
    u64 n = p / (L + 1);
    u32 o = (u32)(p % (L + 1));
    u64 src0 = 0;
    bool ok = false;
    auto open_read = [&]() {  // source position of the read's first base, checked against its record
        ok = false;
        if (n >= n_reads) return;
        const u64 r = iso[n];
        if (r >= n_rec) {
            *err = 1;
            return;
        }
        const u64 a = rec_start[r], len = rec_start[r + 1] - a - 1;
        if ((u64)start[n] + L > len) {
            *err = 1;
            return;
        }
        src0 = a + start[n];
        ok = true;
    };
    open_read();
    // The 32 positions are copied in pieces of up to 16 that cross neither a read boundary nor an output word: a
    // piece's bases are a bit-field of two consecutive source words, its validity bits of two validity words (both
    // source arrays carry 4 words of padding).
    u32 w[2] = {0u, 0u}, v = 0;
    u32 j = 0;  // position inside this thread's 32
    while (j < 32 && p < n_pos_out) {
        u32 len = 16 - (j & 15);                       // to the end of the output word
        if (o < L) len = min(len, L - o);              // to the end of the read's bases
        else len = 1;                                  // the separator
        len = (u32)min((u64)len, n_pos_out - p);
        if (o < L && ok) {
            const u64 s = src0 + o;
            const u64 b64 = ((u64)bases[s >> 4] << 32) | bases[(s >> 4) + 1];
            const u32 field = (u32)((b64 << (2 * (u32)(s & 15))) >> (64 - 2 * len));      // len bases, right-aligned
            w[j >> 4] |= field << (32 - 2 * ((j & 15) + len));
            const u64 v64 = (u64)valid[s >> 5] | ((u64)valid[(s >> 5) + 1] << 32);
            const u32 bits = (u32)(v64 >> (u32)(s & 31)) & (len == 32 ? 0xFFFFFFFFu : ((1u << len) - 1u));
            v |= bits << j;
        }
        j += len;
        p += len;
        o += len;
        if (o == L + 1) {
            o = 0;
            ++n;
            open_read();
        }
    }
    const u32 w0 = w[0], w1 = w[1];
    if (2 * g < out_bases_words) out_bases[2 * g] = w0;
    if (2 * g + 1 < out_bases_words) out_bases[2 * g + 1] = w1;
    if (g < out_valid_words) out_valid[g] = v;
}

}  // namespace

extern "C" int kmers_sim_starts(const uint32_t *raw, uint64_t n_raw, const uint32_t *n_choices, uint64_t n,
                                uint32_t *start_out, uint64_t *reads_done, uint64_t *raw_used) {
    if (!raw || !n_choices || !start_out || !reads_done || !raw_used) return KMERS_ERR_BAD_ARG;
    uint64_t k = 0, i = 0;
    for (; i < n; ++i) {
        if (n_choices[i] == 0) return KMERS_ERR_BAD_ARG;
        const uint32_t rng = n_choices[i] - 1;
        if (rng == 0) {  // one choice: numpy returns it without drawing
            start_out[i] = 0;
            continue;
        }
        uint32_t mask = rng;
        mask |= mask >> 1, mask |= mask >> 2, mask |= mask >> 4, mask |= mask >> 8, mask |= mask >> 16;
        const uint64_t k0 = k;
        uint32_t v = 0;
        bool got = false;
        while (k < n_raw) {
            v = raw[k++] & mask;
            if (v <= rng) {
                got = true;
                break;
            }
        }
        if (!got) {  // the block of raw outputs ran out inside this read: it is redone with the next block
            k = k0;
            break;
        }
        start_out[i] = v;
    }
    *reads_done = i;
    *raw_used = k;
    return KMERS_OK;
}

extern "C" int kmers_hip_simulate_reads(const uint32_t *d_bases, const uint32_t *d_valid, const uint64_t *d_rec_start,
                                        uint64_t n_rec, const uint32_t *d_iso, const uint32_t *d_start, uint64_t n_reads,
                                        uint32_t L, uint32_t *d_out_bases, size_t out_bases_words, uint32_t *d_out_valid,
                                        size_t out_valid_words, void *stream) {
    if (!d_bases || !d_valid || !d_rec_start || !d_out_bases || !d_out_valid || L == 0 || (n_reads && (!d_iso || !d_start)))
        return KMERS_ERR_BAD_ARG;
    const uint64_t n_pos = n_reads * ((uint64_t)L + 1);
    if (out_bases_words < (n_pos + 15) / 16 + 4 || out_valid_words < (n_pos + 31) / 32 + 4) return KMERS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    u32 *d_err = nullptr;
    auto fail = [&](hipError_t e, const char *what) {
        fprintf(stderr, "libkmers: HIP error %s (%s)\n", hipGetErrorString(e), what);
        if (d_err) (void)hipFree(d_err);
        return KMERS_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipMalloc((void **)&d_err, 4)) != hipSuccess) return fail(e, "hipMalloc");
    if ((e = hipMemsetAsync(d_err, 0, 4, s)) != hipSuccess) return fail(e, "hipMemsetAsync");
    const uint64_t groups = out_valid_words > (out_bases_words + 1) / 2 ? out_valid_words : (out_bases_words + 1) / 2;
    hipLaunchKernelGGL(k_sim_reads, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, d_bases, d_valid, d_rec_start,
                       (u64)n_rec, d_iso, d_start, (u64)n_reads, L, d_out_bases, (u64)out_bases_words, d_out_valid,
                       (u64)out_valid_words, d_err);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "k_sim_reads");
    u32 h_err = 0;
    if ((e = hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) return fail(e, "hipMemcpyAsync");
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return fail(e, "hipStreamSynchronize");
    (void)hipFree(d_err);
    return h_err ? KMERS_ERR_BAD_ARG : KMERS_OK;  // an isoform index or a start outside its record (the read is left empty)
}
