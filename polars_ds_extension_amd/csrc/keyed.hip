// keyed.hip -- from an int64 key column in ANY row order to the contiguous-group form the grouped kernels take:
// what Polars' `group_by(key)` does on the host before it calls `pl_lr` once per group (the reference's call pattern,
// tests/test_linear_exprs.py:435-474, 918-953), done on the device for the key-aware symbol `pl_lr_by`.
//   1. one pass decides whether the keys are already non-decreasing (a frame sorted by its key needs no data movement);
//   2. otherwise a radix sort of (key, row index) pairs (hipCUB; stable, so rows keep their order inside a group) and a
//      gather of the frame through the permutation.  Gathering column by column reads one 64-byte sector per 8-byte
//      element (19 of the 25 ms of a shuffled 1e8-row x 9-column frame, profiles/r02_keyed_kernel_stats_before.csv), so the
//      frame is first transposed to row-major records (one streaming pass, LDS tiles) and the permutation then fetches
//      whole ROWS: 72 contiguous bytes per random access instead of nine sectors;
//   3. run-length encoding of the sorted keys gives the distinct keys and the group sizes, an exclusive sum the offsets.
// Groups come out in ascending key order.
#include <hipcub/hipcub.hpp>
#include <type_traits>

#include "common.hpp"

namespace pds {

__global__ __launch_bounds__(256) void key_descent_kernel(const int64_t* __restrict__ keys, int64_t n, unsigned* __restrict__ flag) {
    bool found = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += (int64_t)gridDim.x * blockDim.x)
        found = found || keys[i] < keys[i - 1];
    if (__any(found) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// one pass over the keys: is any key smaller than its predecessor, and the smallest / largest key (state: [flag, -, min, max, runs]
// as int64 slots; min / max start at INT64_MAX / INT64_MIN, runs at 0)
// Every wave takes 128-key pieces (1 KiB per load instruction, 16 bytes per lane, U pieces in flight); a key's successor is the
// lane's own second key, the next lane's first (DPP shift) or -- lane 63 -- the first key of the next piece (one extra 8-byte load).
// (The first version read every key twice with 8-byte loads: 0.40 ms for 1e8 keys = 2 TB/s, a quarter of every ordered by-key call.)
// run_counts (nullable; npieces + 2 slots: [0] the unaligned first key, [1 + piece], [npieces + 1] the keys behind the last whole
// piece): how many keys j of the slot differ from their successor j + 1 < n -- for keys that ARE in order, the number of groups that
// start at j + 1.  run_masks (with run_counts; two 64-bit words per piece): WHICH keys -- bit l of word 0: key 2l of the piece
// differs from key 2l + 1, bit l of word 1: key 2l + 1 differs from key 2l + 2.  state[4] receives the sum of all counts, so the host
// knows the number of groups with the order flag (one synchronisation).  An exclusive scan of the slots and a second pass over the
// MASKS (key_run_starts_kernel: n / 8 bytes instead of the 8 n bytes of the keys; it fetches a key only where a group starts) then
// write the distinct keys and the group offsets: the run-length encoding of an ordered key column with ONE pass over the keys and
// no library pass of its own (hipCUB's: 0.35 ms per 1e8 keys; round 3/4 read the keys a second time: + 0.15 ms).
// slot_counts (nullable; kKeySlots x 8 counters, zeroed by the caller; needs a 16-byte aligned key buffer and a grid that is a multiple
// of 16): the histogram the partition route (keyed_partition.hip) starts from, taken in the same pass -- slot = (key >> hist_shift) mod
// kKeySlots, stream = (row / 4096) mod 8 (a block takes 4 x U x 128 = 4096 keys per round with U = 8: all of one stream, blockIdx mod 8).  Valid when the key range
// turns out to span at most kKeySlots buckets; bucket b is then slot (b + (min >> hist_shift)) mod kKeySlots.  A wave whose 128 keys
// share one slot -- ordered keys -- adds once.
#ifndef PDS_KEY_ORDER_U
#define PDS_KEY_ORDER_U 8
#endif
#ifndef PDS_KEY_ORDER_BPC
#define PDS_KEY_ORDER_BPC 4
#endif
__global__ __launch_bounds__(256) void key_order_minmax_kernel(const int64_t* __restrict__ keys, int64_t n, long long* __restrict__ state,
                                                               uint32_t* __restrict__ run_counts, unsigned long long* __restrict__ run_masks,
                                                               int hist_shift, unsigned* __restrict__ slot_counts) {
    typedef long long ll2 __attribute__((ext_vector_type(2), aligned(16)));
    typedef unsigned long long ull2 __attribute__((ext_vector_type(2), aligned(16)));
    __shared__ unsigned hist[kKeySlots];
    const bool do_hist = slot_counts != nullptr;
    if (do_hist) {
        for (int i = threadIdx.x; i < kKeySlots; i += 256) hist[i] = 0u;
        __syncthreads();
    }
    bool found = false;
    long long mn = 0x7fffffffffffffffll, mx = -0x7fffffffffffffffll - 1;
    unsigned long long runs = 0;  // (wave-uniform)
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    // 16-byte alignment of the buffer decides where the vector part starts
    const int64_t head = (((uintptr_t)keys & 15) != 0 && n > 0) ? 1 : 0;
    const int64_t npieces = (n - head) / 128;
    auto take = [&](long long k, long long next) __attribute__((always_inline)) {
        found = found || next < k;
        mn = k < mn ? k : mn;
        mx = k > mx ? k : mx;
    };
    constexpr int U = PDS_KEY_ORDER_U;
    static_assert(U == 4 || U == 8, "the histogram's stream of a block: 4 x U x 128 keys per round must divide or equal a 4096-row chunk");
    for (int64_t p0 = wave * U; p0 < npieces; p0 += nwaves * U) {
        ll2 v[U];
        long long edge[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t pc = p0 + u < npieces ? p0 + u : npieces - 1;  // (clamped: unconditional loads)
            const int64_t base = head + pc * 128;
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const ll2*>(keys + base) + lane);
        }
        // a piece's successor key is the next piece's first key: inside the wave's U pieces it is already in a register
        {
            const int64_t pl = p0 + U - 1 < npieces ? p0 + U - 1 : npieces - 1;
            const int64_t e = head + pl * 128 + 128 < n ? head + pl * 128 + 128 : n - 1;
            edge[U - 1] = keys[e];  // (wave-uniform address: one scalar-like broadcast load)
        }
#pragma unroll
        for (int u = 0; u + 1 < U; ++u) {
            // (valid whenever piece p0 + u + 1 exists; a clamped duplicate otherwise -- then piece p0 + u is the last one or beyond,
            //  and the last whole piece takes its successor from memory below)
            edge[u] = __shfl(v[u + 1].x, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p0 + u < npieces) {
                long long nxt = __shfl_down(v[u].x, 1);
                if (lane == 63) {
                    long long ed = edge[u];
                    if (u + 1 < U && p0 + u + 1 >= npieces) {  // the last whole piece inside an unrolled group
                        const int64_t e = head + (p0 + u) * 128 + 128 < n ? head + (p0 + u) * 128 + 128 : n - 1;
                        ed = keys[e];
                    }
                    nxt = ed;  // (the frame's last key is its own successor: the clamped index)
                }
                take(v[u].x, v[u].y);
                take(v[u].y, nxt);
                if (run_counts) {
                    const unsigned long long bx = __ballot(v[u].x != v[u].y), by = __ballot(v[u].y != nxt);
                    const unsigned c = (unsigned)__popcll(bx) + (unsigned)__popcll(by);
                    runs += c;
                    if (lane == 0) {
                        run_counts[1 + p0 + u] = c;
                        ull2 m;
                        m.x = bx;
                        m.y = by;
                        *reinterpret_cast<ull2*>(run_masks + 2 * (p0 + u)) = m;
                    }
                }
                if (do_hist) {
                    const unsigned sx = (unsigned)((v[u].x >> hist_shift) & (long long)(kKeySlots - 1));
                    const unsigned sy = (unsigned)((v[u].y >> hist_shift) & (long long)(kKeySlots - 1));
                    const unsigned f = (unsigned)__builtin_amdgcn_readfirstlane((int)sx);
                    if (__all(sx == f && sy == f)) {
                        if (lane == 0) atomicAdd(&hist[f], 128u);
                    } else {
                        atomicAdd(&hist[sx], 1u);
                        atomicAdd(&hist[sy], 1u);
                    }
                }
            }
        }
    }
    // the unaligned first key and the keys behind the last whole piece
    if (wave == 0) {
        if (head && lane == 0) take(keys[0], n > 1 ? keys[1] : keys[0]);
        unsigned ct = 0;
        for (int64_t i0 = head + npieces * 128; i0 < n; i0 += 64) {  // (wave-uniform trip count: the ballot below)
            const int64_t i = i0 + lane;
            const bool in = i < n;
            const long long k = in ? keys[i] : 0, nx = (in && i + 1 < n) ? keys[i + 1] : k;
            if (in) take(k, nx);
            ct += (unsigned)__popcll(__ballot(in && k != nx));
            if (do_hist && in)  // (fewer than 128 keys: straight to the counters of their own chunk's stream)
                atomicAdd(&slot_counts[(unsigned)((k >> hist_shift) & (long long)(kKeySlots - 1)) * 8u + (unsigned)((i >> 12) & 7)], 1u);
        }
        if (run_counts) {
            const unsigned c0 = (head && n > 1 && keys[0] != keys[1]) ? 1u : 0u;
            runs += c0 + ct;
            if (lane == 0) {
                run_counts[0] = c0;
                run_counts[1 + npieces] = ct;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const long long a = __shfl_down(mn, o), b = __shfl_down(mx, o);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    // one set of atomics per BLOCK (the 8192 waves of the first version queued 24 000 atomics on three addresses: most of its 0.4 ms)
    __shared__ long long red[2][4];
    __shared__ unsigned long long red_runs[4];
    __shared__ int any_found[4];
    const int wv = threadIdx.x >> 6;
    const bool wf = __any(found);
    if (lane == 0) {
        red[0][wv] = mn;
        red[1][wv] = mx;
        red_runs[wv] = runs;
        any_found[wv] = wf ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            mn = red[0][w] < mn ? red[0][w] : mn;
            mx = red[1][w] > mx ? red[1][w] : mx;
        }
        if (any_found[0] | any_found[1] | any_found[2] | any_found[3]) atomicOr(reinterpret_cast<unsigned long long*>(state), 1ull);
        atomicMin(state + 2, mn);
        atomicMax(state + 3, mx);
        const unsigned long long rs = red_runs[0] + red_runs[1] + red_runs[2] + red_runs[3];
        if (rs) atomicAdd(reinterpret_cast<unsigned long long*>(state + 4), rs);
    }
    if (do_hist) {  // (the __syncthreads above also closed the histogram)
        const unsigned stream = (unsigned)(((int64_t)blockIdx.x * (PDS_KEY_ORDER_U * 512)) >> 12) & 7u;  // (rows of a block / 4096) mod 8
        for (int i = threadIdx.x; i < kKeySlots; i += 256) {
            const unsigned c = hist[i];
            if (c) atomicAdd(&slot_counts[(unsigned)i * 8u + stream], c);
        }
    }
}

__global__ __launch_bounds__(256) void iota_u32_kernel(uint32_t* __restrict__ idx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) idx[i] = (uint32_t)i;
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src, const uint32_t* __restrict__ perm, int64_t n,
                                                          T* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[perm[i]];
}

// ---- columns -> row-major records: block = 256 rows, transposed through LDS (both sides coalesced)
constexpr int kAosRows = 256;
template <typename T>
__global__ __launch_bounds__(256) void cols_to_rows_kernel(const T* const* __restrict__ cols, int nc, int64_t n, T* __restrict__ rows) {
    extern __shared__ __attribute__((aligned(16))) char aos_lds[];
    T* tile = reinterpret_cast<T*>(aos_lds);
    const int ncp = nc | 1;  // odd stride: conflict-free column writes
    const int64_t nblk = (n + kAosRows - 1) / kAosRows;
    for (int64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        const int64_t r0 = b * kAosRows;
        const int rows_here = (int)((n - r0 < kAosRows) ? n - r0 : kAosRows);
        for (int c = 0; c < nc; ++c) {
            const gptr<T> col = as_global(cols[c]);
            if ((int)threadIdx.x < rows_here) tile[threadIdx.x * ncp + c] = __builtin_nontemporal_load(col + r0 + threadIdx.x);
        }
        __syncthreads();
        T* out = rows + r0 * nc;
        const int total = rows_here * nc;
        for (int e = threadIdx.x; e < total; e += 256) {
            const int r = e / nc, c = e - r * nc;
            out[e] = tile[r * ncp + c];
        }
        __syncthreads();
    }
}

// ---- dst_c[i] = rows[perm[i]][c]: one random access per ROW (nc contiguous values), writes coalesced per column
template <typename T>
__global__ __launch_bounds__(256) void gather_records_kernel(const T* __restrict__ rows, const uint32_t* __restrict__ perm, int nc,
                                                             int64_t n, T* const* __restrict__ dst) {
    constexpr int E = 16 / (int)sizeof(T);  // elements per 16-byte load (records are only element aligned)
    using V = typename std::conditional<sizeof(T) == 8, double __attribute__((ext_vector_type(2), aligned(8))),
                                        float __attribute__((ext_vector_type(4), aligned(4)))>::type;
    constexpr int MAXV = 10;  // records of up to 10 vectors (20 f64 / 40 f32 values) are fetched whole before anything is stored
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T* src = rows + (int64_t)perm[i] * nc;
        const int nv = nc / E;
        if (nv <= MAXV) {
            // every load of the record is in flight before the first store (a load -> store -> load chain per lane cost
            // 7.3 ms for 1e8 records of 9 doubles)
            V v[MAXV];
            T tl[E];
#pragma unroll
            for (int k = 0; k < MAXV; ++k)
                if (k < nv) v[k] = *reinterpret_cast<const V*>(src + k * E);
#pragma unroll
            for (int e = 0; e < E - 1; ++e)
                if (nv * E + e < nc) tl[e] = src[nv * E + e];
#pragma unroll
            for (int k = 0; k < MAXV; ++k)
                if (k < nv) {
#pragma unroll
                    for (int e = 0; e < E; ++e) dst[k * E + e][i] = v[k][e];
                }
#pragma unroll
            for (int e = 0; e < E - 1; ++e)
                if (nv * E + e < nc) dst[nv * E + e][i] = tl[e];
            continue;
        }
        int c = 0;
        for (; c + E <= nc; c += E) {
            const V v = *reinterpret_cast<const V*>(src + c);
#pragma unroll
            for (int e = 0; e < E; ++e) dst[c + e][i] = v[e];
        }
        for (; c < nc; ++c) dst[c][i] = src[c];
    }
}

__global__ void close_offsets_kernel(int64_t* __restrict__ off, const int64_t* __restrict__ n_runs, int64_t n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) off[*n_runs] = n;
}

static inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

int keys_nondecreasing(pds_ctx* ctx, const int64_t* d_keys, int64_t n, unsigned* d_flag, bool* sorted) {
    PDS_HIP_CHECK(hipMemsetAsync(d_flag, 0, sizeof(unsigned), ctx->stream));
    const int nb = (int)std::min<int64_t>(std::max<int64_t>((n + 255) / 256, 1), (int64_t)ctx->num_cus * 8);
    hipLaunchKernelGGL(key_descent_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_keys, n, d_flag);
    unsigned h = 0;
    PDS_HIP_CHECK(hipMemcpyAsync(&h, d_flag, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *sorted = h == 0;
    return PDS_OK;
}

// order check + key range (+ run counts / masks / total of an ordered column) in ONE pass (d_state: 8 int64 slots on the device;
// d_state + 2 is the {min, max} pair keyed_sort and the partition route read on the device; d_state[4] the number of keys that differ
// from their successor = n_groups - 1 of an ordered column, returned through n_runs)
int keys_order_minmax(pds_ctx* ctx, const int64_t* d_keys, int64_t n, int64_t* d_state, bool* sorted, int64_t* mm, uint32_t* d_run_counts,
                      unsigned long long* d_run_masks, int64_t* n_runs, int hist_shift, unsigned* d_slot_counts, bool* hist_taken) {
    if (int rc = ensure_pinned(ctx, 4096)) return rc;
    long long* init = reinterpret_cast<long long*>(static_cast<char*>(ctx->pinned) + 2048);  // (pinned: the copies below are truly asynchronous)
    init[0] = 0;
    init[1] = 0;
    init[2] = 0x7fffffffffffffffll;
    init[3] = -0x7fffffffffffffffll - 1;
    init[4] = 0;
    PDS_HIP_CHECK(hipMemcpyAsync(d_state, init, 5 * sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
    constexpr int64_t per_block = 4 * 128 * PDS_KEY_ORDER_U;  // keys of one block and iteration
    int nb = (int)std::min<int64_t>(std::max<int64_t>((n + per_block - 1) / per_block, 1), (int64_t)ctx->num_cus * PDS_KEY_ORDER_BPC);
    // the fused histogram: aligned keys, a grid in which a block's pieces all belong to one stream, enough keys to be worth it
    const bool hist = d_slot_counts && hist_shift >= 0 && ((uintptr_t)d_keys & 15) == 0 && nb >= 16;
    if (hist) {
        nb = nb / 16 * 16;
        PDS_HIP_CHECK(hipMemsetAsync(d_slot_counts, 0, (size_t)kKeySlots * 8 * sizeof(unsigned), ctx->stream));
    }
    if (hist_taken) *hist_taken = hist;
    if (!d_run_masks) d_run_counts = nullptr;
    hipLaunchKernelGGL(key_order_minmax_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_keys, n, reinterpret_cast<long long*>(d_state), d_run_counts,
                       d_run_masks, hist ? hist_shift : 0, hist ? d_slot_counts : (unsigned*)nullptr);
    long long* h = init + 8;
    PDS_HIP_CHECK(hipMemcpyAsync(h, d_state, 5 * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *sorted = h[0] == 0;
    mm[0] = h[2];
    mm[1] = h[3];
    if (n_runs) *n_runs = h[4];
    return PDS_OK;
}

// second pass of the run-length encoding of an ORDERED key column: prefix = exclusive scan of key_order_minmax_kernel's slots,
// masks = its change bits.  Group r > 0 starts at j + 1 where key j differs from its successor, r = 1 + prefix[slot of j] + (such keys
// in front of j in the slot); group 0 starts at row 0.  A WAVE takes 64 consecutive 128-key pieces: lane = piece reads the piece's two
// mask words (16 bytes, coalesced) and lists its run starts in the wave's LDS list -- the ranks of a wave's starts are consecutive, so
// entry e of the list is group 1 + prefix[first piece] + e -- and then the lanes walk the LIST: one key fetch and two coalesced stores
// per entry, all of a wave's fetches in flight together.  (Round 3 / 4 re-read the whole key column here: 0.8 GB per 1e8 keys against
// 12.5 MB of masks + the fetched keys; the first mask version looped over a lane's set bits with one dependent fetch per trip: 31 us.)
constexpr int kRunListCap = 1024;  // list entries per round; a wave whose 64 pieces start more groups than that goes round again
__global__ __launch_bounds__(256) void key_run_starts_kernel(const int64_t* __restrict__ keys, int64_t n, const uint32_t* __restrict__ prefix,
                                                             const unsigned long long* __restrict__ masks, int64_t* __restrict__ out_keys,
                                                             int64_t* __restrict__ offsets, int64_t cap) {
    typedef unsigned long long ull2 __attribute__((ext_vector_type(2), aligned(16)));
    __shared__ uint16_t list[4][kRunListCap];  // offset of a run start inside the wave's 8192 keys, minus 1 (fits 13 bits)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t head = (((uintptr_t)keys & 15) != 0 && n > 0) ? 1 : 0;
    const int64_t npieces = (n - head) / 128;
    const unsigned long long ltl = (1ull << lane) - 1ull;
    auto put = [&](int64_t r, long long k, int64_t start) __attribute__((always_inline)) {
        if (r < cap) {
            out_keys[r] = k;
            offsets[r] = start;
        }
    };
    const int64_t nwt = (npieces + 63) / 64;  // wave tiles (one per wave when the grid allows: the pass is a chain of latencies,
                                              // mask -> list -> key -> store, so it wants every tile in flight at once)
    for (int64_t wt = (int64_t)blockIdx.x * 4 + wv; wt < nwt; wt += (int64_t)gridDim.x * 4) {
        const int64_t pc = wt * 64 + lane;
        ull2 m;
        m.x = 0ull;
        m.y = 0ull;
        if (pc < npieces) m = __builtin_nontemporal_load(reinterpret_cast<const ull2*>(masks) + pc);
        const unsigned long long bx = m.x, by = m.y;
        const int c = __popcll(bx) + __popcll(by);
        // exclusive scan of the counts over the lanes (the wave's starts in key order: piece by piece, inside a piece x before y per lane)
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        const int total = __shfl(inc, 63), exc = inc - c;
        if (total == 0) continue;
        const int64_t r0 = 1 + (int64_t)prefix[1 + wt * 64];
        uint16_t* const li = list[wv];
        const int64_t base = head + wt * 64 * 128 + 1;
        for (int e0 = 0; e0 < total; e0 += kRunListCap) {  // (one round unless more than 1024 groups start inside 8192 keys)
            unsigned long long rx = bx, ry = by;
            while (rx) {  // key 2l differs from key 2l + 1: a group starts at 2l + 1
                const int l = __builtin_ctzll(rx);
                rx &= rx - 1;
                const unsigned long long lt = (1ull << l) - 1ull;
                const int e = exc + __popcll(bx & lt) + __popcll(by & lt) - e0;
                if (e >= 0 && e < kRunListCap) li[e] = (uint16_t)(lane * 128 + 2 * l);
            }
            while (ry) {  // key 2l + 1 differs from key 2l + 2: a group starts at 2l + 2 (the x change of the same lane comes first)
                const int l = __builtin_ctzll(ry);
                ry &= ry - 1;
                const unsigned long long lt = (1ull << l) - 1ull;
                const int e = exc + __popcll(bx & (lt | (1ull << l))) + __popcll(by & lt) - e0;
                if (e >= 0 && e < kRunListCap) li[e] = (uint16_t)(lane * 128 + 2 * l + 1);
            }
            PDS_WAVE_LDS_SYNC();
            const int cnt = total - e0 < kRunListCap ? total - e0 : kRunListCap;
            for (int e = lane; e < cnt; e += 64) {
                const int64_t st = base + (int64_t)li[e];
                put(r0 + e0 + e, keys[st], st);
            }
            PDS_WAVE_LDS_SYNC();
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        if (lane == 0) {
            put(0, keys[0], 0);
            if (head && n > 1 && keys[0] != keys[1]) put(1 + (int64_t)prefix[0], keys[1], 1);
            const int64_t ng = 1 + (int64_t)prefix[npieces + 2];  // (the scan runs over one slot more: its last entry is the total)
            if (ng <= cap) offsets[ng] = n;
        }
        int64_t r = 1 + (int64_t)prefix[1 + npieces];
        for (int64_t i0 = head + npieces * 128; i0 < n; i0 += 64) {
            const int64_t i = i0 + lane;
            const bool in = i < n;
            const long long k = in ? keys[i] : 0, nx = (in && i + 1 < n) ? keys[i + 1] : k;
            const bool f = in && k != nx;
            const unsigned long long b = __ballot(f);
            if (f) put(r + __popcll(b & ltl), nx, i + 1);
            r += __popcll(b);
        }
    }
}

size_t key_run_slots(int64_t n) { return (size_t)(n / 128 + 3); }  // head, pieces, tail (+ one for the scan's total)
size_t key_run_mask_bytes(int64_t n) { return ((size_t)(n / 128 + 1) * 16 + 255) & ~(size_t)255; }

// distinct keys + offsets (n_groups + 1 entries) of an ORDERED key column from the counts / masks key_order_minmax_kernel left
// (d_counts: key_run_slots(n) entries; d_prefix: as many + 1; d_masks: key_run_mask_bytes(n)).  cap: capacity of d_unique (d_offsets
// holds cap + 1).  Nothing is read back: the number of groups came with the order check (n_runs + 1); the launches are left on the stream.
int keyed_runs_ordered(pds_ctx* ctx, const int64_t* d_keys, int64_t n, uint32_t* d_counts, uint32_t* d_prefix,
                       const unsigned long long* d_masks, int64_t cap, int64_t* d_unique, int64_t* d_offsets, void* d_temp, size_t temp_bytes) {
    const int64_t head = (((uintptr_t)d_keys & 15) != 0 && n > 0) ? 1 : 0;
    const int64_t npieces = (n - head) / 128;
    const int slots = (int)(npieces + 2);
    // the slot behind the last one is zero: the exclusive scan over slots + 1 entries ends with the total
    PDS_HIP_CHECK(hipMemsetAsync(d_counts + slots, 0, sizeof(uint32_t), ctx->stream));
    PDS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, (const uint32_t*)d_counts, d_prefix, slots + 1, ctx->stream));
    const int nb = (int)std::min<int64_t>(std::max<int64_t>((npieces + 255) / 256, 1), (int64_t)ctx->num_cus * 8);
    hipLaunchKernelGGL(key_run_starts_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_keys, n, (const uint32_t*)d_prefix, d_masks, d_unique, d_offsets, cap);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

// temp bytes of the ordered route alone (the scan of the order check's run counts): what a frame whose keys are already in order
// needs -- the sort's temp storage is 12 bytes per row
size_t keyed_ordered_temp_bytes(int64_t n) {
    size_t c = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, c, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)std::min<int64_t>(n / 128 + 4, INT32_MAX));
    return ((c + 255) & ~(size_t)255) + 256;
}

// temp bytes of the sort + run-length + scan stages for n rows
size_t keyed_temp_bytes(int64_t n) {
    size_t a = 0, b = 0, c = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const int64_t*)nullptr, (int64_t*)nullptr, (const uint32_t*)nullptr,
                                             (uint32_t*)nullptr, (int)std::min<int64_t>(n, INT32_MAX));
    (void)hipcub::DeviceRunLengthEncode::Encode(nullptr, b, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t*)nullptr,
                                                (int64_t*)nullptr, (int)std::min<int64_t>(n, INT32_MAX));
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, c, (const int64_t*)nullptr, (int64_t*)nullptr, (int)std::min<int64_t>(n, INT32_MAX));
    size_t a32 = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a32, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                             (uint32_t*)nullptr, (int)std::min<int64_t>(n, INT32_MAX));
    a = std::max(a, a32);
    size_t d = 0;
    (void)hipcub::DeviceReduce::Min(nullptr, d, (const int64_t*)nullptr, (int64_t*)nullptr, (int)std::min<int64_t>(n, INT32_MAX));
    return up256(std::max(std::max(a, d), std::max(b, c))) + 256;
}

// key - min as an unsigned number, and back: the radix sort then only walks the bits the key range needs
__global__ __launch_bounds__(256) void rebase_keys_kernel(const int64_t* __restrict__ in, int64_t n, const int64_t* __restrict__ kmin,
                                                          int sign, int64_t* __restrict__ out) {
    const uint64_t base = (uint64_t)*kmin;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (int64_t)(sign > 0 ? (uint64_t)in[i] - base : (uint64_t)in[i] + base);
}

// key ranges below 2^32 (the usual case: group ids) sort as 32-bit numbers -- 8 bytes per (key, row) pair and pass, not 12
__global__ __launch_bounds__(256) void rebase_keys_u32_kernel(const int64_t* __restrict__ in, int64_t n, const int64_t* __restrict__ kmin,
                                                              uint32_t* __restrict__ out) {
    const uint64_t base = (uint64_t)*kmin;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (uint32_t)((uint64_t)in[i] - base);
}
__global__ __launch_bounds__(256) void widen_keys_kernel(const uint32_t* __restrict__ in, int64_t n, const int64_t* __restrict__ kmin,
                                                         int64_t* __restrict__ out) {
    const uint64_t base = (uint64_t)*kmin;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (int64_t)((uint64_t)in[i] + base);
}

// smallest and largest key: d_minmax[0..1] on the device, mm[0..1] on the host (one stream synchronisation)
int keyed_minmax(pds_ctx* ctx, const int64_t* d_keys, int64_t n, void* d_temp, size_t temp_bytes, int64_t* d_minmax, int64_t* mm) {
    PDS_HIP_CHECK(hipcub::DeviceReduce::Min(d_temp, temp_bytes, d_keys, d_minmax, (int)n, ctx->stream));
    PDS_HIP_CHECK(hipcub::DeviceReduce::Max(d_temp, temp_bytes, d_keys, d_minmax + 1, (int)n, ctx->stream));
    PDS_HIP_CHECK(hipMemcpyAsync(mm, d_minmax, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PDS_OK;
}

// (key, row) radix sort: d_sorted_keys / d_perm out; d_idx_in is scratch of n uint32; d_scratch_keys n int64; d_minmax / mm: the
// result of keyed_minmax
int keyed_sort(pds_ctx* ctx, const int64_t* d_keys, int64_t n, uint32_t* d_idx_in, int64_t* d_sorted_keys, uint32_t* d_perm,
               void* d_temp, size_t temp_bytes, int64_t* d_scratch_keys, const int64_t* d_minmax, const int64_t* mm) {
    const int nb = (int)std::min<int64_t>(std::max<int64_t>((n + 255) / 256, 1), (int64_t)ctx->num_cus * 16);
    hipLaunchKernelGGL(iota_u32_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_idx_in, n);
    // key range -> number of significant bits of (key - min): a million groups sort in 3 radix passes instead of 8
    const uint64_t range = (uint64_t)mm[1] - (uint64_t)mm[0];
    int bits = 1;
    while (bits < 64 && (range >> bits) != 0) ++bits;
    if (bits <= 32) {
        uint32_t* k32_in = reinterpret_cast<uint32_t*>(d_scratch_keys);   // the scratch holds n int64 = 2 n uint32
        uint32_t* k32_out = k32_in + n;
        hipLaunchKernelGGL(rebase_keys_u32_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_keys, n, d_minmax, k32_in);
        PDS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, (const uint32_t*)k32_in, k32_out, (const uint32_t*)d_idx_in,
                                                         d_perm, (int)n, 0, bits, ctx->stream));
        hipLaunchKernelGGL(widen_keys_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const uint32_t*)k32_out, n, d_minmax, d_sorted_keys);
        PDS_HIP_CHECK(hipGetLastError());
        return PDS_OK;
    }
    hipLaunchKernelGGL(rebase_keys_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_keys, n, d_minmax, 1, d_scratch_keys);
    // (as unsigned numbers in [0, range]: the signed sort order of int64 agrees below bit 63, and bit 63 is only
    //  walked when the range needs it, where the unsigned key type below sorts it correctly)
    PDS_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, reinterpret_cast<const uint64_t*>(d_scratch_keys),
                                                     reinterpret_cast<uint64_t*>(d_sorted_keys), (const uint32_t*)d_idx_in, d_perm, (int)n,
                                                     0, bits, ctx->stream));
    hipLaunchKernelGGL(rebase_keys_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_sorted_keys, n, d_minmax, -1, d_sorted_keys);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

// sorted keys -> distinct keys, offsets (n_groups + 1 entries), n_groups (host)
int keyed_runs(pds_ctx* ctx, const int64_t* d_sorted_keys, int64_t n, int64_t* d_unique, int64_t* d_counts, int64_t* d_offsets,
               int64_t* d_nruns, void* d_temp, size_t temp_bytes, int64_t* n_groups) {
    PDS_HIP_CHECK(hipcub::DeviceRunLengthEncode::Encode(d_temp, temp_bytes, d_sorted_keys, d_unique, d_counts, d_nruns, (int)n, ctx->stream));
    int64_t g = 0;
    PDS_HIP_CHECK(hipMemcpyAsync(&g, d_nruns, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    PDS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    PDS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, (const int64_t*)d_counts, d_offsets, (int)g, ctx->stream));
    hipLaunchKernelGGL(close_offsets_kernel, dim3(1), dim3(64), 0, ctx->stream, d_offsets, d_nruns, n);
    PDS_HIP_CHECK(hipGetLastError());
    *n_groups = g;
    return PDS_OK;
}

template <typename T>
int launch_gather_rows(pds_ctx* ctx, const T* d_src, const uint32_t* d_perm, int64_t n, T* d_dst) {
    const int nb = (int)std::min<int64_t>(std::max<int64_t>((n + 255) / 256, 1), (int64_t)ctx->num_cus * 32);
    hipLaunchKernelGGL((gather_rows_kernel<T>), dim3(nb), dim3(256), 0, ctx->stream, d_src, d_perm, n, d_dst);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

// the whole frame through the permutation: d_src / d_dst are DEVICE tables of nc column pointers, d_records n * nc elements
template <typename T>
int launch_gather_frame(pds_ctx* ctx, const T* const* d_src, const uint32_t* d_perm, int nc, int64_t n, T* d_records, T* const* d_dst) {
    static_assert(kAosRows == 256, "gather_frame_fits() sizes the tile for 256 rows");
    const size_t lds = (size_t)kAosRows * (size_t)(nc | 1) * sizeof(T);
    if (!gather_frame_fits<T>(nc)) return fail(PDS_ERR_UNSUPPORTED, "launch_gather_frame: the transposition tile exceeds the LDS of one launch");
    const int64_t nblk = (n + kAosRows - 1) / kAosRows;
    const int nb1 = (int)std::min<int64_t>(std::max<int64_t>(nblk, 1), (int64_t)ctx->num_cus * 8);
    hipLaunchKernelGGL((cols_to_rows_kernel<T>), dim3(nb1), dim3(256), lds, ctx->stream, d_src, nc, n, d_records);
    const int nb2 = (int)std::min<int64_t>(std::max<int64_t>((n + 255) / 256, 1), (int64_t)ctx->num_cus * 32);
    hipLaunchKernelGGL((gather_records_kernel<T>), dim3(nb2), dim3(256), 0, ctx->stream, (const T*)d_records, d_perm, nc, n, d_dst);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}
template int launch_gather_frame<double>(pds_ctx*, const double* const*, const uint32_t*, int, int64_t, double*, double* const*);
template int launch_gather_frame<float>(pds_ctx*, const float* const*, const uint32_t*, int, int64_t, float*, float* const*);

// weighted groups: x_c * sqrt(w) (and sqrt(w) itself as the bias column) turn X' W X into a plain Gram matrix
template <typename T>
__global__ __launch_bounds__(256) void scale_sqrt_w_kernel(const T* __restrict__ src /*nullable: the ones column*/,
                                                           const T* __restrict__ w, int64_t n, T* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T s = (T)sqrt((double)w[i]);
        dst[i] = src ? src[i] * s : s;
    }
}
template <typename T>
int launch_scale_sqrt_w(pds_ctx* ctx, const T* d_src, const T* d_w, int64_t n, T* d_dst) {
    const int nb = (int)std::min<int64_t>(std::max<int64_t>((n + 255) / 256, 1), (int64_t)ctx->num_cus * 32);
    hipLaunchKernelGGL((scale_sqrt_w_kernel<T>), dim3(nb), dim3(256), 0, ctx->stream, d_src, d_w, n, d_dst);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}
template int launch_scale_sqrt_w<double>(pds_ctx*, const double*, const double*, int64_t, double*);
template int launch_scale_sqrt_w<float>(pds_ctx*, const float*, const float*, int64_t, float*);

template int launch_gather_rows<double>(pds_ctx*, const double*, const uint32_t*, int64_t, double*);
template int launch_gather_rows<float>(pds_ctx*, const float*, const uint32_t*, int64_t, float*);

}  // namespace pds
