// keyed_partition.hip -- grouped fits of a frame whose rows are in ANY order, without sorting the rows.
//
// keyed.hip brings such a frame into key order: a radix sort of (key, row) pairs, a transposition of the frame to row-major
// records and a gather of the records through the permutation -- 1e8 random 72-byte reads, which run at the random-access
// rate of the HBM stacks (6.75 of the 14.7 ms of the 1e6-group x ~100-row x 8-feature frame, profiles/r02_keyed_*).  A group's
// fit does not need its rows in order, or even adjacent: it needs their MOMENTS.  So, for integer keys whose range is not
// much wider than the frame is long (group ids: the usual case; anything else keeps the sorting route):
//
//   1. histogram   dense id = key - base (base: the smallest key rounded down to a bucket boundary); bucket = id >> shift, a bucket =
//                  the 2^shift ids whose moment records the accumulate kernel holds in its registers; rows counted per (bucket,
//                  stream), stream = the XCD a block runs on (its index mod 8).  Normally taken along by the order check of keyed.hip
//                  (same buckets: key >> shift); part_hist_kernel is the separate pass for key buffers that are not 16-byte aligned
//                  or ranges beyond kKeySlots buckets;
//   2. scatter     ONE pass over the frame: every row becomes a record [x_0 .. x_{p-1}, y | id in bucket, row] appended to its
//                  (bucket, stream) region.  A region is written front to back by the blocks of one XCD only, so its open
//                  cache line sits in that XCD's L2 until it is full.  A wave stages its 64 records in LDS and writes them
//                  piece-cooperatively (the lanes that share a record write its 16-byte pieces side by side);
//   3. accumulate  a workgroup streams (a chunk of) one bucket's records in tiles, counting-sorts each tile by id in LDS and lets
//                  the thread(s) that own an id walk that id's records: the moment records (upper triangles of z z', z = [x, 1, y])
//                  live in registers; they go to the id-indexed table in HBM through LDS -- plain stores when the bucket has one
//                  chunk (the usual case: its rows of the table need no memset), atomics where several chunks meet;
//   4. compact     ids with rows -> groups in ascending key order: distinct keys, sizes, offsets;
//   5. solve       the register solver of solve_reg.hip reads the table's packed triangles through the id list (OLS / ridge); the
//                  other methods take chunks of triangles expanded to (p+2)^2 records (part_expand_kernel).
//   Per-row predictions (grouped_pred.hip MODE 2) look a row's group up as rank[key - base]: nothing is ever permuted.
//
// Traffic: the frame once (scatter in), records out and in: 3.9x the frame by the PMC counters (profiles/r03_keyed_traffic.json)
// instead of ~5x plus a random-access pass.  Records arrive in their regions in the order of the scatter's cursor atomics: results
// agree with the sorting route to rounding (tests hold both to 1e-10 against the oracle), not bit for bit, and not run to run.
#include <hipcub/hipcub.hpp>
#include <type_traits>

#include "common.hpp"

namespace pds {

namespace {

constexpr int kPartStreams = 8;        // one per XCD
constexpr int kPartChunkRows = 4096;   // rows a block takes at a time (histogram and scatter use the same chunk -> stream map)
constexpr int kPartLdsBytes = 152 * 1024;  // of the 160 KB of a CU
constexpr int kPartAccumChunk = 32768; // records per accumulate workgroup
constexpr int kAccumThreads = 256;     // four waves = (id groups of 64) x (parts of an id's moment record); two workgroups per CU, up to 256 registers per lane

__host__ __device__ constexpr int tri_count(int q) { return q * (q + 1) / 2; }
__host__ __device__ constexpr int tri_index(int i, int j, int q) { return i * q - (i * (i - 1)) / 2 + (j - i); }  // i <= j

// threads that share one id's moment record in the accumulate kernel (NV entries of the upper triangle)
__host__ __device__ constexpr int accum_tpi(int nv) { return nv <= 55 ? 1 : (nv <= 110 ? 2 : 4); }

struct PartLayout {
    int pc;        // padded feature count (compile-time variant of the accumulate kernel)
    int qp;        // pc + 2
    int nv;        // tri_count(qp)
    int nvp;       // nv | 1: odd stride of a moment record in LDS / in the table
    int meta_off;  // byte offset of {u32 id in bucket, u32 row} in a record
    int rs;        // record bytes (multiple of 16)
    int shift;     // log2(ids per bucket)
};

inline int padded_features(int p) {
    static const int steps[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16};
    for (int s : steps)
        if (p <= s) return s;
    return -1;
}

template <typename T>
PartLayout make_layout(int p) {
    PartLayout L;
    L.pc = padded_features(p);
    L.qp = L.pc + 2;
    L.nv = tri_count(L.qp);
    L.nvp = L.nv | 1;
    L.meta_off = ((L.pc + 1) * (int)sizeof(T) + 7) & ~7;
    L.rs = (L.meta_off + 8 + 15) & ~15;
#ifdef PDS_PART_RS_ALIGN32
    L.rs = (L.rs + 31) & ~31;  // (experiment: whole 32-byte sectors per record)
#endif
    // ids per bucket: the accumulate kernel keeps their moment records in the registers of its 1024 threads, TPI threads per id
    int shift = 0;
    while ((2 << shift) <= kAccumThreads / accum_tpi(L.nv)) ++shift;
#ifdef PDS_DEV_SWITCHES  // (development builds only: EXTRA=-DPDS_DEV_SWITCHES; the accumulate launch refuses a foreign bucket width)
    if (const char* e = dev_env("PDS_PART_SHIFT")) shift = std::atoi(e);
#endif
    L.shift = shift;
    return L;
}

// ---- 1. histogram: counts[bucket * 8 + stream]
__global__ __launch_bounds__(256) void part_hist_kernel(const int64_t* __restrict__ keys, int64_t n, const int64_t* __restrict__ kmin,
                                                        int shift, int64_t n_buckets, unsigned* __restrict__ counts) {
    extern __shared__ unsigned hist_lds[];
    for (int64_t b = threadIdx.x; b < n_buckets; b += 256) hist_lds[b] = 0u;
    __syncthreads();
    const uint64_t base = (uint64_t)*kmin;
    const int64_t nchunks = (n + kPartChunkRows - 1) / kPartChunkRows;
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int64_t r0 = c * kPartChunkRows, r1 = (r0 + kPartChunkRows < n) ? r0 + kPartChunkRows : n;
        for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
            const uint64_t id = (uint64_t)__builtin_nontemporal_load(keys + r) - base;
            atomicAdd(&hist_lds[id >> shift], 1u);
        }
    }
    __syncthreads();
    const int stream = blockIdx.x & (kPartStreams - 1);
    for (int64_t b = threadIdx.x; b < n_buckets; b += 256) {
        const unsigned v = hist_lds[b];
        if (v) atomicAdd(&counts[b * kPartStreams + stream], v);
    }
}

// ---- 1'. the histogram the order check took along (keyed.hip: slot = (key >> shift) mod kKeySlots) -> counts[bucket * 8 + stream]
__global__ __launch_bounds__(256) void part_counts_from_slots_kernel(const unsigned* __restrict__ slots, unsigned first_slot, int64_t n_buckets,
                                                                     unsigned* __restrict__ counts) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_buckets * kPartStreams; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / kPartStreams;
        counts[i] = slots[(((unsigned)b + first_slot) & (unsigned)(kKeySlots - 1)) * kPartStreams + (unsigned)(i - b * kPartStreams)];
    }
}

// ---- 2. scatter.  PPR = 16-byte pieces per record.
// (Round 5, measured and parked -- tools/experiments/keyed_partition_scatter_block_reserve.hip.txt: a block counts its 16 384 rows per
//  bucket in LDS, reserves each bucket's space with ONE global atomic and places its rows through LDS atomics, so a bucket's ~4
//  records per block land side by side.  It does cut the write amplification -- 12.15 -> 10.4 GB written for 8.0 GB of records -- and is
//  SLOWER: 6.7 against 4.6 ms.  Two passes over a block's rows between block-wide barriers expose the latency that 24 independent waves
//  per CU hide here; the scatter is not bound by the bytes it writes.  The other direction -- no LDS stage at all, every lane writes
//  its own record's pieces, 32 / 48 / 64 waves per CU -- costs 7.2 ms whatever the occupancy: 5e8 lane-private 16-byte stores are the
//  slower way into the L2, the piece-cooperative write (a record's pieces side by side in adjacent lanes) stays.
//  profiles/r05_scatter_block_reserve_ab.txt.)
template <typename T, int PPR>
__global__ __launch_bounds__(256) void part_scatter_kernel(const T* const* __restrict__ cols /*x_0..x_{p-1}, y*/, int p, int pc,
                                                           const int64_t* __restrict__ keys, int64_t n,
                                                           const int64_t* __restrict__ kmin, int shift, int meta_off,
                                                           unsigned* __restrict__ cursor, char* __restrict__ records) {
    constexpr int RS = PPR * 16;
    __shared__ __attribute__((aligned(16))) char stage[4][64 * RS];
    __shared__ unsigned pos_lds[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    char* st = stage[wv];
    const uint64_t base = (uint64_t)*kmin;
    const unsigned idmask = (1u << shift) - 1u;
    const int stream = blockIdx.x & (kPartStreams - 1);
    const int64_t nchunks = (n + kPartChunkRows - 1) / kPartChunkRows;
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int64_t r0 = c * kPartChunkRows, r1 = (r0 + kPartChunkRows < n) ? r0 + kPartChunkRows : n;
        for (int64_t rb = r0 + (int64_t)wv * 64; rb < r1; rb += 256) {
            const int64_t r = rb + lane;
            const bool in = r < r1;
            // ---- the lane's row -> its record in the wave's LDS stage
            if (in) {
                const uint64_t id = (uint64_t)__builtin_nontemporal_load(keys + r) - base;
                const unsigned pos = atomicAdd(&cursor[(id >> shift) * kPartStreams + stream], 1u);
                pos_lds[wv][lane] = pos;
                T* rec = reinterpret_cast<T*>(st + lane * RS);
                for (int cc = 0; cc < p; ++cc) rec[cc] = __builtin_nontemporal_load(as_global(cols[cc]) + r);
                for (int cc = p; cc < pc; ++cc) rec[cc] = T(0);
                rec[pc] = __builtin_nontemporal_load(as_global(cols[p]) + r);
                unsigned* meta = reinterpret_cast<unsigned*>(st + lane * RS + meta_off);
                meta[0] = (unsigned)id & idmask;
                meta[1] = (unsigned)r;
            }
            PDS_WAVE_LDS_SYNC();
            // ---- piece-cooperative write: piece j of the wave's stage belongs to record j / PPR
            const int nrec = (int)((r1 - rb < 64) ? r1 - rb : 64);
#pragma unroll
            for (int k = 0; k < PPR; ++k) {
                const int j = k * 64 + lane;
                const int rec = j / PPR, piece = j - rec * PPR;
                if (rec < nrec) {
                    const uint4 v = *reinterpret_cast<const uint4*>(st + j * 16);
                    *reinterpret_cast<uint4*>(records + (size_t)pos_lds[wv][rec] * RS + piece * 16) = v;
                }
            }
            PDS_WAVE_LDS_SYNC();
        }
    }
}

// ---- 3. accumulate.  One workgroup per (bucket, chunk of kPartAccumChunk records).
// The moment records of a bucket's ids live in REGISTERS: TPI threads own one id (EPT = ceil(NV / TPI) entries of its upper triangle
// each -- up to 8 features ONE thread holds all 55); the waves of a workgroup are (id group of 64) x (part), so every lane of a
// wave runs the same entries and the code of a part is straight-line.  (Sixteen waves with four threads per id read every record
// four times: the random 16-byte LDS reads cost ~20 clk per wave instruction in bank conflicts and bounded the pass at 3.1 ms.)  A tile of TILE records is staged in LDS (the next one waits in registers), counting-sorted by id there
// (one returning LDS atomic per record, a scan over the ids, one 16-bit slot per record), and every owner walks its id's records:
// ~QP LDS reads and EPT FMAs per record and part -- against NV LDS atomics per record in the first version (55 ds_add_f64 at 8
// features: ~3 lanes / clk with same-id lanes serialising, 4.6 ms of the 11.3 ms of the shuffled C3 frame; tools/valu_lds_rate.hip).
// An id with more than kHeavy records in a tile (a giant group) leaves the owners' loop: the whole workgroup forms its records'
// products, reduces them per wave and hands the sum to the owners through an LDS scratch.
// Tried and slower (tools/experiments/): the matrix cores on id-sorted half-tiles (keyed_partition_accum_mfma: 8.8 ms); wave-owned
// ids with register-resident records (keyed_partition_accum_wave_owned_ids: 8.3 ms); LDS atomics (keyed_partition_accum_lds_atomics).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__host__ __device__ constexpr int accum_tile(int rs) { return rs <= 96 ? 512 : 256; }
constexpr int kHeavy = 64;
// ids whose records go through LDS to the table at a time (64, or 32 where 64 would need more LDS than the tile does)
__host__ __device__ constexpr int accum_flush_ids(int nv, int rs) {
    return (size_t)64 * (nv | 1) * 8 <= (size_t)accum_tile(rs) * rs + 4096 ? 64 : 32;
}
__host__ __device__ constexpr size_t accum_lds_bytes(int nv, int rs) {
    const int tpi = accum_tpi(nv), ids = kAccumThreads / tpi, tile = accum_tile(rs), max_heavy = tile / kHeavy;
    const size_t lists = (size_t)(ids + ids + 1 + 16 + max_heavy + 1) * 4 + (size_t)tile * 2 + 16 + (size_t)nv * 8;
    const size_t work = (size_t)tile * rs + lists, flush = (size_t)accum_flush_ids(nv, rs) * (nv | 1) * 8;
    return (work > flush ? work : flush) + 16;
}

template <typename T, int PC, int PART, int EPT>
__device__ __forceinline__ void accum_record(const char* rec, double (&acc)[EPT]) {
    constexpr int QP = PC + 2;
    const T* vals = reinterpret_cast<const T*>(rec);
    double z[QP];
#pragma unroll
    for (int c = 0; c < PC; ++c) z[c] = (double)vals[c];
    z[PC] = 1.0;
    z[PC + 1] = (double)vals[PC];
    int v = 0;
#pragma unroll
    for (int a = 0; a < QP; ++a) {
#pragma unroll
        for (int b = a; b < QP; ++b) {
            if (v >= PART * EPT && v < (PART + 1) * EPT) acc[v - PART * EPT] = fma(z[a], z[b], acc[v - PART * EPT]);
            ++v;
        }
    }
}

template <typename T, int PC, int PPR>
__global__ __launch_bounds__(kAccumThreads) void part_accum_kernel(const char* __restrict__ records, const unsigned* __restrict__ bucket_start /*n_buckets * 8 + 1*/,
                                                                   const unsigned* __restrict__ chunk_prefix /*n_buckets + 1*/, int64_t n_buckets,
                                                                   int meta_off, double* __restrict__ table, int debug) {
    constexpr int QP = PC + 2, NV = tri_count(QP), NVP = NV | 1, RS = PPR * 16;
    constexpr int TPI = accum_tpi(NV), EPT = (NV + TPI - 1) / TPI, IDS = kAccumThreads / TPI, TILE = accum_tile(RS);
    constexpr int KPT = (PPR * TILE + kAccumThreads - 1) / kAccumThreads;  // 16-byte pieces of the next tile a thread holds
    constexpr int kMaxHeavy = TILE / kHeavy;
    extern __shared__ __attribute__((aligned(16))) char acc_lds[];
    char* tile = acc_lds;
    unsigned* cnt = reinterpret_cast<unsigned*>(acc_lds + (size_t)TILE * RS);
    unsigned* off = cnt + IDS;          // IDS + 1
    unsigned* wtot = off + IDS + 1;     // 16
    unsigned* heavy = wtot + 16;        // kMaxHeavy ids, then their count
    unsigned short* sorted = reinterpret_cast<unsigned short*>(heavy + kMaxHeavy + 1);
    double* scratch = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(sorted + TILE) + 7) & ~(uintptr_t)7);
    double* mom = reinterpret_cast<double*>(acc_lds);  // (the flush at the end reuses the tile)
    const unsigned total_chunks = chunk_prefix[n_buckets];
    const unsigned w = blockIdx.x;
    if (w >= total_chunks) return;
    // bucket of work item w: last b with chunk_prefix[b] <= w
    int64_t lo = 0, hi = n_buckets;
    while (hi - lo > 1) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (chunk_prefix[mid] <= w) lo = mid;
        else hi = mid;
    }
    const int64_t bucket = lo;
    const unsigned chunk = w - chunk_prefix[bucket];
    const bool single = chunk_prefix[bucket + 1] - chunk_prefix[bucket] == 1u;
    const int64_t b0 = bucket_start[bucket * kPartStreams], b1 = bucket_start[(bucket + 1) * kPartStreams];
    const int64_t r0 = b0 + (int64_t)chunk * kPartAccumChunk, r1 = (r0 + kPartAccumChunk < b1) ? r0 + kPartAccumChunk : b1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int part = __builtin_amdgcn_readfirstlane(wv % TPI);
    const int my = (wv / TPI) * 64 + lane;  // the id (within the bucket) this thread owns a part of
    for (int i = tid; i < IDS; i += kAccumThreads) cnt[i] = 0u;
    if (tid < NV) scratch[tid] = 0.0;
    static_assert(NV <= kAccumThreads && TILE % kAccumThreads == 0 && TPI <= 4, "scratch is cleared by one thread per entry");
    double acc[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) acc[e] = 0.0;
    u32x4 nxt[KPT];
    auto fetch = [&](int64_t base) __attribute__((always_inline)) {
        const int np = (int)((r1 - base < TILE) ? r1 - base : TILE) * PPR;
        const u32x4* src = reinterpret_cast<const u32x4*>(records + (size_t)base * RS);
        // unconditional loads from a clamped index: a load under its own exec mask made the compiler wait for the previous one
        // (vmcnt(0) in front of every address computation) -- five HBM round trips in a row per tile, 1.7 of the kernel's 3.6 ms
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const int j = k * kAccumThreads + tid;
            nxt[k] = __builtin_nontemporal_load(src + (j < np ? j : np - 1));
        }
    };
    if (r0 < r1) fetch(r0);
    for (int64_t base = r0; base < r1; base += TILE) {
        const int nrec = (int)((r1 - base < TILE) ? r1 - base : TILE);
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const int j = k * kAccumThreads + tid;
            if (j < nrec * PPR) reinterpret_cast<u32x4*>(tile)[j] = nxt[k];
        }
        if (tid == 0) heavy[kMaxHeavy] = 0u;
        __syncthreads();  // (A) the tile is in LDS; the counters are zero
        if (base + TILE < r1) fetch(base + TILE);
        // ---- counting sort of the tile's records by id
        constexpr int RPT = TILE / kAccumThreads;  // records a thread files
        unsigned lid[RPT], rank[RPT];
        if (debug & 2) continue;  // (experiment: stream the records only)
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int t = q * kAccumThreads + tid;
            lid[q] = rank[q] = 0u;
            if (t < nrec) {
                lid[q] = *reinterpret_cast<const unsigned*>(tile + (size_t)t * RS + meta_off);
                rank[q] = atomicAdd(&cnt[lid[q]], 1u);
            }
        }
        __syncthreads();  // (B)
        unsigned c = 0u, incl = 0u;
        if (tid < IDS) {
            c = cnt[tid];
            incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned t = __shfl_up(incl, d);
                if (lane >= d) incl += t;
            }
            if (lane == 63) wtot[wv] = incl;
        }
        __syncthreads();  // (C)
        if (tid < IDS) {
            unsigned o = 0u;
            for (int ww = 0; ww < wv; ++ww) o += wtot[ww];
            off[tid] = o + incl - c;
            if (tid == IDS - 1) off[IDS] = o + incl;
            cnt[tid] = 0u;  // (read above; next written behind the next (A))
        }
        __syncthreads();  // (D)
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int t = q * kAccumThreads + tid;
            if (t < nrec) sorted[off[lid[q]] + rank[q]] = (unsigned short)t;
        }
        __syncthreads();  // (E)
        // ---- every owner walks its id's records
        {
            const unsigned k0 = off[my], k1 = off[my + 1];
            if (k1 - k0 > (unsigned)kHeavy) {
                if (part == 0) heavy[atomicAdd(&heavy[kMaxHeavy], 1u)] = (unsigned)my;
            } else if (k0 < k1 && !(debug & 1)) {  // (debug 1 -- experiment: everything but the owners' loop)
                auto run = [&](auto part_c) __attribute__((always_inline)) {
                    constexpr int PART = decltype(part_c)::value;
                    unsigned r = sorted[k0];
                    for (unsigned k = k0; k < k1; ++k) {
                        const unsigned rn = sorted[k + 1 < k1 ? k + 1 : k];
                        accum_record<T, PC, PART, EPT>(tile + (size_t)r * RS, acc);
                        r = rn;
                    }
                };
                switch (part) {
#define PDS_PART_CASE(N) case N: if constexpr (TPI > N) run(std::integral_constant<int, N>{}); break;
                    PDS_PART_CASE(0) PDS_PART_CASE(1) PDS_PART_CASE(2) PDS_PART_CASE(3)
#undef PDS_PART_CASE
                    default: break;
                }
            }
        }
        __syncthreads();  // (F) everybody is done with the tile -- but for the heavy ids
        const unsigned nh = heavy[kMaxHeavy];
        for (unsigned i = 0; i < nh; ++i) {
            const unsigned h = heavy[i], hk0 = off[h], hk1 = off[h + 1];
            for (unsigned kb = hk0; kb < hk1; kb += kAccumThreads) {
                if (kb + (unsigned)wv * 64 >= hk1) continue;  // (wave-uniform)
                const unsigned k = kb + tid;
                const bool in = k < hk1;
                // (rare path: run-time loops, values straight from LDS -- no register arrays next to the owners' accumulators)
                const T* vals = reinterpret_cast<const T*>(tile + (size_t)sorted[in ? k : hk0] * RS);
                auto zval = [&](int idx) __attribute__((always_inline)) {
                    return !in ? 0.0 : (idx == PC ? 1.0 : (double)vals[idx < PC ? idx : PC]);
                };
                int v = 0;
#pragma unroll 1
                for (int a = 0; a < QP; ++a) {
                    const double za = zval(a);
#pragma unroll 1
                    for (int b = a; b < QP; ++b) {
                        double pv = za * zval(b);
#pragma unroll
                        for (int o = 32; o >= 1; o >>= 1) pv += __shfl_xor(pv, o);
                        if (lane == 0) __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)(scratch + v), pv);
                        ++v;
                    }
                }
            }
            __syncthreads();
            if ((unsigned)my == h) {
#pragma unroll
                for (int e = 0; e < EPT; ++e)
                    if (part * EPT + e < NV) acc[e] += scratch[part * EPT + e];
            }
            __syncthreads();
            if (tid < NV) scratch[tid] = 0.0;
            __syncthreads();
        }
    }
    __syncthreads();
    // ---- the owners' registers -> LDS, 64 ids at a time -> the table (several chunks of one bucket, and nobody else, meet here)
    constexpr int CNT = tri_index(PC, PC, QP);
    constexpr int FID = accum_flush_ids(NV, RS);
    for (int id0 = 0; id0 < IDS; id0 += FID) {
        if (my >= id0 && my < id0 + FID) {
            const int l = my - id0;
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                if (part * EPT + e < NV) mom[l * NVP + part * EPT + e] = acc[e];
            if (part == 0 && NVP > NV) mom[l * NVP + NV] = 0.0;
        }
        __syncthreads();
        double* tb = table + ((size_t)bucket * IDS + (size_t)id0) * NVP;
        if (single) {  // the bucket's only workgroup: plain stores of every entry (ids without rows: zeros) -- its rows need no memset
            for (int i = tid; i < FID * NVP; i += kAccumThreads) tb[i] = mom[i];
        } else {
            for (int i = tid; i < FID * NVP; i += kAccumThreads) {
                const int l = i / NVP;
                if (mom[l * NVP + CNT] > 0.0) {
                    const double v = mom[i];
                    if (v != 0.0) unsafeAtomicAdd(tb + i, v);
                }
            }
        }
        __syncthreads();
    }
}

// table rows of the buckets that are NOT written by exactly one accumulate workgroup (no records at all, or several chunks that meet
// through atomics) start from zero; the usual bucket -- one chunk -- is overwritten whole by its workgroup instead (a memset of the
// whole id-indexed table was 0.1 ms of the C3 frame's 7.7)
__global__ __launch_bounds__(256) void part_zero_rows_kernel(double* __restrict__ table, const unsigned* __restrict__ chunks, int64_t n_buckets,
                                                             int64_t row_doubles) {
    for (int64_t b = blockIdx.x; b < n_buckets; b += gridDim.x) {
        if (chunks[b] == 1u) continue;
        double* t = table + (size_t)b * row_doubles;
        for (int64_t i = threadIdx.x; i < row_doubles; i += 256) t[i] = 0.0;
    }
}

// ---- 4. compaction helpers
template <int DUMMY = 0>
__global__ __launch_bounds__(256) void part_flags_kernel(const double* __restrict__ table, int64_t n_ids, int nvp, int cnt_index,
                                                         unsigned* __restrict__ flags) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_ids; i += (int64_t)gridDim.x * 256)
        flags[i] = table[(size_t)i * nvp + cnt_index] > 0.0 ? 1u : 0u;
}
__global__ __launch_bounds__(256) void part_compact_kernel(const double* __restrict__ table, int64_t n_ids, int nvp, int cnt_index,
                                                           const unsigned* __restrict__ rank, const int64_t* __restrict__ kmin,
                                                           int64_t max_groups, int64_t* __restrict__ out_keys,
                                                           unsigned* __restrict__ ids, int64_t* __restrict__ sizes) {
    const uint64_t base = (uint64_t)*kmin;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_ids; i += (int64_t)gridDim.x * 256) {
        const double c = table[(size_t)i * nvp + cnt_index];
        if (c > 0.0) {
            const unsigned r = rank[i];
            if ((int64_t)r < max_groups) {
                out_keys[r] = (int64_t)(base + (uint64_t)i);
                ids[r] = (unsigned)i;
                sizes[r] = (int64_t)(c + 0.5);
            }
        }
    }
}
__global__ void part_close_offsets_kernel(int64_t* __restrict__ off, int64_t ng, int64_t n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) off[ng] = n;
}
__global__ __launch_bounds__(256) void part_chunk_counts_kernel(const unsigned* __restrict__ bucket_start, int64_t n_buckets,
                                                                unsigned* __restrict__ chunks) {
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < n_buckets; b += (int64_t)gridDim.x * 256) {
        const unsigned sz = bucket_start[(b + 1) * kPartStreams] - bucket_start[b * kPartStreams];
        chunks[b] = (sz + kPartAccumChunk - 1) / kPartAccumChunk;
    }
}

// ---- 5. table rows (upper triangles over [x_0..x_{PC-1}, 1, y]) -> (p+2)^2 records over [x_0..x_{p-1}, 1, y]
template <typename T>
__global__ __launch_bounds__(256) void part_expand_kernel(const double* __restrict__ table, const unsigned* __restrict__ ids, int64_t g0,
                                                          int64_t gc, int p, int pc, int nvp, T* __restrict__ recs) {
    const int q = p + 2, qp = pc + 2;
    const int64_t total = gc * q * q;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t g = e / (q * q);
        const int ij = (int)(e - g * q * q);
        int i = ij % q, j = ij / q;
        if (i > j) {
            const int t = i;
            i = j;
            j = t;
        }
        const int pi = i < p ? i : pc + (i - p), pj = j < p ? j : pc + (j - p);
        recs[e] = (T)table[(size_t)ids[g0 + g] * nvp + tri_index(pi, pj, qp)];
    }
}

template <typename T, int PPR>
void launch_scatter(dim3 g, hipStream_t st, const T* const* cols, int p, const PartLayout& L, const int64_t* keys, int64_t n,
                    const int64_t* kmin, unsigned* cursor, char* records) {
    hipLaunchKernelGGL((part_scatter_kernel<T, PPR>), g, dim3(256), 0, st, cols, p, L.pc, keys, n, kmin, L.shift, L.meta_off, cursor, records);
}
template <typename T, int PC>
int launch_accum(pds_ctx* ctx, unsigned grid, const PartLayout& L, const char* records, const unsigned* bucket_start,
                 const unsigned* chunk_prefix, int64_t n_buckets, double* table) {
#ifdef PDS_PART_RS_ALIGN32
    constexpr int PPR = (((((((PC + 1) * (int)sizeof(T) + 7) & ~7) + 8 + 15) & ~15) + 31) & ~31) / 16;
#else
    constexpr int PPR = ((((PC + 1) * (int)sizeof(T) + 7) & ~7) + 8 + 15) / 16;  // = make_layout<T>(PC).rs / 16
#endif
    if (L.rs != PPR * 16) return fail(PDS_ERR_INVALID, "internal: record size");
    constexpr size_t lds = accum_lds_bytes(tri_count(PC + 2), PPR * 16);
    static_assert(lds <= (size_t)kPartLdsBytes / 2, "two workgroups share the CU's LDS");
    if ((1 << L.shift) != kAccumThreads / accum_tpi(tri_count(PC + 2))) return fail(PDS_ERR_INVALID, "internal: ids per bucket");
    auto kern = part_accum_kernel<T, PC, PPR>;
    if (lds > 64 * 1024) PDS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#ifdef PDS_DEV_SWITCHES  // timing experiments of development builds (EXTRA=-DPDS_DEV_SWITCHES): the results are wrong with it
    const char* dbg = dev_env("PDS_PART_DEBUG");
#else
    const char* dbg = nullptr;
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kAccumThreads), lds, ctx->stream, records, bucket_start, chunk_prefix, n_buckets, L.meta_off, table,
                       dbg ? std::atoi(dbg) : 0);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}

}  // namespace

// Is the partition route applicable?  range = max - min + 1 of the keys (host); returns the bucket count or 0.
template <typename T>
int64_t keyed_partition_buckets(int n_feat, int64_t n_rows, uint64_t range) {
    if (n_feat < 1 || n_feat > 16 || n_rows < (int64_t)1 << 16 || n_rows >= ((int64_t)1 << 31)) return 0;
    if (range == 0 || range > ((uint64_t)1 << 31) || range > (uint64_t)n_rows * 4) return 0;  // sparse keys: the sorting route
    const PartLayout L = make_layout<T>(n_feat);
    const int64_t nb = (int64_t)((range + ((uint64_t)1 << L.shift) - 1) >> L.shift);
    if (nb > 32768) return 0;  // (the histogram of one block lives in LDS: 128 KB of counters)
    // the id-indexed moment table (one upper triangle per POSSIBLE key) must stay within twice the frame's own size and 16 GiB
    const uint64_t table_bytes = ((uint64_t)nb << L.shift) * (uint64_t)L.nvp * 8;
    if (table_bytes > 2 * (uint64_t)n_rows * (uint64_t)(n_feat + 1) * sizeof(T) || table_bytes > ((uint64_t)16 << 30)) return 0;
    return nb;
}
template int64_t keyed_partition_buckets<double>(int, int64_t, uint64_t);
template int64_t keyed_partition_buckets<float>(int, int64_t, uint64_t);

template <typename T>
size_t keyed_partition_workspace(int n_feat, int64_t n_rows, int64_t n_buckets) {
    const PartLayout L = make_layout<T>(n_feat);
    const size_t ids = (size_t)n_buckets << L.shift;
    size_t temp = 0, t2 = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, temp, (const unsigned*)nullptr, (unsigned*)nullptr, (int)std::min<size_t>(ids + 1, INT32_MAX));
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t2, (const int64_t*)nullptr, (int64_t*)nullptr, (int)std::min<size_t>(ids + 1, INT32_MAX));
    temp = std::max(temp, t2);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return up((size_t)n_rows * L.rs) + up(ids * L.nvp * 8) + 3 * up(((size_t)n_buckets * kPartStreams + 1) * 4) + 2 * up(((size_t)n_buckets + 1) * 4) +
           2 * up((ids + 1) * 4) + up(ids * 4) + up(temp) + 8192;
}
template size_t keyed_partition_workspace<double>(int, int64_t, int64_t);
template size_t keyed_partition_workspace<float>(int, int64_t, int64_t);

// The whole route up to the moment table and the group list.  d_cols: DEVICE table in kernel order (x_0..x_{p-1}, y).
//   out: *n_groups (host; a stream synchronisation), d_out_keys / d_sizes_as_offsets (n_groups + 1, exclusive scan + n) for up to
//   max_groups groups, and the handles the solve stage needs (table, ids, layout) in `st`.
template <typename T>
int keyed_partition_shift(int n_feat) { return make_layout<T>(n_feat).shift; }
template int keyed_partition_shift<double>(int);
template int keyed_partition_shift<float>(int);

// d_kmin: the BASE of the dense ids on the device -- the smallest key rounded down to a multiple of the bucket width, so that
// (key - base) >> shift = (key >> shift) - (base >> shift): the order check's histogram (d_slot_counts, nullable; first_slot =
// (base >> shift) mod kKeySlots) uses the same buckets
template <typename T>
int keyed_partition_build(pds_ctx* ctx, const T* const* d_cols, const int64_t* d_keys, const int64_t* d_kmin, uint64_t range, int n_feat,
                          int64_t n_rows, int64_t n_buckets, char* ws, int64_t max_groups, int64_t* d_out_keys, int64_t* d_offsets,
                          int64_t* n_groups, KeyedPartitionState& st, const unsigned* d_slot_counts, unsigned first_slot, int phases,
                          int64_t n_rows_total) {
    // phases (capi_multi.hpp: one frame over several contexts): 1 = histogram .. accumulate only -- the id-indexed moment table of
    // THESE rows, complete (st.table; tables of row slices are additive: keyed_partition_add_table) -- 2 = the group list from the
    // table that is there (same workspace: the carving below is the same); 3 = both, the single-context route.
    const PartLayout L = make_layout<T>(n_feat);
    const size_t ids = (size_t)n_buckets << L.shift;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    auto take = [&](size_t b) { char* r = ws; ws += up(b); return r; };
    char* records = take((size_t)n_rows * L.rs);
    double* table = reinterpret_cast<double*>(take(ids * L.nvp * 8));
    const size_t ncnt = (size_t)n_buckets * kPartStreams + 1;
    unsigned* counts = reinterpret_cast<unsigned*>(take(ncnt * 4));
    unsigned* starts = reinterpret_cast<unsigned*>(take(ncnt * 4));
    unsigned* cursor = reinterpret_cast<unsigned*>(take(ncnt * 4));
    unsigned* chunks = reinterpret_cast<unsigned*>(take(((size_t)n_buckets + 1) * 4));
    unsigned* chunk_prefix = reinterpret_cast<unsigned*>(take(((size_t)n_buckets + 1) * 4));
    unsigned* flags = reinterpret_cast<unsigned*>(take((ids + 1) * 4));
    unsigned* rank = reinterpret_cast<unsigned*>(take((ids + 1) * 4));
    unsigned* id_of = reinterpret_cast<unsigned*>(take(ids * 4));
    size_t temp_bytes = 0, t2 = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (int)std::min<size_t>(ids + 1, INT32_MAX));
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t2, (const int64_t*)nullptr, (int64_t*)nullptr, (int)std::min<size_t>(ids + 1, INT32_MAX));
    temp_bytes = std::max(temp_bytes, t2);
    void* d_temp = take(temp_bytes);
    hipStream_t s = ctx->stream;
    st.table = table;
    st.ids = id_of;
    st.rank = rank;
    st.pc = L.pc;
    st.nvp = L.nvp;
    const int cnt_index = tri_index(L.pc, L.pc, L.qp);
    const int fb = (int)std::min<int64_t>(std::max<int64_t>(((int64_t)ids + 255) / 256, 1), (int64_t)ctx->num_cus * 16);
    if (phases & 1) {
    if (!d_slot_counts) PDS_HIP_CHECK(hipMemsetAsync(counts, 0, ncnt * 4, s));
    else PDS_HIP_CHECK(hipMemsetAsync(counts + (ncnt - 1), 0, 4, s));
    // (the table is zeroed bucket by bucket where that is needed: part_zero_rows_kernel below)
    PDS_HIP_CHECK(hipMemsetAsync(chunks + n_buckets, 0, 4, s));
    PDS_HIP_CHECK(hipMemsetAsync(flags + ids, 0, 4, s));
    // ---- 1. histogram (a grid that is resident at once and a multiple of the stream count: block b runs on XCD b mod 8)
    const int64_t nchunks = (n_rows + kPartChunkRows - 1) / kPartChunkRows;
    const size_t hist_lds = (size_t)n_buckets * 4;
    if (hist_lds > 64 * 1024)
        PDS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(part_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_lds));
    int hb = (int)std::min<int64_t>(nchunks, (int64_t)ctx->num_cus * (hist_lds > 32 * 1024 ? 1 : 4));
    hb = std::max(kPartStreams, hb / kPartStreams * kPartStreams);
    if (d_slot_counts) {
        const int pb = (int)std::min<int64_t>((n_buckets * kPartStreams + 255) / 256, 1024);
        hipLaunchKernelGGL(part_counts_from_slots_kernel, dim3(pb), dim3(256), 0, s, d_slot_counts, first_slot, n_buckets, counts);
    } else {
        hipLaunchKernelGGL(part_hist_kernel, dim3(hb), dim3(256), hist_lds, s, d_keys, n_rows, d_kmin, L.shift, n_buckets, counts);
    }
    PDS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, (const unsigned*)counts, starts, (int)ncnt, s));
    PDS_HIP_CHECK(hipMemcpyAsync(cursor, starts, ncnt * 4, hipMemcpyDeviceToDevice, s));
    // ---- 2. scatter
    static const int scatter_bpc = [] { const char* e = dev_env("PDS_PART_SCATTER_BPC"); return e ? std::max(1, std::atoi(e)) : 6; }();  // (2: 7.1 ms, 4: 4.76, 6: 4.59, 8: 5.52 on the C3 frame -- tools/gpu_scatter_bpc.sh)
    int sb = (int)std::min<int64_t>(nchunks, (int64_t)ctx->num_cus * scatter_bpc);
    sb = std::max(kPartStreams, sb / kPartStreams * kPartStreams);
    const int ppr = L.rs / 16;
    switch (ppr) {
#define PDS_PART_PPR(N) case N: launch_scatter<T, N>(dim3(sb), s, d_cols, n_feat, L, d_keys, n_rows, d_kmin, cursor, records); break;
        PDS_PART_PPR(1) PDS_PART_PPR(2) PDS_PART_PPR(3) PDS_PART_PPR(4) PDS_PART_PPR(5) PDS_PART_PPR(6) PDS_PART_PPR(7) PDS_PART_PPR(8)
        PDS_PART_PPR(9) PDS_PART_PPR(10)
#undef PDS_PART_PPR
        default: return fail(PDS_ERR_UNSUPPORTED, "keyed partition: record too wide");
    }
    PDS_HIP_CHECK(hipGetLastError());
    // ---- 3. accumulate: work items = (bucket, chunk); their count is bounded on the host, surplus workgroups leave at once
    const int cb = (int)std::min<int64_t>(std::max<int64_t>((n_buckets + 255) / 256, 1), 1024);
    hipLaunchKernelGGL(part_chunk_counts_kernel, dim3(cb), dim3(256), 0, s, (const unsigned*)starts, n_buckets, chunks);
    PDS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, (const unsigned*)chunks, chunk_prefix, (int)n_buckets + 1, s));
    hipLaunchKernelGGL(part_zero_rows_kernel, dim3((unsigned)std::min<int64_t>(n_buckets, (int64_t)ctx->num_cus * 8)), dim3(256), 0, s, table,
                       (const unsigned*)chunks, n_buckets, (int64_t)(((size_t)1 << L.shift) * L.nvp));
    const unsigned max_items = (unsigned)(n_buckets + (n_rows + kPartAccumChunk - 1) / kPartAccumChunk);
    int rc = PDS_OK;
    switch (L.pc) {
#define PDS_PART_PC(N) case N: rc = launch_accum<T, N>(ctx, max_items, L, records, starts, chunk_prefix, n_buckets, table); break;
        PDS_PART_PC(1) PDS_PART_PC(2) PDS_PART_PC(3) PDS_PART_PC(4) PDS_PART_PC(5) PDS_PART_PC(6) PDS_PART_PC(7) PDS_PART_PC(8)
        PDS_PART_PC(10) PDS_PART_PC(12) PDS_PART_PC(14) PDS_PART_PC(16)
#undef PDS_PART_PC
        default: return fail(PDS_ERR_UNSUPPORTED, "keyed partition: feature count");
    }
    if (rc) return rc;
    }  // phases & 1
    if (!(phases & 2)) return PDS_OK;
    // ---- 4. groups = ids with rows, ascending
    PDS_HIP_CHECK(hipMemsetAsync(flags + ids, 0, 4, s));
    hipLaunchKernelGGL(part_flags_kernel<0>, dim3(fb), dim3(256), 0, s, (const double*)table, (int64_t)ids, L.nvp, cnt_index, flags);
    PDS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, (const unsigned*)flags, rank, (int)ids + 1, s));
    unsigned h_ng = 0;
    PDS_HIP_CHECK(hipMemcpyAsync(&h_ng, rank + ids, 4, hipMemcpyDeviceToHost, s));
    PDS_HIP_CHECK(hipStreamSynchronize(s));
    *n_groups = (int64_t)h_ng;
    if ((int64_t)h_ng > max_groups) return fail(PDS_ERR_INVALID, "more distinct keys than max_groups");
    int64_t* sizes = d_offsets;  // sizes are scanned in place into offsets
    hipLaunchKernelGGL(part_compact_kernel, dim3(fb), dim3(256), 0, s, (const double*)table, (int64_t)ids, L.nvp, cnt_index, (const unsigned*)rank,
                       d_kmin, max_groups, d_out_keys, id_of, sizes);
    PDS_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, (const int64_t*)sizes, d_offsets, (int)h_ng, s));
    hipLaunchKernelGGL(part_close_offsets_kernel, dim3(1), dim3(64), 0, s, d_offsets, (int64_t)h_ng, n_rows_total > 0 ? n_rows_total : n_rows);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}
template int keyed_partition_build<double>(pds_ctx*, const double* const*, const int64_t*, const int64_t*, uint64_t, int, int64_t, int64_t,
                                           char*, int64_t, int64_t*, int64_t*, int64_t*, KeyedPartitionState&, const unsigned*, unsigned, int, int64_t);
template int keyed_partition_build<float>(pds_ctx*, const float* const*, const int64_t*, const int64_t*, uint64_t, int, int64_t, int64_t, char*,
                                          int64_t, int64_t*, int64_t*, int64_t*, KeyedPartitionState&, const unsigned*, unsigned, int, int64_t);

// table += other (both id-indexed moment tables of the same layout, both readable from ctx's device): the exchange step of the
// row-sliced route -- moment tables of row slices are additive whatever the row order
__global__ __launch_bounds__(256) void part_add_table_kernel(double* __restrict__ dst, const double* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] += src[i];
}
int keyed_partition_add_table(pds_ctx* ctx, const KeyedPartitionState& st, int64_t n_ids, const double* d_other) {
    const size_t n = (size_t)n_ids * (size_t)st.nvp;
    const int nb = (int)std::min<size_t>((n + 255) / 256, (size_t)ctx->num_cus * 16);
    hipLaunchKernelGGL(part_add_table_kernel, dim3(nb), dim3(256), 0, ctx->stream, st.table, d_other, n);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}
template <typename T>
int64_t keyed_partition_table_ids(int n_feat, int64_t n_buckets) { return n_buckets << make_layout<T>(n_feat).shift; }
template int64_t keyed_partition_table_ids<double>(int, int64_t);
template int64_t keyed_partition_table_ids<float>(int, int64_t);

// (p+2)^2 moment records of groups [g0, g0 + gc) for the batched solvers
template <typename T>
int keyed_partition_records(pds_ctx* ctx, const KeyedPartitionState& st, int n_feat, int64_t g0, int64_t gc, T* d_records) {
    const int q = n_feat + 2;
    const int nb = (int)std::min<int64_t>(std::max<int64_t>((gc * q * q + 255) / 256, 1), (int64_t)ctx->num_cus * 16);
    hipLaunchKernelGGL((part_expand_kernel<T>), dim3(nb), dim3(256), 0, ctx->stream, (const double*)st.table, (const unsigned*)st.ids, g0, gc, n_feat,
                       st.pc, st.nvp, d_records);
    PDS_HIP_CHECK(hipGetLastError());
    return PDS_OK;
}
template int keyed_partition_records<double>(pds_ctx*, const KeyedPartitionState&, int, int64_t, int64_t, double*);
template int keyed_partition_records<float>(pds_ctx*, const KeyedPartitionState&, int, int64_t, int64_t, float*);

}  // namespace pds
