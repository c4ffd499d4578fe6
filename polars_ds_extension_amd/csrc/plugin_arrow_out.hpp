// plugin_arrow_out.hpp -- result Series: owned Arrow schemas / arrays (List, Struct), storage the device copies land in
// Part of the one translation unit plugin.cpp (included there, inside its anonymous namespace, in dependency order).
#pragma once

// ------------------------------------------------------------------------------------------------- Arrow export
struct OwnedSchema {
    std::string format, name;
    std::vector<std::unique_ptr<ArrowSchema>> kids;
    std::vector<ArrowSchema*> kid_ptrs;
};
void release_schema(ArrowSchema* s) {
    if (!s || !s->release) return;
    auto* o = static_cast<OwnedSchema*>(s->private_data);
    for (auto& k : o->kids)
        if (k->release) k->release(k.get());
    delete o;
    s->release = nullptr;
}
void fill_schema(ArrowSchema* s, const std::string& format, const std::string& name,
                 std::vector<std::unique_ptr<ArrowSchema>> kids = {}) {
    auto* o = new OwnedSchema();
    o->format = format;
    o->name = name;
    o->kids = std::move(kids);
    for (auto& k : o->kids) o->kid_ptrs.push_back(k.get());
    std::memset(s, 0, sizeof(*s));
    s->format = o->format.c_str();
    s->name = o->name.c_str();
    s->flags = 2;  // ARROW_FLAG_NULLABLE
    s->n_children = (int64_t)o->kid_ptrs.size();
    s->children = o->kid_ptrs.empty() ? nullptr : o->kid_ptrs.data();
    s->release = release_schema;
    s->private_data = o;
}
std::unique_ptr<ArrowSchema> make_schema(const std::string& format, const std::string& name,
                                         std::vector<std::unique_ptr<ArrowSchema>> kids = {}) {
    auto s = std::make_unique<ArrowSchema>();
    fill_schema(s.get(), format, name, std::move(kids));
    return s;
}

// Large result buffers live in page-locked host memory (pds_host_alloc): the copy back from the device then runs at the link
// rate (57 GB/s) instead of the pageable rate (17 GB/s: 7.8 ms for the 136 MB of the headline frame's coefficients).  Pinning
// pages is slow (milliseconds per 100 MB), so released blocks are kept for the next result of the same size class: a pool
// keyed by the block size, at most kPinnedCacheBytes cached; Polars frees a result through the array's release callback,
// which brings its blocks back here.  No device / no pinned memory left: pageable storage as before.
struct PinnedPool {
    static constexpr size_t kMinBytes = (size_t)4 << 20, kMaxBytes = (size_t)4 << 30;
    // page-locked memory kept cached between results: 4 GiB unless PDS_PLUGIN_PINNED_CACHE_MB says otherwise (0: nothing is cached;
    // the per-row results of the headline frame are two 0.8 GB blocks: with 1 GiB one of them was pinned afresh on every call, +40 ms).
    // Pinned pages are taken from every other process on the host, so the cache is bounded, evicts its largest blocks first when a
    // returning block would exceed the bound, and serves a request from any cached block up to twice its size.
    static size_t cache_cap() { return settings().pinned_cache_bytes; }
    std::mutex m;
    std::multimap<size_t, void*> free_blocks;   // size -> block
    std::map<void*, size_t> live;               // blocks handed out (size)
    size_t cached = 0;
    bool disabled = false;
    static PinnedPool& get() {
        static PinnedPool* p = new PinnedPool();  // (never destroyed: results may outlive static destruction order)
        return *p;
    }
    static size_t size_class(size_t bytes) {  // next multiple of 2 MiB: repeated calls on one frame shape reuse their blocks
        const size_t q = (size_t)2 << 20;
        return (bytes + q - 1) / q * q;
    }
    void* take(size_t bytes) {
        if (!settings().pinned_results || bytes < kMinBytes || bytes > kMaxBytes) return nullptr;
        const size_t sz = size_class(bytes);
        {
            std::lock_guard<std::mutex> g(m);
            if (disabled) return nullptr;
            auto it = free_blocks.lower_bound(sz);  // best fit: the smallest cached block that holds the request, up to 2x its size
            if (it != free_blocks.end() && it->first <= 2 * sz) {
                void* p = it->second;
                const size_t have = it->first;
                free_blocks.erase(it);
                cached -= have;
                live[p] = have;
                return p;
            }
        }
        void* p = nullptr;
        if (pds_host_alloc(sz, &p) != PDS_OK || !p) {
            std::lock_guard<std::mutex> g(m);
            if (live.empty() && free_blocks.empty()) disabled = true;  // (no device / not supported: do not ask again)
            return nullptr;
        }
        std::lock_guard<std::mutex> g(m);
        live[p] = sz;
        return p;
    }
    bool give_back(void* p) {  // false: not one of ours
        size_t sz = 0;
        std::vector<void*> evicted;
        {
            std::lock_guard<std::mutex> g(m);
            auto it = live.find(p);
            if (it == live.end()) return false;
            sz = it->second;
            live.erase(it);
            if (sz <= cache_cap()) {
                // make room by unpinning the largest cached blocks (freed below, outside the lock)
                while (cached + sz > cache_cap() && !free_blocks.empty()) {
                    auto last = std::prev(free_blocks.end());
                    evicted.push_back(last->second);
                    cached -= last->first;
                    free_blocks.erase(last);
                }
                free_blocks.emplace(sz, p);
                cached += sz;
                p = nullptr;
            }
        }
        for (void* e : evicted) (void)pds_host_free(e);
        if (p) (void)pds_host_free(p);
        return true;
    }
};

// Byte storage whose sizing does not zero: a std::vector<uint8_t>(n) touches every page of a multi-GB result once more
// before the D2H copy writes it.  assign(n, 0) still zeroes.
template <typename U>
struct NoInitAlloc : std::allocator<U> {
    template <typename V> struct rebind { using other = NoInitAlloc<V>; };
    NoInitAlloc() = default;
    template <typename V> NoInitAlloc(const NoInitAlloc<V>&) {}
    // large results: 2 MiB aligned and advised for transparent huge pages -- the first touch of a fresh multi-GB buffer by the
    // copy back from the device is otherwise one page fault per 4 KiB
    static constexpr size_t kHuge = (size_t)2 << 20;
    U* allocate(size_t n) {
        const size_t bytes = n * sizeof(U);
        if (bytes >= PinnedPool::kMinBytes)
            if (void* pin = PinnedPool::get().take(bytes)) return static_cast<U*>(pin);
        if (bytes >= 2 * kHuge) {
            void* p = nullptr;
            if (posix_memalign(&p, kHuge, (bytes + kHuge - 1) & ~(kHuge - 1)) != 0) throw std::bad_alloc();
            (void)madvise(p, (bytes + kHuge - 1) & ~(kHuge - 1), MADV_HUGEPAGE);
            return static_cast<U*>(p);
        }
        void* p = std::malloc(bytes ? bytes : 1);
        if (!p) throw std::bad_alloc();
        return static_cast<U*>(p);
    }
    void deallocate(U* p, size_t n) {
        if (n * sizeof(U) >= PinnedPool::kMinBytes && PinnedPool::get().give_back(p)) return;
        std::free(p);
    }
    template <typename V, typename... A>
    void construct(V* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new (static_cast<void*>(p)) V;
        else ::new (static_cast<void*>(p)) V(std::forward<A>(a)...);
    }
};
using ByteVec = std::vector<uint8_t, NoInitAlloc<uint8_t>>;
template <typename V>
using RawVec = std::vector<V, NoInitAlloc<V>>;
template <typename V>
ByteVec raw_buffer(size_t n) { return ByteVec(n * sizeof(V) + 8); }  // (8 spare bytes: readers may fetch whole words)
template <typename V>
V* as(ByteVec& b) { return reinterpret_cast<V*>(b.data()); }

struct OwnedArray {
    std::vector<ByteVec> bufs;   // owned buffer storage (empty = null buffer)
    std::vector<const void*> buf_ptrs;
    std::vector<std::unique_ptr<ArrowArray>> kids;
    std::vector<ArrowArray*> kid_ptrs;
};
void release_array(ArrowArray* a) {
    if (!a || !a->release) return;
    auto* o = static_cast<OwnedArray*>(a->private_data);
    for (auto& k : o->kids)
        if (k->release) k->release(k.get());
    delete o;
    a->release = nullptr;
}
// `skip`: the Arrow buffer i starts skip[i] bytes into its storage (a result whose leading rows are not part of the array)
std::unique_ptr<ArrowArray> make_array(int64_t length, int64_t null_count, std::vector<ByteVec> bufs, std::vector<bool> present,
                                       std::vector<std::unique_ptr<ArrowArray>> kids = {}, std::vector<size_t> skip = {}) {
    auto* o = new OwnedArray();
    o->bufs = std::move(bufs);
    for (size_t i = 0; i < o->bufs.size(); ++i)
        o->buf_ptrs.push_back(present[i] ? (const void*)(o->bufs[i].data() + (i < skip.size() ? skip[i] : 0)) : nullptr);
    o->kids = std::move(kids);
    for (auto& k : o->kids) o->kid_ptrs.push_back(k.get());
    auto a = std::make_unique<ArrowArray>();
    std::memset(a.get(), 0, sizeof(ArrowArray));
    a->length = length;
    a->null_count = null_count;
    a->n_buffers = (int64_t)o->buf_ptrs.size();
    a->buffers = o->buf_ptrs.data();
    a->n_children = (int64_t)o->kid_ptrs.size();
    a->children = o->kid_ptrs.empty() ? nullptr : o->kid_ptrs.data();
    a->release = release_array;
    a->private_data = o;
    return a;
}
template <typename V>
ByteVec bytes_of(const V* p, size_t n) {
    ByteVec b = raw_buffer<V>(n);
    if (n) std::memcpy(b.data(), p, n * sizeof(V));
    std::memset(b.data() + n * sizeof(V), 0, 8);
    return b;
}
// validity bitmap from byte flags (0 / non-zero); returns null_count
int64_t pack_validity(const uint8_t* flags, int64_t n, ByteVec& bm) {
    bm.assign((n + 7) / 8 + 8, 0);
    int64_t set = 0, i = 0;
    for (; i + 8 <= n; i += 8) {  // eight flags per step: normalise to 0 / 1 bytes, gather their low bits with one multiply
        uint64_t w;
        std::memcpy(&w, flags + i, 8);
        w = ((w | (w >> 4)) & 0x0f0f0f0f0f0f0f0full);
        w = ((w | (w >> 2)) & 0x0303030303030303ull);
        w = ((w | (w >> 1)) & 0x0101010101010101ull);
        const uint8_t bits = (uint8_t)((w * 0x0102040810204080ull) >> 56);
        bm[i >> 3] = bits;
        set += __builtin_popcount(bits);
    }
    for (; i < n; ++i)
        if (flags[i]) {
            bm[i >> 3] |= (uint8_t)(1u << (i & 7));
            ++set;
        }
    return n - set;
}
// any non-zero byte among n flags?  (a result without null rows needs no validity bitmap at all: eight flags per compare)
inline bool any_flag(const uint8_t* flags, int64_t n) {
    int64_t i = 0;
    uint64_t acc = 0;
    for (; i + 64 <= n; i += 64) {
        uint64_t w[8];
        std::memcpy(w, flags + i, 64);
        acc |= (w[0] | w[1]) | (w[2] | w[3]) | (w[4] | w[5]) | (w[6] | w[7]);
        if (acc) return true;
    }
    for (; i < n; ++i) acc |= flags[i];
    return acc != 0;
}
template <typename T>
const char* fmt_of() { return sizeof(T) == 8 ? "g" : "f"; }

// primitive array that takes over `values` (storage of skip_rows + n elements: the copy back from the device wrote it)
template <typename T>
std::unique_ptr<ArrowArray> prim_array_take(ByteVec&& values, int64_t n, const uint8_t* valid_flags /*nullable*/, int64_t skip_rows = 0) {
    ByteVec bm;
    int64_t nulls = 0;
    if (valid_flags) nulls = pack_validity(valid_flags, n, bm);
    std::vector<ByteVec> bufs;
    bufs.push_back(std::move(bm));
    bufs.push_back(std::move(values));
    return make_array(n, nulls, std::move(bufs), {nulls > 0, true}, {}, {0, (size_t)skip_rows * sizeof(T)});
}
// ... with a validity bitmap that is already packed (`bm`: (n + 7) / 8 + 8 bytes; `nulls` zero bits among the first n)
template <typename T>
std::unique_ptr<ArrowArray> prim_array_take_bitmap(ByteVec&& values, int64_t n, ByteVec&& bm, int64_t nulls) {
    std::vector<ByteVec> bufs;
    bufs.push_back(std::move(bm));
    bufs.push_back(std::move(values));
    return make_array(n, nulls, std::move(bufs), {nulls > 0, true}, {}, {0, 0});
}
template <typename T>
std::unique_ptr<ArrowArray> prim_array(const T* v, int64_t n, const uint8_t* valid_flags /*nullable*/) {
    return prim_array_take<T>(bytes_of(v, (size_t)n), n, valid_flags);
}
// LargeList<T>: row i = values[off[i] .. off[i+1])
template <typename T>
std::unique_ptr<ArrowArray> list_array(const std::vector<int64_t>& offsets, const uint8_t* valid_flags, const T* values,
                                       int64_t n_values) {
    const int64_t n = (int64_t)offsets.size() - 1;
    ByteVec bm;
    int64_t nulls = 0;
    if (valid_flags) nulls = pack_validity(valid_flags, n, bm);
    std::vector<ByteVec> bufs;
    bufs.push_back(std::move(bm));
    bufs.push_back(bytes_of(offsets.data(), offsets.size()));
    std::vector<std::unique_ptr<ArrowArray>> kids;
    kids.push_back(prim_array<T>(values, n_values, nullptr));
    return make_array(n, nulls, std::move(bufs), {nulls > 0, true}, std::move(kids));
}
// LargeList<T> of fixed-width rows that takes over its storage: `rows` = n x width values as the device wrote them, invalid
// rows included.  Arrow wants the values of the valid rows back to back: the leading invalid rows (a rolling fit: the first
// window - 1) are skipped by starting the values buffer behind them, later ones are closed up in place -- usually nothing moves.
template <typename T>
std::unique_ptr<ArrowArray> list_array_take_rows(ByteVec&& rows, int64_t n, int width, const uint8_t* valid_flags) {
    ByteVec obuf = raw_buffer<int64_t>((size_t)n + 1);
    int64_t* off = as<int64_t>(obuf);
    T* v = as<T>(rows);
    int64_t lead = 0;
    while (lead < n && !valid_flags[lead]) ++lead;
    for (int64_t i = 0; i < lead; ++i) off[i] = 0;
    int64_t dst = lead;
    for (int64_t i = lead; i < n; ++i) {
        off[i] = (dst - lead) * width;
        if (valid_flags[i]) {
            if (dst != i) std::memcpy(v + dst * width, v + i * width, (size_t)width * sizeof(T));
            ++dst;
        }
    }
    off[n] = (dst - lead) * width;
    ByteVec bm;
    const int64_t nulls = pack_validity(valid_flags, n, bm);
    std::vector<ByteVec> bufs;
    bufs.push_back(std::move(bm));
    bufs.push_back(std::move(obuf));
    std::vector<std::unique_ptr<ArrowArray>> kids;
    kids.push_back(prim_array_take<T>(std::move(rows), (dst - lead) * width, nullptr, lead * width));
    return make_array(n, nulls, std::move(bufs), {nulls > 0, true}, std::move(kids));
}
std::unique_ptr<ArrowArray> utf8_array(const std::vector<std::string>& strs) {
    std::vector<int64_t> off(strs.size() + 1, 0);
    std::string data;
    for (size_t i = 0; i < strs.size(); ++i) {
        data += strs[i];
        off[i + 1] = (int64_t)data.size();
    }
    std::vector<ByteVec> bufs;
    bufs.emplace_back();
    bufs.push_back(bytes_of(off.data(), off.size()));
    bufs.push_back(bytes_of(data.data(), data.size()));
    return make_array((int64_t)strs.size(), 0, std::move(bufs), {false, true, true});
}
std::unique_ptr<ArrowArray> struct_array(int64_t n, std::vector<std::unique_ptr<ArrowArray>> kids) {
    std::vector<ByteVec> bufs;
    bufs.emplace_back();
    return make_array(n, 0, std::move(bufs), {false}, std::move(kids));
}

template <typename T>
std::unique_ptr<ArrowSchema> list_schema(const std::string& name) {
    std::vector<std::unique_ptr<ArrowSchema>> kids;
    kids.push_back(make_schema(fmt_of<T>(), "item"));
    return make_schema("+L", name, std::move(kids));
}

struct OwnedSeries {
    std::unique_ptr<ArrowSchema> field;
    std::unique_ptr<ArrowArray> array;
    ArrowArray* arr_ptr[1];
};
void release_series(SeriesExport* s) {
    if (!s || !s->release) return;
    auto* o = static_cast<OwnedSeries*>(s->private_data);
    if (o->field && o->field->release) o->field->release(o->field.get());
    if (o->array && o->array->release) o->array->release(o->array.get());
    delete o;
    s->release = nullptr;
}
void export_series(SeriesExport* out, std::unique_ptr<ArrowSchema> field, std::unique_ptr<ArrowArray> arr) {
    auto* o = new OwnedSeries();
    o->field = std::move(field);
    o->array = std::move(arr);
    o->arr_ptr[0] = o->array.get();
    out->field = o->field.get();
    out->arrays = o->arr_ptr;
    out->len = 1;
    out->release = release_series;
    out->private_data = o;
}
