// plugin_arrow_in.hpp -- Series import over the Arrow C Data Interface: borrowed single chunks, casts, multi-chunk gather, validity bitmaps
// Part of the one translation unit plugin.cpp (included there, inside its anonymous namespace, in dependency order).
#pragma once

// ------------------------------------------------------------------------------------------------- Series import
// A single chunk whose dtype already is T is BORROWED: `view` points into Polars' Arrow buffer (inputs stay alive for the
// call) and the staging copy to HBM reads it directly -- the reference's marshalling memcpy (series_to_slice_inner,
// src/utils/mod.rs:101-206) has no counterpart on that path.  Several chunks, or a dtype that needs a cast, are
// gathered into `values`; so is any column a null policy rewrites (`own()`).
template <typename T>
struct Column {
    std::string name;
    const T* view = nullptr;         // borrowed Arrow values (offset applied), or nullptr when `values` holds the column
    int64_t len = 0;
    std::vector<T> values;           // owned: contiguous, all chunks concatenated, cast to T
    std::vector<uint8_t> validity;   // bitmap (LSB first, bit 0 = row 0), empty = no nulls
    int64_t null_count = 0;
    const T* data() const { return view ? view : values.data(); }
    int64_t size() const { return len; }
    T at(int64_t i) const { return data()[i]; }
    std::vector<T>& own() {          // writable storage (copies a borrowed chunk once)
        if (view) {
            values.assign(view, view + len);
            view = nullptr;
        }
        return values;
    }
    void shrink(int64_t n) {         // after a compaction of own()
        values.resize((size_t)n);
        len = n;
    }
};

bool bit_get(const uint8_t* bm, int64_t i) { return (bm[i >> 3] >> (i & 7)) & 1; }
void bit_set(std::vector<uint8_t>& bm, int64_t i) { bm[i >> 3] |= (uint8_t)(1u << (i & 7)); }

template <typename T, typename S>
void append_cast(std::vector<T>& dst, const void* buf, int64_t off, int64_t len) {
    const S* s = static_cast<const S*>(buf) + off;
    const size_t base = dst.size();
    dst.resize(base + len);
    for (int64_t i = 0; i < len; ++i) dst[base + i] = (T)s[i];
}

template <typename T> constexpr const char* arrow_fmt_of();
template <> constexpr const char* arrow_fmt_of<double>() { return "g"; }
template <> constexpr const char* arrow_fmt_of<float>() { return "f"; }
template <> constexpr const char* arrow_fmt_of<int64_t>() { return "l"; }

// validity of one chunk -> bits [pos, pos + length) of `dst`; returns the chunk's null count
inline int64_t append_validity(std::vector<uint8_t>& dst, int64_t pos, const ArrowArray* a) {
    const uint8_t* bm = static_cast<const uint8_t*>(a->buffers[0]);
    const int64_t n = a->length;
    if (bm == nullptr || a->null_count == 0) {
        int64_t i = 0;
        for (; i < n && ((pos + i) & 7); ++i) bit_set(dst, pos + i);
        const int64_t whole = (n - i) / 8;
        if (whole > 0) std::memset(dst.data() + ((pos + i) >> 3), 0xff, (size_t)whole);
        for (i += whole * 8; i < n; ++i) bit_set(dst, pos + i);
        return 0;
    }
    int64_t nulls = 0;
    if (((pos | a->offset) & 7) == 0) {  // byte aligned on both sides: copy whole bytes, count with popcount
        const int64_t whole = n / 8;
        const uint8_t* src = bm + (a->offset >> 3);
        std::memcpy(dst.data() + (pos >> 3), src, (size_t)whole);
        int64_t set = 0, k = 0;
        for (; k + 8 <= whole; k += 8) {
            uint64_t w;
            std::memcpy(&w, src + k, 8);
            set += __builtin_popcountll(w);
        }
        for (; k < whole; ++k) set += __builtin_popcount(src[k]);
        nulls = whole * 8 - set;
        for (int64_t i = whole * 8; i < n; ++i) {
            if (bit_get(bm, a->offset + i)) bit_set(dst, pos + i);
            else ++nulls;
        }
        return nulls;
    }
    for (int64_t i = 0; i < n; ++i) {
        if (bit_get(bm, a->offset + i)) bit_set(dst, pos + i);
        else ++nulls;
    }
    return nulls;
}

template <typename T>
Column<T> import_series(const SeriesExport& se) {
    Column<T> c;
    if (!se.field || !se.field->format) raise("input series without a schema");
    c.name = se.field->name ? se.field->name : "";
    const std::string fmt = se.field->format;
    int64_t total = 0;
    for (size_t k = 0; k < se.len; ++k) {
        if (se.arrays[k]->n_buffers < 2) raise("unsupported input layout for " + c.name);
        total += se.arrays[k]->length;
    }
    c.len = total;
    bool any_null = false;
    for (size_t k = 0; k < se.len; ++k) any_null |= se.arrays[k]->null_count != 0 && se.arrays[k]->buffers[0] != nullptr;
    if (any_null) c.validity.assign((total + 7) / 8, 0);
    if (se.len == 1 && fmt == arrow_fmt_of<T>() && se.arrays[0]->buffers[1] != nullptr) {
        c.view = static_cast<const T*>(se.arrays[0]->buffers[1]) + se.arrays[0]->offset;  // no host copy
    } else {
        c.values.reserve(total);
        for (size_t k = 0; k < se.len; ++k) {
            const ArrowArray* a = se.arrays[k];
            const void* vb = a->buffers[1];
            if (a->length == 0) continue;
            if (fmt == "g") append_cast<T, double>(c.values, vb, a->offset, a->length);
            else if (fmt == "f") append_cast<T, float>(c.values, vb, a->offset, a->length);
            else if (fmt == "l") append_cast<T, int64_t>(c.values, vb, a->offset, a->length);
            else if (fmt == "i") append_cast<T, int32_t>(c.values, vb, a->offset, a->length);
            else if (fmt == "s") append_cast<T, int16_t>(c.values, vb, a->offset, a->length);
            else if (fmt == "c") append_cast<T, int8_t>(c.values, vb, a->offset, a->length);
            else if (fmt == "L") append_cast<T, uint64_t>(c.values, vb, a->offset, a->length);
            else if (fmt == "I") append_cast<T, uint32_t>(c.values, vb, a->offset, a->length);
            else if (fmt == "S") append_cast<T, uint16_t>(c.values, vb, a->offset, a->length);
            else if (fmt == "C") append_cast<T, uint8_t>(c.values, vb, a->offset, a->length);
            else raise("column '" + c.name + "' has a non-numeric dtype (" + fmt + ")");
        }
        if (fmt != "g" && fmt != "f" && fmt != "l" && fmt != "i" && fmt != "s" && fmt != "c" && fmt != "L" && fmt != "I" && fmt != "S" &&
            fmt != "C")
            raise("column '" + c.name + "' has a non-numeric dtype (" + fmt + ")");
    }
    if (any_null) {
        int64_t pos = 0;
        for (size_t k = 0; k < se.len; ++k) {
            c.null_count += append_validity(c.validity, pos, se.arrays[k]);
            pos += se.arrays[k]->length;
        }
    }
    if (c.null_count == 0) c.validity.clear();
    return c;
}
