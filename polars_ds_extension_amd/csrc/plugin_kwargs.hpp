// plugin_kwargs.hpp -- the pickled kwargs of the expressions (protocol-5 flat dict parser) and the null policy
// Part of the one translation unit plugin.cpp (included there, inside its anonymous namespace, in dependency order).
#pragma once

// ------------------------------------------------------------------------------------------------- pickle (flat dict)
struct KwVal {
    enum Kind { NONE, BOOL, INT, FLOAT, STR } kind = NONE;
    bool b = false;
    int64_t i = 0;
    double f = 0.0;
    std::string s;
};
using Kwargs = std::map<std::string, KwVal>;

Kwargs parse_pickle_dict(const uint8_t* p, size_t n) {
    Kwargs out;
    if (n == 0) return out;
    std::vector<KwVal> stack, memo;
    std::vector<size_t> marks;
    size_t i = 0;
    auto need = [&](size_t k) {
        if (i + k > n) raise("kwargs: truncated pickle");
    };
    auto rd_le = [&](int bytes) {
        need(bytes);
        uint64_t v = 0;
        for (int b = 0; b < bytes; ++b) v |= (uint64_t)p[i + b] << (8 * b);
        i += bytes;
        return v;
    };
    size_t dict_pos = (size_t)-1;
    auto setitems = [&](size_t from) {
        if (dict_pos == (size_t)-1) raise("kwargs: SETITEM without a dict");
        for (size_t k = from; k + 1 < stack.size(); k += 2) {
            if (stack[k].kind != KwVal::STR) raise("kwargs: non-string key");
            out[stack[k].s] = stack[k + 1];
        }
        stack.resize(from);
    };
    while (i < n) {
        const uint8_t op = p[i++];
        switch (op) {
            case 0x80: need(1); ++i; break;                 // PROTO
            case 0x95: need(8); i += 8; break;               // FRAME
            case '}': {                                       // EMPTY_DICT
                KwVal v;
                stack.push_back(v);
                dict_pos = stack.size() - 1;
                break;
            }
            case 0x94: memo.push_back(stack.empty() ? KwVal() : stack.back()); break;  // MEMOIZE
            case '(': marks.push_back(stack.size()); break;  // MARK
            case 0x8c: {                                      // SHORT_BINUNICODE
                const size_t len = rd_le(1);
                need(len);
                KwVal v;
                v.kind = KwVal::STR;
                v.s.assign(reinterpret_cast<const char*>(p + i), len);
                i += len;
                stack.push_back(v);
                break;
            }
            case 'X': {                                       // BINUNICODE
                const size_t len = rd_le(4);
                need(len);
                KwVal v;
                v.kind = KwVal::STR;
                v.s.assign(reinterpret_cast<const char*>(p + i), len);
                i += len;
                stack.push_back(v);
                break;
            }
            case 0x88: case 0x89: {                           // NEWTRUE / NEWFALSE
                KwVal v;
                v.kind = KwVal::BOOL;
                v.b = op == 0x88;
                v.i = v.b;
                v.f = v.b;
                stack.push_back(v);
                break;
            }
            case 'N': stack.push_back(KwVal()); break;        // NONE
            case 'K': case 'M': case 'J': {                   // BININT1 / BININT2 / BININT
                const int bytes = op == 'K' ? 1 : (op == 'M' ? 2 : 4);
                uint64_t u = rd_le(bytes);
                KwVal v;
                v.kind = KwVal::INT;
                v.i = (op == 'J') ? (int64_t)(int32_t)(uint32_t)u : (int64_t)u;
                v.f = (double)v.i;
                stack.push_back(v);
                break;
            }
            case 0x8a: {                                      // LONG1
                const size_t len = rd_le(1);
                need(len);
                int64_t val = 0;
                for (size_t b = 0; b < len && b < 8; ++b) val |= (int64_t)p[i + b] << (8 * b);
                if (len > 0 && len < 8 && (p[i + len - 1] & 0x80)) val |= -((int64_t)1 << (8 * len));
                i += len;
                KwVal v;
                v.kind = KwVal::INT;
                v.i = val;
                v.f = (double)val;
                stack.push_back(v);
                break;
            }
            case 'G': {                                       // BINFLOAT (big endian)
                need(8);
                uint64_t u = 0;
                for (int b = 0; b < 8; ++b) u = (u << 8) | p[i + b];
                i += 8;
                KwVal v;
                v.kind = KwVal::FLOAT;
                std::memcpy(&v.f, &u, 8);
                v.i = (int64_t)v.f;
                stack.push_back(v);
                break;
            }
            case 'h': {                                       // BINGET
                const size_t idx = rd_le(1);
                if (idx >= memo.size()) raise("kwargs: bad memo index");
                stack.push_back(memo[idx]);
                break;
            }
            case 'j': {                                       // LONG_BINGET
                const size_t idx = rd_le(4);
                if (idx >= memo.size()) raise("kwargs: bad memo index");
                stack.push_back(memo[idx]);
                break;
            }
            case 'u': {                                       // SETITEMS
                if (marks.empty()) raise("kwargs: SETITEMS without MARK");
                const size_t from = marks.back();
                marks.pop_back();
                setitems(from);
                break;
            }
            case 's':                                         // SETITEM
                if (stack.size() < 3) raise("kwargs: SETITEM underflow");
                setitems(stack.size() - 2);
                break;
            case '.': return out;                             // STOP
            default: raise("kwargs: unsupported pickle opcode " + std::to_string((int)op));
        }
    }
    return out;
}

bool kw_bool(const Kwargs& k, const char* n, bool dflt = false) {
    auto it = k.find(n);
    if (it == k.end() || it->second.kind == KwVal::NONE) return dflt;
    return it->second.kind == KwVal::STR ? !it->second.s.empty() : (it->second.f != 0.0);
}
double kw_f64(const Kwargs& k, const char* n, double dflt = 0.0) {
    auto it = k.find(n);
    if (it == k.end() || it->second.kind == KwVal::NONE || it->second.kind == KwVal::STR) return dflt;
    return it->second.f;
}
int64_t kw_i64(const Kwargs& k, const char* n, int64_t dflt = 0) {
    auto it = k.find(n);
    if (it == k.end() || it->second.kind == KwVal::NONE || it->second.kind == KwVal::STR) return dflt;
    return it->second.kind == KwVal::FLOAT ? (int64_t)it->second.f : it->second.i;
}
std::string kw_str(const Kwargs& k, const char* n, const char* dflt = "") {
    auto it = k.find(n);
    if (it == k.end() || it->second.kind != KwVal::STR) return dflt;
    return it->second.s;
}

// NullPolicy::try_from (src/linear/mod.rs:43-66)
struct Policy {
    enum Kind { RAISE, SKIP, FILL, IGNORE, SKIP_WINDOW } kind;
    double fill = 0.0;
};
Policy parse_policy(const std::string& v) {
    std::string s;
    for (char c : v) s.push_back((char)std::tolower((unsigned char)c));
    if (s == "raise") return {Policy::RAISE};
    if (s == "skip") return {Policy::SKIP};
    if (s == "zero") return {Policy::FILL, 0.0};
    if (s == "one") return {Policy::FILL, 1.0};
    if (s == "ignore") return {Policy::IGNORE};
    if (s == "skip_window") return {Policy::SKIP_WINDOW};
    char* end = nullptr;
    const double x = std::strtod(v.c_str(), &end);
    if (end != v.c_str() && *end == '\0') return {Policy::FILL, x};
    raise("Invalid NullPolicy.");
}
