// flat_vocab.h -- piece -> id lookup for tokenizer vocabularies (hf_tokenizer.cpp, host.cpp's vocab.txt tokenizer).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace semtools {

// piece -> id lookup without a node or a std::string per entry: one string pool, one open-addressing table built in a
// single pass once all pieces are known (500 k pieces: ~10 ms to build; std::unordered_map<std::string, ..> took 100).
class FlatVocab {
public:
    void reserve_bytes(size_t bytes) { pool_.reserve(bytes); entries_.reserve(bytes / 12); }
    // a repeated piece: the later id wins (like the crate's HashMap insert)
    void add(const std::string &piece, uint32_t id)
    {
        entries_.push_back(Entry{(uint32_t)pool_.size(), (uint32_t)piece.size(), id});
        pool_.append(piece);
    }
    void build()
    {
        size_t cap = 16;
        while (cap < entries_.size() * 2 + 2) cap <<= 1;
        slots_.assign(cap, 0);
        mask_ = cap - 1;
        for (uint32_t e = 0; e < entries_.size(); ++e) {
            const Entry &x = entries_[e];
            size_t h = hash(pool_.data() + x.off, x.len) & mask_;
            for (;;) {
                const uint32_t cur = slots_[h];
                if (!cur) { slots_[h] = e + 1; break; }
                const Entry &y = entries_[cur - 1];
                if (y.len == x.len && memcmp(pool_.data() + y.off, pool_.data() + x.off, x.len) == 0) { slots_[h] = e + 1; break; }  // later wins
                h = (h + 1) & mask_;
            }
        }
    }
    // id of the piece, or -1
    int64_t find(const char *p, size_t n) const
    {
        if (slots_.empty()) return -1;
        size_t h = hash(p, n) & mask_;
        for (;;) {
            const uint32_t cur = slots_[h];
            if (!cur) return -1;
            const Entry &y = entries_[cur - 1];
            if (y.len == n && memcmp(pool_.data() + y.off, p, n) == 0) return y.id;
            h = (h + 1) & mask_;
        }
    }
    int64_t find(const std::string &s) const { return find(s.data(), s.size()); }
    size_t size() const { return entries_.size(); }

private:
    struct Entry { uint32_t off, len, id; };
    static size_t hash(const char *p, size_t n)
    {
        uint64_t h = 0xcbf29ce484222325ull;   // FNV-1a, then a finaliser (the low bits of FNV alone cluster on short keys)
        for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 0x100000001b3ull; }
        h ^= h >> 32;
        h *= 0x9E3779B97F4A7C15ull;
        return (size_t)(h >> 20);
    }
    std::string pool_;
    std::vector<Entry> entries_;
    std::vector<uint32_t> slots_;
    size_t mask_ = 0;
};

}  // namespace semtools
