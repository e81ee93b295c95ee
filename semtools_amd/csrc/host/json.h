// json.h -- minimal JSON value + parser + serde_json-style pretty writer (2-space indent,
// insertion-ordered objects), enough for workspace config / metadata files and CLI output.
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "fmt.h"

namespace semtools {
namespace json {

struct Value {
    enum Kind { Null, Bool, Int, UInt, Float, String, Array, Object } kind = Null;
    bool b = false;
    int64_t i = 0;
    uint64_t u = 0;
    double f = 0.0;
    std::string s;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;

    static Value str(const std::string &v) { Value x; x.kind = String; x.s = v; return x; }
    static Value uint(uint64_t v) { Value x; x.kind = UInt; x.u = v; return x; }
    static Value sint(int64_t v) { Value x; x.kind = Int; x.i = v; return x; }
    static Value num(double v) { Value x; x.kind = Float; x.f = v; return x; }
    static Value array() { Value x; x.kind = Array; return x; }
    static Value object() { Value x; x.kind = Object; return x; }
    Value &set(const std::string &k, Value v) { obj.emplace_back(k, std::move(v)); return *this; }
    const Value *get(const std::string &k) const
    {
        for (auto &kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
    uint64_t as_u64() const { return kind == UInt ? u : kind == Int ? (uint64_t)i : (uint64_t)f; }
    int64_t as_i64() const { return kind == Int ? i : kind == UInt ? (int64_t)u : (int64_t)f; }
};

inline void write(const Value &v, std::string &out, int indent, bool pretty)
{
    auto nl = [&](int ind) { if (pretty) { out.push_back('\n'); out.append((size_t)ind * 2, ' '); } };
    switch (v.kind) {
        case Value::Null: out += "null"; break;
        case Value::Bool: out += v.b ? "true" : "false"; break;
        case Value::Int: out += std::to_string(v.i); break;
        case Value::UInt: out += std::to_string(v.u); break;
        case Value::Float: out += fmt::json_f64(v.f); break;
        case Value::String: out += fmt::json_string(v.s); break;
        case Value::Array:
            if (v.arr.empty()) { out += "[]"; break; }
            out.push_back('[');
            for (size_t k = 0; k < v.arr.size(); ++k) {
                nl(indent + 1);
                write(v.arr[k], out, indent + 1, pretty);
                if (k + 1 < v.arr.size()) out.push_back(',');
            }
            nl(indent);
            out.push_back(']');
            break;
        case Value::Object:
            if (v.obj.empty()) { out += "{}"; break; }
            out.push_back('{');
            for (size_t k = 0; k < v.obj.size(); ++k) {
                nl(indent + 1);
                out += fmt::json_string(v.obj[k].first);
                out += pretty ? ": " : ":";
                write(v.obj[k].second, out, indent + 1, pretty);
                if (k + 1 < v.obj.size()) out.push_back(',');
            }
            nl(indent);
            out.push_back('}');
            break;
    }
}

inline std::string to_string_pretty(const Value &v)
{
    std::string out;
    write(v, out, 0, true);
    return out;
}

class Parser {
public:
    explicit Parser(const std::string &t) : t_(t) {}
    Value parse()
    {
        Value v = value();
        ws();
        if (p_ != t_.size()) throw std::runtime_error("trailing characters in JSON");
        return v;
    }
    // Large members can be left out of the DOM: the value of `key` in an object at nesting depth `depth` (root = 1)
    // is skipped -- a Null stays in its place -- and its byte span in the text is reported, for a purpose-built scan
    // (hf_tokenizer.cpp: a 500 k-entry vocabulary costs 150 B per entry as Values).
    void skip_member(const std::string &key, int depth) { skip_key_ = key; skip_depth_ = depth; }
    bool skipped(size_t &begin, size_t &end) const { begin = skip_begin_; end = skip_end_; return skip_end_ > skip_begin_; }
    // ... or consumed on the spot by `reader` (which must leave the parser just behind the value): one pass over the text
    void read_member(const std::string &key, int depth, std::function<void(Parser &)> reader)
    {
        skip_key_ = key;
        skip_depth_ = depth;
        reader_ = std::move(reader);
    }
    // low-level access for such scans
    void seek(size_t pos) { p_ = pos; }
    size_t pos() const { return p_; }

    void ws() { while (p_ < t_.size() && (t_[p_] == ' ' || t_[p_] == '\n' || t_[p_] == '\r' || t_[p_] == '\t')) ++p_; }
    char peek() { ws(); if (p_ >= t_.size()) throw std::runtime_error("unexpected end of JSON"); return t_[p_]; }
    void expect(char c) { if (peek() != c) throw std::runtime_error(std::string("expected '") + c + "' in JSON"); ++p_; }
    bool accept(char c) { if (peek() == c) { ++p_; return true; } return false; }
    double number_f64()
    {
        ws();
        double v = 0.0;   // std::from_chars: correctly rounded like strtod / serde_json, several times faster (no locale)
        const char *b = t_.data() + p_ + (p_ < t_.size() && t_[p_] == '+' ? 1 : 0);
        const auto r = std::from_chars(b, t_.data() + t_.size(), v);
        if (r.ec != std::errc() || r.ptr == b) throw std::runtime_error("invalid JSON number");
        p_ = (size_t)(r.ptr - t_.data());
        return v;
    }
    void skip_value()   // brackets matched outside strings
    {
        const char c = peek();
        if (c == '"') { (void)string(); return; }
        if (c != '{' && c != '[') { (void)number_or_literal(); return; }
        int depth = 0;
        while (p_ < t_.size()) {
            const char x = t_[p_];
            if (x == '"') { (void)string(); continue; }
            if (x == '{' || x == '[') ++depth;
            else if (x == '}' || x == ']') { if (--depth == 0) { ++p_; return; } }
            ++p_;
        }
        throw std::runtime_error("unterminated JSON value");
    }

private:
    const std::string &t_;
    size_t p_ = 0;
    std::string skip_key_;
    int skip_depth_ = -1, depth_ = 0;
    size_t skip_begin_ = 0, skip_end_ = 0;
    std::function<void(Parser &)> reader_;
    Value number_or_literal()
    {
        if (t_.compare(p_, 4, "true") == 0) { p_ += 4; Value v; v.kind = Value::Bool; v.b = true; return v; }
        if (t_.compare(p_, 5, "false") == 0) { p_ += 5; Value v; v.kind = Value::Bool; v.b = false; return v; }
        if (t_.compare(p_, 4, "null") == 0) { p_ += 4; return Value(); }
        return number();
    }
    Value value()
    {
        const char c = peek();
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') return Value::str(string());
        return number_or_literal();
    }
    Value object()
    {
        Value v = Value::object();
        expect('{');
        if (peek() == '}') { ++p_; return v; }
        ++depth_;
        for (;;) {
            ws();
            std::string k = string();
            expect(':');
            if (depth_ == skip_depth_ && k == skip_key_ && skip_end_ == 0) {
                ws();
                skip_begin_ = p_;
                if (reader_) reader_(*this);
                else skip_value();
                skip_end_ = p_;
                v.obj.emplace_back(std::move(k), Value());
            } else {
                v.obj.emplace_back(std::move(k), value());
            }
            if (peek() == ',') { ++p_; continue; }
            expect('}');
            --depth_;
            return v;
        }
    }
    Value array()
    {
        Value v = Value::array();
        expect('[');
        if (peek() == ']') { ++p_; return v; }
        for (;;) {
            v.arr.push_back(value());
            if (peek() == ',') { ++p_; continue; }
            expect(']');
            return v;
        }
    }
public:
    static void put_utf8(std::string &o, unsigned cp)
    {
        if (cp < 0x80) o.push_back((char)cp);
        else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) { o.push_back((char)(0xE0 | (cp >> 12))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else { o.push_back((char)(0xF0 | (cp >> 18))); o.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
    }
    std::string string()
    {
        if (peek() != '"') throw std::runtime_error("expected string in JSON");
        ++p_;
        std::string o;
        while (p_ < t_.size() && t_[p_] != '"') {
            // bulk-copy the run up to the next quote or backslash
            size_t e = p_;
            while (e < t_.size() && t_[e] != '"' && t_[e] != '\\') ++e;
            if (e > p_) { o.append(t_, p_, e - p_); p_ = e; continue; }
            char c = t_[p_++];
            if (c != '\\') { o.push_back(c); continue; }
            if (p_ >= t_.size()) break;
            c = t_[p_++];
            switch (c) {
                case 'n': o.push_back('\n'); break;
                case 't': o.push_back('\t'); break;
                case 'r': o.push_back('\r'); break;
                case 'b': o.push_back('\b'); break;
                case 'f': o.push_back('\f'); break;
                case 'u': {
                    unsigned cp = (unsigned)std::strtoul(t_.substr(p_, 4).c_str(), nullptr, 16);
                    p_ += 4;
                    if (cp >= 0xD800 && cp < 0xDC00 && p_ + 6 <= t_.size() && t_[p_] == '\\' && t_[p_ + 1] == 'u') {
                        unsigned lo = (unsigned)std::strtoul(t_.substr(p_ + 2, 4).c_str(), nullptr, 16);
                        p_ += 6;
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    put_utf8(o, cp);
                    break;
                }
                default: o.push_back(c);
            }
        }
        if (p_ >= t_.size()) throw std::runtime_error("unterminated string in JSON");
        ++p_;
        return o;
    }
private:
    Value number()
    {
        const size_t s = p_;
        bool is_float = false;
        if (p_ < t_.size() && (t_[p_] == '-' || t_[p_] == '+')) ++p_;
        while (p_ < t_.size() && (isdigit((unsigned char)t_[p_]) || t_[p_] == '.' || t_[p_] == 'e' || t_[p_] == 'E' || t_[p_] == '-' || t_[p_] == '+')) {
            if (t_[p_] == '.' || t_[p_] == 'e' || t_[p_] == 'E') is_float = true;
            ++p_;
        }
        if (s == p_) throw std::runtime_error("invalid JSON value");
        const std::string tok = t_.substr(s, p_ - s);
        if (is_float) return Value::num(std::strtod(tok.c_str(), nullptr));
        if (tok[0] == '-') return Value::sint(std::strtoll(tok.c_str(), nullptr, 10));
        return Value::uint(std::strtoull(tok.c_str(), nullptr, 10));
    }
};

inline Value parse(const std::string &text) { return Parser(text).parse(); }

}  // namespace json
}  // namespace semtools
