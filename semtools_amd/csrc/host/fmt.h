// fmt.h -- number/string formatting that reproduces what the reference prints:
//   * Rust `{}` Display for f64/f32 (shortest round-trip digits, never scientific)   -> text output
//   * serde_json (ryu) f64 formatting and string escaping, to_string_pretty layout    -> JSON output
// Reference: src/cmds/search.rs:35-110 (println! formats), src/json_mode.rs (field order).
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace semtools {
namespace fmt {

// shortest round-trip decimal digits and exponent: value = 0.d1d2...dn * 10^point
template <typename F>
inline void shortest_digits(F v, std::string &digits, int &point, bool &neg)
{
    char buf[64];
    neg = std::signbit(v);
    if (neg) v = -v;
    auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::scientific);
    std::string s(buf, r.ptr);  // d.ddddde[+-]XX
    const size_t e = s.find('e');
    std::string mant = s.substr(0, e);
    const int exp10 = std::stoi(s.substr(e + 1));
    digits.clear();
    for (char c : mant) if (c != '.') digits.push_back(c);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    point = exp10 + 1;
}

// Rust's Display for floats: plain decimal, shortest digits, "inf"/"NaN", -0 prints "-0".
template <typename F>
inline std::string rust_display(F v)
{
    if (v != v) return "NaN";
    if (v == (F)INFINITY) return "inf";
    if (v == -(F)INFINITY) return "-inf";
    if (v == 0) return std::signbit(v) ? "-0" : "0";
    std::string d; int point; bool neg;
    shortest_digits(v, d, point, neg);
    std::string out = neg ? "-" : "";
    const int n = (int)d.size();
    if (point <= 0) { out += "0."; out.append((size_t)(-point), '0'); out += d; }
    else if (point >= n) { out += d; out.append((size_t)(point - n), '0'); }
    else { out += d.substr(0, (size_t)point); out += "."; out += d.substr((size_t)point); }
    return out;
}

// serde_json f64 (ryu "pretty" rules): decimal for -5 < kk <= 16, else d.ddde[-]X; always a ".0"
// on integral decimals; non-finite -> null.
inline std::string json_f64(double v)
{
    if (v != v || v == INFINITY || v == -INFINITY) return "null";
    if (v == 0) return std::signbit(v) ? "-0.0" : "0.0";
    std::string d; int point; bool neg;
    shortest_digits(v, d, point, neg);
    std::string out = neg ? "-" : "";
    const int n = (int)d.size();
    const int kk = point;            // position of the decimal point relative to the first digit
    if (n <= kk && kk <= 16) { out += d; out.append((size_t)(kk - n), '0'); out += ".0"; }
    else if (0 < kk && kk <= 16) { out += d.substr(0, (size_t)kk); out += "."; out += d.substr((size_t)kk); }
    else if (-5 < kk && kk <= 0) { out += "0."; out.append((size_t)(-kk), '0'); out += d; }
    else {
        out += d.substr(0, 1);
        if (n > 1) { out += "."; out += d.substr(1); }
        out += "e"; out += std::to_string(kk - 1);
    }
    return out;
}

inline std::string json_string(const std::string &s)
{
    std::string o = "\"";
    for (unsigned char c : s) {
        switch (c) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break;
            case '\t': o += "\\t"; break;
            case '\b': o += "\\b"; break;
            case '\f': o += "\\f"; break;
            default:
                if (c < 0x20) { char b[8]; snprintf(b, sizeof(b), "\\u%04x", c); o += b; }
                else o.push_back((char)c);
        }
    }
    o += "\"";
    return o;
}

}  // namespace fmt
}  // namespace semtools
