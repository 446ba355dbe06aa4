// rust_fmt.hpp -- text forms of f64 that the reference's output layer produces, so that the CLI's stdout is
// byte-compatible with turbo-metrics-cli (crates/turbo-metrics-cli/src/output.rs):
//   display()     Rust `{}`   (CSV fields, output.rs:57,122): shortest round-trip digits, never an exponent, "100" for 100.0
//   debug()       Rust `{:?}` (`{:#?}` of Stats, output.rs:85-96): like display but "100.0", exponent form outside [1e-4, 1e16)
//   json_number() serde_json / ryu (output.rs:52,98-105): "100.0", exponent form outside [1e-5, 1e16), non-finite -> null
#pragma once
#include <charconv>
#include <cmath>
#include <cstring>
#include <string>

namespace tm_host {

// shortest round-trip decimal digits of |x| and the decimal exponent e10 such that |x| = 0.d1d2d3... * 10^e10
inline void shortest_digits(double x, std::string &digits, int &e10)
{
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, std::fabs(x), std::chars_format::scientific);
    std::string s(buf, r.ptr); // d[.ddd]e[+-]XX
    const size_t epos = s.find('e');
    const int exp = std::stoi(s.substr(epos + 1));
    digits.clear();
    for (size_t i = 0; i < epos; ++i)
        if (s[i] != '.') digits.push_back(s[i]);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    e10 = exp + 1;
}

inline std::string fixed_from_digits(const std::string &d, int e10, bool force_fraction)
{
    std::string out;
    if (e10 <= 0) {
        out = "0.";
        out.append((size_t)(-e10), '0');
        out += d;
    } else if ((size_t)e10 >= d.size()) {
        out = d;
        out.append((size_t)e10 - d.size(), '0');
        if (force_fraction) out += ".0";
    } else {
        out = d.substr(0, (size_t)e10) + "." + d.substr((size_t)e10);
    }
    return out;
}

inline std::string exp_from_digits(const std::string &d, int e10)
{
    std::string out = d.substr(0, 1);
    if (d.size() > 1) out += "." + d.substr(1);
    out += "e" + std::to_string(e10 - 1);
    return out;
}

inline std::string display(double x)
{
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x < 0 ? "-inf" : "inf";
    if (x == 0.0) return std::signbit(x) ? "-0" : "0";
    std::string d; int e;
    shortest_digits(x, d, e);
    return (x < 0 ? "-" : "") + fixed_from_digits(d, e, false);
}

inline std::string debug(double x)
{
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x < 0 ? "-inf" : "inf";
    if (x == 0.0) return std::signbit(x) ? "-0.0" : "0.0";
    std::string d; int e;
    shortest_digits(x, d, e);
    const double a = std::fabs(x);
    const bool sci = a < 1e-4 || a >= 1e16; // core::fmt::float: float_to_general_debug
    return (x < 0 ? "-" : "") + (sci ? exp_from_digits(d, e) : fixed_from_digits(d, e, true));
}

inline std::string json_number(double x)
{
    if (!std::isfinite(x)) return "null"; // serde_json serialises non-finite floats as null
    if (x == 0.0) return std::signbit(x) ? "-0.0" : "0.0";
    std::string d; int e;
    shortest_digits(x, d, e);
    const int kk = e; // ryu's `kk`: position of the decimal point relative to the digit string
    std::string body; // ryu::pretty::format64
    if (kk > 0 && kk <= 16) body = fixed_from_digits(d, e, true);        // 1234e7 -> 12340000000.0, 1234e-2 -> 12.34
    else if (kk > -5 && kk <= 0) body = fixed_from_digits(d, e, true);   // 1234e-6 -> 0.001234
    else body = exp_from_digits(d, e);                                  // 1e30, 1.234e33, 1.2e-7
    return (x < 0 ? "-" : "") + body;
}

} // namespace tm_host
