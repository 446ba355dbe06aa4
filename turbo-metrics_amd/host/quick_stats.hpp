// quick_stats.hpp -- counterpart of crates/quick-stats/src/lib.rs:4-97 (`full::Stats`): same definitions, same
// evaluation order (mean = sum of the SORTED values / n, variances from the unsorted values, percentiles by linear
// interpolation on (len-1)*pct/100).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace tm_host {

struct Stats {
    double min, max, mean, var, sample_var, stddev, sample_stddev, p1, p5, p50, p95, p99;

    static double percentile_of_sorted(const std::vector<double> &s, double pct)
    {
        if (s.size() == 1) return s[0];
        if (pct == 100.0) return s.back();
        const double length = (double)(s.size() - 1);
        const double rank = (pct / 100.0) * length;
        const double lrank = std::floor(rank);
        const double d = rank - lrank;
        const size_t n = (size_t)lrank;
        const double lo = s[n], hi = s[n + 1];
        return lo + (hi - lo) * d;
    }

    static double compute_var(const std::vector<double> &v, double mean, bool sample)
    {
        if (v.size() < 2) return 0.0;
        double acc = 0.0;
        for (double s : v) { const double x = s - mean; acc += x * x; }
        return acc / (double)(sample ? v.size() - 1 : v.size());
    }

    // the reference indexes sorted[0] and panics on an empty slice (quick-stats/src/lib.rs:25); here: an exception
    static Stats compute(const std::vector<double> &values)
    {
        if (values.empty()) throw std::out_of_range("Stats::compute: no values (the reference panics here: index out of bounds)");
        std::vector<double> sorted(values);
        std::sort(sorted.begin(), sorted.end(), [](double a, double b) { // f64::total_cmp
            auto key = [](double x) { long long i; memcpy(&i, &x, 8); return i ^ (long long)((unsigned long long)(i >> 63) >> 1); };
            return key(a) < key(b);
        });
        Stats s;
        s.min = sorted.front();
        s.max = sorted.back();
        double sum = 0.0;
        for (double x : sorted) sum += x;
        s.mean = sum / (double)values.size();
        s.var = compute_var(values, s.mean, false);
        s.sample_var = compute_var(values, s.mean, true);
        s.stddev = std::sqrt(s.var);
        s.sample_stddev = std::sqrt(s.sample_var);
        s.p1 = percentile_of_sorted(sorted, 1.0);
        s.p5 = percentile_of_sorted(sorted, 5.0);
        s.p50 = percentile_of_sorted(sorted, 50.0);
        s.p95 = percentile_of_sorted(sorted, 95.0);
        s.p99 = percentile_of_sorted(sorted, 99.0);
        return s;
    }
};

} // namespace tm_host
