// sample.hpp -- descriptive statistics of the timed runs and their JSON form.
//
// Field for field what the reference prints for "execution_time"
// (src/util/sample.hpp:137-165), including its conventions:
//   median   = sorted[n/2]                      (upper median; :43-54, `n % 1 == 0`)
//   variance = sum (v-mean)^2 / (n-1)           (NaN for one sample; :95-106)
//   skewness = m3 / sqrt(variance^3)            (population m3 over SAMPLE variance; :117-125)
//   kurtosis = m4 / m2^2                        (population moments; :127-135)
// Empty samples give NaN for everything but min/max, which keep the type's extremes.
#pragma once

#include <algorithm>
#include <cmath>
#include <limits>
#include <ostream>
#include <string>
#include <vector>

template <typename T>
struct SampleStats
{
    std::size_t samples = 0;
    T min = std::numeric_limits<T>::max();
    T max = std::numeric_limits<T>::min();
    double mean = std::numeric_limits<double>::quiet_NaN();
    double median = std::numeric_limits<double>::quiet_NaN();
    double variance = std::numeric_limits<double>::quiet_NaN();
    double standard_deviation = std::numeric_limits<double>::quiet_NaN();
    double skewness = std::numeric_limits<double>::quiet_NaN();
    double kurtosis = std::numeric_limits<double>::quiet_NaN();
};

template <typename T>
SampleStats<T> sample_stats(std::vector<T> const & v)
{
    SampleStats<T> s;
    std::size_t const n = v.size();
    s.samples = n;
    for (T const & x : v) {
        s.min = std::min(s.min, x);
        s.max = std::max(s.max, x);
    }
    if (n == 0)
        return s;

    double sum = 0.0;
    for (T const & x : v)
        sum = sum + x;
    double const mu = sum / (double) n;
    s.mean = mu;

    std::vector<T> sorted(v);
    std::sort(sorted.begin(), sorted.end());
    s.median = (double) sorted[n / 2];

    // central sums of order 2, 3, 4 (each term multiplied out left to right)
    double c2 = 0.0, c3 = 0.0, c4 = 0.0;
    for (T const & x : v) {
        double const d = x - mu;
        c2 = c2 + d * d;
        c3 = c3 + d * d * d;
        c4 = c4 + d * d * d * d;
    }
    double const m2 = c2 / (double) n, m3 = c3 / (double) n, m4 = c4 / (double) n;
    s.variance = c2 / (double) (n - 1);
    s.standard_deviation = std::sqrt(s.variance);
    s.skewness = m3 / std::sqrt(s.variance * s.variance * s.variance);
    s.kurtosis = m4 / (m2 * m2);
    return s;
}

template <typename T>
std::ostream & print_sample(std::ostream & o, std::vector<T> const & v, std::string const & unit)
{
    SampleStats<T> const s = sample_stats(v);
    return o << "{\n"
             << "\"samples\": " << s.samples << ",\n"
             << "\"min\": " << s.min << ",\n"
             << "\"max\": " << s.max << ",\n"
             << "\"mean\": " << s.mean << ",\n"
             << "\"median\": " << s.median << ",\n"
             << "\"variance\": " << s.variance << ",\n"
             << "\"standard_deviation\": " << s.standard_deviation << ",\n"
             << "\"skewness\": " << s.skewness << ",\n"
             << "\"kurtosis\": " << s.kurtosis << ",\n"
             << "\"unit\": \"" << unit << "\""
             << "\n}";
}
