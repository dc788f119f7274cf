#include "matrix-market.hpp"
#include "synthetic.hpp"

#include "matrix-error.hpp"
#include "matrix-reorder.hpp"
#include "matrix-cache.hpp"

#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <istream>
#include <iterator>
#include <numeric>
#include <ostream>
#include <sstream>

#ifdef _OPENMP
#include <omp.h>
#endif

using matrix::matrix_error;

namespace matrix_market {

// Below this many entries the converters run on the calling thread: waking an OpenMP team costs
// tens of milliseconds on a busy host, the work itself microseconds (BASELINE configs[0] is a
// 1138-row matrix).
static constexpr std::size_t kSerialBelow = 1u << 16;


Matrix::Matrix(Header header, std::vector<std::string> comments, Size size,
               std::vector<index_type> i, std::vector<index_type> j, std::vector<real_type> a,
               std::vector<real_type> imag)
    : header_(header)
    , comments_(std::move(comments))
    , size_(size)
    , i_(std::move(i))
    , j_(std::move(j))
    , a_(std::move(a))
    , imag_(std::move(imag))
{
}

std::vector<real_type> Matrix::values_real() const
{
    if (header_.field == Field::pattern)
        return std::vector<real_type>((std::size_t) size_.num_entries, 1.0);
    return a_;
}

std::vector<index_type> Matrix::row_lengths() const
{
    std::vector<index_type> len((std::size_t) std::max<index_type>(size_.rows, 0), 0);
    for (index_type r : i_) {
        if (r < 1 || r > size_.rows)
            throw matrix_error("Row index out of bounds: " + std::to_string(r));
        ++len[(std::size_t) r - 1];
    }
    return len;
}

index_type Matrix::max_row_length() const
{
    auto const len = row_lengths();
    return len.empty() ? 0 : *std::max_element(len.begin(), len.end());
}

// ------------------------------------------------------------------------------------
// text -> Matrix
// ------------------------------------------------------------------------------------
namespace {

inline bool is_space(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

std::string lower(std::string s)
{
    for (char & c : s)
        if (c >= 'A' && c <= 'Z')
            c = (char) (c - 'A' + 'a');
    return s;
}

// one line starting at pos (without the newline); pos moves past it
std::string take_line(char const * d, std::size_t n, std::size_t & pos)
{
    std::size_t e = pos;
    while (e < n && d[e] != '\n')
        ++e;
    std::string line(d + pos, e - pos);
    pos = (e < n) ? e + 1 : e;
    return line;
}

Header parse_header(std::string const & line)
{
    std::istringstream s(line);
    std::string banner, object, format, field, symmetry;
    s >> banner;
    if (banner != "%%MatrixMarket")
        throw matrix_error("Failed to parse header: Expected \"%%MatrixMarket\", got \"" + banner + "\"");
    s >> object >> format >> field >> symmetry;
    object = lower(object);
    format = lower(format);
    field = lower(field);
    symmetry = lower(symmetry);
    Header h;
    if (object != "matrix")
        throw matrix_error("Failed to parse header: Expected \"matrix\", got \"" + object + "\"");
    if (format == "coordinate") h.format = Format::coordinate;
    else if (format == "array") h.format = Format::array;
    else throw matrix_error("Expected \"coordinate\" or \"array\", got \"" + format + "\"");
    if (field == "real") h.field = Field::real;
    else if (field == "complex") h.field = Field::complex;
    else if (field == "integer") h.field = Field::integer;
    else if (field == "pattern") h.field = Field::pattern;
    else throw matrix_error("Expected \"real\", \"complex\", \"integer\", or \"pattern\", got \"" + field + "\"");
    if (symmetry == "general") h.symmetry = Symmetry::general;
    else if (symmetry == "symmetric") h.symmetry = Symmetry::symmetric;
    else if (symmetry == "skew-symmetric") h.symmetry = Symmetry::skew_symmetric;
    else if (symmetry == "hermitian") h.symmetry = Symmetry::hermitian;
    else throw matrix_error("Expected \"general\", \"symmetric\", \"skew-symmetric\", or \"hermitian\", got \"" + symmetry + "\"");
    return h;
}

bool parse_i32(char const * b, char const * e, int32_t & out)
{
    if (b < e && *b == '+') {
        ++b;
        if (b < e && (*b == '+' || *b == '-'))
            return false; // "+-1": one sign at most
    }
    auto r = std::from_chars(b, e, out);
    return r.ec == std::errc() && r.ptr == e;
}

Size parse_size(std::string const & line, Format format)
{
    std::istringstream s(line);
    std::string tok[3];
    s >> tok[0] >> tok[1] >> tok[2];
    Size size;
    if (!parse_i32(tok[0].data(), tok[0].data() + tok[0].size(), size.rows))
        throw matrix_error("Failed to parse size: Integer overflow when reading number of rows");
    if (!parse_i32(tok[1].data(), tok[1].data() + tok[1].size(), size.columns))
        throw matrix_error("Failed to parse size: Integer overflow when reading number of columns");
    if (format == Format::array)
        return size;
    if (!parse_i32(tok[2].data(), tok[2].data() + tok[2].size(), size.num_entries))
        throw matrix_error("Failed to parse size: Integer overflow when reading number of non-zeros");
    if (size.rows < 0 || size.columns < 0 || size.num_entries < 0)
        throw matrix_error("Failed to parse size: negative dimension");
    return size;
}

bool parse_f64(char const * b, char const * e, double & out)
{
    if (b < e && *b == '+') {
        ++b;
        if (b < e && (*b == '+' || *b == '-'))
            return false; // "+-0.5": one sign at most
    }
    auto r = std::from_chars(b, e, out);
    if (r.ec == std::errc() && r.ptr == e)
        return true;
    // from_chars reports out-of-range for values strtod would round to 0 / inf; let strtod decide
    std::string tmp(b, e);
    char * end = nullptr;
    errno = 0;
    out = std::strtod(tmp.c_str(), &end);
    return end && *end == '\0' && end != tmp.c_str();
}

struct Chunk
{
    std::size_t begin, end; // byte range, boundaries on whitespace
    long long tokens = 0;
    long long first_token = 0;
};

} // namespace

Matrix fromBuffer(char const * d, std::size_t n)
{
    std::size_t pos = 0;
    Header const header = parse_header(take_line(d, n, pos));
    std::vector<std::string> comments;
    while (pos < n && d[pos] == '%')
        comments.push_back(take_line(d, n, pos));
    if (pos >= n)
        throw matrix_error("Failed to parse size");
    Size const size = parse_size(take_line(d, n, pos), header.format);
    if (header.format == Format::array)
        return Matrix(header, comments, size, {}, {}, {});

    int const per = header.field == Field::pattern ? 2 : (header.field == Field::complex ? 4 : 3);
    std::size_t const N = (std::size_t) size.num_entries;
    std::vector<index_type> vi(N), vj(N);
    std::vector<real_type> va(header.field == Field::pattern ? 0 : N);
    std::vector<real_type> vimag(header.field == Field::complex ? N : 0);

    // cut the entry text into chunks on whitespace boundaries, one or more per thread
#ifdef _OPENMP
    int const threads = omp_get_max_threads();
#else
    int const threads = 1;
#endif
    std::size_t const body = n - pos;
    std::size_t nchunks = std::max<std::size_t>(1, std::min<std::size_t>((std::size_t) threads * 4, body / 65536));
    std::vector<Chunk> chunks(nchunks);
    {
        std::size_t b = pos;
        for (std::size_t c = 0; c < nchunks; ++c) {
            std::size_t e = (c + 1 == nchunks) ? n : pos + body * (c + 1) / nchunks;
            if (e < b)
                e = b;
            while (e < n && !is_space(d[e]))
                ++e; // never split a token
            chunks[c].begin = b;
            chunks[c].end = e;
            b = e;
        }
    }
    // pass 1: tokens per chunk (a file below 64 KiB per chunk is one chunk and is parsed by the
    // calling thread alone: waking a team costs more than the parse)
#pragma omp parallel for schedule(dynamic, 1) if (nchunks > 1)
    for (std::size_t c = 0; c < nchunks; ++c) {
        long long t = 0;
        std::size_t k = chunks[c].begin, e = chunks[c].end;
        while (k < e) {
            while (k < e && is_space(d[k]))
                ++k;
            if (k < e)
                ++t;
            while (k < e && !is_space(d[k]))
                ++k;
        }
        chunks[c].tokens = t;
    }
    long long total = 0;
    for (auto & c : chunks) {
        c.first_token = total;
        total += c.tokens;
    }
    long long const wanted = (long long) N * per;
    if (total < wanted) {
        std::ostringstream s;
        s << "Failed to parse entries: Expected " << N << " entries, got " << total / per << " entries.";
        throw matrix_error(s.str());
    }
    // pass 2: every chunk knows the global index of its first token, hence which field of
    // which entry each token is
    bool bad = false;
    long long bad_token = -1;
#pragma omp parallel for schedule(dynamic, 1) if (nchunks > 1)
    for (std::size_t c = 0; c < nchunks; ++c) {
        long long t = chunks[c].first_token;
        std::size_t k = chunks[c].begin, e = chunks[c].end;
        while (k < e && t < wanted) {
            while (k < e && is_space(d[k]))
                ++k;
            if (k >= e)
                break;
            std::size_t b = k;
            while (k < e && !is_space(d[k]))
                ++k;
            std::size_t const entry = (std::size_t) (t / per);
            int const f = (int) (t % per);
            bool ok;
            if (f == 0) {
                ok = parse_i32(d + b, d + k, vi[entry]);
            } else if (f == 1) {
                ok = parse_i32(d + b, d + k, vj[entry]);
            } else if (f == 2) {
                if (header.field == Field::integer) {
                    int32_t iv = 0;
                    ok = parse_i32(d + b, d + k, iv);
                    va[entry] = (double) iv;
                } else {
                    ok = parse_f64(d + b, d + k, va[entry]);
                }
            } else {
                ok = parse_f64(d + b, d + k, vimag[entry]);
            }
            if (!ok) {
#pragma omp critical(mtx_parse_error)
                {
                    if (!bad || t < bad_token) {
                        bad = true;
                        bad_token = t;
                    }
                }
            }
            ++t;
        }
    }
    if (bad) {
        std::ostringstream s;
        s << "Failed to parse entries: bad token in entry " << bad_token / per + 1;
        throw matrix_error(s.str());
    }
    return Matrix(header, std::move(comments), size, std::move(vi), std::move(vj), std::move(va),
                  std::move(vimag));
}

Matrix fromStream(std::istream & i)
{
    std::string text((std::istreambuf_iterator<char>(i)), std::istreambuf_iterator<char>());
    return fromBuffer(text.data(), text.size());
}

// ------------------------------------------------------------------------------------
// files: plain, gzip, gzip'ed tar
// ------------------------------------------------------------------------------------
namespace {

bool ends_with(std::string const & s, std::string const & t)
{
    return s.size() > t.size() && std::equal(t.rbegin(), t.rend(), s.rbegin());
}

// Inflate a gzip (or zlib) stream from `f`, handing each decompressed block to `sink`;
// sink returns false to stop early.
void inflate_stream(std::ifstream & f, std::function<bool(unsigned char const *, std::size_t)> const & sink)
{
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    int rc = inflateInit2(&zs, 15 + 32); // gzip or zlib header, auto-detected
    if (rc != Z_OK)
        throw matrix_error(std::string("inflateInit2: ") + zError(rc));
    std::vector<unsigned char> in(1 << 20), out(1 << 22);
    bool done = false;
    while (!done) {
        f.read(reinterpret_cast<char *>(in.data()), (std::streamsize) in.size());
        std::streamsize got = f.gcount();
        if (got <= 0)
            break;
        zs.next_in = in.data();
        zs.avail_in = (uInt) got;
        while (zs.avail_in > 0 && !done) {
            zs.next_out = out.data();
            zs.avail_out = (uInt) out.size();
            rc = inflate(&zs, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END) {
                inflateEnd(&zs);
                throw matrix_error(std::string("inflate: ") + zError(rc));
            }
            std::size_t produced = out.size() - zs.avail_out;
            if (produced && !sink(out.data(), produced))
                done = true;
            if (rc == Z_STREAM_END)
                done = true;
        }
    }
    inflateEnd(&zs);
}

// Incremental ustar reader: collects the data of the first member whose name starts with
// `member` (the reference compares with strncmp over the wanted name's length,
// src/util/tarstream.cpp:74-78).
class TarMember
{
public:
    explicit TarMember(std::string member) : member_(std::move(member)) {}

    bool feed(unsigned char const * p, std::size_t n) // false = finished
    {
        while (n > 0) {
            if (state_ == State::header) {
                std::size_t take = std::min<std::size_t>(512 - hdr_fill_, n);
                std::memcpy(hdr_ + hdr_fill_, p, take);
                hdr_fill_ += take;
                p += take;
                n -= take;
                if (hdr_fill_ < 512)
                    return true;
                hdr_fill_ = 0;
                bool all_zero = true;
                for (unsigned char c : hdr_)
                    if (c) {
                        all_zero = false;
                        break;
                    }
                if (all_zero) {
                    if (++zero_blocks_ == 2)
                        return false; // end of archive
                    continue;
                }
                zero_blocks_ = 0;
                std::uint64_t size = numeric_field(hdr_ + 124, 12);
                std::size_t len = std::min<std::size_t>(member_.size(), 100);
                bool match = std::strncmp(member_.c_str(), reinterpret_cast<char const *>(hdr_), len) == 0;
                remaining_ = size;
                padding_ = (512 - size % 512) % 512;
                if (match) {
                    found = true;
                    data.reserve((std::size_t) size);
                    state_ = State::copy;
                } else {
                    state_ = State::skip;
                }
                if (remaining_ == 0 && padding_ == 0) {
                    if (state_ == State::copy)
                        return false;
                    state_ = State::header;
                }
            } else {
                std::size_t take = (std::size_t) std::min<std::uint64_t>(remaining_, n);
                if (state_ == State::copy)
                    data.insert(data.end(), p, p + take);
                p += take;
                n -= take;
                remaining_ -= take;
                if (remaining_ == 0) {
                    if (state_ == State::copy)
                        return false; // got the member
                    std::size_t pad = (std::size_t) std::min<std::uint64_t>(padding_, n);
                    p += pad;
                    n -= pad;
                    padding_ -= pad;
                    if (padding_ == 0)
                        state_ = State::header;
                    else
                        state_ = State::pad;
                }
            }
            if (state_ == State::pad) {
                std::size_t pad = (std::size_t) std::min<std::uint64_t>(padding_, n);
                p += pad;
                n -= pad;
                padding_ -= pad;
                if (padding_ == 0)
                    state_ = State::header;
            }
        }
        return true;
    }

    bool found = false;
    std::vector<char> data;

private:
    enum class State { header, copy, skip, pad };

    static std::uint64_t numeric_field(unsigned char const * s, std::size_t n)
    {
        if (s[0] & 0x80) { // base-256
            std::uint64_t x = s[0] & 0x7F;
            for (std::size_t i = 1; i < n; ++i)
                x = x * 256 + s[i];
            return x;
        }
        std::uint64_t x = 0;
        std::size_t i = 0;
        while (i < n && (s[i] == ' ' || s[i] == '\0'))
            ++i;
        for (; i < n && s[i] >= '0' && s[i] <= '7'; ++i)
            x = x * 8 + (std::uint64_t) (s[i] - '0');
        return x;
    }

    std::string member_;
    State state_ = State::header;
    unsigned char hdr_[512];
    std::size_t hdr_fill_ = 0;
    int zero_blocks_ = 0;
    std::uint64_t remaining_ = 0, padding_ = 0;
};

} // namespace

Matrix load_tar_gz_member(std::string const & path, std::string const & member)
{
    std::ifstream f(path, std::ios::binary);
    if (!f)
        throw matrix_error(std::strerror(errno));
    TarMember tar(member);
    inflate_stream(f, [&](unsigned char const * p, std::size_t n) { return tar.feed(p, n); });
    if (!tar.found)
        throw matrix_error("Failed to parse header: Expected \"%%MatrixMarket\", got \"\"");
    return fromBuffer(tar.data.data(), tar.data.size());
}

namespace {

Matrix load_file(std::string const & path, std::ostream & o, bool verbose)
{
    std::ifstream f(path, std::ios::binary);
    if (!f)
        throw matrix_error(std::strerror(errno));

    std::string tar_suffix;
    if (ends_with(path, ".tar.gz")) tar_suffix = ".tar.gz";
    else if (ends_with(path, ".tgz")) tar_suffix = ".tgz";

    if (!tar_suffix.empty()) {
        auto start = path.find_last_of('/');
        start = (start == std::string::npos) ? 0 : start + 1;
        std::string name = path.substr(start, path.size() - tar_suffix.size() - start);
        std::string member = name + "/" + name + ".mtx";
        if (verbose)
            o << "Loading compressed matrix from " << path << ':' << member << '\n';
        f.close();
        return load_tar_gz_member(path, member);
    }
    if (ends_with(path, ".gz")) {
        std::vector<char> text;
        inflate_stream(f, [&](unsigned char const * p, std::size_t n) {
            text.insert(text.end(), p, p + n);
            return true;
        });
        return fromBuffer(text.data(), text.size());
    }
    std::vector<char> text;
    f.seekg(0, std::ios::end);
    std::streamoff len = f.tellg();
    f.seekg(0, std::ios::beg);
    if (len > 0) {
        text.resize((std::size_t) len);
        f.read(text.data(), len);
        text.resize((std::size_t) f.gcount());
    }
    return fromBuffer(text.data(), text.size());
}

} // namespace

// The reference's suffix parser (src/matrix/matrix-market.cpp:786-802): the last "__RCM" and everything behind it goes,
// then the last "__GP" with the number behind it.  "__GPX<n>" (extension) is the same position with an X in front of the number
// -- the reference would strip it just the same and read no number.
ReorderSuffixes parse_reorder_suffixes(std::string const & path)
{
    ReorderSuffixes r;
    r.file = path;
    auto pos = r.file.rfind("__RCM");
    if (pos != std::string::npos) {
        r.rcm = true;
        r.file.erase(pos);
    }
    pos = r.file.rfind("__GP");
    if (pos != std::string::npos) {
        std::size_t at = pos + 4;
        if (at < r.file.size() && r.file[at] == 'X') {
            r.gpx = true;
            ++at;
        } else {
            r.gp = true;
        }
        if (at < r.file.size())
            r.nparts = std::atoi(r.file.c_str() + at);
        r.file.erase(pos);
    }
    return r;
}

std::string reordering_of(std::string const & path)
{
    ReorderSuffixes const s = parse_reorder_suffixes(path);
    std::string out;
    if (s.rcm)
        out = "rcm";
    if (s.gp)
        out += std::string(out.empty() ? "" : "+") + "gp (no METIS in this build: order unchanged, as in a reference build without USE_METIS)";
    if (s.gpx)
        out += std::string(out.empty() ? "" : "+") + "gpx:" + kPartitionerName + ":" + std::to_string(s.nparts <= 1 ? 16 : s.nparts);
    return out;
}

Matrix load_matrix(std::string const & path, std::ostream & o, bool verbose)
{
    // "<file>__RCM" / "<file>__GP<n>": load <file>, then reorder it
    // (src/matrix/matrix-market.cpp:782-802); "__GPX<n>" is this build's extension (matrix-reorder.hpp)
    ReorderSuffixes const sfx = parse_reorder_suffixes(path);
    std::string const & file = sfx.file;
    bool const rcm = sfx.rcm, gp = sfx.gp || sfx.gpx;
    if (verbose) {
        o << "Loading matrix from " << file << '\n';
        if (rcm)
            o << "The input matrix will be reordered using reverse Cuthill-McKee\n";
        if (gp)
            o << "The input matrix will be reordered using graph partitioning\n";
    }
    // optional binary cache of the parsed entries (matrix-cache.hpp); reordering comes after it
    Matrix m;
    std::string const cache = cache_directory();
    if (synthetic::is_spec(file)) {
        // EXTENSION: "synthetic:<family>[:<parameters>]" is generated, not read (synthetic.hpp)
        m = synthetic::generate(file);
    } else if (!cache.empty() && load_cached(file, cache, m)) {
        if (verbose)
            o << "Read the parsed entries from " << cache_file_for(file, cache) << '\n';
    } else {
        m = load_file(file, o, verbose);
        if (!cache.empty())
            store_cached(file, cache, m);
    }
    if (rcm)
        m = permute(m, find_new_order_RCM(m, o, verbose));
    if (sfx.gpx)
        m = permute(m, find_new_order_GPX(m, sfx.nparts, o, verbose));
    else if (sfx.gp) // (the identity: the reference's Matrix::permute would change nothing, or -- for a matrix that is not square
        (void) find_new_order_GP(m, sfx.nparts, o, verbose); //  and real -- say so on stderr and change nothing either, :309-334)
    return m;
}

// ------------------------------------------------------------------------------------
// ordering
// ------------------------------------------------------------------------------------
namespace {

// Stable order by (major, minor): counting sort on the major index, then each bucket
// (one row / column) is ordered by the minor index with a stable sort.
std::vector<size_type> two_key_order(std::vector<index_type> const & major,
                                     std::vector<index_type> const & minor, index_type nmajor,
                                     char const * what)
{
    std::size_t const N = major.size();
    std::vector<std::size_t> start((std::size_t) std::max<index_type>(nmajor, 0) + 1, 0);
    for (index_type r : major) {
        if (r < 1 || r > nmajor)
            throw matrix_error(std::string(what) + " index out of bounds: " + std::to_string(r));
        ++start[(std::size_t) r];
    }
    for (std::size_t r = 1; r < start.size(); ++r)
        start[r] += start[r - 1];
    std::vector<size_type> order(N);
    {
        std::vector<std::size_t> fill(start.begin(), start.end() - 1);
        for (std::size_t k = 0; k < N; ++k)
            order[fill[(std::size_t) major[k] - 1]++] = (size_type) k;
    }
#pragma omp parallel for schedule(dynamic, 1024) if (N > kSerialBelow)
    for (long long r = 0; r < (long long) start.size() - 1; ++r) {
        auto b = order.begin() + (std::ptrdiff_t) start[(std::size_t) r];
        auto e = order.begin() + (std::ptrdiff_t) start[(std::size_t) r + 1];
        if (e - b > 1 && !std::is_sorted(b, e, [&](size_type p, size_type q) { return minor[p] < minor[q]; }))
            std::stable_sort(b, e, [&](size_type p, size_type q) { return minor[p] < minor[q]; });
    }
    return order;
}

template <typename T>
std::vector<T> permuted(std::vector<T> const & v, std::vector<size_type> const & order)
{
    if (v.empty())
        return {};
    std::vector<T> out(order.size());
    for (std::size_t k = 0; k < order.size(); ++k)
        out[k] = v[(std::size_t) order[k]];
    return out;
}

} // namespace

RowMajorEntries row_major_entries(Matrix const & m)
{
    RowMajorEntries out;
    index_type const rows = m.rows();
    auto const & ri = m.row_indices();
    auto const & ci = m.column_indices();
    std::size_t const N = ri.size();
    bool const pattern = m.field() == Field::pattern;
    std::vector<real_type> const ones;
    auto const & va = m.a_;

    auto const len = m.row_lengths(); // validates the row indices
    for (auto c : ci)
        if (c < 1 || c > m.columns())
            throw matrix_error("Column index out of bounds: " + std::to_string(c));
    out.start.assign((std::size_t) rows + 1, 0);
    for (index_type r = 0; r < rows; ++r)
        out.start[(std::size_t) r + 1] = out.start[(std::size_t) r] + (std::size_t) len[(std::size_t) r];
    out.col.resize(N);
    out.val.resize(N);
    std::vector<std::size_t> cursor(out.start.begin(), out.start.end() - 1);

#ifdef _OPENMP
    int const threads = N > kSerialBelow ? std::max(1, omp_get_max_threads()) : 1;
#else
    int const threads = 1;
#endif
    // every thread owns a contiguous range of rows holding ~N/threads entries, reads the whole
    // entry list and places the entries of its rows: no atomics, writes stay inside its range
    std::vector<index_type> bound((std::size_t) threads + 1, rows);
    bound[0] = 0;
    for (int t = 1; t < threads; ++t) {
        std::size_t const target = N / (std::size_t) threads * (std::size_t) t;
        bound[(std::size_t) t] = (index_type) (std::lower_bound(out.start.begin(), out.start.end(), target) - out.start.begin());
        bound[(std::size_t) t] = std::min(bound[(std::size_t) t], rows);
    }
#pragma omp parallel for schedule(static, 1) num_threads(threads) if (threads > 1)
    for (int t = 0; t < threads; ++t) {
        index_type const lo = bound[(std::size_t) t], hi = bound[(std::size_t) t + 1];
        if (lo >= hi)
            continue;
        for (std::size_t k = 0; k < N; ++k) {
            index_type const r = ri[k] - 1;
            if (r >= lo && r < hi) {
                std::size_t const dst = cursor[(std::size_t) r]++;
                out.col[dst] = ci[k] - 1;
                out.val[dst] = pattern ? 1.0 : va[k];
            }
        }
    }
    // order each row by column; rows arrive in file order, so already-sorted rows are common
#pragma omp parallel for schedule(dynamic, 2048) if (N > kSerialBelow)
    for (long long r = 0; r < (long long) rows; ++r) {
        std::size_t const b = out.start[(std::size_t) r], e = out.start[(std::size_t) r + 1];
        bool sorted = true;
        for (std::size_t k = b + 1; k < e; ++k)
            if (out.col[k] < out.col[k - 1]) {
                sorted = false;
                break;
            }
        if (sorted)
            continue;
        if (e - b <= 64) { // stable insertion sort of (column, value) pairs
            for (std::size_t k = b + 1; k < e; ++k) {
                index_type const c = out.col[k];
                real_type const v = out.val[k];
                std::size_t q = k;
                while (q > b && out.col[q - 1] > c) {
                    out.col[q] = out.col[q - 1];
                    out.val[q] = out.val[q - 1];
                    --q;
                }
                out.col[q] = c;
                out.val[q] = v;
            }
        } else {
            std::vector<std::size_t> idx(e - b);
            std::iota(idx.begin(), idx.end(), b);
            std::stable_sort(idx.begin(), idx.end(), [&](std::size_t p, std::size_t q) { return out.col[p] < out.col[q]; });
            std::vector<index_type> c2(e - b);
            std::vector<real_type> v2(e - b);
            for (std::size_t k = 0; k < e - b; ++k) {
                c2[k] = out.col[idx[k]];
                v2[k] = out.val[idx[k]];
            }
            std::copy(c2.begin(), c2.end(), out.col.begin() + (std::ptrdiff_t) b);
            std::copy(v2.begin(), v2.end(), out.val.begin() + (std::ptrdiff_t) b);
        }
    }
    return out;
}

std::vector<size_type> row_major_order(Matrix const & m)
{
    return two_key_order(m.row_indices(), m.column_indices(), m.rows(), "Row");
}

Matrix sort_matrix_row_major(Matrix const & m)
{
    auto const order = row_major_order(m);
    return Matrix(m.header_, m.comments_, m.size_, permuted(m.i_, order), permuted(m.j_, order),
                  permuted(m.a_, order), permuted(m.imag_, order));
}

Matrix sort_matrix_column_major(Matrix const & m)
{
    auto const order = two_key_order(m.column_indices(), m.row_indices(), m.columns(), "Column");
    return Matrix(m.header_, m.comments_, m.size_, permuted(m.i_, order), permuted(m.j_, order),
                  permuted(m.a_, order), permuted(m.imag_, order));
}

Matrix expand_symmetry(Matrix const & m)
{
    if (m.symmetry() == Symmetry::general || m.format() != Format::coordinate)
        return Matrix(m.header_, m.comments_, m.size_, m.i_, m.j_, m.a_, m.imag_);
    std::size_t const N = m.i_.size();
    std::size_t off = 0;
    for (std::size_t k = 0; k < N; ++k)
        off += (m.i_[k] != m.j_[k]);
    if (N + off > (std::size_t) INT32_MAX)
        throw matrix_error("Failed to expand symmetry: Integer overflow when computing number of non-zeros");
    std::vector<index_type> i(m.i_), j(m.j_);
    std::vector<real_type> a(m.a_), imag(m.imag_);
    i.reserve(N + off);
    j.reserve(N + off);
    double const sign = (m.symmetry() == Symmetry::skew_symmetric) ? -1.0 : 1.0;
    for (std::size_t k = 0; k < N; ++k) {
        if (m.i_[k] == m.j_[k])
            continue;
        i.push_back(m.j_[k]);
        j.push_back(m.i_[k]);
        if (!m.a_.empty())
            a.push_back(sign * m.a_[k]);
        if (!m.imag_.empty())
            imag.push_back(m.symmetry() == Symmetry::hermitian ? -m.imag_[k] : sign * m.imag_[k]);
    }
    Header h = m.header_;
    h.symmetry = Symmetry::general;
    Size s = m.size_;
    s.num_entries = (size_type) i.size();
    return Matrix(h, m.comments_, s, std::move(i), std::move(j), std::move(a), std::move(imag));
}

std::ostream & operator<<(std::ostream & o, Matrix const & m)
{
    static char const * const fields[] = {"real", "complex", "integer", "pattern"};
    static char const * const syms[] = {"general", "symmetric", "skew-symmetric", "hermitian"};
    o << "%%MatrixMarket matrix " << (m.format() == Format::coordinate ? "coordinate" : "array") << ' '
      << fields[(int) m.field()] << ' ' << syms[(int) m.symmetry()] << '\n';
    for (auto const & c : m.comments())
        o << c << '\n';
    o << m.rows() << ' ' << m.columns() << ' ' << m.num_entries() << '\n';
    auto const & i = m.row_indices();
    auto const & j = m.column_indices();
    auto const a = m.values_real();
    for (std::size_t k = 0; k < i.size(); ++k) {
        o << i[k] << ' ' << j[k];
        if (m.field() == Field::complex)
            o << ' ' << a[k] << ' ' << m.values_imag()[k];
        else if (m.field() != Field::pattern)
            o << ' ' << a[k];
        o << '\n';
    }
    return o;
}

void write_matrix(std::string const & path, Matrix const & m)
{
    static char const * const fields[] = {"real", "complex", "integer", "pattern"};
    static char const * const syms[] = {"general", "symmetric", "skew-symmetric", "hermitian"};
    std::string head = std::string("%%MatrixMarket matrix ") + (m.format() == Format::coordinate ? "coordinate" : "array") + ' '
        + fields[(int) m.field()] + ' ' + syms[(int) m.symmetry()] + '\n';
    for (auto const & c : m.comments())
        head += c + '\n';
    head += std::to_string(m.rows()) + ' ' + std::to_string(m.columns()) + ' ' + std::to_string(m.num_entries()) + '\n';
    if (m.format() != Format::coordinate)
        throw matrix::matrix_error("write_matrix: only coordinate matrices are written");
    auto const & ri = m.row_indices();
    auto const & ci = m.column_indices();
    auto const a = m.values_real();
    bool const complex = m.field() == Field::complex, pattern = m.field() == Field::pattern, integer = m.field() == Field::integer;
    std::size_t const n = ri.size();
    // one text block per chunk of entries, formatted in parallel, written in order
    std::size_t const chunk = 1u << 20;
    std::size_t const nchunks = (n + chunk - 1) / chunk;
    bool const gz = path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
    std::FILE * f = nullptr;
    gzFile g = nullptr;
    if (gz)
        g = gzopen(path.c_str(), "wb1");
    else
        f = std::fopen(path.c_str(), "wb");
    if (!f && !g)
        throw matrix::matrix_error(path + ": cannot open for writing");
    auto put = [&](char const * data, std::size_t len) {
        bool ok = true;
        if (gz) {
            for (std::size_t off = 0; ok && off < len;) {
                unsigned const part = (unsigned) std::min<std::size_t>(len - off, 1u << 30);
                ok = gzwrite(g, data + off, part) == (int) part;
                off += part;
            }
        } else {
            ok = std::fwrite(data, 1, len, f) == len;
        }
        if (!ok) {
            if (f) std::fclose(f);
            if (g) gzclose(g);
            throw matrix::matrix_error(path + ": write failed");
        }
    };
    put(head.data(), head.size());
    int const group = 16; // chunks formatted side by side before they are written
    std::vector<std::string> text((std::size_t) group);
    for (std::size_t c0 = 0; c0 < nchunks; c0 += group) {
        std::size_t const c1 = std::min(nchunks, c0 + group);
#pragma omp parallel for schedule(dynamic, 1)
        for (long long c = (long long) c0; c < (long long) c1; ++c) {
            std::string & out = text[(std::size_t) c - c0];
            out.clear();
            out.reserve(chunk * 40);
            char buf[96];
            std::size_t const k1 = std::min(n, ((std::size_t) c + 1) * chunk);
            for (std::size_t k = (std::size_t) c * chunk; k < k1; ++k) {
                char * q = buf;
                q = std::to_chars(q, buf + sizeof(buf), (long long) ri[k]).ptr;
                *q++ = ' ';
                q = std::to_chars(q, buf + sizeof(buf), (long long) ci[k]).ptr;
                if (!pattern) {
                    *q++ = ' ';
                    if (integer)
                        q = std::to_chars(q, buf + sizeof(buf), (long long) a[k]).ptr;
                    else
                        q = std::to_chars(q, buf + sizeof(buf), (double) a[k]).ptr;
                    if (complex) {
                        *q++ = ' ';
                        q = std::to_chars(q, buf + sizeof(buf), (double) m.values_imag()[k]).ptr;
                    }
                }
                *q++ = '\n';
                out.append(buf, (std::size_t) (q - buf));
            }
        }
        for (std::size_t c = c0; c < c1; ++c)
            put(text[c - c0].data(), text[c - c0].size());
    }
    bool ok = true;
    if (f) ok = std::fclose(f) == 0;
    if (g) ok = gzclose(g) == Z_OK;
    if (!ok)
        throw matrix::matrix_error(path + ": close failed");
}

} // namespace matrix_market
