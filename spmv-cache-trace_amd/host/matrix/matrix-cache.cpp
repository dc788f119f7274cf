#include "matrix-cache.hpp"

#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

namespace matrix_market
{

namespace
{

char const kMagic[8] = {'S', 'P', 'M', 'V', 'M', 'M', '0', '2'}; // 02: payload checksum at the end

struct Identity
{
    std::string path; // canonical
    std::int64_t size = 0;
    std::int64_t mtime_ns = 0;
};

bool identify(std::string const & source, Identity & id)
{
    char real[PATH_MAX];
    if (!realpath(source.c_str(), real))
        return false;
    struct stat st;
    if (stat(real, &st) != 0 || !S_ISREG(st.st_mode))
        return false;
    id.path = real;
    id.size = (std::int64_t) st.st_size;
    id.mtime_ns = (std::int64_t) st.st_mtim.tv_sec * 1000000000ll + (std::int64_t) st.st_mtim.tv_nsec;
    return true;
}

std::uint64_t fnv1a(std::string const & s)
{
    std::uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) {
        h ^= c;
        h *= 1099511628211ull;
    }
    return h;
}

struct File
{
    std::FILE * f = nullptr;
    explicit File(std::FILE * f) : f(f) {}
    ~File() { if (f) std::fclose(f); }
    File(File const &) = delete;
    File & operator=(File const &) = delete;
};

template <typename T>
bool put(std::FILE * f, T const & v) { return std::fwrite(&v, sizeof(T), 1, f) == 1; }
template <typename T>
bool get(std::FILE * f, T & v) { return std::fread(&v, sizeof(T), 1, f) == 1; }

// Payload checksum: a multiply-and-add hash over the arrays' bytes (order-dependent; one pass at
// memory speed).  A flipped or missing byte anywhere makes the cache file unusable -> the source is parsed.
std::uint64_t checksum(void const * data, std::size_t bytes, std::uint64_t h)
{
    unsigned char const * p = static_cast<unsigned char const *>(data);
    std::size_t const words = bytes / 8;
    for (std::size_t k = 0; k < words; ++k) {
        std::uint64_t w;
        std::memcpy(&w, p + 8 * k, 8);
        h = (h ^ w) * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    }
    for (std::size_t k = 8 * words; k < bytes; ++k)
        h = (h ^ p[k]) * 0x100000001B3ull;
    return h;
}

template <typename T>
bool put_array(std::FILE * f, std::vector<T> const & a, std::uint64_t & sum)
{
    std::uint64_t n = a.size();
    sum = checksum(a.data(), (std::size_t) n * sizeof(T), sum ^ n);
    return put(f, n) && (n == 0 || std::fwrite(a.data(), sizeof(T), (std::size_t) n, f) == n);
}

// `budget` = bytes the cache file still has: an array can never be longer than what is left of the
// file, whatever its header claims (a corrupt count must not become a multi-gigabyte allocation)
template <typename T>
bool get_array(std::FILE * f, std::vector<T> & a, std::uint64_t limit, std::uint64_t & budget, std::uint64_t & sum)
{
    std::uint64_t n = 0;
    if (budget < sizeof n || !get(f, n) || n > limit)
        return false;
    budget -= sizeof n;
    if (n > budget / sizeof(T))
        return false;
    try {
        a.resize((std::size_t) n);
    } catch (std::bad_alloc const &) {
        return false;
    }
    if (n != 0 && std::fread(a.data(), sizeof(T), (std::size_t) n, f) != n)
        return false;
    budget -= n * sizeof(T);
    sum = checksum(a.data(), (std::size_t) n * sizeof(T), sum ^ n);
    return true;
}

} // namespace

std::string cache_directory()
{
    char const * e = std::getenv("SPMV_MATRIX_CACHE");
    return e ? std::string(e) : std::string();
}

std::string cache_file_for(std::string const & source, std::string const & directory)
{
    Identity id;
    if (directory.empty() || !identify(source, id))
        return std::string();
    std::string base = id.path.substr(id.path.find_last_of('/') + 1);
    char hex[32];
    std::snprintf(hex, sizeof hex, "%016llx",
                  (unsigned long long) fnv1a(id.path + "|" + std::to_string(id.size) + "|" + std::to_string(id.mtime_ns)));
    return directory + "/" + base + "." + hex + ".mmbin";
}

bool load_cached(std::string const & source, std::string const & directory, Matrix & m)
{
    Identity id;
    std::string const path = cache_file_for(source, directory);
    if (path.empty() || !identify(source, id))
        return false;
    File file(std::fopen(path.c_str(), "rb"));
    if (!file.f)
        return false;
    std::FILE * f = file.f;
    char magic[8];
    std::int64_t size = 0, mtime = 0, nnz = 0;
    std::int32_t object = 0, format = 0, field = 0, symmetry = 0, rows = 0, cols = 0;
    if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, kMagic, 8) != 0)
        return false;
    if (!get(f, size) || !get(f, mtime) || size != id.size || mtime != id.mtime_ns)
        return false; // the source changed under the same name
    if (!get(f, object) || !get(f, format) || !get(f, field) || !get(f, symmetry) || !get(f, rows) || !get(f, cols)
        || !get(f, nnz))
        return false;
    if (object != 0 || format < 0 || format > 1 || field < 0 || field > 3 || symmetry < 0 || symmetry > 3 || rows < 0
        || cols < 0 || nnz < 0)
        return false;
    std::uint64_t ncomments = 0;
    if (!get(f, ncomments) || ncomments > (1u << 20))
        return false;
    std::vector<std::string> comments((std::size_t) ncomments);
    for (auto & c : comments) {
        std::uint64_t len = 0;
        if (!get(f, len) || len > (1u << 24))
            return false;
        c.resize((std::size_t) len);
        if (len && std::fread(&c[0], 1, (std::size_t) len, f) != len)
            return false;
    }
    // array format stores rows * columns values; coordinate format at most INT32_MAX entries
    if (nnz > INT32_MAX || ((Format) format == Format::array && nnz != (std::int64_t) rows * (std::int64_t) cols))
        return false;
    std::uint64_t const limit = (std::uint64_t) nnz;
    std::uint64_t budget = 0;
    {
        long const here = std::ftell(f);
        if (here < 0 || std::fseek(f, 0, SEEK_END) != 0)
            return false;
        long const end = std::ftell(f);
        if (end < here || std::fseek(f, here, SEEK_SET) != 0)
            return false;
        budget = (std::uint64_t) (end - here);
    }
    std::vector<index_type> i, j;
    std::vector<real_type> a, imag;
    std::uint64_t sum = 0x5350'4D56ull, stored = 0;
    if (!get_array(f, i, limit, budget, sum) || !get_array(f, j, limit, budget, sum) || !get_array(f, a, limit, budget, sum)
        || !get_array(f, imag, limit, budget, sum))
        return false;
    if (!get(f, stored) || stored != sum)
        return false; // truncated or altered payload: parse the source instead
    Field const fld = (Field) field;
    bool const coordinate = (Format) format == Format::coordinate;
    if (coordinate && (i.size() != limit || j.size() != limit))
        return false;
    if (a.size() != (fld == Field::pattern ? 0 : limit) || imag.size() != (fld == Field::complex ? limit : 0))
        return false;
    Header h;
    h.object = Object::matrix;
    h.format = (Format) format;
    h.field = fld;
    h.symmetry = (Symmetry) symmetry;
    Size s;
    s.rows = rows;
    s.columns = cols;
    s.num_entries = (size_type) nnz;
    m = Matrix(h, std::move(comments), s, std::move(i), std::move(j), std::move(a), std::move(imag));
    return true;
}

void store_cached(std::string const & source, std::string const & directory, Matrix const & m)
{
    Identity id;
    std::string const path = cache_file_for(source, directory);
    if (path.empty() || !identify(source, id))
        return;
    std::string const tmp = path + ".tmp" + std::to_string((long long) getpid());
    bool ok = false;
    {
        File file(std::fopen(tmp.c_str(), "wb"));
        if (!file.f)
            return;
        std::FILE * f = file.f;
        std::int32_t const object = 0, format = (std::int32_t) m.format(), field = (std::int32_t) m.field(),
                           symmetry = (std::int32_t) m.symmetry(), rows = m.rows(), cols = m.columns();
        std::int64_t const nnz = (std::int64_t) m.num_entries();
        ok = std::fwrite(kMagic, 1, 8, f) == 8 && put(f, id.size) && put(f, id.mtime_ns) && put(f, object)
             && put(f, format) && put(f, field) && put(f, symmetry) && put(f, rows) && put(f, cols) && put(f, nnz);
        std::uint64_t const ncomments = m.comments().size();
        ok = ok && put(f, ncomments);
        for (auto const & c : m.comments()) {
            std::uint64_t const len = c.size();
            ok = ok && put(f, len) && (len == 0 || std::fwrite(c.data(), 1, (std::size_t) len, f) == len);
        }
        std::vector<real_type> const a = m.field() == Field::pattern ? std::vector<real_type>() : m.values_real();
        std::uint64_t sum = 0x5350'4D56ull;
        ok = ok && put_array(f, m.row_indices(), sum) && put_array(f, m.column_indices(), sum) && put_array(f, a, sum)
             && put_array(f, m.values_imag(), sum);
        ok = ok && put(f, sum);
        ok = ok && std::fflush(f) == 0;
    }
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0)
        std::remove(tmp.c_str());
}

} // namespace matrix_market
