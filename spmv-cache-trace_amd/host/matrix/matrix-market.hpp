// matrix-market.hpp -- Matrix Market input: .mtx, .gz, .tgz / .tar.gz (SuiteSparse tarballs).
//
// Same input surface as the reference loader (src/matrix/matrix-market.cpp):
//   * banner "%%MatrixMarket" exact case, the other header words case-insensitive (:339-436)
//   * comment lines = lines starting with '%' right after the banner (:438-447)
//   * size line "rows columns entries" (array format: "rows columns", parsed for its size only)
//   * entries are whitespace-separated records, not necessarily one per line (:484-528)
//   * field real | complex | integer | pattern; values_real() gives the real part of complex
//     entries and 1.0 for pattern entries (:243-277)
//   * the symmetry word is parsed and KEPT, but entries are never mirrored: a symmetric file
//     yields only the triangle it stores unless expand_symmetry() is asked for
//   * a tarball named <name>.tar.gz is searched for the member <name>/<name>.mtx (:746-775)
//   * a path ending in "__RCM" / "__GP<n>" loads the file before the suffix and reorders it
//     (:782-802; matrix-reorder.hpp)
//
// New here: entries live in structure-of-arrays form (no AoS -> SoA copies), the text is
// tokenised in memory by all OpenMP threads, and sorting is a counting sort on the row index.
#pragma once

#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>

namespace matrix_market {

typedef int32_t size_type;
typedef int32_t index_type;
typedef double real_type;

enum class Object { matrix };
enum class Format { coordinate, array };
enum class Field { real, complex, integer, pattern };
enum class Symmetry { general, symmetric, skew_symmetric, hermitian };

struct Header
{
    Object object = Object::matrix;
    Format format = Format::coordinate;
    Field field = Field::real;
    Symmetry symmetry = Symmetry::general;
};

struct Size
{
    index_type rows = 0;
    index_type columns = 0;
    size_type num_entries = 0;
};

struct RowMajorEntries;

class Matrix
{
public:
    Matrix() = default;
    Matrix(Header header, std::vector<std::string> comments, Size size, std::vector<index_type> i,
           std::vector<index_type> j, std::vector<real_type> a, std::vector<real_type> imag = {});

    Header const & header() const { return header_; }
    Format format() const { return header_.format; }
    Field field() const { return header_.field; }
    Symmetry symmetry() const { return header_.symmetry; }
    std::vector<std::string> const & comments() const { return comments_; }
    Size const & size() const { return size_; }
    index_type rows() const { return size_.rows; }
    index_type columns() const { return size_.columns; }
    size_type num_entries() const { return size_.num_entries; }

    // 1-based, in file order
    std::vector<index_type> const & row_indices() const { return i_; }
    std::vector<index_type> const & column_indices() const { return j_; }
    // real: the value; complex: its real part; integer: the value as double; pattern: 1.0
    std::vector<real_type> values_real() const;
    std::vector<real_type> const & values_imag() const { return imag_; }

    index_type max_row_length() const;
    std::vector<index_type> row_lengths() const;

private:
    friend Matrix sort_matrix_row_major(Matrix const &);
    friend Matrix sort_matrix_column_major(Matrix const &);
    friend Matrix expand_symmetry(Matrix const &);
    friend struct RowMajorEntries row_major_entries(Matrix const &);
    Header header_;
    std::vector<std::string> comments_;
    Size size_;
    std::vector<index_type> i_, j_;
    std::vector<real_type> a_;    // empty for pattern
    std::vector<real_type> imag_; // complex only
};

// Parse a whole Matrix Market document held in memory / read from a stream.
Matrix fromBuffer(char const * data, std::size_t size);
Matrix fromStream(std::istream & i);

// Load by path: plain text, or by suffix .gz (gzip), .tgz / .tar.gz (gzip'ed tar with member
// <name>/<name>.mtx).  `o` receives progress lines when verbose.
Matrix load_matrix(std::string const & path, std::ostream & o, bool verbose = false);
// The named member of a gzip'ed tar archive (what load_matrix does for *.tar.gz after it has
// derived the member name from the path).
Matrix load_tar_gz_member(std::string const & path, std::string const & member);

// The entries grouped by row and ordered by column inside each row (stable: duplicate (i, j) keep
// their file order), built by all OpenMP threads: what every row-major format converter starts from.
// start[r] .. start[r+1] are the entries of row r; col is 0-based; val follows values_real().
struct RowMajorEntries
{
    std::vector<std::size_t> start;
    std::vector<index_type> col;
    std::vector<real_type> val;
};
RowMajorEntries row_major_entries(Matrix const & m);

// Stable permutation that orders the entries by (row, column) / (column, row).
std::vector<size_type> row_major_order(Matrix const & m);
Matrix sort_matrix_row_major(Matrix const & m);
Matrix sort_matrix_column_major(Matrix const & m);

// EXTENSION (not in the reference, off by default): mirror the off-diagonal entries of a
// symmetric / skew-symmetric / hermitian file so the full matrix is multiplied.
Matrix expand_symmetry(Matrix const & m);

std::ostream & operator<<(std::ostream & o, Matrix const & m);

// EXTENSION: the matrix as a Matrix Market file that load_matrix reads back bit for bit -- values in their
// shortest decimal form that round-trips (std::to_chars), entries formatted by all OpenMP threads.  `.gz`
// paths are compressed with zlib.  Used by `--write-mtx` (files of the generated stand-ins at full size).
void write_matrix(std::string const & path, Matrix const & m);

} // namespace matrix_market
