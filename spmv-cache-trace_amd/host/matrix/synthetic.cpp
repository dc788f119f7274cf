#include "synthetic.hpp"

#include "matrix-error.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <numeric>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace synthetic {

namespace {

using csr_matrix::index_type;
using csr_matrix::size_type;

// counter-based randomness: every number is a hash of what it belongs to (splitmix64 finaliser)
inline std::uint64_t mix(std::uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline std::uint64_t h2(std::uint64_t a, std::uint64_t b) { return mix(mix(a) ^ (b * 0xD6E8FEB86659FD93ull + 0x2545F4914F6CDD1Dull)); }
inline double u01(std::uint64_t h) { return (double) (h >> 11) * (1.0 / 9007199254740992.0); }   // [0, 1)
inline double u11(std::uint64_t h) { return (double) (h >> 11) * (2.0 / 9007199254740992.0) - 1.0; } // [-1, 1)

std::vector<long long> numbers(std::string const & s)
{
    std::vector<long long> out;
    std::size_t pos = 0;
    while (pos < s.size()) {
        std::size_t end = s.find(',', pos);
        if (end == std::string::npos)
            end = s.size();
        std::string const tok = s.substr(pos, end - pos);
        char * e = nullptr;
        long long const v = std::strtoll(tok.c_str(), &e, 10);
        if (tok.empty() || *e != '\0' || v < 0)
            throw matrix::matrix_error("synthetic matrix: expected non-negative integers, got '" + s + "'");
        out.push_back(v);
        pos = end + 1;
    }
    return out;
}

// Two passes over the rows [rb, re): lengths -> row_ptr, then every row fills its own slice.
template <class Len, class Fill>
csr_matrix::Matrix build(long long rows_total, long long cols, long long rb, long long re, Len len, Fill fill)
{
    if (rows_total > INT32_MAX || cols > INT32_MAX)
        throw matrix::matrix_error("synthetic matrix: more than 2^31-1 rows or columns");
    long long const nr = re - rb;
    csr_matrix::size_array_type row_ptr((std::size_t) nr + 1, 0);
#pragma omp parallel for schedule(static)
    for (long long r = 0; r < nr; ++r)
        row_ptr[(std::size_t) r + 1] = (size_type) len(rb + r);
    long long k = 0;
    for (long long r = 0; r < nr; ++r) {
        k += row_ptr[(std::size_t) r + 1];
        if (k > INT32_MAX)
            throw matrix::matrix_error("synthetic matrix: Integer overflow when computing number of non-zeros");
        row_ptr[(std::size_t) r + 1] = (size_type) k;
    }
    csr_matrix::index_array_type col((std::size_t) k);
    csr_matrix::value_array_type val((std::size_t) k);
#pragma omp parallel for schedule(dynamic, 4096)
    for (long long r = 0; r < nr; ++r)
        fill(rb + r, col.data() + row_ptr[(std::size_t) r], val.data() + row_ptr[(std::size_t) r]);
    return csr_matrix::Matrix((index_type) nr, (index_type) cols, (size_type) k, 1, std::move(row_ptr), std::move(col),
                              std::move(val));
}

// ---- poisson2d: the arrays of python/spmv_amd/synth.py poisson2d(), value for value --------------
// hashed != 0: the "pessimistic twin" -- same structure, every coefficient perturbed by a hash of its place (a
// variable-coefficient operator: 84 M distinct values, so no value dictionary applies)
csr_matrix::Matrix poisson2d(long long n, long long rb, long long re, long long * total, long long hashed = 0)
{
    if (n < 1 || n > 46340)
        throw matrix::matrix_error("synthetic:poisson2d: grid edge must be in 1..46340");
    long long const N = n * n;
    if (total) *total = N;
    if (re < 0) re = N;
    auto len = [n](long long r) {
        long long const i = r / n, j = r % n;
        return 1 + (i > 0) + (j > 0) + (j < n - 1) + (i < n - 1);
    };
    auto fill = [n, hashed](long long r, index_type * c, double * v) {
        long long const i = r / n, j = r % n;
        double * const v0 = v;
        index_type * const c0 = c;
        if (i > 0) { *c++ = (index_type) (r - n); *v++ = -1.0; }
        if (j > 0) { *c++ = (index_type) (r - 1); *v++ = -1.0; }
        *c++ = (index_type) r; *v++ = 4.0;
        if (j < n - 1) { *c++ = (index_type) (r + 1); *v++ = -1.0; }
        if (i < n - 1) { *c++ = (index_type) (r + n); *v++ = -1.0; }
        if (hashed)
            for (long long q = 0; q < v - v0; ++q)
                v0[q] *= 1.0 + 0.25 * u11(h2(0x9015507ull + (std::uint64_t) r, (std::uint64_t) c0[q]));
    };
    return build(N, N, rb, re, len, fill);
}

// ---- kkt: [H 0 A'; 0 R C'; A C 0], unknowns ordered states, controls, multipliers ----------------
// jitter_pct > 0: the "pessimistic twin" -- that share of the 27-point links of A (and, independently, of A') ends
// 3 cells to either side of its grid neighbour, hashed per row, so that no two rows are shifted copies of each
// other any more (the structure is then no longer exactly symmetric: only the multiply's access pattern matters)
csr_matrix::Matrix kkt(long long n, long long rb, long long re, long long * total, long long jitter_pct = 0)
{
    if (n < 1 || n > 1000)
        throw matrix::matrix_error("synthetic:kkt: grid edge must be in 1..1000");
    if (jitter_pct < 0 || jitter_pct > 100)
        throw matrix::matrix_error("synthetic:kkt:<n>,<jitter>: jitter is a percentage");
    long long const ny = n * n * n, nu = 6 * n * n, N = 2 * ny + nu;
    if (total) *total = N;
    if (re < 0) re = N;
    long long const n2 = n * n;
    std::uint64_t const seedA = 0xA11CE, seedH = 0xB0B, seedC = 0xC0DE;
    auto ext = [n](long long a) { return 1 + (a > 0) + (a < n - 1); };
    auto faces = [n](long long a, long long b, long long c) {
        return (a == 0) + (a == n - 1) + (b == 0) + (b == n - 1) + (c == 0) + (c == n - 1);
    };
    auto len = [=](long long r) -> long long {
        if (r < ny) { // state: diagonal of H, then A' (one entry per constraint whose stencil holds this state)
            long long const a = r / n2, b = (r / n) % n, c = r % n;
            return 1 + ext(a) * ext(b) * ext(c);
        }
        if (r < ny + nu)
            return 2; // control: diagonal of R, its boundary cell's constraint
        long long const s = r - ny - nu, a = s / n2, b = (s / n) % n, c = s % n;
        return ext(a) * ext(b) * ext(c) + faces(a, b, c); // constraint: 27-point row of A, controls of its faces
    };
    auto cell_of_control = [=](long long k) {
        long long const f = k / n2, u = (k % n2) / n, v = k % n;
        switch (f) {
        case 0: return (0 * n + u) * n + v;
        case 1: return ((n - 1) * n + u) * n + v;
        case 2: return (u * n + 0) * n + v;
        case 3: return (u * n + (n - 1)) * n + v;
        case 4: return (u * n + v) * n + 0;
        default: return (u * n + v) * n + (n - 1);
        }
    };
    // moves the jittered share of the columns [lo, hi) of a row's stencil part (which lie in [base, base + ny)),
    // keeps them ascending and distinct
    auto jitter = [=](long long r, index_type * lo, index_type * hi, long long base) {
        if (jitter_pct == 0 || hi - lo < 2)
            return;
        for (index_type * q = lo; q < hi; ++q) {
            std::uint64_t const h = h2(0x71773Eull + (std::uint64_t) r, (std::uint64_t) (q - lo));
            if ((long long) (h % 100) < jitter_pct) {
                long long t = (long long) *q + ((h >> 32) & 1 ? 3 : -3);
                t = std::max(base, std::min(base + ny - 1, t));
                *q = (index_type) t;
            }
        }
        std::sort(lo, hi);
        for (index_type * q = lo + 1; q < hi; ++q)
            if (*q <= q[-1])
                *q = q[-1] + 1;
        if ((long long) hi[-1] > base + ny - 1) {
            hi[-1] = (index_type) (base + ny - 1);
            for (index_type * q = hi - 2; q >= lo && *q >= q[1]; --q)
                *q = q[1] - 1;
        }
    };
    auto fill = [=](long long r, index_type * col, double * val) {
        index_type * const col0 = col;
        if (r < ny) {
            long long const a = r / n2, b = (r / n) % n, c = r % n;
            *col++ = (index_type) r;
            *val++ = 2.0 + u11(h2(seedH, (std::uint64_t) r));
            for (int da = -1; da <= 1; ++da)
                for (int db = -1; db <= 1; ++db)
                    for (int dc = -1; dc <= 1; ++dc) {
                        long long const aa = a + da, bb = b + db, cc = c + dc;
                        if (aa < 0 || aa >= n || bb < 0 || bb >= n || cc < 0 || cc >= n)
                            continue;
                        long long const s = (aa * n + bb) * n + cc;
                        int const code = (da + 1) * 9 + (db + 1) * 3 + (dc + 1);
                        *col++ = (index_type) (ny + nu + s);
                        *val++ = u11(h2(seedA, (std::uint64_t) (s * 27 + (26 - code)))); // A[s][r]
                    }
            jitter(r, col0 + 1, col, ny + nu);
        } else if (r < ny + nu) {
            long long const k = r - ny;
            *col++ = (index_type) r;
            *val++ = 1.5 + 0.5 * u11(h2(seedH, (std::uint64_t) r));
            *col++ = (index_type) (ny + nu + cell_of_control(k));
            *val++ = u11(h2(seedC, (std::uint64_t) k));
        } else {
            long long const s = r - ny - nu, a = s / n2, b = (s / n) % n, c = s % n;
            for (int da = -1; da <= 1; ++da)
                for (int db = -1; db <= 1; ++db)
                    for (int dc = -1; dc <= 1; ++dc) {
                        long long const aa = a + da, bb = b + db, cc = c + dc;
                        if (aa < 0 || aa >= n || bb < 0 || bb >= n || cc < 0 || cc >= n)
                            continue;
                        int const code = (da + 1) * 9 + (db + 1) * 3 + (dc + 1);
                        *col++ = (index_type) ((aa * n + bb) * n + cc);
                        *val++ = u11(h2(seedA, (std::uint64_t) (s * 27 + code))); // A[s][j]
                    }
            jitter(r, col0, col, 0);
            long long const ks[6] = {a == 0 ? 0 * n2 + b * n + c : -1, a == n - 1 ? 1 * n2 + b * n + c : -1,
                                     b == 0 ? 2 * n2 + a * n + c : -1, b == n - 1 ? 3 * n2 + a * n + c : -1,
                                     c == 0 ? 4 * n2 + a * n + b : -1, c == n - 1 ? 5 * n2 + a * n + b : -1};
            for (long long k : ks)
                if (k >= 0) {
                    *col++ = (index_type) (ny + k);
                    *val++ = u11(h2(seedC, (std::uint64_t) k));
                }
        }
    };
    return build(N, N, rb, re, len, fill);
}

// ---- poisson3d: the 7-point Laplacian on an n^3 grid (x fastest), values hashed like poisson2d's pessimistic twin -----
// What the 2-D stand-in of BASELINE configs[1] looks like one dimension up: grid lines of n cells, so that with n = 256 every
// tile of 64 ... 73 rows holds the end of a line (round 5: masked stencil tiles, csr_stenciltile.hpp).
csr_matrix::Matrix poisson3d(long long n, long long rb, long long re, long long * total, long long constant = 0)
{
    if (n < 1 || n > 1290)
        throw matrix::matrix_error("synthetic:poisson3d: grid edge must be in 1..1290");
    long long const N = n * n * n;
    if (total) *total = N;
    if (re < 0) re = N;
    auto len = [n](long long r) {
        long long const x = r % n, y = (r / n) % n, z = r / (n * n);
        return 1 + (x > 0) + (x < n - 1) + (y > 0) + (y < n - 1) + (z > 0) + (z < n - 1);
    };
    auto fill = [n, constant](long long r, index_type * c, double * v) {
        long long const x = r % n, y = (r / n) % n, z = r / (n * n);
        index_type * const c0 = c;
        double * const v0 = v;
        if (z > 0) { *c++ = (index_type) (r - n * n); *v++ = -1.0; }
        if (y > 0) { *c++ = (index_type) (r - n); *v++ = -1.0; }
        if (x > 0) { *c++ = (index_type) (r - 1); *v++ = -1.0; }
        *c++ = (index_type) r; *v++ = 6.0;
        if (x < n - 1) { *c++ = (index_type) (r + 1); *v++ = -1.0; }
        if (y < n - 1) { *c++ = (index_type) (r + n); *v++ = -1.0; }
        if (z < n - 1) { *c++ = (index_type) (r + n * n); *v++ = -1.0; }
        if (!constant) // (constant = 1: the constant-coefficient operator, two distinct values: the plan's value dictionary applies)
            for (long long q = 0; q < v - v0; ++q)
                v0[q] *= 1.0 + 0.25 * u11(h2(0x9015507ull + (std::uint64_t) r, (std::uint64_t) c0[q]));
    };
    return build(N, N, rb, re, len, fill);
}

// ---- queen: 3 unknowns per node of a jittered mesh, symmetric structure, dense 3x3 blocks ------------
// The less tidy twins (round 5; what real finite-element files do): `broken` per mille of the off-diagonal 3 x 3 blocks have one
// or two of their nine entries missing (explicit zeros the assembly dropped), and with `odd_every` = K > 0 every K-th node has
// only one or two unknowns (a constraint node, a pressure node), which moves the grid of row and column triples behind it.
// `dof` (7th number of the spec; default 3): unknowns per node -- 2 or 4 give a mesh without 3 x 3 blocks whose rows still come in groups
// with the same columns (the group tiles of csrc/csr_blocktile.hpp); broken blocks and odd nodes are defined for 3 only.
csr_matrix::Matrix queen(long long gx, long long gy, long long gz, long long rb, long long re, long long * total, long long jitter_nodes = 3,
                         long long broken = 0, long long odd_every = 0, long long dof = 3)
{
    if (gx < 1 || gy < 1 || gz < 1 || gx > 100000 || gy > 100000 || gz > 100000 || gx * gy > 700000000LL / gz)
        throw matrix::matrix_error("synthetic:queen: bad mesh dimensions");
    if (jitter_nodes < 3 || jitter_nodes > 4096)
        throw matrix::matrix_error("synthetic:queen:gx,gy,gz,<J>: a jittered link ends J nodes away, 3 <= J <= 4096");
    if (broken < 0 || broken > 1000 || odd_every < 0 || odd_every == 1)
        throw matrix::matrix_error("synthetic:queen:gx,gy,gz,J,<broken per mille>,<odd node every K>: 0 <= broken <= 1000, K = 0 or K >= 2");
    if (dof < 1 || dof > 8 || (dof != 3 && (broken != 0 || odd_every != 0)) || gx * gy > 2000000000LL / gz / dof)
        throw matrix::matrix_error("synthetic:queen:gx,gy,gz,J,broken,odd,<unknowns per node>: 1 .. 8, and broken blocks / odd nodes only with 3");
    long long const nodes = gx * gy * gz;
    int const D = (int) dof;
    std::uint64_t const seedJ = 0x51DE, seedQ = 0x0EE2, seedB = 0xB20CE, seedO = 0x0DD;
    // unknowns per node and the first row (= column) of every node
    auto dofs = [=](long long n) -> int {
        if (odd_every == 0 || n % odd_every != odd_every / 2)
            return D;
        return 1 + (int) (h2(seedO, (std::uint64_t) n) & 1);
    };
    std::vector<long long> first_row;
    if (odd_every > 0) {
        first_row.resize((std::size_t) nodes + 1);
        first_row[0] = 0;
        for (long long n = 0; n < nodes; ++n)
            first_row[(std::size_t) n + 1] = first_row[(std::size_t) n] + dofs(n);
    }
    auto row0 = [&](long long n) { return odd_every > 0 ? first_row[(std::size_t) n] : D * n; };
    long long const N = row0(nodes);
    if (total) *total = N;
    if (re < 0) re = N;
    auto node_of = [&](long long r) {
        if (odd_every == 0)
            return r / D;
        return (long long) (std::upper_bound(first_row.begin(), first_row.end(), r) - first_row.begin()) - 1;
    };
    long long const J = jitter_nodes; // a jittered link's far end is moved by J nodes (3 by default; more = the pessimistic twin)
    // the 13 "forward" neighbour offsets (x fastest); the other 13 are their mirror images
    int fd[13][3];
    long long foff[13];
    {
        int k = 0;
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    long long const off = ((long long) dz * gy + dy) * gx + dx;
                    if (off > 0) {
                        fd[k][0] = dx; fd[k][1] = dy; fd[k][2] = dz;
                        foff[k++] = off;
                    }
                }
    }
    auto inside = [=](long long m, int const * d) {
        long long const x = m % gx + d[0], y = (m / gx) % gy + d[1], z = m / (gx * gy) + d[2];
        return x >= 0 && x < gx && y >= 0 && y < gy && z >= 0 && z < gz;
    };
    // half of the links keep their grid neighbour, the others end J nodes to either side of it.  J = 3
    // keeps the three links of an x-triple (offsets c-1, c, c+1) apart whatever their jitter, so an
    // interior node has 26 distinct neighbours like a node of a hexahedral mesh
    auto jit = [=](long long m, int k) {
        std::uint64_t const h = h2(seedJ, (std::uint64_t) (m * 13 + k));
        return (h & 1) ? 0LL : ((h & 2) ? J : -J);
    };
    // neighbours of node n (itself included), ascending, no duplicates; returns their number (<= 13 + 13*5 + 1)
    auto neighbours = [=](long long n, long long * out) {
        int c = 0;
        out[c++] = n;
        for (int k = 0; k < 13; ++k) {
            if (inside(n, fd[k])) { // the link n starts
                long long const t = n + foff[k] + jit(n, k);
                if (t >= 0 && t < nodes && t != n)
                    out[c++] = t;
            }
            for (long long j = -J; j <= J; j += J) { // links that end at n
                long long const m = n - foff[k] - j;
                if (m >= 0 && m < nodes && m != n && inside(m, fd[k]) && jit(m, k) == j)
                    out[c++] = m;
            }
        }
        std::sort(out, out + c);
        return (int) (std::unique(out, out + c) - out);
    };
    // entries of block (n, m) that are NOT stored: bit 3 a + b; only off-diagonal blocks of two full nodes are ever broken
    auto dropped = [=](long long n, long long m) -> unsigned {
        if (broken == 0 || n == m)
            return 0u;
        std::uint64_t const h = h2(h2(seedB, (std::uint64_t) n), (std::uint64_t) m);
        if ((long long) (h % 1000) >= broken)
            return 0u;
        unsigned mask = 1u << ((h >> 10) % 9);
        if ((h >> 20) & 1)
            mask |= 1u << ((h >> 24) % 9);
        return mask;
    };
    // node graph first (150 MB at full size), then the rows asked for
    std::vector<long long> node_ptr((std::size_t) nodes + 1, 0);
#pragma omp parallel for schedule(static)
    for (long long n = 0; n < nodes; ++n) {
        long long tmp[96];
        node_ptr[(std::size_t) n + 1] = neighbours(n, tmp);
    }
    for (long long n = 0; n < nodes; ++n)
        node_ptr[(std::size_t) n + 1] += node_ptr[(std::size_t) n];
    std::vector<index_type> nbr((std::size_t) node_ptr[(std::size_t) nodes]);
#pragma omp parallel for schedule(static)
    for (long long n = 0; n < nodes; ++n) {
        long long tmp[96];
        int const c = neighbours(n, tmp);
        for (int i = 0; i < c; ++i)
            nbr[(std::size_t) node_ptr[(std::size_t) n] + i] = (index_type) tmp[i];
    }
    bool const plain = broken == 0 && odd_every == 0;
    auto len = [&](long long r) -> long long {
        long long const n = node_of(r);
        if (plain)
            return D * (node_ptr[(std::size_t) n + 1] - node_ptr[(std::size_t) n]);
        int const a = (int) (r - row0(n));
        long long c = 0;
        for (long long q = node_ptr[(std::size_t) n]; q < node_ptr[(std::size_t) n + 1]; ++q) {
            long long const m = nbr[(std::size_t) q];
            unsigned const gone = dropped(n, m) >> (3 * a);
            for (int b = 0; b < dofs(m); ++b)
                c += !((gone >> b) & 1u);
        }
        return c;
    };
    auto fill = [&](long long r, index_type * col, double * val) {
        long long const n = node_of(r);
        int const a = (int) (r - row0(n));
        for (long long q = node_ptr[(std::size_t) n]; q < node_ptr[(std::size_t) n + 1]; ++q) {
            long long const m = nbr[(std::size_t) q];
            std::uint64_t const edge = h2(h2(seedQ, (std::uint64_t) std::min(n, m)), (std::uint64_t) std::max(n, m));
            unsigned const gone = plain ? 0u : dropped(n, m) >> (3 * a);
            int const mb = plain ? D : dofs(m);
            for (int b = 0; b < mb; ++b) {
                if ((gone >> b) & 1u)
                    continue;
                // block(lo, hi)[a][b]; the mirrored block is its transpose, the diagonal block symmetric
                int const ia = n < m ? a : b, ib = n < m ? b : a;
                double v;
                if (n == m) {
                    v = u11(h2(edge, (std::uint64_t) (std::min(a, b) * (D > 3 ? D : 3) + std::max(a, b))));
                    if (a == b)
                        v += 30.0;
                } else {
                    v = u11(h2(edge, (std::uint64_t) (ia * (D > 3 ? D : 3) + ib)));
                }
                *col++ = (index_type) (row0(m) + b);
                *val++ = v;
            }
        }
    };
    return build(N, N, rb, re, len, fill);
}

// ---- webbase / powerlaw ------------------------------------------------------------------------------
long long gcd_ll(long long a, long long b) { return b == 0 ? a : gcd_ll(b, a % b); }

csr_matrix::Matrix webbase(long long N, long long Z, long long maxrow, int locality_pct, int popularity_exp,
                           long long rb, long long re, long long * total)
{
    if (N < 2 || N > INT32_MAX || Z < N || Z > INT32_MAX || maxrow < 1 || maxrow > N || maxrow > 1000000 ||
        locality_pct < 0 || locality_pct > 100 || (double) Z > (double) N * (double) maxrow)
        throw matrix::matrix_error("synthetic:webbase: need 2 <= N, N <= Z <= N*maxrow, 1 <= maxrow <= min(N, 10^6), locality 0..100");
    if (total) *total = N;
    if (re < 0) re = N;
    std::uint64_t const seedL = 0x3EB, seedF = 0xF1C5, seedHost = 0x4057, seedC = 0xC01;
    // row lengths: P(len = k) ~ k^-alpha on 1..maxrow with alpha solved for the mean Z/N
    double const mean = (double) Z / (double) N;
    auto mean_of = [maxrow](double alpha) {
        double s0 = 0.0, s1 = 0.0;
        for (long long k = maxrow; k >= 1; --k) {
            double const w = std::pow((double) k, -alpha);
            s0 += w;
            s1 += w * (double) k;
        }
        return s1 / s0;
    };
    double lo = 0.0, hi = 12.0;
    for (int it = 0; it < 80; ++it) {
        double const mid = 0.5 * (lo + hi);
        (mean_of(mid) > mean ? lo : hi) = mid;
    }
    double const alpha = 0.5 * (lo + hi);
    std::vector<double> cdf((std::size_t) maxrow);
    {
        double s = 0.0;
        for (long long k = 1; k <= maxrow; ++k)
            cdf[(std::size_t) k - 1] = (s += std::pow((double) k, -alpha));
        for (double & c : cdf)
            c /= s;
    }
    std::vector<index_type> lens((std::size_t) N);
#pragma omp parallel for schedule(static)
    for (long long r = 0; r < N; ++r) {
        double const u = u01(h2(seedL, (std::uint64_t) r));
        lens[(std::size_t) r] = (index_type) (std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin()) + 1;
    }
    long long const longest = (long long) (h2(seedL, 0xFFFFFFFFull) % (std::uint64_t) N);
    lens[(std::size_t) longest] = (index_type) maxrow;
    long long sum = 0;
    for (index_type l : lens)
        sum += l;
    // exactly Z entries: move single entries in and out of hashed rows (never the longest row)
    for (std::uint64_t it = 0; sum != Z; ++it) {
        // hashed rows first; should they all be full (Z close to N * maxrow), walk the rows in order
        long long const t = it < 64ull * (std::uint64_t) N ? (long long) (h2(seedF, it) % (std::uint64_t) N)
                                                          : (long long) (it % (std::uint64_t) N);
        if (t == longest)
            continue;
        if (sum < Z && lens[(std::size_t) t] < maxrow) {
            ++lens[(std::size_t) t];
            ++sum;
        } else if (sum > Z && lens[(std::size_t) t] > 1) {
            --lens[(std::size_t) t];
            --sum;
        }
    }
    // host blocks: many small, a few of thousands of pages
    std::vector<index_type> hstart((std::size_t) N), hsize((std::size_t) N);
    for (long long s = 0, hidx = 0; s < N; ++hidx) {
        double const u = std::max(u01(h2(seedHost, (std::uint64_t) hidx)), 1e-12);
        long long size = (long long) (8.0 * std::pow(u, -0.8));
        size = std::max(1LL, std::min({size, 50000LL, N - s}));
        for (long long r = s; r < s + size; ++r) {
            hstart[(std::size_t) r] = (index_type) s;
            hsize[(std::size_t) r] = (index_type) size;
        }
        s += size;
    }
    long long P = 611953 % N;
    while (P < 2 || gcd_ll(P, N) != 1)
        ++P;
    auto len = [&](long long r) { return (long long) lens[(std::size_t) r]; };
    auto fill = [&](long long r, index_type * col, double * val) {
        long long const L = lens[(std::size_t) r], S = hsize[(std::size_t) r], s0 = hstart[(std::size_t) r];
        std::uint64_t const hr = h2(seedC, (std::uint64_t) r);
        long long nl = (long long) ((double) L * locality_pct / 100.0 + u01(h2(hr, 1)));
        nl = std::min({nl, S, L});
        if (nl > 0) { // distinct pages of the row's own host: an arithmetic progression modulo its size
            long long const a = (long long) (h2(hr, 2) % (std::uint64_t) S);
            long long g = 1 + (long long) (h2(hr, 3) % (std::uint64_t) S);
            while (gcd_ll(g, S) != 1)
                g = g % S + 1;
            for (long long i = 0; i < nl; ++i)
                col[i] = (index_type) (s0 + (a + i * g) % S);
        }
        for (long long i = nl; i < L; ++i) { // the rest: pages by popularity rank, ranks scattered over the index space
            double u = u01(h2(hr, (std::uint64_t) (16 + i)));
            double w = u;
            for (int e = 1; e < popularity_exp; ++e)
                w *= u;
            long long const rank = std::min(N - 1, (long long) (w * (double) N));
            col[i] = (index_type) (popularity_exp > 1 ? (rank * P + 12345) % N : rank);
        }
        std::sort(col, col + L);
        for (long long i = 1; i < L; ++i) // duplicates move up ...
            if (col[i] <= col[i - 1])
                col[i] = col[i - 1] + 1;
        if (col[L - 1] > (index_type) (N - 1)) { // ... and back down where they ran off the end
            col[L - 1] = (index_type) (N - 1);
            for (long long i = L - 2; i >= 0 && col[i] >= col[i + 1]; --i)
                col[i] = col[i + 1] - 1;
        }
        for (long long i = 0; i < L; ++i)
            val[i] = u11(h2(hr, (std::uint64_t) (1u << 20) + (std::uint64_t) i));
    };
    return build(N, N, rb, re, len, fill);
}

// ---- banded: SURVEY 8d "S-banded(N, b, seed)": N rows, the 2b + 1 diagonals -b .. +b, values U(-1, 1) ----------------
csr_matrix::Matrix banded(long long N, long long b, std::uint64_t seed, long long rb, long long re, long long * total)
{
    if (N < 1 || b < 0 || 2 * b + 1 > 32768)
        throw matrix::matrix_error("synthetic:banded:<N>,<b>[,seed]: N >= 1, 0 <= b <= 16383");
    if (total) *total = N;
    if (re < 0) re = N;
    auto len = [=](long long r) { return std::min(N - 1, r + b) - std::max(0LL, r - b) + 1; };
    auto fill = [=](long long r, index_type * c, double * v) {
        long long const lo = std::max(0LL, r - b), hi = std::min(N - 1, r + b);
        for (long long q = lo; q <= hi; ++q) {
            *c++ = (index_type) q;
            *v++ = u11(h2(seed * 0x9E3779B97F4A7C15ull + (std::uint64_t) r, (std::uint64_t) q));
        }
    };
    return build(N, N, rb, re, len, fill);
}

// ---- scrambled: banded(N, b, seed) with rows and columns renumbered by a pseudo-random permutation: P B P^T ----------------
// What a mesh matrix looks like whose nodes were numbered in no particular order: every row still has its 2b + 1 entries, but
// they are scattered over the whole index space and no two rows are shifted copies of each other -- the input the reordering
// suffixes (file__RCM, file__GP<n>) exist for.  The permutation is a four-round Feistel network on the smallest even number of
// bits that holds N, walked until it lands below N (a bijection of [0, N) with a cheap inverse; an affine map r -> r P + c
// would keep consecutive rows shifted copies of each other, which the kernels exploit).
csr_matrix::Matrix scrambled(long long N, long long b, std::uint64_t seed, long long rb, long long re, long long * total)
{
    if (N < 1 || N > INT32_MAX || b < 0 || 2 * b + 1 > 4096)
        throw matrix::matrix_error("synthetic:scrambled:<N>,<b>[,seed]: 1 <= N <= 2^31-1, 0 <= b <= 2047");
    if (total) *total = N;
    if (re < 0) re = N;
    int half = 1;
    while ((1LL << (2 * half)) < N)
        ++half;
    std::uint64_t const mask = (1ull << half) - 1, key = seed * 0xD6E8FEB86659FD93ull + 0x5C4A;
    auto round_fn = [=](std::uint64_t v, int k) { return h2(key + (std::uint64_t) k, v) & mask; };
    auto feistel = [=](std::uint64_t v, bool inverse) {
        std::uint64_t l = v >> half, r = v & mask;
        if (!inverse)
            for (int k = 0; k < 4; ++k) {
                std::uint64_t const t = l ^ round_fn(r, k);
                l = r;
                r = t;
            }
        else
            for (int k = 3; k >= 0; --k) {
                std::uint64_t const t = r ^ round_fn(l, k);
                r = l;
                l = t;
            }
        return (l << half) | r;
    };
    auto fwd = [=](long long r) {
        std::uint64_t v = (std::uint64_t) r;
        do
            v = feistel(v, false);
        while (v >= (std::uint64_t) N);
        return (long long) v;
    };
    auto back = [=](long long q) {
        std::uint64_t v = (std::uint64_t) q;
        do
            v = feistel(v, true);
        while (v >= (std::uint64_t) N);
        return (long long) v;
    };
    auto len = [=](long long q) { long long const r = back(q); return std::min(N - 1, r + b) - std::max(0LL, r - b) + 1; };
    auto fill = [=](long long q, index_type * c, double * v) {
        long long const r = back(q), lo = std::max(0LL, r - b), hi = std::min(N - 1, r + b);
        std::pair<index_type, double> tmp[4096];
        int n = 0;
        for (long long o = lo; o <= hi; ++o)
            tmp[n++] = {(index_type) fwd(o), u11(h2(seed * 0x9E3779B97F4A7C15ull + (std::uint64_t) r, (std::uint64_t) o))};
        std::sort(tmp, tmp + n, [](auto const & x, auto const & y) { return x.first < y.first; });
        for (int i = 0; i < n; ++i) {
            c[i] = tmp[i].first;
            v[i] = tmp[i].second;
        }
    };
    return build(N, N, rb, re, len, fill);
}

// ---- random: SURVEY 8d "S-random(N, k, seed)": k distinct uniform columns per row, ascending, values U(-1, 1) --------
csr_matrix::Matrix random_columns(long long N, long long k, std::uint64_t seed, long long rb, long long re, long long * total)
{
    if (N < 1 || k < 1 || k > 1024 || k > N)
        throw matrix::matrix_error("synthetic:random:<N>,<k>[,seed]: N >= 1, 1 <= k <= min(N, 1024)");
    if (total) *total = N;
    if (re < 0) re = N;
    auto len = [=](long long) { return k; };
    auto fill = [=](long long r, index_type * c, double * v) {
        // k distinct columns: draw, sort, and push duplicates up to the next free column (wrapping keeps them in range)
        std::uint64_t const hr = seed * 0xD1B54A32D192ED03ull + (std::uint64_t) r;
        for (long long i = 0; i < k; ++i)
            c[i] = (index_type) (h2(hr, (std::uint64_t) i) % (std::uint64_t) N);
        std::sort(c, c + k);
        for (long long i = 1; i < k; ++i)
            if (c[i] <= c[i - 1])
                c[i] = c[i - 1] + 1;
        if (c[k - 1] >= N) { // ran over the edge: the last columns of the matrix, still distinct and ascending
            for (long long i = k - 1, q = N - 1; i >= 0 && c[i] > q; --i, --q)
                c[i] = (index_type) q;
        }
        for (long long i = 0; i < k; ++i)
            v[i] = u11(h2(hr, (std::uint64_t) (1u << 20) + (std::uint64_t) i));
    };
    return build(N, N, rb, re, len, fill);
}

// ---- ":tril": what a Matrix Market file with a `symmetric` header holds of a structurally symmetric matrix ----------------
// The reference stores exactly the entries of the file and mirrors nothing (src/matrix/matrix-market.cpp:396-414 parses the
// token, :530-555 keeps the entries as read; README.md:106: 1138_bus has "nonzeros": 2596, its stored triangle), so what it
// multiplies when handed Queen_4147.tar.gz / nlpkkt200.tar.gz is the LOWER TRIANGLE, diagonal included: rows whose length
// grows with the number of neighbours numbered before them, triangular diagonal blocks, half stencils.  Keeps the entries with
// column <= row of the rows [rb, rb + A.rows) of a generated matrix; two passes, every row on its own.
csr_matrix::Matrix lower_triangle(csr_matrix::Matrix const & A, long long rb)
{
    long long const nr = A.rows;
    csr_matrix::size_array_type row_ptr((std::size_t) nr + 1, 0);
#pragma omp parallel for schedule(static)
    for (long long r = 0; r < nr; ++r) {
        index_type const * const c = A.column_index.data();
        size_type const k0 = A.row_ptr[(std::size_t) r], k1 = A.row_ptr[(std::size_t) r + 1];
        // columns ascend inside a row: the stored part is a prefix
        row_ptr[(std::size_t) r + 1] = (size_type) (std::upper_bound(c + k0, c + k1, (index_type) (rb + r)) - (c + k0));
    }
    for (long long r = 0; r < nr; ++r)
        row_ptr[(std::size_t) r + 1] += row_ptr[(std::size_t) r];
    std::size_t const Z = (std::size_t) row_ptr[(std::size_t) nr];
    csr_matrix::index_array_type col(Z);
    csr_matrix::value_array_type val(Z);
#pragma omp parallel for schedule(dynamic, 4096)
    for (long long r = 0; r < nr; ++r) {
        size_type const k0 = A.row_ptr[(std::size_t) r], d0 = row_ptr[(std::size_t) r], n = row_ptr[(std::size_t) r + 1] - d0;
        std::copy_n(A.column_index.data() + k0, n, col.data() + d0);
        std::copy_n(A.value.data() + k0, n, val.data() + d0);
    }
    return csr_matrix::Matrix((index_type) nr, A.columns, (size_type) Z, 1, std::move(row_ptr), std::move(col), std::move(val));
}

} // namespace

bool is_spec(std::string const & path) { return path.compare(0, 10, "synthetic:") == 0; }

bool is_stored_triangle(std::string const & spec)
{
    return is_spec(spec) && spec.size() >= 15 && spec.compare(spec.size() - 5, 5, ":tril") == 0;
}

csr_matrix::Matrix generate_csr(std::string const & spec, long long rb, long long re, long long * total)
{
    if (!is_spec(spec))
        throw matrix::matrix_error("not a synthetic matrix specification: " + spec);
    std::string rest = spec.substr(10);
    bool const tril = is_stored_triangle(spec);
    if (tril)
        rest.resize(rest.size() - 5);
    std::string family = rest, params;
    std::size_t const colon = rest.find(':');
    if (colon != std::string::npos) {
        family = rest.substr(0, colon);
        params = rest.substr(colon + 1);
    }
    std::vector<long long> const v = numbers(params);
    long long tot = 0;
    auto check_range = [&](long long N) {
        if (rb < 0 || (re >= 0 && (re < rb || re > N)) || rb > N)
            throw matrix::matrix_error("synthetic matrix: row range out of bounds");
    };
    csr_matrix::Matrix A;
    // every family's parameters are bounded BEFORE a row count is formed from them (a product of two numbers
    // parsed from the command line overflows a signed 64-bit integer long before the generator refuses them)
    if (family == "poisson2d") {
        if (v.size() < 1 || v.size() > 2)
            throw matrix::matrix_error("synthetic:poisson2d:<n>[,<hashed values 0|1>] takes one or two numbers");
        if (v[0] < 1 || v[0] > 46340)
            throw matrix::matrix_error("synthetic:poisson2d: grid edge must be in 1..46340");
        check_range(v[0] * v[0]);
        A = poisson2d(v[0], rb, re, &tot, v.size() > 1 ? v[1] : 0);
    } else if (family == "poisson3d") {
        if (v.size() < 1 || v.size() > 2)
            throw matrix::matrix_error("synthetic:poisson3d:<n>[,<constant coefficients 0|1>] takes one or two numbers");
        if (v[0] < 1 || v[0] > 1290)
            throw matrix::matrix_error("synthetic:poisson3d: grid edge must be in 1..1290");
        check_range(v[0] * v[0] * v[0]);
        A = poisson3d(v[0], rb, re, &tot, v.size() > 1 ? v[1] : 0);
    } else if (family == "kkt") {
        if (v.size() > 2)
            throw matrix::matrix_error("synthetic:kkt[:<n>[,<jitter %>]] takes at most two numbers");
        long long const n = v.empty() ? 200 : v[0];
        if (n < 1 || n > 1000)
            throw matrix::matrix_error("synthetic:kkt: grid edge must be in 1..1000");
        check_range(2 * n * n * n + 6 * n * n);
        A = kkt(n, rb, re, &tot, v.size() > 1 ? v[1] : 0);
    } else if (family == "queen") {
        if (!v.empty() && (v.size() < 3 || v.size() > 7))
            throw matrix::matrix_error("synthetic:queen[:gx,gy,gz[,J[,broken per mille[,odd node every K[,unknowns per node]]]]] takes three to seven numbers");
        long long const gx = v.empty() ? 110 : v[0], gy = v.empty() ? 71 : v[1], gz = v.empty() ? 177 : v[2];
        if (gx < 1 || gy < 1 || gz < 1 || gx > 100000 || gy > 100000 || gz > 100000 || gx * gy > 700000000LL / gz)
            throw matrix::matrix_error("synthetic:queen: bad mesh dimensions");
        check_range((v.size() > 6 ? std::max(1LL, std::min(8LL, v[6])) : 3) * gx * gy * gz);
        A = queen(gx, gy, gz, rb, re, &tot, v.size() > 3 ? v[3] : 3, v.size() > 4 ? v[4] : 0, v.size() > 5 ? v[5] : 0, v.size() > 6 ? v[6] : 3);
    } else if (family == "webbase" || family == "powerlaw") {
        bool const web = family == "webbase";
        if (v.size() > (web ? 4u : 3u))
            throw matrix::matrix_error("synthetic:" + family + ": too many parameters");
        long long const N = v.size() > 0 ? v[0] : 1000005;
        if (N < 2 || N > INT32_MAX)
            throw matrix::matrix_error("synthetic:" + family + ": need 2 <= N <= 2^31-1");
        long long const Z = v.size() > 1 ? v[1] : (v.empty() ? 3105536 : 3 * N);
        long long const maxrow = v.size() > 2 ? v[2] : std::min(4700LL, N);
        int const loc = web ? (v.size() > 3 ? (int) v[3] : 75) : 0;
        check_range(N);
        A = webbase(N, Z, maxrow, loc, web ? 4 : 1, rb, re, &tot);
    } else if (family == "banded" || family == "random" || family == "scrambled") {
        if (v.size() < 2 || v.size() > 3)
            throw matrix::matrix_error("synthetic:" + family + ":<N>,<" + (family == "random" ? "k" : "b") + ">[,seed]");
        check_range(v[0]);
        std::uint64_t const seed = v.size() > 2 ? (std::uint64_t) v[2] : 1;
        A = family == "banded" ? banded(v[0], v[1], seed, rb, re, &tot)
            : family == "scrambled" ? scrambled(v[0], v[1], seed, rb, re, &tot) : random_columns(v[0], v[1], seed, rb, re, &tot);
    } else {
        throw matrix::matrix_error("unknown synthetic matrix family '" + family +
                                   "' (poisson2d, poisson3d, queen, kkt, webbase, powerlaw, banded, scrambled, random)");
    }
    if (total)
        *total = tot;
    if (tril)
        return lower_triangle(A, rb);
    return A;
}

matrix_market::Matrix generate(std::string const & spec)
{
    csr_matrix::Matrix const A = generate_csr(spec);
    std::size_t const Z = (std::size_t) A.row_ptr[(std::size_t) A.rows];
    std::vector<matrix_market::index_type> i(Z), j(Z);
    std::vector<matrix_market::real_type> a(Z);
#pragma omp parallel for schedule(static)
    for (long long r = 0; r < (long long) A.rows; ++r)
        for (size_type k = A.row_ptr[(std::size_t) r]; k < A.row_ptr[(std::size_t) r + 1]; ++k) {
            i[(std::size_t) k] = (matrix_market::index_type) r + 1;
            j[(std::size_t) k] = A.column_index[(std::size_t) k] + 1;
            a[(std::size_t) k] = A.value[(std::size_t) k];
        }
    matrix_market::Header h;
    if (is_stored_triangle(spec)) // the header such a file carries (parsed and otherwise ignored, as in the reference)
        h.symmetry = matrix_market::Symmetry::symmetric;
    matrix_market::Size s;
    s.rows = A.rows;
    s.columns = A.columns;
    s.num_entries = (matrix_market::size_type) Z;
    return matrix_market::Matrix(h, {"% generated: " + spec}, s, std::move(i), std::move(j), std::move(a));
}

} // namespace synthetic
