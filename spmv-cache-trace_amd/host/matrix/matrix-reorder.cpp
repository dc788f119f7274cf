#include "matrix-reorder.hpp"

#include "matrix-error.hpp"

#include <algorithm>
#include <cstdlib>
#include <numeric>
#include <ostream>
#include <queue>

using matrix::matrix_error;

namespace matrix_market {

int bandwidth(Matrix const & m)
{
    auto const & i = m.row_indices();
    auto const & j = m.column_indices();
    int b = 0;
    for (std::size_t k = 0; k < i.size(); ++k)
        b = std::max(b, std::abs(i[k] - j[k]));
    return b;
}

namespace {

void require_square_real(Matrix const & m)
{
    if (m.format() != Format::coordinate)
        throw matrix_error("Expected matrix in coordinate format");
    if (m.rows() != m.columns())
        throw matrix_error("Expected a square matrix");
    if (m.field() != Field::real)
        throw matrix_error("Expected matrix with real values");
}

} // namespace

std::vector<int> find_new_order_RCM(Matrix const & m, std::ostream & log, bool verbose)
{
    require_square_real(m);
    int const n = m.rows();
    auto const & ri = m.row_indices();
    auto const & ci = m.column_indices();
    for (std::size_t k = 0; k < ri.size(); ++k)
        if (ri[k] < 1 || ri[k] > n || ci[k] < 1 || ci[k] > n)
            throw matrix_error("Row or column index out of bounds");

    // out-neighbours of every node in file order, self-loops dropped; for a symmetric file that
    // stores one triangle this is a directed graph, exactly as in the reference
    std::vector<int> degree((std::size_t) n, 0);
    for (std::size_t k = 0; k < ri.size(); ++k)
        if (ri[k] != ci[k])
            ++degree[(std::size_t) ri[k] - 1];
    std::vector<std::size_t> first((std::size_t) n + 1, 0);
    for (int v = 0; v < n; ++v)
        first[(std::size_t) v + 1] = first[(std::size_t) v] + (std::size_t) degree[(std::size_t) v];
    std::vector<int> adjacency(first[(std::size_t) n]);
    {
        std::vector<std::size_t> fill(first.begin(), first.end() - 1);
        for (std::size_t k = 0; k < ri.size(); ++k)
            if (ri[k] != ci[k])
                adjacency[fill[(std::size_t) ri[k] - 1]++] = ci[k] - 1;
    }
    if (verbose)
        log << "Bandwidth of the matrix is " << bandwidth(m) << '\n'
            << "Running the reverse Cuthill-McKee algorithm to find another permutation\n";

    // start nodes: the not-yet-taken node of smallest degree, lowest index first (the reference
    // rescans all nodes per component; walking a (degree, index)-sorted list picks the same node)
    std::vector<int> by_degree((std::size_t) n);
    std::iota(by_degree.begin(), by_degree.end(), 0);
    std::stable_sort(by_degree.begin(), by_degree.end(),
                     [&](int a, int b) { return degree[(std::size_t) a] < degree[(std::size_t) b]; });
    std::size_t next_start = 0;

    std::vector<int> order; // Cuthill-McKee order
    order.reserve((std::size_t) n);
    std::vector<char> taken((std::size_t) n, 0), visited((std::size_t) n, 0);
    std::vector<int> batch;
    auto const by_deg = [&degree](int a, int b) { return degree[(std::size_t) a] < degree[(std::size_t) b]; };

    // unvisited out-neighbours of v, ordered by degree with std::sort like the reference (ties
    // land where this libstdc++'s introsort puts them, which is what the reference gets as well)
    auto const collect = [&](int v) {
        batch.clear();
        for (std::size_t q = first[(std::size_t) v]; q < first[(std::size_t) v + 1]; ++q) {
            int const w = adjacency[q];
            if (!visited[(std::size_t) w]) {
                batch.push_back(w);
                visited[(std::size_t) w] = 1;
            }
        }
        if (batch.size() > 1)
            std::sort(batch.begin(), batch.end(), by_deg);
    };

    while ((int) order.size() < n) {
        while (taken[(std::size_t) by_degree[next_start]])
            ++next_start;
        int const root = by_degree[next_start];
        order.push_back(root);
        taken[(std::size_t) root] = 1;
        visited[(std::size_t) root] = 1;
        std::queue<int> frontier;
        collect(root);
        for (int w : batch)
            frontier.push(w);
        while (!frontier.empty()) {
            int const v = frontier.front();
            frontier.pop();
            if (taken[(std::size_t) v])
                continue;
            order.push_back(v);
            taken[(std::size_t) v] = 1;
            collect(v);
            for (int w : batch)
                frontier.push(w);
        }
    }
    std::reverse(order.begin(), order.end());

    std::vector<int> new_order((std::size_t) n, -1);
    for (int k = 0; k < n; ++k)
        new_order[(std::size_t) order[(std::size_t) k]] = k;
    if (verbose) {
        int b = 0;
        for (std::size_t k = 0; k < ri.size(); ++k)
            b = std::max(b, std::abs(new_order[(std::size_t) ri[k] - 1] - new_order[(std::size_t) ci[k] - 1]));
        log << "New bandwidth of the matrix is " << b << '\n';
    }
    return new_order;
}

std::vector<int> find_new_order_GP(Matrix const & m, int, std::ostream & log, bool)
{
    require_square_real(m);
    // same outcome as the reference built without USE_METIS (matrix-market-reorder.cpp:172-181)
    log << "Warning: No reordering is done. Graph partitioning needs METIS, which this build does not have\n";
    std::vector<int> same((std::size_t) m.rows());
    std::iota(same.begin(), same.end(), 0);
    return same;
}

Matrix permute(Matrix const & m, std::vector<int> const & new_order)
{
    require_square_real(m);
    if ((int) new_order.size() != m.rows())
        throw matrix_error("The dimension of the matrix doesn't match the permutation");
    std::vector<index_type> i(m.row_indices()), j(m.column_indices());
    for (std::size_t k = 0; k < i.size(); ++k) {
        i[k] = new_order[(std::size_t) i[k] - 1] + 1;
        j[k] = new_order[(std::size_t) j[k] - 1] + 1;
    }
    return Matrix(m.header(), m.comments(), m.size(), std::move(i), std::move(j), m.values_real());
}

} // namespace matrix_market
