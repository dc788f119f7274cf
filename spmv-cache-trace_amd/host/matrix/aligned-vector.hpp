// aligned-vector.hpp -- page-aligned std::vector storage for the host copies of A, x, y.
//
// The reference keeps its arrays in 4096-byte aligned vectors
// (src/util/aligned-allocator.hpp:65-87); the C ABI receives `.data()` of these, and page
// alignment also makes them cheap to pin / DMA.  NUMA page placement (distribute_pages) is a
// host-only concern of the reference's cache study and is not reproduced.
#pragma once

#include <cstddef>
#include <cstdlib>
#include <new>
#include <system_error>
#include <vector>

template <typename T, std::size_t Alignment = 4096>
class aligned_allocator
{
public:
    using value_type = T;
    template <class U> struct rebind { using other = aligned_allocator<U, Alignment>; };

    aligned_allocator() noexcept = default;
    template <class U> aligned_allocator(aligned_allocator<U, Alignment> const &) noexcept {}

    T * allocate(std::size_t n)
    {
        void * p = nullptr;
        std::size_t bytes = n * sizeof(T);
        if (bytes == 0)
            bytes = Alignment;
        int rc = posix_memalign(&p, Alignment, bytes);
        if (rc != 0)
            throw std::system_error(rc, std::generic_category());
        return static_cast<T *>(p);
    }
    void deallocate(T * p, std::size_t) noexcept { std::free(p); }

    template <class U> bool operator==(aligned_allocator<U, Alignment> const &) const noexcept { return true; }
    template <class U> bool operator!=(aligned_allocator<U, Alignment> const &) const noexcept { return false; }
};

template <typename T> using aligned_vector = std::vector<T, aligned_allocator<T, 4096>>;
