// matrix-error.hpp -- the exception type of the matrix layer (reference src/matrix/matrix-error.hpp).
#pragma once

#include <stdexcept>
#include <string>

namespace matrix {

class matrix_error : public std::runtime_error
{
public:
    explicit matrix_error(std::string const & s) : std::runtime_error(s) {}
};

} // namespace matrix
