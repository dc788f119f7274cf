// triad-kernel.hpp -- STREAM triad a = b + 3.1*c behind the Kernel interface
// (reference src/kernels/triad.cpp): on the GPU it calibrates the attainable HBM bandwidth
// that the SpMV roofline fractions are quoted against.
#pragma once

#include "kernel.hpp"

#include <memory>

std::unique_ptr<Kernel> make_triad_kernel(std::size_t num_entries, bool hip, int device);
