# host/host.mk -- C++ host side: Matrix Market loader, formats, Kernel adapters, timed loop, CLI.
# Included by ../Makefile (paths are relative to spmv-cache-trace_amd/).
#
#   libspmv_host.so         everything but main(): loader, formats, kernels, timed loop, and the
#                           C ABI of include/spmv_host.h (host-api.cpp)
#   libspmv_host_test.so    test-hooks.cpp only: entry points for tests/ (not shipped with the product)
#   spmv-cache-trace-hip    the CLI; finds libspmv_hip.so / libspmv_host.so next to itself
#
# -ffp-contract=off keeps the CPU kernels' arithmetic identical to the reference build
# (plain -O3 on x86-64: multiply, then add).

HOST_CXXFLAGS := -std=c++17 -O3 -fopenmp -fPIC -ffp-contract=off -Wall -Wextra -Wno-unused-parameter \
                 -D__HIP_PLATFORM_AMD__ $(INC) -I$(ROCM)/include -Ihost
HOST_SRCS := host/util/json-value.cpp host/util/cpu-budget.cpp host/trace-config.cpp host/matrix/matrix-market.cpp host/matrix/matrix-cache.cpp \
             host/matrix/csr-matrix.cpp host/matrix/coo-matrix.cpp host/matrix/ell-matrix.cpp \
             host/matrix/hybrid-matrix.cpp host/matrix/matrix-reorder.cpp host/matrix/synthetic.cpp \
             host/kernels/spmv-kernels.cpp host/kernels/triad-kernel.cpp host/profile-kernel.cpp \
             host/host-api.cpp
HOST_OBJS := $(HOST_SRCS:.cpp=.o)
HOST_HDRS := $(wildcard host/*.hpp host/*/*.hpp) $(wildcard $(ROOT)/include/spmv_hip*.h)
HOST_LIB  := libspmv_host.so
CLI       := spmv-cache-trace-hip

HOST_TEST_LIB := libspmv_host_test.so

host: $(HOST_LIB) $(HOST_TEST_LIB) $(CLI)

$(HOST_TEST_LIB): host/test-hooks.o $(HOST_LIB)
	$(CXX) -shared -fopenmp host/test-hooks.o -o $@ -L. -lspmv_host -Wl,-rpath,'$$ORIGIN'

host/%.o: host/%.cpp $(HOST_HDRS)
	$(CXX) $(HOST_CXXFLAGS) -c $< -o $@

$(HOST_LIB): $(HOST_OBJS) $(LIB)
	$(CXX) -shared -fopenmp $(HOST_OBJS) -o $@ -L. -lspmv_hip -L$(ROCM)/lib -lamdhip64 -lz \
		-Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(ROCM)/lib

$(CLI): host/main.o $(HOST_LIB)
	$(CXX) -fopenmp host/main.o -o $@ -L. -lspmv_host -lspmv_hip -L$(ROCM)/lib -lamdhip64 \
		-Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(ROCM)/lib
