# Builds everything in-tree (no install step):
#   tsdf_amd/lib/libtsdf_hip.so   HIP kernels + C ABI (include/tsdf_amd.h), gfx950 only
#   tsdf_amd/lib/libtsdf_host.so  C++ class surface (TSDFVolume, GPURaycaster, Camera, ...) over the C ABI
#   oracle/libtsdf_oracle.so      CPU oracle (test infrastructure), oracle/_ref/*.so: what of the reference compiles from its own sources here (BilateralFilter.cpp, cuda_coordinate_transforms.cu), when the reference is mounted
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
# -ffp-contract=off: every fp32 op rounds on its own, in the reference's order (parity contract)
HIPFLAGS  = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -Itsdf_amd/csrc -Wall -Wno-unused-function
CSRC      = tsdf_amd/csrc
# `make DIAG=1`: a library with the host-side diagnostics of diagnostics.hip (TSDF_DEBUG_SORT / TSDF_DEBUG_BRICKS); not the product build
ifeq ($(DIAG),1)
HIPFLAGS += -DTSDF_DIAGNOSTICS
endif
HIP_SRCS  = $(CSRC)/diagnostics.hip $(CSRC)/volume.hip $(CSRC)/integrate.hip $(CSRC)/integrate_packed.hip $(CSRC)/weights.hip $(CSRC)/raycast.hip $(CSRC)/bilateral.hip $(CSRC)/icp.hip $(CSRC)/mcubes.hip $(CSRC)/pipeline.hip
HIP_OBJS  = $(HIP_SRCS:.hip=.o)
LIBDIR    = tsdf_amd/lib

# Host C++ (class surface).  A real Eigen wins when one is installed; otherwise the bundled
# minimal Eigen-compatible header is used.
CXX      ?= g++
HOSTDIR   = tsdf_amd/host
EIGEN_INC := $(shell for d in /usr/include/eigen3 /usr/local/include/eigen3; do [ -f $$d/Eigen/Core ] && echo -I$$d && break; done)
ifeq ($(EIGEN_INC),)
EIGEN_INC = -I$(HOSTDIR)/eigen_compat
endif
# the same for Sophus (used only by the ICPOdometry interface)
SOPHUS_INC := $(shell for d in /usr/include /usr/local/include; do [ -f $$d/sophus/se3.hpp ] && echo -I$$d && break; done)
ifeq ($(SOPHUS_INC),)
SOPHUS_INC = -I$(HOSTDIR)/sophus_compat
endif
HOSTFLAGS = -std=c++11 -O2 -pthread -ffp-contract=off -fPIC -Wall -Iinclude -I$(HOSTDIR)/include -I$(HOSTDIR)/third_party $(EIGEN_INC) $(SOPHUS_INC)
HOST_SRCS = $(wildcard $(HOSTDIR)/src/*.cpp)
HOST_OBJS = $(HOST_SRCS:.cpp=.o)

all: hip host oracle cpptest

host: $(LIBDIR)/libtsdf_host.so

$(HOSTDIR)/src/%.o: $(HOSTDIR)/src/%.cpp $(wildcard $(HOSTDIR)/include/*.hpp) $(wildcard $(HOSTDIR)/src/*.hpp) $(wildcard $(HOSTDIR)/eigen_compat/Eigen/*) $(wildcard $(HOSTDIR)/sophus_compat/sophus/*) $(wildcard $(HOSTDIR)/third_party/ICP_CUDA/*) include/tsdf_amd.h
	$(CXX) $(HOSTFLAGS) -c $< -o $@

$(LIBDIR)/libtsdf_host.so: $(HOST_OBJS) $(LIBDIR)/libtsdf_hip.so
	$(CXX) -shared -fPIC -pthread -o $@ $(HOST_OBJS) -L$(LIBDIR) -ltsdf_hip -lz -Wl,-rpath,'$$ORIGIN'

hip: $(LIBDIR)/libtsdf_hip.so

# integrate_packed_kernel is bound by its vector instruction stream: the scheduler's max-ILP strategy is worth 1-1.5 % there
# (0.0804 against 0.0816 ms; it costs the latency-bound ray kernels 13 % and the bilateral filter 11 %: those keep the default)
$(CSRC)/integrate_packed.o: HIPFLAGS += -mllvm -amdgpu-sched-strategy=max-ilp

$(CSRC)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.hpp) include/tsdf_amd.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/libtsdf_hip.so: $(HIP_OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(HIP_OBJS) -ldl

oracle:
	$(MAKE) -C oracle -s all

# C++ test program of the class surface (run by tests/test_cpp_surface.py on the GPU box)
cpptest: build/test_surface build/kinfu_stream

# C++ driver of BASELINE configs[2] (TUM directory -> tsdf_pipeline_step, no Python): tools/kinfu_stream.cpp
build/kinfu_stream: tools/kinfu_stream.cpp $(LIBDIR)/libtsdf_host.so include/tsdf_amd.h
	@mkdir -p build
	$(CXX) $(HOSTFLAGS) -o $@ tools/kinfu_stream.cpp -L$(LIBDIR) -ltsdf_host -ltsdf_hip -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)'

build/test_surface: tests/cpp/test_surface.cpp $(LIBDIR)/libtsdf_host.so
	@mkdir -p build
	$(CXX) $(HOSTFLAGS) -o $@ tests/cpp/test_surface.cpp -L$(LIBDIR) -ltsdf_host -ltsdf_hip -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)'

clean:
	rm -f $(HIP_OBJS) $(HOST_OBJS) $(LIBDIR)/*.so
	$(MAKE) -C oracle clean

.PHONY: all hip host oracle cpptest clean
