# Host-side sanitizer build of libvagnmt (CPU box only; listed in .gpurunignore: it never travels to a GPU box).
#   make -C vag-nmt_amd/csrc -f asan.mk          ->  ../lib/libvagnmt_asan.so
# The HOST halves of every translation unit (argument checks, workspace layouts, thread-local request queues, group planning, the
# prezeroed set, defer lists) under AddressSanitizer + UBSan: `-Xarch_host` hands the sanitizer flags to the host pass only, the device
# code is the product's.  tests/test_abi.py::test_host_side_under_sanitizers loads it in a child process with the sanitizer runtime
# preloaded.  Not a GPU sanitizer: those are unavailable on this pool.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CXXFLAGS = -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function
SAN = address,undefined
ASAN_HOST = -Xarch_host -fsanitize=$(SAN) -Xarch_host -fno-omit-frame-pointer -Xarch_host -fno-sanitize-recover=undefined
SRCS = gemm.hip elem.hip attn.hip head.hip vse.hip optim.hip beam.hip api.hip step.hip persist.hip comm.hip
ASAN_OBJS = $(SRCS:%.hip=build_asan/%.o)

asan: ../lib/libvagnmt_asan.so

build_asan/%.o: %.hip common.h kernels.h gemm_shared.h ../../include/vag_nmt.h
	@mkdir -p build_asan
	$(HIPCC) $(CXXFLAGS) $(ASAN_HOST) -c $< -o $@

../lib/libvagnmt_asan.so: $(ASAN_OBJS)
	@mkdir -p ../lib
	$(HIPCC) --offload-arch=$(ARCH) -fsanitize=$(SAN) -shared-libsan -shared -fPIC -o $@ $(ASAN_OBJS) -ldl

.PHONY: asan
