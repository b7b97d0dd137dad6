# Builds the product library (HIP, gfx950) and the test-only CPU oracle.
#   make            -> raft_amd/librq.so  oracle/librq_oracle.so
HIPCC   ?= hipcc
CC      ?= gcc
ARCH    ?= gfx950
# -amdgpu-kernarg-preload-count: the leading scalar arguments of a kernel (up to 14 dwords on
# gfx950) arrive in SGPRs with the wave instead of being fetched by its first scalar loads;
# rq_round_fixed orders its arguments for it (rq_round_impl.h).
# PLAIN=1 also builds the round kernels with the default cache policy (RQ_VARIANT_PLAIN;
# a library without them runs the streaming builds for that variant, results are the same)
HIPFLAGS = -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function $(if $(PLAIN),-DRQ_BUILD_PLAIN) -mllvm -amdgpu-kernarg-preload-count=16
CSRC     = raft_amd/csrc

all: raft_amd/librq.so raft_amd/libraft_host.so oracle/librq_oracle.so examples/rq_cli examples/rq_bench

build/%.o: $(CSRC)/%.hip $(CSRC)/rq_kernels.h $(CSRC)/rq_round_impl.h $(CSRC)/rq_append_impl.h $(CSRC)/rq_groups_impl.h include/rq.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

raft_amd/librq.so: build/rq_kernels.o build/rq_kernels64.o build/rq_engine.o build/rq_comm.o build/rq_noalloc.o
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $^ -ldl -o $@

HOST = $(CSRC)/host
raft_amd/libraft_host.so: $(HOST)/raft_host.cpp $(HOST)/raft_host_resident.cpp $(HOST)/raft_host_capi.cpp $(HOST)/raft_host.hpp include/rq.h raft_amd/librq.so
	g++ -O2 -std=c++17 -fPIC -shared -Wall -Wextra -pthread $(HOST)/raft_host.cpp $(HOST)/raft_host_resident.cpp $(HOST)/raft_host_capi.cpp \
	    -Lraft_amd -lrq -Wl,-rpath,'$$ORIGIN' -o $@

oracle/librq_oracle.so: oracle/raft_oracle.c oracle/raft_oracle.h
	$(CC) -O2 -std=c11 -fPIC -shared -Wall -Wextra -pthread $< -o $@

# the C-ABI from plain C (no HIP headers, no torch)
examples/rq_%: examples/rq_%.c include/rq.h raft_amd/librq.so
	$(CC) -O2 -std=c11 -Wall -Wextra -Iinclude $< -Lraft_amd -lrq -Wl,-rpath,'$$ORIGIN/../raft_amd' -o $@

clean:
	rm -rf build raft_amd/librq.so raft_amd/libraft_host.so oracle/librq_oracle.so examples/rq_cli examples/rq_bench

.PHONY: all clean
