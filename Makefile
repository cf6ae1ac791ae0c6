# Top-level build: the gfx950 library, the test oracle, the CPU single-stepper.
#   make lib     -> hevcbitstream_amd/libhevcbitstream_amd.so   (hipcc, gfx950 only)
#   make oracle  -> oracle/liboracle.so (+ oracle/_ref/* when /root/reference exists)
#   make sim     -> tests/sim/libhbs_sim.so
HIPCC ?= hipcc
ARCH  ?= gfx950
CSRC  := hevcbitstream_amd/csrc
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -Wall -Wno-unused-function
LIB   := hevcbitstream_amd/libhevcbitstream_amd.so
HIP_SRCS := $(wildcard $(CSRC)/*.hip)
HDRS  := $(wildcard $(CSRC)/*.h) $(wildcard include/*.h)

all: lib oracle sim

lib: $(LIB)

$(LIB): $(HIP_SRCS) $(HDRS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_SRCS)

oracle:
	$(MAKE) -C oracle all

sim:
	$(MAKE) -C tests/sim

clean:
	rm -f $(LIB) tests/sim/libhbs_sim.so
	$(MAKE) -C oracle clean

.PHONY: all lib oracle sim clean
