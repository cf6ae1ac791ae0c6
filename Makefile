# Top-level build: the gfx950 library, the test oracle, the CPU single-stepper.
#   make lib     -> hevcbitstream_amd/libhevcbitstream_amd.so   (hipcc, gfx950 only)
#   make oracle  -> oracle/liboracle.so (+ oracle/_ref/* when /root/reference exists)
#   make sim     -> tests/sim/libhbs_sim.so
# no built-in rules: `%: %.o` would try to LINK a source file from its object and delete it on failure
MAKEFLAGS += -r
.SUFFIXES:

HIPCC ?= hipcc
ARCH  ?= gfx950
CSRC  := hevcbitstream_amd/csrc
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -Wall -Wno-unused-function
LIB   := hevcbitstream_amd/libhevcbitstream_amd.so
HIP_SRCS := $(wildcard $(CSRC)/*.hip)
HDRS  := $(wildcard $(CSRC)/*.h) $(wildcard include/*.h)

all: lib oracle sim analyze

lib: $(LIB)

# the legacy single-NAL API is plain C over the C ABI (no HIP headers): gcc compiles it
LEGACY_OBJ := build/obj/hbs_legacy_c.o
$(LEGACY_OBJ): $(CSRC)/hbs_legacy.c $(HDRS)
	@mkdir -p build/obj
	$(CC) -std=c99 -O2 -fPIC -Wall -Wextra -Iinclude -I$(CSRC) -c -o $@ $<

HIP_OBJS := $(patsubst $(CSRC)/%.hip,build/obj/%.o,$(HIP_SRCS))
build/obj/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

# only the C ABI is exported (csrc/exports.map; tests/test_abi_exports.py checks nm -D against include/*.h)
EXPORTS := -Wl,--version-script=hevcbitstream_amd/csrc/exports.map
$(LIB): $(HIP_OBJS) $(LEGACY_OBJ) hevcbitstream_amd/csrc/exports.map
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(EXPORTS) -o $@ $(HIP_OBJS) $(LEGACY_OBJ) -ldl -lpthread
	ln -sf libhevcbitstream_amd.so hevcbitstream_amd/libhevcbitstream.so

# the reference's own CLI, unmodified, against OUR headers and library (dev container only)
REF ?= /root/reference
analyze: $(LIB)
	@if [ -f $(REF)/hevc_analyze.c ]; then mkdir -p oracle/_ref && \
	  $(CC) -std=c99 -O2 -w -D_GNU_SOURCE -Iinclude -o oracle/_ref/hevc_analyze_amd $(REF)/hevc_analyze.c \
	    -Lhevcbitstream_amd -lhevcbitstream_amd -Wl,-rpath,'$$ORIGIN/../../hevcbitstream_amd' && echo built oracle/_ref/hevc_analyze_amd; \
	 else echo "reference sources absent: keeping prebuilt oracle/_ref/hevc_analyze_amd (if any)"; fi

oracle:
	$(MAKE) -C oracle all

sim:
	$(MAKE) -C tests/sim

clean:
	rm -f $(LIB) tests/sim/libhbs_sim.so
	$(MAKE) -C oracle clean

# diagnostic build with per-phase shader-clock sums (tests/tools/phase_timing*.py); never the shipped library
DIAG_OBJS := $(patsubst $(CSRC)/%.hip,build/diag/%.o,$(HIP_SRCS))
build/diag/%.o: $(CSRC)/%.hip $(HDRS)
	mkdir -p build/diag && $(HIPCC) $(HIPFLAGS) -DHBS_PHASE_TIMING -c -o $@ $<
diag: $(DIAG_OBJS) $(LEGACY_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(EXPORTS) -o build/diag/libhbs_diag.so $(DIAG_OBJS) $(LEGACY_OBJ)

# development variant of the library with extra -D flags, for A/B timing (HBS_LIB=build/variants/<NAME>/libhbs.so)
#   make variant NAME=ntload DEFS="-DHBS_NT_LOAD=1"
#   make variant NAME=d4 DEFS="-DHBS4_COPY_DEPTH=4" ONLY="hbs_scan4"      (recompile only the named files; the rest from build/obj)
ONLY ?= $(patsubst $(CSRC)/%.hip,%,$(HIP_SRCS))
VAR_OBJS := $(patsubst %,build/variants/$(NAME)/%.o,$(ONLY))
VAR_REST := $(filter-out $(patsubst %,build/obj/%.o,$(ONLY)),$(HIP_OBJS))
build/variants/$(NAME)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build/variants/$(NAME)
	$(HIPCC) $(HIPFLAGS) $(DEFS) -c -o $@ $<
variant: $(VAR_OBJS) $(VAR_REST) $(LEGACY_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(EXPORTS) -o build/variants/$(NAME)/libhbs.so $(VAR_OBJS) $(VAR_REST) $(LEGACY_OBJ)

.PHONY: all lib oracle sim clean analyze diag variant
