# Top-level build without Python, for a C host (the reference builds with make/build.sh):
#   make            libsynth_mi355x.so (hipcc, gfx950), the host programs, the test oracle
#   make lib        only synth_tools_amd/libsynth_mi355x.so
# `python -m synth_tools_amd.build` / `__graft_entry__.build()` produce the same files.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := synth_tools_amd/csrc
OBJDIR  := synth_tools_amd/build
LIB     := synth_tools_amd/libsynth_mi355x.so
SRCS    := saw_bank.hip pdm_bank.hip poly_bank.hip pwm_bank.hip osc_bank.hip cproc_bank.hip \
           abi_core.cpp abi_saw.cpp abi_pdm.cpp abi_pwm.cpp abi_poly.cpp abi_osc.cpp abi_cproc.cpp abi_fw.cpp
OBJS    := $(addprefix $(OBJDIR)/,$(addsuffix .o,$(SRCS)))
HDRS    := $(CSRC)/smx_common.h $(CSRC)/abi_internal.h include/synth_mi355x.h
HFLAGS  := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result

all: lib hosts oracle
lib: $(LIB)

$(OBJDIR)/%.o: $(CSRC)/% $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HFLAGS) -c $< -o $@

$(LIB): $(OBJS) $(CSRC)/exports.map
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -L/opt/rocm/lib -lrccl \
	    -Wl,-rpath,/opt/rocm/lib -Wl,--version-script=$(CSRC)/exports.map

hosts: $(LIB)
	$(MAKE) -C host

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(OBJDIR) $(LIB)
	$(MAKE) -C host clean
	$(MAKE) -C oracle clean
.PHONY: all lib hosts oracle clean
