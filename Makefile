# Builds libmvlpt_hip.so (hand-written gfx950 kernels + C ABI) in-tree, and the CPU oracle helpers.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := mvlpt_amd/csrc
OBJDIR := build/obj
SRCS  := $(CSRC)/gemm.hip $(CSRC)/gemm_duo.hip $(CSRC)/norm.hip $(CSRC)/attention.hip $(CSRC)/attention_stream.hip $(CSRC)/attention32.hip $(CSRC)/glue.hip $(CSRC)/preprocess.hip $(CSRC)/engine.hip
OBJS  := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -Wno-inline-asm
LIB   := mvlpt_amd/libmvlpt_hip.so

ORACLE_SO := oracle/_build/libresample_oracle.so

all: $(LIB) $(ORACLE_SO)

# CPU oracle of the input pipeline (test infrastructure only: never linked into $(LIB))
$(ORACLE_SO): oracle/resample_oracle.c
	@mkdir -p oracle/_build
	gcc -O2 -ffp-contract=off -shared -fPIC -o $@ $< -lm

$(OBJDIR)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/kernels.h $(CSRC)/attn_common.h $(CSRC)/gemm_epi.h include/mvlpt_hip.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -rf build $(LIB) oracle/_build

.PHONY: all clean
