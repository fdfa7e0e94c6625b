# Builds libmvlpt_hip.so (hand-written gfx950 kernels + C ABI) in-tree, and the CPU oracle helpers.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := mvlpt_amd/csrc
OBJDIR := build/obj
SRCS  := $(CSRC)/gemm.hip $(CSRC)/gemm_duo.hip $(CSRC)/norm.hip $(CSRC)/attention.hip $(CSRC)/attention_stream.hip $(CSRC)/attention32.hip $(CSRC)/glue.hip $(CSRC)/preprocess.hip $(CSRC)/engine.hip
OBJS  := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))
# -target-feature -packed-fp32-ops: NO packed fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32) in
# the device code.  On gfx950 a packed fp32 op whose LOW lane selects the HIGH register of a source pair (op_sel) can read 0 in lanes
# 48-63 when the pair was written one instruction (or one LDS return) earlier and another wave owns the matrix pipe — a hazard ROCm 7.2's
# hipcc does not pad (NOTES_experiments.md round 6; tools/pkfma_hazard.hip; tests/test_isa_audit.py keeps the library free of them).
# Packed fp32 is not a rate doubler on gfx950: the image tower is 0.7 % FASTER without it.  (The host pass ignores the feature with a
# warning, filtered below.)
NOPK  := -Xclang -target-feature -Xclang -packed-fp32-ops
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -Wno-inline-asm $(NOPK)
SHELL := /bin/bash
QUIET := 2> >(grep -v "is not a recognized feature for this target" >&2)
LIB   := mvlpt_amd/libmvlpt_hip.so

ORACLE_SO := oracle/_build/libresample_oracle.so

# Identity of the binary (mvlpt_version()): sha256 over the sources it is compiled from, and the git commit when there is one.
# bench.py compares it with the hash recorded in profiles/gemm_hbm_traffic*.json and drops a traffic figure taken on another binary.
HDRS     := $(sort $(wildcard $(CSRC)/*.h) include/mvlpt_hip.h)
SRC_HASH := $(shell cat $(sort $(SRCS)) $(HDRS) Makefile | sha256sum | cut -c1-12)
GIT_HASH := $(shell git rev-parse --short=12 HEAD 2>/dev/null || echo nogit)
STAMP    := $(OBJDIR)/version.stamp
$(shell mkdir -p $(OBJDIR); echo "$(SRC_HASH) $(GIT_HASH)" | cmp -s - $(STAMP) || echo "$(SRC_HASH) $(GIT_HASH)" > $(STAMP))

all: $(LIB) $(ORACLE_SO)

# CPU oracle of the input pipeline (test infrastructure only: never linked into $(LIB))
$(ORACLE_SO): oracle/resample_oracle.c
	@mkdir -p oracle/_build
	gcc -O2 -ffp-contract=off -shared -fPIC -o $@ $< -lm

$(OBJDIR)/engine.o: $(CSRC)/engine.hip $(HDRS) $(STAMP) Makefile
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -DMVLPT_SRC_HASH='"$(SRC_HASH)"' -DMVLPT_GIT_HASH='"$(GIT_HASH)"' -c $< -o $@ $(QUIET)

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS) Makefile
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -c $< -o $@ $(QUIET)

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -rf build $(LIB) oracle/_build

.PHONY: all clean
