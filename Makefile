# Builds the gfx950 HIP library (C ABI in include/lgteun_hip.h) in-tree.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := lgteun_amd/csrc
SRCS  := $(CSRC)/api.hip $(CSRC)/k_pixel.hip $(CSRC)/k_dstep.hip $(CSRC)/k_fft.hip $(CSRC)/k_attn.hip $(CSRC)/k_attn_m.hip $(CSRC)/k_ffn.hip $(CSRC)/k_ffn_prep.hip $(CSRC)/k_ffn_x.hip $(CSRC)/k_ffn_xr.hip $(CSRC)/k_ffn_x32.hip $(CSRC)/k_ffn_x64.hip $(CSRC)/k_bwd.hip $(CSRC)/k_bwd_pixel.hip $(CSRC)/k_wgrad.hip $(CSRC)/k_ffn_bwd.hip $(CSRC)/k_ffn_bwd_x.hip $(CSRC)/k_ffn_dwbwd_x.hip $(CSRC)/k_ffn_dwbwd_h.hip $(CSRC)/k_ffn1_bwd_x32.hip $(CSRC)/k_attn_bwd.hip $(CSRC)/k_attn_bwd_f.hip $(CSRC)/k_attn_bwd_m.hip
# A/B-only kernels stay out of the product library: `make AB=1` adds k_ffn_xp (LG_FFN_IMPL=xp: the software-pipelined variant of the
# fused FFN forward, bitwise the same results, measured 2.5 % slower)
# (AB objects get their own suffix, so a product build never links objects compiled with the other flag set and vice versa)
ifdef AB
SRCS  += $(CSRC)/k_ffn_xp.hip
ABFLAGS := -DLG_BUILD_AB=1
OSUF  := .ab.o
LIB   := lgteun_amd/_lgteun_hip_ab.so   # use it with LGTEUN_HIP_LIB=$PWD/lgteun_amd/_lgteun_hip_ab.so
else
OSUF  := .o
LIB   := lgteun_amd/_lgteun_hip.so
endif
OBJS  := $(SRCS:.hip=$(OSUF))
# -fno-slp-vectorize: the SLP vectoriser turns scalar fp32 chains into v_pk_mul_f32 / v_pk_add_f32 pairs; packed fp32 issues at
# half rate on gfx950 and the pairing blocks mul+add -> fma contraction (k_attn: 3140 VALU instructions, 502 of them packed,
# vs 3099 unpacked).  Measured on one box, alternating runs: 9.28 -> 9.10 ms/step fp32, 8.96 -> 8.68 bf16 mode.
FLAGS := -O3 -std=c++17 -fno-slp-vectorize -fPIC -fvisibility=hidden --offload-arch=$(ARCH) -Wall -Wno-unused-function -Wno-unused-value $(ABFLAGS)

all: $(LIB)

# k_attn_m.hip: matrix-core results straight into VGPRs (the default form parks them in AGPRs at this register pressure: one
# v_accvgpr_read per value) and no canonicalising v_max in front of the softmax's fmaxf chain
$(CSRC)/k_attn_m$(OSUF): FLAGS += -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans
$(CSRC)/k_attn_bwd_m$(OSUF): FLAGS += -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans
# k_ffn_xr: the iterative ILP scheduler instead of the default max-occupancy one (the kernel is pinned at two waves per SIMD anyway): 77.25 -> 76.36 us
# per fused-FFN launch, 5.134 -> 5.119 ms per step (same-box A/B; max-ilp measures the same; neither pays in k_attn_m, k_ffn1_bwd_xs, k_fftmix_r, k_attn_bwd_f)
$(CSRC)/k_ffn_xr$(OSUF): FLAGS += -mllvm -amdgpu-sched-strategy=iterative-ilp

# prerequisites come from the compiler (-MMD writes one .d file per object: every header a source includes, hstore.h too)
# (every object also depends on this file: a changed flag must rebuild what it applies to)
$(CSRC)/%$(OSUF): $(CSRC)/%.hip Makefile
	$(HIPCC) $(FLAGS) -MMD -MP -c $< -o $@

-include $(OBJS:.o=.d)

$(LIB): $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(OBJS) -o $@

clean:
	rm -f $(CSRC)/*.o $(CSRC)/*.d $(LIB)
.PHONY: all clean
