# Builds the product library (HIP, gfx950) in-tree.  `make` = product; `make oracle` = CPU checkers.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := vkfft_amd/csrc
LIBDIR := vkfft_amd/lib
# --offload-compress: the gfx950 code objects are stored zstd-compressed in the library (165 MB -> a third) and unpacked by the HIP runtime when the module loads
CXXFLAGS := -O3 -std=c++17 -fPIC -fvisibility=hidden -Iinclude -I$(CSRC) -Wno-unused-result --offload-compress
OBJS := build/obj/api.o build/obj/planner.o build/obj/kernels.o build/obj/kernels_pow2.o build/obj/kernels_blue_r2r.o build/obj/kernels_fused.o build/obj/kernels_mixfused.o build/obj/kernels_aux.o build/obj/kernels_mixed_0.o build/obj/kernels_mixed_1.o build/obj/kernels_mixed_2.o build/obj/kernels_mixed_3.o build/obj/kernels_mixed_4.o build/obj/kernels_mixed_5.o build/obj/kernels_mixed_6.o build/obj/kernels_mixed_7.o build/obj/kernels_mixed_8.o build/obj/kernels_mixed_9.o build/obj/kernels_mixed_10.o build/obj/kernels_mixed_11.o build/obj/kernels_mixed_12.o build/obj/kernels_mixed_13.o build/obj/kernels_mixed_14.o build/obj/kernels_mixed_15.o build/obj/kernels_mixed_16.o build/obj/kernels_mixed_17.o build/obj/kernels_mixed_18.o build/obj/kernels_mixed_19.o \
        build/obj/kernels_mixconv_0.o build/obj/kernels_mixconv_1.o build/obj/kernels_mixconv_2.o build/obj/kernels_mixconv_3.o build/obj/kernels_mixconv_4.o build/obj/kernels_mixconv_5.o \
        $(foreach t,f32_row f32_col f64_row f64_col,build/obj/kernels_opfft_$(t)_0.o build/obj/kernels_opfft_$(t)_1.o) build/obj/kernels_opfft_f32_col_2.o
# header dependencies come from the compiler (-MMD): a change to one kernel family rebuilds only the translation units that include it
DEPFLAGS = -MMD -MP -MF build/obj/$*.d

all: $(LIBDIR)/libvkfft_mi355x.so build/vkfft_mi355x_cli $(if $(wildcard /opt/rocm/lib/librccl.so),build/vkfft_mi355x_multi)
multi: build/vkfft_mi355x_multi

# caller-side benchmark driver (flag-compatible in spirit with the reference's VkFFT_TestSuite): links the C-ABI only
build/vkfft_mi355x_cli: tools/vkfft_cli.cpp include/vkFFT.h $(LIBDIR)/libvkfft_mi355x.so
	@mkdir -p build
	$(HIPCC) -O2 -std=c++17 -Wno-unused-value -Wno-unused-result -Iinclude tools/vkfft_cli.cpp -L$(LIBDIR) -lvkfft_mi355x -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -o $@

# multi-GPU host drivers (one thread per GPU; batch sharding, slab 3D over RCCL or device-to-device copies): links the C-ABI and librccl
build/vkfft_mi355x_multi: tools/vkfft_multi.cpp include/vkFFT.h $(LIBDIR)/libvkfft_mi355x.so
	@mkdir -p build
	$(HIPCC) -O2 -std=c++17 -pthread -Wno-unused-value -Wno-unused-result -Iinclude tools/vkfft_multi.cpp -L$(LIBDIR) -lvkfft_mi355x -L/opt/rocm/lib -lrccl -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -Wl,-rpath,/opt/rocm/lib -o $@

build/obj/%.o: $(CSRC)/%.cpp
	@mkdir -p build/obj
	$(HIPCC) $(CXXFLAGS) $(DEPFLAGS) --offload-arch=$(ARCH) -c $< -o $@

build/obj/%.o: $(CSRC)/%.hip
	@mkdir -p build/obj
	$(HIPCC) $(CXXFLAGS) $(DEPFLAGS) --offload-arch=$(ARCH) -c $< -o $@

$(LIBDIR)/libvkfft_mi355x.so: $(OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(OBJS) -o $@

# development build of the fused Four-Step kernels (per-phase cycle profile, arithmetic-free variants): tools/prof_fused.py,
# selected with VKFFT_MI355X_LIB=build/libvkfft_mi355x_dev.so; never shipped
build/obj/kernels_fused_dev.o: $(CSRC)/kernels_fused.hip
	@mkdir -p build/obj
	$(HIPCC) $(CXXFLAGS) -DVKFFT_MI355X_DEV -MMD -MP -MF build/obj/kernels_fused_dev.d --offload-arch=$(ARCH) -c $< -o $@
build/libvkfft_mi355x_dev.so: $(OBJS) build/obj/kernels_fused_dev.o
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(filter-out build/obj/kernels_fused.o,$(OBJS)) build/obj/kernels_fused_dev.o -o $@
dev: build/libvkfft_mi355x_dev.so

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build/obj $(LIBDIR)/*.so

-include $(wildcard build/obj/*.d)

.PHONY: all oracle clean dev multi
