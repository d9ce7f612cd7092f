# CPU-only: the emulator's farm build under the sanitizers (tests/test_emu_kernel.py runs it in a subprocess with libasan
# preloaded).  Kept out of the Makefile and out of the GPU snapshot (.gpurunignore): sanitizer builds are for the CPU.
CXX ?= g++
CSRC := ../../slip_lu_amd/csrc
libslip_emu_san.so: $(CSRC)/slip_hip.hip $(CSRC)/ref_lu_pipe.h $(CSRC)/ref_lu_pipe_cols.h $(CSRC)/ref_lu_pipe_commit.h $(CSRC)/wave_bigint.h $(CSRC)/wave_bigint_reg.h $(CSRC)/wave_shim.h fiber_emu.h hip_rt_emu.h
	$(CXX) -O1 -g -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -DSLIP_EMULATE -DSLIP_FARM_MIN_COST=0 -DSLIP_FARM_MIN_ITEMS=2 -DSLIP_FARM_NEAR_DIV=0 -DSLIP_FARM_KIND2=1 -DSLIP_FARM_KIND2_COST=0 -I. -I$(CSRC) -x c++ $(CSRC)/slip_hip.hip -o $@
