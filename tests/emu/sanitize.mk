# Sanitizer builds of the emulated kernels: CPU only (g++ on the host emulation of the kernel sources).  Included by
# tests/emu/Makefile when present; listed in .gpurunignore — sanitizers are not run on the GPU pool, and nothing a GPU box
# runs needs these recipes.
# AddressSanitizer build of the emulated kernels (heap out-of-bounds reads / writes of device buffers and LDS in any kernel):
#   make -C tests/emu asan && cp tests/emu/asan/libplonk_emu.so tests/emu/ && \
#   LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
#     python -m pytest tests/test_emu_parity.py tests/test_distributed_gloo.py tests/test_bench_cli.py ; make -C tests/emu clean all
# (ucontext fibers: ASan prints one "doesn't fully support swapcontext" warning; round 2's final sources pass clean.
#  The same recipe with -fsanitize=signed-integer-overflow,bounds and libubsan.so checks the signed-limb arithmetic.)
asan:
	@mkdir -p asan
	$(MAKE) -s BUILD_ASAN=1 asan/libplonk_emu.so
asan/libplonk_emu.so: $(SRCS) hip_emu.cpp comm_stub.cpp $(HDRS)
	for f in $(SRCS); do $(CXX) $(CXXFLAGS) -O1 -g -fsanitize=address -fno-omit-frame-pointer -x c++ -c $$f -o asan/$$(basename $$f .hip).o || exit 1; done
	$(CXX) $(CXXFLAGS) -O1 -g -fsanitize=address -fno-omit-frame-pointer -c hip_emu.cpp -o asan/hip_emu.o
	$(CXX) $(CXXFLAGS) -O1 -g -fsanitize=address -fno-omit-frame-pointer -c comm_stub.cpp -o asan/comm_stub.o
	$(CXX) -shared -fsanitize=address -o $@ asan/*.o

# the same with UBSan (signed overflow of the limb arithmetic, array bounds):  make -C tests/emu ubsan && cp tests/emu/ubsan/libplonk_emu.so tests/emu/ &&
#   LD_PRELOAD=$(gcc -print-file-name=libubsan.so) python -m pytest tests/test_emu_parity.py ; make -C tests/emu clean all
UBSAN := -fsanitize=signed-integer-overflow,bounds -fno-sanitize-recover=signed-integer-overflow,bounds
ubsan:
	@mkdir -p ubsan
	for f in $(SRCS); do $(CXX) $(CXXFLAGS) -O1 -g $(UBSAN) -x c++ -c $$f -o ubsan/$$(basename $$f .hip).o || exit 1; done
	$(CXX) $(CXXFLAGS) -O1 -g $(UBSAN) -c hip_emu.cpp -o ubsan/hip_emu.o
	$(CXX) $(CXXFLAGS) -O1 -g $(UBSAN) -c comm_stub.cpp -o ubsan/comm_stub.o
	$(CXX) -shared $(UBSAN) -o ubsan/libplonk_emu.so ubsan/*.o

