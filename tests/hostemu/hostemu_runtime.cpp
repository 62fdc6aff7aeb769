// TEST INFRASTRUCTURE ONLY — fiber scheduler behind hostemu_runtime.h (x86-64 System V).
#include "hostemu_runtime.h"
#include <vector>
#include <cstdio>

hostemu_uint3 threadIdx, blockIdx, blockDim, gridDim;

extern "C" void hostemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hostemu_switch
.type hostemu_switch,@function
hostemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hostemu_switch,.-hostemu_switch
)");

namespace hostemu {

char* dyn_smem = nullptr;

enum State { READY, WAIT_BLOCK, WAIT_WAVE, DONE };
struct Fiber { void* sp; State st; char* stack; };
static std::vector<Fiber> fibers;
static std::vector<char*> stackPool;
static void* schedSp = nullptr;
static int cur = -1;
static const std::function<void()>* curBody = nullptr;
static const size_t kStack = 256 * 1024;

static void yield_to_sched(State s) {
	fibers[cur].st = s;
	hostemu_switch(&fibers[cur].sp, schedSp);
}
void syncthreads() { yield_to_sched(WAIT_BLOCK); }
void wave_sync() { yield_to_sched(WAIT_WAVE); }

static void trampoline() {
	(*curBody)();
	yield_to_sched(DONE);
	abort(); // never resumed
}

static void set_tid(unsigned i) {
	threadIdx.x = i % blockDim.x;
	threadIdx.y = (i / blockDim.x) % blockDim.y;
	threadIdx.z = i / (blockDim.x * blockDim.y);
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
	const unsigned nthr = block.x * block.y * block.z;
	std::vector<char> smem(shmem + 64);
	dyn_smem = (char*)(((uintptr_t)smem.data() + 63) & ~(uintptr_t)63);
	while (stackPool.size() < nthr) stackPool.push_back((char*)aligned_alloc(64, kStack));
	blockDim = {block.x, block.y, block.z};
	gridDim = {grid.x, grid.y, grid.z};
	curBody = &body;
	fibers.resize(nthr);
	for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bx = 0; bx < grid.x; bx++) {
		blockIdx = {bx, by, bz};
		for (unsigned i = 0; i < nthr; i++) {
			char* top = stackPool[i] + kStack;
			void** sp = (void**)(((uintptr_t)top) & ~(uintptr_t)15);
			*--sp = nullptr;              // alignment slot: after `ret` pops the entry, rsp % 16 == 8 as at a call
			*--sp = (void*)&trampoline;   // return address for the first switch
			for (int k = 0; k < 6; k++) *--sp = nullptr; // rbp rbx r12 r13 r14 r15
			fibers[i] = {(void*)sp, READY, stackPool[i]};
		}
		unsigned live = nthr;
		// Thread order between barriers (VKFFT_HOSTEMU_ORDER): 0 ascending (default), 1 descending, 2 pseudo-random (changes every
		// round).  A missing barrier shows up as a wrong result only when a thread reads what a LATER-scheduled thread writes, so
		// the suite is also run with the other orders (tests/test_emu_fuzz.py) to expose hazards ascending order would mask.
		static const int order = getenv("VKFFT_HOSTEMU_ORDER") ? atoi(getenv("VKFFT_HOSTEMU_ORDER")) : 0;
		uint32_t rng = 0x9E3779B9u ^ (bx * 2654435761u);
		std::vector<unsigned> perm(nthr);
		while (live) {
			for (unsigned i = 0; i < nthr; i++) perm[i] = order == 1 ? nthr - 1 - i : i;
			if (order == 2) for (unsigned i = nthr; i > 1; i--) { rng = rng * 1664525u + 1013904223u; std::swap(perm[i - 1], perm[(rng >> 8) % i]); }
			for (unsigned ii = 0; ii < nthr; ii++) {
				const unsigned i = perm[ii];
				if (fibers[i].st != READY) continue;
				cur = (int)i;
				set_tid(i);
				hostemu_switch(&schedSp, fibers[i].sp);
				if (fibers[i].st == DONE) live--;
			}
			if (!live) break;
			// release barriers
			bool allBlock = true;
			for (unsigned i = 0; i < nthr; i++) if (fibers[i].st != DONE && fibers[i].st != WAIT_BLOCK) { allBlock = false; break; }
			if (allBlock) { for (unsigned i = 0; i < nthr; i++) if (fibers[i].st == WAIT_BLOCK) fibers[i].st = READY; continue; }
			bool progress = false;
			for (unsigned w = 0; w < nthr; w += 64) {
				bool allWave = true, any = false;
				for (unsigned i = w; i < w + 64 && i < nthr; i++) {
					if (fibers[i].st == DONE) continue;
					any = true;
					if (fibers[i].st != WAIT_WAVE) { allWave = false; break; }
				}
				if (any && allWave) { for (unsigned i = w; i < w + 64 && i < nthr; i++) if (fibers[i].st == WAIT_WAVE) fibers[i].st = READY; progress = true; }
			}
			if (!progress) { fprintf(stderr, "hostemu: divergent barrier (deadlock) in block %u\n", bx); abort(); }
		}
	}
	dyn_smem = nullptr;
}

hipError_t e_malloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) & ~(size_t)255); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t e_free(void* p) { free(p); return hipSuccess; }
hipError_t e_memcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t e_attr(int* v, hipDeviceAttribute_t a, int) {
	switch (a) {
	case hipDeviceAttributeWarpSize: *v = 64; break;
	case hipDeviceAttributeMaxThreadsPerBlock: *v = 1024; break;
	case hipDeviceAttributeMaxGridDimX: *v = 2147483647; break;
	case hipDeviceAttributeMaxGridDimY: case hipDeviceAttributeMaxGridDimZ: *v = 65536; break;
	case hipDeviceAttributeMaxSharedMemoryPerBlock: *v = 163840; break;
	case hipDeviceAttributeComputeCapabilityMajor: *v = 9; break;
	case hipDeviceAttributeComputeCapabilityMinor: *v = 5; break;
	default: *v = 0; break;
	}
	return hipSuccess;
}
hipError_t e_event_create(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t e_event_destroy(hipEvent_t e) { free((void*)e); return hipSuccess; }

} // namespace hostemu
