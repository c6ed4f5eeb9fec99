// Which physical CUs does a hipExtStreamCreateWithCUMask stream use?  For a few masks, launch 2048 one-wave
// workgroups that record (XCC_ID, SE_ID, CU_ID) and print the set of distinct locations.
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench_cumask.hip -o tools/microbench_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>

__global__ void where(uint32_t* out) {
	uint32_t hw, xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	// spin a little so that workgroups spread over every CU the queue may use
	long long t0 = clock64();
	while (clock64() - t0 < 20000) {}
	if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
	hipStream_t s;
	hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
	if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return; }
	const int n = 2048;
	uint32_t* d;
	hipMalloc(&d, n * 4);
	hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, s, d);
	hipStreamSynchronize(s);
	std::vector<uint32_t> h(n);
	hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
	std::set<uint32_t> locs;
	int per_xcc[16] = {};
	for (auto v : h) {
		// HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
		uint32_t cu = (v >> 8) & 0xf, sh = (v >> 12) & 1, se = (v >> 13) & 7, xcc = v >> 16;
		uint32_t key = xcc << 12 | se << 8 | sh << 4 | cu;
		if (locs.insert(key).second) per_xcc[xcc]++;
	}
	printf("%s: %zu distinct CUs; per XCC:", name, locs.size());
	for (int i = 0; i < 8; i++) printf(" %d", per_xcc[i]);
	printf("\n");
	if (locs.size() <= 16) {
		printf("   ");
		for (auto k : locs) printf(" (xcc%u se%u cu%u)", k >> 12, (k >> 8) & 7, k & 15);
		printf("\n");
	}
	hipFree(d);
	hipStreamDestroy(s);
}

int main() {
	run("all 256", std::vector<uint32_t>(8, 0xffffffffu));
	run("bits 0-7", { 0xffu, 0, 0, 0, 0, 0, 0, 0 });
	run("bits 0-31", { 0xffffffffu, 0, 0, 0, 0, 0, 0, 0 });
	run("bit 0", { 1u, 0, 0, 0, 0, 0, 0, 0 });
	run("bit 8", { 0x100u, 0, 0, 0, 0, 0, 0, 0 });
	run("bits 248-255", { 0, 0, 0, 0, 0, 0, 0, 0xff000000u });
	run("all but 0-7", { 0xffffff00u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu });
	run("every 32nd", { 1u, 1u, 1u, 1u, 1u, 1u, 1u, 1u });
	return 0;
}
