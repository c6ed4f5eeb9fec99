// tools/microbench_mall.hip -- which round trips of the chain reach HBM?  ("Missing 5")
// The intermediates of a step (48 kHz channels 0.10 GB, FIR outputs 0.10 GB, PhaseSearch scratch 0.03 GB) are written by one kernel and
// read by another ~0.5 ms later, while the front end streams 1.6 GB of input through the chip with the non-temporal hint.  The
// fabric-side counters (TCC_EA0_RDREQ -> FETCH_SIZE) sit between L2 and the Infinity Cache (256 MiB, memory side), so they cannot tell
// an Infinity-Cache hit from an HBM read.  Time can: this program writes a buffer C, optionally streams a big buffer A past it, and
// times the read-back of C -- cold (after the caches were flushed with other data), hot (right after the write), and behind a stream
// of A with and without the non-temporal hint.  Each read-back kernel has its own name (template tag), so a `rocprofv3 --pmc
// FETCH_SIZE` pass of the same binary shows whether the counter moves with the timing or not.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_mall.hip -o tools/microbench_mall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void fill(float4* p, size_t n4, float v) {
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = make_float4(v, v + 1, v + 2, v + 3);
}
template <int TAG>
__global__ __launch_bounds__(256) void readback(const float4* __restrict__ p, size_t n4, float* out) {
	float acc = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
	if (acc == 12345.678f) out[0] = acc;
}
// the front end's read pattern: one-wave workgroups, 8 KB tiles straight into LDS, AUX = 2: non-temporal (what k1_dpp uses), 0: default
template <int AUX>
__global__ __launch_bounds__(64) void stream_tiles(const float4* __restrict__ in, float* out, int tiles_per_span) {
	__shared__ float4 tile[512];
	const float4* src = in + (size_t)blockIdx.x * tiles_per_span * 512;
	const int lane = threadIdx.x;
	float acc = 0;
	for (int e = 0; e < 8; e++) __builtin_amdgcn_global_load_lds((const void*)(src + e * 64 + lane), (__attribute__((address_space(3))) void*)(tile + e * 64), 16, 0, AUX);
	for (int t = 0; t < tiles_per_span; t++) {
		__builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
		float4 cur[8];
		for (int e = 0; e < 8; e++) cur[e] = tile[e * 64 + lane];
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
		const int tn = t + 1 < tiles_per_span ? t + 1 : t;
		for (int e = 0; e < 8; e++) __builtin_amdgcn_global_load_lds((const void*)(src + (size_t)tn * 512 + e * 64 + lane), (__attribute__((address_space(3))) void*)(tile + e * 64), 16, 0, AUX);
		for (int e = 0; e < 8; e++) acc += cur[e].x + cur[e].y + cur[e].z + cur[e].w;
	}
	if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char** argv) {
	const size_t A_BYTES = (size_t)256 * 786432 * 8;  // one step's input
	const size_t F_BYTES = (size_t)1 << 30;           // flush buffer
	float4 *A, *F, *C;
	float* out;
	CHK(hipMalloc(&A, A_BYTES)); CHK(hipMalloc(&F, F_BYTES)); CHK(hipMalloc(&C, (size_t)512 << 20)); CHK(hipMalloc(&out, 64));
	fill<<<4096, 256>>>(A, A_BYTES / 16, 1.0f);
	fill<<<4096, 256>>>(F, F_BYTES / 16, 2.0f);
	CHK(hipDeviceSynchronize());
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const int spans = 256 * 24, tps = (int)(A_BYTES / 8192 / spans); // 6,144 one-wave workgroups x 32 tiles, like the front end
	auto flush = [&]() { fill<<<4096, 256>>>(F, F_BYTES / 16, 3.0f); readback<99><<<4096, 256>>>(F, F_BYTES / 16, out); };
	auto timed = [&](auto&& launch) { CHK(hipEventRecord(e0)); launch(); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1000.0f; };
	const int REP = 7;
	printf("read-back of a buffer C (grid-stride, 16 B per lane, 4,096 x 256 threads), median of %d, microseconds (GB/s)\n", REP);
	printf("%8s | %16s %16s %16s %16s %16s %16s | %14s %14s\n", "C", "cold", "after write", "after read", "W, stream nt", "W, stream dflt", "W, 2x stream nt", "stream nt", "stream dflt");
	for (size_t mb : { (size_t)26, (size_t)100, (size_t)200, (size_t)400 }) {
		const size_t n4 = (mb << 20) / 16;
		std::vector<float> t[6], ts[2];
		for (int r = 0; r < REP; r++) {
			flush(); fill<<<4096, 256>>>(C, n4, 1.0f); flush();
			t[0].push_back(timed([&]() { readback<0><<<4096, 256>>>(C, n4, out); }));
			flush(); fill<<<4096, 256>>>(C, n4, 1.0f);
			t[1].push_back(timed([&]() { readback<1><<<4096, 256>>>(C, n4, out); }));
			t[2].push_back(timed([&]() { readback<2><<<4096, 256>>>(C, n4, out); }));
			flush(); fill<<<4096, 256>>>(C, n4, 1.0f);
			ts[0].push_back(timed([&]() { stream_tiles<2><<<spans, 64>>>(A, out, tps); }));
			t[3].push_back(timed([&]() { readback<3><<<4096, 256>>>(C, n4, out); }));
			flush(); fill<<<4096, 256>>>(C, n4, 1.0f);
			ts[1].push_back(timed([&]() { stream_tiles<0><<<spans, 64>>>(A, out, tps); }));
			t[4].push_back(timed([&]() { readback<4><<<4096, 256>>>(C, n4, out); }));
			flush(); fill<<<4096, 256>>>(C, n4, 1.0f);
			stream_tiles<2><<<spans, 64>>>(A, out, tps); stream_tiles<2><<<spans, 64>>>(A, out, tps);
			t[5].push_back(timed([&]() { readback<5><<<4096, 256>>>(C, n4, out); }));
		}
		auto med = [](std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
		printf("%5zu MB |", mb);
		for (int k = 0; k < 6; k++) { const float us = med(t[k]); printf(" %7.1f (%6.0f)", us, (double)(mb << 20) / us * 1e-3); }
		printf(" |");
		for (int k = 0; k < 2; k++) { const float us = med(ts[k]); printf(" %6.1f (%5.0f)", us, (double)A_BYTES / us * 1e-3); }
		printf("\n");
	}
	// the same question for a WRITE into a resident buffer: does writing C again (as every step does) cost HBM write bandwidth?
	printf("\nwrite of C (fill kernel), microseconds (GB/s): cold (after flush) / again right after\n");
	for (size_t mb : { (size_t)26, (size_t)100, (size_t)200, (size_t)400 }) {
		const size_t n4 = (mb << 20) / 16;
		std::vector<float> a, b;
		for (int r = 0; r < REP; r++) {
			flush();
			a.push_back(timed([&]() { fill<<<4096, 256>>>(C, n4, 1.0f); }));
			b.push_back(timed([&]() { fill<<<4096, 256>>>(C, n4, 2.0f); }));
		}
		std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
		printf("%5zu MB | %7.1f (%6.0f)   %7.1f (%6.0f)\n", mb, a[REP / 2], (double)(mb << 20) / a[REP / 2] * 1e-3, b[REP / 2], (double)(mb << 20) / b[REP / 2] * 1e-3);
	}
	return 0;
}
