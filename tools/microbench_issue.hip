// tools/microbench_issue.hip -- issue cost per instruction CLASS on gfx950, in SIMD cycles per wave64 instruction: the numbers
// behind the per-kernel "issue ms" column of profiles/r03_issue_model.txt (SQ_INSTS_VALU x class share x cost).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_issue.hip -o /tmp/mb_issue && /tmp/mb_issue
// Every kernel is a loop of 32 independent instructions of one class (8 registers x 4), W one-wave workgroups per SIMD
// (W = 1, 2, 4: a single wave's issue rate against the SIMD's).  The clock comes from a loop of `s_nop 15`: 16 wait states of one
// 4-cycle issue slot each = 64 cycles (with 16 cycles the plain f32 rate would come out at 0.6 cycles per wave64 instruction, i.e.
// 100 lanes per clock on a 32-lane SIMD; the device's reported clockRate is printed next to it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define KERNEL(name, TYPE, INITV, I0, I1, I2, I3, I4, I5, I6, I7) \
	__global__ __launch_bounds__(64) void name(float* out, int iters) { \
		TYPE r0 = INITV, r1 = INITV, r2 = INITV, r3 = INITV, r4 = INITV, r5 = INITV, r6 = INITV, r7 = INITV; \
		TYPE c = INITV; unsigned long long sm = 0; \
		(void)c; (void)sm; \
		for (int k = 0; k < iters; k++) { REP4(I0 I1 I2 I3 I4 I5 I6 I7) } \
		if (iters < 0) out[threadIdx.x] = (float)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7)SUMX + (float)sm; \
	}
#define SUMX
#define V1(op, r) asm volatile(op " %0, %0, %1" : "+v"(r) : "v"(c));
#define VSELF(op, r) asm volatile(op : "+v"(r) : "v"(c));

KERNEL(k_add, float, 1.0f, V1("v_add_f32", r0), V1("v_add_f32", r1), V1("v_add_f32", r2), V1("v_add_f32", r3), V1("v_add_f32", r4), V1("v_add_f32", r5), V1("v_add_f32", r6), V1("v_add_f32", r7))
KERNEL(k_mul, float, 1.0f, V1("v_mul_f32", r0), V1("v_mul_f32", r1), V1("v_mul_f32", r2), V1("v_mul_f32", r3), V1("v_mul_f32", r4), V1("v_mul_f32", r5), V1("v_mul_f32", r6), V1("v_mul_f32", r7))
KERNEL(k_and, float, 1.0f, V1("v_and_b32", r0), V1("v_and_b32", r1), V1("v_and_b32", r2), V1("v_and_b32", r3), V1("v_and_b32", r4), V1("v_and_b32", r5), V1("v_and_b32", r6), V1("v_and_b32", r7))
#define DPPMOV(r, s) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(s));
KERNEL(k_mov_dpp, float, 1.0f, DPPMOV(r0, c), DPPMOV(r1, c), DPPMOV(r2, c), DPPMOV(r3, c), DPPMOV(r4, c), DPPMOV(r5, c), DPPMOV(r6, c), DPPMOV(r7, c))
#define DPPADD(r, s) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(s));
KERNEL(k_add_dpp, float, 1.0f, DPPADD(r0, c), DPPADD(r1, c), DPPADD(r2, c), DPPADD(r3, c), DPPADD(r4, c), DPPADD(r5, c), DPPADD(r6, c), DPPADD(r7, c))
#define QPERM(r, s) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(s));
KERNEL(k_mov_dpp_quad, float, 1.0f, QPERM(r0, c), QPERM(r1, c), QPERM(r2, c), QPERM(r3, c), QPERM(r4, c), QPERM(r5, c), QPERM(r6, c), QPERM(r7, c))
#define BFE(r) asm volatile("v_bfe_i32 %0, %0, 1, 3" : "+v"(r));
KERNEL(k_bfe, int, 77, BFE(r0), BFE(r1), BFE(r2), BFE(r3), BFE(r4), BFE(r5), BFE(r6), BFE(r7))
#define ADD3(r) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(r) : "v"(c));
KERNEL(k_add3, int, 77, ADD3(r0), ADD3(r1), ADD3(r2), ADD3(r3), ADD3(r4), ADD3(r5), ADD3(r6), ADD3(r7))
#define CND(r) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(c) : "vcc");
KERNEL(k_cndmask, int, 77, CND(r0), CND(r1), CND(r2), CND(r3), CND(r4), CND(r5), CND(r6), CND(r7))
#define CMPV(r) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(r), "v"(c) : "vcc");
KERNEL(k_cmp_vcc, float, 1.0f, CMPV(r0), CMPV(r1), CMPV(r2), CMPV(r3), CMPV(r4), CMPV(r5), CMPV(r6), CMPV(r7))
#define CMPS(r) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(sm) : "v"(r), "v"(c));
KERNEL(k_cmp_sgpr, float, 1.0f, CMPS(r0), CMPS(r1), CMPS(r2), CMPS(r3), CMPS(r4), CMPS(r5), CMPS(r6), CMPS(r7))
// a compare and the scalar instruction that consumes its mask (the PhaseSearch neighbour move): VALU -> SALU dependency
#define CMPBAL(r) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2\n s_and_b64 %0, %0, exec" : "=s"(sm) : "v"(r), "v"(c) : "scc");
KERNEL(k_cmp_sand, float, 1.0f, CMPBAL(r0), CMPBAL(r1), CMPBAL(r2), CMPBAL(r3), CMPBAL(r4), CMPBAL(r5), CMPBAL(r6), CMPBAL(r7))
#define SXOR(r) asm volatile("s_xor_b64 %0, %0, exec" : "+s"(sm) : : "scc");
KERNEL(k_salu, float, 1.0f, SXOR(r0), SXOR(r1), SXOR(r2), SXOR(r3), SXOR(r4), SXOR(r5), SXOR(r6), SXOR(r7))
#define NOP16(r) asm volatile("s_nop 15");
KERNEL(k_nop16, float, 1.0f, NOP16(r0), NOP16(r1), NOP16(r2), NOP16(r3), NOP16(r4), NOP16(r5), NOP16(r6), NOP16(r7))
#undef SUMX
#define SUMX .x
#define PK(op, r) asm volatile(op " %0, %0, %1" : "+v"(r) : "v"(c));
KERNEL(k_pk_add, v2f, (v2f{ 1.0f, 1.0f }), PK("v_pk_add_f32", r0), PK("v_pk_add_f32", r1), PK("v_pk_add_f32", r2), PK("v_pk_add_f32", r3), PK("v_pk_add_f32", r4), PK("v_pk_add_f32", r5), PK("v_pk_add_f32", r6), PK("v_pk_add_f32", r7))
KERNEL(k_pk_mul, v2f, (v2f{ 1.0f, 1.0f }), PK("v_pk_mul_f32", r0), PK("v_pk_mul_f32", r1), PK("v_pk_mul_f32", r2), PK("v_pk_mul_f32", r3), PK("v_pk_mul_f32", r4), PK("v_pk_mul_f32", r5), PK("v_pk_mul_f32", r6), PK("v_pk_mul_f32", r7))
#undef SUMX
#define SUMX
#define SHL64(r) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(r));
KERNEL(k_shl64, unsigned long long, 77ull, SHL64(r0), SHL64(r1), SHL64(r2), SHL64(r3), SHL64(r4), SHL64(r5), SHL64(r6), SHL64(r7))
#define FMA64(r) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(r) : "v"(c));
KERNEL(k_fma64, double, 1.0, FMA64(r0), FMA64(r1), FMA64(r2), FMA64(r3), FMA64(r4), FMA64(r5), FMA64(r6), FMA64(r7))
#define BPERM(r) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(7)" : "+v"(r) : "v"(c));
KERNEL(k_bpermute, int, 4, BPERM(r0), BPERM(r1), BPERM(r2), BPERM(r3), BPERM(r4), BPERM(r5), BPERM(r6), BPERM(r7))

typedef void (*kern_t)(float*, int);
struct Entry { const char* name; kern_t k; };

static float time_kernel(kern_t k, int grid, int iters, float* d) {
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, d, iters); hipDeviceSynchronize();
	float best = 1e30f;
	for (int r = 0; r < 3; r++) {
		hipEventRecord(a); hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, d, iters); hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
	}
	return best;
}

int main() {
	float* d; hipMalloc(&d, 1 << 20);
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int simds = prop.multiProcessorCount * 4, iters = 20000;
	// clock: one wave per SIMD, 32 x s_nop 15 per iteration = 2048 cycles (+ loop overhead, ~1 %)
	const float tn = time_kernel(k_nop16, simds, iters, d);
	const double ghz = (double)iters * 32 * 64 / (tn * 1e-3) / 1e9;
	printf("%d CUs; clock from the s_nop loop: %.3f GHz (%.3f ms); hipDeviceProp clockRate %.3f GHz\n", prop.multiProcessorCount, ghz, tn, prop.clockRate * 1e-6);
	const Entry es[] = { { "v_add_f32", k_add }, { "v_mul_f32", k_mul }, { "v_and_b32", k_and }, { "v_pk_add_f32", k_pk_add }, { "v_pk_mul_f32", k_pk_mul },
		{ "v_mov_b32_dpp wave_shr:1", k_mov_dpp }, { "v_add_f32_dpp wave_shr:1", k_add_dpp }, { "v_mov_b32_dpp quad_perm", k_mov_dpp_quad },
		{ "v_bfe_i32", k_bfe }, { "v_add3_u32", k_add3 }, { "v_cndmask_b32 (vcc)", k_cndmask }, { "v_cmp_gt_f32 -> vcc", k_cmp_vcc },
		{ "v_cmp_gt_f32 -> sgpr pair", k_cmp_sgpr }, { "v_cmp + s_and_b64 (pair)", k_cmp_sand }, { "s_xor_b64", k_salu },
		{ "v_lshlrev_b64", k_shl64 }, { "v_fma_f64", k_fma64 }, { "ds_bpermute_b32", k_bpermute } };
	printf("%-28s %10s %10s %10s   (SIMD cycles per wave64 instruction; W one-wave workgroups per SIMD)\n", "class", "W=1", "W=2", "W=4");
	for (const Entry& e : es) {
		printf("%-28s", e.name);
		for (int W = 1; W <= 4; W *= 2) {
			const float t = time_kernel(e.k, simds * W, iters, d);
			printf(" %10.2f", t * 1e-3 * ghz * 1e9 / ((double)W * iters * 32));
		}
		printf("\n");
	}
	return 0;
}
