// ais-catcher_amd/csrc/kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the AIS GMSK demodulation chain.
//
// Written for 64-wide wavefronts, 160 KiB LDS per CU and HBM3E streaming; compiled with
// -ffp-contract=off: every float operation below is an individually rounded IEEE binary32 op in
// exactly the association order of the reference, so results are
// bit-identical to the reference CPU chain built with strict FP flags.
// File:line citations are relative to the reference's Source/ directory.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "kernels.h"

namespace aisk {

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
typedef float c2 __attribute__((ext_vector_type(2))); // complex sample as a native 2-vector (one VGPR pair)
typedef float c4 __attribute__((ext_vector_type(4)));
// streamed once: loads / stores with the non-temporal hint (the builtins want native vector types)
__device__ __forceinline__ float2 nt_load(const float2* p) { const c2 v = __builtin_nontemporal_load(reinterpret_cast<const c2*>(p)); return make_float2(v.x, v.y); }
__device__ __forceinline__ void nt_store(float4 v, float4* p) { __builtin_nontemporal_store(c4{ v.x, v.y, v.z, v.w }, reinterpret_cast<c4*>(p)); }

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// std::complex<float> product as the strict-FP reference evaluates it: (ac-bd, ad+bc), 4 mul + 2 add
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
	return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// glibc 2.35 hypotf == (float)sqrt((double)x*x + (double)y*y)  (sysdeps/ieee754/flt-32/e_hypotf.c);
// std::abs(std::complex<float>) in the reference resolves to it (DSP/DSP.cpp:315,434,439,449,465).
__device__ __forceinline__ float hypot_ref(float x, float y) {
	double dx = (double)x, dy = (double)y;
	return (float)__dsqrt_rn(dx * dx + dy * dy);
}

// Decimating CIC5 on a register chunk (DSP/DSP.cpp:85-117):
// v[0 .. 2*NOUT+3] = x[2*j0-5 .. 2*j0+2*NOUT-2]; out[q] = s4[2*(j0+q)] * 2^-5 with
// s_k[n] = s_{k-1}[n] + s_{k-1}[n-1].  In-place Pascal triangle; pairing identical to the reference.
template <int NOUT>
__device__ __forceinline__ void cic5_dec_chunk(float2 (&v)[2 * NOUT + 4], float2 (&out)[NOUT]) {
#pragma unroll
	for (int lvl = 0; lvl < 4; lvl++) {
#pragma unroll
		for (int i = 0; i < 2 * NOUT + 3 - lvl; i++) v[i] = cadd(v[i + 1], v[i]);
	}
#pragma unroll
	for (int q = 0; q < NOUT; q++) {
		float2 s = cadd(v[2 * q + 1], v[2 * q]);
		out[q] = make_float2(s.x * 0.03125f, s.y * 0.03125f);
	}
}

// LDS is addressed in 16-byte units (float4 = two complex samples) wherever a thread moves more than one
// sample, so that every such access is a single ds_read_b128 / ds_write_b128 (the padded row strides are
// only conflict free for the 128-bit lane grouping).
// 16-byte LDS load whose four components are all kept live, so the compiler cannot narrow it and re-pair
// the halves into ds_read2_b64 (whose 16-lane grouping conflicts on the padded rows: measured 30 % of the
// LDS cycles as bank conflicts before this)
__device__ __forceinline__ float4 lds4(const float4* p) {
	float4 v = *p;
	asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
	return v;
}
// the same for a batch: all loads are issued first, the keep-alive statements follow (an asm right behind each load
// would force a wait per load and serialise the LDS round trips)
template <int N>
__device__ __forceinline__ void lds4_batch(const float4* p, float4 (&v)[N]) {
#pragma unroll
	for (int e = 0; e < N; e++) v[e] = p[e];
#pragma unroll
	for (int e = 0; e < N; e++) asm volatile("" : "+v"(v[e].x), "+v"(v[e].y), "+v"(v[e].z), "+v"(v[e].w));
}
__device__ __forceinline__ float2 lo(float4 v) { return make_float2(v.x, v.y); }
__device__ __forceinline__ float2 hi(float4 v) { return make_float2(v.z, v.w); }
__device__ __forceinline__ float4 pack(float2 a, float2 b) { return make_float4(a.x, a.y, b.x, b.y); }

// one decimating CIC5 output from in[2t-5 .. 2t] of a contiguous level with 8 leading history slots
// (lvl4 = level base in float4 units: sample i lives at float2 index 8 + i)
__device__ __forceinline__ float2 cic5_small(const float4* lvl4, int t) {
	float2 v[6];
	float4 c[4];
	lds4_batch<4>(lvl4 + t + 1, c); // samples 2t-6 .. 2t+1
	v[0] = hi(c[0]);
	v[1] = lo(c[1]); v[2] = hi(c[1]);
	v[3] = lo(c[2]); v[4] = hi(c[2]);
	v[5] = lo(c[3]);
	float2 o[1];
	cic5_dec_chunk<1>(v, o);
	return o[0];
}

// ------------------------------------------------------------------------------------------
// K1 (register variant): the same ladder with the CIC5 stages, the droop filter and the rotator entirely in
// registers.  One wave = one autonomous stream processor; lane l owns 2^K consecutive input samples of the
// wave-tile (64 * 2^K samples) and ends up with exactly one 96 kHz sample.  The five samples of history every
// stage needs come from the neighbouring lane through DPP wave shifts (lane 0 receives lane 63's value of the
// PREVIOUS tile, kept in a shadow register), so between the coalesced-load transposition and the 96 kHz point
// there is no LDS traffic and no synchronisation at all; only the two channel filters behind the rotator use a
// small wave-private LDS ring.  Arithmetic and pairing are identical to the LDS variant (cic5_dec_chunk).
// ------------------------------------------------------------------------------------------
// complex samples as native 2-vectors: an add is one v_pk_add_f32 on an aligned register pair and (re, im) stay
// together (with the float2 struct the SLP vectoriser re-pairs components of different samples and pays for it in moves)

__device__ __forceinline__ float dpp_wave_shr1(float old_lane0, float src) { // lane l <- src[l-1]; lane 0 keeps old_lane0
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old_lane0), __float_as_int(src), 0x138, 0xF, 0xF, false));
}
// lanes 0..3 <- src[(l + 63) % 64], the other lanes keep `keep` (row_mask 0x1, bank_mask 0x1): only lane 0 matters
__device__ __forceinline__ float dpp_carry_lane63(float keep, float src) {
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(src), 0x13C, 0x1, 0x1, false));
}
// value of the previous lane; for lane 0 the value lane 63 held one tile earlier.  `prev` is a shadow register whose
// lane 0 holds that carried value.  Both DPP moves work IN PLACE on the shadow register (the shift's `old` operand
// and, after the caller's last use of the result, the carry for the next tile), so the loop-carried shadow never
// needs a copy: call from_prev_lane(), use the result, then call carry_to_next_tile() with the same registers.
__device__ __forceinline__ c2 from_prev_lane(c2 cur, c2 prev) {
	c2 r;
	r.x = dpp_wave_shr1(prev.x, cur.x);
	r.y = dpp_wave_shr1(prev.y, cur.y);
	return r;
}
__device__ __forceinline__ c2 carry_to_next_tile(c2 halo, c2 cur) {
	c2 p;
	p.x = dpp_carry_lane63(halo.x, cur.x);
	p.y = dpp_carry_lane63(halo.y, cur.y);
	return p;
}
// the same two moves for a sample packed into one register (fixed-point ladder: I in bits 0..15, Q in bits 16..31)
__device__ __forceinline__ unsigned from_prev_lane(unsigned cur, unsigned prev) {
	return (unsigned)__builtin_amdgcn_update_dpp((int)prev, (int)cur, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ unsigned carry_to_next_tile(unsigned halo, unsigned cur) {
	return (unsigned)__builtin_amdgcn_update_dpp((int)halo, (int)cur, 0x13C, 0x1, 0x1, false);
}

// cic5_dec_chunk on native vectors (same pairing, same rounding)
// T = c2: float stage (DSP.cpp:93-117).  T = unsigned: one stage of the fixed-point ladder DS_UINT16::Run (DSP.cpp:499-522):
// I and Q are 16-bit fields of one word, the five cascaded sums are plain 32-bit additions (no field ever overflows: 8-bit
// input, gain 32 per stage, SHIFT bits dropped per stage), the stage output is (z >> SHIFT) & mask.  Integer sums do not
// care about the order, so the same pairing as the float stage is used.
template <int NOUT, typename T, int SHIFT>
__device__ __forceinline__ void cic5_dec_chunk_v(T (&v)[2 * NOUT + 4], T (&out)[NOUT]) {
#pragma unroll
	for (int lvl = 0; lvl < 4; lvl++) {
#pragma unroll
		for (int i = 0; i < 2 * NOUT + 3 - lvl; i++) v[i] = v[i + 1] + v[i];
	}
#pragma unroll
	for (int q = 0; q < NOUT; q++) {
		if constexpr (sizeof(T) == sizeof(c2)) out[q] = (v[2 * q + 1] + v[2 * q]) * 0.03125f;
		else out[q] = ((v[2 * q + 1] + v[2 * q]) >> SHIFT) & ((0xFFFFu >> SHIFT) * 0x10001u);
	}
}

template <int C, typename T> struct HaloState;           // shadow registers of one stage (all zero = silence before the stream)
template <typename T> struct HaloState<64, T> { T p[5]; };
template <typename T> struct HaloState<32, T> { T p[5]; };
template <typename T> struct HaloState<16, T> { T p[5]; };
template <typename T> struct HaloState<8, T> { T p[5]; };
template <typename T> struct HaloState<4, T> { T p[4], q; };
template <typename T> struct HaloState<2, T> { T p1[2], p2[2], p3; };

// h[i] = sample (chunk_start - 5 + i) of the stage's input stream
template <int C, typename T>
__device__ __forceinline__ void get_halo(const T (&x)[C], const HaloState<C, T>& hs, T (&h)[5]) {
	if constexpr (C >= 5) {
#pragma unroll
		for (int i = 0; i < 5; i++) h[i] = from_prev_lane(x[C - 5 + i], hs.p[i]);
	} else if constexpr (C == 4) {
#pragma unroll
		for (int j = 0; j < 4; j++) h[1 + j] = from_prev_lane(x[j], hs.p[j]);
		h[0] = from_prev_lane(h[4], hs.q); // two lanes back
	} else { // C == 2
#pragma unroll
		for (int j = 0; j < 2; j++) h[3 + j] = from_prev_lane(x[j], hs.p1[j]);
#pragma unroll
		for (int j = 0; j < 2; j++) h[1 + j] = from_prev_lane(h[3 + j], hs.p2[j]);
		h[0] = from_prev_lane(h[2], hs.p3); // three lanes back
	}
}
// after the halo values have been consumed: their registers become the shadows of the next tile
template <int C, typename T>
__device__ __forceinline__ void put_halo(const T (&x)[C], HaloState<C, T>& hs, const T (&h)[5]) {
	if constexpr (C >= 5) {
#pragma unroll
		for (int i = 0; i < 5; i++) hs.p[i] = carry_to_next_tile(h[i], x[C - 5 + i]);
	} else if constexpr (C == 4) {
		hs.q = carry_to_next_tile(h[0], h[4]);
#pragma unroll
		for (int j = 0; j < 4; j++) hs.p[j] = carry_to_next_tile(h[1 + j], x[j]);
	} else {
		hs.p3 = carry_to_next_tile(h[0], h[2]);
#pragma unroll
		for (int j = 0; j < 2; j++) hs.p2[j] = carry_to_next_tile(h[1 + j], h[3 + j]);
#pragma unroll
		for (int j = 0; j < 2; j++) hs.p1[j] = carry_to_next_tile(h[3 + j], x[j]);
	}
}

template <int SHIFT, int C, typename T>
__device__ __forceinline__ void reg_stage(const T (&x)[C], HaloState<C, T>& hs, T (&out)[C / 2]) {
	T h[5];
#ifdef ABL_NO_HALO // (ablation builds only: wrong results -- no halo exchange at all)
#pragma unroll
	for (int i = 0; i < 5; i++) h[i] = x[i % C];
#else
	get_halo<C, T>(x, hs, h);
#endif
	T v[C + 4];
#pragma unroll
	for (int i = 0; i < 5; i++) v[i] = h[i];
#pragma unroll
	for (int i = 0; i < C - 1; i++) v[5 + i] = x[i];
	cic5_dec_chunk_v<C / 2, T, SHIFT>(v, out);
#if !defined(ABL_NO_PUT) && !defined(ABL_NO_HALO) // (ablation builds only: wrong results, measures what the carry moves cost)
	put_halo<C, T>(x, hs, h);
#endif
}
template <int C>
__device__ __forceinline__ void reg_stage(const c2 (&x)[C], HaloState<C, c2>& hs, c2 (&out)[C / 2]) { reg_stage<0, C, c2>(x, hs, out); }

// ------------------------------------------------------------------------------------------
// Round 6: the halo exchange as ONE neighbour value per Pascal level ("level carry") instead of five input samples per stage.
// s_k[n] = s_{k-1}[n] + s_{k-1}[n-1]: for the first sample of a lane's chunk the second operand is the LAST element of the previous
// lane's level k-1 -- the neighbour has computed it anyway -- so the lane needs one value per level (five per stage: x[C-1], L1[C-1] ..
// L4[C-1]) and adds it with the shift fused into the addition (v_add_f32_dpp wave_shr:1), instead of fetching five raw samples (ten DPP
// moves) and re-adding the neighbour's triangle over them (4 C + 6 + C/2 additions per stage then, 4 C + C/2 now: the floor of this
// pairing).  Every sum is the same pair of operands as before (float addition commutes), so the results are the same bits.
// Lane 0's neighbour is lane 63 of the PREVIOUS tile: lane 63 leaves its five level tails of every stage in 40 bytes of LDS per
// stage (a one-lane write), every lane reads them back at the next tile (a broadcast read; only lane 0's copy matters) and forms
// `own + carry` with a packed addition, which the fused DPP additions then overwrite in lanes 1..63 -- no carry moves through the
// DPP network (rounds 1-5: ten per stage and tile).  Tile loop of the four-stage ladder: 224 packed + 88 DPP -> 200 + 48 instructions.
// Used by the integer input formats only (see k1_dpp: LC).
// (`s_nop 1` in front of the DPP pair: a VGPR written by the preceding VALU instruction may not be read through DPP for two wait
// states, and the compiler's hazard recogniser does not look into inline assembly.)
// ------------------------------------------------------------------------------------------
#ifndef K1_LEVEL_CARRY
#define K1_LEVEL_CARRY 1
#endif
__device__ __forceinline__ void wave_sync();
__device__ __forceinline__ c2 nb_add(c2 own, c2 tail, c2 carry) { // lanes 1..63: tail[lane - 1] + own; lane 0: carry + own
	const c2 r = own + carry;
	float rx = r.x, ry = r.y;
	asm("s_nop 1\n\tv_add_f32_dpp %0, %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %3, %5 wave_shr:1 row_mask:0xf bank_mask:0xf"
	    : "+v"(rx), "+v"(ry) : "v"(tail.x), "v"(tail.y), "v"(own.x), "v"(own.y));
	return c2{ rx, ry };
}
__device__ __forceinline__ unsigned nb_add(unsigned own, unsigned tail, unsigned carry) {
	unsigned r = own + carry;
	asm("s_nop 1\n\tv_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(tail), "v"(own));
	return r;
}
template <int C, typename T>
__device__ __forceinline__ void level_lc(const T (&in)[C], T carry, T (&out)[C]) {
	out[0] = nb_add(in[0], in[C - 1], carry);
#pragma unroll
	for (int i = 1; i < C; i++) out[i] = in[i] + in[i - 1];
}
// one decimating CIC5 stage: x = the lane's C consecutive inputs, carry[k] = lane 63's tail of level k of the previous tile (valid in
// lane 0), tails[k] = this lane's (k = 0: the last input, k = 1..4: the last element of level k); out = C / 2 outputs
template <int SHIFT, int C, typename T>
__device__ __forceinline__ void reg_stage_lc(const T (&x)[C], const T (&carry)[5], T (&tails)[5], T (&out)[C / 2]) {
	T a[C], b[C], c[C], d[C];
	level_lc<C, T>(x, carry[0], a);
	level_lc<C, T>(a, carry[1], b);
	level_lc<C, T>(b, carry[2], c);
	level_lc<C, T>(c, carry[3], d);
	tails[0] = x[C - 1]; tails[1] = a[C - 1]; tails[2] = b[C - 1]; tails[3] = c[C - 1]; tails[4] = d[C - 1];
	T e[C / 2];
	e[0] = nb_add(d[0], d[C - 1], carry[4]);
#pragma unroll
	for (int q = 1; q < C / 2; q++) e[q] = d[2 * q] + d[2 * q - 1];
#pragma unroll
	for (int q = 0; q < C / 2; q++) {
		if constexpr (sizeof(T) == sizeof(c2)) out[q] = e[q] * 0.03125f;
		else out[q] = (e[q] >> SHIFT) & ((0xFFFFu >> SHIFT) * 0x10001u);
	}
}
// The carry traffic of a whole ladder in two batches: every lane reads the 5 K slots at the top of the tile (one round trip for all
// stages: a read per stage in front of its stage cost the wave four LDS latencies per tile and was slower than the moves it saved),
// lane 63 writes its tails behind the last stage, and ONE wavefront-scope fence follows.
// Lane 63 writes, every lane reads at the next tile: without that fence this is a data race to the compiler, which then forwards a
// lane's OWN (never executed) store and keeps the carries of lanes 0..62 in registers for ever -- seen with the pre-decimation
// passes, whose loop has no other fence.  Wavefront scope: ordering only, the LDS unit executes a wave's instructions in order.
template <int NS, typename T, typename LT>
__device__ __forceinline__ void lc_load(const LT* lc, T (&carry)[NS][5]) {
#pragma unroll
	for (int s = 0; s < NS; s++)
#pragma unroll
		for (int k = 0; k < 5; k++) {
			if constexpr (sizeof(T) == sizeof(c2)) { const float2 v = lc[5 * s + k]; carry[s][k] = c2{ v.x, v.y }; }
			else carry[s][k] = lc[5 * s + k];
		}
}
template <int NS, typename T, typename LT>
__device__ __forceinline__ void lc_store(LT* lc, const T (&tails)[NS][5], int lane) {
	if (lane == 63) {
#pragma unroll
		for (int s = 0; s < NS; s++)
#pragma unroll
			for (int k = 0; k < 5; k++) {
				if constexpr (sizeof(T) == sizeof(c2)) lc[5 * s + k] = make_float2(tails[s][k].x, tails[s][k].y);
				else lc[5 * s + k] = tails[s][k];
			}
	}
	wave_sync();
}
template <int K>
__device__ __forceinline__ c2 run_ladder_lc(const c2 (&x)[1 << K], float2* lc, int lane) {
	c2 cr[K][5], tl[K][5];
	lc_load<K, c2>(lc, cr);
	c2 r;
	if constexpr (K == 6) {
		c2 y[32], z[16], a[8], b[4], c[2], d[1];
		reg_stage_lc<0, 64>(x, cr[0], tl[0], y); reg_stage_lc<0, 32>(y, cr[1], tl[1], z); reg_stage_lc<0, 16>(z, cr[2], tl[2], a);
		reg_stage_lc<0, 8>(a, cr[3], tl[3], b); reg_stage_lc<0, 4>(b, cr[4], tl[4], c); reg_stage_lc<0, 2>(c, cr[5], tl[5], d);
		r = d[0];
	} else if constexpr (K == 5) {
		c2 z[16], a[8], b[4], c[2], d[1];
		reg_stage_lc<0, 32>(x, cr[0], tl[0], z); reg_stage_lc<0, 16>(z, cr[1], tl[1], a); reg_stage_lc<0, 8>(a, cr[2], tl[2], b);
		reg_stage_lc<0, 4>(b, cr[3], tl[3], c); reg_stage_lc<0, 2>(c, cr[4], tl[4], d);
		r = d[0];
	} else if constexpr (K == 4) {
		c2 a[8], b[4], c[2], d[1];
		reg_stage_lc<0, 16>(x, cr[0], tl[0], a); reg_stage_lc<0, 8>(a, cr[1], tl[1], b); reg_stage_lc<0, 4>(b, cr[2], tl[2], c); reg_stage_lc<0, 2>(c, cr[3], tl[3], d);
		r = d[0];
	} else if constexpr (K == 3) {
		c2 b[4], c[2], d[1];
		reg_stage_lc<0, 8>(x, cr[0], tl[0], b); reg_stage_lc<0, 4>(b, cr[1], tl[1], c); reg_stage_lc<0, 2>(c, cr[2], tl[2], d);
		r = d[0];
	} else if constexpr (K == 2) {
		c2 c[2], d[1];
		reg_stage_lc<0, 4>(x, cr[0], tl[0], c); reg_stage_lc<0, 2>(c, cr[1], tl[1], d);
		r = d[0];
	} else {
		c2 d[1];
		reg_stage_lc<0, 2>(x, cr[0], tl[0], d);
		r = d[0];
	}
	lc_store<K, c2>(lc, tl, lane);
	return r;
}
// Downsample16_CU8 (see run_fix_ladder) with the level carry: packed 16-bit fields, integer additions
__device__ __forceinline__ c2 run_fix_ladder_lc(const unsigned (&x)[16], unsigned* lc, int lane) {
	unsigned cr[4][5], tl[4][5];
	lc_load<4, unsigned>(lc, cr);
	unsigned a[8], b[4], c[2], d[1];
	reg_stage_lc<3, 16, unsigned>(x, cr[0], tl[0], a); reg_stage_lc<4, 8, unsigned>(a, cr[1], tl[1], b);
	reg_stage_lc<5, 4, unsigned>(b, cr[2], tl[2], c); reg_stage_lc<0, 2, unsigned>(c, cr[3], tl[3], d);
	lc_store<4, unsigned>(lc, tl, lane);
	const unsigned z = d[0] ^ 0x80008000u;
	return c2{ (float)(int)(short)(z & 0xffffu) * 0.000030517578125f, (float)(int)(short)(z >> 16) * 0.000030517578125f };
}

template <int K> struct RegLadder;
template <> struct RegLadder<6> { HaloState<64, c2> s64; HaloState<32, c2> s32; HaloState<16, c2> s16; HaloState<8, c2> s8; HaloState<4, c2> s4; HaloState<2, c2> s2; };
template <> struct RegLadder<5> { HaloState<32, c2> s32; HaloState<16, c2> s16; HaloState<8, c2> s8; HaloState<4, c2> s4; HaloState<2, c2> s2; };
template <> struct RegLadder<4> { HaloState<16, c2> s16; HaloState<8, c2> s8; HaloState<4, c2> s4; HaloState<2, c2> s2; };
template <> struct RegLadder<3> { HaloState<8, c2> s8; HaloState<4, c2> s4; HaloState<2, c2> s2; };
template <> struct RegLadder<2> { HaloState<4, c2> s4; HaloState<2, c2> s2; };
template <> struct RegLadder<1> { HaloState<2, c2> s2; };

// Downsample16_CU8 (DSP.cpp:639-651, `-go FP_DS on` at 1536 kSPS): four DS_UINT16 stages with shifts 3, 4, 5, 0, then
// uint16 -> int16 by flipping the sign bits and / 32768.0f (DSP.cpp:587-607)
struct FixLadder { HaloState<16, unsigned> s16; HaloState<8, unsigned> s8; HaloState<4, unsigned> s4; HaloState<2, unsigned> s2; };
__device__ __forceinline__ c2 run_fix_ladder(const unsigned (&x)[16], FixLadder& st) {
	unsigned a[8], b[4], c[2], d[1];
	reg_stage<3, 16, unsigned>(x, st.s16, a); reg_stage<4, 8, unsigned>(a, st.s8, b);
	reg_stage<5, 4, unsigned>(b, st.s4, c); reg_stage<0, 2, unsigned>(c, st.s2, d);
	const unsigned z = d[0] ^ 0x80008000u;
	return c2{ (float)(int)(short)(z & 0xffffu) * 0.000030517578125f, (float)(int)(short)(z >> 16) * 0.000030517578125f };
}

template <int K>
__device__ __forceinline__ c2 run_ladder(const c2 (&x)[1 << K], RegLadder<K>& st) {
	if constexpr (K == 6) { // 6144 kSPS in one pass: 64 samples per lane, one wave per SIMD
		c2 y[32], z[16], a[8], b[4], c[2], d[1];
		reg_stage<64>(x, st.s64, y); reg_stage<32>(y, st.s32, z); reg_stage<16>(z, st.s16, a); reg_stage<8>(a, st.s8, b); reg_stage<4>(b, st.s4, c); reg_stage<2>(c, st.s2, d);
		return d[0];
	} else if constexpr (K == 5) { // 3072 kSPS in one pass (round 5, late): 32 samples per lane
		c2 z[16], a[8], b[4], c[2], d[1];
		reg_stage<32>(x, st.s32, z); reg_stage<16>(z, st.s16, a); reg_stage<8>(a, st.s8, b); reg_stage<4>(b, st.s4, c); reg_stage<2>(c, st.s2, d);
		return d[0];
	} else if constexpr (K == 4) {
		c2 a[8], b[4], c[2], d[1];
		reg_stage<16>(x, st.s16, a); reg_stage<8>(a, st.s8, b); reg_stage<4>(b, st.s4, c); reg_stage<2>(c, st.s2, d);
		return d[0];
	} else if constexpr (K == 3) {
		c2 b[4], c[2], d[1];
		reg_stage<8>(x, st.s8, b); reg_stage<4>(b, st.s4, c); reg_stage<2>(c, st.s2, d);
		return d[0];
	} else if constexpr (K == 2) {
		c2 c[2], d[1];
		reg_stage<4>(x, st.s4, c); reg_stage<2>(c, st.s2, d);
		return d[0];
	} else {
		c2 d[1];
		reg_stage<2>(x, st.s2, d);
		return d[0];
	}
}

// A workgroup of ONE wave needs no s_barrier and, above all, no "s_waitcnt vmcnt(0)" (which __syncthreads() implies
// and which would drain the next tile's prefetch in the middle of this tile's arithmetic): the LDS unit executes a
// wave's DS instructions in order, so a wavefront-scope fence (compiler ordering only) is all that is needed.
__device__ __forceinline__ void wave_sync() {
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

// issue priorities (measured, 0.60 vs 0.65 ms per step): the front end must keep its loads flowing (2); raising
// PhaseSearchEMA above 0 only takes issue slots from it
#ifndef K1_PRIO
#define K1_PRIO 2
#endif
// cache policy of the input stream (aux operand of global_load_lds: 0 default, 2 = nt: read once, do not keep).  nt: the kernel
// alone 3 % faster, the step 0-3 % depending on the box (profiles/r02_expF.txt, r02_expG.txt); the SAME hint on the
// intermediates (FIR outputs, 48 kHz channels) costs 5-10 %: their consumers do find them in L2 / Infinity Cache
#ifndef K1_LOAD_AUX
#define K1_LOAD_AUX 2
#endif
// Register budget: three front-end waves per SIMD must leave room for one PhaseSearchEMA wave (96 VGPRs) in the
// 512-entry file, or the two kernels evict each other instead of overlapping (HBM-bound next to VALU-bound).
#ifndef K1_WAVES
#define K1_WAVES 3
#endif
// FMT: input sample format (Utilities/StreamHelpers.cpp:51-133): 0 = CF32, 1 = CU8, 2 = CS8, 3 = CS16;
// 4 = CU8 through the fixed-point ladder Downsample16_CU8 (K = 4 only)
constexpr int fmt_bytes(int fmt) { return fmt == 0 || fmt == 5 ? 8 : fmt == 3 ? 4 : 2; } // (5: Upsample outputs computed in the wave, see K1Params::us_idx)

__device__ __forceinline__ void k1_fft_tail(const K1Params& p, int rx, int span, float2* X); // below, with the FFT
template <class P> __device__ __forceinline__ void wave_fft_tail(const P& q, size_t row, bool two_rows, int w0, int nw, float2* X); // (the same for k1x_wave / k1k_wave)

template <int E> struct K1Const { static constexpr int value = E; };
template <int E, int N, class F>
__device__ __forceinline__ void k1_static_for(F&& f) {
	if constexpr (E < N) { f(K1Const<E>{}); k1_static_for<E + 1, N>(f); }
}

template <int K, int FMT, bool PRE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(K == 6 ? 1 : K == 5 ? 2 : K1_WAVES, 4))) void k1_dpp(K1Params p) { // (K = 5: 32 samples per lane, two waves per SIMD with 16 KB tiles; K = 6: 64, one wave with 32 KB tiles)
	constexpr int C0 = 1 << K;        // input samples per lane per tile
	constexpr int TILE_IN = 64 * C0;  // input samples per wave-tile
	constexpr bool DMA = C0 >= 4 && FMT == 0; // tiles come straight from HBM into LDS (global_load_lds), no staging registers
	constexpr int W4 = C0 / 2;            // 16-byte pieces per lane
	__shared__ __attribute__((aligned(16))) float4 xt[DMA ? 64 * W4 : 1]; // the tile, linear, XOR-swizzled in units of 16 B
	__shared__ __attribute__((aligned(16))) float2 x5[2][8 + 64];  // rotated up/down with 8 samples of history
	__shared__ __attribute__((aligned(16))) float2 x6[2][8 + 32];  // DS2_a/b output
	constexpr bool XFFT_IN_XT = DMA && K >= 4; // the FFT tail's exchange / search buffer (8 KB) reuses the tile buffer where that is big enough
	__shared__ __attribute__((aligned(16))) float2 xfft[(PRE || XFFT_IN_XT) ? 1 : 1024];
	// Level carry (run_ladder_lc) where the kernel is bound by its instructions -- the integer input formats, which are converted in
	// the lanes and read a quarter or half of the bytes: -3 % per step with CU8 input, +2 % GS/s with FP_DS.  The CF32 forms keep the
	// five-sample halos: they are paced by their tile stream, and what the level carry saves in instructions (346 -> 283 per tile)
	// it costs them in latency per tile (an LDS round trip for the carries, wait states in front of the fused DPP additions):
	// +0.4 ... 0.9 % per step, front end alone +3 % (profiles/r06_expF_level_carry.txt).
	constexpr bool LC = K1_LEVEL_CARRY != 0 && FMT >= 1 && FMT <= 4;
	__shared__ __attribute__((aligned(16))) float2 lc_f[LC ? 5 * K + 2 : 1]; // level carries: lane 63's five tails per stage
	__shared__ unsigned lc_u[LC && FMT == 4 ? 20 : 1];
	const int lane = threadIdx.x;
	const int rx = blockIdx.y;
	const int span = blockIdx.x;
#ifndef K1US_PRIO
#define K1US_PRIO K1_PRIO
#endif
	__builtin_amdgcn_s_setprio(FMT == 5 ? K1US_PRIO : K1_PRIO);
	// Occupancy cap: ~120 VGPRs and ~10 KB of LDS let FOUR of these one-wave workgroups share a SIMD, and sixteen of them hold
	// 156 of a CU's 160 KB of LDS -- nothing that needs LDS (the FFT, the staged PhaseSearch) gets on the CU beside them.  The
	// kernel is as fast with three (HBM-bound); naming v135 as clobbered makes its allocation 136 registers = three per SIMD.
#ifdef K1US_NO_CLOBBER
	if constexpr (FMT != 5)
#endif
	asm volatile("" ::: "v135");

	if (lane < 8) { x5[0][lane] = x5[1][lane] = x6[0][lane] = x6[1][lane] = make_float2(0.f, 0.f); }
	if (LC) { // silence before the span's warm-up tile, as the zeroed shadow registers were
		if (lane < 5 * K) lc_f[lane] = make_float2(0.f, 0.f);
		if (FMT == 4 && lane < 20) lc_u[lane] = 0u;
	}
	RegLadder<K> st = {};
	FixLadder fst = {};
	c2 fdc_p1 = { 0.f, 0.f }, fdc_p2 = { 0.f, 0.f };

	const int tile_first = span * p.tiles_per_span - 1; // warm-up tile
	int tile_last = tile_first + p.tiles_per_span;
	if (tile_last >= p.tiles_per_block) tile_last = p.tiles_per_block - 1;

	constexpr int TILE_BYTES = TILE_IN * fmt_bytes(FMT);
	constexpr int LANE_BYTES = TILE_BYTES / 64;                    // 2^K * (2 or 8)
	constexpr int NV = LANE_BYTES >= 16 ? LANE_BYTES / 16 : 1;      // 16-byte pieces per lane
	uint4 pre[DMA ? 1 : NV];
	// DMA path.  Piece q (16 B) of lane r's row lives in LDS slot r*W4 + (q ^ (r % W4)): with the XOR the 8 lanes a
	// ds_read_b128 serves per clock hit 8 different bank groups.  global_load_lds writes lane l of instruction e to slot
	// 64*e + l, so that lane FETCHES the piece that belongs there (the swizzle is applied to the global address).
	const int dma_r = lane / W4, dma_q = (lane % W4) ^ ((lane / W4) % W4); // slot 64*e + lane -> row 64/W4*e + dma_r, piece dma_q
	// The warm-up tile only has to fill the filters: nothing a later tile puts out depends on more than the last 603 input samples
	// (dependency cone of the whole ladder), so of its 64 lane rows only the last 64 - WARM_SKIP_ROWS are fetched; the first rows
	// keep whatever the tile buffer held -- the values computed from them are finite-window sums that never reach a stored output.
	constexpr int WARM_SKIP_E = (DMA && C0 == 16) ? 3 : (DMA && C0 == 32) ? 6 : (DMA && C0 == 64) ? 12 : 0; // 3 of the 8 load instructions = rows 0 .. 23 = 384 of 1024 samples (K = 5: the cone is 2 x 603 + 5 samples; 6 of 16 instructions = rows 0 .. 23 = 768 of 2048; K = 6: 2,427 samples; 12 of 32 = rows 0 .. 23 = 1,536 of 4,096)
	auto prefetch = [&](int tile) {
		const unsigned char* base;
		if (tile < 0) base = (const unsigned char*)p.hist + (size_t)rx * TILE_BYTES;
		else base = (const unsigned char*)p.in + ((size_t)rx * p.in_stride + (size_t)tile * TILE_IN) * fmt_bytes(FMT);
		if constexpr (DMA) {
			const bool warm = WARM_SKIP_E > 0 && tile == tile_first; // wave-uniform
			if constexpr (W4 <= 8) { // an instruction covers a multiple of W4 rows: the row's swizzle term is the same for every e
				const uint4* src = (const uint4*)base + dma_r * W4 + dma_q;
#pragma unroll
				for (int e = 0; e < NV; e++)
					if (e >= WARM_SKIP_E || !warm)
						__builtin_amdgcn_global_load_lds((const void*)(src + e * 64), (__attribute__((address_space(3))) void*)(xt + e * 64), 16, 0, K1_LOAD_AUX);
			} else { // W4 = 16 (K = 5): four rows per instruction, row = 4 e + dma_r, so its swizzle term is dma_r ^ 4 (e % 4) (disjoint bits)
				const uint4* src = (const uint4*)base + dma_r * W4;
#pragma unroll
				for (int e = 0; e < NV; e++)
					if (e >= WARM_SKIP_E || !warm)
						__builtin_amdgcn_global_load_lds((const void*)(src + e * 64 + (dma_q ^ ((64 / W4 * e) % W4))), (__attribute__((address_space(3))) void*)(xt + e * 64), 16, 0, K1_LOAD_AUX);
			}
		} else if constexpr (LANE_BYTES >= 16) {
			const uint4* src = (const uint4*)base;
#pragma unroll
			for (int e = 0; e < NV; e++) pre[e] = src[lane * NV + e]; // the lane's own contiguous bytes
		} else if constexpr (LANE_BYTES == 8) {
			const uint2 v = ((const uint2*)base)[lane];
			pre[0] = make_uint4(v.x, v.y, 0, 0);
		} else {
			pre[0] = make_uint4(((const unsigned*)base)[lane], 0, 0, 0);
		}
	};
	// FMT 5: the tile's samples are Upsample outputs n = 4 C0' ... computed from the pre-decimated stream.  Two loads deep: the table
	// entries (input index b, alpha) of a lane's C0 outputs travel two tiles ahead, the input samples they point at one tile ahead.
	// Output n of the flush lives at table index US_HIST + n; the warm-up tile of span 0 reaches back to n = -64 C0, the tables to
	// n = -US_HIST, which covers the dependency cone of everything behind (83 + 8 samples, like k1u_resample_frontend's halo): the lanes
	// in front of that produce zeros that never reach a stored output.
	static_assert(FMT != 5 || (K == 2 && !PRE), "k1_dpp: the resampled form is the two-stage tail of the ladder");
	struct UsTab { int ib[FMT == 5 ? C0 : 1]; float al[FMT == 5 ? C0 : 1]; };
	struct UsXs { float2 a[FMT == 5 ? C0 : 1], b[FMT == 5 ? C0 : 1]; float al[FMT == 5 ? C0 : 1]; bool dead; };
	UsTab us_tab = {};
	UsXs us_xs = {};
	XRow us_row = {};
	if constexpr (FMT == 5) us_row = make_xrow(p, rx);
	const auto us_tab_fetch = [&](int tile, UsTab& t) {
		if constexpr (FMT == 5) {
			int n0 = tile * TILE_IN + lane * C0;
			n0 = n0 < -US_HIST ? -US_HIST : n0;
			const int4 ib = *reinterpret_cast<const int4*>(p.us_idx + US_HIST + n0);
			const float4 al = *reinterpret_cast<const float4*>(p.us_alpha + US_HIST + n0);
			t.ib[0] = ib.x; t.ib[1] = ib.y; t.ib[2] = ib.z; t.ib[3] = ib.w;
			t.al[0] = al.x; t.al[1] = al.y; t.al[2] = al.z; t.al[3] = al.w;
		}
	};
	const auto us_xs_fetch = [&](int tile, const UsTab& t, UsXs& q) {
		if constexpr (FMT == 5) {
			q.dead = tile * TILE_IN + lane * C0 < -US_HIST;
			// the input samples of a whole tile nearly always lie in ONE of the ring's blocks: a wave-uniform base then (us_idx is non-decreasing)
			const int lo = __builtin_amdgcn_readfirstlane(t.ib[0]) - 1, hi = __builtin_amdgcn_readlane(t.ib[C0 - 1], 63);
			const XSpan x(us_row, lo, hi);
			if (!x.mixed) { // (wave-uniform)
#pragma unroll
				for (int i = 0; i < C0; i++) { q.a[i] = x.base[t.ib[i] - 1]; q.b[i] = x.base[t.ib[i]]; }
			} else {
#pragma unroll
				for (int i = 0; i < C0; i++) { q.a[i] = us_row[t.ib[i] - 1]; q.b[i] = us_row[t.ib[i]]; }
			}
#pragma unroll
			for (int i = 0; i < C0; i++) q.al[i] = t.al[i];
		}
	};
	if constexpr (FMT == 5) {
		static_assert(FMT != 5 || C0 == 4, "k1_dpp: one int4 / float4 of table entries per lane");
		UsTab t0 = {};
		us_tab_fetch(tile_first, t0);
		us_xs_fetch(tile_first, t0, us_xs);
		us_tab_fetch(tile_first + 1 <= tile_last ? tile_first + 1 : tile_last, us_tab);
	} else prefetch(tile_first);
	float2 rot_next = make_float2(1.0f, 0.0f);
	if (!PRE) rot_next = p.rot[(size_t)ROT_HIST + (long long)tile_first * 64 + lane];

	for (int tile = tile_first; tile <= tile_last; tile++) {
		c2 x[C0];
		unsigned xi[FMT == 4 ? C0 : 1];
		if constexpr (FMT == 4) { // z = I | Q << 16 (DSP.cpp:532-533)
			const unsigned* w = reinterpret_cast<const unsigned*>(pre);
#pragma unroll
			for (int i = 0; i < C0; i++) {
				const unsigned v = w[i >> 1] >> ((i & 1) * 16);
				xi[i] = (v & 255u) | ((v & 0xff00u) << 8);
			}
		} else if constexpr (FMT == 1 || FMT == 2) { // Utilities/Convert.cpp:255-275: ((int)u - 128) / 128.0f, (int8) / 128.0f (exact)
			const unsigned* w = reinterpret_cast<const unsigned*>(pre);
#pragma unroll
			for (int i = 0; i < C0; i++) {
				const unsigned v = w[i >> 1] >> ((i & 1) * 16);
				const int re = FMT == 1 ? (int)(v & 255u) - 128 : (int)(signed char)(v & 255u);
				const int im = FMT == 1 ? (int)((v >> 8) & 255u) - 128 : (int)(signed char)((v >> 8) & 255u);
				x[i] = c2{ (float)re * 0.0078125f, (float)im * 0.0078125f };
			}
		} else if constexpr (FMT == 5) { // Upsample (DSP.cpp:192-212): (1 - alpha) * a + alpha * b, products and sum rounded separately (as k1u_resample_frontend)
#pragma unroll
			for (int i = 0; i < C0; i++) {
				const float al = us_xs.al[i], w0 = 1 - al;
				const float2 a = us_xs.a[i], b = us_xs.b[i];
				x[i] = us_xs.dead ? c2{ 0.0f, 0.0f } : c2{ w0 * a.x + al * b.x, w0 * a.y + al * b.y };
			}
		} else if constexpr (FMT == 3) { // Utilities/Convert.cpp:277-286: (int16) / 32768.0f (exact)
			const unsigned* w = reinterpret_cast<const unsigned*>(pre);
#pragma unroll
			for (int i = 0; i < C0; i++)
				x[i] = c2{ (float)(int)(short)(w[i] & 0xffffu) * 0.000030517578125f, (float)(int)(short)(w[i] >> 16) * 0.000030517578125f };
		} else if constexpr (C0 < 4) {
			x[0] = c2{ __uint_as_float(pre[0].x), __uint_as_float(pre[0].y) };
			x[1] = c2{ __uint_as_float(pre[0].z), __uint_as_float(pre[0].w) };
		} else {
#pragma unroll
			for (int e = 0; e < W4; e++) { // every component is used, so these stay 128-bit loads
				const float4 v = xt[lane * W4 + (e ^ (lane % W4))];
				x[2 * e] = c2{ v.x, v.y }; x[2 * e + 1] = c2{ v.z, v.w };
			}
			wave_sync(); // the tile is in registers: the next one may land in xt
			if (p.hist_out && tile == p.tiles_per_block - 1) { // the block's last tile is the next block's warm-up tile
				float4* ho = reinterpret_cast<float4*>((unsigned char*)p.hist_out + (size_t)rx * TILE_BYTES) + lane * W4;
#pragma unroll
				for (int e = 0; e < W4; e++) ho[e] = make_float4(x[2 * e].x, x[2 * e].y, x[2 * e + 1].x, x[2 * e + 1].y);
			}
		}
		// the Rotate phasor travels one tile ahead like the samples: with a global_load_lds in flight the compiler
		// waits for ALL vector memory operations at the first use of an ordinary load, so the only such use sits
		// at the top of the loop, where the tile itself is awaited anyway
		const float2 rotv = rot_next;
		const int tile_n = tile + 1 <= tile_last ? tile + 1 : tile_last;
		if constexpr (FMT == 5) {
			us_xs_fetch(tile_n, us_tab, us_xs); // (x[] above holds this tile: its registers are free)
			us_tab_fetch(tile + 2 <= tile_last ? tile + 2 : tile_last, us_tab);
		} else prefetch(tile_n);
		if (!PRE) rot_next = p.rot[(size_t)ROT_HIST + (long long)tile_n * 64 + lane];

		c2 x96;
		if constexpr (FMT == 4) {
			if constexpr (LC) x96 = run_fix_ladder_lc(xi, lc_u, lane);
			else x96 = run_fix_ladder(xi, fst);
			// before the stream starts the fixed-point stages hold zeros, which is -1.0 after the sign flip, but the float stages
			// behind them have seen nothing at all: the warm-up tile of the very first block contributes zeros
			if (p.stream_start && tile < 0) x96 = c2{ 0.0f, 0.0f };
		}
		else if constexpr (LC) x96 = run_ladder_lc<K>(x, lc_f, lane);
		else x96 = run_ladder<K>(x, st);
		if constexpr (PRE) {
			if (tile > tile_first) p.pre_out[(size_t)rx * p.pre_stride + (size_t)tile * 64 + lane] = make_float2(x96.x, x96.y);
		} else {
			// ---- FDC (DSP.cpp:283-293) + Rotate (DSP.cpp:296-316)
			const c2 xm1 = from_prev_lane(x96, fdc_p1);
			const c2 xm2 = from_prev_lane(xm1, fdc_p2);
			const auto fdc_carry = [&]() { fdc_p2 = carry_to_next_tile(xm2, xm1); fdc_p1 = carry_to_next_tile(xm1, x96); };
			c2 y = x96;
			if (p.has_fdc) { // alpha * (h1 + x) + h2 * beta: add, mul, mul, add (componentwise)
				const c2 s2 = xm2 + x96;
				y = s2 * p.alpha + xm1 * p.beta;
			}
			fdc_carry();
			const float RR = y.x * rotv.x, II = y.y * rotv.y, RI = y.x * rotv.y, IR = y.y * rotv.x;
			wave_sync(); // previous tile's x5/x6 reads are complete
			x5[0][8 + lane] = make_float2(RR - II, IR + RI); // up   -> channel A
			x5[1][8 + lane] = make_float2(RR + II, IR - RI); // down -> channel B
			wave_sync();
			// ---- DS2_a / DS2_b: lanes 0..31 channel A, lanes 32..63 channel B
			const int ch = lane >> 5, j = lane & 31;
#ifdef ABL_NO_LDS_TAIL // (ablation builds only: wrong results -- the same arithmetic on registers, no LDS round trip to wait for)
			{
				float2 vv[6] = { make_float2(RR, II), make_float2(RI, IR), make_float2(II, RR), make_float2(IR, RI), make_float2(RR, RI), make_float2(II, IR) };
				float2 oo[1];
				cic5_dec_chunk<1>(vv, oo);
				x6[ch][8 + j] = oo[0];
			}
#else
			x6[ch][8 + j] = cic5_small(reinterpret_cast<const float4*>(&x5[ch][0]), j);
#endif
			wave_sync();
			// ---- FilterCIC5 (DSP.cpp:132-157)
			{
				const float2* src = &x6[ch][8 + j - 5];
				float2 v[6];
#ifdef ABL_NO_LDS_TAIL
				for (int e = 0; e < 6; e++) v[e] = make_float2(RR * (float)e, II);
				(void)src;
#else
#pragma unroll
				for (int e = 0; e < 6; e++) v[e] = src[e];
#endif
#pragma unroll
				for (int lvl = 0; lvl < 5; lvl++) {
#pragma unroll
					for (int i = 0; i < 5 - lvl; i++) v[i] = cadd(v[i + 1], v[i]);
				}
				if (tile > tile_first) {
					float2* dst = p.c48 + ((size_t)rx * 2 + ch) * p.c48_stride + (size_t)tile * 32 + j;
					*dst = make_float2(v[0].x * 0.03125f, v[0].y * 0.03125f);
				}
			}
			wave_sync();
			if (lane < 8) { x5[0][lane] = x5[0][64 + lane]; x5[1][lane] = x5[1][64 + lane]; }
			else if (lane < 16) { x6[0][lane - 8] = x6[0][32 + lane - 8]; x6[1][lane - 8] = x6[1][32 + lane - 8]; }
		}
	}
	if constexpr (!PRE) {
		if (p.fft_windows > 0) k1_fft_tail(p, rx, span, XFFT_IN_XT ? reinterpret_cast<float2*>(xt) : xfft);
	}
}

// ------------------------------------------------------------------------------------------
// K1u: the tail of an interpolated ladder (sample rates between two 2^k buckets, Model.cpp:163-189, e.g. 6 MSPS):
// Upsample (linear fractional resampler, DSP/DSP.cpp:192-212) -> DS2_2 -> DS2_1 -> FDC -> Rotate -> DS2_a/b ->
// FilterCIC5, on the pre-decimated stream written by the PRE pass.  The resampler's (input index, alpha) sequence
// is data independent and comes as a host-generated table (the float accumulation `alpha += increment` is
// replayed exactly there), so every output is a pure function of nearby inputs: a span of K1U_M = 128 output samples
// per channel is computed from scratch, with the short halos of every stage recomputed (no carried state), and a
// workgroup walks a few consecutive spans with the next one's loads in flight (round 5).  The stream carries 1/16 of
// the input rate; the kernel is 48 us per flush of the 6 MSPS ladder when alone, a tenth of that step.
// ------------------------------------------------------------------------------------------
constexpr int K1U_T = 320; // threads per workgroup of these front ends: their stages have 2 M + 15 .. 2 M + 19 = 271 .. 275 items at M = 128 -- with 256 threads a
                           // second round for the last 15 .. 19 of them, in the FIR stage of the decimate-by-3 ladder (26 taps) half of the kernel
// spans a workgroup of k1u_resample_frontend walks: at most K1U_SPW, as few as leave K1U_MIN_WGS workgroups (four rounds of a chip that holds
// 1,536 of them; longer walks measured no better, profiles/r05_expL_resampler_span_walk.txt)
#ifndef K1U_MIN_WGS
#define K1U_MIN_WGS (4 * 1536)
#endif
#ifndef K1U_SPW
#define K1U_SPW 8
#endif
constexpr int K1U_M = 128; // 48 kHz outputs per channel per workgroup (every block is a whole number of 512-sample windows, aisgpu.cpp)
// Round 4: 128, was 32.  With 32 a workgroup's stages had 64 .. 339 items for its 256 threads and six barriers for them -- 85 us per
// block on the 6 MSPS ladder's front stream, a fifth of the step; the halos are 83 samples per 8 M inputs either way.

__device__ __forceinline__ float2 cic5_at(const float2* a, int pos2j) { // decimating CIC5 output from a[pos2j-5 .. pos2j]
	float2 v[6];
#pragma unroll
	for (int e = 0; e < 6; e++) v[e] = a[pos2j - 5 + e];
	float2 o[1];
	cic5_dec_chunk<1>(v, o);
	return o[0];
}

// NPOST: CIC5 stages between the resampler and the 96 kHz point -- 2 (buckets 384k ... 12288k), 1 (rates resampled into the 192k
// bucket: US >> DS2_1, Model.cpp:323-329), 0 (96 kSPS input, no resampler at all: convert >> ROT, Model.cpp:332-334; xin is the
// converted input itself)
// LDS-only barrier of a multi-wave workgroup: __syncthreads() also waits for every outstanding global load (s_waitcnt vmcnt(0)), which would
// drain the next span's prefetch at the first barrier of this span's stages
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Round 5 (late): a workgroup walks p.spw CONSECUTIVE spans of its receiver, and what a span needs from memory -- table entries, Rotate
// phasor, the staged input span -- is requested one span ahead into registers (the span's bounds two ahead: the input loads depend on
// them).  Alone the kernel was two dependent round trips and six barriers per 128 outputs, 2 x its instruction floor (60 us per flush of
// the 6 MSPS ladder, 111 us at 2.4 MSPS); the barriers between the stages wait for LDS only, so the prefetch stays in flight.
template <int NPOST, int M>
__global__ __launch_bounds__(K1U_T) void k1u_resample_frontend(K1uParams p) {
	// One pool, so that the staged input span (XS) can lie over the buffers of the later stages:
	constexpr int UN = NPOST == 2 ? 8 * M + 88 : 0;   // u(n),  n  in [8 m0 - 83, 8 m0 + 8 M)
	constexpr int S1N = NPOST >= 1 ? 4 * M + 40 : 0;  // 192 kHz-equivalent level, j in [4 m0 - 39, 4 m0 + 4 M)
	constexpr int S2N = 2 * M + 18;                   // 96 kHz level,           i in [2 m0 - 17, 2 m0 + 2 M)
	constexpr int RUN = 2 * M + 16;                   // rotated up / down,      i in [2 m0 - 15, 2 m0 + 2 M)   (two rows)
	constexpr int DDN = M + 6;                        // DS2_a/b output,         j in [m0 - 5, m0 + M)          (two rows)
	constexpr bool DD_IN_U = NPOST == 2;              // (u is dead long before: 19.9 KB, eight workgroups per CU instead of seven)
	__shared__ float2 pool[UN + S1N + S2N + 2 * RUN + (DD_IN_U ? 0 : 2 * DDN)];
	float2* const U = pool; float2* const S1 = pool + UN; float2* const S2 = S1 + S1N; float2* const RU = S2 + S2N; float2* const DD = DD_IN_U ? U : RU + 2 * RUN;
	const int t = threadIdx.x;
	const int rx = blockIdx.y;
	const XRow xr = make_xrow(p, rx); // xr[i]: i relative to the current block start
	static_assert(2 * M + 15 <= K1U_T, "k1u: one Rotate item per thread");
	// Upsample (DSP.cpp:192-212): output n = (1 - alpha) * x[b - 1] + alpha * x[b], products rounded separately (DSP.cpp:199), (b, alpha) from
	// the tables: the entries of ALL of a thread's outputs, and the input span they point into (us_idx is non-decreasing; an upsampler's
	// span is no longer than its outputs), staged in LDS with coalesced loads; the interpolation reads LDS.
	constexpr int NU = NPOST == 2 ? 8 * M + 83 : NPOST == 1 ? 4 * M + 39 : 1; // resampled samples a span needs
	constexpr int NQ = (NU + K1U_T - 1) / K1U_T;
	constexpr int XS_CAP = NPOST == 0 ? 2 * M + 17 : (NPOST == 2 ? S1N : 0) + S2N + 2 * RUN; // the staged span lies over the later stages' buffers
	constexpr int NXS = (XS_CAP + K1U_T - 1) / K1U_T;
	float2* const XS = NPOST == 2 ? S1 : S2;
	struct Pre { float2 rot; int ib[NQ]; float al[NQ]; int lo, hi; float2 xs[NXS]; };
	const auto n_lo_of = [&](int m0) { return NPOST == 2 ? 8 * m0 - 83 : NPOST == 1 ? 4 * m0 - 39 : 2 * m0 - 17; };
	const auto bounds = [&](int m0, int& lo, int& hi) { // the input samples a span touches (NPOST = 0: the span itself)
		if constexpr (NPOST >= 1) { lo = p.us_idx[US_HIST + n_lo_of(m0)] - 1; hi = p.us_idx[US_HIST + n_lo_of(m0) + NU - 1]; }
		else { lo = 2 * m0 - 17; hi = 2 * m0 + 2 * M - 1; }
	};
	const auto fetch = [&](int m0, int lo, int hi, Pre& q) {
		q.rot = t < 2 * M + 15 ? p.rot[ROT_HIST + 2 * m0 - 15 + t] : make_float2(0.0f, 0.0f);
		if constexpr (NPOST >= 1) {
#pragma unroll
			for (int k = 0; k < NQ; k++) {
				const int i = t + K1U_T * k;
				q.ib[k] = i < NU ? p.us_idx[US_HIST + n_lo_of(m0) + i] : 0;
				q.al[k] = i < NU ? p.us_alpha[US_HIST + n_lo_of(m0) + i] : 0.0f;
			}
		}
		q.lo = lo; q.hi = hi;
		if (hi - lo + 1 <= XS_CAP) { // (workgroup-uniform)
			const XSpan x(xr, lo, hi);
#pragma unroll
			for (int k = 0; k < NXS; k++) {
				const int i = t + K1U_T * k;
				q.xs[k] = i <= hi - lo ? x[lo + i] : make_float2(0.0f, 0.0f);
			}
		}
	};
	const int span0 = blockIdx.x * p.spw;
	int lo1 = 0, hi1 = 0, lo2 = 0, hi2 = 0;
	Pre cur;
	{
		int lo0, hi0;
		bounds(span0 * M, lo0, hi0);
		if (p.spw > 1) bounds((span0 + 1) * M, lo1, hi1);
		fetch(span0 * M, lo0, hi0, cur);
	}
#pragma unroll 1
	for (int sp = 0; sp < p.spw; sp++) {
		const int m0 = (span0 + sp) * M;
		Pre nxt = cur;
		if constexpr (NPOST >= 1) {
			float2* const dst = NPOST == 2 ? U : S1;
			const int lo = cur.lo, hi = cur.hi;
			if (hi - lo + 1 <= XS_CAP) {
#pragma unroll
				for (int k = 0; k < NXS; k++) {
					const int i = t + K1U_T * k;
					if (i <= hi - lo) XS[i] = cur.xs[k];
				}
				// this span's registers are consumed (ib / al / rot stay): the next span's loads go out now and land during the stages below
				if (sp + 1 < p.spw) fetch(m0 + M, lo1, hi1, nxt);
				if (sp + 2 < p.spw) bounds(m0 + 2 * M, lo2, hi2);
				lds_barrier();
#pragma unroll
				for (int k = 0; k < NQ; k++) {
					const int i = t + K1U_T * k;
					if (i < NU) { // (dst and XS do not overlap)
						const float2 a = XS[cur.ib[k] - 1 - lo], b = XS[cur.ib[k] - lo];
						const float w0 = 1 - cur.al[k];
						dst[i] = make_float2(w0 * a.x + cur.al[k] * b.x, w0 * a.y + cur.al[k] * b.y);
					}
				}
			} else { // (not an upsampler's table: straight from global memory)
				const XSpan x(xr, lo, hi);
#pragma unroll
				for (int k = 0; k < NQ; k++) {
					const int i = t + K1U_T * k;
					if (i < NU) {
						const float2 a = x[cur.ib[k] - 1], b = x[cur.ib[k]];
						const float w0 = 1 - cur.al[k];
						dst[i] = make_float2(w0 * a.x + cur.al[k] * b.x, w0 * a.y + cur.al[k] * b.y);
					}
				}
				if (sp + 1 < p.spw) fetch(m0 + M, lo1, hi1, nxt);
				if (sp + 2 < p.spw) bounds(m0 + 2 * M, lo2, hi2);
			}
			lds_barrier();
		}
		if constexpr (NPOST == 2) {
			for (int q = t; q < 4 * M + 39; q += K1U_T) { // j = 4 m0 - 39 + q needs u(2j-5..2j): U index 2j - n_lo
				const int j = 4 * m0 - 39 + q;
				S1[q] = cic5_at(U, 2 * j - (8 * m0 - 83));
			}
			lds_barrier();
		}
		if constexpr (NPOST >= 1) {
			for (int q = t; q < 2 * M + 17; q += K1U_T) { // i = 2 m0 - 17 + q needs s1(2i-5..2i)
				const int i = 2 * m0 - 17 + q;
				S2[q] = cic5_at(S1, 2 * i - (4 * m0 - 39));
			}
		} else {
#pragma unroll
			for (int k = 0; k < NXS; k++) {
				const int i = t + K1U_T * k;
				if (i < 2 * M + 17) S2[i] = cur.xs[k];
			}
			if (sp + 1 < p.spw) fetch(m0 + M, lo1, hi1, nxt);
			if (sp + 2 < p.spw) bounds(m0 + 2 * M, lo2, hi2);
		}
		lds_barrier();
		for (int q = t; q < 2 * M + 15; q += K1U_T) { // i = 2 m0 - 15 + q: FDC (DSP.cpp:283-293) + Rotate (DSP.cpp:296-316)
			const int i = 2 * m0 - 15 + q;
			const int si = i - (2 * m0 - 17);
			const float2 xm2 = S2[si - 2], xm1 = S2[si - 1], xv = S2[si];
			float2 y = xv;
			if (p.has_fdc) {
				const float2 s2 = cadd(xm2, xv);
				y = make_float2(p.alpha * s2.x + xm1.x * p.beta, p.alpha * s2.y + xm1.y * p.beta);
			}
			const float2 rot = cur.rot; // (= p.rot[ROT_HIST + i]: q == t, one item per thread)
			const float RR = y.x * rot.x, II = y.y * rot.y, RI = y.x * rot.y, IR = y.y * rot.x;
			RU[q] = make_float2(RR - II, IR + RI);
			RU[RUN + q] = make_float2(RR + II, IR - RI);
		}
		lds_barrier();
		for (int q = t; q < 2 * (M + 5); q += K1U_T) { // DS2_a / DS2_b: j = m0 - 5 + jj needs up(2j-5..2j)
			const int ch = q / (M + 5), jj = q % (M + 5);
			const int j = m0 - 5 + jj;
			DD[ch * DDN + jj] = cic5_at(RU + ch * RUN, 2 * j - (2 * m0 - 15));
		}
		lds_barrier();
		if (t < 2 * M) { // FilterCIC5 (DSP.cpp:132-157)
			const int ch = t / M, mm = t % M;
			float2 v[6];
#pragma unroll
			for (int e = 0; e < 6; e++) v[e] = DD[ch * DDN + mm + e]; // d(m-5 .. m)
#pragma unroll
			for (int lvl = 0; lvl < 5; lvl++) {
#pragma unroll
				for (int i = 0; i < 5 - lvl; i++) v[i] = cadd(v[i + 1], v[i]);
			}
			p.c48[((size_t)rx * 2 + ch) * p.c48_stride + m0 + mm] = make_float2(v[0].x * 0.03125f, v[0].y * 0.03125f);
		}
		lds_barrier(); // (the next span's staging overwrites the pool)
		cur = nxt; lo1 = lo2; hi1 = hi2;
	}
}

// ------------------------------------------------------------------------------------------
// K1x: channel mode X (`-c X`, Model.cpp:35-107): ONE channel, already centred, at 48 / 96 / 192 kSPS (or resampled into the
// next of these): convert >> [US] >> [DS2_2] >> [DS2_1] >> [FDC] >> FCIC5_a.  Same tile scheme as K1u (a workgroup produces
// K1U_M outputs at 48 kHz and recomputes the short halos of every stage); only channel A's row of c48 is written, channel
// B's stays silent (zero), so everything behind the front end runs unchanged.
// ------------------------------------------------------------------------------------------
template <int NPOST, int M>
__global__ __launch_bounds__(K1U_T) void k1x_single_channel(K1uParams p) {
	__shared__ float2 U[NPOST == 2 ? 4 * M + 44 : 1];  // 192 kHz level, n in [4 m0 - 43, 4 m0 + 4 M)
	__shared__ float2 S1[NPOST >= 1 ? 2 * M + 20 : 1]; // 96 kHz level,  j in [2 m0 - 19, 2 m0 + 2 M)
	__shared__ float2 T[M + 8];                        // 48 kHz level,  m in [m0 - 7, m0 + M)
	__shared__ float2 F[M + 6];                        // behind the droop filter, m in [m0 - 5, m0 + M)
	const int t = threadIdx.x;
	const int rx = blockIdx.y;
	const int m0 = blockIdx.x * M;
	const XRow x = make_xrow(p, rx);
	const auto level0 = [&](int n) -> float2 { // sample n of the stream the first CIC5 stage (or the 48 kHz point) sees
		if (!p.us_idx) return x[n];
		const int i = p.us_idx[US_HIST + n];
		const float al = p.us_alpha[US_HIST + n];
		const float2 a = x[i - 1], b = x[i];
		const float w0 = 1 - al; // DSP.cpp:199, products rounded separately
		return make_float2(w0 * a.x + al * b.x, w0 * a.y + al * b.y);
	};
	if constexpr (NPOST == 2) {
		for (int q = t; q < 4 * M + 43; q += K1U_T) U[q] = level0(4 * m0 - 43 + q);
		__syncthreads();
		for (int q = t; q < 2 * M + 19; q += K1U_T) S1[q] = cic5_at(U, 2 * (2 * m0 - 19 + q) - (4 * m0 - 43));
		__syncthreads();
	} else if constexpr (NPOST == 1) {
		for (int q = t; q < 2 * M + 19; q += K1U_T) S1[q] = level0(2 * m0 - 19 + q);
		__syncthreads();
	}
	for (int q = t; q < M + 7; q += K1U_T) {
		if constexpr (NPOST >= 1) T[q] = cic5_at(S1, 2 * (m0 - 7 + q) - (2 * m0 - 19));
		else T[q] = level0(m0 - 7 + q);
	}
	__syncthreads();
	for (int q = t; q < M + 5; q += K1U_T) { // FDC (DSP.cpp:283-293): alpha * (h1 + x) + h2 * beta
		const float2 xm2 = T[q], xm1 = T[q + 1], xv = T[q + 2];
		float2 y = xv;
		if (p.has_fdc) {
			const float2 s2 = cadd(xm2, xv);
			y = make_float2(p.alpha * s2.x + xm1.x * p.beta, p.alpha * s2.y + xm1.y * p.beta);
		}
		F[q] = y;
	}
	__syncthreads();
	for (int q = t; q < M; q += K1U_T) { // FilterCIC5 (DSP.cpp:132-157)
		float2 v[6];
#pragma unroll
		for (int e = 0; e < 6; e++) v[e] = F[q + e]; // f(m-5 .. m)
#pragma unroll
		for (int lvl = 0; lvl < 5; lvl++) {
#pragma unroll
			for (int i = 0; i < 5 - lvl; i++) v[i] = cadd(v[i + 1], v[i]);
		}
		p.c48[((size_t)rx * p.c48_rows_per_rx) * p.c48_stride + m0 + q] = make_float2(v[0].x * 0.03125f, v[0].y * 0.03125f);
	}
}

// K1x at 96 kSPS without a resampler (NPOST = 1, us_idx == nullptr: `-c X` at the rate mode X is usually fed with), register / DPP
// form (round 6, late) -- k1x_single_channel<1, 512> is five waves x four barriers per 512 outputs, ~4 wave-instructions per output
// (0.66 ms per 4,096 receivers x 49,152 samples alone, issue-bound like everything behind it, so the step was the sum of its kernels).
// Here one wave walks a span of tiles of 1,024 input samples like k1_dpp: lane l owns 16 consecutive samples, Downsample2CIC5 is
// reg_stage<16> (eight 48 kHz samples per lane), the droop filter takes the two samples in front of them and FilterCIC5 the five in
// front of its eight from the neighbouring lane through DPP wave shifts (lane 0: lane 63's values of the previous tile, shadow
// registers); a tile comes straight from memory into the wave's LDS buffer (global_load_lds, swizzled as in k1_dpp), the next one is
// requested as soon as this one is in registers.  A span starts one tile early to fill the filters:
// nothing put out depends on more than the 19 input samples in front of it (FCIC5 5 + FDC 2 at 48 kHz, CIC5 5 at 96 kHz), so only
// the last 128 samples of that tile are fetched (one load instruction; in front of the block: the library's look-back of DSK_HIST samples).  Same sums in the same pairs as
// k1x_single_channel (cic5_dec_chunk; the shared pyramid of FilterCIC5 forms exactly the pair sums of the six-sample one): same bits.
// NPOST = 0 / 2 (round 6, last): the same waves at 48 kSPS (no CIC5 stage: a tile is 512 samples, a lane's eight samples are its 48 kHz
// samples) and at 192 kSPS (two stages, DS2_2 >> DS2_1: a tile is 2,048 samples, 32 per lane); the pieces of a lane's row are swizzled
// by (row / 4) % 4, row % 8 and row % 16 (4, 8, 16 pieces per row: a row starts every 64, 128, 256 bytes).
template <int NPOST>
__global__ __launch_bounds__(64) void k1x_wave(K1uParams p, int tiles_per_span) {
	constexpr int C0 = 8 << NPOST;       // input samples per lane and tile
	constexpr int TS = 64 * C0;          // input samples per tile (512 at 48 kHz: one window of the row)
	constexpr int W4 = C0 / 2;           // 16-byte pieces per lane row
	constexpr int NE = TS / 128;         // load instructions per tile (128 samples each)
	__shared__ __attribute__((aligned(16))) float4 xt[TS / 2 < 512 ? 512 : TS / 2]; // the tile, XOR-swizzled in units of 16 B (global_load_lds); at least the 8 KB of the analysis
	const int lane = threadIdx.x, rx = blockIdx.y, span = blockIdx.x;
	const int tiles = p.L / 512;
	const int tile_first = span * tiles_per_span - 1; // warm-up tile
	int tile_last = tile_first + tiles_per_span;
	if (tile_last >= tiles) tile_last = tiles - 1;
	const XRow xr = make_xrow(p, rx);
	float2* out = p.c48 + ((size_t)rx * p.c48_rows_per_rx) * p.c48_stride;
	HaloState<32, c2> s32 = {};
	HaloState<16, c2> s16 = {};
	HaloState<8, c2> sf = {};
	c2 th0 = { 0.f, 0.f }, th1 = { 0.f, 0.f };
	// physical slot of piece q of row r: r W4 + (q ^ swz(r)), swz(r) = (r / 4) % 4 (W4 = 4), r % W4 (W4 = 8, 16): the lanes a 128-bit read serves
	// together hit different bank groups.  The DMA writes slot 64 e + lane, so that lane fetches the piece that belongs there.
	const auto swz = [](int r) { return W4 == 4 ? (r >> 2) & 3 : r & (W4 - 1); };
	const auto dma_piece = [&](int e) { // global piece (16 B) of the tile for slot 64 e + lane
		const int sl = 64 * e + lane, r = sl / W4, qs = sl % W4;
		return r * W4 + (qs ^ swz(r));
	};
	const auto prefetch = [&](int tile) {
		if (tile == tile_first) { // (wave-uniform) the last 128 samples of the warm-up tile = one load instruction; in front of the block: the look-back
			static_assert(DSK_HIST == 128, "k1x_wave: the look-back in front of a block is the warm-up tile's last load instruction");
			const float2* b128 = tile >= 0 ? xr.cur + (size_t)tile * TS + (TS - 128) : (xr.prev ? xr.prev + (xr.n - 128) : xr.cur - 128);
			__builtin_amdgcn_global_load_lds((const void*)(reinterpret_cast<const uint4*>(b128) + (dma_piece(NE - 1) - 64 * (NE - 1))),
			                                 (__attribute__((address_space(3))) void*)(xt + (NE - 1) * 64), 16, 0, K1_LOAD_AUX);
		} else {
			const uint4* src = reinterpret_cast<const uint4*>(xr.cur + (size_t)tile * TS);
#pragma unroll
			for (int e = 0; e < NE; e++)
				__builtin_amdgcn_global_load_lds((const void*)(src + dma_piece(e)), (__attribute__((address_space(3))) void*)(xt + e * 64), 16, 0, K1_LOAD_AUX);
		}
	};
	prefetch(tile_first);
	for (int tile = tile_first; tile <= tile_last; tile++) {
		c2 x[C0];
#pragma unroll
		for (int e = 0; e < W4; e++) { // every component is used, so these stay 128-bit loads
			const float4 v = xt[lane * W4 + (e ^ swz(lane))];
			x[2 * e] = c2{ v.x, v.y }; x[2 * e + 1] = c2{ v.z, v.w };
		}
		wave_sync(); // the tile is in registers: the next one may land in xt
		if (tile < tile_last) prefetch(tile + 1);
		c2 t8[8];
		if constexpr (NPOST == 2) { // DS2_2 >> DS2_1: Downsample2CIC5 twice (DSP.cpp:93-117)
			c2 y16[16];
			reg_stage<32>(x, s32, y16);
			reg_stage<16>(y16, s16, t8);
		} else if constexpr (NPOST == 1) reg_stage<16>(x, s16, t8); // DS2_1
		else {
#pragma unroll
			for (int j = 0; j < 8; j++) t8[j] = x[j];
		}
		c2 f8[8];
		if (p.has_fdc) { // FDC (DSP.cpp:283-293): alpha * (h1 + x) + h2 * beta
			const c2 tm2 = from_prev_lane(t8[6], th0), tm1 = from_prev_lane(t8[7], th1);
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const c2 xm2 = j >= 2 ? t8[j - 2] : (j == 0 ? tm2 : tm1), xm1 = j >= 1 ? t8[j - 1] : tm1;
				const c2 s2 = xm2 + t8[j];
				f8[j] = s2 * p.alpha + xm1 * p.beta;
			}
			th0 = carry_to_next_tile(tm2, t8[6]);
			th1 = carry_to_next_tile(tm1, t8[7]);
		} else {
#pragma unroll
			for (int j = 0; j < 8; j++) f8[j] = t8[j];
		}
		// FilterCIC5 (DSP.cpp:132-157): out(m) from f(m-5 .. m), five levels of pair sums, * 1/32
		c2 h[5];
		get_halo<8, c2>(f8, sf, h);
		c2 g[13];
#pragma unroll
		for (int i = 0; i < 5; i++) g[i] = h[i];
#pragma unroll
		for (int i = 0; i < 8; i++) g[5 + i] = f8[i];
#pragma unroll
		for (int lvl = 0; lvl < 5; lvl++) {
#pragma unroll
			for (int i = 0; i < 12 - lvl; i++) g[i] = g[i + 1] + g[i];
		}
		put_halo<8, c2>(f8, sf, h);
		if (tile > tile_first) {
			float4* dst = reinterpret_cast<float4*>(out + (size_t)tile * 512 + lane * 8);
#pragma unroll
			for (int e = 0; e < 4; e++) {
				const c2 a = g[2 * e] * 0.03125f, b = g[2 * e + 1] * 0.03125f;
				dst[e] = make_float4(a.x, a.y, b.x, b.y);
			}
		}
	}
	// a tile is a window of this row: the span's windows in pairs (launch_k1x: spans of an even number of tiles where p.fz is set)
	static_assert(sizeof(xt) >= 1024 * sizeof(float2), "k1x_wave: the spectral analysis works in 8 KB of the tile buffer");
	if (p.fz) wave_fft_tail(p, (size_t)rx * p.c48_rows_per_rx, false, tile_first + 1, tile_last - tile_first, reinterpret_cast<float2*>(xt));
}

// ------------------------------------------------------------------------------------------
// K1k: the tail of a decimate-by-3 ladder (rates 288k * 2^k, Model.cpp:207-219,248-259,278-289,308-313):
// DownsampleKFilter (26-tap Blackman-Harris FIR, keep every 3rd output, DSP.cpp:160-189, DSP.h:195-201) ->
// Rotate -> DS2_a/b -> FilterCIC5 (no droop filter on these ladders).  With block lengths that are a multiple
// of 3 the filter's input phase is 0 at every block start, so 96 kHz sample i of a block is
// sum_t taps[t] * x[3 i + t - 25] (accumulated left to right from 0).  Same tile scheme as K1u: a workgroup
// produces 32 outputs per channel and recomputes the short halos of every stage.
// ------------------------------------------------------------------------------------------
template <int M>
__global__ __launch_bounds__(K1U_T) void k1k_dsk_frontend(K1kParams p) {
	__shared__ float2 X[6 * M + 72];       // x(n), n in [6 m0 - 70, 6 m0 + 6 M)
	__shared__ float2 RU[2][2 * M + 16];   // rotated up/down, i in [2 m0 - 15, 2 m0 + 2 M)
	__shared__ float2 DD[2][M + 6];        // DS2_a/b output,  j in [m0 - 5, m0 + M)
	const int t = threadIdx.x;
	const int rx = blockIdx.y;
	const int m0 = blockIdx.x * M;
	const XRow xr = make_xrow(p, rx);
	const int n_lo = 6 * m0 - 70;
	if (p.us_idx) { // Upsample in front of the filter (rates below a decimate-by-3 bucket): sample n of the flush is interpolated
		const XSpan x(xr, p.us_idx[US_HIST + n_lo] - 1, p.us_idx[US_HIST + n_lo + 6 * M + 69]);
		for (int q = t; q < 6 * M + 70; q += K1U_T) { // from the input stream like in K1u (DSP.cpp:199: products rounded separately)
			const int n = n_lo + q;
			const int i = p.us_idx[US_HIST + n];
			const float al = p.us_alpha[US_HIST + n];
			const float2 a = x[i - 1], b = x[i];
			const float w0 = 1 - al;
			X[q] = make_float2(w0 * a.x + al * b.x, w0 * a.y + al * b.y);
		}
	} else
	for (int q = t; q < 6 * M + 70; q += K1U_T) X[q] = xr[n_lo + q];
	__syncthreads();
	for (int q = t; q < 2 * M + 15; q += K1U_T) { // i = 2 m0 - 15 + q
		const int i = 2 * m0 - 15 + q;
		const float2* d = X + (3 * i - 25 - n_lo);
		float2 acc = make_float2(0.0f, 0.0f);
#pragma unroll
		for (int k = 0; k < 26; k++) acc = make_float2(acc.x + p.taps[k] * d[k].x, acc.y + p.taps[k] * d[k].y);
		const float2 rot = p.rot[ROT_HIST + i];
		const float RR = acc.x * rot.x, II = acc.y * rot.y, RI = acc.x * rot.y, IR = acc.y * rot.x; // DSP.cpp:296-316
		RU[0][q] = make_float2(RR - II, IR + RI);
		RU[1][q] = make_float2(RR + II, IR - RI);
	}
	__syncthreads();
	for (int q = t; q < 2 * (M + 5); q += K1U_T) { // DS2_a / DS2_b: j = m0 - 5 + jj needs up(2j-5..2j)
		const int ch = q / (M + 5), jj = q % (M + 5);
		const int j = m0 - 5 + jj;
		DD[ch][jj] = cic5_at(RU[ch], 2 * j - (2 * m0 - 15));
	}
	__syncthreads();
	for (int q = t; q < 2 * M; q += K1U_T) { // FilterCIC5 (DSP.cpp:132-157)
		const int ch = q / M, mm = q % M;
		float2 v[6];
#pragma unroll
		for (int e = 0; e < 6; e++) v[e] = DD[ch][mm + e]; // d(m-5 .. m)
#pragma unroll
		for (int lvl = 0; lvl < 5; lvl++) {
#pragma unroll
			for (int i = 0; i < 5 - lvl; i++) v[i] = cadd(v[i + 1], v[i]);
		}
		p.c48[((size_t)rx * 2 + ch) * p.c48_stride + m0 + mm] = make_float2(v[0].x * 0.03125f, v[0].y * 0.03125f);
	}
}

// K1k without a resampler in front (the 288k * 2^k ladders themselves), register / DPP form (round 6, late; the form of k1x_wave).
// k1k_dsk_frontend<128> is five waves x four barriers per 128 outputs per channel: 0.73 ms per 1,024 receivers x 196,608 samples alone,
// issue-bound like everything behind it.  Here one wave walks a span of tiles of 1,536 input samples (512 at 96 kHz, 256 per channel
// at 48 kHz); a tile comes straight from memory into the wave's LDS buffer (global_load_lds, 1 KB contiguous per instruction), with
// the previous tile's last 32 samples kept in front of it.  Lane l owns the 96 kHz samples i = 8 l .. 8 l + 7: DownsampleKFilter
// (DSP.cpp:160-189) is sum_k taps[k] * x[3 i - 25 + k], accumulated from the left, over a window of 48 samples the lane reads from LDS
// in 24 128-bit pieces (the pieces are XOR-swizzled by their 256-byte row -- through the global addresses of the DMA -- so that the
// sixteen lanes a read serves hit sixteen different bank groups: a lane's window starts every 192 bytes); Rotate (DSP.cpp:296-316) from
// the table (requested a tile ahead); DS2_a / DS2_b as reg_stage<8>, FilterCIC5 on the lane's four 48 kHz samples per channel with the
// five in front of them from the two neighbouring lanes (get_halo<4>).  A span starts a tile early: nothing put out depends on more than
// 70 input samples in front of it, so of that tile only the last 128 samples are fetched (one load instruction; in front of the block:
// the library's look-back of DSK_HIST samples).  Same products and sums in the same order as k1k_dsk_frontend: same bits.
__device__ __forceinline__ int k1kw_slot(int pf) { return (pf & ~15) | ((pf & 15) ^ ((pf >> 4) & 3)); } // logical 16-byte piece (32 look-back samples = 16 pieces in front) -> LDS slot
// FIR = false (P = K1uParams; round 6, last): the same waves for dual-channel 96 kSPS input, the ladder's last bucket (convert >> ROT >> DS2_a/b >>
// FCIC5, Model.cpp:332-334; k1u_resample_frontend<0> before: 4.96 ms per step of 4,096 receivers x 49,152 samples) -- a tile is 512 samples,
// a lane's eight 96 kHz samples come straight out of the tile (pieces swizzled by (row / 4) % 4: a lane's 64 bytes start every 64 bytes).
// US = true (FIR only; round 6, last): Upsample (DSP.cpp:192-212) in front of the filter -- the rates resampled into a decimate-by-3 bucket
// (250 kSPS -> 288k, ...).  A tile's 1,536 samples are not fetched but formed: the input samples they interpolate between are one
// contiguous span of the pre-decimated stream (us_idx is non-decreasing, the ratio at most 1), which goes into a second LDS buffer by
// global_load_lds (every 16-byte piece from the block of the ring of three it lies in); lane l then forms the samples 64 j + l of the tile
// from its table entries (coalesced) and the span (products and sum rounded separately, as k1k_dsk_frontend forms them) and puts them
// where the DMA would have.  Simple form: span, tables and arithmetic of a tile one after the other (the other wave of the SIMD overlaps).
template <bool FIR, class P, bool US = false>
__global__ __launch_bounds__(64) void k1k_wave(P p, int tiles_per_span) {
	static_assert(!US || FIR, "k1k_wave: Upsample sits in front of the filter");
	constexpr int TS = FIR ? 1536 : 512;  // input samples per tile (512 at 96 kHz)
	constexpr int NE = TS / 128;          // load instructions per tile (128 samples each)
	constexpr int LB = FIR ? 16 : 0;      // pieces of look-back in front of the tile
	__shared__ __attribute__((aligned(16))) float4 xt[LB + TS / 2 < 512 ? 512 : LB + TS / 2]; // (at least the 8 KB the spectral analysis at the end of the span works in)
	constexpr int XSP = US ? TS / 2 + 64 : 1; // pieces of the input span of a tile (at most TS + 2 samples, from an even sample on, in whole load instructions)
	__shared__ __attribute__((aligned(16))) float4 xs[XSP];
	const int lane = threadIdx.x, rx = blockIdx.y, span = blockIdx.x;
	const int tiles = p.L / 256;
	const int tile_first = span * tiles_per_span - 1; // warm-up tile
	int tile_last = tile_first + tiles_per_span;
	if (tile_last >= tiles) tile_last = tiles - 1;
	const XRow xr = make_xrow(p, rx);
	float2* outa = p.c48 + ((size_t)rx * 2) * p.c48_stride;
	float2* outb = outa + p.c48_stride;
	HaloState<8, c2> sup = {}, sdn = {};
	HaloState<4, c2> sfa = {}, sfb = {};
	// DMA: slot LB + 64 e + lane <- the piece g of the tile that belongs there (the swizzle term does not depend on e)
	const int dma_g = FIR ? (lane & ~15) + ((lane & 15) ^ ((1 + (lane >> 4)) & 3)) : (lane & ~3) + ((lane & 3) ^ ((lane >> 4) & 3));
	float4 rot_next[4];
	// US: the input span of a tile's samples [first, TS) (first = TS - 128 for the warm-up tile) requested into xs -- a tile ahead, during
	// the previous tile's filter, which works from registers; returns the span's first sample
	const auto span_issue = [&](int tile) -> int {
		int base = 0;
		if constexpr (US) {
			const int first = tile == tile_first ? TS - 128 : 0;
			const int n0 = tile * TS + first, n1 = tile * TS + TS - 1; // outputs of the flush; in front of the tables' look-back: zeros (nothing kept depends on them)
			const int na = n0 < -US_HIST ? -US_HIST : n0;
			const int lo = __builtin_amdgcn_readfirstlane(p.us_idx[US_HIST + na]) - 1, hi = __builtin_amdgcn_readfirstlane(p.us_idx[US_HIST + n1]);
			base = lo & ~1;
			const int npieces = ((hi - base) >> 1) + 1; // <= TS / 2 + 2
#pragma unroll 1
			for (int e = 0; e * 64 < npieces; e++) {
				const int pi = e * 64 + lane;
				const int i = base + 2 * (pi < npieces ? pi : npieces - 1); // (the last piece again: a load instruction writes all 64 slots)
				const float2* src = (i >= 0 || !xr.prev) ? xr.cur + i : (i >= -xr.n ? xr.prev + (xr.n + i) : xr.prev2 + (2 * xr.n + i));
				__builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(xs + e * 64), 16, 0, 0);
			}
		}
		return base;
	};
	// ... and the tile formed from it in LDS, where the DMA of the unresampled ladders would have put it
	int span_base = 0;
	const auto form_tile = [&](int tile) {
		if constexpr (US) {
			const int first = tile == tile_first ? TS - 128 : 0;
			const int base = span_base;
			int ib[TS / 64]; float al[TS / 64];
#pragma unroll
			for (int j = 0; j < TS / 64; j++) {
				const int n = tile * TS + 64 * j + lane;
				const bool on = 64 * j >= first && n >= -US_HIST; // (64 j >= first: wave-uniform)
				ib[j] = on ? p.us_idx[US_HIST + n] : base + 1;
				al[j] = on ? p.us_alpha[US_HIST + n] : 0.0f;
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // (s_waitcnt vmcnt(0): the span has landed)
			wave_sync();
			const float2* xss = reinterpret_cast<const float2*>(xs);
			float2* xtw = reinterpret_cast<float2*>(xt);
#pragma unroll
			for (int j = 0; j < TS / 64; j++) {
				if (64 * j < first) continue; // (wave-uniform)
				const int nl = 64 * j + lane, n = tile * TS + nl;
				const float2 a = xss[ib[j] - 1 - base], b = xss[ib[j] - base];
				const float w0 = 1 - al[j]; // DSP.cpp:199, products rounded separately
				float2 v = make_float2(w0 * a.x + al[j] * b.x, w0 * a.y + al[j] * b.y);
				if (n < -US_HIST) v = make_float2(0.0f, 0.0f);
				xtw[2 * k1kw_slot((nl + 32) >> 1) + (nl & 1)] = v;
			}
			wave_sync();
		}
	};
	const auto prefetch = [&](int tile) {
		if constexpr (US) {
			span_base = span_issue(tile);
		} else if (tile == tile_first) { // (wave-uniform) only the tile's last 128 samples = its last load instruction
			static_assert(DSK_HIST == 128, "k1k_wave: the look-back in front of a block is the warm-up tile's last load instruction");
			const float2* b128 = tile >= 0 ? xr.cur + (size_t)tile * TS + (TS - 128) : (xr.prev ? xr.prev + (xr.n - 128) : xr.cur - 128);
			__builtin_amdgcn_global_load_lds((const void*)(reinterpret_cast<const uint4*>(b128) + dma_g), (__attribute__((address_space(3))) void*)(xt + LB + (NE - 1) * 64), 16, 0, K1_LOAD_AUX);
		} else {
			const uint4* src = reinterpret_cast<const uint4*>(xr.cur + (size_t)tile * TS) + dma_g;
#pragma unroll
			for (int e = 0; e < NE; e++)
				__builtin_amdgcn_global_load_lds((const void*)(src + e * 64), (__attribute__((address_space(3))) void*)(xt + LB + e * 64), 16, 0, K1_LOAD_AUX);
		}
		int i0 = tile * 512 + lane * 8; // the lane's eight Rotate phasors (in front of the table's look-back: lanes whose outputs nobody keeps)
		i0 = i0 < -ROT_HIST ? -ROT_HIST : i0;
		const float4* r4 = reinterpret_cast<const float4*>(p.rot + ROT_HIST + i0);
#pragma unroll
		for (int e = 0; e < 4; e++) rot_next[e] = r4[e];
	};
	prefetch(tile_first);
	for (int tile = tile_first; tile <= tile_last; tile++) {
		form_tile(tile);
		c2 w[FIR ? 48 : 8];
		if constexpr (FIR) { // the lane's window: samples 24 l - 26 .. 24 l + 21 of the tile = pieces 12 l + 3 .. 12 l + 26 (look-back included)
#pragma unroll
			for (int k = 0; k < 24; k++) {
				const float4 v = xt[k1kw_slot(12 * lane + 3 + k)];
				w[2 * k] = c2{ v.x, v.y }; w[2 * k + 1] = c2{ v.z, v.w };
			}
		} else { // the lane's eight samples
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const float4 v = xt[4 * lane + (k ^ ((lane >> 2) & 3))];
				w[2 * k] = c2{ v.x, v.y }; w[2 * k + 1] = c2{ v.z, v.w };
			}
		}
		float4 rot[4];
#pragma unroll
		for (int e = 0; e < 4; e++) rot[e] = rot_next[e];
		wave_sync();
		if constexpr (FIR) {
			if (lane < 16) xt[lane] = xt[768 + lane]; // the tile's last 32 samples become the look-back (rows 0 and 48: no swizzle term)
			wave_sync(); // the window is in registers, the look-back is in place: the next tile may land
		}
		if (tile < tile_last) prefetch(tile + 1);
		c2 up[8], dn[8];
#pragma unroll
		for (int o = 0; o < 8; o++) {
			c2 acc = { 0.0f, 0.0f };
			if constexpr (FIR) {
#pragma unroll
				for (int k = 0; k < 26; k++) acc = acc + w[3 * o + 1 + k] * p.taps[k]; // x[3 i - 25 + k] (DSP.cpp:160-189, DSP.h:195-201)
			} else acc = w[o];
			const float rx_ = (o & 1) ? rot[o >> 1].z : rot[o >> 1].x, ry_ = (o & 1) ? rot[o >> 1].w : rot[o >> 1].y;
			const float RR = acc.x * rx_, II = acc.y * ry_, RI = acc.x * ry_, IR = acc.y * rx_; // DSP.cpp:296-316
			up[o] = c2{ RR - II, IR + RI };
			dn[o] = c2{ RR + II, IR - RI };
		}
		c2 a4[4], b4[4];
		reg_stage<8>(up, sup, a4); // DS2_a / DS2_b: Downsample2CIC5
		reg_stage<8>(dn, sdn, b4);
		// FilterCIC5 (DSP.cpp:132-157): out(m) from d(m-5 .. m), five levels of pair sums, * 1/32
		const auto fcic5 = [&](const c2 (&d4)[4], HaloState<4, c2>& st, float2* dst) {
			c2 h[5];
			get_halo<4, c2>(d4, st, h);
			c2 g[9];
#pragma unroll
			for (int i = 0; i < 5; i++) g[i] = h[i];
#pragma unroll
			for (int i = 0; i < 4; i++) g[5 + i] = d4[i];
#pragma unroll
			for (int lvl = 0; lvl < 5; lvl++) {
#pragma unroll
				for (int i = 0; i < 8 - lvl; i++) g[i] = g[i + 1] + g[i];
			}
			put_halo<4, c2>(d4, st, h);
			if (tile > tile_first) {
				float4* o4 = reinterpret_cast<float4*>(dst + (size_t)tile * 256 + lane * 4);
#pragma unroll
				for (int e = 0; e < 2; e++) {
					const c2 x0 = g[2 * e] * 0.03125f, x1 = g[2 * e + 1] * 0.03125f;
					o4[e] = make_float4(x0.x, x0.y, x1.x, x1.y);
				}
			}
		};
		fcic5(a4, sfa, outa);
		fcic5(b4, sfb, outb);
	}
	// two tiles are a window of each channel: the span's windows, both channels side by side (launch_k1k: spans of an even number of tiles)
	static_assert(sizeof(xt) >= 1024 * sizeof(float2), "k1k_wave: the spectral analysis works in 8 KB of the tile buffer");
	if (p.fz) wave_fft_tail(p, (size_t)rx * 2, true, (tile_first + 1) / 2, (tile_last - tile_first) / 2, reinterpret_cast<float2*>(xt));
}

// raw input rows -> complex float rows (Utilities/Convert.cpp:255-264 for CU8), for ladders without a CIC5 pre-pass
__global__ void k_convert_rows(const unsigned char* in, long long in_stride_bytes, int fmt, float2* dst, long long dst_stride, int n) {
	const int rx = blockIdx.y;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		float2 v;
		const unsigned char* row = in + (size_t)rx * in_stride_bytes;
		if (fmt == 1) {
			const unsigned char* u = row + 2 * (size_t)i;
			v = make_float2((float)((int)u[0] - 128) * 0.0078125f, (float)((int)u[1] - 128) * 0.0078125f);
		} else if (fmt == 2) {
			const signed char* u = reinterpret_cast<const signed char*>(row) + 2 * (size_t)i;
			v = make_float2((float)(int)u[0] * 0.0078125f, (float)(int)u[1] * 0.0078125f);
		} else if (fmt == 3) {
			const short* u = reinterpret_cast<const short*>(row) + 2 * (size_t)i;
			v = make_float2((float)(int)u[0] * 0.000030517578125f, (float)(int)u[1] * 0.000030517578125f);
		} else v = reinterpret_cast<const float2*>(row)[i];
		dst[(size_t)rx * dst_stride + i] = v;
	}
}

// DownsampleMovingAverage (DSP/DSP.cpp:60-82, `-go MA on`) at an integer ratio m = sample_rate / 96000: every output is the sum of
// its own m inputs, accumulated from zero in input order (D += data[i]), divided by (float) m -- complex / real, two true
// divisions.  Outputs are independent of each other, so one thread per output; a thread's m inputs are contiguous (m * 8 bytes of
// CF32: whole cache lines from 1536 kSPS on).
__global__ void k_ma_rows(const unsigned char* in, long long in_stride_bytes, int fmt, int m, float2* dst, long long dst_stride, int n) {
	const int rx = blockIdx.y;
	const unsigned char* row = in + (size_t)rx * in_stride_bytes;
	const float fm = (float)m;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		float dr = 0.0f, di = 0.0f;
		const size_t i0 = (size_t)i * (size_t)m;
		if (fmt == 0) {
			const float2* x = reinterpret_cast<const float2*>(row) + i0;
			if ((m & 1) == 0) { // 16-byte loads
				const float4* x4 = reinterpret_cast<const float4*>(x);
				for (int j = 0; j < m / 2; j++) { const float4 v = x4[j]; dr += v.x; di += v.y; dr += v.z; di += v.w; }
			} else for (int j = 0; j < m; j++) { const float2 v = x[j]; dr += v.x; di += v.y; }
		} else if (fmt == 1) { // Utilities/Convert.cpp:255-264
			const unsigned char* u = row + 2 * i0;
			for (int j = 0; j < m; j++) { dr += (float)((int)u[2 * j] - 128) * 0.0078125f; di += (float)((int)u[2 * j + 1] - 128) * 0.0078125f; }
		} else if (fmt == 2) {
			const signed char* u = reinterpret_cast<const signed char*>(row) + 2 * i0;
			for (int j = 0; j < m; j++) { dr += (float)(int)u[2 * j] * 0.0078125f; di += (float)(int)u[2 * j + 1] * 0.0078125f; }
		} else {
			const short* u = reinterpret_cast<const short*>(row) + 2 * i0;
			for (int j = 0; j < m; j++) { dr += (float)(int)u[2 * j] * 0.000030517578125f; di += (float)(int)u[2 * j + 1] * 0.000030517578125f; }
		}
		dst[(size_t)rx * dst_stride + i] = make_float2(__fdiv_rn(dr, fm), __fdiv_rn(di, fm));
	}
}

// copy rows of float2 (history carry of the pre-decimated stream)
__global__ void k_copy_rows(const float2* src, long long src_stride, float2* dst, long long dst_stride, int n) {
	const int rx = blockIdx.y;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
		dst[(size_t)rx * dst_stride + i] = src[(size_t)rx * src_stride + i];
}

// K1b: keep the last tile of the block as history for the next block's warm-up tile
__global__ void k1_tail(const unsigned char* in, long long in_stride_bytes, long long block_bytes,
                        unsigned char* hist, int tail_bytes) {
	const int rx = blockIdx.y;
	const uint4* src = (const uint4*)(in + (size_t)rx * in_stride_bytes + block_bytes - tail_bytes);
	uint4* dst = (uint4*)(hist + (size_t)rx * tail_bytes);
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tail_bytes / 16; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------
// K2a: SquareFreqOffsetCorrection analysis (DSP/DSP.cpp:417-456 + 475-489, DSP/FFT.h:94-129).
// One wave per 512-sample window: x^2 scattered bit-reversed into LDS, radix-2 DIT with the
// reference's float twiddle table (every butterfly is the same three complex ops as the
// reference's; butterflies of one stage are independent, so lane-parallel evaluation is exact),
// |X| via the glibc-equivalent hypot, the *sequential* float prefix sum (one lane, 511 adds: a
// parallel scan would round differently), then the two first-maximum searches as wave reductions.
// ------------------------------------------------------------------------------------------
// Wave-wide "first maximum": the lane values v >= 0 (sums of magnitudes; a lane without a candidate holds `none`, which is below every
// candidate) with their indices i; afterwards every lane holds the largest v and, among the lanes that had it, the smallest i -- what
// the reference's upward scan with a strict '>' finds.  For non-negative floats the IEEE order is the order of the bit patterns, so the
// maximum is an unsigned integer maximum, taken in six DPP steps (two inside the quads, half-row and row mirror, then the row
// broadcasts 15 and 31 that gfx9 has: the result arrives in lane 63) plus a v_readlane; the index is a second reduction (minimum
// over the lanes that hold the maximum).  Rounds 1-5 did this with ds_bpermute butterflies: 12 LDS round trips and ~120 instructions
// with branches per reduction, a third of the spectral analysis' instructions (profiles/r06_expB_fft_tail_argmax.txt).
// Precondition: no lane holds NaN or a negative value other than `none` (the callers' `v > best` never lets a NaN in).
template <bool IS_MIN>
__device__ __forceinline__ unsigned wave_reduce_u32(unsigned k) {
	const auto op = [](unsigned a, unsigned b) { return IS_MIN ? (a < b ? a : b) : (a > b ? a : b); };
	k = op(k, (unsigned)__builtin_amdgcn_update_dpp((int)k, (int)k, 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
	k = op(k, (unsigned)__builtin_amdgcn_update_dpp((int)k, (int)k, 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
	k = op(k, (unsigned)__builtin_amdgcn_update_dpp((int)k, (int)k, 0x141, 0xF, 0xF, false)); // row_half_mirror
	k = op(k, (unsigned)__builtin_amdgcn_update_dpp((int)k, (int)k, 0x140, 0xF, 0xF, false)); // row_mirror: every lane has its row's result
	k = op(k, (unsigned)__builtin_amdgcn_update_dpp((int)k, (int)k, 0x142, 0xA, 0xF, false)); // row_bcast:15 into rows 1 and 3
	k = op(k, (unsigned)__builtin_amdgcn_update_dpp((int)k, (int)k, 0x143, 0xC, 0xF, false)); // row_bcast:31 into rows 2 and 3
	return (unsigned)__builtin_amdgcn_readlane((int)k, 63);
}
__device__ __forceinline__ void wave_argmax_first(float& v, int& i, float none) {
#ifdef ABL_NO_ARGMAX
	return;
#endif
	// key 0: no candidate; a candidate's key is its bit pattern + 1 (so that a candidate of +0.0 still beats "none")
	const unsigned key = v > none ? __float_as_uint(v) + 1u : 0u;
	const unsigned m = wave_reduce_u32<false>(key);
	const unsigned first = wave_reduce_u32<true>(key == m ? (unsigned)i : 0x7fffffffu);
	v = m ? __uint_as_float(m - 1u) : none;
	i = (int)first;
}

// K2a is two kernels.
//
// k2_fft_mag: one wave per FFT_NW consecutive windows.  The reference's radix-2 DIT butterflies (FFT.h:104-129:
// t = Omega[j * (N >> (s+1))] * x[hi]; x[hi] = x[lo] - t; x[lo] += t) are evaluated unchanged, but three stages
// at a time in registers: a lane owns the 8 points whose indices differ in bits {0,1,2}, then {3,4,5}, then
// {6,7,8}; between the three passes the points change lanes through a padded (conflict-free) LDS buffer.  The
// 14 lane-dependent twiddles stay in registers for all windows of the wave.  |X| goes through the
// glibc-equivalent hypot, is staged in LDS for the wave's windows and leaves transposed: magT[W / 64][q][W % 64]
// with q = (bin + 256) % 512, so that the search kernel's lanes (one per window) read consecutive floats.
//
// k2_cgf_search: the order-sensitive part -- the float prefix sum and the two first-maximum searches, which the
// reference evaluates strictly left to right -- one LANE per window, 64 windows per wave, branch-free.
#ifndef FFT_NW_
#define FFT_NW_ 8
#endif
constexpr int FFT_NW = FFT_NW_;                 // windows per wave: the staging buffer below bounds the occupancy (LDS)
constexpr int MAG_STRIDE = 512 + 64 / FFT_NW;   // floats; 64/NW (mod 64) banks between the windows of a wave

// twiddle o with its rotated copy (-o.y, o.x): o * c = c.xx * o + c.yy * (-o.y, o.x) = (o.x c.x - o.y c.y, o.x c.y + o.y c.x),
// the same two products and one addition per component as std::complex's operator* (x - y == x + (-y) exactly)
// FFT_SLIM (default): only o stays in registers and the rotated copy is rebuilt where it is used (two sign/move operations per
// twiddle and window): 96 instead of 124 VGPRs, so that a wave of this kernel still fits on a SIMD that already holds a
// derotation/FIR wave and three PhaseSearch waves (it sits on the front stream: whatever delays it delays the next front end).
struct Tw { c2 o; };
__device__ __forceinline__ Tw make_tw(float2 o) { return Tw{ c2{ o.x, o.y } }; }
__device__ __forceinline__ c2 cmul_tw(const Tw& w, c2 c) { const c2 r = c2{ -w.o.y, w.o.x }; return c.xx * w.o + c.yy * r; }
__device__ __forceinline__ void tw_pin(Tw& w) { asm volatile("" : "+v"(w.o)); } // keeps the rotated copy from being hoisted out of the window loop

// three radix-2 stages on the 8 points of a lane; tw0: twiddle of the first stage (all 4 butterflies), tw1[b]:
// second stage for local index bit 0 = b, tw2[c]: third stage for local index bits (1,0) = c
__device__ __forceinline__ void fft_pass(c2 (&v)[8], const Tw& tw0, const Tw (&tw1)[2], const Tw (&tw2)[4]) {
#pragma unroll
	for (int r = 0; r < 8; r += 2) {
		const c2 t = cmul_tw(tw0, v[r + 1]);
		v[r + 1] = v[r] - t;
		v[r] = v[r] + t;
	}
#pragma unroll
	for (int r = 0; r < 8; r++) {
		if (r & 2) continue;
		const c2 t = cmul_tw(tw1[r & 1], v[r + 2]);
		v[r + 2] = v[r] - t;
		v[r] = v[r] + t;
	}
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const c2 t = cmul_tw(tw2[r], v[r + 4]);
		v[r + 4] = v[r] - t;
		v[r] = v[r] + t;
	}
}

// hypot_ref for the FFT bins: the same correctly rounded double sqrt (rsq seed + the Goldschmidt/Newton fma chain the
// compiler emits for sqrt(double)), minus its denormal-range rescaling, which x*x + y*y of two floats can never need:
// a non-zero sum is >= 2^-298.  A zero sum is lifted to 2^-600, whose root still converts to 0.0f.
__device__ __forceinline__ float hypot_bins(float x, float y) {
	const double dx = (double)x, dy = (double)y;
	double s = __builtin_fma(dy, dy, dx * dx); // dx * dx is exact, so this is the one rounding of dx*dx + dy*dy
	s = __builtin_fmax(s, 0x1p-600);
	const double y0 = __builtin_amdgcn_rsq(s);
	const double g0 = s * y0, h0 = y0 * 0.5;
	const double r0 = __builtin_fma(-h0, g0, 0.5);
	const double g1 = __builtin_fma(g0, r0, g0), h1 = __builtin_fma(h0, r0, h0);
	const double d0 = __builtin_fma(-g1, g1, s);
	const double g2 = __builtin_fma(d0, h1, g1);
	const double d1 = __builtin_fma(-g2, g2, s);
	return (float)__builtin_fma(d1, h1, g2);
}

// debug/self-test: count inputs on which hypot_bins differs from hypot_ref (tests only)
__global__ void k_selftest_hypot(const float2* in, int n, unsigned* mismatches) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float a = hypot_ref(in[i].x, in[i].y), b = hypot_bins(in[i].x, in[i].y);
	if (__float_as_uint(a) != __float_as_uint(b)) atomicAdd(mismatches, 1u);
}

// the lane-dependent twiddles of the three register passes (pass 1's are the same for every lane: scalar loads)
struct FftTwiddles { Tw a0, a1[2], a2[4], b0, b1[2], b2[4], c0, c1[2], c2_[4]; };
__device__ __forceinline__ FftTwiddles fft_twiddles(const float2* omega, int lane) {
	const int l7 = lane & 7;
	const auto tw = [&](int idx) { return make_tw(omega[idx]); };
	FftTwiddles t;
	t.a0 = tw(0); t.a1[0] = tw(0); t.a1[1] = tw(128); t.a2[0] = tw(0); t.a2[1] = tw(64); t.a2[2] = tw(128); t.a2[3] = tw(192);
	t.b0 = tw(l7 << 5); t.b1[0] = tw(l7 << 4); t.b1[1] = tw((l7 + 8) << 4);
	t.b2[0] = tw(l7 << 3); t.b2[1] = tw((l7 + 8) << 3); t.b2[2] = tw((l7 + 16) << 3); t.b2[3] = tw((l7 + 24) << 3);
	t.c0 = tw(lane << 2); t.c1[0] = tw(lane << 1); t.c1[1] = tw((lane + 64) << 1);
	t.c2_[0] = tw(lane); t.c2_[1] = tw(lane + 64); t.c2_[2] = tw(lane + 128); t.c2_[3] = tw(lane + 192);
	return t;
}
// lane's sample r of a window: bit-reversed storage (DSP.cpp:480): position 8*lane + r <- sample brev6(lane) + 64*brev3(r)
__device__ __forceinline__ int fft_src_lane(int lane) { return (int)(__brev((unsigned)lane) >> 26); }
__device__ __forceinline__ int fft_src_step(int r) { return 64 * (((r & 1) << 2) | (r & 2) | (r >> 2)); }

// One 512-point window by one wave: v[r] holds the lane's 8 squared samples on entry, m[r] = |X[lane + 64 r]| on exit.
// X: 584 float2 of wave-private LDS.  A one-wave workgroup orders its LDS exchanges with wavefront-scope fences (wave_sync);
// __syncthreads() would also wait for whatever global loads the caller has in flight.
__device__ __forceinline__ void fft512_mag(c2 (&v)[8], float2* X, FftTwiddles& t, int lane, float (&m)[8]) {
	const int l7 = lane & 7, l8 = lane >> 3;
	tw_pin(t.b0); tw_pin(t.b1[0]); tw_pin(t.b1[1]); tw_pin(t.c0); tw_pin(t.c1[0]); tw_pin(t.c1[1]);
#pragma unroll
	for (int r = 0; r < 4; r++) { tw_pin(t.b2[r]); tw_pin(t.c2_[r]); }
	fft_pass(v, t.a0, t.a1, t.a2); // stages 0-2: position 8*lane + r
#pragma unroll
	for (int r = 0; r < 8; r++) X[9 * lane + r] = make_float2(v[r].x, v[r].y); // index P + (P >> 3)
	wave_sync();
#pragma unroll
	for (int r = 0; r < 8; r++) { const float2 d = X[l7 + 9 * r + 72 * l8]; v[r] = c2{ d.x, d.y }; }
	wave_sync();
	fft_pass(v, t.b0, t.b1, t.b2); // stages 3-5: position l7 + 8*r + 64*l8
#pragma unroll
	for (int r = 0; r < 8; r++) X[l7 + 8 * r + 72 * l8] = make_float2(v[r].x, v[r].y); // index P + 8 * (P >> 6)
	wave_sync();
#pragma unroll
	for (int r = 0; r < 8; r++) { const float2 d = X[lane + 72 * r]; v[r] = c2{ d.x, d.y }; }
	wave_sync();
	fft_pass(v, t.c0, t.c1, t.c2_); // stages 6-8: bin lane + 64*r
#pragma unroll
	for (int r = 0; r < 8; r++) m[r] = hypot_bins(v[r].x, v[r].y);
}
__device__ __forceinline__ void fft_square(const float2 (&d)[8], c2 (&v)[8]) {
#pragma unroll
	for (int r = 0; r < 8; r++) v[r] = c2{ d[r].x * d[r].x - d[r].y * d[r].y, d[r].x * d[r].y + d[r].y * d[r].x }; // data[i] * data[i]
}

__global__ __launch_bounds__(64) void k2_fft_mag(K2Params p) {
	__shared__ __attribute__((aligned(16))) float2 X[584];
	__shared__ __attribute__((aligned(16))) float mag[FFT_NW * MAG_STRIDE];

	__builtin_amdgcn_s_setprio(1);
	const int lane = threadIdx.x;
	const int W0 = blockIdx.x * FFT_NW, n_win_total = p.n_chan * p.n_windows;
	FftTwiddles t = fft_twiddles(p.omega, lane);
	const int src = fft_src_lane(lane);

	// the 8 samples of a lane for window W0 + wi; the next window's are requested before this one's butterflies start, so the
	// wave (there are only one or two per SIMD: LDS) does not sit through a memory round trip per window
	float2 dn[8];
	const auto fetch = [&](int wi) {
		int W = W0 + wi;
		W = W < n_win_total ? W : n_win_total - 1;
		const int chan = W / p.n_windows, w = W - chan * p.n_windows;
		const float2* x = p.c48 + (size_t)chan * p.c48_stride + (size_t)w * 512 + src;
#pragma unroll
		for (int r = 0; r < 8; r++) dn[r] = x[fft_src_step(r)];
	};
	fetch(0);
	for (int wi = 0; wi < FFT_NW; wi++) {
		const int W = W0 + wi;
		if (W >= n_win_total) break;
		c2 v[8];
		fft_square(dn, v);
		__builtin_amdgcn_sched_barrier(0); // (the scheduler would otherwise sink the requests to the end of the iteration)
		if (wi + 1 < FFT_NW) fetch(wi + 1);
		__builtin_amdgcn_sched_barrier(0);
		float m[8];
		fft512_mag(v, X, t, lane, m);
		float* mg = mag + wi * MAG_STRIDE;
#pragma unroll
		for (int r = 0; r < 8; r++) mg[(lane + 64 * r + 256) & 511] = m[r];
	}
	wave_sync();
	// transposed store: 64/NW bins x NW windows per instruction, 4*NW contiguous bytes per bin
	const int wl = lane % FFT_NW, qs = lane / FFT_NW;
	const int W = W0 + wl;
	if (W < n_win_total) {
		float* dst = p.magT + (size_t)(W >> 6) * (512 * 64) + (W & 63);
		const float* mg = mag + wl * MAG_STRIDE;
#pragma unroll 8
		for (int it = 0; it < 8 * FFT_NW; it++) {
			const int q = it * (64 / FFT_NW) + qs;
			dst[(size_t)q * 64] = mg[q];
		}
	}
}

// SquareFreqOffsetCorrection::correctFrequency (DSP.cpp:426-456) by ONE WAVE for the windows whose magnitudes it holds in
// registers (NWIN of them side by side: their dependent chains interleave).  m[w][r] = |X[lane + 64 r]| of window w.
//  * shifted index q = (bin + 256) % 512 lives at lane q % 64 of register q / 64;
//  * cumsum[q] = cumsum[q - 1] + mag[q] is evaluated in exactly that order: a chain of 511 dependent additions per window, by one
//    lane per window through LDS (round 5; a wave-wide DPP shift-add chain before);
//  * the 379 candidates of the wide search and the 36 of the second search are then independent: cumsum and magnitudes go
//    through LDS once (S: 1024 floats per window), every lane evaluates its candidates with the reference's expression, and
//    "first maximum under a strict >" is a wave reduction (larger value wins, ties go to the lower index; NaN never wins,
//    like `v > best`).
// Returns fz (DSP.cpp:449-456: N/2 - (i + delta/2), -1 without a positive peak) in every lane.
template <int NWIN>
__device__ __forceinline__ void spectral_search(const float (&m)[NWIN][8], float* S, int lane, int wide, int (&fz)[NWIN]) {
	wave_sync(); // the FFT's last exchange has been read
#pragma unroll
	for (int w = 0; w < NWIN; w++)
#pragma unroll
		for (int q8 = 0; q8 < 8; q8++) S[1024 * w + 512 + lane + 64 * q8] = m[w][(q8 + 4) & 7]; // shifted magnitudes: index q at lane q % 64 of register q / 64
	wave_sync();
	// The prefix sum, one LANE per window: cumsum[0] = 0, cumsum[q] = cumsum[q - 1] + mag[q] in exactly that order -- 511 dependent
	// additions whoever does them.  Rounds 1-4 ran the chain as a wave-wide DPP shift-add (one instruction advances it by one element,
	// 63 steps per register, two windows interleaved): 3 issue slots per element and window pair, 12 M VALU instructions per launch, an
	// eighth of the front end's -- and the front end shares its SIMDs' issue slots with PhaseSearch.  A lane that walks its
	// window through LDS needs 1.5 per element for ALL windows of the wave together (128-bit reads and writes, four additions each).
	if (lane < NWIN) {
		float4* Cs4 = reinterpret_cast<float4*>(S + 1024 * lane);
		const float4* Ms4 = reinterpret_cast<const float4*>(S + 1024 * lane + 512);
		float c = 0.0f;
#pragma unroll 4
		for (int q4 = 0; q4 < 128; q4++) {
			const float4 v = Ms4[q4];
			float4 o;
			o.x = q4 == 0 ? 0.0f : c + v.x; // cumsum[0] = 0 (DSP.cpp:431)
			o.y = o.x + v.y;
			o.z = o.y + v.z;
			o.w = o.z + v.w;
			c = o.w;
			Cs4[q4] = o;
		}
	}
	wave_sync();
	constexpr int BIG = 0x7fffffff;
#pragma unroll
	for (int w = 0; w < NWIN; w++) {
		const float* Cs = S + 1024 * w;
		const float* Ms = Cs + 512;
		int wi = 0;
		if (wide) { // v(i) = cumsum[i + M] - cumsum[i] + 0.6f * (mag[i + ofs] + mag[i + ofs + delta]), i = 0 .. 378 (DSP.cpp:426-447)
			float best = -1.0f;
			int bi = BIG;
#pragma unroll
			for (int t = 0; t < 6; t++) {
				const int i = lane + 64 * t;
				if (i <= 512 - 134) {
					const float v = Cs[i + 133] - Cs[i] + 0.6f * (Ms[i + 15] + Ms[i + 117]);
					if (v > best) { best = v; bi = i; }
				}
			}
			wave_argmax_first(best, bi, -1.0f);
			if (!(best > -1.0f)) bi = 0;
			wi = bi + 66 - 256; // wi + M/2 - N/2
		}
		// second search (DSP.cpp:449-456): i in [wi + 187, wi + 223), first maximum above 0
		float h = 0.0f;
		int hidx = BIG;
		if (lane < 36) {
			const int i = wi + 187 + lane;
			const float v = Ms[(i + 512) & 511] + Ms[(i + 102 + 512) & 511];
			if (v > 0.0f) { h = v; hidx = i; }
		}
		wave_argmax_first(h, hidx, 0.0f);
		fz[w] = (h > 0.0f) ? (205 - hidx) : -1;
	}
}

// The whole spectral analysis riding at the end of a front-end wave (k1_dpp): a span of 16 tiles is 512 samples of each 48 kHz
// channel, i.e. one window of SquareFreqOffsetCorrection per channel (p.fft_windows of them for longer spans), all written by
// this very wave a moment ago.  FFT, magnitudes, prefix sum and both peak searches happen in the wave's registers and LDS; what
// leaves is fz and ppm of the window -- 8 bytes instead of 2 KB of magnitudes -- so neither the FFT kernel (0.05 ms alone,
// 0.16-0.20 ms on the front stream next to the previous block's back end), nor a search kernel, nor 0.2 GB of magnitude
// traffic per step remain, and the arithmetic fills issue slots of a kernel that waits for HBM most of the time.
// the analysis of two windows side by side (their dependent chains interleave): window wa of c48 row ra and window wb of row rb
__device__ __forceinline__ void k1_fft_pair(const K1Params& p, size_t ra, int wa, size_t rb, int wb, float2* X, FftTwiddles& t, int lane) {
	const int src = fft_src_lane(lane);
	float* S = reinterpret_cast<float*>(X); // 2 x 1024 floats for the searches; the FFT's exchange buffer (584 float2) is the front of it
	float2 d[2][8];
#pragma unroll
	for (int ch = 0; ch < 2; ch++) {
		const float2* x = p.c48 + (ch ? rb : ra) * p.c48_stride + (size_t)(ch ? wb : wa) * 512 + src;
#pragma unroll
		for (int r = 0; r < 8; r++) d[ch][r] = x[fft_src_step(r)];
	}
	float m[2][8];
#pragma unroll
	for (int ch = 0; ch < 2; ch++) {
		c2 v[8];
		fft_square(d[ch], v);
		fft512_mag(v, X, t, lane, m[ch]);
	}
	int fz[2];
	spectral_search<2>(m, S, lane, p.wide, fz);
	if (lane < 2) {
		const int f = lane == 0 ? fz[0] : fz[1];
		const size_t W = (lane == 0 ? ra : rb) * p.n_windows + (lane == 0 ? wa : wb);
		p.fz[W] = f;
		p.ppm[W] = p.ppm_table[f + 205];
	}
}
__device__ __forceinline__ void k1_fft_tail(const K1Params& p, int rx, int span, float2* X) {
	const int lane = threadIdx.x;
	// This wave's own c48 stores must have reached L2 (its L1 never held those lines), and the last tile's LDS-DMA must have
	// landed before the tile buffer is reused: a workgroup-scope fence is exactly "s_waitcnt vmcnt(0)" -- an agent-scope one
	// (__threadfence) adds an L2 write-back and an L1 invalidate per span, which cost more than the analysis itself.
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	FftTwiddles t = fft_twiddles(p.omega, lane);
	const int nw = p.fft_windows;
	for (int i = 0; i < nw; i++) {
		const int w = span * nw + i;
		k1_fft_pair(p, (size_t)rx * 2, w, (size_t)rx * 2 + 1, w, X, t, lane);
	}
}
// the spectral analysis at the end of the waves of k1x_wave / k1k_wave (round 6, late): the windows [w0, w0 + nw) this wave has just written,
// of ONE c48 row in pairs of consecutive windows (mode X: nw even), or of the rows row, row + 1 window by window (two channels)
template <class P>
__device__ __forceinline__ void wave_fft_tail(const P& q, size_t row, bool two_rows, int w0, int nw, float2* X) {
	const int lane = threadIdx.x;
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // (see k1_fft_tail)
	K1Params p{};
	p.c48 = q.c48; p.c48_stride = q.c48_stride; p.omega = q.omega; p.ppm_table = q.ppm_table; p.fz = q.fz; p.ppm = q.ppm; p.n_windows = q.n_windows; p.wide = q.wide;
	FftTwiddles t = fft_twiddles(p.omega, lane);
	if (two_rows) { for (int i = 0; i < nw; i++) k1_fft_pair(p, row, w0 + i, row + 1, w0 + i, X, t, lane); }
	else { for (int i = 0; i < nw; i += 2) k1_fft_pair(p, row, w0 + i, row, w0 + i + 1, X, t, lane); }
}

// The same analysis as a kernel of its own, one wave per (receiver, window), for the front ends that do not end in k1_dpp (the resampled
// and decimate-by-3 ladders, mode X): it replaces k2_fft_mag + k2_cgf_search there -- 21 KB of LDS and 88 VGPRs per wave, 2 KB of
// magnitudes per window out and in again -- which beside the 6 MSPS ladder's pass over the input took 0.23 + 0.06 ms instead of
// 0.02 + 0.03 alone and were the longest link of that path's back-end chain (profiles/r04_config3_timeline.txt).
__global__ __launch_bounds__(64) void k2_fft_search_win(K1Params p) {
	__shared__ __attribute__((aligned(16))) float2 X[1024];
	k1_fft_tail(p, blockIdx.y, blockIdx.x, X); // (p.fft_windows = 1: "span" = window)
}

__global__ __launch_bounds__(64) void k2_cgf_search(K2Params p) {
	__builtin_amdgcn_s_setprio(3); // a few latency-bound waves (one lane per window) next to the HBM-bound passes: 0.26 -> 0.05 ms at 6 MSPS
	const int lane = threadIdx.x;
	const int W = blockIdx.x * 64 + lane, n_win_total = p.n_chan * p.n_windows;
	const float* mrow = p.magT + (size_t)blockIdx.x * (512 * 64); // wave-uniform: mrow[64 * q + lane] = |X[(q + 256) % 512]| of this lane's window
#define MG(q) mrow[64 * (q) + lane]
	__builtin_amdgcn_s_setprio(2);
	// ---- wide search (DSP.cpp:426-447), strictly sequential like the reference:
	// cumsum[i] = cumsum[i-1] + mag[i];  v(i) = cumsum[i+M] - cumsum[i] + 0.6f * (mag[i+ofs] + mag[i+ofs+delta]);
	// two running sums 133 apart reproduce cumsum[i+M] and cumsum[i] with the same additions in the same order.
	int wi = 0;
	if (p.wide) {
		float hi = 0.0f; // cumsum[0] = 0
#pragma unroll 7
		for (int i = 1; i <= 133; i++) hi = hi + MG(i);
		float lo_ = 0.0f, best = -1.0f;
		int bi = 0;
#pragma unroll 9
		for (int i = 0; i < 512 - 134; i++) { // 378 = 9 * 42 iterations, then the last candidate without the sum updates
			const float v = hi - lo_ + 0.6f * (MG(i + 15) + MG(i + 117));
			const bool gt = v > best;
			best = gt ? v : best;
			bi = gt ? i : bi;
			hi = hi + MG(i + 134); // cumsum[i+1+M]
			lo_ = lo_ + MG(i + 1); // cumsum[i+1]
		}
		{
			const int i = 512 - 134;
			const float v = hi - lo_ + 0.6f * (MG(i + 15) + MG(i + 117));
			bi = v > best ? i : bi;
		}
		wi = bi + 66 - 256; // wi + M/2 - N/2
	}
	// ---- second search (DSP.cpp:449-456): i in [wi+187, wi+223), first maximum above 0
	float h = 0.0f;
	int hidx = 0;
#pragma unroll 6
	for (int c = 0; c < 36; c++) {
		const int i = wi + 187 + c;
		const float v = MG((i + 512) & 511) + MG((i + 102 + 512) & 511);
		const bool gt = v > h;
		h = gt ? v : h;
		hidx = gt ? i : hidx;
	}
	if (W < n_win_total) {
		const int fz = (h > 0.0f) ? (205 - hidx) : -1; // fz = N/2 - (i + delta/2) = 205 - i (integer valued); default -1
		p.fz[W] = fz;
		p.ppm[W] = p.ppm_table[fz + 205];
	}
#undef MG
}

// ------------------------------------------------------------------------------------------
// K2b: CGF derotation phasors (DSP/DSP.cpp:457-466).  rot *= rot_step per sample, renormalised per
// window and carried across windows and blocks: a strictly sequential float recurrence of 24,576 steps
// per (receiver, channel) and block -- no re-association is allowed, so time cannot be parallelised.
// What CAN be done is to make the recurrence cost almost no issue slots: one LANE per chain (64
// chains per wave), 3 packed VALU ops per step (P = (rx*sx, rx*sy), Q = (ry*-sy, ry*sx), rot' = P + Q;
// x*-y == -(x*y) exactly, so this is the reference's (ac - bd, ad + bc) bit for bit), and the phasor of
// every step is stored time-major (rotT[n][chain]) so the 64 lanes write 512 contiguous bytes.
// The kernel is latency bound (~20 cycles per step) but occupies only n_chan/64 waves, so it hides
// behind the bandwidth-bound kernels of the next block; K2c applies the phasors in parallel.
// ------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(64) void k2_cgf_phasor(K2Params p) {
	const int lane = threadIdx.x;
	const int chan_raw = blockIdx.x * 64 + lane;
	const bool live = chan_raw < p.n_chan;
	const int chan = live ? chan_raw : p.n_chan - 1;
	const int L = p.n_windows * 512;

	// this handful of waves is the longest dependency chain of the whole pipeline: always issue first
	__builtin_amdgcn_s_setprio(3);
	(void)L;
	const float2 r0 = p.rot_state[chan];
	v2f cur = { r0.x, r0.y };
	// wave-uniform row base + lane: the store addresses are SGPR base + constant lane offset
	float2* const out = p.rotT + (size_t)blockIdx.x * 64; // padded columns exist for dead lanes
	const size_t stride = p.rotT_stride;
	for (int w = 0; w < p.n_windows; w++) {
		const int fz = p.fz[(size_t)chan * p.n_windows + w];
		const float2 stp = p.step_table[fz + 205];
		const v2f st = { stp.x, stp.y }, st_sw = { -stp.y, stp.x };
		float2* o = out + (size_t)w * 512 * stride;
#pragma unroll 8
		for (int k = 0; k < 512; k++) {
			cur = cur.xx * st + cur.yy * st_sw; // rot *= rot_step
			o[lane] = make_float2(cur.x, cur.y);
			o += stride;
		}
		const float a = hypot_ref(cur.x, cur.y); // rot /= std::abs(rot), once per window (DSP.cpp:465)
		cur.x = __fdiv_rn(cur.x, a);
		cur.y = __fdiv_rn(cur.y, a);
	}
	if (live) p.rot_state[chan] = make_float2(cur.x, cur.y);
}

// The same recurrence without the per-sample stores: only its state at the start of every window is kept (after the
// renormalisation).  3 packed VALU ops per step and nothing else -- in particular no store per few steps: beside the front end a
// store is acknowledged after ~20 us, a wave may have 63 memory operations outstanding, and 57 checkpoints per 4.8 us window
// stalled the recurrence at that limit (0.32 -> 0.45 ms beside the front end, the pipeline's longest chain).  The checkpoints
// inside the windows are k2_cgf_refine's.
__global__ __launch_bounds__(64) void k2_cgf_phasor_ck(K2Params p) {
	const int lane = threadIdx.x;
	const int chan_raw = blockIdx.x * 64 + lane;
	const bool live = chan_raw < p.n_chan;
	const int chan = live ? chan_raw : p.n_chan - 1;
	__builtin_amdgcn_s_setprio(3);
	const float2 r0 = p.rot_state[chan];
	v2f cur = { r0.x, r0.y };
	float2* const ckw = p.ck + (size_t)blockIdx.x * 64 + lane; // slot 0 of every window; padded columns exist for dead lanes
	// A window's step is two dependent loads (the window's frequency bin, then the table entry): fetched where they are needed they
	// were two memory round trips in front of every window's 512 steps -- 48 times a few microseconds next to the front end, a
	// third of the kernel.  The bin is requested two windows ahead and the table entry one window ahead.
	const int* fzrow = p.fz + (size_t)chan * p.n_windows;
	int fz_next = fzrow[p.n_windows > 1 ? 1 : 0];
	float2 stp_next = p.step_table[fzrow[0] + 205];
	for (int w = 0; w < p.n_windows; w++) {
		const float2 stp = stp_next;
		stp_next = p.step_table[fz_next + 205];
		fz_next = fzrow[w + 2 < p.n_windows ? w + 2 : p.n_windows - 1];
		const v2f st = { stp.x, stp.y }, st_sw = { -stp.y, stp.x };
		ckw[(size_t)w * CK_SLOTS * p.ck_stride] = make_float2(cur.x, cur.y); // state at the window start (after the renormalisation)
#pragma unroll 8
		for (int k = 0; k < 512; k++) cur = cur.xx * st + cur.yy * st_sw; // rot *= rot_step
		const float a = hypot_ref(cur.x, cur.y); // rot /= std::abs(rot), once per window (DSP.cpp:465)
		cur.x = __fdiv_rn(cur.x, a);
		cur.y = __fdiv_rn(cur.y, a);
	}
	if (live) p.rot_state[chan] = make_float2(cur.x, cur.y);
}

// The same once more with a chain on a PAIR of lanes (even lane: real part, odd lane: imaginary part; 32 chains per wave).
// The packed form above takes 24 cycles per step -- three packed operations of a dependent chain, 8 cycles each; here a step
// is own * c, partner * (-/+ s) with the partner's value through the DPP operand, and their sum: three plain operations, 4 cycles
// each.  Every product and sum is the one of the packed form (x c + y (-s); y c + x s = x s + y c), so the states are the same bits.
// Twice the waves: the launcher takes this form while every wave still has a SIMD of the reserved CUs to itself.  Measured alone,
// 512 chains x 48 windows: 183 us against 247 (18 against 24 cycles per step: the DPP operand wants two wait states behind the
// sum that wrote it, one of them an s_nop); with (own c, own s') as ONE packed product and the sum taking the partner's half through
// DPP: 267 us -- a packed operation is 8 cycles to its dependant (profiles/r04_expC_phasor_pairs.txt).
__global__ __launch_bounds__(64) void k2_cgf_phasor_ck_pairs(K2Params p) {
	const int lane = threadIdx.x, comp = lane & 1;
	const int chan_raw = blockIdx.x * 32 + (lane >> 1);
	const bool live = chan_raw < p.n_chan;
	const int chan = live ? chan_raw : p.n_chan - 1;
	__builtin_amdgcn_s_setprio(3);
	float own = reinterpret_cast<const float*>(p.rot_state)[(size_t)chan * 2 + comp];
	float* const ckw = reinterpret_cast<float*>(p.ck + (size_t)blockIdx.x * 32) + lane; // (padded columns exist for dead lanes)
	const int* fzrow = p.fz + (size_t)chan * p.n_windows;
	int fz_next = fzrow[p.n_windows > 1 ? 1 : 0];
	float2 stp_next = p.step_table[fzrow[0] + 205];
	for (int w = 0; w < p.n_windows; w++) {
		const float2 stp = stp_next;
		stp_next = p.step_table[fz_next + 205];
		fz_next = fzrow[w + 2 < p.n_windows ? w + 2 : p.n_windows - 1];
		const float c = stp.x, s = comp ? stp.y : -stp.y;
		ckw[(size_t)w * CK_SLOTS * p.ck_stride * 2] = own; // state at the window start (after the renormalisation)
#pragma unroll 16
		for (int k = 0; k < 512; k++) {
			const float u = own * c;
			const float partner = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own), 0xB1, 0xf, 0xf, true)); // quad_perm [1,0,3,2]
			own = u + partner * s; // rot *= rot_step
		}
		const float partner = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own), 0xB1, 0xf, 0xf, true));
		const float a = comp ? hypot_ref(partner, own) : hypot_ref(own, partner); // rot /= std::abs(rot), once per window (DSP.cpp:465)
		own = __fdiv_rn(own, a);
	}
	if (live) reinterpret_cast<float*>(p.rot_state)[(size_t)chan * 2 + comp] = own;
}

// The recurrence once more inside every window, all windows in parallel (one lane per (chain, window), 64 chains of one window
// per wave): from the window's start state, its state in front of every CK_SEG-th sample -- what lane i of k6_window_fir restarts
// from.  Time-major: a store is 512 contiguous bytes.  384 short waves; the stores of a wave are fewer than it may have outstanding.
__global__ __launch_bounds__(64) void k2_cgf_refine(K2Params p) {
	const int lane = threadIdx.x, w = blockIdx.y;
	const int chan_raw = blockIdx.x * 64 + lane;
	const int chan = chan_raw < p.n_chan ? chan_raw : p.n_chan - 1;
	float2* o = p.ck + (size_t)w * CK_SLOTS * p.ck_stride + (size_t)blockIdx.x * 64 + lane;
	const float2 r0 = o[0];
	const float2 stp = p.step_table[p.fz[(size_t)chan * p.n_windows + w] + 205];
	v2f cur = { r0.x, r0.y };
	const v2f st = { stp.x, stp.y }, st_sw = { -stp.y, stp.x };
#pragma unroll 8
	for (int i = 1; i < CK_USED; i++) {
#pragma unroll
		for (int k = 0; k < CK_SEG; k++) cur = cur.xx * st + cur.yy * st_sw; // rot *= rot_step
		o[(size_t)i * p.ck_stride] = make_float2(cur.x, cur.y);
	}
}

// ------------------------------------------------------------------------------------------
// K6: output[i] *= rot (DSP.cpp:457-466) + FilterComplex(Filters::Coherent) + ScatterPLL (DSP.cpp:215-246,
// DSP.h:95-117) in one pass, without the phasor array and without the derotated-sample array in HBM.
// One wave = ONE chain x one 512-sample window, lanes over time (round 4; rounds 1-3 had one lane per chain walking a segment
// with 20 samples of history in registers: 178 VGPRs, which never fitted beside three front-end waves and displaced one).
//  * lane i < 60 owns nine consecutive samples: it restarts the recurrence from the checkpoint in front of them (lanes 0..2: the
//    last three segments of the previous window, the FIR's history; in the block's first window the previous block's carried
//    samples instead), derotates them and leaves them in LDS -- every sample's arithmetic is exactly K2b/K2c's;
//  * a lane then owns one ScatterPLL group (two rounds: a window completes 102 or 103 groups): 21 derotated samples out of LDS,
//    the five 17-tap sums each left to right from 0 (DSP.h:224-230), the level, the (1j)^n pre-rotation of PhaseSearchEMA; the
//    groups of a window are consecutive elements of the channel's five rows: 512 contiguous bytes per store.
// The groups of window w are those whose LAST sample lies in it.  Workgroup ids put a channel's windows on one XCD, one behind the
// other (the rows' lines that two windows share meet in that L2).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ c2 pk_sub_add(c2 a, c2 b) { // (a.x - b.x, a.y + b.y) in one packed add (x - y == x + (-y) exactly)
	c2 r;
	asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

constexpr int K6_HL = 5;                    // segments of the previous window derotated again as the window's halo
constexpr int K6_HALO = K6_HL * CK_SEG - 1; // = 44 samples: segments 52..56 (the FIR needs 20; the FM branch 37: 36 of the Receiver filter + Demod::FM's prev)
static_assert(DF_HIST <= K6_HALO && DF_HIST >= FM_HIST + 1, "k6_window_fir: the carried tail is the first window's halo");
#ifndef K6_WAVES
#define K6_WAVES 8
#endif
__device__ __forceinline__ float atan2f_ref(float y, float x);
// FM (round 4, late): ModelChallenger's FM branch (Model.cpp:638-639) inside this kernel -- Demod::FM (Demod.cpp:27-37) and
// Filter(Receiver) (DSP.h:257-263) on the derotated window while it sits in LDS, the signs out as bits: the derotated samples never
// reach HBM (100 MB out, 100 MB in per step of 256 receivers) and k5_fm_filter's launch is gone.  Every value is k5_fm_filter's: the
// discriminator of sample n from samples n and n - 1, the 37-tap sum left to right; the 36 discriminator values in front of the window
// are computed again from the halo (in a block's first window: from the carried tail, whose first samples are zero before the stream
// starts -- atan2f(+0, +0) / pi = 0, the filter's zero history).
constexpr int K6_YBUF = K6_HALO + 512 + 6; // derotated samples of a window in LDS: halo + window + the over-read of the last group's window
// one (channel, window) by one wave (a one-wave workgroup, or one whose other waves do not share ybuf / s_fm: the barriers below
// are then ordering only)
template <bool FM>
__device__ __forceinline__ void k6_window_body(const K6Params& p, int chain, int w, float2* ybuf, float* s_fm) {
	const int lane = threadIdx.x & 63;
	const int W = p.n_windows;
	const float2* xrow = p.c48 + (size_t)chain * p.c48_stride;
	// ---- derotation: lane -> segment (lanes 0..2: the previous window's last three)
	{
		const int seg = lane - K6_HL;
		const int ws = seg >= 0 ? w : w - 1, si = seg >= 0 ? seg : CK_USED + seg;
		if (lane < CK_USED + K6_HL && ws >= 0) {
			const int n0 = ws * 512 + si * CK_SEG;
			const float2 r0 = p.ck[((size_t)ws * CK_SLOTS + si) * p.ck_stride + chain];
			const float2 stp = p.step_table[p.fz[(size_t)chain * W + ws] + 205];
			float2 d[CK_SEG];
#pragma unroll
			for (int m = 0; m < CK_SEG; m++) d[m] = xrow[n0 + m]; // (segment 56: the ninth is the next window's first sample, unused)
			c2 rot = { r0.x, r0.y };
			const c2 st = { stp.x, stp.y }, st_sw = { -stp.y, stp.x };
			float2* y = ybuf + K6_HALO + (n0 - w * 512);
#pragma unroll
			for (int m = 0; m < CK_SEG; m++) {
				rot = rot.xx * st + rot.yy * st_sw;                       // rot *= rot_step
				const c2 v = pk_sub_add(rot * d[m].x, rot.yx * d[m].y);   // data * rot = (d.x r.x - d.y r.y, d.x r.y + d.y r.x)
				if (m < CK_SEG - 1 || si < CK_USED - 1) y[m] = make_float2(v.x, v.y);
			}
		}
		if (w == 0 && lane < DF_HIST) ybuf[K6_HALO - DF_HIST + lane] = p.hist_in[(size_t)chain * DF_HIST + lane]; // the previous block's tail
	}
	__syncthreads(); // (one wave: ordering)
	if (w == W - 1 && lane < DF_HIST && p.hist_out) p.hist_out[(size_t)chain * DF_HIST + lane] = ybuf[K6_HALO + 512 - DF_HIST + lane];
	if constexpr (FM) {
		// discriminator value i = sample (i - FM_HIST) of the window: data[i] * std::conj(prev) -> atan2f / pi, as in k5_fm_filter
		for (int i = lane; i < FM_HIST + 512; i += 64) {
			const float2 d = ybuf[K6_HALO - FM_HIST + i], pv = ybuf[K6_HALO - FM_HIST + i - 1];
			const float npi = -pv.y;
			const float re = d.x * pv.x - d.y * npi;
			const float im = d.x * npi + d.y * pv.x;
			s_fm[i] = __fdiv_rn(atan2f_ref(im, re), 3.14159265358979323846f);
		}
		__syncthreads(); // (one wave: ordering)
		uint32_t* o = p.fmbits + (size_t)chain * p.fmbits_stride + w * 16;
#pragma unroll 1
		for (int q = 0; q < 8; q++) {
			float acc = 0.0f;
#pragma unroll
			for (int i = 0; i < 37; i++) acc += p.fm_taps[i] * s_fm[q * 64 + lane + i]; // x += taps[i] * *data++ (DSP.h:257-263)
			const unsigned long long b = __ballot(acc > 0);
			if (lane == 0) { o[2 * q] = (uint32_t)b; o[2 * q + 1] = (uint32_t)(b >> 32); }
		}
	}
	// ---- FIR + ScatterPLL: block-local group gl covers samples n_rel0 + 5 gl .. + 4
	const int g_lo = (w * 512 - p.n_rel0) / 5;                                              // ceil((512 w - 4 - n_rel0) / 5), numerator + 4 >= 0
	const int g_end = w == W - 1 ? p.n_groups : min(p.n_groups, ((w + 1) * 512 - p.n_rel0) / 5);
	const int r0sel = (int)(p.first_group & 3);
	for (int gl = g_lo + lane; gl < g_end; gl += 64) {
		const int a = p.n_rel0 + 5 * gl - w * 512; // the group's first sample relative to the window: -4 .. 507
		const float2* wsrc = ybuf + K6_HALO + a - 16;
		// the five 17-tap sums of the group advance together over its 21 samples -- each still x += taps[i] * d[i] from i = 0 upwards
		// (DSP.h:224-230) -- so a sample is dead once its five products are formed: ten accumulator registers instead of a 42-register
		// window (round 5: the kernel then fits beside a PhaseSearch wave in the registers three front-end waves leave on a SIMD)
		c2 accs[5] = { { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f } };
#pragma unroll
		for (int i = 0; i < 21; i++) {
			const float2 v = wsrc[i];
			const c2 x = c2{ v.x, v.y };
#pragma unroll
			for (int j = 0; j < 5; j++)
				if (i - j >= 0 && i - j < 17) accs[j] = accs[j] + x * p.taps[i - j];
		}
		// PhaseSearchEMA multiplies symbol n of a chain by (1j)^(n & 3) with swaps/negations (Demod.cpp:44-61); every chain has
		// consumed exactly first_group + gl symbols, so that exact rotation is applied here: rot 1: (-y, x)  rot 2: (-x, -y)  rot 3: (y, -x)
		const int rsel = (r0sel + gl) & 3;
		const unsigned nx = (rsel == 1 || rsel == 2) ? 0x80000000u : 0u, ny = rsel >= 2 ? 0x80000000u : 0u;
		const bool swap = (rsel & 1) != 0;
		float level = 0.0f;
		float2* srow = p.sym + sym_row_base(chain, 0, p.sym_stride) + gl;
#pragma unroll
		for (int j = 0; j < 5; j++) {
			const c2 acc = accs[j];
			level = level + (acc.x * acc.x + acc.y * acc.y); // std::norm
			const float sx = swap ? acc.y : acc.x, sy = swap ? acc.x : acc.y;
			srow[(size_t)j * p.sym_stride] = make_float2(__uint_as_float(__float_as_uint(sx) ^ nx), __uint_as_float(__float_as_uint(sy) ^ ny));
		}
		p.lvl[(size_t)chain * p.sym_stride + gl] = __fdiv_rn(level, 5.0f);
	}
}

template <bool FM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FM ? 6 : K6_WAVES, FM ? 6 : K6_WAVES))) void k6_window_fir(K6Params p) { // (38 / 39 registers since round 5; the FM form's scalar registers allow six waves per SIMD)
	__shared__ __attribute__((aligned(16))) float2 ybuf[K6_YBUF];
	__shared__ float s_fm[FM ? FM_HIST + 512 : 1];
	// id = (((chain / 64) W + w) 8 + chain % 8) 8 + (chain / 8) % 8: the XCD (id % 8) depends on the chain alone, a chain's windows are
	// 64 ids apart, and the eight chains that share a 64-byte run of the time-major checkpoints are neighbours on one XCD
	const int W = p.n_windows;
	const int t = blockIdx.x >> 6, w = t % W, chain = (t / W) * 64 + (blockIdx.x & 7) * 8 + ((blockIdx.x >> 3) & 7);
	if (chain >= p.n_chan) return;
	k6_window_body<FM>(p, chain, w, ybuf, s_fm);
}

// carry the tail of the previous block's derotated samples (FIR-17 history + partial ScatterPLL group) to the
// front of each row, before K2c overwrites the row (same stream: K3 of the previous block has finished)
__global__ __launch_bounds__(64) void k2_cgf_carry(K2Params p) {
	const int lane = threadIdx.x;
	const int L = p.n_windows * 512;
	float2* y = p.cgf + (size_t)blockIdx.x * p.cgf_stride;
	if (lane < CGF_HIST) y[lane] = y[L + lane];
}

// K2c: output[i] *= rot (DSP.cpp:463), fully parallel: a 64-sample x 64-chain tile of phasors is read
// time-major (coalesced along chains), transposed through LDS, multiplied into the chain-major 48 kHz
// samples (coalesced along time) and written to the CGF output rows.
__global__ __launch_bounds__(256) void k2_cgf_apply(K2Params p) {
	__shared__ float2 tile[64][65];
	const int t = threadIdx.x;
	const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
	if (blockIdx.x == gridDim.x - 1) {
		// the workgroup that overwrites the row's last CGF_HIST samples first carries them (the previous block's tail: FIR-17
		// history + a partial ScatterPLL group) to the front of the row; nobody else touches either range in this launch
		const int L = p.n_windows * 512;
#pragma unroll
		for (int q = 0; q < 16; q++) {
			const int c = c0 + (t >> 6) + 4 * q;
			if (c < p.n_chan) {
				float2* y = p.cgf + (size_t)c * p.cgf_stride;
				y[t & 63] = y[L + (t & 63)];
			}
		}
		__syncthreads();
	}
	{
		const int cc = t & 63, nn = t >> 6; // 4 time rows per pass
#pragma unroll
		for (int q = 0; q < 16; q++) tile[nn + 4 * q][cc] = p.rotT[(size_t)(n0 + nn + 4 * q) * p.rotT_stride + c0 + cc];
	}
	__syncthreads();
	{
		const int nn = t & 63, cq = t >> 6; // 4 chain rows per pass
#pragma unroll
		for (int q = 0; q < 16; q++) {
			const int c = c0 + cq + 4 * q;
			if (c < p.n_chan) {
				const float2 xv = p.c48[(size_t)c * p.c48_stride + n0 + nn];
				p.cgf[(size_t)c * p.cgf_stride + CGF_HIST + n0 + nn] = cmul(xv, tile[nn][cq + 4 * q]);
			}
		}
	}
}

// ------------------------------------------------------------------------------------------
// K3: FilterComplex(Filters::Coherent) + ScatterPLL (DSP/DSP.cpp:215-246, DSP/DSP.h:95-117).
// One thread per complete 5-sample group: five 17-tap dot products accumulated left-to-right
// from 0 (DSP.h:224-230), the group level ((((0+n0)+n1)+n2)+n3)+n4)/5, and the de-interleave
// into five per-phase sample streams for the PhaseSearchEMA chains.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k3_fir_scatter(K3Params p) {
	const int chan = blockIdx.y;
	const int g = blockIdx.x * blockDim.x + threadIdx.x; // group index inside the block
	if (g >= p.n_groups) return;
	// stream index of the group's first sample relative to the block start (may be -4..0 for g = 0)
	const long long n_rel = (p.first_group + g) * 5 - p.first_sample48;
	const float2* x = p.cgf + (size_t)chan * p.cgf_stride + CGF_HIST + n_rel - 16;
	float2 win[21];
#pragma unroll
	for (int i = 0; i < 21; i++) win[i] = x[i];
	float level = 0.0f;
#pragma unroll
	for (int j = 0; j < 5; j++) {
		float2 acc = make_float2(0.0f, 0.0f);
#pragma unroll
		for (int i = 0; i < 17; i++) {
			float tp = p.taps[i];
			acc = make_float2(acc.x + tp * win[j + i].x, acc.y + tp * win[j + i].y);
		}
		level = level + (acc.x * acc.x + acc.y * acc.y); // std::norm
		// PhaseSearchEMA multiplies symbol n of a chain by (1j)^(n & 3) with swaps/negations (Demod.cpp:44-61);
		// every chain has consumed exactly first_group + g symbols, so that exact rotation is applied here
		const int rot = (int)((p.first_group + g) & 3);
		float2 sv = (rot & 1) ? make_float2(acc.y, acc.x) : acc; // rot 1: (-y, x)   rot 3: (y, -x)
		if (rot == 1 || rot == 2) sv.x = -sv.x;
		if (rot >= 2) sv.y = -sv.y;
		p.sym[sym_row_base(chan, j, p.sym_stride) + sym_offset(g)] = sv;
		if (p.fir_tap) p.fir_tap[(size_t)chan * p.fir_tap_stride + (n_rel + j + 4)] = acc;
	}
	p.lvl[(size_t)chan * p.sym_stride + g] = __fdiv_rn(level, 5.0f);
}

// ------------------------------------------------------------------------------------------
// K4: PhaseSearchEMA (DSP/Demod.cpp:39-101).  One 16-lane row per (receiver, channel, phase)
// chain, one lane per phase hypothesis: lane k keeps ma[k] and bits[k]; the +-1 neighbourhood
// argmax becomes two per-lane predicates gathered with wave ballots, and every lane of the row
// tracks max_idx redundantly from the ballot words (pure integer recurrence).
// ------------------------------------------------------------------------------------------
__constant__ float2 c_ps_phase[8] = { // DSP/Demod.h:29-31
	{ 9.9518472640441780e-01f, 9.8017143048367339e-02f }, { 9.5694033335306883e-01f, 2.9028468509743588e-01f },
	{ 8.8192125790916542e-01f, 4.7139674887287397e-01f }, { 7.7301044123076901e-01f, 6.3439329894649099e-01f },
	{ 6.3439326515712957e-01f, 7.7301046896098113e-01f }, { 4.7139671032286945e-01f, 8.8192127851457169e-01f },
	{ 2.9028464326824349e-01f, 9.5694034604181499e-01f }, { 9.8017099547459546e-02f, 9.9518473068888236e-01f } };

// neighbour exchange inside a 16-lane row: DPP row rotate (a VALU modifier, no LDS round trip).
// The rotate direction is probed once per wave (mode 0: row_ror:1 delivers lane k-1; mode 1: lane k+1;
// mode 2: unexpected -> fall back to ds_bpermute), so correctness never rests on the ISA manual's wording.
// (mov_dpp: every lane is written, so there is no `old` value to initialise -- one instruction instead of two)
__device__ __forceinline__ float dpp_ror1(float v) {
	return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x121, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_ror15(float v) {
	return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x12F, 0xF, 0xF, true));
}

// Per-lane state is just ma[k]; the 8-bit decision shift registers bits[k] of the reference live in
// wave ballots: D[j] = ballot(decision of j+1 symbols ago), so "bit(nDelay) XOR bit(nDelay+1)" of any
// hypothesis is a scalar XOR of two ballot words and costs no vector instruction.
struct PsWave {
	unsigned long long h1, h2, h3, h4; // decisions of 1..4 symbols ago, one bit per lane
};

// M.y: the hypothesis' EMA (M.x is unused).  The step is issue-bound next to the front end, and a packed f32
// instruction takes two issue slots on gfx950: the EMA update as one packed multiply and one packed cross add (plus the v_and for
// |t|, which packed operands cannot take as a modifier) was five slots, the three scalar instructions below are three.
__device__ __forceinline__ float ps_ema(float ma, float tt) { // w * ma + (1 - w) * |t|, each product and the sum rounded (Demod.cpp:71)
	const float w = 0.85f;
	const float w1 = 1 - w; // (1 - weight) evaluated in float
	float p_t, p_m, r;
	asm("v_mul_f32 %0, |%1|, %2" : "=v"(p_t) : "v"(tt), "s"(w1));
	asm("v_mul_f32 %0, %1, %2" : "=v"(p_m) : "v"(ma), "s"(w));
	asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(p_m), "v"(p_t));
	return r;
}

// One symbol for all 16 hypotheses of a row (DSP/Demod.cpp:39-101); v is already multiplied by (1j)^n (K3).
// Returns the emitted bit (0/1).
template <int MODE>
__device__ __forceinline__ unsigned ps_step(float2 v, float pc, float psn, c2& M, PsWave& hs, int& idx, int k, int rowbase) {
	const float a = v.x * pc, b = v.y * psn;
	const float tt = a + b;
	const unsigned long long dn = __ballot(tt > 0); // bits[k] = (bits[k] << 1) | (t > 0)
	M.y = ps_ema(M.y, tt);
	const float ma = M.y;
	float left, right;
	if (MODE == 2) {
		left = __shfl(ma, (k + 15) & 15, 16);
		right = __shfl(ma, (k + 1) & 15, 16);
	} else {
		const float r1 = dpp_ror1(ma), r15 = dpp_ror15(ma);
		left = MODE == 0 ? r1 : r15;
		right = MODE == 0 ? r15 : r1;
	}
#ifdef ABL_NO_WALK // (ablation builds only: wrong results)
	{ const unsigned long long X0 = hs.h3 ^ hs.h4; hs.h4 = hs.h3; hs.h3 = hs.h2; hs.h2 = hs.h1; hs.h1 = dn; (void)left; (void)right; return (unsigned)X0 & 1u; }
#endif
	const bool p0 = ma > left;         // centre beats idx-1
	const float bestc = p0 ? ma : left;
	const bool p1 = right > bestc;     // idx+1 beats the better of the two
	const unsigned long long B0 = __ballot(p0), B1 = __ballot(p1);
	const int sh = rowbase | (idx & 15); // (rowbase is a multiple of 16)
	// prev-1, prev, prev+1 with first-maximum preference: +1 where the third candidate wins (p1), -1 where neither the second
	// nor the third does; the two ballot words are combined on the scalar unit, a lane only extracts its bit of each
	const unsigned long long UP = B1, DN = ~(B0 | B1);
	const int up = (int)((unsigned)(UP >> sh) & 1u);
	int down, moved; // (pinned: the compiler turns the sign-extending extract into and + sub and the three-operand add into two)
	asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(down) : "v"((unsigned)(DN >> sh))); // -1 or 0
	asm("v_add3_u32 %0, %1, %2, %3" : "=v"(moved) : "v"(idx), "v"(up), "v"(down));
	idx = moved; // (only its low four bits count: masked where it is used -- one v_and_or with the row base per step -- and where it is stored)
	// nDelay = 3 (Model.cpp:560-561): after the shift-in, bit 3 = decision of 3 symbols ago, bit 4 = 4 symbols ago
	const unsigned long long X = hs.h3 ^ hs.h4;
	hs.h4 = hs.h3; hs.h3 = hs.h2; hs.h2 = hs.h1; hs.h1 = dn;
	return (unsigned)(X >> (rowbase | (idx & 15))) & 1u;
}

#ifndef PS_BATCH_
#define PS_BATCH_ 8
#endif
constexpr int PS_BATCH = PS_BATCH_;  // symbols whose samples are fetched together (8; 4 with a tighter register budget)

template <int MODE>
__device__ __forceinline__ void ps_chain(const SymRow x, uint32_t* __restrict__ out, int n, bool writer, float pc,
                                         float psn, c2& ma, PsWave& hs, int& idx, int k, int rowbase) {
	const int nb = n - (n % PS_BATCH);
	uint32_t word = 0;
	float2 cur[PS_BATCH];
	if (nb > 0) {
#pragma unroll
		for (int e = 0; e < PS_BATCH; e++) cur[e] = x[e];
	}
#pragma unroll 1
	for (int g0 = 0; g0 < nb; g0 += PS_BATCH) {
		float2 nxt[PS_BATCH];
		const int gn = g0 + PS_BATCH < nb ? g0 + PS_BATCH : g0; // next batch in flight while this one is processed
#pragma unroll
		for (int e = 0; e < PS_BATCH; e++) nxt[e] = x[gn + e];
		uint32_t part = 0;
#pragma unroll
		for (int e = 0; e < PS_BATCH; e++) part |= ps_step<MODE>(cur[e], pc, psn, ma, hs, idx, k, rowbase) << e;
		word |= part << (g0 & 31);
		if (((g0 + PS_BATCH) & 31) == 0) {
			if (writer) out[g0 >> 5] = word;
			word = 0;
		}
#pragma unroll
		for (int e = 0; e < PS_BATCH; e++) cur[e] = nxt[e];
	}
	for (int g = nb; g < n; g++) {
		word |= ps_step<MODE>(x[g], pc, psn, ma, hs, idx, k, rowbase) << (g & 31);
		if ((g & 31) == 31) {
			if (writer) out[g >> 5] = word;
			word = 0;
		}
	}
	if ((n & 31) != 0 && writer) out[n >> 5] = word;
}

// Sequential reference search: one pass over the whole block for the four chains of workgroup bx.  On its own (k4_phase_search)
// for small batches / one-chunk blocks, and as the exact fallback of the chunk-parallel path inside k4_assemble.
__device__ __forceinline__ void ps_search_quad(const K4Params& p, int bx) {
	const int lane = threadIdx.x;
	const int k = lane & 15, row = lane >> 4;
	const int chain = bx * 4 + row; // (rx*2 + ch) * 5 + j
	const bool live = chain < p.n_chains;
	const int cidx = live ? chain : p.n_chains - 1;
	const int rowbase = row * 16;

	const int src = __builtin_amdgcn_update_dpp(0, k, 0x121, 0xF, 0xF, false);
	const bool all_left = __all(src == ((k + 15) & 15)), all_right = __all(src == ((k + 1) & 15));

	const int jj = k < 8 ? k : 15 - k;
	const float pc = c_ps_phase[jj].x;
	const float psn = k < 8 ? c_ps_phase[jj].y : -c_ps_phase[jj].y; // a - b == a + (im * -s) exactly

	const EmaState* st = p.state_in + cidx;
	EmaState* sto = p.state_out + cidx;
	c2 ma = c2{ st->ma[k], st->ma[k] };
	const unsigned bits = st->bits[k]; // bit j = decision of j+1 symbols ago
	PsWave hs;
	hs.h1 = __ballot((bits & 1u) != 0);
	hs.h2 = __ballot((bits & 2u) != 0);
	hs.h3 = __ballot((bits & 4u) != 0);
	hs.h4 = __ballot((bits & 8u) != 0);
	int idx = st->max_idx;

	const SymRow x(p.sym, cidx, p.sym_stride);
	uint32_t* out = p.bits + (size_t)cidx * p.bits_stride;
	const bool writer = live && k == 0;
	if (all_left) ps_chain<0>(x, out, p.n_groups, writer, pc, psn, ma, hs, idx, k, rowbase);
	else if (all_right) ps_chain<1>(x, out, p.n_groups, writer, pc, psn, ma, hs, idx, k, rowbase);
	else ps_chain<2>(x, out, p.n_groups, writer, pc, psn, ma, hs, idx, k, rowbase);
	if (live) {
		sto->ma[k] = ma.y;
		// only the last four decisions can ever be read again (bits 3 and 4 after the next shift-in)
		sto->bits[k] = (unsigned)((hs.h1 >> lane) & 1ull) | ((unsigned)((hs.h2 >> lane) & 1ull) << 1) |
		               ((unsigned)((hs.h3 >> lane) & 1ull) << 2) | ((unsigned)((hs.h4 >> lane) & 1ull) << 3);
		if (k == 0) { sto->max_idx = idx & 15; sto->rot = (st->rot + p.n_groups) & 3; }
	}
}
__global__ __launch_bounds__(64) void k4_phase_search(K4Params p) { ps_search_quad(p, blockIdx.x); }

// ------------------------------------------------------------------------------------------
// K4': Demod::PhaseSearch (Demod.cpp:103-170), the boxcar variant behind `-go PS_EMA off` (nHistory = 12, nDelay = 3,
// nSearch = 2; Model.h:218-219).  Same row layout (lane k = hypothesis k, 4 chains per wave), sequential over the
// block.  memory[k][slot] lives in LDS; the average is the sum of the 12 slots in SLOT order (not time order),
// exactly like the reference's inner loop; lane k evaluates the 5-candidate first-maximum search for "previous
// maximum = k", and the row then takes the answer of the lane its previous maximum points at.
// An optional, unoptimised mode: one dependent LDS/bpermute round trip per symbol.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k4_phase_search_box(K4Params p) {
	__shared__ float mem[64][13]; // [lane][slot], padded
	const int lane = threadIdx.x;
	const int k = lane & 15, row = lane >> 4;
	const int chain = blockIdx.x * 4 + row;
	const bool live = chain < p.n_chains;
	const int cidx = live ? chain : p.n_chains - 1;
	const int rowbase = row * 16;
	const int jj = k < 8 ? k : 15 - k;
	const float pc = c_ps_phase[jj].x;
	const float psn = k < 8 ? c_ps_phase[jj].y : -c_ps_phase[jj].y; // a - b == a + (im * -s) exactly
	const PsBoxState* st = p.box_in + cidx;
	PsBoxState* sto = p.box_out + cidx;
#pragma unroll
	for (int l = 0; l < 12; l++) mem[lane][l] = st->mem[l][k];
	unsigned bits = st->bits[k];
	int idx = st->max_idx;
	int last = (int)(p.first_group % 12); // every chain has consumed first_group symbols
	const SymRow x(p.sym, cidx, p.sym_stride);
	uint32_t* out = p.bits + (size_t)cidx * p.bits_stride;
	uint32_t word = 0;
	for (int g = 0; g < p.n_groups; g++) {
		const float2 v = x[g]; // already multiplied by (1j)^n (K3)
		const float tt = v.x * pc + v.y * psn;
		bits = (bits << 1) | (tt > 0 ? 1u : 0u);
		mem[lane][last] = fabsf(tt);
		last = last == 11 ? 0 : last + 1;
		float avg = mem[lane][0];
#pragma unroll
		for (int l = 1; l < 12; l++) avg += mem[lane][l];
		// candidates prev-2 .. prev+2 in that order, strict '>' against a running maximum that starts at 0
		float max_val = 0.0f;
		int res = k;
#pragma unroll
		for (int d = -2; d <= 2; d++) {
			const float a = __shfl(avg, (k + d + 16) & 15, 16);
			if (a > max_val) { max_val = a; res = (k + d + 16) & 15; }
		}
		idx = __shfl(res, idx, 16);
		const unsigned b = (unsigned)__shfl((int)bits, idx, 16);
		word |= (((b >> 4) ^ (b >> 3)) & 1u) << (g & 31); // bit(nDelay + 1) XOR bit(nDelay)
		if ((g & 31) == 31) {
			if (live && k == 0) out[g >> 5] = word;
			word = 0;
		}
	}
	if ((p.n_groups & 31) != 0 && live && k == 0) out[p.n_groups >> 5] = word;
	if (live) {
#pragma unroll
		for (int l = 0; l < 12; l++) sto->mem[l][k] = mem[lane][l];
		sto->bits[k] = bits & 0xffu;
		if (k == 0) sto->max_idx = idx & 15;
	}
	(void)rowbase;
}

// ------------------------------------------------------------------------------------------
// K4' chunk-parallel (round 2): Demod::PhaseSearch's float state is the |t| of the last 12 symbols (slot = symbol count % 12) and
// the 8-bit decision registers -- a chunk that replays the 16 symbols in front of it has EXACTLY the sequential state, no
// speculation and nothing to verify.  Only max_idx is carried through time, and it is tracked for all 16 possible starts like
// in k4_phase_chunks (lane k: the trajectory that starts at k); k4_assemble selects.  Scratch layout as k4_phase_chunks
// (words, fin; ma_start / ma_fin are written as zeros so that the verification in k4_assemble is vacuous).
// ------------------------------------------------------------------------------------------
// Round 5 (late): the ring in REGISTERS.  The first form kept memory[k][slot] in LDS behind a running slot index -- 13 LDS operations
// and five ds_bpermute per symbol, 0.6 ms per step beside the front end.  Now a chunk starts where slot 0 is written (its warm-up is
// 16 .. 27 symbols instead of 16: any look-back of at least 12 gives the exact state), the loop walks twelve symbols per turn and
// every one of them writes ITS register of r[12]; the sum is the same twelve additions in slot order.  The four neighbours of the
// candidate search are DPP row rotations (direction probed per wave, ds_bpermute as the fallback, like the EMA search), the decisions
// live in ballots (PsWave), the samples come through the EMA kernel's LDS staging (one load = 16 symbols of a chain).  Only the
// block's first chunk, whose ring comes from the previous block with the slot counter wherever it stands, first walks up to eleven
// symbols with a run-time slot.
constexpr int BOX_WARM = 16;
constexpr int BOX_SB = 48;             // symbols per chain per super-batch: four turns of twelve
constexpr int BOX_SB_PAD = BOX_SB + 4; // row pitch (as PS_SB_PAD)
#ifndef BOX_NUM_VGPR
#define BOX_NUM_VGPR 96
#endif
template <int MODE, int D> // the value of lane k + D of the row (D = -2 .. 2)
__device__ __forceinline__ float box_neighbour(float v, int k) {
	if constexpr (D == 0) return v;
	else if constexpr (MODE == 2) return __shfl(v, (k + D + 16) & 15, 16);
	else { // MODE 0: row_ror:n delivers lane k - n, MODE 1: lane k + n
		constexpr int n = MODE == 0 ? (16 - D) & 15 : (16 + D) & 15;
		return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x120 + n, 0xF, 0xF, true));
	}
}
template <int MODE>
__device__ __forceinline__ void box_chunk_body(const K4Params& p, int chain, int chunk, bool live, int k, int rowbase, int lane,
                                               float2 (*stage)[4][BOX_SB_PAD]) {
	const int jj = k < 8 ? k : 15 - k;
	const float pc = c_ps_phase[jj].x;
	const float psn = k < 8 ? c_ps_phase[jj].y : -c_ps_phase[jj].y;
	const int g0 = chunk * PS_CHUNK;
	const int g1 = g0 + PS_CHUNK < p.n_groups ? g0 + PS_CHUNK : p.n_groups;
	const size_t slot = (size_t)chain * p.n_chunks + chunk;
	const int row = rowbase >> 4;
	const SymRow x(p.sym, chain, p.sym_stride);
	uint32_t* wout = p.words + slot * (PS_CHUNK / 32) * 16 + k;
	uint32_t word = 0;
	float r[12]; // memory[k][slot]: t itself, |t| where it is summed (a source modifier)
	PsWave hs;
	int idx = k; // the trajectory that starts at max_idx == k
	// what follows the ring write of a symbol: average in slot order, the five-candidate first-maximum search for "previous max_idx
	// == this lane", the hop of the trajectory, the delayed decision of the hypothesis it lands on
	const auto search = [&]() -> unsigned { // (the ring holds this symbol's t, the ballots the decisions in front of it)
		float avg = fabsf(r[0]);
#pragma unroll
		for (int l = 1; l < 12; l++) avg += fabsf(r[l]);
		float max_val = 0.0f;
		int res = k;
		k1_static_for<0, 5>([&](auto e) {
			constexpr int d = decltype(e)::value - 2;
			const float a = box_neighbour<MODE, d>(avg, k);
			if (a > max_val) { max_val = a; res = (k + d + 16) & 15; }
		});
		idx = __shfl(res, idx, 16);
		const unsigned long long X = hs.h3 ^ hs.h4; // bit(nDelay) XOR bit(nDelay + 1) after this symbol's shift-in
		return (unsigned)(X >> (rowbase | idx)) & 1u;
	};
	const auto finish = [&](float tt, int g) { // one symbol, wherever it stands (warm-up, the turns at a chunk's ends, the first chunk's peel)
		const unsigned long long dn = __ballot(tt > 0);
		if (g >= g0) { // wave-uniform
			const int q = g - g0;
			word |= search() << (q & 31);
			if ((q & 31) == 31) {
				if (live) wout[(q >> 5) * 16] = word;
				word = 0;
			}
		}
		hs.h4 = hs.h3; hs.h3 = hs.h2; hs.h2 = hs.h1; hs.h1 = dn;
	};
	int start;
	if (chunk == 0) { // the true state; the slot counter stands at first_group % 12
		const PsBoxState* st = p.box_in + chain;
#pragma unroll
		for (int l = 0; l < 12; l++) r[l] = st->mem[l][k];
		const unsigned bits = st->bits[k];
		hs.h1 = __ballot((bits & 1u) != 0); hs.h2 = __ballot((bits & 2u) != 0);
		hs.h3 = __ballot((bits & 4u) != 0); hs.h4 = __ballot((bits & 8u) != 0);
		const int s0 = (int)(p.first_group % 12);
		const int peel = (12 - s0) % 12 < g1 ? (12 - s0) % 12 : g1;
		float2 v[11];
#pragma unroll
		for (int e = 0; e < 11; e++) v[e] = x[e < peel ? e : 0];
#pragma unroll 1
		for (int e = 0; e < peel; e++) {
			float2 ve = v[0];
#pragma unroll
			for (int i = 1; i < 11; i++) ve = e == i ? v[i] : ve;
			const float tt = ve.x * pc + ve.y * psn;
			const int sl = s0 + e;
#pragma unroll
			for (int l = 0; l < 12; l++) r[l] = l == sl ? tt : r[l];
			finish(tt, e);
		}
		start = peel;
	} else { // every slot is rewritten during the warm-up
#pragma unroll
		for (int l = 0; l < 12; l++) r[l] = 0.0f;
		hs.h1 = hs.h2 = hs.h3 = hs.h4 = 0;
		start = g0 - BOX_WARM;
		start -= (int)((p.first_group + start) % 12);
	}
	const int last_i = (int)p.sym_stride - 1;
	float2 pre[BOX_SB / 16];
	const auto fetch = [&](int sb) {
#pragma unroll
		for (int q = 0; q < BOX_SB / 16; q++) {
			int i = start + sb * BOX_SB + q * 16 + k;
			i = i < last_i ? i : last_i;
			pre[q] = x[i];
		}
	};
	const auto stash = [&](int buf) {
#pragma unroll
		for (int q = 0; q < BOX_SB / 16; q++) stage[buf][row][q * 16 + k] = pre[q];
	};
	const int nsb = (g1 - start + BOX_SB - 1) / BOX_SB;
	if (nsb > 0) { fetch(0); stash(0); }
	if (nsb > 1) fetch(1);
#pragma unroll 1
	for (int sb = 0; sb < nsb; sb++) {
		const int buf = sb & 1;
		wave_sync();
#pragma unroll 1
		for (int s12 = 0; s12 < BOX_SB; s12 += 12) {
			const int g = start + sb * BOX_SB + s12;
			if (g >= g1) break;
			float2 v[12];
			{
				const float4* src = reinterpret_cast<const float4*>(&stage[buf][row][s12]);
#pragma unroll
				for (int e = 0; e < 12; e += 2) { const float4 t = src[e >> 1]; v[e] = make_float2(t.x, t.y); v[e + 1] = make_float2(t.z, t.w); }
			}
			if (g + 12 <= g1) {
#pragma unroll
				for (int e = 0; e < 12; e++) { const float tt = v[e].x * pc + v[e].y * psn; r[e] = tt; finish(tt, g + e); }
			} else {
#pragma unroll
				for (int e = 0; e < 12; e++)
					if (g + e < g1) { const float tt = v[e].x * pc + v[e].y * psn; r[e] = tt; finish(tt, g + e); } // wave-uniform
			}
		}
		if (sb + 1 < nsb) {
			stash(buf ^ 1);
			if (sb + 2 < nsb) fetch(sb + 2);
		}
	}
	const int n = g1 - g0;
	if ((n & 31) != 0 && live) wout[(n >> 5) * 16] = word;
	if (live) {
		const unsigned dec = (unsigned)((hs.h1 >> lane) & 1ull) | ((unsigned)((hs.h2 >> lane) & 1ull) << 1) |
		                     ((unsigned)((hs.h3 >> lane) & 1ull) << 2) | ((unsigned)((hs.h4 >> lane) & 1ull) << 3);
		p.ma_fin[slot * 16 + k] = 0.0f;
		if (chunk > 0) p.ma_start[slot * 16 + k] = 0.0f;
		p.fin[slot * 16 + k] = (unsigned)(idx & 15) | (dec << 4); // (only the last four decisions can ever be read again)
		if (chunk == p.n_chunks - 1) { // the block's final float state (max_idx: k4_assemble)
			PsBoxState* sto = p.box_out + chain;
#pragma unroll
			for (int l = 0; l < 12; l++) sto->mem[l][k] = fabsf(r[l]);
			sto->bits[k] = dec;
		}
	}
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(BOX_NUM_VGPR / 2))) void k4_box_chunks(K4Params p) { // (beside three front-end waves a SIMD has 104 registers left)
	__shared__ __attribute__((aligned(16))) float2 stage[2][4][BOX_SB_PAD];
	const int lane = threadIdx.x;
	const int k = lane & 15, row = lane >> 4;
	const int chunk = blockIdx.y;
	const int j = blockIdx.x % 5, chan = (blockIdx.x / 5) * 4 + row;
	const int chain_raw = chan * 5 + j;
	const bool live = chain_raw < p.n_chains;
	const int chain = live ? chain_raw : p.n_chains - 1;
	const int rowbase = row * 16;
	const int s1 = __builtin_amdgcn_update_dpp(0, k, 0x121, 0xF, 0xF, false), s2 = __builtin_amdgcn_update_dpp(0, k, 0x122, 0xF, 0xF, false);
	const int s14 = __builtin_amdgcn_update_dpp(0, k, 0x12E, 0xF, 0xF, false), s15 = __builtin_amdgcn_update_dpp(0, k, 0x12F, 0xF, 0xF, false);
	const bool all_left = __all(s1 == ((k + 15) & 15) && s2 == ((k + 14) & 15) && s14 == ((k + 2) & 15) && s15 == ((k + 1) & 15));
	const bool all_right = __all(s1 == ((k + 1) & 15) && s2 == ((k + 2) & 15) && s14 == ((k + 14) & 15) && s15 == ((k + 15) & 15));
	if (all_left) box_chunk_body<0>(p, chain, chunk, live, k, rowbase, lane, stage);
	else if (all_right) box_chunk_body<1>(p, chain, chunk, live, k, rowbase, lane, stage);
	else box_chunk_body<2>(p, chain, chunk, live, k, rowbase, lane, stage);
}

// ------------------------------------------------------------------------------------------
// K4 chunk-parallel: the block's symbols are cut into chunks of PS_CHUNK; one 16-lane row per (chain, chunk).
//  * ma[k] is a contraction (x0.85 per symbol), so a chunk starts from ma = 0 and first replays the `warm`
//    symbols in front of it; after that the float state is (with overwhelming probability) bit-identical to
//    the sequential one.  That is VERIFIED: k4_assemble compares the post-warm-up values with the previous
//    chunk's final values bit for bit; on any difference in its four chains the assembling wave runs the sequential
//    search above over the block, from the true state.  Results are therefore always bit-exact.
//  * the decisions (t > 0) do not depend on state at all.
//  * max_idx is not contractive, so it is not guessed: lane k of the row tracks the trajectory that STARTS at
//    max_idx = k (same instructions as one trajectory, the row's 16 lanes just stop being redundant);
//    k4_assemble then walks the chunks sequentially, picking for each the trajectory of the true start.
// ------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void ps_warm_step(float2 v, float pc, float psn, c2& M, PsWave& hs) {
	const float tt = v.x * pc + v.y * psn;
	const unsigned long long dn = __ballot(tt > 0);
	M.y = ps_ema(M.y, tt);
	hs.h4 = hs.h3; hs.h3 = hs.h2; hs.h2 = hs.h1; hs.h1 = dn;
}

// The 16 lanes of a row consume the same sample, and a wave that waits for its own 8-symbol prefetch is latency-bound as soon
// as the front end loads the memory system (0.17 ms alone, 0.55 ms next to it).  So the wave stages its input: every lane
// fetches a different symbol (one load instruction = 16 symbols per chain, whole 128-byte segments), a super-batch of PS_SB
// symbols per chain goes through a double-buffered LDS tile, and the loads of super-batch n+2 are in flight while n is
// processed (64 symbols x 20 instructions of cover).  The steps then read their sample as an LDS broadcast.
constexpr int PS_SB = 64;            // symbols per chain per super-batch
constexpr int PS_SB_PAD = PS_SB + 4; // row pitch: the four rows' broadcast reads fall into different banks

template <int MODE>
__device__ __forceinline__ void ps_chunk_body(const K4Params& p, int chain, int chunk, bool live, int k, int rowbase, int lane,
                                              float2 (*stage)[4][PS_SB_PAD]) {
	const int jj = k < 8 ? k : 15 - k;
	const float pc = c_ps_phase[jj].x;
	const float psn = k < 8 ? c_ps_phase[jj].y : -c_ps_phase[jj].y;
	const SymRow x(p.sym, chain, p.sym_stride);
	const int g0 = chunk * PS_CHUNK;
	const int g1 = g0 + PS_CHUNK < p.n_groups ? g0 + PS_CHUNK : p.n_groups;
	const size_t slot = (size_t)chain * p.n_chunks + chunk;
	const int row = rowbase >> 4;

	c2 ma;
	PsWave hs;
	int idx = k; // trajectory that starts at max_idx == k
	int start = g0;
	if (chunk == 0) { // the true state
		const EmaState* st = p.state_in + chain;
		ma = c2{ st->ma[k], st->ma[k] };
		const unsigned bits = st->bits[k];
		hs.h1 = __ballot((bits & 1u) != 0); hs.h2 = __ballot((bits & 2u) != 0);
		hs.h3 = __ballot((bits & 4u) != 0); hs.h4 = __ballot((bits & 8u) != 0);
	} else { // speculative: replay the `warm` symbols in front of the chunk from zero (warm: multiple of 8, <= PS_CHUNK)
		ma = c2{ 0.0f, 0.0f };
		hs.h1 = hs.h2 = hs.h3 = hs.h4 = 0;
		start = g0 - p.warm;
	}
	const int last_i = (int)p.sym_stride - 1;
	float2 r[PS_SB / 16];
	const auto fetch = [&](int sb) {
#pragma unroll
		for (int q = 0; q < PS_SB / 16; q++) {
			int i = start + sb * PS_SB + q * 16 + k;
			i = i < last_i ? i : last_i; // (past the chunk's end: any readable sample, it is never used)
			r[q] = x[i];
		}
	};
	const auto stash = [&](int buf) {
#pragma unroll
		for (int q = 0; q < PS_SB / 16; q++) stage[buf][row][q * 16 + k] = r[q];
	};
	const int nsb = (g1 - start + PS_SB - 1) / PS_SB;
	fetch(0);
	stash(0);
	if (nsb > 1) fetch(1);

	uint32_t* wout = p.words + slot * (PS_CHUNK / 32) * 16 + k;
	uint32_t word = 0;
#pragma unroll 1
	for (int sb = 0; sb < nsb; sb++) {
		const int buf = sb & 1;
		wave_sync(); // the tile written by this wave's own stash() is read below (one-wave workgroup: ordering only)
#pragma unroll 1
		for (int s8 = 0; s8 < PS_SB; s8 += PS_BATCH) {
			const int g = start + sb * PS_SB + s8;
			if (g >= g1) break;
			float2 v[PS_BATCH];
			{
				const float4* src = reinterpret_cast<const float4*>(&stage[buf][row][s8]);
#pragma unroll
				for (int e = 0; e < PS_BATCH; e += 2) { const float4 t = src[e >> 1]; v[e] = make_float2(t.x, t.y); v[e + 1] = make_float2(t.z, t.w); }
			}
			if (g < g0) { // warm-up: EMA and decision history only
#pragma unroll
				for (int e = 0; e < PS_BATCH; e++) ps_warm_step<MODE>(v[e], pc, psn, ma, hs);
				if (g + PS_BATCH == g0 && live) p.ma_start[slot * 16 + k] = ma.y;
			} else {
				const int q = g - g0;
				uint32_t part = 0;
				if (g + PS_BATCH <= g1) {
#pragma unroll
					for (int e = 0; e < PS_BATCH; e++) part |= ps_step<MODE>(v[e], pc, psn, ma, hs, idx, k, rowbase) << e;
				} else {
#pragma unroll
					for (int e = 0; e < PS_BATCH; e++)
						if (g + e < g1) part |= ps_step<MODE>(v[e], pc, psn, ma, hs, idx, k, rowbase) << e; // wave-uniform
				}
				word |= part << (q & 31);
				// (a last, partial batch never completes a word -- the write behind the loop is the one that stores it;
				// flushed here as well it was overwritten by an empty word whenever n % 32 was 25 .. 31)
				if (((q + PS_BATCH) & 31) == 0 && g + PS_BATCH <= g1) {
					if (live) wout[(q >> 5) * 16] = word;
					word = 0;
				}
			}
		}
		if (sb + 1 < nsb) {
			stash(buf ^ 1);                  // super-batch sb + 1 has been in flight for one super-batch of work
			if (sb + 2 < nsb) fetch(sb + 2);
		}
	}
	const int n = g1 - g0;
	if ((n & 31) != 0 && live) wout[(n >> 5) * 16] = word;
	if (live) {
		p.ma_fin[slot * 16 + k] = ma.y;
		const unsigned dec = (unsigned)((hs.h1 >> lane) & 1ull) | ((unsigned)((hs.h2 >> lane) & 1ull) << 1) |
		                     ((unsigned)((hs.h3 >> lane) & 1ull) << 2) | ((unsigned)((hs.h4 >> lane) & 1ull) << 3);
		p.fin[slot * 16 + k] = (unsigned)(idx & 15) | (dec << 4);
	}
}

#ifndef K4_WAVES
#define K4_WAVES 8
#endif
// Register budget: a SIMD that holds three front-end waves (3 x 136 VGPRs) has 104 registers left; what the PhaseSearch waves
// are allowed decides how many of them fit into that gap (64: one, 48: two, 32: three).  (The attribute counts architectural VGPRs and
// LLVM doubles it on targets with the unified register file -- a request above the waves-per-eu limit is silently dropped, which is
// why rounds 2-4 never saw 48 honoured: hence the / 2.)  Measured in round 5 (profiles/r05_expH_k4_two_waves.txt): with 48 registers and
// PS_BATCH_ = 4 the loop has no scratch traffic and two of these waves share a SIMD with the front end's three -- and the step does
// not move (+-0.3 % against the same batch size with 56 registers): it is not PhaseSearch's residency that the step waits for.
#ifndef K4_NUM_VGPR
#define K4_NUM_VGPR 64
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(K4_WAVES, K4_WAVES))) __attribute__((amdgpu_num_vgpr(K4_NUM_VGPR / 2))) void k4_phase_chunks(K4Params p) {
	__shared__ __attribute__((aligned(16))) float2 stage[2][4][PS_SB_PAD];
	const int lane = threadIdx.x;
	const int k = lane & 15, row = lane >> 4;
	const int chunk = blockIdx.y;
	// the four rows of a wave: the same sampling phase of four ADJACENT channels (their symbol pairs are 64 contiguous bytes in
	// the SymRow layout) and the same chunk (equal trip counts)
	const int j = blockIdx.x % 5, chan = (blockIdx.x / 5) * 4 + row;
	const int chain_raw = chan * 5 + j;
	const bool live = chain_raw < p.n_chains;
	const int chain = live ? chain_raw : p.n_chains - 1;
	const int rowbase = row * 16;
	const int src = __builtin_amdgcn_update_dpp(0, k, 0x121, 0xF, 0xF, false);
	const bool all_left = __all(src == ((k + 15) & 15)), all_right = __all(src == ((k + 1) & 15));
	if (all_left) ps_chunk_body<0>(p, chain, chunk, live, k, rowbase, lane, stage);
	else if (all_right) ps_chunk_body<1>(p, chain, chunk, live, k, rowbase, lane, stage);
	else ps_chunk_body<2>(p, chain, chunk, live, k, rowbase, lane, stage);
}

// sequential over the (few) chunks of a chain, 16 lanes per chain: verify the speculative warm-ups, select the
// trajectory of the true start index per chunk, emit the packed decisions and the new state
__device__ __forceinline__ bool ps_assemble_rows(const K4Params& p, int chain, int k) { // returns: a speculative warm-up of this row failed
	const EmaState* st = p.state_in + chain;
	EmaState* sto = p.state_out + chain;
	int start = p.box_out ? p.box_in[chain].max_idx : st->max_idx; // (boxcar variant: its own state block)
	uint32_t* out = p.bits + (size_t)chain * p.bits_stride;
	bool bad = false;
	// Next to the front end a dependent global load costs microseconds, so the walk over the chunks must not contain any:
	// lane k fetches fin[.][k] of AS chunks at once, the start index then hops from chunk to chunk through lane shuffles,
	// and only after that are the word gathers (whose addresses are all known by then) issued together.
	constexpr int AS = 8;
	static_assert(PS_CHUNK % 512 == 0, "k4_assemble: whole words per lane");
	constexpr int WPL = PS_CHUNK / 32 / 16; // words per lane and chunk
	const size_t base = (size_t)chain * p.n_chunks;
	unsigned fin_last = 0, ma_last = 0;
	for (int c0 = 0; c0 < p.n_chunks; c0 += AS) {
		unsigned f[AS], ms[AS], mf[AS];
#pragma unroll
		for (int e = 0; e < AS; e++) {
			const int c = c0 + e;
			const bool on = c < p.n_chunks;
			const size_t slot = base + (on ? c : p.n_chunks - 1);
			f[e] = p.fin[slot * 16 + k];
			mf[e] = __float_as_uint(p.ma_fin[slot * 16 + k]);
			ms[e] = (on && c > 0) ? __float_as_uint(p.ma_start[slot * 16 + k]) : 0u;
		}
		int st_e[AS];
#pragma unroll
		for (int e = 0; e < AS; e++) {
			const int c = c0 + e;
			st_e[e] = start;
			if (c < p.n_chunks) { // wave-uniform
				if (c > 0) bad = bad || (ms[e] != (e > 0 ? mf[e - 1] : ma_last));
				start = __shfl((int)f[e], start, 16) & 15;
				fin_last = f[e];
			}
		}
		uint32_t wv[AS][WPL];
#pragma unroll
		for (int e = 0; e < AS; e++) {
			const int c = c0 + e;
			if (c >= p.n_chunks) break;
			const int g0 = c * PS_CHUNK;
			const int n = (g0 + PS_CHUNK < p.n_groups ? PS_CHUNK : p.n_groups - g0);
			const int nw = (n + 31) >> 5;
			const uint32_t* w = p.words + (base + c) * (PS_CHUNK / 32) * 16 + st_e[e];
#pragma unroll
			for (int q = 0; q < WPL; q++) { const int i = k + 16 * q; wv[e][q] = i < nw ? w[i * 16] : 0u; }
		}
#pragma unroll
		for (int e = 0; e < AS; e++) {
			const int c = c0 + e;
			if (c >= p.n_chunks) break;
			const int g0 = c * PS_CHUNK;
			const int n = (g0 + PS_CHUNK < p.n_groups ? PS_CHUNK : p.n_groups - g0);
			const int nw = (n + 31) >> 5;
#pragma unroll
			for (int q = 0; q < WPL; q++) { const int i = k + 16 * q; if (i < nw) out[(g0 >> 5) + i] = wv[e][q]; }
		}
		ma_last = mf[AS - 1]; // only read when another batch follows, i.e. when all AS chunks of this one were live
	}
	const size_t last = base + (p.n_chunks - 1);
	sto->ma[k] = p.ma_fin[last * 16 + k];
	sto->bits[k] = fin_last >> 4;
	if (k == 0) { sto->max_idx = start; sto->rot = (st->rot + p.n_groups) & 3; }
	if (p.box_out && k == 0) p.box_out[chain].max_idx = start; // boxcar variant: its own state block
	return bad;
}

__global__ __launch_bounds__(64) void k4_assemble(K4Params p) {
	const int lane = threadIdx.x;
	const int k = lane & 15, row = lane >> 4;
	const int chain = blockIdx.x * 4 + row;
	if (chain >= p.n_chains) return;
	const bool bad = ps_assemble_rows(p, chain, k);
	// A speculative warm-up that did not reproduce the sequential EMA anywhere in the wave's four chains: the wave itself recomputes
	// them sequentially from the true state, here and now (the same wave would have been the one to do it in a kernel of its own --
	// which cost a launch boundary on the PhaseSearch stream, the pipeline's critical one, for every block).  (Rows of chains that
	// do not exist have left above: ballots and DPP row operations of the search only ever look inside a row.)
	if (__any(bad)) {
		if (__builtin_amdgcn_readfirstlane(lane) == lane && p.fb_count) atomicAdd(p.fb_count, 1);
		ps_search_quad(p, blockIdx.x);
	}
}

// ------------------------------------------------------------------------------------------
// K46 (round 5): derotation + FilterComplex(17) + ScatterPLL + PhaseSearchEMA in ONE workgroup -- the FIR outputs (`sym`, as large
// as the 48 kHz channels: 105 MB written and 123 MB read back per step of 256 receivers) never leave the chip, and k6_window_fir's
// launch is gone from the default path.  Reference: DSP.cpp:457-466, :215-246, DSP.h:95-117, Demod.cpp:39-101 -- every value is the
// one k6_window_fir / k4_phase_chunks compute, in the same order.
//  * workgroup = four waves = sixteen PhaseSearch rows: the five sampling phases of THREE adjacent channels (row R -> channel R / 5,
//    phase R % 5; the sixteenth row idles), one chunk of PS_CHUNK symbols with its speculative warm-up in front, like k4_phase_chunks
//    (round 2's k46 had five waves of 105 registers: they found no room beside three front-end waves per SIMD; four waves of <= 104
//    registers sit in the registers and the 40 KB of LDS the front end cannot use);
//  * the chunk is walked window by window (512 samples of each channel = 102 or 103 symbols of each phase).  Per window, wave c < 3
//    is k6_window_fir for channel c: lanes over time restart the phasor recurrence from the checkpoints (k2_cgf_refine), derotate
//    into the wave's own LDS window, then own one ScatterPLL group each; the five FIR outputs of a group go to an LDS row per
//    (channel, phase) instead of `sym`.  All four waves then run PhaseSearch over the window's symbols, samples as LDS broadcasts;
//  * software pipeline: the rows are double buffered, so ONE barrier per window; wave c computes window w + 1 in front of its
//    PhaseSearch pass over window w, and the global loads of window w + 2 (nine samples and one checkpoint per lane) are in flight
//    during that pass -- beside the front end a load takes 10-20 us, a pass about as long;
//  * a chunk's warm-up (256 symbols) recomputes 2.5 windows of FIR: +25 % of the FIR arithmetic (a tenth of PhaseSearch's).
// The groups of window w are those whose LAST sample lies in it (as in k6_window_fir); `lvl` of a group is written by the chunk that
// owns the group.  k46_assemble = k4_assemble; its exact fallback (a speculative warm-up that failed, never seen with 256 symbols
// of warm-up) first materialises `sym` for its channels with k6_window_fir's body, then runs the sequential search as before.
// ------------------------------------------------------------------------------------------
constexpr int K46_CH = 3;                               // channels per workgroup
constexpr int K46_HL = 3;                               // halo segments (26 samples: the FIR reaches 20 back)
constexpr int K46_HALO = K46_HL * CK_SEG - 1;
constexpr int K46_WIN = K46_HALO + 512 + 2;             // a wave's LDS window: halo, window, the unused ninth sample of the last segment
constexpr int K46_YBUF = K46_WIN + 64;                  // + the lanes' checkpoints (x parts, then y parts: two 256-byte runs)
constexpr int K46_ROW = 110;                            // LDS row of a (channel, phase): 103 symbols + 7 of alignment slack; 110 x 8 B puts the four rows of a wave into different banks
static_assert(K46_HALO >= 20 && K46_HALO <= DF_HIST, "k46: halo");
constexpr int K46_RAW4 = (K46_HALO + 512) / 2;          // 16-byte pieces of a window's raw samples (halo + window)
static_assert((K46_HALO & 1) == 0 && (K46_WIN & 1) == 0, "k46: the halo begins on a 16-byte boundary of the channel's row");
// LDS-DMA by hand: lane l's 16 (4) bytes at g land at LDS byte address lds_base + 16 l (4 l); lds_base wave-uniform.  Inline assembly
// on purpose: the compiler does not know these operations, so it puts no vmcnt wait in front of LDS reads that cannot alias them
// (with the builtin every ds_read of the PhaseSearch pass waited for vmcnt(0) -- i.e. also for the pass's own stores, which are
// acknowledged after 10-20 us beside the front end).  The reader waits itself: k46_wait_vm().  (m0 is not used by anything else here.)
__device__ __forceinline__ void k46_dma16(const void* g, unsigned lds_base) {
	asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void k46_dma4(const void* g, unsigned lds_base) {
	asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(g), "s"(lds_base) : "memory");
}
// wait until at most n (wave-uniform) of the wave's vector memory operations are outstanding: everything older than its n youngest
// -- the stores of the PhaseSearch pass in between -- has completed (loads and stores of a wave retire in order on gfx9)
__device__ __forceinline__ void k46_wait_vm(int n) {
	switch (n) {
	case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
	case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
	case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
	case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
	case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
	case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
	default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break; // (more stores in between: a stronger wait than needed)
	}
}
// workgroup barrier for LDS data only: __syncthreads() also waits for vmcnt(0) -- the prefetch in flight and the pass's stores
__device__ __forceinline__ void k46_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MODE>
__device__ __forceinline__ void k46_body(const K46Params& q, int trip, int chunk, float2 (*ybufs)[K46_YBUF], float2 (*symL)[K46_CH][5][K46_ROW]) {
	const K6Params& p = q.f;
	const K4Params& s = q.s;
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int W = p.n_windows;
	// ---- PhaseSearch rows
	const int k = lane & 15, row = lane >> 4, rowbase = row * 16;
	const int R = wv * 4 + row;
	const int c_ps = R < 15 ? R / 5 : K46_CH - 1, j_ps = R < 15 ? R % 5 : 4;
	const int chan_ps = trip * K46_CH + c_ps;
	const bool live = R < 15 && chan_ps < p.n_chan;
	const int chain = live ? chan_ps * 5 + j_ps : s.n_chains - 1;
	const int jj = k < 8 ? k : 15 - k;
	const float pc = c_ps_phase[jj].x;
	const float psn = k < 8 ? c_ps_phase[jj].y : -c_ps_phase[jj].y;
	float pc_l, psn_l; // (the same two behind the wait in front of the window loop)
	const int g0 = chunk * PS_CHUNK;
	const int g1 = g0 + PS_CHUNK < s.n_groups ? g0 + PS_CHUNK : s.n_groups;
	const size_t slot = (size_t)chain * s.n_chunks + chunk;
	c2 ma;
	PsWave hs;
	int idx = k;
	int start = g0;
	if (chunk == 0) {
		const EmaState* st = s.state_in + chain;
		ma = c2{ st->ma[k], st->ma[k] };
		const unsigned bits = st->bits[k];
		hs.h1 = __ballot((bits & 1u) != 0); hs.h2 = __ballot((bits & 2u) != 0);
		hs.h3 = __ballot((bits & 4u) != 0); hs.h4 = __ballot((bits & 8u) != 0);
	} else {
		ma = c2{ 0.0f, 0.0f };
		hs.h1 = hs.h2 = hs.h3 = hs.h4 = 0;
		start = g0 - s.warm;
	}
	uint32_t* wout = s.words + slot * (PS_CHUNK / 32) * 16 + k;
	uint32_t word = 0;
	const int wave_stores = __any(live) ? 1 : 0; // a store under `live` is an instruction of this wave iff one of its rows exists
	int n_st = 0; // vector memory operations (stores) this wave has issued since its last fir_load()
	// ---- windows of the chunk (group gl: samples n_rel0 + 5 gl .. + 4 of the block; it belongs to the window of its last sample)
	const int w_first = (p.n_rel0 + 5 * start + 4) >> 9, w_last = (p.n_rel0 + 5 * (g1 - 1) + 4) >> 9;
	const auto g_lo_of = [&](int w) { return (w * 512 - p.n_rel0) / 5; };
	const auto g_end_of = [&](int w) { return w >= W - 1 ? p.n_groups : min(p.n_groups, ((w + 1) * 512 - p.n_rel0) / 5); };
	// rows in LDS start at a group index that is a multiple of 8 away from `start`: PhaseSearch's batches of 8 then are aligned reads
	const auto base_of = [&](int w) { const int gl = g_lo_of(w); return gl - ((gl - start) & 7); };
	// ---- the derotation / FIR wave of channel wv (waves 0 .. 2)
	const int chan_f = trip * K46_CH + (wv < K46_CH ? wv : 0);
	const bool fir_wave = wv < K46_CH && chan_f < p.n_chan;
	const int chf = chan_f < p.n_chan ? chan_f : p.n_chan - 1;
	float2* ybuf = ybufs[wv < K46_CH ? wv : 0];
	const float2* xrow = p.c48 + (size_t)chf * p.c48_stride;
	// the windows' phasor steps: lane i keeps the step of window w_first - 1 + i (one dependent pair of loads per chunk, not per window)
	float2 stp_l;
	{
		int wi = w_first - 1 + lane;
		wi = wi < 0 ? 0 : (wi > W - 1 ? W - 1 : wi);
		stp_l = p.step_table[p.fz[(size_t)chf * W + wi] + 205];
	}
	const int seg = lane - K46_HL;
	const int si = seg >= 0 ? seg : CK_USED + seg;
	// the window's raw samples (with the halo in front) and the lanes' checkpoints travel from HBM straight into the wave's LDS window
	// (no staging registers: nine samples per lane held across a PhaseSearch pass were twenty registers the pass does not have).
	// Issued one PhaseSearch pass ahead of fir_compute(), which waits for them with k46_wait_vm().
	const unsigned ybuf_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) float2*)ybuf);
	const auto fir_load = [&](int w) {
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (the window's last reads have returned before anything lands in it)
		const int ws = seg >= 0 ? w : w - 1;
		if (lane < CK_USED + K46_HL && ws >= 0) {
			const float* ckp = reinterpret_cast<const float*>(p.ck + ((size_t)ws * CK_SLOTS + si) * p.ck_stride + chf);
			k46_dma4(ckp, ybuf_lds + K46_WIN * 8);
			k46_dma4(ckp + 1, ybuf_lds + K46_WIN * 8 + 256);
		}
		const float2* src = xrow + (w * 512 - K46_HALO) + 2 * lane;
#pragma unroll
		for (int e = 0; e < (K46_RAW4 + 63) / 64; e++) {
			const int piece = e * 64 + lane;
			if (piece < K46_RAW4 && (w > 0 || piece >= K46_HALO / 2)) // (the block's first window: its halo is the carried tail)
				k46_dma16(src + e * 128, ybuf_lds + e * 1024);
		}
	};
	const auto fir_compute = [&](int w, int younger, int buf) { // younger: vector memory operations of this wave since fir_load(w)
		k46_wait_vm(younger);
		const int ws = seg >= 0 ? w : w - 1;
		const int iw = w - (w_first - 1);
		const auto lane_value = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
		const float sx_c = lane_value(stp_l.x, iw), sy_c = lane_value(stp_l.y, iw), sx_p = lane_value(stp_l.x, iw - 1), sy_p = lane_value(stp_l.y, iw - 1);
		if (lane < CK_USED + K46_HL && ws >= 0) {
			const int n0 = ws * 512 + si * CK_SEG;
			const float stx = seg >= 0 ? sx_c : sx_p, sty = seg >= 0 ? sy_c : sy_p;
			const float* ckl = reinterpret_cast<const float*>(ybuf + K46_WIN);
			c2 rot = { ckl[lane], ckl[64 + lane] };
			const c2 st = { stx, sty }, st_sw = { -sty, stx };
			float2* y = ybuf + K46_HALO + (n0 - w * 512); // in place: every sample belongs to exactly one lane
			float2 d[CK_SEG];
#pragma unroll
			for (int m = 0; m < CK_SEG; m++) d[m] = y[m]; // (segment 56: the ninth is beyond the window, unused)
#pragma unroll
			for (int m = 0; m < CK_SEG; m++) {
				rot = rot.xx * st + rot.yy * st_sw;                       // rot *= rot_step
				const c2 v = pk_sub_add(rot * d[m].x, rot.yx * d[m].y);   // data * rot
				if (m < CK_SEG - 1 || si < CK_USED - 1) y[m] = make_float2(v.x, v.y);
			}
		}
		if (w == 0 && lane < K46_HALO) ybuf[lane] = p.hist_in[(size_t)chf * DF_HIST + (DF_HIST - K46_HALO) + lane]; // the previous block's tail
		wave_sync();
		if (w == W - 1 && lane < DF_HIST) p.hist_out[(size_t)chf * DF_HIST + lane] = ybuf[K46_HALO + 512 - DF_HIST + lane];
		const int g_lo = g_lo_of(w), g_end = g_end_of(w), base = base_of(w);
		const int r0sel = (int)(p.first_group & 3);
		for (int gl = g_lo + lane; gl < g_end; gl += 64) {
			const int a = p.n_rel0 + 5 * gl - w * 512; // the group's first sample relative to the window: -4 .. 507
			const float2* wsrc = ybuf + K46_HALO + a - 16;
			// the five 17-tap sums of the group advance together over its 21 samples -- each still x += taps[i] * d[i] from i = 0 upwards
			// (DSP.h:224-230) -- so a sample is dead once its five products are formed: ten accumulator registers instead of a 42-register window
			c2 acc[5] = { { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f } };
#pragma unroll
			for (int i = 0; i < 21; i++) {
				const float2 v = wsrc[i];
				const c2 x = c2{ v.x, v.y };
#pragma unroll
				for (int j = 0; j < 5; j++)
					if (i - j >= 0 && i - j < 17) acc[j] = acc[j] + x * p.taps[i - j];
			}
			const int rsel = (r0sel + gl) & 3; // the (1j)^n of PhaseSearchEMA (Demod.cpp:44-61), as in k6_window_fir
			const unsigned nx = (rsel == 1 || rsel == 2) ? 0x80000000u : 0u, ny = rsel >= 2 ? 0x80000000u : 0u;
			const bool swap = (rsel & 1) != 0;
			float level = 0.0f;
#pragma unroll
			for (int j = 0; j < 5; j++) {
				level = level + (acc[j].x * acc[j].x + acc[j].y * acc[j].y); // std::norm
				const float sx = swap ? acc[j].y : acc[j].x, sy = swap ? acc[j].x : acc[j].y;
				symL[buf][wv][j][gl - base] = make_float2(__uint_as_float(__float_as_uint(sx) ^ nx), __uint_as_float(__float_as_uint(sy) ^ ny));
			}
			if (gl >= g0 && gl < g1) p.lvl[(size_t)chf * p.sym_stride + gl] = __fdiv_rn(level, 5.0f); // (a warm-up group is the previous chunk's)
		}
		wave_sync(); // (the next window's derotation overwrites ybuf: LDS operations of a wave execute in order)
	};
	// ---- PhaseSearch over the symbols of window w that belong to this chunk (or its warm-up)
	const auto ps_window = [&](int w, int buf) {
		const int gl_lo = g_lo_of(w), gl_end = g_end_of(w), base = base_of(w);
		const int a = gl_lo > start ? gl_lo : start, b = gl_end < g1 ? gl_end : g1;
		const float2* srow = &symL[buf][c_ps][j_ps][0];
#pragma unroll 1
		for (int gb = a - ((a - start) & 7); gb < b; gb += 8) {
			float2 v[8];
			{
				const float4* src = reinterpret_cast<const float4*>(srow + (gb - base));
#pragma unroll
				for (int e = 0; e < 8; e += 2) { const float4 t = src[e >> 1]; v[e] = make_float2(t.x, t.y); v[e + 1] = make_float2(t.z, t.w); }
			}
			const bool full = gb >= a && gb + 8 <= b;
			const int gend = gb + 8 < b ? gb + 8 : b;
			if (gb < g0) { // warm-up: EMA and decision history only (a batch never straddles g0: warm and PS_CHUNK are multiples of 8)
				if (full) {
#pragma unroll
					for (int e = 0; e < 8; e++) ps_warm_step<MODE>(v[e], pc_l, psn_l, ma, hs);
				} else {
#pragma unroll
					for (int e = 0; e < 8; e++) if (gb + e >= a && gb + e < b) ps_warm_step<MODE>(v[e], pc_l, psn_l, ma, hs); // wave-uniform
				}
				if (gend == g0) { if (live) s.ma_start[slot * 16 + k] = ma.y; n_st += wave_stores; }
			} else {
				const int qb = gb - g0;
				if (full) {
					uint32_t part = 0;
#pragma unroll
					for (int e = 0; e < 8; e++) part |= ps_step<MODE>(v[e], pc_l, psn_l, ma, hs, idx, k, rowbase) << e;
					word |= part << (qb & 31);
					if (((qb + 8) & 31) == 0) { if (live) wout[(qb >> 5) * 16] = word; word = 0; n_st += wave_stores; }
				} else {
#pragma unroll
					for (int e = 0; e < 8; e++) {
						if (gb + e >= a && gb + e < b) { // wave-uniform
							word |= ps_step<MODE>(v[e], pc_l, psn_l, ma, hs, idx, k, rowbase) << ((qb + e) & 31);
							if (((qb + e) & 31) == 31) { if (live) wout[(qb >> 5) * 16] = word; word = 0; n_st += wave_stores; }
						}
					}
				}
			}
		}
	};
	// Everything fetched with ordinary loads so far is made final HERE: a value whose load the compiler still counts as pending when the
	// window loop begins would get its vmcnt(0) wait inside the loop -- where it waits for the PhaseSearch pass's stores.
	asm volatile("s_waitcnt vmcnt(0)" : "+v"(stp_l.x), "+v"(stp_l.y), "+v"(ma), "+v"(idx) :: "memory");
	{ float pc_ = pc, psn_ = psn; asm volatile("" : "+v"(pc_), "+v"(psn_)); pc_l = pc_; psn_l = psn_; }
	if (fir_wave) {
		fir_load(w_first);
		fir_compute(w_first, 0, w_first & 1);
		if (w_first + 1 <= w_last) fir_load(w_first + 1);
	}
	k46_barrier();
#pragma unroll 1
	for (int w = w_first; w <= w_last; w++) {
		if (fir_wave && w + 1 <= w_last) {
			fir_compute(w + 1, n_st, (w + 1) & 1);   // its loads were issued a whole PhaseSearch pass ago
			if (w + 2 <= w_last) fir_load(w + 2);    // in flight during the pass below
		}
		n_st = 0;
		ps_window(w, w & 1);
		k46_barrier(); // rows [w & 1] are free for window w + 2, rows [(w + 1) & 1] complete
	}
	const int n = g1 - g0;
	if ((n & 31) != 0 && live) wout[(n >> 5) * 16] = word;
	if (live) {
		s.ma_fin[slot * 16 + k] = ma.y;
		const unsigned dec = (unsigned)((hs.h1 >> lane) & 1ull) | ((unsigned)((hs.h2 >> lane) & 1ull) << 1) |
		                     ((unsigned)((hs.h3 >> lane) & 1ull) << 2) | ((unsigned)((hs.h4 >> lane) & 1ull) << 3);
		s.fin[slot * 16 + k] = (unsigned)(idx & 15) | (dec << 4);
	}
}

#ifndef K46_WAVES
#define K46_WAVES 5 // occupancy target of five waves per SIMD = at most 96 registers: three front-end waves of 136 leave 104
#endif
// (The LDS is dynamic: with a static 40 KB the compiler sees "four workgroups per CU" and budgets 128 registers whatever it is asked
// for; what the kernel has to fit into is the gap beside the front end, which the compiler cannot know.)
// (and the two sides of the LDS are different objects: the waits the compiler puts in front of reads of the DMA's target must not land in
// front of PhaseSearch's reads of the rows)
constexpr size_t K46_LDS = sizeof(float2) * ((size_t)2 * K46_CH * 5 * K46_ROW);
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(K46_WAVES, K46_WAVES))) void k46_window_search(K46Params q) {
	__shared__ __attribute__((aligned(16))) float2 ybufs[K46_CH][K46_YBUF];
	static_assert(sizeof(float2) * K46_CH * K46_YBUF + K46_LDS <= 40960, "k46: the LDS twelve front-end workgroups leave on a CU");
	extern __shared__ __attribute__((aligned(16))) float2 k46_rows[];
	float2 (*symL)[K46_CH][5][K46_ROW] = reinterpret_cast<float2 (*)[K46_CH][5][K46_ROW]>(k46_rows);
	// workgroup id -> (channel triple, chunk): id % 8 is the XCD; an XCD takes a contiguous eighth of the triples, neighbours in
	// time one behind the other -- the 128-byte lines of the time-major checkpoints (sixteen chains each) stay in ONE L2
	const int per = q.trips_pad >> 3;
	const int xcd = blockIdx.x & 7, sq = blockIdx.x >> 3;
	const int trip = xcd * per + sq % per, chunk = sq / per;
	if (trip * K46_CH >= q.f.n_chan) return; // (the whole workgroup: no barrier has been reached)
	const int lane = threadIdx.x & 63, k = lane & 15;
	const int src = __builtin_amdgcn_update_dpp(0, k, 0x121, 0xF, 0xF, false);
	const bool all_left = __all(src == ((k + 15) & 15)), all_right = __all(src == ((k + 1) & 15));
	if (all_left) k46_body<0>(q, trip, chunk, ybufs, symL);
	else if (all_right) k46_body<1>(q, trip, chunk, ybufs, symL);
	else k46_body<2>(q, trip, chunk, ybufs, symL);
}

// k4_assemble for the fused workgroups: there is no `sym` in HBM, so the exact fallback first produces the rows of its (at most two)
// channels with k6_window_fir's body, window by window -- into the global `sym` of this block parity, which nothing else touches in
// this mode (two waves that share a channel write the same values) -- and then runs the sequential search over them.
__global__ __launch_bounds__(64) void k46_assemble(K46Params q) {
	__shared__ __attribute__((aligned(16))) float2 ybuf[K6_YBUF];
	const K4Params& p = q.s;
	const int lane = threadIdx.x;
	const int k = lane & 15, row = lane >> 4;
	const int chain = blockIdx.x * 4 + row;
	// (rows of chains that do not exist stay: the fallback's derotation / FIR body needs all 64 lanes of the wave)
	const bool bad = chain < p.n_chains && ps_assemble_rows(p, chain, k);
	if (__any(bad)) {
		if (lane == 0 && p.fb_count) atomicAdd(p.fb_count, 1);
		const int c_lo = (blockIdx.x * 4) / 5, c_hi = min((int)blockIdx.x * 4 + 3, p.n_chains - 1) / 5;
		K6Params f = q.f;
		f.hist_out = nullptr; // (the fused workgroups have written the carried tail)
		for (int c = c_lo; c <= c_hi; c++)
			for (int w = 0; w < f.n_windows; w++) { k6_window_body<false>(f, c, w, ybuf, nullptr); __syncthreads(); } // (one wave: ordering)
		__threadfence();
		ps_search_quad(p, blockIdx.x);
	}
}

// ------------------------------------------------------------------------------------------
// K5: ModelChallenger's non-coherent branch (DSP/Model.cpp:638-639): Demod::FM (DSP/Demod.cpp:27-37) ->
// DSP::Filter with Filters::Receiver (37 taps, DSP/DSP.cpp:249-280) -> Deinterleave(5) -> AIS::Decoder.
// The decoders only look at the sign, so the device hands back one bit per 48 kHz sample.
// atan2f is glibc's (fdlibm e_atan2f.c / s_atanf.c) restated operation by operation; the CPU test
// tests/test_atan2f.py checks the same restatement against the host libm on 10^7 inputs, and the GPU parity
// tests compare every decision with the reference chain.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float atanf_ref(float x) {
	const float atanhi[4] = { 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f };
	const float atanlo[4] = { 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f };
	const float aT[11] = { 3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f,
	                       -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f,
	                       1.6285819933e-02f };
	const int hx = __float_as_int(x);
	const int ix = hx & 0x7fffffff;
	int id;
	if (ix >= 0x4c000000) { // |x| >= 2^25
		if (ix > 0x7f800000) return x + x;
		return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
	}
	if (ix < 0x3ee00000) { // |x| < 0.4375
		if (ix < 0x31000000) return x; // |x| < 2^-29 (the reference raises inexact here; the value is x)
		id = -1;
	} else {
		x = fabsf(x);
		if (ix < 0x3f980000) { // |x| < 1.1875
			if (ix < 0x3f300000) { id = 0; x = __fdiv_rn(2.0f * x - 1.0f, 2.0f + x); }
			else { id = 1; x = __fdiv_rn(x - 1.0f, x + 1.0f); }
		} else {
			if (ix < 0x401c0000) { id = 2; x = __fdiv_rn(x - 1.5f, 1.0f + 1.5f * x); }
			else { id = 3; x = __fdiv_rn(-1.0f, x); }
		}
	}
	const float z = x * x;
	const float w = z * z;
	const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
	const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
	if (id < 0) return x - x * (s1 + s2);
	const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
	return hx < 0 ? -r : r;
}

__device__ __forceinline__ float atan2f_ref(float y, float x) {
	const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
	const int hx = __float_as_int(x), hy = __float_as_int(y);
	const int ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
	if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
	if (hx == 0x3f800000) return atanf_ref(y);
	const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
	if (iy == 0) {
		if (m < 2) return y;
		return m == 2 ? pi + tiny : -pi - tiny;
	}
	if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
	if (ix == 0x7f800000) {
		if (iy == 0x7f800000) {
			switch (m) {
			case 0: return pi_o_4 + tiny;
			case 1: return -pi_o_4 - tiny;
			case 2: return 3.0f * pi_o_4 + tiny;
			default: return -3.0f * pi_o_4 - tiny;
			}
		}
		switch (m) {
		case 0: return 0.0f;
		case 1: return -0.0f;
		case 2: return pi + tiny;
		default: return -pi - tiny;
		}
	}
	if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
	const int k = (iy - ix) >> 23;
	float z;
	if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
	else if (hx < 0 && k < -60) z = 0.0f;
	else z = atanf_ref(fabsf(__fdiv_rn(y, x)));
	switch (m) {
	case 0: return z;
	case 1: return __int_as_float(__float_as_int(z) ^ 0x80000000);
	case 2: return pi - (z - pi_lo);
	default: return (z - pi_lo) - pi;
	}
}

// The FM receiver as ONE kernel: Demod::FM (Demod.cpp:27-37) into LDS, DSP::Filter with the 37 Receiver taps (DSP.cpp:249-280) out
// of LDS, the sign into the bit row.  A workgroup makes 256 outputs and needs the discriminator at 36 samples in front of them: the
// first tile of a block takes those from the previous block's last tile (hist_in / hist_out, two buffers by block parity -- first
// and last tile of one launch run side by side), every other tile computes them again (14 % more atan2, and the discriminator
// output -- 50 MB per block written and read 37 times through the caches -- never leaves the chip; it is stored only as a tap).
__global__ __launch_bounds__(256) void k5_fm_filter(K5Params p) {
	__shared__ float s_fm[256 + FM_HIST];
	const int chan = blockIdx.y, t = threadIdx.x;
	const int t0 = blockIdx.x * 256; // L is a multiple of 256
	const float2* y = p.x + (size_t)chan * p.x_stride + p.x_off;
	const auto disc = [&](int n) { // sample n >= 0 of the block
		const float2 d = y[n];
		const float2 pv = (n == 0 && p.prev_in) ? p.prev_in[chan] : y[n - 1];
		if (n == p.L - 1 && p.prev_out) p.prev_out[chan] = d;
		// data[i] * std::conj(prev): (xr*pr - xi*(-pi), xr*(-pi) + xi*pr)
		const float npi = -pv.y;
		const float re = d.x * pv.x - d.y * npi;
		const float im = d.x * npi + d.y * pv.x;
		return __fdiv_rn(atan2f_ref(im, re), 3.14159265358979323846f);
	};
	const float f = disc(t0 + t);
	s_fm[FM_HIST + t] = f;
	if (p.fm) p.fm[(size_t)chan * p.fm_stride + FM_HIST + t0 + t] = f;
	if (t < FM_HIST) s_fm[t] = t0 == 0 ? p.hist_in[(size_t)chan * FM_HIST + t] : disc(t0 - FM_HIST + t);
	if (t0 + 256 == p.L && t >= 256 - FM_HIST) p.hist_out[(size_t)chan * FM_HIST + (t - (256 - FM_HIST))] = f;
	__syncthreads();
	float acc = 0.0f;
#pragma unroll
	for (int i = 0; i < 37; i++) acc += p.taps[i] * s_fm[t + i]; // x += taps[i] * *data++ (DSP.h:257-263)
	const int n = t0 + t;
	if (p.fir_out) p.fir_out[(size_t)chan * p.fir_stride + n] = acc;
	const unsigned long long b = __ballot(acc > 0);
	if ((t & 63) == 0) {
		uint32_t* o = p.fmbits + (size_t)chan * p.fmbits_stride + (n >> 5);
		o[0] = (uint32_t)b;
		o[1] = (uint32_t)(b >> 32);
	}
}

// ------------------------------------------------------------------------------------------
// KV2: the decoder-independent part of V2::Engine (ModelEngineV2, DSP/Decoder/V2/V2Engine.cpp), see kernels.h.
// ------------------------------------------------------------------------------------------
// sample n of the stream relative to this block's first sample; n in [-V2_HIST, L)
__device__ __forceinline__ float2 v2_sample(const KV2Params& p, int chan, int n) {
	return n < 0 ? p.hist[(size_t)chan * V2_HIST + (V2_HIST + n)] : p.c48[(size_t)chan * p.c48_stride + n];
}

// FreqOffset::Estimate (:56-131) for V2E_NW windows per wave: the FFT of the squared window with the reference's butterflies
// (fft512 passes as in k2_fft_mag), magnitudes sqrtf(re^2 + im^2) in float (norm2, :33-36) in fftshift order into LDS, then the
// reference's loops (:74-131) in their order -- what is a dependent chain by lanes that do nothing else, what is not by the whole wave
// (round 6, late; one lane per window ran all of it before: 166 us per 256 receivers, most of it that lane's ~15 instructions and LDS
// round trips per candidate):
//  * the rolling 133-bin sum (rolling - m[i-1] + m[i+132]: two dependent operations per step) by lanes 0-7, one per window, which
//    leave every step's value in a small ring; `total` (the sum of all 512 bins from the left) starts with the same 133 additions
//    and goes on with exactly the bins the rolling sum takes in (m[133] ... m[511]), so lanes 8-15 run it in the same instructions
//    (they subtract 0.0f, which changes nothing: the sums are >= +0 or NaN);
//  * the 380 candidates rolling + 0.6 (m[i+15] + m[i+117]) and "first maximum under a strict >" by all lanes in chunks of 64 steps:
//    eight lanes per window, each over its candidates in rising order, then the larger value wins among them and on equal values
//    the lower index; a NaN candidate never wins, and a NaN at candidate 0 keeps index 0, as `v > best` has it;
//  * the peak pair 102 bins apart, the prominence and the parabola through the three pair sums by one lane per window.
constexpr int V2E_NW = 8;                    // windows per wave (eight lanes per window in the candidate search)
constexpr int V2E_MS = 512 + 64 / V2E_NW;    // floats per window of magnitudes: 8 banks between the windows of a wave
constexpr int V2E_CH = 64, V2E_RP = V2E_CH + 8; // steps per chunk of the rolling sum, row pitch of its ring
// the wave-wide part: magnitudes of the window whose squared samples sit in v[] (bit-reversed positions, as fft_square leaves them),
// fftshift order, into mg[512]
__device__ __forceinline__ void v2_fft_mag_sq(c2 (&v)[8], float2* X, float* mg, FftTwiddles& t, int lane) {
	const int l7 = lane & 7, l8 = lane >> 3;
	fft_pass(v, t.a0, t.a1, t.a2);
#pragma unroll
	for (int r = 0; r < 8; r++) X[9 * lane + r] = make_float2(v[r].x, v[r].y);
	wave_sync();
#pragma unroll
	for (int r = 0; r < 8; r++) { const float2 d = X[l7 + 9 * r + 72 * l8]; v[r] = c2{ d.x, d.y }; }
	wave_sync();
	fft_pass(v, t.b0, t.b1, t.b2);
#pragma unroll
	for (int r = 0; r < 8; r++) X[l7 + 8 * r + 72 * l8] = make_float2(v[r].x, v[r].y);
	wave_sync();
#pragma unroll
	for (int r = 0; r < 8; r++) { const float2 d = X[lane + 72 * r]; v[r] = c2{ d.x, d.y }; }
	wave_sync();
	fft_pass(v, t.c0, t.c1, t.c2_); // bin lane + 64 r
#pragma unroll
	// sqrtf(norm2(x)) (:69-72): the float square root through the correctly rounded double one (53 >= 2 * 24 + 2 bits: exact)
	for (int r = 0; r < 8; r++) mg[(lane + 64 * r + 256) & 511] = (float)__dsqrt_rn((double)(v[r].x * v[r].x + v[r].y * v[r].y));
}
// behind the candidate search (:89-131), one lane per window: wi_ = the winning candidate, total = the sum of the 512 bins
__device__ __forceinline__ void v2_search_tail(const float* m, int wi_, float total, float& f_out, float& prom_out) {
	constexpr int N = 512, delta = 102, M = 133;
	int fz = -1;
	float peak = 0.0f;
	for (int i = wi_; i < wi_ + (M - delta); i++) {
		const float h = m[i] + m[i + delta];
		if (h > peak) { peak = h; fz = i; }
	}
	prom_out = total > 0.0f ? __fdiv_rn(peak * (float)(N / 2), total) : 0.0f;
	float f = 0.0f;
	if (fz >= 0) {
		float frac = 0.0f;
		if (fz > 0 && fz + delta + 1 < N) {
			const float a = m[fz - 1] + m[fz - 1 + delta];
			const float c = m[fz + 1] + m[fz + 1 + delta];
			const float den = a - 2.0f * peak + c;
			if (den < 0.0f) {
				frac = __fdiv_rn(0.5f * (a - c), den);
				frac = frac > 0.5f ? 0.5f : (frac < -0.5f ? -0.5f : frac);
			}
		}
		f = __fdiv_rn(__fdiv_rn((float)(N / 2) - ((float)fz + frac + 51.0f), 2.0f), (float)N);
	}
	f_out = f;
}

// The same estimate by one lane for ONE window (the engine's slot-locked estimates inside kv2_engine / kv2_engine_roles: rare):
// the wave-wide part: magnitudes of the window that starts at sample s0 of the channel, fftshift order, into mg[512]
__device__ __forceinline__ void v2_fft_mag_window(const KV2Params& p, int chan, int s0, float2* X, float* mg, FftTwiddles& t, int lane) {
	const int src = fft_src_lane(lane);
	float2 dn[8];
#pragma unroll
	for (int r = 0; r < 8; r++) dn[r] = v2_sample(p, chan, s0 + src + fft_src_step(r));
	c2 v[8];
	fft_square(dn, v); // window[n] * window[n] into the bit-reversed position (:63-64)
	const int l7 = lane & 7, l8 = lane >> 3;
	fft_pass(v, t.a0, t.a1, t.a2);
#pragma unroll
	for (int r = 0; r < 8; r++) X[9 * lane + r] = make_float2(v[r].x, v[r].y);
	wave_sync();
#pragma unroll
	for (int r = 0; r < 8; r++) { const float2 d = X[l7 + 9 * r + 72 * l8]; v[r] = c2{ d.x, d.y }; }
	wave_sync();
	fft_pass(v, t.b0, t.b1, t.b2);
#pragma unroll
	for (int r = 0; r < 8; r++) X[l7 + 8 * r + 72 * l8] = make_float2(v[r].x, v[r].y);
	wave_sync();
#pragma unroll
	for (int r = 0; r < 8; r++) { const float2 d = X[lane + 72 * r]; v[r] = c2{ d.x, d.y }; }
	wave_sync();
	fft_pass(v, t.c0, t.c1, t.c2_); // bin lane + 64 r
#pragma unroll
	// sqrtf(norm2(x)) (:69-72): the float square root through the correctly rounded double one (53 >= 2 * 24 + 2 bits: exact)
	for (int r = 0; r < 8; r++) mg[(lane + 64 * r + 256) & 511] = (float)__dsqrt_rn((double)(v[r].x * v[r].x + v[r].y * v[r].y));
}
// the sequential part (:74-131), one lane: the reference's loops in their order
__device__ __forceinline__ void v2_search(const float* m, float& f_out, float& prom_out) {
	constexpr int N = 512, delta = 102, M = 133, ofs = 15;
	float rolling = 0.0f;
	for (int jx = 0; jx < M; jx++) rolling += m[jx];
	float best = rolling + 0.6f * (m[ofs] + m[ofs + delta]);
	int wi_ = 0;
	for (int i = 1; i <= N - M; i++) {
		rolling = rolling - m[i - 1] + m[i + M - 1];
		const float v = rolling + 0.6f * (m[i + ofs] + m[i + ofs + delta]);
		if (v > best) { best = v; wi_ = i; }
	}
	int fz = -1;
	float peak = 0.0f;
	for (int i = wi_; i < wi_ + (M - delta); i++) {
		const float h = m[i] + m[i + delta];
		if (h > peak) { peak = h; fz = i; }
	}
	float total = 0.0f;
	for (int i = 0; i < N; i++) total += m[i];
	prom_out = total > 0.0f ? __fdiv_rn(peak * (float)(N / 2), total) : 0.0f;
	float f = 0.0f;
	if (fz >= 0) {
		float frac = 0.0f;
		if (fz > 0 && fz + delta + 1 < N) {
			const float a = m[fz - 1] + m[fz - 1 + delta];
			const float c = m[fz + 1] + m[fz + 1 + delta];
			const float den = a - 2.0f * peak + c;
			if (den < 0.0f) {
				frac = __fdiv_rn(0.5f * (a - c), den);
				frac = frac > 0.5f ? 0.5f : (frac < -0.5f ? -0.5f : frac);
			}
		}
		f = __fdiv_rn(__fdiv_rn((float)(N / 2) - ((float)fz + frac + 51.0f), 2.0f), (float)N);
	}
	f_out = f;
}

__global__ __launch_bounds__(64) void kv2_estimate(KV2Params p) {
	__shared__ __attribute__((aligned(16))) float2 X[584];
	__shared__ __attribute__((aligned(16))) float mag[V2E_NW * V2E_MS];
	static_assert(2 * V2E_NW * V2E_RP <= 2 * 584, "kv2_estimate: the ring of rolling sums (and as many rows of scratch) lives in the FFT's exchange buffer");
	const int lane = threadIdx.x;
	const int nw = 2 * p.n_windows;
	const int W0 = blockIdx.x * V2E_NW, n_win_total = p.n_chan * nw;
	FftTwiddles t = fft_twiddles(p.omega, lane);
	const int src = fft_src_lane(lane);
	// the 8 samples of a lane for window W0 + wi; the next window's are requested before this one's butterflies start (as k2_fft_mag)
	float2 dn[8];
	const auto fetch = [&](int wi) {
		int W = W0 + wi;
		W = W < n_win_total ? W : n_win_total - 1;
		const int chan = W / nw, w = W - chan * nw;
		const int s0 = -V2_HIST + 256 * w + src;
#pragma unroll
		for (int r = 0; r < 8; r++) dn[r] = v2_sample(p, chan, s0 + fft_src_step(r));
	};
	fetch(0);
	for (int wi = 0; wi < V2E_NW; wi++) {
		if (W0 + wi >= n_win_total) break;
		c2 v[8];
		fft_square(dn, v); // window[n] * window[n] into the bit-reversed position (:63-64)
		__builtin_amdgcn_sched_barrier(0);
		if (wi + 1 < V2E_NW) fetch(wi + 1);
		__builtin_amdgcn_sched_barrier(0);
		v2_fft_mag_sq(v, X, mag + wi * V2E_MS, t, lane);
	}
	wave_sync();

	constexpr int N = 512, delta = 102, M = 133, ofs = 15, NC = N - M + 1; // 380 candidates
	float* rl = reinterpret_cast<float*>(X); // [2 * V2E_NW][V2E_RP]: the chunk's rolling sums in rows 0-7 (the FFTs are through with X)
	const int cw = lane & 7;
	const bool chain = lane < 16, is_total = lane >= 8; // lanes 0-7: rolling sum of window cw, lanes 8-15: its `total`
	const float* mc = mag + cw * V2E_MS;
	const int ew = lane >> 3, es = lane & 7;            // candidate search: window ew, candidates es + 8 k of a chunk
	const float* me = mag + ew * V2E_MS;
	float r = 0.0f;
	if (chain) {
#pragma unroll 19
		for (int jx = 0; jx < M; jx++) r += mc[jx];
	}
	float best = -1.0f; // (every candidate is >= +0 or NaN)
	int bi = 0;
	bool nan0 = false;
#pragma unroll 1
	for (int c0 = 0; c0 < NC; c0 += V2E_CH) {
		if (chain) {
#pragma unroll 1
			for (int s0 = 0; s0 < V2E_CH; s0 += 16) {
				float a[16], b[16];
#pragma unroll
				for (int e = 0; e < 16; e++) {
					const int i = c0 + s0 + e;
					const bool on = i >= 1 && i < NC; // (step 0 is the initial sum itself; zeros leave r as it is)
					const float av = mc[i >= 1 ? i - 1 : 0], bv = mc[i + M - 1]; // (i + 132 <= 515 < V2E_MS)
					a[e] = (on && !is_total) ? av : 0.0f;
					b[e] = on ? bv : 0.0f;
				}
#pragma unroll
				for (int e = 0; e < 16; e++) {
					r = (r - a[e]) + b[e];
					rl[lane * V2E_RP + s0 + e] = r; // (rows 8-15: the `total` lanes' running values, which nobody reads -- no branch in the chain)
				}
			}
		}
		wave_sync();
#pragma unroll
		for (int k = 0; k < V2E_CH / 8; k++) {
			const int s = es + 8 * k, i = c0 + s;
			const float v = rl[ew * V2E_RP + s] + 0.6f * (me[i + ofs] + me[i + ofs + delta]); // (i + 117 <= 500)
			if (i == 0) nan0 = v != v;
			if (i < NC && v > best) { best = v; bi = i; }
		}
		wave_sync();
	}
	// among the eight lanes of a window: the larger value, on equal values the lower index
#pragma unroll
	for (int d = 1; d < 8; d <<= 1) {
		const float ob = __shfl_xor(best, d);
		const int oi = __shfl_xor(bi, d);
		if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
	}
	if (__shfl((int)nan0, lane & ~7)) bi = 0;
	const float total = __shfl(r, 8 + ew);
	const int W = W0 + ew;
	if (es == 0 && W < n_win_total) {
		float f, prom;
		v2_search_tail(me, bi, total, f, prom);
		p.est_f[W] = f;
		p.est_prom[W] = prom;
	}
}

// half-block energies (midWins, :281-291): one lane per (channel, block start -512 + 512 i), 256 terms in order
__global__ __launch_bounds__(64) void kv2_energy(KV2Params p) {
	const int id = blockIdx.x * 64 + threadIdx.x;
	const int per = p.n_windows + 1;
	if (id >= p.n_chan * per) return;
	const int chan = id / per, i = id - chan * per;
	const int s0 = -V2_HIST + 512 * i;
	float e = 0.0f;
	for (int n = 0; n < 256; n++) {
		const float2 z = v2_sample(p, chan, s0 + n);
		e += z.x * z.x + z.y * z.y;
	}
	p.energy[id] = e;
}

// octant-reduced polynomial arctangent (:244-263), operation by operation
__device__ __forceinline__ float atan2_fast_ref(float y, float x) {
	const float ax = fabsf(x), ay = fabsf(y);
	const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
	if (mx == 0.0f) return 0.0f;
	const float a = __fdiv_rn(mn, mx), s = a * a;
	float r = ((-0.0464964749f * s + 0.15931422f) * s - 0.327622764f) * s * a + a;
	if (ay > ax) r = 1.57079637f - r;
	if (x < 0.0f) r = 3.14159274f - r;
	return y < 0.0f ? -r : r;
}

// FMDemod::Run (:265-273) + FilterFL37 (:48-54, :175-188) as ONE kernel (round 6, the form of k5_fm_filter; kv2_fm + kv2_filter before:
// 67 + 95 us per 256 receivers, the discriminator -- 50 MB per block -- written and read 19 times per output through the caches).  A workgroup
// makes 256 outputs: out[n] = sum_{i<18} (a[i] + a[36 - i]) * taps[i] + a[18] * taps[18] over a = disc[n-36 .. n], the discriminator in LDS.  The 36
// values in front of a workgroup's outputs: the first workgroup of a block takes them from the previous block's tail (disc[0 .. 36), put there by
// kv2_carry), every other one computes them again (14 % more arctangents).  (Four outputs per lane and tiles of 1,024 -- 3.5 % more arctangents, a
// lane's 40 values in ten 128-bit LDS reads -- take 48 instead of 56 us alone and 139 instead of 130 us beside the estimates, which is where
// the kernel runs: measured, not kept.)  Only the block's last 36 discriminator values are stored (for
// kv2_carry), all of them where the taps are asked for (disc_full).
__global__ __launch_bounds__(256) void kv2_fm_filter(KV2Params p) {
	__shared__ float s_fm[256 + FM_HIST];
	const int t = threadIdx.x;
	const int t0 = blockIdx.x * 256; // L is a multiple of 512
	if ((int)blockIdx.y < p.energy_rows) { // the launch's FIRST rows of workgroups (so that these long lanes start first): the half-block energies, kv2_energy's lanes
		const int id = ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * 256 + t;
		const int per = p.n_windows + 1;
		if (id < p.n_chan * per) {
			const int ch = id / per, i = id - ch * per;
			const int s0 = -V2_HIST + 512 * i;
			float e = 0.0f;
			for (int n = 0; n < 256; n++) {
				const float2 z = v2_sample(p, ch, s0 + n);
				e += z.x * z.x + z.y * z.y;
			}
			p.energy[id] = e;
		}
		return;
	}
	const int chan = (int)blockIdx.y - p.energy_rows;
	const float2* y = p.c48 + (size_t)chan * p.c48_stride;
	float* dsc = p.disc + (size_t)chan * (FM_HIST + p.L);
	const auto disc = [&](int n) { // sample n >= 0 of the block
		const float2 d = y[n];
		const float2 pv = n == 0 ? p.fmprev[chan] : y[n - 1];
		const float npi = -pv.y; // input[i] * std::conj(prev)
		const float re = d.x * pv.x - d.y * npi;
		const float im = d.x * npi + d.y * pv.x;
		return __fdiv_rn(atan2_fast_ref(im, re), 3.14159265358979323846f);
	};
	const float f = disc(t0 + t);
	s_fm[FM_HIST + t] = f;
	if (p.disc_full || (t0 + 256 == p.L && t >= 256 - FM_HIST)) dsc[FM_HIST + t0 + t] = f;
	if (t < FM_HIST) s_fm[t] = t0 == 0 ? dsc[t] : disc(t0 - FM_HIST + t);
	__syncthreads();
	const float* a = s_fm + t; // a[i] = disc[n - 36 + i]
	float sum = 0.0f;
#pragma unroll
	for (int i = 0; i < 18; i++) sum += (a[i] + a[36 - i]) * p.taps[i];
	sum = sum + a[18] * p.taps[18];
	const int n = t0 + t;
	if (p.fir_out) p.fir_out[(size_t)chan * p.fir_stride + n] = sum;
	const unsigned long long b = __ballot(sum > 0.0f);
	if ((t & 63) == 0) {
		uint32_t* o = p.fmbits + (size_t)chan * p.fmbits_stride + (n >> 5);
		o[0] = (uint32_t)b;
		o[1] = (uint32_t)(b >> 32);
	}
}

// after everything above has read them: the block's tail becomes the next block's look-back -- one channel by the 64 lanes of a wave.
// (Round 6, late: done by the engine's own workgroups instead, at their start or at the end of wave 2's loop, it cost the engine kernel
// 110 us of its 2,066 -- measured on one box, A/B -- so it stays a kernel, and leaves the step's chain by its stream: aisgpu.cpp.)
__device__ __forceinline__ void v2_carry_chan(const KV2Params& p, int chan, int lane) {
	float* dsc = p.disc + (size_t)chan * (FM_HIST + p.L);
	if (lane < FM_HIST) dsc[lane] = dsc[p.L + lane];
	const float2* x = p.c48 + (size_t)chan * p.c48_stride + (p.L - V2_HIST);
	for (int i = lane; i < V2_HIST; i += 64) p.hist_out[(size_t)chan * V2_HIST + i] = x[i];
	if (lane == 0) p.fmprev[chan] = p.c48[(size_t)chan * p.c48_stride + p.L - 1];
	if (p.fmtail_out && lane < 16) p.fmtail_out[(size_t)chan * 16 + lane] = p.fmbits[(size_t)chan * p.fmbits_stride + (p.L - 512) / 32 + lane];
}
__global__ __launch_bounds__(64) void kv2_carry(KV2Params p) { v2_carry_chan(p, blockIdx.x, threadIdx.x); }

// ------------------------------------------------------------------------------------------
// K7: AIS::Decoder (Marine/AIS.h:82-181, Marine/AIS.cpp:33-142) on the device -- NRZI, training / start-flag state machine,
// bit de-stuffing, CRC-16 residue check, the early-abort heuristics and the Reset mesh between the five decoders of a
// channel (DSP/Model.cpp:566-573).  One lane per decoder, the five decoders of a channel in adjacent lanes (12 channels per
// wave), one pass over the block's symbols.  The reference feeds the five decoders of a group one after the other; here
// they step together, and only when one of them completes a frame with a good CRC is the order restored: the decoders
// after it are rolled back, reset and stepped again, the ones before it are reset afterwards (a decoder that has just been
// reset cannot complete a frame, so one repair pass is enough).  Frames leave as records (bits as received, level sum,
// indices); length validation, the level in dB and the NMEA text stay on the host (Marine/Message.cpp), they are string work.
// ------------------------------------------------------------------------------------------
#include "dec_core.h"

__global__ __launch_bounds__(64) void k7_decode(K7Params p) {
	__shared__ uint32_t fdata[DEC_DATA_WORDS * 64]; // [word][lane]
	__shared__ float lv[64 * 33];                   // [lane][32 groups], padded
	const int lane = threadIdx.x;
	if (p.cond) { // exact fallback of the event-driven kernels: only for a block whose candidate lists overflowed
		if (*p.cond == 0) return;
		if (blockIdx.x == 0 && lane == 0 && p.cond_count) atomicAdd(p.cond_count, 1);
	}
	const int mesh = lane / 5, j = lane - 5 * mesh;  // lanes 60..63 idle
	const int chan_raw = blockIdx.x * 12 + mesh;
	const bool live = lane < 60 && chan_raw < p.n_chan;
	const int chan = live ? chan_raw : 0;
	const int dec = chan * 5 + (live ? j : 0);
	uint32_t* data = fdata + lane;
	DecState* st = p.state + dec;
	DecReg r;
	r.state = st->state; r.lastBit = st->lastBit; r.prev = st->prev; r.position = st->position; r.osc = st->osc;
	r.level = st->level; r.start_idx = st->start_idx;
	for (int w = 0; w < DEC_DATA_WORDS; w++) data[64 * w] = st->data[w];
	r.crc = st->crc[0]; r.cw = st->crc[1]; r.cwi = (int)st->crc[2]; r.tail = st->crc[3]; r.abort_pos = (int)st->crc[4];
	const uint32_t* brow = p.bits + (size_t)dec * p.bits_stride;
	const float* lrow = p.lvl + (size_t)chan * p.lvl_stride;
	for (int g0 = 0; g0 < p.n_groups; g0 += 32) {
		const uint32_t word = brow[g0 >> 5];
		const int n = p.n_groups - g0 < 32 ? p.n_groups - g0 : 32;
		{ // the row is padded to a multiple of 32 groups: eight unconditional 16-byte loads, issued together
			const float4* src = reinterpret_cast<const float4*>(lrow + g0);
			float4 t[8];
#pragma unroll
			for (int q = 0; q < 8; q++) t[q] = src[q];
#pragma unroll
			for (int q = 0; q < 8; q++) { float* d = &lv[lane * 33 + 4 * q]; d[0] = t[q].x; d[1] = t[q].y; d[2] = t[q].z; d[3] = t[q].w; }
		}
		for (int e = 0; e < n; e++) {
			const int g = g0 + e;
			const int dd = (int)((word >> e) & 1u);
			const float slvl = lv[lane * 33 + e];
			const long long sidx = 5 * (p.first_group + g) + j; // tag.sample_idx of this symbol
			const DecReg before = r;
			bool found = live && dec_step(r, dd, slvl, sidx, data);
			const unsigned long long F = __ballot(found);
			if (F != 0) { // rare: restore the order in which the reference runs the five decoders of a group
				const unsigned m5 = (unsigned)(F >> (5 * mesh)) & 31u;
				if (live && m5 != 0) {
					const int jmin = __builtin_ctz(m5);
					if (j > jmin) { // would have been reset before its step
						r = before;
						r.state = DST_TRAINING; r.position = 0; r.osc = 0;
						dec_step(r, dd, slvl, sidx, data);
					} else if (j < jmin) { // reset after its step
						r.state = DST_TRAINING; r.position = 0; r.osc = 0;
					} else { // the decoder that found the message
						const unsigned slot = atomicAdd(p.frame_count, 1u) % (unsigned)p.max_frames;
						{
							uint32_t* f = p.frames + (size_t)slot * DEC_FRAME_WORDS;
							f[0] = (uint32_t)dec; f[1] = (uint32_t)g; f[2] = (uint32_t)r.position; f[3] = __float_as_uint(r.level);
							f[4] = (uint32_t)(unsigned long long)r.start_idx; f[5] = (uint32_t)((unsigned long long)r.start_idx >> 32);
							f[6] = (uint32_t)(unsigned long long)sidx; f[7] = (uint32_t)((unsigned long long)sidx >> 32);
							f[8] = p.block; f[9] = p.sub;
							for (int w = 0; w < DEC_DATA_WORDS; w++) f[10 + w] = data[64 * w];
						}
						r.state = DST_TRAINING; r.position = 0; r.osc = 0; // FOUNDMESSAGE (resets the siblings), then TRAINING
					}
				}
			}
		}
	}
	if (live) {
		st->state = r.state; st->lastBit = r.lastBit; st->prev = r.prev; st->position = r.position; st->osc = r.osc;
		st->level = r.level; st->start_idx = r.start_idx;
		data[64 * r.cwi] = r.cw;
		for (int w = 0; w < DEC_DATA_WORDS; w++) st->data[w] = data[64 * w];
		st->crc[0] = r.crc; st->crc[1] = r.cw; st->crc[2] = (uint32_t)r.cwi; st->crc[3] = r.tail; st->crc[4] = (uint32_t)r.abort_pos;
	}
}

// ------------------------------------------------------------------------------------------
// KV2E: V2::Engine's coherent branch with its decoders on the device (kernels.h: KV2EParams).  Per channel everything below
// happens in the reference's order; what is parallel is the channels (one per wave) and, inside a channel, the block-wise stages and
// the six decoders of a group of five samples -- until one of them completes a message: then the group is redone sample by sample.
// ------------------------------------------------------------------------------------------
// sinf / cosf of glibc 2.35 (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h) for |y| < 120, operation by operation, in the
// variant glibc's ifunc selects on CPUs with FMA (every a * b + c fused): double-precision polynomials on the argument reduced by
// multiples of pi / 2, the result rounded to float once.  Coefficients: __sincosf_table of the image's libm.so.6.  The CPU test
// tests/test_sincosf.py checks the same restatement against the host libm on 2 x 10^7 inputs.
// (host + device: aisgpu_create() runs the same code on the host against the host's own sinf / cosf before it accepts the engine
// on the device -- a host whose libm is not this one would otherwise diverge from the reference silently)
#if defined(__HIP_DEVICE_COMPILE__)
#define V2_FMA(a, b, c) __fma_rn((a), (b), (c))
#else
#define V2_FMA(a, b, c) fma((a), (b), (c))
#endif
__host__ __device__ __forceinline__ unsigned v2_float_bits(float v) { unsigned u; memcpy(&u, &v, 4); return u; }
__host__ __device__ __forceinline__ float sincosf_poly_ref(double x, double x2, bool neg, int n) {
	const double sg = neg ? -1.0 : 1.0; // the second table entry: the cosine coefficients negated
	if ((n & 1) == 0) {
		const double x3 = x * x2;
		const double s1 = V2_FMA(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
		const double x7 = x3 * x2;
		const double s = V2_FMA(x3, -0x1.555545995a603p-3, x);
		return (float)V2_FMA(x7, s1, s);
	}
	const double x4 = x2 * x2;
	const double c2 = V2_FMA(x2, sg * 0x1.99343027bf8c3p-16, sg * -0x1.6c087e89a359dp-10);
	const double c1 = V2_FMA(x2, sg * -0x1.ffffffd0c621cp-2, sg * 0x1p0);
	const double x6 = x4 * x2;
	const double c = V2_FMA(x4, sg * 0x1.55553e1068f19p-5, c1);
	return (float)V2_FMA(x6, c2, c);
}
__host__ __device__ __forceinline__ float sin_or_cos_ref(float y, int iscos) {
	const auto abstop12 = [](float v) { return (v2_float_bits(v) >> 20) & 0x7ffu; };
	double x = (double)y;
	if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
		if (abstop12(y) < abstop12(0x1p-12f)) return iscos ? 1.0f : y;
		return sincosf_poly_ref(x, x * x, false, iscos);
	}
	const double r = x * 0x1.45F306DC9C883p+23;
	const int n = ((int)r + 0x800000) >> 24;
	x = V2_FMA(-(double)n, 0x1.921FB54442D18p0, x);
	const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
	return sincosf_poly_ref(x * sgn, x * x, (n & 2) != 0, n ^ iscos);
}
// The restatement above against the host's libm on the arguments FreqOffset::Derotate can form (|f| <= 0.25 cycles per sample:
// |theta| <= pi / 2) and beyond: every float of a 2^-20 grid over [-1.7, 1.7] and a sweep of small magnitudes; true = identical bits.
bool sincos_restatement_matches_host_libm() {
	for (int i = -1782580; i <= 1782580; i++) {
		const float y = (float)i * 0x1p-20f;
		if (v2_float_bits(sin_or_cos_ref(y, 0)) != v2_float_bits(sinf(y)) || v2_float_bits(sin_or_cos_ref(y, 1)) != v2_float_bits(cosf(y))) return false;
	}
	for (float y = 1.0f; y > 1e-30f; y *= 0.9990234375f)
		if (v2_float_bits(sin_or_cos_ref(y, 0)) != v2_float_bits(sinf(y)) || v2_float_bits(sin_or_cos_ref(-y, 1)) != v2_float_bits(cosf(-y))) return false;
	return true;
}

struct V2Lane { DecReg r; V2Tracker t; float pll_phase; int pll_last; };
#ifdef V2_PROF // experiment build: cycles of the engine's phases, summed over the waves and launches (tools/build_variant.sh v2prof -DV2_PROF)
__device__ unsigned long long v2_prof[16];
__device__ unsigned long long v2_prof_wg[1024][4]; // per workgroup (last launch wins): tracker wave's run, FM wave's run, FM wave's resident time, messages
#define V2P_T0() const unsigned long long v2p_t0 = __builtin_readcyclecounter()
#define V2P_ADD(slot) do { if (lane == 0) atomicAdd(&v2_prof[slot], __builtin_readcyclecounter() - v2p_t0); } while (0)
void v2_prof_dump() {
	unsigned long long h[16];
	if (hipMemcpyFromSymbol(h, HIP_SYMBOL(v2_prof), sizeof h) != hipSuccess) return;
	fprintf(stderr, "v2_prof:");
	for (int i = 0; i < 16; i++) fprintf(stderr, " %llu", h[i]);
	fprintf(stderr, "\n");
	static unsigned long long w[1024][4];
	if (hipMemcpyFromSymbol(w, HIP_SYMBOL(v2_prof_wg), sizeof w) != hipSuccess) return;
	for (int i = 0; i < 1024; i++) if (w[i][2]) fprintf(stderr, "v2_wg %d %llu %llu %llu %llu\n", i, w[i][0], w[i][1], w[i][2], w[i][3]);
}
#else
#define V2P_T0() do {} while (0)
#define V2P_ADD(slot) do {} while (0)
#endif

__device__ __forceinline__ c2 v2_dot17(const float2* a, const float* taps) { // dot17 (:38-45)
	c2 sum = { 0.0f, 0.0f };
#pragma unroll
	for (int i = 0; i < 8; i++) {
		const float2 u = a[i], v = a[16 - i];
		const c2 w = c2{ u.x + v.x, u.y + v.y };
		sum = sum + w * taps[i];
	}
	const float2 m = a[8];
	return sum + c2{ m.x, m.y } * taps[8];
}
__device__ __forceinline__ int v2_track(V2Tracker& t, c2 z, bool training, float w_train, float w_track) { // PhaseTracker::Run (:190-223)
	const unsigned rot = t.rot;
	const float sre = (rot & 1) ? z.y : z.x, sim = (rot & 1) ? z.x : z.y;
	const float zr = ((rot ^ (rot >> 1)) & 1) ? -sre : sre, zi = (rot & 2) ? -sim : sim;
	t.rot = (rot + 1) & 3;
	const float alpha = training ? w_train : w_track;
	const float beta = 1.0f - alpha;
	const float proj = zr * t.s.x + zi * t.s.y;
	const float d = proj >= 0.0f ? 1.0f : -1.0f;
	const float bd = beta * d;
	t.s = make_float2(alpha * t.s.x + bd * zr, alpha * t.s.y + bd * zi);
	const int decision = proj > 0.0f ? 1 : 0;
	const int bit = decision ^ t.prev_decision;
	t.prev_decision = decision;
	return bit;
}
// PhaseTracker::Run (:190-223) behind its Rotate90 (:175-188): (zr, zi) is the sample already turned by the tracker's `rot` (kv2_engine's
// three-wave form turns the block's FilterFL17 outputs where it computes them, lanes over time; t.rot is advanced once per block)
__device__ __forceinline__ int v2_track_pre(V2Tracker& t, float zr, float zi, bool training, float w_train, float w_track) {
	const float alpha = training ? w_train : w_track;
	const float beta = 1.0f - alpha;
	const float proj = zr * t.s.x + zi * t.s.y;
	const float bd = beta * (proj >= 0.0f ? 1.0f : -1.0f);
	t.s = make_float2(alpha * t.s.x + bd * zr, alpha * t.s.y + bd * zi);
	const int decision = proj > 0.0f ? 1 : 0;
	const int bit = decision ^ t.prev_decision;
	t.prev_decision = decision;
	return bit;
}
__device__ __forceinline__ bool v2_pll(float& phase, int& last_bit, int bit, bool training) { // BitPLL::Run (:225-242)
	if (bit != last_bit) phase += (0.5f - phase) * (training ? 0.6f : 0.05f);
	last_bit = bit;
	phase += 0.2f;
	if (phase < 1.0f) return false;
	phase -= (float)(int)phase;
	return true;
}
__device__ __forceinline__ void v2_reset(DecReg& r) { r.state = DST_TRAINING; r.position = 0; r.osc = 0; }

// Round 5: ONE channel per wave (round 4 had ten channels x six lanes in a wave: 52 waves on a chip of 1,024 SIMDs, every wave as slow
// as its slowest channel and every roll-back of one channel paid by ten).  What the reference computes block-wise is block-wise here,
// with lanes over time, and only the decoders' loop is serial:
//  * Derotate (:133-146): the phasor is rounded at every step, so every lane runs the recurrence and keeps the state in front of its
//    own eight samples (512 steps of three packed operations -- the chain a single lane needed anyway), then derotates those eight;
//  * FilterFL17 (:154-167, Engine::processBlock's filter17.Run(freq_corrected, coh_filtered), :351): all 512 outputs once per block
//    into LDS, eight per lane -- the group loop and its roll-back read them instead of filtering per decoder and sample;
//  * the group loop: five tracker lanes + the FM decoder's lane as before; while none of the six decoders is inside a frame
//    (DATAFCS) -- four fifths of the time on the bench signal -- the decoder step is its TRAINING / STARTFLAG half only.
// The channel's scalar state (phasor, slot predictor, sample index ...) lives in every lane, identically: no leader, no broadcasts.
__global__ __launch_bounds__(64) void kv2_engine(KV2EParams q) {
	__shared__ uint32_t fdata[DEC_DATA_WORDS * 64]; // [word][lane]
	__shared__ __attribute__((aligned(16))) float2 dero[16 + 512 + 2]; // FilterFL17's carry, then the block: raw, derotated in place
	__shared__ __attribute__((aligned(16))) float2 X[584];              // Estimate()'s exchange space; afterwards the block's 512 FilterFL17 outputs
	__shared__ __attribute__((aligned(16))) float mag[512 + 8];
	__shared__ uint32_t fmw[16];
	float2* const zb = X;
	const KV2Params& p = q.k;
	const int lane = threadIdx.x, j = lane;
	const int chan = blockIdx.x;
	const bool dl = lane < 6; // the six decoder lanes: 0..4 behind the trackers, 5 the FM decoder behind its BitPLL
	const int dec = chan * 6 + (dl ? j : 0);
	uint32_t* data = fdata + lane;
	V2ChanState* cs = q.st + chan;
	V2Lane L;
	{
		const DecState* st = q.dec + dec;
		DecReg& r = L.r;
		r.state = st->state; r.lastBit = st->lastBit; r.prev = st->prev; r.position = st->position; r.osc = st->osc;
		r.level = st->level; r.start_idx = st->start_idx;
		for (int w = 0; w < DEC_DATA_WORDS; w++) data[64 * w] = st->data[w];
		r.crc = st->crc[0]; r.cw = st->crc[1]; r.cwi = (int)st->crc[2]; r.tail = st->crc[3]; r.abort_pos = (int)st->crc[4];
		L.t = cs->trk[j < 5 ? j : 0];
		L.pll_phase = cs->pll_phase; L.pll_last = cs->pll_last;
	}
	float2 rot = cs->rot, slot_ema = cs->slot_ema;
	float last_f = cs->last_f, ppm = cs->ppm, ppm_prev = cs->ppm_prev;
	int slot_phase = cs->slot_phase, di = cs->di;
	long long sample_idx = cs->sample_idx;
	if (lane < 16) dero[lane] = cs->carry17[lane];
	FftTwiddles tw = fft_twiddles(p.omega, lane);
	const uint32_t* fm_cur = p.fmbits + (size_t)chan * p.fmbits_stride;
	const uint32_t* fm_old = q.fm_prev + (size_t)chan * 16; // the previous block's last sixteen words (kv2_carry's fmtail_out)
	// sign of the filtered discriminator at sample k of the engine block that is being decoded (the block's sixteen words: LDS)
	const auto fm_sign = [&](int k) -> int { return (int)((fmw[k >> 5] >> (k & 31)) & 1u); };
	const auto emit = [&](const DecReg& r, long long sidx, float tag_ppm) {
		const unsigned slot = atomicAdd(q.frame_count, 1u) % (unsigned)q.max_frames;
		uint32_t* f = q.frames + (size_t)slot * DEC_FRAME_WORDS;
		f[0] = (uint32_t)dec; f[1] = __float_as_uint(tag_ppm); f[2] = (uint32_t)r.position; f[3] = __float_as_uint(r.level);
		f[4] = (uint32_t)(unsigned long long)r.start_idx; f[5] = (uint32_t)((unsigned long long)r.start_idx >> 32);
		f[6] = (uint32_t)(unsigned long long)sidx; f[7] = (uint32_t)((unsigned long long)sidx >> 32);
		f[8] = q.block; f[9] = q.sub;
		for (int w = 0; w < DEC_DATA_WORDS; w++) f[10 + w] = data[64 * w];
	};
	const auto learn_slot = [&](long long start_idx) { // learnSlotPhase (:328-337), every lane alike
		const long long a = start_idx - 155;
		const int m = (int)((a % 1280 + 1280) % 1280);
		const float2 csv = q.slot_cs[m];
		slot_ema = make_float2((1.0f - 0.2f) * slot_ema.x + 0.2f * csv.x, (1.0f - 0.2f) * slot_ema.y + 0.2f * csv.y);
		const float ph = atan2f_ref(slot_ema.y, slot_ema.x) * (1280.0f / (2.0f * 3.14159265358979323846f));
		slot_phase = (int)(ph + 1280.0f + 0.5f) % 1280;
	};
	// FreqOffset::Derotate (:133-146) over samples [from, to) of the staged block: lane l owns samples from + 8 l .. + 7
	const auto derotate = [&](float f, int from, int to) {
		const float th = f * 2.0f * 3.14159265358979323846f; // std::polar(1.0f, f * 2.0f * PI): (rho * cos(theta), rho * sin(theta))
		const float sn = 1.0f * sin_or_cos_ref(th, 0), cn = 1.0f * sin_or_cos_ref(th, 1);
		const c2 st = { cn, sn }, st_sw = { -sn, cn };
		c2 r = { rot.x, rot.y }, mine = r;
		const int n = to - from;
		for (int c = 0; c * 8 < n; c++) { // the state in front of sample from + 8 c: lane c's
			if (c == lane) mine = r;
			const int m = n - c * 8 < 8 ? n - c * 8 : 8;
			if (m == 8) {
#pragma unroll
				for (int i = 0; i < 8; i++) r = r.xx * st + r.yy * st_sw; // r *= rot_step
			} else {
				for (int i = 0; i < m; i++) r = r.xx * st + r.yy * st_sw;
			}
		}
		float2* d = &dero[16 + from + 8 * lane];
#pragma unroll
		for (int i = 0; i < 8; i++) {
			if (8 * lane + i < n) {
				const float2 x = d[i];
				mine = mine.xx * st + mine.yy * st_sw;
				d[i] = make_float2(x.x * mine.x - x.y * mine.y, x.x * mine.y + x.y * mine.x); // src * r
			}
		}
		const float a = hypot_ref(r.x, r.y);
		rot = make_float2(__fdiv_rn(r.x, a), __fdiv_rn(r.y, a));
		last_f = f;
	};

	for (int blk = 0; blk < p.n_windows; blk++) {
		const int n0 = -V2_HIST + 512 * blk; // the decoded block; [n0 + 512, n0 + 1024) is the look-ahead
#pragma unroll
		for (int i = 0; i < 8; i++) dero[16 + i * 64 + lane] = v2_sample(p, chan, n0 + i * 64 + lane);
		if (lane < 16) { // (n0 is a multiple of 512: whole words; block 0 decodes the previous device block's tail)
			const int m0 = n0 < 0 ? n0 + p.L : n0;
			fmw[lane] = n0 < 0 ? fm_old[lane] : fm_cur[(m0 >> 5) + lane];
		}
		wave_sync();
		// ---- Engine::processBlock (:345-352): slot predictor decay, busy, CGF
		const bool busy = __ballot(dl && j < 5 && L.r.state != DST_TRAINING) != 0;
		slot_ema = make_float2(slot_ema.x * 0.9999f, slot_ema.y * 0.9999f);
		const bool locked = slot_ema.x * slot_ema.x + slot_ema.y * slot_ema.y >= 0.64f;
		const int e_slot = (int)((((long long)slot_phase - sample_idx) % 1280 + 1280) % 1280);
		ppm_prev = ppm;
		int split = 0;
		float f = 0.0f;
		if (locked && e_slot < 512) {
			// a slot starts inside this block: [0, e) keeps the previous frequency, Estimate() works on the 512 samples from e on -- a window
			// the assist kernels did not compute: FFT by the wave, the sequential search by one lane
			split = e_slot;
			derotate(last_f, 0, split); // (also for e == 0: the renormalisation happens)
			v2_fft_mag_window(p, chan, n0 + split, X, mag, tw, lane);
			wave_sync();
			if (lane == 0) { float prom; v2_search(mag, f, prom); if (q.locked_estimates) atomicAdd(q.locked_estimates, 1); }
			f = __shfl(f, 0);
			wave_sync();
		} else {
			const float* en = p.energy + (size_t)chan * (p.n_windows + 1);
			const bool louder = en[blk + 1] > en[blk];
			const int w = 2 * blk + ((!busy && louder) ? 1 : 0);
			f = p.est_f[(size_t)chan * 2 * p.n_windows + w];
			const float prom = p.est_prom[(size_t)chan * 2 * p.n_windows + w];
			if (busy && prom < 5.5f) f = last_f; // tone gate: hold while a decode is in flight
		}
		derotate(f, split, 512);
		ppm = __fdiv_rn(f * 48000.0f, 162.0f);
		wave_sync();
		// ---- FilterFL17 (:154-167) of the whole block: output k from carry + block samples k .. k + 16
#pragma unroll
		for (int i = 0; i < 8; i++) {
			const c2 z = v2_dot17(&dero[i * 64 + lane], q.taps17);
			zb[i * 64 + lane] = make_float2(z.x, z.y);
		}
		wave_sync();
		// ---- the 512 samples: tracker di handles sample i (di runs on across blocks), the FM decoder sees every sample through its PLL
		const int off = j < 5 ? (j - di + 5) % 5 : 0; // this tracker's sample inside a group of five
		// (a generic lambda: the 102 whole groups of a block are one straight-line body, the two samples behind them another instance)
		const auto group = [&](const int g5, auto whole_group) {
			constexpr bool WHOLE = decltype(whole_group)::value != 0;
			const V2Lane before = L;
			const int ng = WHOLE ? 5 : 512 - g5;
			// The FM decoder's lane: its BitPLL (:225-242) over the group's samples.  The symbol of the first sample on which it fires goes
			// through the decoder together with the trackers'; the samples behind that one see the PLL gain of the decoder's state AFTER
			// that step in the reference.  Here all five run first, straight-line, with the gain of the state before the group: exact
			// unless the step flips TRAINING <-> not-TRAINING AND a later sample of the group has a sign change (the only place the gain
			// enters) -- then, as for a second symbol inside one group, the group is redone sample by sample below.
			const unsigned fm5 = (unsigned)(((((unsigned long long)fmw[(g5 >> 5) + 1 < 16 ? (g5 >> 5) + 1 : 15]) << 32) | fmw[g5 >> 5]) >> (g5 & 31));
			const bool tr0 = L.r.state == DST_TRAINING;
			int k_pll = -1;
			bool again = false, chg_after = false;
			{
				float ph = L.pll_phase;
				int lastb = L.pll_last;
				const float gain = tr0 ? 0.6f : 0.05f;
#pragma unroll
				for (int s5 = 0; s5 < 5; s5++) {
					if (s5 < ng) { // (wave-uniform: only a block's last group is short)
						const int b = (int)((fm5 >> s5) & 1u);
						const bool chg = b != lastb;
						const float ph_c = ph + (0.5f - ph) * gain;
						ph = chg ? ph_c : ph;
						lastb = b;
						ph += 0.2f;
						const bool fire = !(ph < 1.0f);
						const float ph_w = ph - (float)(int)ph;
						ph = fire ? ph_w : ph;
						chg_after = chg_after || (chg && k_pll >= 0);
						again = again || (fire && k_pll >= 0);
						k_pll = (fire && k_pll < 0) ? g5 + s5 : k_pll;
					}
				}
				if (j == 5) { L.pll_phase = ph; L.pll_last = lastb; }
			}
			const int my_k = j == 5 ? k_pll : ((j < 5 && off < ng) ? g5 + off : -1);
			const bool have = dl && my_k >= 0;
			const int kk = have ? my_k : 0;
			const float2 zf = zb[kk];
			const c2 z = { zf.x, zf.y };
			const int bit = j < 5 ? v2_track(L.t, z, tr0, q.w_train, q.w_track) : (int)((fm5 >> (kk - g5)) & 1u);
			if (!have && j < 5) L.t = before.t;
			const float slvl = z.x * z.x + z.y * z.y;
			const long long sidx = sample_idx + kk;
			// no decoder of the channel inside a frame: the step's TRAINING / STARTFLAG half is the whole step (it cannot complete a message)
			const bool in_frame = __ballot(dl && L.r.state == DST_DATAFCS) != 0;
			bool found = false;
			if (in_frame) { if (have) found = dec_step(L.r, bit, slvl, sidx, data); }
			else dec_step_idle(L.r, bit, sidx, have ? 1 : 0); // (every lane, no branch)
			again = j == 5 && (again || (chg_after && (L.r.state == DST_TRAINING) != tr0));
			// anything that breaks the lockstep -- a completed message (it resets the other five at ITS sample), or an FM decoder that
			// clocks two symbols inside one group -- sends the channel through the reference's own order, sample by sample
			if (__ballot(found || (again && dl)) != 0) {
				L = before;
				// (the frame buffers: a lane that is rolled back may have written a word of its column: dec_step rewrites what it needs)
				for (int s5 = 0; s5 < ng; s5++) {
					const int k5 = g5 + s5;
					const float tag_ppm = k5 >= split ? ppm : ppm_prev;
					const long long si = sample_idx + k5;
					const float2 zq = zb[k5];
					const c2 zz = { zq.x, zq.y };
					const float lv = zz.x * zz.x + zz.y * zz.y;
					bool fnd = false;
					if (j < 5 && off == s5) {
						const int b2 = v2_track(L.t, zz, L.r.state == DST_TRAINING, q.w_train, q.w_track);
						fnd = dec_step(L.r, b2, lv, si, data);
						if (fnd) emit(L.r, si, tag_ppm);
					}
					unsigned long long FF = __ballot(fnd);
					if (FF != 0) { // learnSlotPhase(dec[di]) + resetDecoders()
						const long long sidx0 = __shfl(L.r.start_idx, __builtin_ctzll(FF));
						learn_slot(sidx0);
						v2_reset(L.r);
					}
					fnd = false;
					if (j == 5 && v2_pll(L.pll_phase, L.pll_last, fm_sign(k5), L.r.state == DST_TRAINING)) {
						fnd = dec_step(L.r, fm_sign(k5), lv, si, data);
						if (fnd) emit(L.r, si, tag_ppm);
					}
					FF = __ballot(fnd);
					if (FF != 0) v2_reset(L.r);
				}
			}
		};
		{
#pragma unroll 1
			for (int g5 = 0; g5 + 5 <= 512; g5 += 5) group(g5, K1Const<1>{});
			group(510, K1Const<0>{}); // 512 = 102 x 5 + 2
		}
		sample_idx += 512;
		di = (di + 512) % 5;
		wave_sync();
		if (lane < 16) dero[lane] = dero[512 + lane]; // FilterFL17's carry
		wave_sync();
	}
	if (dl) {
		DecState* st = q.dec + dec;
		const DecReg& r = L.r;
		st->state = r.state; st->lastBit = r.lastBit; st->prev = r.prev; st->position = r.position; st->osc = r.osc;
		st->level = r.level; st->start_idx = r.start_idx;
		data[64 * r.cwi] = r.cw;
		for (int w = 0; w < DEC_DATA_WORDS; w++) st->data[w] = data[64 * w];
		st->crc[0] = r.crc; st->crc[1] = r.cw; st->crc[2] = (uint32_t)r.cwi; st->crc[3] = r.tail; st->crc[4] = (uint32_t)r.abort_pos;
		if (j < 5) cs->trk[j] = L.t;
		if (j == 5) { cs->pll_phase = L.pll_phase; cs->pll_last = L.pll_last; }
	}
	if (lane == 0) {
		cs->rot = rot; cs->last_f = last_f; cs->ppm = ppm; cs->ppm_prev = ppm_prev; cs->slot_ema = slot_ema; cs->slot_phase = slot_phase;
		cs->di = di; cs->sample_idx = sample_idx;
	}
	if (lane < 16) cs->carry17[lane] = dero[lane];
}


// ------------------------------------------------------------------------------------------
// Round 6: kv2_engine_roles -- the engine of one channel on THREE waves of a workgroup.  A wave that is alone on its SIMD pays 5-9
// cycles per instruction whatever it does (tools/microbench_lonewave.hip), so what counts is the length of the longest dependent
// instruction stream per 512-sample block, and the engine has three streams that meet only at rare events:
//  * wave 0: the five PhaseTrackers with their decoders, a group of five samples per turn (lanes 0..4);
//  * wave 1: the FM decoder behind its BitPLL (one lane; the turn as vector-side arithmetic);
//  * wave 2: Engine::CGF + Derotate + FilterFL17 of the NEXT block while the other two decode this one.  What CGF needs from the
//    decoders is one bit (busy: some tracker's decoder is not in TRAINING when the block starts, :347-349) and the slot predictor,
//    which only a completed message moves: the wave prepares the block for both answers where they differ (two frequencies ->
//    two derotated, filtered, pre-rotated blocks in LDS), and the right one is picked when the decoders arrive.  A message completed by
//    a tracker's decoder in between (learnSlotPhase) voids the preparation: the block is prepared again, in the open.
// Waves 0 and 1 speculate that nobody completes a message inside a block: a wave that does complete one stops there, the waves exchange
// the positions at the end of the block, and where there is one, both restore the state they had at the last agreed position, run up to
// the earliest completion k* exactly, handle sample k* in the reference's order (the tracker's decoder first: message out,
// learnSlotPhase, all six reset; then the FM decoder), and speculate on from k* + 1.  Completed messages leave the kernel only from
// that exact pass.  Everything is the reference's arithmetic in the reference's order.
// ------------------------------------------------------------------------------------------
#define V2R_KERNEL_HEAD __global__ __launch_bounds__(192) void kv2_engine_roles(KV2EParams q)
#include "kv2_roles.inc"
#undef V2R_KERNEL_HEAD
// (amdgpu_num_vgpr takes half of the budget on this compiler: see check_resources.py)
#define V2R_KERNEL_HEAD __global__ __launch_bounds__(192) __attribute__((amdgpu_num_vgpr(168 / 2))) void kv2_engine_roles_dense(KV2EParams q)
#include "kv2_roles.inc"
#undef V2R_KERNEL_HEAD

// ------------------------------------------------------------------------------------------
// K7 for the other engines, sequential forms (ModelStandard and ModelChallenger run event-driven by default -- K7e below, kinds 1 / 2 --
// and come here with AISGPU_K7=seq or blocks of more than 8191 groups; ModelBase always).
//  * k7_pack_fm: the FM receivers hand every 48 kHz sample n to decoder n % 5 (Deinterleave, DSP.h:51-74): regroup the sign bits
//    of the filtered discriminator per decoder (row j, bit g = sample 5 g + j), for the groups completed inside this block
//    (the first one may have started in the previous block: those bits come from the previous block's row).
//  * k7_decode_mesh<1>: ModelStandard -- five decoders with their Reset mesh on those rows (tag.sample_lvl is never set in
//    this engine: level 0; tag.sample_idx = n).
//  * k7_decode_mesh<2>: ModelChallenger -- ten decoders per channel.  Per group the reference runs FM0..FM3 (samples 5g..5g+3),
//    then, with sample 5g+4, the five coherent decoders and FM4 (Model.cpp:630-639); any of the ten that completes
//    a message resets the other nine (Model.cpp:658-674).  The ten step together in that order's lanes and the order is
//    restored by the same roll-back as in k7_decode.  tag.sample_lvl is written by ScatterPLL when a group completes, so
//    FM0..FM3 still see the previous group's level.
//  * k7_base: ModelBase -- DSP::SimplePLL (DSP.cpp:28-44) in front of ONE decoder per channel, sequential over the 48 kHz
//    samples; the sampler's fast / slow loop follows the decoder's StartTraining / StopTraining signals (DSP.cpp:46-57,
//    AIS.cpp:41-46), which amounts to "the decoder is in TRAINING".  tag.sample_idx / sample_lvl are never set in this engine.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k7_pack_fm(K7Params p) {
	const int dec = blockIdx.y, chan = dec / 5, j = dec - 5 * chan;
	const int w = blockIdx.x * 256 + threadIdx.x;
	if (w >= (int)p.fmrows_stride) return;
	const uint32_t* cur = p.fm_cur + (size_t)chan * p.fm_stride;
	const uint32_t* prev = p.fm_prev + (size_t)chan * p.fm_stride;
	uint32_t word = 0;
	for (int e = 0; e < 32; e++) {
		const int g = 32 * w + e;
		if (g >= p.n_groups) break;
		const int n = p.n_rel0 + 5 * g + j; // block-relative sample
		const uint32_t b = n >= 0 ? (cur[n >> 5] >> (n & 31)) & 1u : (prev[(p.L + n) >> 5] >> ((p.L + n) & 31)) & 1u;
		word |= b << e;
	}
	p.fmrows[(size_t)dec * p.fmrows_stride + w] = word;
}

template <int KIND> // 1: ModelStandard (mesh of 5 on the FM rows), 2: ModelChallenger (mesh of 10)
__global__ __launch_bounds__(64) void k7_decode_mesh(K7Params p) {
	constexpr int MESH = KIND == 2 ? 10 : 5, PER_WAVE = 60 / MESH;
	__shared__ uint32_t fdata[DEC_DATA_WORDS * 64]; // [word][lane]
	__shared__ float lv[64 * 33];
	const int lane = threadIdx.x;
	if (p.cond) { // exact fallback of the event-driven kernels (see k7_decode)
		if (*p.cond == 0) return;
		if (blockIdx.x == 0 && lane == 0 && p.cond_count) atomicAdd(p.cond_count, 1);
	}
	const int mesh = lane / MESH, o = lane - MESH * mesh; // o: position in the reference's order within a group
	const int chan_raw = blockIdx.x * PER_WAVE + mesh;
	const bool live = lane < 60 && chan_raw < p.n_chan;
	const int chan = live ? chan_raw : 0;
	// ModelChallenger: o = 0..3 FM0..FM3, 4..8 the coherent decoders 0..4, 9 FM4
	const bool is_fm = KIND == 1 || o < 4 || o == 9;
	const int j = KIND == 1 ? o : (o < 4 ? o : o == 9 ? 4 : o - 4);
	const int dec = chan * MESH + (live ? o : 0);
	uint32_t* data = fdata + lane;
	DecState* st = p.state + dec;
	DecReg r;
	r.state = st->state; r.lastBit = st->lastBit; r.prev = st->prev; r.position = st->position; r.osc = st->osc;
	r.level = st->level; r.start_idx = st->start_idx;
	for (int w = 0; w < DEC_DATA_WORDS; w++) data[64 * w] = st->data[w];
	r.crc = st->crc[0]; r.cw = st->crc[1]; r.cwi = (int)st->crc[2]; r.tail = st->crc[3]; r.abort_pos = (int)st->crc[4];
	const uint32_t* brow = is_fm ? p.fmrows + (size_t)(chan * 5 + j) * p.fmrows_stride : p.bits + (size_t)(chan * 5 + j) * p.bits_stride;
	const float* lrow = KIND == 2 ? p.lvl + (size_t)chan * p.lvl_stride : nullptr;
	// tag.sample_lvl as FM0..FM3 of a group see it: what the previous group's ScatterPLL left.  The TAG is ONE object per device,
	// shared by both channels, and Rotate hands channel A its whole block before channel B (DSP.cpp:312-313): the first FM
	// samples of channel A's block still see channel B's last level of the PREVIOUS block, those of channel B's block see
	// channel A's last level of THIS block; samples of a group that began in the previous block saw their own channel's.
	float lvl_prev = 0.0f, lvl_other = 0.0f;
	if (KIND == 2) {
		lvl_prev = p.last_lvl_in[chan];
		lvl_other = (chan & 1) ? (p.n_groups > 0 ? p.lvl[(size_t)(chan ^ 1) * p.lvl_stride + p.n_groups - 1] : p.last_lvl_in[chan ^ 1]) : p.last_lvl_in[chan ^ 1];
	}
	const unsigned mesh_mask = (1u << MESH) - 1u;
	for (int g0 = 0; g0 < p.n_groups; g0 += 32) {
		const uint32_t word = brow[g0 >> 5];
		const int n = p.n_groups - g0 < 32 ? p.n_groups - g0 : 32;
		if (KIND == 2) {
			const float4* src = reinterpret_cast<const float4*>(lrow + g0);
			float4 t[8];
#pragma unroll
			for (int q = 0; q < 8; q++) t[q] = src[q];
#pragma unroll
			for (int q = 0; q < 8; q++) { float* d = &lv[lane * 33 + 4 * q]; d[0] = t[q].x; d[1] = t[q].y; d[2] = t[q].z; d[3] = t[q].w; }
		}
		for (int e = 0; e < n; e++) {
			const int g = g0 + e;
			const int dd = (int)((word >> e) & 1u);
			const float lv_g = KIND == 2 ? lv[lane * 33 + e] : 0.0f;
			float slvl = 0.0f;
			if (KIND == 2) {
				slvl = (is_fm && o < 4) ? lvl_prev : lv_g;
				if (g == 0 && is_fm && o < 4 && p.n_rel0 + j >= 0) slvl = lvl_other; // the block's first samples: the other channel ran last
			}
			lvl_prev = lv_g;
			const long long sidx = 5 * (p.first_group + g) + j; // coherent: ScatterPLL's sample_idx; FM: the sample number (Deinterleave)
			const DecReg before = r;
			bool found = live && dec_step(r, dd, slvl, sidx, data);
			const unsigned long long F = __ballot(found);
			if (F != 0) { // rare: restore the order in which the reference runs the decoders of a group
				const unsigned mm = (unsigned)(F >> (MESH * mesh)) & mesh_mask;
				if (live && mm != 0) {
					const int omin = __builtin_ctz(mm);
					if (o > omin) { // would have been reset before its step
						r = before;
						r.state = DST_TRAINING; r.position = 0; r.osc = 0;
						dec_step(r, dd, slvl, sidx, data);
					} else if (o < omin) { // reset after its step
						r.state = DST_TRAINING; r.position = 0; r.osc = 0;
					} else { // the decoder that found the message
						const unsigned slot = atomicAdd(p.frame_count, 1u) % (unsigned)p.max_frames;
						uint32_t* f = p.frames + (size_t)slot * DEC_FRAME_WORDS;
						f[0] = (uint32_t)dec; f[1] = (uint32_t)g; f[2] = (uint32_t)r.position; f[3] = __float_as_uint(r.level);
						f[4] = (uint32_t)(unsigned long long)r.start_idx; f[5] = (uint32_t)((unsigned long long)r.start_idx >> 32);
						f[6] = (uint32_t)(unsigned long long)sidx; f[7] = (uint32_t)((unsigned long long)sidx >> 32);
						f[8] = p.block; f[9] = p.sub;
						for (int w = 0; w < DEC_DATA_WORDS; w++) f[10 + w] = data[64 * w];
						r.state = DST_TRAINING; r.position = 0; r.osc = 0;
					}
				}
			}
		}
	}
	if (live) {
		st->state = r.state; st->lastBit = r.lastBit; st->prev = r.prev; st->position = r.position; st->osc = r.osc;
		st->level = r.level; st->start_idx = r.start_idx;
		data[64 * r.cwi] = r.cw;
		for (int w = 0; w < DEC_DATA_WORDS; w++) st->data[w] = data[64 * w];
		st->crc[0] = r.crc; st->crc[1] = r.cw; st->crc[2] = (uint32_t)r.cwi; st->crc[3] = r.tail; st->crc[4] = (uint32_t)r.abort_pos;
		if (KIND == 2 && o == 0) p.last_lvl[chan] = lvl_prev; // (a block without a complete group hands on what it was given)
	}
}

__global__ __launch_bounds__(64) void k7_base(K7Params p) {
	__shared__ uint32_t fdata[DEC_DATA_WORDS * 64]; // [word][lane]
	const int lane = threadIdx.x;
	const int chan_raw = blockIdx.x * 64 + lane;
	bool live = chan_raw < p.n_chan;
	if (p.cond) { // exact fallback of the chunk-parallel kernels (k7b_*): only the channels they flagged, from the untouched carried state
		const bool mine = live && p.cond[chan_raw] != 0;
		if (!__any(mine)) return;
		if (mine) { p.cond[chan_raw] = 0; atomicAdd(p.cond_count, 1); }
		live = mine;
	}
	const int chan = live ? chan_raw : 0;
	uint32_t* data = fdata + lane;
	DecState* st = p.state + chan;
	DecReg r;
	r.state = st->state; r.lastBit = st->lastBit; r.prev = st->prev; r.position = st->position; r.osc = st->osc;
	r.level = st->level; r.start_idx = st->start_idx;
	for (int w = 0; w < DEC_DATA_WORDS; w++) data[64 * w] = st->data[w];
	r.crc = st->crc[0]; r.cw = st->crc[1]; r.cwi = (int)st->crc[2]; r.tail = st->crc[3]; r.abort_pos = (int)st->crc[4];
	float pll = __uint_as_float(st->crc[5]); // SimplePLL::PLL
	int pprev = (int)st->crc[6];              // SimplePLL::prev
	const uint32_t* brow = p.fm_cur + (size_t)chan * p.fm_stride;
	for (int n0 = 0; n0 < p.L; n0 += 32) {
		const uint32_t word = brow[n0 >> 5];
		for (int e = 0; e < 32; e++) {
			const int bit = (int)((word >> e) & 1u);       // data[i] > 0
			const bool fast = r.state == DST_TRAINING;     // FastPLL (StartTraining / StopTraining, AIS.cpp:41-46)
			if (bit != pprev) pll += (0.5f - pll) * (fast ? 0.6f : 0.05f);
			pll += 0.2f;
			const bool emit = pll >= 1.0f;
			if (emit) pll -= (float)(int)pll;
			pprev = bit;
			bool found = false;
			if (live && emit) found = dec_step(r, bit, 0.0f, 0ll, data); // tag.sample_lvl / sample_idx are never set in this engine
			if (found) {
				const unsigned slot = atomicAdd(p.frame_count, 1u) % (unsigned)p.max_frames;
				uint32_t* f = p.frames + (size_t)slot * DEC_FRAME_WORDS;
				f[0] = (uint32_t)chan; f[1] = (uint32_t)(n0 + e); f[2] = (uint32_t)r.position; f[3] = __float_as_uint(r.level);
				f[4] = 0; f[5] = 0; f[6] = 0; f[7] = 0;
				f[8] = p.block; f[9] = p.sub;
				for (int w = 0; w < DEC_DATA_WORDS; w++) f[10 + w] = data[64 * w];
				r.state = DST_TRAINING; r.position = 0; r.osc = 0;
			}
		}
	}
	if (live) {
		st->state = r.state; st->lastBit = r.lastBit; st->prev = r.prev; st->position = r.position; st->osc = r.osc;
		st->level = r.level; st->start_idx = r.start_idx;
		data[64 * r.cwi] = r.cw;
		for (int w = 0; w < DEC_DATA_WORDS; w++) st->data[w] = data[64 * w];
		st->crc[0] = r.crc; st->crc[1] = r.cw; st->crc[2] = (uint32_t)r.cwi; st->crc[3] = r.tail; st->crc[4] = (uint32_t)r.abort_pos;
		st->crc[5] = __float_as_uint(pll); st->crc[6] = (uint32_t)pprev;
	}
}

// ------------------------------------------------------------------------------------------
// K7b: ModelBase's SimplePLL + decoder loop, chunk-parallel (see kernels.h)
// ------------------------------------------------------------------------------------------
struct BaseReg { DecReg r; float pll; int pprev; };

__device__ __forceinline__ void base_load(BaseReg& b, const DecState* st, uint32_t* data) {
	DecReg& r = b.r;
	r.state = st->state; r.lastBit = st->lastBit; r.prev = st->prev; r.position = st->position; r.osc = st->osc;
	r.level = st->level; r.start_idx = st->start_idx;
	for (int w = 0; w < DEC_DATA_WORDS; w++) data[64 * w] = st->data[w];
	r.crc = st->crc[0]; r.cw = st->crc[1]; r.cwi = (int)st->crc[2]; r.tail = st->crc[3]; r.abort_pos = (int)st->crc[4];
	b.pll = __uint_as_float(st->crc[5]); b.pprev = (int)st->crc[6];
}
__device__ __forceinline__ void base_store(const BaseReg& b, DecState* st, uint32_t* data) {
	const DecReg& r = b.r;
	st->state = r.state; st->lastBit = r.lastBit; st->prev = r.prev; st->position = r.position; st->osc = r.osc;
	st->level = r.level; st->start_idx = r.start_idx;
	data[64 * r.cwi] = r.cw;
	for (int w = 0; w < DEC_DATA_WORDS; w++) st->data[w] = data[64 * w];
	st->crc[0] = r.crc; st->crc[1] = r.cw; st->crc[2] = (uint32_t)r.cwi; st->crc[3] = r.tail; st->crc[4] = (uint32_t)r.abort_pos;
	st->crc[5] = __float_as_uint(b.pll); st->crc[6] = (uint32_t)b.pprev; st->crc[7] = 0;
}
__device__ __forceinline__ void base_fresh(BaseReg& b, int pprev, uint32_t* data) {
	DecReg& r = b.r;
	r.state = DST_TRAINING; r.lastBit = 0; r.prev = 0; r.position = 0; r.osc = 0; r.level = 0.0f; r.start_idx = 0;
	r.crc = 0; r.cw = 0; r.cwi = 0; r.tail = 0; r.abort_pos = 0;
	for (int w = 0; w < DEC_DATA_WORDS; w++) data[64 * w] = 0;
	b.pll = 0.0f; b.pprev = pprev;
}
__device__ __forceinline__ K7bCkpt base_ckpt(const BaseReg& b) {
	K7bCkpt c;
	c.pll = __float_as_uint(b.pll); c.position = (uint32_t)b.r.position;
	c.flags = (uint32_t)b.pprev | (uint32_t)b.r.state << 1 | (uint32_t)b.r.lastBit << 3 | (uint32_t)b.r.prev << 4 | (uint32_t)b.r.osc << 5;
	return c;
}
// the two trajectories are the same from here on: identical sampler state, identical decoder state, the decoder in TRAINING (where
// the rest of its registers -- CRC, frame buffer, abort position -- is dead: all of it is initialised again when a frame opens)
__device__ __forceinline__ bool base_same(const K7bCkpt& a, const K7bCkpt& b) {
	return a.pll == b.pll && a.position == b.position && a.flags == b.flags && ((a.flags >> 1) & 3u) == (uint32_t)DST_TRAINING;
}
__device__ __forceinline__ int base_bit(const uint32_t* brow, int n) { return (int)((brow[n >> 5] >> (n & 31)) & 1u); }
// one sample of SimplePLL::Receive (DSP.cpp:28-44); true when the sampler hands this sample to the decoder.
// The loop is one chain of dependent operations in a wave that has nothing else to issue, so it is written for the length of that
// chain, without a branch: a correction with gain +0 leaves pll as it is (x * 0 = +-0, pll + +-0 = pll); pll stays in [0, 1.2) (a
// correction moves it towards 0.5), so the (int)pll that an emission subtracts is 1 and pll - 1 is exact: v_fract_f32 is the wrap.
// (k7_base keeps the reference's own form; the parity tests compare the two.)
__device__ __forceinline__ bool base_pll_step(BaseReg& b, int bit) {
	const uint32_t gain = b.r.state == DST_TRAINING ? __builtin_bit_cast(uint32_t, 0.6f) : __builtin_bit_cast(uint32_t, 0.05f); // FastPLL (StartTraining / StopTraining, AIS.cpp:41-46)
	const uint32_t g = bit != b.pprev ? gain : 0u;
	float pll = b.pll + (0.5f - b.pll) * __uint_as_float(g);
	pll = pll + 0.2f;
	b.pll = __builtin_amdgcn_fractf(pll);
	b.pprev = bit;
	return pll >= 1.0f;
}
// The sampler with a FIXED gain over one whole word of 32 samples, n % 32 == 0 in front of it; the decisions it hands to the decoder
// are appended to `bits` from bit ns on (at most 8 per word with the slow gain, 16 with the fast one).
template <bool FAST>
__device__ __forceinline__ void base_gain_word(float& pll_io, int& pprev, uint32_t w, uint32_t& bits, int& ns) {
	const uint32_t T = w ^ ((w << 1) | (uint32_t)pprev); // sign changes
	float pll = pll_io;
	uint32_t E = 0; // sample i emitted: bit 31 - i
#pragma unroll
	for (int i = 0; i < 32; i++) {
		const uint32_t g = (uint32_t)((int32_t)(T << (31 - i)) >> 31) & __builtin_bit_cast(uint32_t, FAST ? 0.6f : 0.05f);
		pll = pll + (0.5f - pll) * __uint_as_float(g);
		pll = pll + 0.2f;
		E = E + E + (pll >= 1.0f ? 1u : 0u);
		pll = __builtin_amdgcn_fractf(pll);
	}
	pll_io = pll;
	pprev = (int)(w >> 31);
	for (E = __builtin_bitreverse32(E); E; E &= E - 1u) {
		bits |= ((w >> __builtin_ctz(E)) & 1u) << ns;
		ns++;
	}
}
// (the list's length travels in a register: a counter in memory is a round trip per frame)
__device__ __forceinline__ void base_record(BaseReg& b, int n, uint32_t* list, uint32_t& cnt, uint32_t cap, uint32_t* data, bool& overflow) {
	if (cnt < cap) {
		uint32_t* f = list + 1 + cnt * K7B_FREC;
		f[0] = (uint32_t)n; f[1] = (uint32_t)b.r.position;
		for (int w = 0; w < DEC_DATA_WORDS; w++) f[2 + w] = data[64 * w];
		cnt++;
	} else overflow = true;
}
__global__ __launch_bounds__(64) void k7b_spec(K7bParams q) {
	__builtin_amdgcn_s_setprio(3); // a long dependent chain in few waves, beside front-end waves that fill every SIMD: issue first
	__shared__ uint32_t fdata[DEC_DATA_WORDS * 64];
	__shared__ uint32_t rowbits[(K7B_CH + K7B_WARM) / 32 * 64]; // the lanes' bit rows for this chunk (+ warm-up), word i of lane l at [64 i + l]
	const K7Params& p = q.k;
	const int lane = threadIdx.x, c = blockIdx.y;
	const int chan_raw = blockIdx.x * 64 + lane;
	const bool live = chan_raw < p.n_chan;
	const int chan = live ? chan_raw : 0;
	uint32_t* data = fdata + lane;
	const size_t slot = (size_t)c * q.n_chan_pad + chan_raw;
	const uint32_t* brow = p.fm_cur + (size_t)chan * p.fm_stride;
	const int n0 = c * K7B_CH, n1 = n0 + K7B_CH < p.L ? n0 + K7B_CH : p.L;
	// Every chunk is speculative, the block's first one too (its warm-up: the tail of the previous block's row): this kernel then
	// depends on nothing the previous block's decoders leave behind, and runs beside them.
	const uint32_t* prow = p.fm_prev + (size_t)chan * p.fm_stride + p.fm_stride; // word -i of this row = word L / 32 - i of the previous block's
	BaseReg b;
	const int start = n0 - K7B_WARM;
	base_fresh(b, start > 0 ? base_bit(brow, start - 1) : (int)((prow[(start - 1) >> 5] >> ((start - 1) & 31)) & 1u), data);
	uint32_t* list = q.frames + slot * (1 + K7B_FCAP * K7B_FREC);
	uint32_t n_rec = 0;
	bool overflow = false;
	K7bCkpt* ck = q.ckpt + (size_t)c * (K7B_CH / 32) * q.n_chan_pad + chan_raw;
	const int w0 = start >> 5, nwd = (n1 - start + 31) >> 5;
#pragma unroll 8
	for (int i = 0; i < nwd; i++) rowbits[64 * i + lane] = w0 + i >= 0 ? brow[w0 + i] : prow[w0 + i];
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); // (one-wave workgroup: every lane reads its own column, ordering only)
	// The lanes of a wave are independent sample streams.  They walk the chunk in step word by word (32 samples: the word fetch and the
	// checkpoint store are then uniform, once per word -- with every lane at a sample of its own some lane crossed a word boundary in
	// nearly every iteration, and the whole wave waited for its LDS read), and inside a word every lane advances to ITS next emission
	// (the sampler emits every fifth sample, give or take) before all lanes run the decoder step together: once per symbol instead of
	// once per sample of whichever lane happens to emit (the step is ~150 instructions of selects).
	uint32_t ahead = rowbits[lane];
	for (int nb = start, wi = 0; nb < n1; nb += 32, wi++) {
		const uint32_t word = ahead;
		ahead = rowbits[64 * (wi + 1 < nwd ? wi + 1 : wi) + lane];
		if (nb >= n0) ck[(size_t)((nb - n0) >> 5) * q.n_chan_pad] = base_ckpt(b); // (channel rows are padded: idle lanes write their own slot)
		int i = 0;
		while (__any(i < 32)) {
			bool emit = false;
			int bit = 0;
			while (i < 32 && !emit) {
				bit = (int)((word >> i) & 1u);
				emit = base_pll_step(b, bit);
				i++;
			}
			if (emit && dec_step(b.r, bit, 0.0f, 0ll, data)) { // (tag.sample_lvl / sample_idx are never set in this engine)
				if (nb >= n0 && live) base_record(b, nb + i - 1, list, n_rec, (uint32_t)q.fcap, data, overflow); // (what the warm-up "completes" is not a frame)
				b.r.state = DST_TRAINING; b.r.position = 0; b.r.osc = 0;
			}
		}
	}
	if (live) {
		base_store(b, q.end + slot, data);
		q.sum_spec[(size_t)chan * K7B_MAXC + c] = (uint8_t)n_rec;
		if (overflow) q.fallback[chan_raw] = 1;
	}
}

// Only K7B_TL lanes of a task wave carry a channel: the tasks of a wave are unrelated loops (a frame here, a closing flag there), every
// one of them costs the whole wave its instructions, and a 64-channel wave had three to eight of them back to back.  With few channels
// per wave most waves find no task and leave at once, and the ones that stay run little more than their own.
// A task is one long dependent chain in a wave that has nothing else to do, so nothing in the loop waits for memory: the bit row
// travels two words ahead in registers, the checkpoint to compare with one ahead.  (Windows of row and checkpoints staged in LDS by
// the whole wave were measured as well: no faster -- what is left is the chain of dependent instructions itself, ~6 clocks each.)
constexpr int K7B_TL = 8;
__global__ __launch_bounds__(64) void k7b_task(K7bParams q) {
	__builtin_amdgcn_s_setprio(3);
	__shared__ uint32_t fdata[DEC_DATA_WORDS * 64];
	__shared__ uint16_t s_crc[256];
	__shared__ uint32_t s_sym[64];
	const K7Params& p = q.k;
	const int lane = threadIdx.x, c = blockIdx.y;
	const int chan0 = blockIdx.x * K7B_TL;
	const bool live = lane < K7B_TL && chan0 + lane < p.n_chan;
	const int chan = live ? chan0 + lane : 0; // (idle lanes read channel 0's rows and write nothing)
	uint32_t* data = fdata + lane;
	const size_t slot = (size_t)c * q.n_chan_pad + chan;
	BaseReg b;
	base_load(b, c ? q.end + (size_t)(c - 1) * q.n_chan_pad + chan : p.state + chan, data); // boundary 0: the state the previous block left
	const auto ckpt_of = [&](int ch, int w) { // the recorded state in front of sample 32 w of channel ch
		const int n = w << 5;
		return q.ckpt[((size_t)(n / K7B_CH) * (K7B_CH / 32) + (size_t)((n % K7B_CH) >> 5)) * q.n_chan_pad + ch];
	};
	const int n0 = c * K7B_CH;
	// what a task needs first -- two words of the bit row, the next checkpoint -- is asked for together with the state (one round trip)
	const int nw_row = (p.L + 31) >> 5;
	const uint32_t* brow = p.fm_cur + (size_t)chan * p.fm_stride;
	const auto row_word = [&](int w) { return brow[w < nw_row ? w : nw_row - 1]; };
	uint32_t w1 = row_word(n0 >> 5), w2 = row_word((n0 >> 5) + 1);
	K7bCkpt next = ckpt_of(chan, (n0 >> 5) + 1 < nw_row ? (n0 >> 5) + 1 : n0 >> 5); // the recorded state one checkpoint ahead
	// (Tasks whose start state is not the true one -- the end of a trajectory that never became true -- are wasted work, but they run
	// beside the useful ones; restricting the launch to boundaries whose predecessor is known to be true, in passes, was measured: the
	// longest USEFUL task sets the time either way.)
	const bool run = live && !base_same(base_ckpt(b), ckpt_of(chan, n0 >> 5));
	int merge = run ? p.L : -1;
	uint32_t* list = q.task_frames + slot * (1 + K7B_FCAP * K7B_FREC);
	uint32_t n_rec = 0;
	if (live && !run) { q.task_merge[slot] = -1; q.sum_task[(size_t)chan * K7B_MAXC + c] = 0u; }
	if (!__any(run)) return;
	for (int i = lane; i < 256; i += 64) dec_crc_table_entry(i, s_crc);
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); // (one-wave workgroup: ordering only)
	bool overflow = false;
	// The task walks its channel word by word (32 samples).  In front of every word: the word itself (travelling two ahead), the
	// comparison with the recorded state (the merge), and the choice of how to walk the word --
	//  A  the decoder is inside a frame (DATAFCS): the sampler's gain is slow whatever the decoder does until the frame ends, so the
	//     sampler runs alone over up to four words (branch-free, base_gain_word), and the word-parallel frame evaluator (dec_run_frame,
	//     the one k7e_sim uses; fuzzed against dec_step on the host) swallows the <= 30 decisions at once.  Only a stretch in which the
	//     frame ENDS (closing flag, abort) is not accepted: the lane goes back to the state in front of it, lets the evaluator take
	//     the decisions before the end (sampler sample by sample, it has to stop at that very emission) and walks the rest in C, so
	//     that the gain switches at exactly the sample the reference switches it.
	//  F  the decoder is in TRAINING: the gain is fast as long as it stays there.  The sampler runs alone over the word, the decoder's
	//     TRAINING step (count alternations: five instructions) over its decisions; if one of them starts a flag, the word is walked
	//     again in C.
	//  C  symbol by symbol: the samples up to the next emission (or the end of the word), then the decoder step.
	// No checkpoint can match inside a frame (both decoders must be in TRAINING).
	int n = n0, bn = -1; // bn: the word boundary whose work (fetch, comparison) is done
	bool active = run, finish = false, cword = false;
	uint32_t word = 0u; // the word of sample n (w1 / w2: the two behind it; in front of a boundary that has not been worked: the words AT n)
	int next_w = (n0 >> 5) + 1;
	bool stale = false; // r.crc / r.tail behind (dec_run_frame<LAZY>)
	const int n_end = p.L;
	while (__any(active && n < n_end)) {
		if (active && n < n_end && (n & 31) == 0 && n != bn) { // ---- a word boundary
			bn = n;
			word = w1; w1 = w2; w2 = row_word((n >> 5) + 2);
			if (n > n0) {
				const int w = n >> 5;
				const K7bCkpt cur = next_w == w ? next : ckpt_of(chan, w);
				next_w = w + 1;
				next = ckpt_of(chan, next_w < nw_row ? next_w : w);
				if (base_same(base_ckpt(b), cur)) { merge = n; active = false; }
			}
		}
		const bool go = active && n < n_end;
		const bool whole = go && (n & 31) == 0 && n + 32 <= n_end;
		if (whole && b.r.state == DST_DATAFCS && !finish) { // ---- A
			const BaseReg snap = b;
			const int n_s = n;
			const uint32_t w_s = word, w1_s = w1, w2_s = w2;
			uint32_t bits = 0;
			int ns = 0;
			base_gain_word<false>(b.pll, b.pprev, word, bits, ns);
			n += 32;
			for (int k = 1; k < 4 && n + 32 <= n_end; k++) { // (at most 30 decisions in four words: the phase advances by 0.225 per sample at most)
				word = w1; w1 = w2; w2 = row_word((n >> 5) + 2);
				base_gain_word<false>(b.pll, b.pprev, word, bits, ns);
				n += 32;
			}
			s_sym[lane] = bits;
			int end = 0;
			const int flags = dec_run_frame<true>(b.r, data, s_sym + lane, nullptr, 0, ns, s_crc, end);
			stale = true;
			if (flags != 2) { // the frame ends at decision `end` of this stretch
				b = snap; n = n_s; word = w_s; w1 = w1_s; w2 = w2_s; finish = true;
				bits = 0;
				for (int left = end; left > 0;) { // the decisions in front of it (the frame goes on behind every one of them: slow gain)
					if ((n & 31) == 0 && n != bn) { bn = n; word = w1; w1 = w2; w2 = row_word((n >> 5) + 2); next_w = -1; }
					const int bit = (int)((word >> (n & 31)) & 1u);
					const bool emit = base_pll_step(b, bit);
					n++;
					if (emit) { bits |= (uint32_t)bit << (end - left); left--; }
				}
				s_sym[lane] = bits;
				int e2 = 0;
				if (end == 0) dec_fix_crc_tail(b.r, data, s_crc);
				else if (dec_run_frame<false>(b.r, data, s_sym + lane, nullptr, 0, end, s_crc, e2) != 2) overflow = true; // (cannot happen: the same decisions; k7_base would decode the block)
				stale = false;
			}
		} else if (whole && b.r.state == DST_TRAINING && !cword) { // ---- F
			const float pll_s = b.pll;
			const int pprev_s = b.pprev, last_s = b.r.lastBit, prev_s = b.r.prev, pos_s = b.r.position, osc_s = b.r.osc;
			uint32_t bits = 0;
			int ns = 0;
			base_gain_word<true>(b.pll, b.pprev, word, bits, ns);
			bool ok = true;
			for (int k = 0; k < ns; k++) { // AIS::Decoder in TRAINING (Marine/AIS.h:109-119; dec_step's isT half)
				const int dd = (int)((bits >> k) & 1u);
				const int Bit = dd == b.r.prev;
				b.r.prev = dd;
				const bool alt = Bit != b.r.lastBit;
				ok = ok && (alt || b.r.position <= 4); // (else: two equal bits after more than four alternations start a flag)
				b.r.position = alt ? b.r.position + 1 : 0;
				b.r.osc = alt ? b.r.osc : 0;
				b.r.lastBit = Bit;
			}
			if (ok) n += 32;
			else { b.pll = pll_s; b.pprev = pprev_s; b.r.lastBit = last_s; b.r.prev = prev_s; b.r.position = pos_s; b.r.osc = osc_s; cword = true; }
		} else if (go) { // ---- C
			bool emit = false;
			int bit = 0;
			do {
				bit = (int)((word >> (n & 31)) & 1u);
				emit = base_pll_step(b, bit);
				n++;
			} while (!emit && (n & 31) != 0 && n < n_end);
			if (emit && dec_step(b.r, bit, 0.0f, 0ll, data)) { // (tag.sample_lvl / sample_idx are never set in this engine)
				base_record(b, n - 1, list, n_rec, (uint32_t)q.fcap, data, overflow);
				b.r.state = DST_TRAINING; b.r.position = 0; b.r.osc = 0;
			}
			if (b.r.state != DST_DATAFCS) finish = false;
			if ((n & 31) == 0) cword = false;
		}
	}
	if (run) {
		q.task_merge[slot] = merge;
		q.sum_task[(size_t)chan * K7B_MAXC + c] = (uint32_t)(merge + 1) | n_rec << 16;
		if (merge == p.L) { // ran to the end of the block: this is the channel's state
			if (stale && b.r.state == DST_DATAFCS) dec_fix_crc_tail(b.r, data, s_crc); // (the block ends inside a frame)
			base_store(b, q.task_end + slot, data);
		}
		if (overflow) q.fallback[chan] = 1;
	}
}

// k7b_walk / k7b_emit.  The decision which trajectory is the channel's is a chain over the boundaries (k7b_walk: one lane per
// channel); copying the frames out is not -- a lane that did it alone went from list to list with two dependent memory round trips
// each (0.28 ms with 256 distinct receivers) -- so the walk only notes per list "yours from sample s on, at offset o of your
// frames", and k7b_emit copies the lists out, one lane per list.  Two kernels of one-wave workgroups, not one of
// eight-wave workgroups with a barrier: beside the front end and the next block's speculative pass a CU rarely has eight wave slots
// with the registers free at once, and the workgroup waited for most of the front end's launch (0.2 ms against 0.04 alone).
// The kernel runs beside the next block's front end, which keeps the memory system full: a round trip takes microseconds, and what
// the kernel costs is the NUMBER of dependent round trips.  So the pass and the tasks leave a summary per (channel, boundary) --
// merge position, list lengths -- in rows per CHANNEL, a lane fetches 48 boundaries of its channel with 15 wide loads in flight at
// once, and the walk is a fully unrolled scan over registers.  (No LDS tile either: the front end's workgroups hold all of a CU's
// LDS, and a workgroup that asked for 30 KB waited for most of the front end's launch.)
constexpr int K7B_SCAN = 48; // boundaries per batch of the scan (K7B_MAXC = 2 batches)
constexpr uint32_t K7B_NOT = 0xFFFFFFFFu;
__global__ __launch_bounds__(64) void k7b_walk(K7bParams q) {
	__builtin_amdgcn_s_setprio(3);
	const K7Params& p = q.k;
	const int lane = threadIdx.x;
	const int chan_raw = blockIdx.x * 64 + lane;
	const bool live = chan_raw < p.n_chan;
	const int chan = live ? chan_raw : 0;
	const size_t per = 1 + K7B_FCAP * K7B_FREC;
	{
		unsigned mine = 0;
		const DecState* fin = nullptr;
		uint4 tw[K7B_SCAN / 4], sw[K7B_SCAN / 16];
		const auto fetch = [&](int c0) {
			const uint4* trow = reinterpret_cast<const uint4*>(q.sum_task + (size_t)chan * K7B_MAXC + c0);
			const uint4* srow = reinterpret_cast<const uint4*>(q.sum_spec + (size_t)chan * K7B_MAXC + c0);
#pragma unroll
			for (int e = 0; e < K7B_SCAN / 4; e++) tw[e] = trow[e];
#pragma unroll
			for (int e = 0; e < K7B_SCAN / 16; e++) sw[e] = srow[e];
		};
		fetch(0); // (in flight together with the flag below: one round trip, not two)
		const bool go = live && __builtin_nontemporal_load(q.fallback + chan) == 0; // (flagged by a frame list that overflowed: k7_base decodes the block from the untouched state)
		constexpr int NEVER = 0x7FFFFFFF;
		int next_c = go ? 0 : NEVER; // the next boundary to decide (the ones before it lie inside a task)
		int pend_c = -1, pend_from = 0; // the chunk whose recorded trajectory the last task joined, and where
		if (go) fin = q.end + (size_t)(q.n_chunks - 1) * q.n_chan_pad + chan;
		static_assert(K7B_MAXC % K7B_SCAN == 0 && K7B_SCAN % 16 == 0, "k7b_walk: whole 16-byte loads per batch");
		for (int c0 = 0; c0 < q.n_chunks; c0 += K7B_SCAN) {
			if (c0) fetch(c0);
#pragma unroll
			for (int e = 0; e < K7B_SCAN; e++) {
				const int c = c0 + e;
				if (c < q.n_chunks) {
					const uint32_t t4 = e % 4 == 0 ? tw[e / 4].x : e % 4 == 1 ? tw[e / 4].y : e % 4 == 2 ? tw[e / 4].z : tw[e / 4].w;
					const uint32_t s4 = (e / 4) % 4 == 0 ? sw[e / 16].x : (e / 4) % 4 == 1 ? sw[e / 16].y : (e / 4) % 4 == 2 ? sw[e / 16].z : sw[e / 16].w;
					const int m = (int)(t4 & 0xFFFFu) - 1;             // -1: the speculative state was the true one
					const uint32_t t = t4 >> 16, a = (s4 >> (8 * (e % 4))) & 255u; // frames in the task's / the chunk's list
					const size_t sl = (size_t)c * q.n_chan_pad + chan_raw;
					uint32_t rs = K7B_NOT, rt = K7B_NOT;
					if (c == next_c) {
						if (m < 0) { rs = a << 12 | mine; mine += a; next_c = c + 1; }
						else {
							rt = t << 12 | mine; mine += t; // the exact loop from the previous chunk's (true) end state up to the merge
							if (m >= p.L) { fin = q.task_end + (size_t)c * q.n_chan_pad + chan; next_c = NEVER; }
							else { pend_c = m / K7B_CH; pend_from = m; next_c = pend_c + 1; }
						}
					}
					if (c == pend_c) { // the trajectory the task joined (possibly inside this very chunk): only what it completed from there on
						unsigned k = 0;
						const uint32_t* list = q.frames + ((size_t)c * q.n_chan_pad + chan) * per;
						for (unsigned i = 0; i < a; i++) k += (int)list[1 + i * K7B_FREC] >= pend_from ? 1u : 0u; // (rare: a dependent read)
						rs = (uint32_t)pend_from << 16 | a << 12 | mine;
						mine += k;
						pend_c = -1;
					}
					q.take_spec[sl] = rs; q.take_task[sl] = rt;
				}
			}
		}
		unsigned incl = mine;
		for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d); if (lane >= d) incl += o; }
		unsigned base = 0;
		if (lane == 63 && incl) base = atomicAdd(p.frame_count, incl);
		q.out_base[chan_raw] = __shfl(base, 63) + incl - mine;
		// where the channel's state after this block is to be found (k7b_emit copies it): 0 = nowhere (k7_base will decode the block),
		// 1 = the last chunk's trajectory, 2 + c = the task of boundary c
		q.fin_sel[chan_raw] = !fin ? 0u : fin == q.end + (size_t)(q.n_chunks - 1) * q.n_chan_pad + chan ? 1u : 2u + (uint32_t)(fin - (q.task_end + chan)) / (uint32_t)q.n_chan_pad;
	}
}

// one lane per (chunk, channel): the chunk's list and its boundary's task list.  Beside the front end a memory round trip is ~10 us
// (tens of MB in flight), so the first record of either list is fetched together with the notes, before it is known whether the
// list is the channel's: one round trip in the common case.
__global__ __launch_bounds__(64) void k7b_emit(K7bParams q) {
	__builtin_amdgcn_s_setprio(3);
	const K7Params& p = q.k;
	const int lane = threadIdx.x, c = blockIdx.y;
	const int chan_raw = blockIdx.x * 64 + lane;
	const bool live = chan_raw < p.n_chan;
	const int chan = live ? chan_raw : 0;
	const size_t per = 1 + K7B_FCAP * K7B_FREC;
	const size_t sl = (size_t)c * q.n_chan_pad + chan_raw;
	const uint32_t* lists[2] = { q.frames + ((size_t)c * q.n_chan_pad + chan) * per, q.task_frames + ((size_t)c * q.n_chan_pad + chan) * per };
	const unsigned my_base = q.out_base[chan_raw];
	const uint32_t notes[2] = { q.take_spec[sl], q.take_task[sl] };
	const uint32_t sel = c == q.n_chunks - 1 ? q.fin_sel[chan_raw] : 0u;
	uint32_t first[2][K7B_FREC];
#pragma unroll
	for (int l = 0; l < 2; l++)
#pragma unroll
		for (int w = 0; w < K7B_FREC; w++) first[l][w] = lists[l][1 + w];
	// note: first sample << 16 | frames in the list << 12 | offset among the channel's frames (< 4096: K7B_MAXC * 2 * K7B_FCAP)
#pragma unroll
	for (int l = 0; l < 2; l++) {
		const uint32_t note = notes[l];
		if (note == K7B_NOT || !live) continue;
		const unsigned cnt = (note >> 12) & 15u;
		const int from = l == 0 ? (int)(note >> 16) : 0;
		unsigned at = my_base + (note & 0xFFFu);
		for (unsigned i = 0; i < cnt; i++) {
			uint32_t r[K7B_FREC];
			if (i == 0) { for (int w = 0; w < K7B_FREC; w++) r[w] = first[l][w]; }
			else { for (int w = 0; w < K7B_FREC; w++) r[w] = lists[l][1 + i * K7B_FREC + w]; }
			if ((int)r[0] < from) continue;
			uint32_t* f = p.frames + (size_t)(at++ % (unsigned)p.max_frames) * DEC_FRAME_WORDS;
			f[0] = (uint32_t)chan; f[1] = r[0]; f[2] = r[1]; f[3] = 0; // (level sum: tag.sample_lvl is never set in this engine)
			f[4] = 0; f[5] = 0; f[6] = 0; f[7] = 0;
			f[8] = p.block; f[9] = p.sub;
			for (int w = 0; w < DEC_DATA_WORDS; w++) f[10 + w] = r[2 + w];
		}
	}
	if (sel != 0u && live) p.state[chan] = sel == 1u ? q.end[(size_t)(q.n_chunks - 1) * q.n_chan_pad + chan] : q.task_end[(size_t)(sel - 2u) * q.n_chan_pad + chan];
}

// ------------------------------------------------------------------------------------------
// K7e: event-driven frame decoders (see kernels.h).  Conventions: symbol g of decoder (chan, j) is bit g of its packed row;
// Bit[g] = (dd[g] == dd[g-1]) is the NRZI bit, alt[g] = (Bit[g] != Bit[g-1]).  In TRAINING the reference counts consecutive
// alternations in `position` and leaves for STARTFLAG at the first symbol with alt == 0 and position > 4 (Marine/AIS.h:109-119).
// ------------------------------------------------------------------------------------------
constexpr uint32_t K7E_CONT = 0xFFFFu;
#ifndef K7E_RW_
#define K7E_RW_ 32
#endif
constexpr int K7E_RW = K7E_RW_; // events a lane of k7e_resolve stages in LDS at a time (a multiple of 16; -DK7E_RW_=16 exercises the re-staging)
#ifndef K7E_SIM_LANES_
#define K7E_SIM_LANES_ 16
#endif
constexpr int K7E_SIM_LANES = K7E_SIM_LANES_; // lanes of k7e_sim per decoder
constexpr int K7E_FOUND = 8;  // completed messages a lane of k7e_resolve notes before it copies them out

__device__ __forceinline__ int k7e_training_pos(const DecState* st) { // alternations counted so far, only "> 4" ever matters
	return st->state == DST_TRAINING ? (st->position < 5 ? st->position : 5) : 0;
}

// The decoders a K7e launch serves: five per channel on the coherent rows (ModelDefault; ModelStandard with the FM rows handed in as
// `bits`), or -- kind 2, ModelChallenger -- ten per channel in the reference's order within a group (o = 0..3: FM0..FM3, 4..8: the
// coherent decoders 0..4, 9: FM4; Model.cpp:630-639), decoder d = chan * 10 + o, the FM ones on the regrouped discriminator rows.
struct K7eDec { const uint32_t* row; int chan, j, o; bool lvl_shift; };
__device__ __forceinline__ int k7e_mesh(const K7Params& p) { return p.kind == 2 ? 10 : 5; }
__device__ __forceinline__ K7eDec k7e_decoder(const K7Params& p, int d) {
	K7eDec r;
	if (p.kind == 2) {
		r.chan = d / 10; r.o = d - 10 * r.chan;
		const bool is_fm = r.o < 4 || r.o == 9;
		r.j = r.o < 4 ? r.o : r.o == 9 ? 4 : r.o - 4;
		r.row = is_fm ? p.fmrows + (size_t)(r.chan * 5 + r.j) * p.fmrows_stride : p.bits + (size_t)(r.chan * 5 + r.j) * p.bits_stride;
		r.lvl_shift = r.o < 4; // FM0..FM3 run before the group's ScatterPLL output: they see the previous group's level
	} else {
		r.chan = d / 5; r.j = d - 5 * r.chan; r.o = r.j;
		r.row = p.bits + (size_t)d * p.bits_stride;
		r.lvl_shift = false;
	}
	return r;
}

// Sixteen lanes per decoder, each a segment of the row's words: every candidate of the block, classified (dec_core.h).  One lane
// per decoder walked 154 words one after the other -- 0.12 ms alone and 0.17 - 0.22 ms next to the front end, a third of the
// decoder pass; a segment is ten words, whose loads are all in flight at once.  The lanes of a decoder then agree on what a
// segment cannot know alone -- whether its last failed candidate blocks the first candidate behind the segment, and where its
// events and runs go in the decoder's lists -- with a suffix minimum and two prefix sums over the 16-lane row.
constexpr int K7E_SCAN_CAP = DEC_SCAN_MAXW * 32 / 6 + 3; // events of one segment (candidates are >= 6 symbols apart)
__global__ __launch_bounds__(64) void k7e_scan(K7eParams q) {
	__shared__ uint32_t s_list[K7E_SCAN_CAP][64];
	const K7Params& p = q.k;
	__builtin_amdgcn_s_setprio(3); // latency-bound waves: they need their few issue slots at once
	const int lane = threadIdx.x, seg = lane & 15;
	if (blockIdx.x == 0 && lane == 0 && q.overflow_clear) *q.overflow_clear = 0;
	const int n_dec = p.n_chan * k7e_mesh(p);
	const int d_raw = blockIdx.x * 4 + (lane >> 4);
	const bool live = d_raw < n_dec;
	const int d = live ? d_raw : 0;
	const DecState* st = p.state + d;
	const uint32_t* brow = k7e_decoder(p, d).row;
	uint32_t* ev = q.ev + (size_t)d * K7E_EVCAP;
	uint16_t* oc = q.open_c + (size_t)d * K7E_OPENCAP;
	const int n = p.n_groups, nw = (n + 31) >> 5, wps = (nw + 15) >> 4;
	const int w_begin = seg * wps;
	const int cnt = w_begin >= nw ? 0 : (nw - w_begin < wps ? nw - w_begin : wps);
	uint32_t W[DEC_SCAN_MAXW + 1];
#pragma unroll
	for (int k = 0; k <= DEC_SCAN_MAXW; k++) W[k] = k <= cnt && w_begin + k < nw ? brow[w_begin + k] : 0u;
	const int state0 = st->state;
	uint32_t prevD, prevB, prevA;
	if (w_begin == 0) { // carry-in: dd[-1], Bit[-1] and the alternations counted so far (they are the last `position` symbols)
		prevD = st->prev ? 0x80000000u : 0u; prevB = st->lastBit ? 0x80000000u : 0u;
		const int p5 = k7e_training_pos(st);
		prevA = p5 ? (0xFFFFFFFFu << (32 - p5)) : 0u;
	} else dec_scan_carry(cnt > 0 ? brow[w_begin - 1] : 0u, prevD, prevB, prevA);
	ScanSeg sg;
	dec_scan_words(W, cnt, w_begin, n, prevD, prevB, prevA, sg, [&](uint32_t e) { s_list[sg.nev][lane] = e; });
	// ---- the lanes of the decoder
	int m = sg.first_c; // first candidate at or behind this segment
#pragma unroll
	for (int o = 1; o < 16; o <<= 1) { const int t = __shfl_down(m, o, 16); if (seg + o < 16) m = t < m ? t : m; }
	int nf = __shfl_down(m, 1, 16); // ... behind it
	if (seg == 15) nf = DEC_SCAN_INF;
	const bool trailing = sg.pend_until >= 0 && (nf != DEC_SCAN_INF ? nf < sg.pend_until : sg.pend_until > n);
	const int mine_ev = sg.nev + (trailing ? 1 : 0);
	int sum_ev = mine_ev, sum_run = sg.nrun; // inclusive prefix sums over the row
#pragma unroll
	for (int o = 1; o < 16; o <<= 1) {
		const int te = __shfl_up(sum_ev, o, 16), tr = __shfl_up(sum_run, o, 16);
		if (seg >= o) { sum_ev += te; sum_run += tr; }
	}
	const int base = state0 != DST_TRAINING ? 1 : 0; // a frame (or a start flag) is in flight: it continues at symbol 0
	const int ev_off = base + sum_ev - mine_ev, run_off = base + sum_run - sg.nrun;
	if (!live) return;
	int over = 0;
	if (seg == 0 && base) { ev[0] = 0u | (K7E_RUN << 13) | (0u << 19); oc[0] = (uint16_t)K7E_CONT; }
	for (int i = 0; i < sg.nev; i++) {
		uint32_t e = s_list[i][lane];
		if (((e >> 13) & 3u) == K7E_RUN) {
			const int slot = (int)(e >> 19) + run_off;
			if (slot >= K7E_OPENCAP) { over |= 1; e = (e & 0x1FFFu) | (K7E_FAIL << 13) | (1u << 15); }
			else { oc[slot] = (uint16_t)(e & 0x1FFFu); e = (e & 0x7FFFFu) | ((uint32_t)slot << 19); }
		}
		if (ev_off + i < K7E_EVCAP) ev[ev_off + i] = e; else over |= 2;
	}
	if (trailing) { if (ev_off + sg.nev < K7E_EVCAP) ev[ev_off + sg.nev] = sg.pend; else over |= 2; }
	if (over) atomicOr(q.overflow, over);
	if (seg == 15) {
		const int tot_ev = base + sum_ev, tot_run = base + sum_run;
		q.cnt[d] = (uint32_t)(tot_ev < K7E_EVCAP ? tot_ev : K7E_EVCAP) | ((uint32_t)(tot_run < K7E_OPENCAP ? tot_run : K7E_OPENCAP) << 16);
	}
}

// one lane per (decoder, run): the reference's state machine from the candidate (or from the carried state) until it is back in
// TRAINING, has completed a message, or the block ends.  Up to the frame's first symbol (the rest of the start flag: nine
// symbols at most) that is the step itself; inside the frame it is dec_run_frame (dec_core.h), which works on words of 32
// symbols -- a frame of 250 symbols is eight rounds of ~100 instructions instead of 250 steps of ~180.
__global__ __launch_bounds__(64) void k7e_sim(K7eParams q) {
	__shared__ uint32_t fdata[DEC_DATA_WORDS * 64]; // [word][lane]
	__shared__ uint16_t s_crc[256];
	const K7Params& p = q.k;
	if (*q.overflow != 0) return; // (uniform: the sequential kernel decodes this block)
	__builtin_amdgcn_s_setprio(3);
	const int lane = threadIdx.x;
	for (int i = lane; i < 256; i += 64) dec_crc_table_entry(i, s_crc);
	__syncthreads();
	// K7E_SIM_LANES lanes per decoder take its runs round robin.  A decoder has ~4.5 runs per block on the bench signal (13 at
	// most) and there are fewer waves than SIMDs: with 16 lanes a second round practically never happens, and idle lanes cost nothing.
	const int d = blockIdx.x * (64 / K7E_SIM_LANES) + lane / K7E_SIM_LANES;
	const int n_dec = p.n_chan * k7e_mesh(p);
	const int dd_ = d < n_dec ? d : 0;
	const int nrun = d < n_dec ? (int)(q.cnt[dd_] >> 16) : 0;
	const K7eDec dc = k7e_decoder(p, dd_);
	const int chan = dc.chan, j = dc.j;
	const DecState* st = p.state + dd_;
	const uint32_t* brow = dc.row;
	const float* lrow = p.lvl ? p.lvl + (size_t)chan * p.lvl_stride : nullptr; // (ModelStandard: no ScatterPLL, no level)
	// ModelChallenger's FM0..FM3: the level of symbol 0 is what tag.sample_lvl held when the block began -- this channel's last
	// group of the block before, or, for samples of this block, what the OTHER channel left (one TAG per device, Rotate hands
	// channel A its whole block before channel B: k7_decode_mesh<2>)
	float lvl_first = 0.0f;
	if (dc.lvl_shift) {
		lvl_first = p.last_lvl_in[chan];
		if (p.n_rel0 + j >= 0)
			lvl_first = (chan & 1) ? (p.n_groups > 0 ? p.lvl[(size_t)(chan ^ 1) * p.lvl_stride + p.n_groups - 1] : p.last_lvl_in[chan ^ 1]) : p.last_lvl_in[chan ^ 1];
	}
	uint32_t* data = fdata + lane;
	const int n = p.n_groups, nw = (n + 31) >> 5;
	const auto dd_at = [&](int g) -> int { return g < 0 ? st->prev : (int)((brow[g >> 5] >> (g & 31)) & 1u); };
	for (int k = lane % K7E_SIM_LANES; __any(k < nrun); k += K7E_SIM_LANES) {
		const bool act = k < nrun;
		const uint32_t c0 = act ? q.open_c[(size_t)dd_ * K7E_OPENCAP + k] : 0u;
		const bool cont = c0 == K7E_CONT;
		const int c = cont ? 0 : (int)c0;
		DecReg r;
		if (cont) {
			r.state = st->state; r.lastBit = st->lastBit; r.prev = st->prev; r.position = st->position; r.osc = st->osc;
			r.level = st->level; r.start_idx = st->start_idx;
			for (int w = 0; w < DEC_DATA_WORDS; w++) data[64 * w] = st->data[w];
			r.crc = st->crc[0]; r.cw = st->crc[1]; r.cwi = (int)st->crc[2]; r.tail = st->crc[3]; r.abort_pos = (int)st->crc[4];
		} else {
			r.state = DST_TRAINING; r.position = 5; r.osc = 0; r.level = 0.0f; r.start_idx = 0;
			r.prev = dd_at(c - 1);
			r.lastBit = c - 1 < 0 ? st->lastBit : (dd_at(c - 1) == (c - 2 < 0 ? st->prev : dd_at(c - 2)));
			r.crc = 0xFFFFu; r.cw = 0u; r.cwi = 0; r.tail = 0u; r.abort_pos = 0;
		}
		bool running = act;
		int g = c, end = n, flags = 2; // (2: still running when the block ends)
		// ---- the rest of the start flag, symbol by symbol (its decisions are in two adjacent words)
		const int i0 = c >> 5;
		const uint32_t w0 = brow[i0 < nw ? i0 : nw - 1], w1 = brow[i0 + 1 < nw ? i0 + 1 : nw - 1];
		while (__any(running && r.state != DST_DATAFCS)) {
			if (running && r.state != DST_DATAFCS) {
				if (g >= n) running = false;
				else {
					const uint32_t word = (g >> 5) == i0 ? w0 : w1;
					dec_step<false>(r, (int)((word >> (g & 31)) & 1u), 0.0f, 5 * (p.first_group + g) + j, data);
					if (r.state == DST_TRAINING) { end = g; flags = 0; running = false; }
					g++;
				}
			}
		}
		// ---- the frame
		if (running) flags = dec_run_frame(r, data, brow, lrow, g, n, s_crc, end, dc.lvl_shift ? 1 : 0, lvl_first);
		if (act) {
			K7Slot* sl = q.slot + (size_t)d * K7E_OPENCAP + k;
			sl->end = end; sl->flags = flags;
			if (flags != 0) { // a message (position / level / start_idx / data are the frame's) or the state to carry on with
				DecState* o = &sl->s;
				o->state = r.state; o->lastBit = r.lastBit; o->prev = r.prev; o->position = r.position; o->osc = r.osc;
				o->level = r.level; o->start_idx = r.start_idx;
				if (flags == 2) data[64 * r.cwi] = r.cw;
				const int used = flags == 1 ? (r.position + 31) >> 5 : r.cwi + 1; // (words behind them are never read)
				for (int w = 0; w < DEC_DATA_WORDS; w++) o->data[w] = w < used ? data[64 * w] : 0u;
				o->crc[0] = r.crc; o->crc[1] = r.cw; o->crc[2] = (uint32_t)r.cwi; o->crc[3] = r.tail; o->crc[4] = (uint32_t)r.abort_pos;
			}
		}
	}
}

// One lane per decoder, the five lanes of a channel (of the eight it owns in the wave) walk together: which of the runs really happened, in the order in which the
// reference runs its five decoders (symbol by symbol, phase 0..4 within a group), with the Reset a decoder sends its siblings
// when it completes a message.  Per round every lane offers the time of its next event (the start of its next candidate, or the
// end of the run it is in), the channel's earliest one is processed by its owner, and a completed message is broadcast.
// minimum / maximum over the eight lanes of a channel: two quad permutes and a half-row mirror (DPP, no LDS round trip -- the walk
// is a chain of dependent rounds, and next to the front end every ds_bpermute of a round waited in the LDS queue)
__device__ __forceinline__ int grp8_min(int v) {
	int t = __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); v = t < v ? t : v;  // quad_perm [1,0,3,2]
	t = __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true); v = t < v ? t : v;      // quad_perm [2,3,0,1]
	t = __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true); return t < v ? t : v;  // row_half_mirror
}
// (sixteen lanes: one more step, the mirror of the whole row)
template <int GROUP>
__device__ __forceinline__ int grp_min(int v) {
	v = grp8_min(v);
	if (GROUP == 16) { const int t = __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true); v = t < v ? t : v; } // row_mirror
	return v;
}
template <int GROUP>
__device__ __forceinline__ int grp_max(int v) { return -grp_min<GROUP>(-v); }

// MESH decoders per channel in GROUP lanes (5 in 8: ModelDefault / ModelStandard; 10 in 16: ModelChallenger, lane o = the decoder's
// place in the reference's order within a group)
template <int MESH, int GROUP>
__global__ __launch_bounds__(64) void k7e_resolve(K7eParams q) {
	const K7Params& p = q.k;
	__builtin_amdgcn_s_setprio(3); // a handful of latency-bound waves: they need their few issue slots at once
	const int lane = threadIdx.x;
	const int mesh = lane / GROUP, j = lane % GROUP; // j: the decoder's place in the order of a group (its phase, for the meshes of five)
	const int chan_raw = blockIdx.x * (64 / GROUP) + mesh;
	const bool live = j < MESH && chan_raw < p.n_chan;
	const int d = (live ? chan_raw : 0) * MESH + (live ? j : 0);
	const K7eDec dc = k7e_decoder(p, d);
	const int n = p.n_groups;
	constexpr int INF = 1 << 26;
	DecState* st = p.state + d;
	const uint32_t* evd = q.ev + (size_t)d * K7E_EVCAP;
	const K7Slot* slots = q.slot + (size_t)d * K7E_OPENCAP;
	// The kernel is a few waves that run beside the next block's front end, where a memory round trip is 10-30 us: what it costs is the
	// NUMBER of dependent round trips (0.33 ms of them per block when every piece was fetched where it was needed).  So everything the
	// walk can need first is asked for at once, before any of it is looked at: the overflow flag, the carried state, the lists'
	// lengths, the first K7E_RW events, and the (end, flags) of the first K7E_RW runs -- run k of a decoder sits in slot k, so its
	// head does not wait for the event that names it.
	__shared__ uint32_t s_ev[K7E_RW][64];
	__shared__ int2 s_ef[K7E_RW][64]; // (end, flags) of run sw0 + i
	int win0 = 0, sw0 = 0, nev = 0;
	const auto stage = [&](int from, int run0, bool first) {
#pragma unroll
		for (int h = 0; h < K7E_RW; h += 16) {
			uint32_t e[16];
			int2 f[16];
#pragma unroll
			for (int i = 0; i < 16; i++) e[i] = (first || from + h + i < nev) ? evd[from + h + i < K7E_EVCAP ? from + h + i : K7E_EVCAP - 1] : 0xFFFFFFFFu;
#pragma unroll
			for (int i = 0; i < 16; i++) f[i] = *reinterpret_cast<const int2*>(slots + (run0 + h + i < K7E_OPENCAP ? run0 + h + i : K7E_OPENCAP - 1));
#pragma unroll
			for (int i = 0; i < 16; i++) { s_ev[h + i][lane] = e[i]; s_ef[h + i][lane] = f[i]; }
		}
		win0 = from; sw0 = run0;
	};
	const int ovf = *q.overflow;
	const int state0 = st->state, prev0 = st->prev, last0 = st->lastBit, position0 = st->position;
	const uint32_t cnt0 = q.cnt[d];
	stage(0, 0, true);
	if (ovf != 0) return; // (uniform: nothing has been touched, the sequential kernel decodes this block from the same state)
	const int pos0 = state0 == DST_TRAINING ? (position0 < 5 ? position0 : 5) : 0; // (k7e_training_pos)
	nev = live ? (int)(cnt0 & 0xFFFFu) : 0;
	int ptr = 0, free_at = state0 == DST_TRAINING ? 5 - pos0 : 0, end_ = INF, slot_ = 0, next_run = 0;
	bool busy = false, fnd = false;
	const auto ev_at = [&](int i) { return i < nev ? s_ev[i - win0][lane] : 0xFFFFFFFFu; }; // (the first window was fetched before nev was known)
	uint32_t head = ev_at(0);
	// A completed message is a record of 46 words behind an atomic counter: copied out inside the walk, every one of them would
	// stall its whole wave for two memory round trips (the walk was 0.16 ms, nine tenths of it these).  The walk only notes
	// (symbol, run) and the records leave together at the end: one atomic per lane, all loads in flight at once.  Their order
	// in the ring is irrelevant (the host sorts by receiver, block, channel, group, phase).
	__shared__ uint32_t s_found[K7E_FOUND][64];
	int nfound = 0;
	const auto emit = [&](int from, int to) {
		const int cnt = to - from;
		if (cnt <= 0) return;
		const unsigned fs0 = atomicAdd(p.frame_count, (unsigned)cnt);
		for (int i = from; i < to; i++) {
			const uint32_t v = s_found[i][lane];
			const int e = (int)(v & 0xFFFFu);
			const K7Slot* s = slots + (v >> 16);
			uint32_t* f = p.frames + (size_t)((fs0 + (unsigned)(i - from)) % (unsigned)p.max_frames) * DEC_FRAME_WORDS;
			const long long sidx = 5 * (p.first_group + e) + dc.j;
			f[0] = (uint32_t)d; f[1] = (uint32_t)e; f[2] = (uint32_t)s->s.position; f[3] = __float_as_uint(s->s.level);
			f[4] = (uint32_t)(unsigned long long)s->s.start_idx; f[5] = (uint32_t)((unsigned long long)s->s.start_idx >> 32);
			f[6] = (uint32_t)(unsigned long long)sidx; f[7] = (uint32_t)((unsigned long long)sidx >> 32);
			f[8] = p.block; f[9] = p.sub;
#pragma unroll
			for (int w = 0; w < DEC_DATA_WORDS; w++) f[10 + w] = s->s.data[w];
		}
	};
	for (;;) {
		const int t = busy ? end_ : (head != 0xFFFFFFFFu ? (int)(head & 0x1FFFu) : INF);
		const int key = live && t < n ? t * 16 + j : INF * 16; // (equal groups: the reference's order within a group)
		const int mk = grp_min<GROUP>(key);
		if (!__any(mk < INF * 16)) break; // (runs that are still going at the end of the block stay busy)
		const bool mine = live && key == mk && mk < INF * 16;
		int bcast = -1; // a completed message: (group << 4) | place in the order, told to the siblings
		if (mine) {
			if (!busy) { // a candidate: real only if the decoder has been counting alternations for five symbols
				const uint32_t e = head;
				ptr++;
				const int c = (int)(e & 0x1FFFu), kind = (int)((e >> 13) & 3u), off = (int)((e >> 15) & 15u), sl = (int)(e >> 19);
				if (kind == K7E_RUN) next_run = sl + 1;
				if (c >= free_at) {
					busy = true; slot_ = sl;
					if (kind == K7E_FAIL) { end_ = c + off; fnd = false; }
					else {
						const int2 ef = sl >= sw0 && sl < sw0 + K7E_RW ? s_ef[sl - sw0][lane] : *reinterpret_cast<const int2*>(slots + sl); // (end, flags) of run sl
						fnd = (ef.y & 1) != 0;
						end_ = (ef.y & 2) ? INF : ef.x;
					}
				}
				if (ptr - win0 == K7E_RW && ptr < nev) stage(ptr, next_run, false); // (only this lane's columns are touched; the runs behind the last one seen)
				head = ev_at(ptr);
			} else {
				const int e = end_;
				busy = false;
				free_at = e + 6; // back in TRAINING with position 0 at symbol e
				if (fnd) { // noted now, copied out behind the walk (or right away when the note pad is full)
					if (nfound == K7E_FOUND) { emit(0, nfound); nfound = 0; }
					s_found[nfound++][lane] = (uint32_t)e | ((uint32_t)slot_ << 16);
					bcast = (e << 4) | j;
				}
			}
		}
		// Reset to the siblings (AIS.cpp:47-49): the phases behind the finder are reset BEFORE their step of that group
		const int msg = grp_max<GROUP>(bcast); // (only the owner of the round can have one)
		if (live && !mine && msg >= 0) {
			const int e = msg >> 4, jw = msg & 15;
			const int fa = e + (j > jw ? 5 : 6);
			if (busy) { busy = false; free_at = fa; }
			else if (fa > free_at) free_at = fa;
		}
	}
	// ---- the completed messages leave together: the wave's notes in one list, one lane per message (46 words, all loads in flight at
	// once), one atomic for the wave -- and what the state for the next block needs is asked for in the same round trip
	const uint32_t* brow = dc.row;
	const int nw = (n + 31) >> 5;
	const uint32_t wl = nw > 0 ? brow[nw - 1] : 0u, wp = nw > 1 ? brow[nw - 2] : 0u; // the last 33+ decisions
	// the run that is still going: its state as k7e_sim left it becomes the decoder's state for the next block -- copied here, with the
	// wave's other requests in flight, not held across the copy-out below (as a local it lived in 216 bytes of scratch per lane)
	if (live && busy) *st = slots[slot_].s;
	__shared__ uint32_t s_note[K7E_FOUND * 64][2];
	{
		int incl = nfound;
		for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
		const int total = __shfl(incl, 63), first = incl - nfound;
		for (int i = 0; i < nfound; i++) { s_note[first + i][0] = s_found[i][lane]; s_note[first + i][1] = (uint32_t)d; }
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); // (one-wave workgroup: ordering only)
		unsigned fs0 = 0;
		for (int b0 = 0; b0 < total; b0 += 64) {
			const bool has = b0 + lane < total;
			const uint32_t v = has ? s_note[b0 + lane][0] : 0u;
			const int dn = has ? (int)s_note[b0 + lane][1] : d;
			const int e = (int)(v & 0xFFFFu);
			const K7Slot* sl = q.slot + (size_t)dn * K7E_OPENCAP + (v >> 16);
			uint32_t rec[4 + DEC_DATA_WORDS];
			rec[0] = (uint32_t)sl->s.position; rec[1] = __float_as_uint(sl->s.level);
			rec[2] = (uint32_t)(unsigned long long)sl->s.start_idx; rec[3] = (uint32_t)((unsigned long long)sl->s.start_idx >> 32);
#pragma unroll
			for (int w = 0; w < DEC_DATA_WORDS; w++) rec[4 + w] = sl->s.data[w];
			if (b0 == 0) { if (lane == 0) fs0 = atomicAdd(p.frame_count, (unsigned)total); fs0 = __shfl(fs0, 0); }
			if (has) {
				uint32_t* f = p.frames + (size_t)((fs0 + (unsigned)(b0 + lane)) % (unsigned)p.max_frames) * DEC_FRAME_WORDS;
				const long long sidx = 5 * (p.first_group + e) + k7e_decoder(p, dn).j;
				f[0] = (uint32_t)dn; f[1] = (uint32_t)e; f[2] = rec[0]; f[3] = rec[1];
				f[4] = rec[2]; f[5] = rec[3];
				f[6] = (uint32_t)(unsigned long long)sidx; f[7] = (uint32_t)((unsigned long long)sidx >> 32);
				f[8] = p.block; f[9] = p.sub;
#pragma unroll
				for (int w = 0; w < DEC_DATA_WORDS; w++) f[10 + w] = rec[4 + w];
			}
		}
	}
	if (!live) return;
	// (ModelChallenger) tag.sample_lvl as the block leaves it for this channel's FM decoders of the next one
	if (MESH == 10 && j == 0) p.last_lvl[dc.chan] = n > 0 ? p.lvl[(size_t)dc.chan * p.lvl_stride + n - 1] : p.last_lvl_in[dc.chan];
	// state for the next block
	if (busy) return;
	// TRAINING: lastBit, prev and the alternations counted (those that end at the last symbol, none before the restart)
	const auto dd_at = [&](int g) -> int {
		if (g < 0) return prev0;
		return (int)(((g >> 5) == nw - 1 ? wl : wp) >> (g & 31)) & 1;
	};
	const auto bit_at = [&](int g) -> int { return g < 0 ? last0 : (dd_at(g) == dd_at(g - 1)); }; // (g >= -1 only)
	int run = 0;
	for (int g = n - 1; run < 5; g--) {
		if (g < 0) { run += pos0 < 5 - run ? pos0 : 5 - run; break; } // the alternations carried in from the block before
		if (bit_at(g) == bit_at(g - 1)) break;
		run++;
	}
	int lim = n + 5 - free_at; // symbols since position was last set to zero
	lim = lim < 0 ? 0 : lim;
	st->state = DST_TRAINING; st->position = run < lim ? run : lim; st->osc = 0;
	st->lastBit = n > 0 ? bit_at(n - 1) : last0; st->prev = n > 0 ? dd_at(n - 1) : prev0;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
// The events of a front-end launch ride on the dispatch packet itself (hipExtLaunchKernelGGL binds them to the kernel command):
// "the front end of block f is done" -- what the phasor recurrence's stream waits for -- and the two time stamps of the roofline
// measurement then cost no barrier packets between two launches of the front stream (0.03 ms per step of 0.48).
struct K1Events { hipEvent_t start = nullptr, stop = nullptr; };
#define K1_LAUNCH(kernel_, ev_, grid_, s_, p_) K1_LAUNCH_LDS(kernel_, ev_, grid_, s_, p_, 0)
#define K1_LAUNCH_LDS(kernel_, ev_, grid_, s_, p_, lds_) \
	do { \
		if ((ev_).start || (ev_).stop) hipExtLaunchKernelGGL(kernel_, grid_, dim3(64), lds_, s_, (ev_).start, (ev_).stop, 0, p_); \
		else hipLaunchKernelGGL(kernel_, grid_, dim3(64), lds_, s_, p_); \
	} while (0)
// The pre-decimation pass needs few registers, so the hardware would take sixteen of its one-wave workgroups per CU -- all of the
// CU's LDS.  The resampler front end of the previous block (256 threads, 6.6 KB of LDS per workgroup), which runs beside it on the
// downstream stream, then only gets LDS as fast as workgroups of the pass retire (0.39 instead of 0.09 ms, BASELINE configs[2]).
// Unused dynamic LDS brings the pass to ten workgroups per CU: 60 KB stay free for the kernels behind the 48 kHz channels, which all
// run beside the next block's pass (12 per CU, like the main front end: 2 % slower per step; 8: no better).  Late in round 4, with the
// resampler front end on the downstream stream: ModelChallenger's 6 MSPS ladder (BASELINE configs[2]), whose back end is the largest,
// gained 1.5 - 3 % with eight per CU while its FM branch was a kernel of its own; with that branch inside k6_window_fir eight, ten
// and seven are within 1 % of each other; the other ladders lose 0 - 10 % with eight or seven.  Ten.
#ifndef K1_PRE_EXTRA_LDS
#define K1_PRE_EXTRA_LDS 6400
#endif

template <int K, int FMT>
static hipError_t launch_k1_dpp_kf(const K1Params& p, int spans, int n_rx, hipStream_t s, const K1Events& ev) {
	if constexpr (K < 6) { // (no pre-decimation pass has six stages: that form is not instantiated; five: 8 / 10 MSPS, two waves per SIMD already)
		if (p.pre_out != nullptr) { K1_LAUNCH_LDS((k1_dpp<K, FMT, true>), ev, dim3(spans, n_rx), s, p, K < 5 ? K1_PRE_EXTRA_LDS : 0); return hipGetLastError(); }
	}
	K1_LAUNCH((k1_dpp<K, FMT, false>), ev, dim3(spans, n_rx), s, p);
	return hipGetLastError();
}

template <int K>
static hipError_t launch_k1_dpp_k(const K1Params& p, int fmt, int spans, int n_rx, hipStream_t s, const K1Events& ev) {
	switch (fmt) {
	case 0: return launch_k1_dpp_kf<K, 0>(p, spans, n_rx, s, ev);
	case 1: return launch_k1_dpp_kf<K, 1>(p, spans, n_rx, s, ev);
	case 2: return launch_k1_dpp_kf<K, 2>(p, spans, n_rx, s, ev);
	case 3: return launch_k1_dpp_kf<K, 3>(p, spans, n_rx, s, ev);
	case 4:
		if constexpr (K == 4) {
			if (p.pre_out != nullptr) return hipErrorInvalidValue;
			K1_LAUNCH((k1_dpp<4, 4, false>), ev, dim3(spans, n_rx), s, p);
			return hipGetLastError();
		}
		break;
	case 5: // the tail of a resampled ladder: Upsample outputs computed in the wave (K1Params::us_idx)
		if constexpr (K == 2) {
			if (p.pre_out != nullptr || !p.us_idx || !p.us_alpha || !p.xin) return hipErrorInvalidValue;
			K1_LAUNCH((k1_dpp<2, 5, false>), ev, dim3(spans, n_rx), s, p);
			return hipGetLastError();
		}
		break;
	}
	return hipErrorInvalidValue;
}

static hipError_t launch_k1_dpp(const K1Params& p, int K, int fmt, int spans, int n_rx, hipStream_t s, const K1Events& ev) {
	switch (K) {
	case 6: return p.pre_out != nullptr ? hipErrorInvalidValue : launch_k1_dpp_k<6>(p, fmt, spans, n_rx, s, ev);
	case 5: return launch_k1_dpp_k<5>(p, fmt, spans, n_rx, s, ev);
	case 4: return launch_k1_dpp_k<4>(p, fmt, spans, n_rx, s, ev);
	case 3: return launch_k1_dpp_k<3>(p, fmt, spans, n_rx, s, ev);
	case 2: return launch_k1_dpp_k<2>(p, fmt, spans, n_rx, s, ev);
	case 1: return launch_k1_dpp_k<1>(p, fmt, spans, n_rx, s, ev);
	}
	return hipErrorInvalidValue;
}

hipError_t launch_k1(const K1Params& p, int K, int fmt, int spans, int n_rx, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
	K1Events ev; ev.start = ev_start; ev.stop = ev_stop;
	return launch_k1_dpp(p, K, fmt, spans, n_rx, s, ev);
}

#define K1U_LAUNCH(kernel_, ...) hipLaunchKernelGGL((kernel_<__VA_ARGS__ K1U_M>), dim3(p.L / K1U_M, n_rx), dim3(K1U_T), 0, s, p)
#define K1U_LAUNCH_SPW(kernel_, q_, ...) hipLaunchKernelGGL((kernel_<__VA_ARGS__ K1U_M>), dim3(p.L / K1U_M / (q_).spw, n_rx), dim3(K1U_T), 0, s, q_)
#define K1U_COMMA ,

#ifndef K1K_WAVE
#define K1K_WAVE 1
#endif
bool k1u96_wave_form(const K1uParams& p, int npost) { return K1K_WAVE && npost == 0 && !p.us_idx && p.L % 256 == 0 && p.spw_force != 2; }
hipError_t launch_k1u(const K1uParams& p, int npost, int n_rx, hipStream_t s) {
	if (k1u96_wave_form(p, npost)) { // dual-channel 96 kSPS input: k1k_wave without the filter (test hook "k1u_spw" = 2: the workgroup form, 4 / 8: spans of that many tiles)
		const int tiles = p.L / 256;
		int tps = 16;
		const int tps_min = p.fz ? 2 : 1;
		while (tps > tps_min && (long long)((tiles + tps - 1) / tps) * n_rx < 4096) tps >>= 1;
		if (p.spw_force > 2) tps = p.spw_force;
		hipLaunchKernelGGL((k1k_wave<false, K1uParams>), dim3((tiles + tps - 1) / tps, n_rx), dim3(64), 0, s, p, tps);
		return hipGetLastError();
	}
	if (p.fz) return hipErrorInvalidValue; // (the analysis in the waves exists in the wave form only)
	// spans per workgroup: as many as leave the chip four rounds of workgroups (every flush is a whole number of 512-sample windows = 4 spans)
	K1uParams q = p;
	const int spans = p.L / K1U_M;
	q.spw = K1U_SPW;
	if (p.spw_force > 1) { q.spw = p.spw_force < K1U_SPW ? p.spw_force : K1U_SPW; while (q.spw > 1 && spans % q.spw != 0) q.spw >>= 1; } // (tests: few receivers, long walks)
	else while (q.spw > 1 && (spans % q.spw != 0 || (long long)(spans / q.spw) * n_rx < K1U_MIN_WGS)) q.spw >>= 1;
	if (npost == 2) K1U_LAUNCH_SPW(k1u_resample_frontend, q, 2 K1U_COMMA);
	else if (npost == 1) K1U_LAUNCH_SPW(k1u_resample_frontend, q, 1 K1U_COMMA);
	else if (npost == 0) K1U_LAUNCH_SPW(k1u_resample_frontend, q, 0 K1U_COMMA);
	else return hipErrorInvalidValue;
	return hipGetLastError();
}

#ifndef K1X_M
#define K1X_M 512 // 48 kHz outputs per workgroup of the mode-X front end: a whole window (K1U_M = 128 made 786 k workgroups of a 4,096-receiver step)
#endif
#ifndef K1X_WAVE
#define K1X_WAVE 1
#endif
bool k1x_wave_form(const K1uParams& p, int npost) { return K1X_WAVE && npost >= 0 && npost <= 2 && !p.us_idx && p.L % 512 == 0 && p.spw_force != 2; }
hipError_t launch_k1x(const K1uParams& p, int npost, int n_rx, hipStream_t s) {
	if (k1x_wave_form(p, npost)) { // the register / DPP form (test hook "k1u_spw" = 2: the workgroup form)
		const int tiles = p.L / 512;
		int tps = 16; // tiles per span: ~4,096 waves or more where the batch has them (test hook "k1u_spw" = 4 / 8: spans of that many tiles)
		const int tps_min = p.fz ? 2 : 1; // (the analysis in the waves takes this row's windows = tiles in pairs)
		while (tps > tps_min && (long long)((tiles + tps - 1) / tps) * n_rx < 4096) tps >>= 1;
		if (p.spw_force > 2) tps = p.spw_force;
		if (p.fz && (tiles % 2 != 0)) return hipErrorInvalidValue; // (the caller sets fz only where a block is an even number of windows)
		const dim3 gw((tiles + tps - 1) / tps, n_rx);
		if (npost == 2) hipLaunchKernelGGL(k1x_wave<2>, gw, dim3(64), 0, s, p, tps);
		else if (npost == 1) hipLaunchKernelGGL(k1x_wave<1>, gw, dim3(64), 0, s, p, tps);
		else hipLaunchKernelGGL(k1x_wave<0>, gw, dim3(64), 0, s, p, tps);
		return hipGetLastError();
	}
	if (p.fz) return hipErrorInvalidValue; // (the analysis in the waves exists in the wave form only: the caller asks k1x_wave_form first)
	const dim3 grid(p.L / K1X_M, n_rx);
	if (npost == 2) hipLaunchKernelGGL((k1x_single_channel<2, K1X_M>), grid, dim3(K1U_T), 0, s, p);
	else if (npost == 1) hipLaunchKernelGGL((k1x_single_channel<1, K1X_M>), grid, dim3(K1U_T), 0, s, p);
	else if (npost == 0) hipLaunchKernelGGL((k1x_single_channel<0, K1X_M>), grid, dim3(K1U_T), 0, s, p);
	else return hipErrorInvalidValue;
	return hipGetLastError();
}

#ifndef K1K_M
#define K1K_M 128 // 48 kHz outputs per channel per workgroup of the decimate-by-3 front end
#endif
#ifndef K1K_WAVE_US
#define K1K_WAVE_US 1
#endif
bool k1k_wave_form(const K1kParams& p, int hook) { return K1K_WAVE && (!p.us_idx || K1K_WAVE_US) && p.L % 256 == 0 && hook != 2; }
hipError_t launch_k1k(const K1kParams& p, int n_rx, hipStream_t s, int hook) { // hook (test hook "k1u_spw"): 2 = the workgroup form, 4 / 8 = k1k_wave with spans of that many tiles
	if (k1k_wave_form(p, hook)) {
		const int tiles = p.L / 256;
		int tps = 16; // tiles per span: ~4,096 waves or more where the batch has them
		const int tps_min = p.fz ? 2 : 1; // (the analysis in the waves: two tiles are a window)
		while (tps > tps_min && (long long)((tiles + tps - 1) / tps) * n_rx < 4096) tps >>= 1;
		if (hook > 2) tps = hook;
		if (p.us_idx) hipLaunchKernelGGL((k1k_wave<true, K1kParams, true>), dim3((tiles + tps - 1) / tps, n_rx), dim3(64), 0, s, p, tps); // Upsample in the lanes
		else hipLaunchKernelGGL((k1k_wave<true, K1kParams>), dim3((tiles + tps - 1) / tps, n_rx), dim3(64), 0, s, p, tps);
		return hipGetLastError();
	}
	if (p.fz) return hipErrorInvalidValue; // (see launch_k1x)
	hipLaunchKernelGGL((k1k_dsk_frontend<K1K_M>), dim3(p.L / K1K_M, n_rx), dim3(K1U_T), 0, s, p);
	return hipGetLastError();
}

hipError_t launch_convert_rows(const void* in, long long in_stride, int fmt, float2* dst, long long dst_stride, int n, int n_rx, hipStream_t s) {
	int blocks = (n + 255) / 256;
	if (blocks > 256) blocks = 256;
	hipLaunchKernelGGL(k_convert_rows, dim3(blocks, n_rx), dim3(256), 0, s, (const unsigned char*)in, in_stride * fmt_bytes(fmt), fmt, dst, dst_stride, n);
	return hipGetLastError();
}

hipError_t launch_ma_rows(const void* in, long long in_stride, int fmt, int m, float2* dst, long long dst_stride, int n, int n_rx, hipStream_t s) {
	int blocks = (n + 255) / 256;
	if (blocks > 256) blocks = 256;
	hipLaunchKernelGGL(k_ma_rows, dim3(blocks, n_rx), dim3(256), 0, s, (const unsigned char*)in, in_stride * fmt_bytes(fmt), fmt, m, dst, dst_stride, n);
	return hipGetLastError();
}

hipError_t launch_copy_rows(const float2* src, long long src_stride, float2* dst, long long dst_stride, int n, int n_rx, hipStream_t s) {
	int blocks = (n + 255) / 256;
	if (blocks > 64) blocks = 64;
	hipLaunchKernelGGL(k_copy_rows, dim3(blocks, n_rx), dim3(256), 0, s, src, src_stride, dst, dst_stride, n);
	return hipGetLastError();
}

hipError_t launch_k1_tail(const void* in, long long in_stride_bytes, long long block_bytes, void* hist, int tail_bytes,
                          int n_rx, hipStream_t s) {
	int blocks = (tail_bytes / 16 + 255) / 256;
	if (blocks > 8) blocks = 8;
	hipLaunchKernelGGL(k1_tail, dim3(blocks, n_rx), dim3(256), 0, s, (const unsigned char*)in, in_stride_bytes, block_bytes,
	                   (unsigned char*)hist, tail_bytes);
	return hipGetLastError();
}

hipError_t launch_selftest_hypot(const float2* in, int n, unsigned* mismatches, hipStream_t s) {
	hipLaunchKernelGGL(k_selftest_hypot, dim3((n + 255) / 256), dim3(256), 0, s, in, n, mismatches);
	return hipGetLastError();
}

hipError_t launch_k2a_fft(const K2Params& p, int n_chan, hipStream_t s) {
	const int n = n_chan * p.n_windows;
	hipLaunchKernelGGL(k2_fft_mag, dim3((n + FFT_NW - 1) / FFT_NW), dim3(64), 0, s, p);
	return hipGetLastError();
}

hipError_t launch_k2a_fft_search(const K2Params& p, int n_chan, hipStream_t s) { // fz / ppm of every window straight from c48 (both channels of a receiver per wave)
	K1Params k{};
	k.c48 = const_cast<float2*>(p.c48); k.c48_stride = p.c48_stride; k.omega = p.omega; k.ppm_table = p.ppm_table; k.fz = p.fz; k.ppm = p.ppm;
	k.fft_windows = 1; k.n_windows = p.n_windows; k.wide = p.wide;
	hipLaunchKernelGGL(k2_fft_search_win, dim3(p.n_windows, n_chan / 2), dim3(64), 0, s, k);
	return hipGetLastError();
}

hipError_t launch_k2a_search(const K2Params& p, int n_chan, hipStream_t s) {
	const int n = n_chan * p.n_windows;
	hipLaunchKernelGGL(k2_cgf_search, dim3((n + 63) / 64), dim3(64), 0, s, p);
	return hipGetLastError();
}

hipError_t launch_k2b(const K2Params& p, int n_chan, hipStream_t s) {
	hipLaunchKernelGGL(k2_cgf_phasor, dim3((n_chan + 63) / 64), dim3(64), 0, s, p);
	return hipGetLastError();
}

hipError_t launch_k2b_ck(const K2Params& p, int n_chan, hipStream_t s, int simds) {
	if ((n_chan + 31) / 32 <= simds) { // one wave per SIMD at most: the latency-bound form with a chain per lane pair
		hipLaunchKernelGGL(k2_cgf_phasor_ck_pairs, dim3((n_chan + 31) / 32), dim3(64), 0, s, p);
		return hipGetLastError();
	}
	hipLaunchKernelGGL(k2_cgf_phasor_ck, dim3((n_chan + 63) / 64), dim3(64), 0, s, p);
	return hipGetLastError();
}

hipError_t launch_k2b_refine(const K2Params& p, int n_chan, hipStream_t s) {
	hipLaunchKernelGGL(k2_cgf_refine, dim3((n_chan + 63) / 64, p.n_windows), dim3(64), 0, s, p);
	return hipGetLastError();
}

hipError_t launch_k6(const K6Params& p, hipStream_t s) {
	const dim3 grid((unsigned)((p.n_chan + 63) / 64 * 64 * p.n_windows));
	if (p.fmbits) hipLaunchKernelGGL((k6_window_fir<true>), grid, dim3(64), 0, s, p);
	else hipLaunchKernelGGL((k6_window_fir<false>), grid, dim3(64), 0, s, p);
	return hipGetLastError();
}

hipError_t launch_k2c(const K2Params& p, int n_chan, hipStream_t s) { // history carry (inside the last tile's workgroups) + apply
	hipLaunchKernelGGL(k2_cgf_apply, dim3(p.n_windows * 8, (n_chan + 63) / 64), dim3(256), 0, s, p);
	return hipGetLastError();
}

hipError_t launch_k3(const K3Params& p, int n_chan, hipStream_t s) {
	if (p.n_groups <= 0) return hipSuccess;
	hipLaunchKernelGGL(k3_fir_scatter, dim3((p.n_groups + 255) / 256, n_chan), dim3(256), 0, s, p);
	return hipGetLastError();
}

hipError_t launch_k7(const K7Params& p, hipStream_t s) {
	hipLaunchKernelGGL(k7_decode, dim3((p.n_chan + 11) / 12), dim3(64), 0, s, p);
	return hipGetLastError();
}

hipError_t launch_kv2_assist(const KV2Params& p, hipStream_t s, int part) { // part 1: estimates; 2: the FM branch; 4: energies; 7: all
	if (part & 1) {
		const int n_est = p.n_chan * 2 * p.n_windows;
		hipLaunchKernelGGL(kv2_estimate, dim3((n_est + V2E_NW - 1) / V2E_NW), dim3(64), 0, s, p);
	}
	// (the energies ride in the FM branch's launch as its first rows of workgroups where both are asked for: a 15 us kernel less on the step's chain)
	if (part & 2) {
		KV2Params pe = p;
		const int per_row = (p.L / 256) * 256;
		pe.energy_rows = (part & 4) ? (p.n_chan * (p.n_windows + 1) + per_row - 1) / per_row : 0;
		hipLaunchKernelGGL(kv2_fm_filter, dim3(p.L / 256, p.n_chan + pe.energy_rows), dim3(256), 0, s, pe);
	} else if (part & 4) hipLaunchKernelGGL(kv2_energy, dim3((p.n_chan * (p.n_windows + 1) + 63) / 64), dim3(64), 0, s, p);
	return hipGetLastError();
}
hipError_t launch_kv2_engine(const KV2EParams& e, hipStream_t s) { // one workgroup per channel (reads the look-back `hist`)
	if (e.roles == 2) hipLaunchKernelGGL(kv2_engine_roles_dense, dim3(e.k.n_chan), dim3(192), 0, s, e);
	else if (e.roles) hipLaunchKernelGGL(kv2_engine_roles, dim3(e.k.n_chan), dim3(192), 0, s, e);
	else hipLaunchKernelGGL(kv2_engine, dim3(e.k.n_chan), dim3(64), 0, s, e);
	return hipGetLastError();
}
hipError_t launch_kv2_carry(const KV2Params& p, hipStream_t s) {
	hipLaunchKernelGGL(kv2_carry, dim3(p.n_chan), dim3(64), 0, s, p);
	return hipGetLastError();
}
hipError_t launch_kv2(const KV2Params& p, hipStream_t s, const KV2EParams* engine) {
	hipError_t e = launch_kv2_assist(p, s, 7);
	if (e == hipSuccess && engine) e = launch_kv2_engine(*engine, s);
	if (e == hipSuccess) e = launch_kv2_carry(p, s);
	return e;
}

hipError_t launch_k7_pack(const K7Params& p, hipStream_t s) {
	if (p.n_groups > 0) hipLaunchKernelGGL(k7_pack_fm, dim3(((unsigned)p.fmrows_stride + 255) / 256, p.n_chan * 5), dim3(256), 0, s, p);
	return hipGetLastError();
}
hipError_t launch_k7_mesh(const K7Params& p, hipStream_t s) {
	if (p.kind == 3) {
		hipLaunchKernelGGL(k7_base, dim3((p.n_chan + 63) / 64), dim3(64), 0, s, p);
		return hipGetLastError();
	}
	if (p.kind == 1) hipLaunchKernelGGL(k7_decode_mesh<1>, dim3((p.n_chan + 11) / 12), dim3(64), 0, s, p);
	else hipLaunchKernelGGL(k7_decode_mesh<2>, dim3((p.n_chan + 5) / 6), dim3(64), 0, s, p);
	return hipGetLastError();
}

hipError_t launch_k7b_spec(const K7bParams& q, hipStream_t s) {
	hipLaunchKernelGGL(k7b_spec, dim3((q.k.n_chan + 63) / 64, q.n_chunks), dim3(64), 0, s, q);
	return hipGetLastError();
}

hipError_t launch_k7b_finish(const K7bParams& q, hipStream_t s) {
	const int gx = (q.k.n_chan + 63) / 64;
	hipLaunchKernelGGL(k7b_task, dim3((q.k.n_chan + K7B_TL - 1) / K7B_TL, q.n_chunks), dim3(64), 0, s, q);
	hipLaunchKernelGGL(k7b_walk, dim3(gx), dim3(64), 0, s, q);
	hipLaunchKernelGGL(k7b_emit, dim3(gx, q.n_chunks), dim3(64), 0, s, q);
	K7Params fb = q.k; // exact fallback: exits at once unless a channel is flagged
	fb.cond = q.fallback; fb.cond_count = q.fallback_count;
	hipLaunchKernelGGL(k7_base, dim3(gx), dim3(64), 0, s, fb);
	return hipGetLastError();
}

hipError_t launch_k7e_runs(const K7eParams& q, hipStream_t s) { // candidate scan + one run per possible frame: wide kernels
	const int n_dec = q.k.n_chan * (q.k.kind == 2 ? 10 : 5);
	if (q.k.n_groups <= 0) return hipSuccess;
	hipLaunchKernelGGL(k7e_scan, dim3((n_dec + 3) / 4), dim3(64), 0, s, q);
	hipLaunchKernelGGL(k7e_sim, dim3((n_dec * K7E_SIM_LANES + 63) / 64), dim3(64), 0, s, q);
	return hipGetLastError();
}

hipError_t launch_k7e_resolve(const K7eParams& q, hipStream_t s) { // the walk: a few dozen latency-bound waves
	if (q.k.n_groups <= 0) return hipSuccess;
	if (q.k.kind == 2) hipLaunchKernelGGL((k7e_resolve<10, 16>), dim3((q.k.n_chan + 3) / 4), dim3(64), 0, s, q);
	else hipLaunchKernelGGL((k7e_resolve<5, 8>), dim3((q.k.n_chan + 7) / 8), dim3(64), 0, s, q);
	return hipGetLastError();
}

hipError_t launch_k5(const K5Params& p, int n_chan, hipStream_t s) {
	if (!p.hist_in || !p.hist_out) return hipErrorInvalidValue; // the first / last tile of a row dereference them
	hipLaunchKernelGGL(k5_fm_filter, dim3(p.L / 256, n_chan), dim3(256), 0, s, p);
	return hipGetLastError();
}

hipError_t launch_k4_box(const K4Params& p, hipStream_t s) {
	if (p.n_chunks > 1 && p.chunked) { // chunk-parallel: exact by construction (16 symbols of look-back), k4_assemble picks the trajectories
		K4Params q = p;
		hipLaunchKernelGGL(k4_box_chunks, dim3((p.n_chains / 5 + 3) / 4 * 5, p.n_chunks), dim3(64), 0, s, q);
		hipLaunchKernelGGL(k4_assemble, dim3((p.n_chains + 3) / 4), dim3(64), 0, s, q);
		return hipGetLastError();
	}
	hipLaunchKernelGGL(k4_phase_search_box, dim3((p.n_chains + 3) / 4), dim3(64), 0, s, p);
	return hipGetLastError();
}

hipError_t launch_k4_sequential(const K4Params& p, hipStream_t s) {
	if (p.n_groups <= 0) return hipSuccess;
	hipLaunchKernelGGL(k4_phase_search, dim3((p.n_chains + 3) / 4), dim3(64), 0, s, p);
	return hipGetLastError();
}

hipError_t launch_k46(K46Params q, hipStream_t st) {
	if (q.s.n_groups <= 0) return hipSuccess;
	const int trips = (q.f.n_chan + K46_CH - 1) / K46_CH;
	q.trips_pad = (trips + 7) / 8 * 8;
	hipLaunchKernelGGL(k46_window_search, dim3((unsigned)q.trips_pad * (unsigned)q.s.n_chunks), dim3(256), K46_LDS, st, q);
	hipLaunchKernelGGL(k46_assemble, dim3((q.s.n_chains + 3) / 4), dim3(64), 0, st, q); // + the exact fallback where a warm-up failed
	return hipGetLastError();
}

hipError_t launch_k4(const K4Params& p, hipStream_t s) {
	if (p.n_groups <= 0) return hipSuccess;
	hipLaunchKernelGGL(k4_phase_chunks, dim3((p.n_chains / 5 + 3) / 4 * 5, p.n_chunks), dim3(64), 0, s, p);
	hipLaunchKernelGGL(k4_assemble, dim3((p.n_chains + 3) / 4), dim3(64), 0, s, p); // + the exact sequential search where a warm-up failed
	return hipGetLastError();
}

} // namespace aisk
