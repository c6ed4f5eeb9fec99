// ais-catcher_amd/csrc/kernels.h -- parameter blocks and launchers shared by kernels.hip and aisgpu.cpp
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aisk {

constexpr int CGF_HIST = 64;   // 48 kHz samples of CGF output carried in front of each block (FIR-17: 16 + 4; FM branch: 37 + 1)
constexpr int ROT_HIST = 256;  // 96 kHz rotator phasors carried in front of each block's table (one tile)
constexpr int FZ_MIN = -205, FZ_COUNT = 414; // CGF peak index range

struct K1Params {
	const void* in;          // [n_rx][in_stride] input samples (float2 or uchar2)
	long long in_stride;     // samples
	const void* hist;        // [n_rx][tile] last tile of the previous block
	void* hist_out;          // != nullptr (k1_dpp, CF32, K >= 2): the wave that reads the block's last tile saves it here for the next block
	const float2* rot;       // [ROT_HIST + block_len >> K] Rotate phasor per 96 kHz sample (host generated)
	float2* c48;             // [n_rx][2][c48_stride] 48 kHz front-end output (FCIC5_a/b.out)
	long long c48_stride;
	int tiles_per_block, tiles_per_span;
	float alpha, beta;       // FilterComplex3Tap
	int has_fdc;
	int stream_start;        // != 0: first block of the stream (only the fixed-point ladder needs to know)
	float2* pre_out;         // != nullptr: pre-decimation pass, write the level after K stages here ([n_rx][pre_stride])
	long long pre_stride;
	// spectral analysis at the end of every span (k1_dpp only): fft_windows = tiles_per_span / 16 windows per channel, 0 = off
	int fft_windows, n_windows, wide;
	const float2* omega;     // [512] FFT twiddles
	const float* ppm_table;  // [FZ_COUNT]
	int* fz;                 // [n_rx * 2][n_windows]
	float* ppm;              // [n_rx * 2][n_windows]
	// FMT 5 (round 6: the tail of a resampled ladder in the front-end waves, k1_dpp<2, 5, false>): the block is one flush of Upsample
	// (DSP.cpp:192-212) and its samples are not read but COMPUTED where the other formats convert -- input sample n of the block is
	// Upsample output n of the flush, (1 - alpha) * x[b - 1] + alpha * x[b], (b, alpha) from the tables of K1uParams ([US_HIST + len]),
	// x the pre-decimated stream with its ring of earlier blocks (the fields of K1uParams: make_xrow)
	const int* us_idx = nullptr; const float* us_alpha = nullptr;
	const float2* xin = nullptr; long long xin_stride = 0; long long xin_off = 0;
	const float2* xprev = nullptr; const float2* xprev2 = nullptr; int n_in = 0;
	const float2* xhist = nullptr; int xhist_len = 0;
};

constexpr int US_HIST = 96;  // resampler table entries carried in front of each flush block (halo of K1u: 84)

struct K1uParams {
	const float2* xin; long long xin_stride; long long xin_off; // xin[rx * stride + off + i]: pre-decimated sample i of the current block (i < 0: history)
	const int* us_idx; const float* us_alpha; // [US_HIST + len]: Upsample output n -> (input index b, alpha); out = (1 - alpha) * x[b - 1] + alpha * x[b]
	const float2* rot;      // [ROT_HIST + len / 4]
	float2* c48; long long c48_stride;
	float alpha, beta; int has_fdc;
	int L;                  // 48 kHz samples per channel per flush block (len / 8)
	// resampled ladders: the input blocks live in a ring of three buffers instead of one buffer with a copied history (a flush may
	// begin a whole input block back): sample i < 0 is xprev[n_in + i], i < -n_in is xprev2[2 n_in + i] (rows of xin_stride, no offset)
	const float2* xprev = nullptr; const float2* xprev2 = nullptr; int n_in = 0;
	// CF32 input read where the caller put it (no converted copy): xin = the caller's rows (xin_off = 0), and the samples in front of the
	// block come from xhist[rx * xhist_len + xhist_len + i], i in [-xhist_len, 0) -- the previous block's tail, kept by the library
	const float2* xhist = nullptr; int xhist_len = 0;
	int c48_rows_per_rx = 2; // k1x_single_channel: rows of c48 per receiver -- 2: the receiver's channel A of a dual-channel layout; 1 (round 6): receivers packed, row = receiver
	int spw = 1;            // spans (of K1U_M outputs per channel) a workgroup of the resampler front end walks (set by launch_k1u)
	int spw_force = 0;      // test hook "k1u_spw" (2 / 4 / 8): the span walk of that length whatever the number of workgroups it leaves	// the spectral analysis at the end of the waves of the one-wave front ends (k1x_wave / k1k_wave, round 6): fz != nullptr -- every wave
	// finishes its span with SquareFreqOffsetCorrection's FFT + searches of the windows it has written (as k1_dpp's k1_fft_tail)
	const float2* omega = nullptr; const float* ppm_table = nullptr; int* fz = nullptr; float* ppm = nullptr; int n_windows = 0, wide = 0;
};
// sample i of the pre-decimated stream relative to the current input block's start (see K1uParams::xprev)
struct XRow {
	const float2 *cur, *prev, *prev2; int n;
	__host__ __device__ float2 operator[](int i) const { return (i >= 0 || !prev) ? cur[i] : (i >= -n ? prev[n + i] : prev2[2 * n + i]); }
};
// the row of receiver rx for either arrangement of the samples in front of the block
template <class P>
__host__ __device__ inline XRow make_xrow(const P& p, int rx) {
	const size_t xrow = (size_t)rx * p.xin_stride + p.xin_off;
	if (p.xhist) return XRow{ p.xin + xrow, p.xhist + (size_t)rx * p.xhist_len, nullptr, p.xhist_len };
	return XRow{ p.xin + xrow, p.xprev ? p.xprev + xrow : nullptr, p.xprev2 ? p.xprev2 + xrow : nullptr, p.n_in };
}
// the same for a workgroup that only touches samples [lo, hi]: nearly always they lie in ONE of the three buffers, and the access is a
// plain pointer again (workgroup-uniform choice); only a span that straddles a block boundary goes through the three-way select
struct XSpan {
	XRow row; const float2* base; bool mixed;
	__host__ __device__ XSpan(const XRow& r, int lo, int hi) : row(r) {
		mixed = false;
		if (lo >= 0 || !r.prev) base = r.cur;
		else if (hi < 0 && lo >= -r.n) base = r.prev + r.n;
		else if (hi < -r.n) base = r.prev2 + 2 * r.n;
		else { base = r.cur; mixed = true; }
	}
	__host__ __device__ float2 operator[](int i) const { return mixed ? row[i] : base[i]; }
};

struct K1kParams { // decimate-by-3 front end (DownsampleKFilter ladders: 288k * 2^k)
	const float2* xin; long long xin_stride; long long xin_off; // 288 kHz stream of the current block (i < 0: history, >= 70 samples)
	const float2* rot;      // [ROT_HIST + n96]
	float2* c48; long long c48_stride;
	float taps[26];         // Filters::BlackmanHarris_28_3
	const int* us_idx; const float* us_alpha; // != nullptr: [US_HIST + len] Upsample in front of the filter (see K1uParams), xin is ITS input
	int L;                  // 48 kHz samples per channel per block
	const float2* xprev = nullptr; const float2* xprev2 = nullptr; int n_in = 0; // ring of three input blocks (see K1uParams)
	const float2* xhist = nullptr; int xhist_len = 0; // CF32 input in place + kept tail (see K1uParams)	// the spectral analysis at the end of the waves of the one-wave front ends (k1x_wave / k1k_wave, round 6): fz != nullptr -- every wave
	// finishes its span with SquareFreqOffsetCorrection's FFT + searches of the windows it has written (as k1_dpp's k1_fft_tail)
	const float2* omega = nullptr; const float* ppm_table = nullptr; int* fz = nullptr; float* ppm = nullptr; int n_windows = 0, wide = 0;
};
constexpr int DSK_HIST = 128; // samples of the 288 kHz stream kept in front of a block

struct K2Params {
	const float2* c48; long long c48_stride;
	float2* cgf; long long cgf_stride;   // [n_chan][CGF_HIST + L]
	const float2* omega;      // [512] FFT twiddles
	const float2* step_table; // [FZ_COUNT] rot_step per fz
	const float* ppm_table;   // [FZ_COUNT]
	float* magT;              // [ceil(n_chan * n_windows / 64)][512][64] shifted FFT magnitudes, window-minor
	int* fz;                  // [n_chan][n_windows]
	float* ppm;               // [n_chan][n_windows]
	float2* rot_state;        // [n_chan]
	float2* rotT; long long rotT_stride; // [L][rotT_stride] derotation phasor per sample, time-major
	int n_windows, wide, n_chan;
	// checkpointed recurrence (k2_cgf_phasor_ck): per (chain, window) CK_SLOTS entries, entry i = the state BEFORE sample
	// 512 w + CK_SEG i of the block (i = 0: the window's start, after the renormalisation), CK_USED of them written
	float2* ck; long long ck_stride; // [n_windows][CK_SLOTS][ck_stride]: time-major, chains padded to 64
};

// Fused derotation + FilterComplex(Coherent) + ScatterPLL (k6_window_fir): one wave = one chain x one 512-sample window, lanes over
// time.  The recurrence kernel leaves its state every CK_SEG samples of every window, so lane i restarts it for its own nine
// samples: the derotated window goes through LDS, the lanes then own one ScatterPLL group each.
constexpr int FM_HIST = 36; // discriminator values in front of a block / window that the 37-tap Receiver filter reaches back to
constexpr int DF_HIST = 40; // derotated samples carried from block to block (17-tap history + a partial group: 20; the FM branch inside k6_window_fir: 37)
constexpr int CK_SEG = 9, CK_USED = 57, CK_SLOTS = 64; // 56 segments of nine samples and one of eight per window
struct K6Params {
	const float2* c48; long long c48_stride;
	const float2* ck; long long ck_stride;       // phasor checkpoints [n_windows][CK_SLOTS][ck_stride]
	const float2* step_table; const int* fz;     // fz[chain][n_windows]
	const float2* hist_in; float2* hist_out;     // [n_chan][DF_HIST]
	float2* sym; long long sym_stride;           // SymRow layout, sym_stride = group capacity (also the row pitch of lvl)
	float* lvl;                                   // [n_chan][sym_stride]
	uint32_t* fmbits = nullptr; long long fmbits_stride = 0; float fm_taps[37] = {}; // optional: ModelChallenger's FM branch inside the kernel -- sign of the filtered discriminator, [n_chan][L / 32]
	float taps[17];
	long long first_group;
	int n_rel0;                                   // first_group * 5 - first_sample48, in [-4, 0]
	int n_groups, L, n_windows, n_chan;
};

struct K3Params {
	const float2* cgf; long long cgf_stride;
	float2* sym; long long sym_stride;   // [n_chan][5][sym_stride]
	float* lvl;                           // [n_chan][sym_stride]
	float2* fir_tap; long long fir_tap_stride; // optional [n_chan][4 + L]
	float taps[17];
	long long first_group, first_sample48;
	int n_groups;
};

// Layout of the FIR / ScatterPLL output `sym` (what PhaseSearch consumes): one row of `gcap` float2 per (channel, sampling phase),
// group g at element g.  Both sides see whole lines: the derotation / FIR kernel has its lanes over the groups of ONE channel's
// window (64 lanes x 8 bytes = 512 contiguous bytes per phase and store), a PhaseSearch row fetches 16 consecutive symbols of its
// chain (128 bytes) per lane group.  (Rounds 1-3 had one lane per CHANNEL in the producer and a layout interleaved over 64
// channels for it.)
__host__ __device__ inline size_t sym_row_base(int chan, int j, long long gcap) { return ((size_t)chan * 5 + (size_t)j) * (size_t)gcap; }
__host__ __device__ inline size_t sym_offset(int g) { return (size_t)g; }
__host__ __device__ inline size_t sym_elems(int n_chan, long long gcap) { return (size_t)n_chan * 5 * (size_t)gcap; }
// one (channel, phase) row: element g at base[g]
struct SymRow {
	const float2* base;
	__host__ __device__ SymRow(const float2* sym, int chain /* chan * 5 + j */, long long gcap) : base(sym + (size_t)chain * (size_t)gcap) {}
	__host__ __device__ float2 operator[](int g) const { return base[g]; }
};

struct EmaState { float ma[16]; unsigned bits[16]; int max_idx, rot; int pad[2]; };

// Demod::PhaseSearch (boxcar, `-go PS_EMA off`): per chain the |t| ring of the 16 hypotheses (slot-major), the decision
// shift registers and max_idx
struct PsBoxState { float mem[12][16]; unsigned bits[16]; int max_idx; int pad[3]; };

#ifndef PS_CHUNK_
#define PS_CHUNK_ 1024
#endif
constexpr int PS_CHUNK = PS_CHUNK_;   // symbols per time chunk of the chunk-parallel PhaseSearchEMA (multiple of 32)
constexpr int PS_MAXCHUNKS = 16;

struct K4Params {
	const float2* sym; long long sym_stride; // chain c row = c * sym_stride
	uint32_t* bits; long long bits_stride;   // words per chain
	const EmaState* state_in;                // state before this block
	EmaState* state_out;                     // state after this block
	// chunk-parallel scratch, all indexed [chain][chunk][...]
	uint32_t* words;   // [PS_CHUNK/32][16] output words per possible start index
	float* ma_start;   // [16] EMA after the warm-up (speculative), chunk > 0
	float* ma_fin;     // [16] EMA at the end of the chunk
	unsigned* fin;     // [16] low 4 bits: final max_idx per start index; bits 4..7: last four decisions of hypothesis k
	int n_chains, n_groups, n_chunks, warm;
	int chunked = 1;   // boxcar variant: 0 = the sequential row kernel (one-chunk blocks), 1 = k4_box_chunks
	// boxcar variant (k4_phase_search_box)
	const PsBoxState* box_in; PsBoxState* box_out; long long first_group;
	int* fb_count = nullptr;                 // statistics: workgroups of the exact fallback that really ran (aisgpu_ps_fallbacks)
};

// K7: AIS::Decoder on the device (frame decoder).  One lane per decoder, 12 meshes of 5 decoders per wave.
constexpr int DEC_DATA_WORDS = 36;  // MAX_AIS_FRAME_LENGTH = 1064 + 16 + 7 bits -> 136 bytes
constexpr int DEC_FRAME_WORDS = 10 + DEC_DATA_WORDS; // record: decoder, group, position, level bits, start_idx (2), end_idx (2), block, sub, data
struct DecState { int state, lastBit, prev, position, osc; float level; long long start_idx; uint32_t crc[8]; uint32_t data[DEC_DATA_WORDS]; };
struct K7Params {
	const uint32_t* bits; long long bits_stride; // [n_chan * 5][bits_stride] packed hard decisions
	const float* lvl; long long lvl_stride;      // [n_chan][lvl_stride]
	DecState* state;                              // [n_chan * 5], updated in place
	uint32_t* frames; unsigned* frame_count; int max_frames; // ring of records, monotonic counter
	long long first_group; int n_groups, n_chan;
	unsigned block, sub;                          // stamped into the records
	// the other engines' decoder wirings (k7_decode_mesh / k7_base); kind: 0 ModelDefault (5 coherent decoders per channel),
	// 1 ModelStandard (5 decoders on the deinterleaved FM discriminator, Model.cpp:505-514), 2 ModelChallenger (10 per channel:
	// coherent + FM, Model.cpp:641-674), 3 ModelBase (SimplePLL + one decoder with its feedback, Model.cpp:428-435)
	int kind = 0;
	const uint32_t* fm_cur = nullptr; const uint32_t* fm_prev = nullptr; long long fm_stride = 0; // [n_chan][L / 32] sign of the filtered discriminator, this / previous block
	uint32_t* fmrows = nullptr; long long fmrows_stride = 0; // [n_chan * 5][words] scratch: the FM bits regrouped per decoder (sample 5 g + j -> row j, bit g)
	const float* last_lvl_in = nullptr; float* last_lvl = nullptr; // [n_chan] ScatterPLL level of the last group of the previous / of this block (what tag.sample_lvl still holds)
	int n_rel0 = 0, L = 0;                        // first_group * 5 - first_sample48; 48 kHz samples per block
	// sequential kernels as the exact fallback of the event-driven ones: run only where *cond != 0, count the passes that ran in *cond_count
	int* cond = nullptr; int* cond_count = nullptr; // (k7_base: cond = one flag per channel, cleared by the kernel)
};
// K7b: ModelBase's sampler + decoder loop (DSP::SimplePLL with the decoder's StartTraining / StopTraining feedback, DSP/DSP.cpp:28-57,
// Model.cpp:428-435), chunk-parallel and exact.  The loop gain of the sampler follows the decoder's state sample by sample, so the
// coupled system is sequential -- but it FORGETS: in fast mode (decoder in TRAINING) every sign change contracts the PLL phase by
// 0.4, and a decoder in TRAINING is a function of its last few symbols.  So:
//   k7b_spec      one lane per (channel, chunk of K7B_CH samples): every chunk starts K7B_WARM samples early from a fresh state (the
//                 block's first chunk on the tail of the previous block's row) and records its trajectory: the sampler / decoder state
//                 in front of every 32nd sample (12 bytes), the frames it completes, and its full state at the chunk's end.  Needs
//                 nothing but the FM rows: runs on its own stream beside the previous block's tasks (two sets of scratch);
//   k7b_task      one lane per chunk boundary whose speculative state is NOT bit-identical with the true one -- the previous chunk's
//                 end state, for boundary 0 the state the previous block left (a frame in flight, a silent stretch): the exact loop
//                 from that state on, until its state equals a recorded one of the speculative trajectory it runs alongside (both
//                 decoders in TRAINING: from there on the two are the same for ever) -- or until the block ends.  Word by word: inside a
//                 frame the sampler alone over four words and the word-parallel frame evaluator; in TRAINING the sampler alone over
//                 a word and the five-instruction TRAINING step; symbol by symbol only where the decoder changes state;
//   k7b_walk      one lane per channel scans the boundaries in order, decides which trajectory is the true one where, notes per frame
//                 list whether (and from which sample on) it is the channel's;
//   k7b_emit      one lane per list copies the noted frames to the ring, and the channel's final state to DecState.
// A frame list that overflows (more than K7B_FCAP frames in a chunk / task) flags the channel; k7_base then decodes it from the
// untouched carried state (K7Params::cond = per-channel flags).  Same DecState between blocks as k7_base: the two can alternate.
#ifndef K7B_CH_
#define K7B_CH_ 512
#endif
constexpr int K7B_CH = K7B_CH_; // samples per chunk (multiple of 32): 48 chunks per channel in a 786,432-sample block at 1536 kSPS
#ifndef K7B_WARM_
#define K7B_WARM_ 256
#endif
constexpr int K7B_WARM = K7B_WARM_;   // samples of warm-up in front of a speculative chunk (<= K7B_CH, multiple of 32): ~50 sign changes of 0.4 each
constexpr int K7B_FCAP = 4;     // frames recorded per chunk / per task
constexpr int K7B_MAXC = 96;    // chunks per block at most (and L < 65534: 16-bit merge positions in k7b_walk's notes); beyond: k7_base alone
constexpr int K7B_FREC = 2 + DEC_DATA_WORDS; // sample index, position, data
struct K7bCkpt { uint32_t pll, position, flags; }; // flags: pprev | state << 1 | lastBit << 3 | prev << 4 | osc << 5
struct K7bParams {
	K7Params k;
	int n_chunks;
	K7bCkpt* ckpt;        // [n_chunks][K7B_CH / 32][n_chan_pad]: state in front of sample 32 i of the chunk
	DecState* end;        // [n_chunks][n_chan_pad] full state at the end of the chunk's trajectory
	uint32_t* frames;     // [n_chunks][n_chan_pad][1 + K7B_FCAP * K7B_FREC] count, records
	int* task_merge;      // [n_chunks][n_chan_pad] per boundary c >= 1: -1 no task (states matched), else the sample at which the task merged (L: never)
	DecState* task_end;   // [n_chunks][n_chan_pad] state of a task that ran to the end of the block
	uint32_t* task_frames;// like frames
	uint8_t* sum_spec;    // [n_chan_pad][K7B_MAXC] frames in the chunk's list (rows per channel: k7b_walk fetches a channel's boundaries with wide loads)
	uint32_t* sum_task;   // [n_chan_pad][K7B_MAXC] per boundary: task_merge + 1 (0: no task) | frames in the task's list << 16
	uint32_t* take_spec;  // [n_chunks][n_chan_pad] k7b_walk's notes: the chunk's list (of (v >> 12) & 15 frames) is the channel's from sample v >> 16 on, at offset v & 0xFFF of its frames; ~0: not
	uint32_t* take_task;  // likewise for the boundary's task list (offset; ~0: not)
	uint32_t* out_base;   // [n_chan_pad] k7b_walk -> k7b_emit: the ring position of the channel's first frame of this block
	uint32_t* fin_sel;    // [n_chan_pad] k7b_walk -> k7b_emit: where the channel's state after this block is (see k7b_walk)
	int* fallback;        // [n_chan_pad] != 0: a frame list overflowed, k7_base decodes the channel's block
	int* fallback_count;  // statistics
	int n_chan_pad;
	int fcap = K7B_FCAP;  // frames a list takes (<= K7B_FCAP; a smaller value is a test hook: it forces the fallback)
};
// k7b_spec needs the block's FM rows only (and scratch of its own: the caller alternates two sets), so it may run beside the previous
// block's launch_k7b_finish(); that one (k7b_task, k7b_walk, k7b_emit, the conditional k7_base) is what carries the channels' state from
// block to block and runs in block order.
hipError_t launch_k7b_spec(const K7bParams& p, hipStream_t s);
hipError_t launch_k7b_finish(const K7bParams& p, hipStream_t s);
hipError_t launch_k7(const K7Params& p, hipStream_t s);
hipError_t launch_k7_pack(const K7Params& p, hipStream_t s); // kind 1 / 2: regroup the FM bits per decoder (on the stream that produced them)
hipError_t launch_k7_mesh(const K7Params& p, hipStream_t s); // kind 1 / 2 / 3: the decoders

// K7e: the same decoders, event driven.  A decoder in TRAINING can only leave it where two equal NRZI bits follow at least five
// alternations -- a pure function of the hard bits -- so the places where a frame could start are found bit-parallel (k7e_scan),
// every possible frame is run from its start to its end by a lane of its own (k7e_sim: O(frame length) dependent steps instead
// of O(block length)), and one lane per channel then decides, in the reference's time order, which of them really happened
// (k7e_resolve: a frame is real if its decoder was in TRAINING long enough before it, and it is cut short by the Reset of a
// sibling that completes a message first).  State between blocks is the same DecState as the sequential kernel's.
constexpr int K7E_EVCAP = 1024;  // events per decoder and block (candidates are >= 6 symbols apart)
constexpr int K7E_OPENCAP = 128; // frames run per decoder and block; a block with more raises *overflow and is decoded by the sequential kernel
struct K7Slot { int end, flags; DecState s; }; // flags: 1 found, 2 still running at the end of the block
struct K7eParams {
	K7Params k;
	uint32_t* ev;        // [n_dec][K7E_EVCAP]  c | kind << 13 | (fail offset) << 15 | slot << 19
	uint32_t* cnt;       // [n_dec]             events | runs << 16
	uint16_t* open_c;    // [n_dec][K7E_OPENCAP] first symbol of run k (0xFFFF: the frame carried over from the previous block)
	K7Slot* slot;        // [n_dec][K7E_OPENCAP]
	int* overflow;       // != 0 behind k7e_scan: a decoder has more candidates / frame starts in this block than the lists hold -- k7e_sim and
	                     // k7e_resolve then leave everything as it is and the sequential kernel (K7Params::cond) decodes the block
	int* overflow_clear; // the flag of the pass before (cleared by this pass's scan: everything that looked at it has run)
};
hipError_t launch_k7e_runs(const K7eParams& p, hipStream_t s);
hipError_t launch_k7e_resolve(const K7eParams& p, hipStream_t s);

// fmt (kernel numbering): 0 = CF32, 1 = CU8, 2 = CS8, 3 = CS16, 4 = CU8 through the fixed-point ladder (K = 4)
// ev_start / ev_stop: events bound to the dispatch itself (no barrier packets): time stamps, and "this launch is done" for other streams
hipError_t launch_k1(const K1Params& p, int K, int fmt, int spans, int n_rx, hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
hipError_t launch_k1u(const K1uParams& p, int npost, int n_rx, hipStream_t s); // npost: CIC5 stages behind the resampler (2, 1; 0 = 96 kSPS input, no resampler)
hipError_t launch_copy_rows(const float2* src, long long src_stride, float2* dst, long long dst_stride, int n, int n_rx, hipStream_t s);
hipError_t launch_k1_tail(const void* in, long long in_stride_bytes, long long block_bytes, void* hist, int tail_bytes,
                          int n_rx, hipStream_t s);
hipError_t launch_k1x(const K1uParams& p, int npost, int n_rx, hipStream_t s); // channel mode X: npost CIC5 stages down to 48 kHz, us_idx == nullptr: no resampler
hipError_t launch_k1k(const K1kParams& p, int n_rx, hipStream_t s, int hook = 0);
bool k1k_wave_form(const K1kParams& p, int hook); // launch_k1k takes the one-wave form (k1k_wave), which can carry the spectral analysis (K1kParams::fz)
bool k1x_wave_form(const K1uParams& p, int npost); // the same for launch_k1x / k1x_wave
bool k1u96_wave_form(const K1uParams& p, int npost); // ... and for launch_k1u at 96 kSPS (npost = 0, no resampler): k1k_wave<false>
hipError_t launch_convert_rows(const void* in, long long in_stride, int fmt, float2* dst, long long dst_stride, int n, int n_rx, hipStream_t s);
// DownsampleMovingAverage (DSP.cpp:60-82) at an integer ratio m: dst[i] = (((0 + x[m i]) + x[m i + 1]) + ...) / m, n outputs per row
hipError_t launch_ma_rows(const void* in, long long in_stride, int fmt, int m, float2* dst, long long dst_stride, int n, int n_rx, hipStream_t s);
hipError_t launch_k2a_fft(const K2Params& p, int n_chan, hipStream_t s);
hipError_t launch_k2a_search(const K2Params& p, int n_chan, hipStream_t s);
hipError_t launch_k2a_fft_search(const K2Params& p, int n_chan, hipStream_t s); // both in one kernel, no magnitudes in HBM (n_chan even: the two channels of a receiver)
hipError_t launch_selftest_hypot(const float2* in, int n, unsigned* mismatches, hipStream_t s);
hipError_t launch_k2b(const K2Params& p, int n_chan, hipStream_t s); // phasor recurrence
hipError_t launch_k2c(const K2Params& p, int n_chan, hipStream_t s); // history carry + apply
hipError_t launch_k3(const K3Params& p, int n_chan, hipStream_t s);
hipError_t launch_k2b_ck(const K2Params& p, int n_chan, hipStream_t s, int simds); // phasor recurrence, the state at every window start only (simds: how many SIMDs the stream may use)
hipError_t launch_k2b_refine(const K2Params& p, int n_chan, hipStream_t s); // ... and from there the checkpoints inside the windows (in front of launch_k6, any stream)
hipError_t launch_k6(const K6Params& p, hipStream_t s);
struct K5Params { // ModelChallenger FM branch (Model.cpp:638-639): Demod::FM -> Filter(Receiver, 37 taps) -> sign
	const float2* x; long long x_stride; long long x_off; // input rows: sample n of the block at x[chan * x_stride + x_off + n]
	const float2* prev_in; float2* prev_out;   // optional [n_chan]: the sample before the block when the rows carry no history (ModelBase)
	float* fm; long long fm_stride;          // optional (AISGPU_FLAG_TAPS) [n_chan][FM_HIST + L]: the discriminator output at [FM_HIST + n]
	const float* hist_in = nullptr; float* hist_out = nullptr; // [n_chan][FM_HIST] the discriminator's last FM_HIST outputs of the previous block / of this one (both required: launch_k5 refuses a null)
	uint32_t* fmbits; long long fmbits_stride; // [n_chan][L/32] bit n: filtered discriminator > 0
	float taps[37];
	int L;
	float* fir_out; long long fir_stride;     // optional [n_chan][fir_stride]: the filter output itself (AISGPU_FLAG_TAPS)
};

hipError_t launch_k5(const K5Params& p, int n_chan, hipStream_t s);

// ModelEngineV2 (-m 11): what V2::Engine computes from the 48 kHz channel ALONE -- i.e. everything that does not depend on the
// state of its decoders -- for all 512-sample blocks of a batch at once (DSP/Decoder/V2/V2Engine.cpp):
//   * FreqOffset::Estimate (:56-131) of every window that Engine::CGF can ask for without a learned slot phase: the windows
//     at offset 0 and 256 of every block (:312-314), i.e. every 256 samples; f and prominence per window
//   * the two half-block energies midWins compares (:281-291)
//   * FMDemod::Run with atan2_fast (:244-273) and FilterFL37 (:175-188): the sign of the filtered discriminator per sample
// The host engine keeps what closes over its decoders every 512 samples: tone gate, derotation (std::polar of an interpolated
// frequency: libm), FilterFL17, PhaseTracker, BitPLL, the decoders, the slot-phase learner (:293-388).
constexpr int V2_HIST = 512; // samples of the previous block kept in front of the current one (one engine block of look-back)
struct KV2Params {
	const float2* c48; long long c48_stride;   // this block's 48 kHz channels
	float2* hist;                               // [n_chan][V2_HIST] last samples of the previous block (zeros before the stream)
	float2* hist_out;                           // where kv2_carry leaves this block's tail (the same buffer, or the other one of a pair: the engine of
	                                            // this block may still be reading `hist` on its own stream while the next block's assist kernels start)
	uint32_t* fmtail_out;                       // optional [n_chan][16]: the last 512 discriminator signs of this block, for the engine of the next one
	const float2* omega;                        // FFT twiddles
	float* est_f; float* est_prom;              // [n_chan][2 * n_windows]: window w starts at sample -512 + 256 w of this block
	float* energy;                              // [n_chan][n_windows + 1]: sum of |x|^2 over [-512 + 512 i, +256)
	float* disc;                                // [n_chan][FM_HIST + L] discriminator (FM_HIST leading history), as K5: [0, FM_HIST) and the block's last FM_HIST values ...
	bool disc_full;                             // ... all of them (AISGPU_FLAG_TAPS: the discriminator output is a tap only)
	float2* fmprev;                             // [n_chan] FMDemod::prev (0 in front of the first sample: the engine's all-zero look-back block has passed)
	uint32_t* fmbits; long long fmbits_stride;  // [n_chan][L / 32]
	float* fir_out; long long fir_stride;       // optional (taps): the FilterFL37 output itself
	float taps[37];
	int n_windows, L, n_chan;
	int energy_rows;                            // kv2_fm_filter: its first rows of workgroups compute the energies (set by launch_kv2_assist)
};
// ModelEngineV2 with AISGPU_FLAG_GPU_DECODE (round 4): the engine's coherent branch on the device as well -- per channel strictly
// sequential over its 512-sample blocks, like the reference (V2Engine.cpp:293-388): the tone gate / slot lock decide the frequency
// from the decoders' states, Derotate is an accumulated phasor, the five PhaseTrackers take their loop weight from their decoder's
// state sample by sample, the six decoders reset each other.  kv2_engine_roles (round 6, see kernels.hip): three waves per channel.
// kv2_engine: one wave per channel (round 5) -- Derotate and FilterFL17
// block-wise with lanes over time, then six lanes (five tracker + decoder lanes and the FM decoder behind its BitPLL) walk the groups
// of five samples in step, the reference's order inside a group restored only where a message completes.  std::polar of the estimated
// frequency: glibc's sinf / cosf restated (sin_or_cos_ref); frames out like the other engines' device decoders.
struct V2Tracker { unsigned rot; float2 s; int prev_decision; };
struct V2ChanState { // zero-initialised but for rot = (1, 0)
	float2 rot; float last_f; float ppm, ppm_prev; float2 slot_ema; int slot_phase, di; long long sample_idx;
	float pll_phase; int pll_last;
	V2Tracker trk[5];
	float2 carry17[16];
};
struct KV2EParams {
	KV2Params k;               // this block's channels, the previous block's tail, estimates, energies, this block's discriminator signs
	const uint32_t* fm_prev;   // [n_chan][16] the previous block's last 512 discriminator signs (kv2_carry's fmtail_out)
	V2ChanState* st;           // [n_chan]
	DecState* dec;             // [n_chan * 6]
	const float2* slot_cs;     // [1280] (cosf, sinf)(k * (2 pi / 1280)) from the host's libm (learnSlotPhase, :328-337)
	float w_train, w_track;
	uint32_t* frames; unsigned* frame_count; int max_frames; unsigned block, sub;
	int* locked_estimates;     // statistics: Estimate() calls at a learned slot phase (the windows the assist kernels cannot know)
	int roles;                 // 1: kv2_engine_roles -- trackers, FM decoder and the next block's front end on three waves of a workgroup (round 6);
	                           // 2: the same kernel compiled for 168 registers (three waves per SIMD: batches of more than 512 channels);
	                           // 0: kv2_engine -- one wave, six lanes in step (round 5; test hook "v2_roles")
	float taps17[17];
};
#ifdef V2_PROF
void v2_prof_dump(); // experiment build: cycles of kv2_engine's phases on stderr
#endif
bool sincos_restatement_matches_host_libm(); // kv2_engine's sinf / cosf (glibc 2.35, FMA variant) on the host against the host's own libm
hipError_t launch_kv2(const KV2Params& p, hipStream_t s, const KV2EParams* engine = nullptr); // engine: kv2_engine runs before the look-back is overwritten
// the same in three parts (round 6: the engine on a stream of its own, beside the next block's front end and assist kernels)
hipError_t launch_kv2_assist(const KV2Params& p, hipStream_t s, int part = 7); // part 1: estimates, 2: the FM branch, 4: energies
hipError_t launch_kv2_engine(const KV2EParams& e, hipStream_t s);
hipError_t launch_kv2_carry(const KV2Params& p, hipStream_t s);
// Derotation + FIR + ScatterPLL + PhaseSearchEMA in one workgroup (k46_window_search, kernels.hip): K6Params without `sym` traffic
// (f.sym = the block parity's global rows: only the exact fallback inside k46_assemble writes and reads them), K4Params as for launch_k4
struct K46Params { K6Params f; K4Params s; int trips_pad; };
hipError_t launch_k46(K46Params q, hipStream_t st);
hipError_t launch_k4(const K4Params& p, hipStream_t s);          // chunk-parallel + assemble (with the exact sequential search where a speculative warm-up failed)
hipError_t launch_k4_sequential(const K4Params& p, hipStream_t s); // the plain sequential kernel only
hipError_t launch_k4_box(const K4Params& p, hipStream_t s);        // Demod::PhaseSearch (boxcar history), sequential

} // namespace aisk
