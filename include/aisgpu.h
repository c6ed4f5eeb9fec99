/*
 * include/aisgpu.h -- C ABI of libaisgpu.so: the MI355X (gfx950) AIS GMSK demodulation chain.
 *
 * This is the drop-in boundary for the hot path behind the reference's AIS::ModelDefault
 * (jvde-github/AIS-catcher v0.70).  Every entry point is plain C (pointers + sizes, int status,
 * never throws) so the reference can bind it from its own C++ or any FFI.
 * File:line citations are relative to the reference's Source/ directory.
 *
 * What one context replaces, per receiver (the blocks between Util::ConvertRAW and the
 * AIS::Decoder objects in DSP/Model.cpp:27-356 + :520-577):
 *   ConvertRAW (CU8 -> CFLOAT32)          Utilities/StreamHelpers.cpp:51-133, Utilities/Convert.cpp:255-264
 *   4x Downsample2CIC5 (1536k -> 96k)     DSP/DSP.cpp:93-117
 *   FilterComplex3Tap (droop comp.)       DSP/DSP.cpp:283-293
 *   Rotate (+-25 kHz channel split)       DSP/DSP.cpp:296-316
 *   Downsample2CIC5 + FilterCIC5 (x2)     DSP/DSP.cpp:93-157
 *   SquareFreqOffsetCorrection (FFT-512)  DSP/DSP.cpp:417-489, DSP/FFT.h:36-131
 *   FilterComplex (Filters::Coherent)     DSP/DSP.cpp:215-246
 *   ScatterPLL (/5, signal level)         DSP/DSP.h:76-117
 *   5x PhaseSearchEMA per channel         DSP/Demod.cpp:39-101
 * and for AIS::ModelChallenger additionally (DSP/Model.cpp:630-639)
 *   Demod::FM + Filter(Receiver, 37 taps) DSP/Demod.cpp:27-37, DSP/DSP.cpp:249-280
 * The consumer -- 5x AIS::Decoder per channel + NMEA (Marine/AIS.h:82-181, Marine/AIS.cpp:33-142,
 * Marine/Message.cpp:569-631) -- stays on the host, unchanged, fed from aisgpu_fetch() (ais-catcher_amd/host/).
 *
 * A context batches n_receivers independent dual-channel receivers; one aisgpu_run() consumes
 * exactly one reference Receive() block (block_len IQ samples) of every receiver.  Block
 * boundaries are part of the numerical contract (Rotate renormalises once per call,
 * DSP/DSP.cpp:315), so results equal the reference driven with the same block size.
 */
#ifndef AISGPU_H
#define AISGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AISGPU_OK 0
#define AISGPU_ERR_ARG 1      /* invalid argument / unsupported configuration */
#define AISGPU_ERR_NODEV 2    /* no usable HIP device */
#define AISGPU_ERR_HIP 3      /* a HIP runtime call failed (see aisgpu_last_error) */
#define AISGPU_ERR_STATE 4    /* call sequence error (e.g. fetch before run) */
#define AISGPU_ERR_OVERFLOW 5 /* an output buffer supplied by the caller is too small */

/* input formats: the RAW formats the reference's file/RTL devices deliver (Library/Common.h:88-99) */
#define AISGPU_FMT_CU8 0
#define AISGPU_FMT_CF32 1
#define AISGPU_FMT_CS8 2   /* Utilities/Convert.cpp:266-275: (int8) / 128.0f */
#define AISGPU_FMT_CS16 3  /* Utilities/Convert.cpp:277-286: (int16) / 32768.0f */

/* models (DSP/Model.h:61-72) */
#define AISGPU_MODEL_STANDARD 0    /* AIS::ModelStandard   (-m 0), DSP/Model.cpp:484-518: FM receiver with five decoders on the deinterleaved
                                    * discriminator; the device path and the outputs are those of AISGPU_MODEL_BASE */
#define AISGPU_MODEL_BASE 1        /* AIS::ModelBase       (-m 1), DSP/Model.cpp:419-438: FM receiver; the GPU delivers the sign of the
                                    * filtered discriminator per 48 kHz sample; SimplePLL + decoder (feedback loop) run on the host, or -- with
                                    * AISGPU_FLAG_GPU_DECODE -- chunk-parallel on the device (aisgpu_frames) */
#define AISGPU_MODEL_DEFAULT 2     /* AIS::ModelDefault    (-m 2), DSP/Model.cpp:520-577 */
#define AISGPU_MODEL_V2 11         /* AIS::ModelEngineV2  (-m 11), DSP/Model.cpp:440-463.  Two forms:
                                    * - with AISGPU_FLAG_GPU_DECODE the whole V2::Engine of every channel runs ON THE DEVICE (kv2_engine_roles, three waves
                                    *   per channel; kv2_engine, one wave per channel, is round 5's form and a test hook: DSP/Decoder/V2/
                                    *   V2Engine.cpp:293-388, results in the reference's order per channel) and only frames come back (aisgpu_frames);
                                    * - without it the device runs the front end and what the engine computes from the channel alone (frequency estimates,
                                    *   energies, FM branch: aisgpu_out.v2_*, fm_bits) and hands over the two 48 kHz channels (aisgpu_out.c48) to an engine
                                    *   on the host (ais-catcher_amd/host/v2_engine.*), whose coherent branch closes over the state of its own decoders
                                    *   every sample (V2Engine.cpp:300-388). */
#define AISGPU_MODEL_CHALLENGER 4  /* AIS::ModelChallenger (-m 4), DSP/Model.cpp:601-678: ModelDefault + the FM branch */

#define AISGPU_FLAG_TAPS 1    /* keep intermediate float taps readable via aisgpu_tap() (tests) */
#define AISGPU_FLAG_SERIAL 2  /* profiling aid: one stream, no overlap between the kernels of consecutive blocks */
#define AISGPU_FLAG_PS_BOXCAR 8 /* KEY_SETTING_PS_EMA off: Demod::PhaseSearch (boxcar history) instead of PhaseSearchEMA (Model.cpp:550-555) */
#define AISGPU_FLAG_GPU_DECODE 16 /* run the AIS::Decoder objects (frame decoder + Reset mesh) on the device too: aisgpu_frames().  ModelDefault (five per
                                   * channel), ModelStandard (five on the deinterleaved discriminator), ModelChallenger (ten: coherent + FM, Model.cpp:641-674)
                                   * and ModelBase (DSP::SimplePLL + one decoder with its feedback loop, DSP.cpp:28-57, Model.cpp:428-435).
                                   * ModelEngineV2 (round 4): the whole V2::Engine per channel -- tone gate / slot lock, Derotate, FilterFL17, five
                                   * PhaseTrackers, BitPLL, six decoders, slot-phase learner (V2Engine.cpp:293-388) -- runs on the device, strictly in the
                                   * reference's order per channel; nothing but frames goes to the host: aisgpu_out.c48, v2_f, v2_prom, v2_energy and fm_bits are
                                   * NULL (no pinned host slots are allocated for them), frames come back (NMEA text, tag.ppm and tag.level equal the
                                   * reference's bit for bit; std::polar = glibc 2.35's sinf / cosf restated in the FMA variant its ifunc selects on x86-64
                                   * hosts with FMA: aisgpu_create() checks the restatement against the host's own sinf / cosf and refuses the flag for
                                   * this model with AISGPU_ERR_ARG where they differ -- see INTEGRATION.md.  That check is a SAMPLE: every float of a
                                   * 2^-20 grid over [-1.7, 1.7] plus a geometric sweep towards 0, 3.6 M of the ~10^9 floats FreqOffset::Derotate's argument
                                   * can take (Estimate interpolates between bins, V2Engine.cpp:113-128, so the set is not finite in practice).  It
                                   * recognises a libm whose algorithm or FMA variant differs; it cannot prove equality on every argument) */
#define AISGPU_FLAG_FP_DS 32  /* KEY_SETTING_FP_DS (`-go FP_DS on`, `-F`): 1536 kSPS CU8 input goes through the fixed-point ladder
                               * Downsample16_CU8 (DSP/DSP.cpp:499-651, Model.cpp:231-237); ignored at other rates like in the reference */
#define AISGPU_FLAG_MODE_X 64 /* channel mode X (`-c X`, Receiver.cpp:87-98, Model.cpp:35-107): ONE channel, already centred, sample_rate
                              * 12000 .. 192000 like the reference (buckets 48k / 96k / 192k, resampled in between; below 24000 one input
                              * block completes up to four downstream blocks, aisgpu_out_count()).  The receiver's channel is channel 0 of aisgpu_fetch();
                              * channel 1 does not exist (round 6: the receivers of a batch are packed, one chain per receiver) -- aisgpu_fetch(rx, 1)
                              * hands out what a chain fed with silence puts out (no decision set, level 0), frames carry ch = 0 */
#define AISGPU_FLAG_DSK 4     /* KEY_SETTING_DSK (`-go DSK on`): 576k / 1152k / 2304k use the decimate-by-3 ladder (Model.cpp:130) */
#define AISGPU_FLAG_MA_DS 128 /* KEY_SETTING_MA (`-go MA on`, Model.cpp:122-126): convert >> DownsampleMovingAverage >> Rotate instead of the CIC5 ladder
                               * -- integrate-and-dump to 96 kHz (DSP.cpp:60-82), handed on in blocks of 8192 samples.  Here for sample rates
                               * that are a multiple of 96000 (192k .. 12288k: 2 .. 128 samples per output) and input blocks that are a whole
                               * number of its output blocks (block_len * 96000 / sample_rate a multiple of 8192, as the reference's file blocks
                               * are); like the reference, the option overrides FP_DS and DSK.  Not with channel mode X. */

typedef struct aisgpu aisgpu_t;

typedef struct aisgpu_cfg {
	int sample_rate;   /* any rate from 96000 to 12288000 (ModelFrontend::buildModel, Model.cpp:109-338): the buckets 96000*2^k
	                    * (k = 0..7), 288000 and -- with AISGPU_FLAG_DSK -- 576000 / 1152000 / 2304000 (DownsampleKFilter ladders),
	                    * and every rate in between, which the reference resamples (Upsample) into the smallest bucket above it,
	                    * e.g. 250000, 2400000, 6000000, 10000000.  On a DownsampleKFilter bucket block_len must be a multiple of
	                    * 24576 * bucket/288000 (whole 8192-sample output blocks of the filter) */
	int n_receivers;   /* independent dual-channel receiver instances batched on this GPU */
	int block_len;     /* IQ samples per receiver per Receive() block; multiple of 512 * bucket_rate/48000 */
	int model;         /* AISGPU_MODEL_STANDARD, _BASE, _DEFAULT, _CHALLENGER or _V2 */
	int input_format;  /* AISGPU_FMT_* */
	int afc_wide;      /* KEY_SETTING_AFC_WIDE (default on, Model.cpp:536-540) */
	int droop;         /* KEY_SETTING_DROOP    (default on, Model.cpp:223-229) */
	int device_id;     /* HIP device ordinal */
	int flags;         /* AISGPU_FLAG_* */
	int tiles_per_span;/* front-end time tiling (0 = auto) */
} aisgpu_cfg;

/* per (receiver, channel) result of one block: what ScatterPLL/PhaseSearchEMA hand to the
 * five AIS::Decoder objects (DSP/DSP.h:95-117, Demod.cpp:96-99), in host memory owned by the
 * context and valid until the next aisgpu_sync_outputs() -- so a caller may hand in and start the NEXT block
 * (aisgpu_submit x R, aisgpu_run) while it still consumes these (pipelined hand-off, GpuBatch::setPipelined). */
typedef struct aisgpu_out {
	int n_groups;            /* complete 5-sample groups emitted in this block */
	long long first_group;   /* stream index of the first group; tag.sample_idx of phase j = 5*(first_group+g)+j */
	const uint32_t* bits[5]; /* phase j: bit g of word g/32 -> PhaseSearchEMA output (+1.0f if set, else -1.0f) */
	const float* lvl;        /* [n_groups] tag.sample_lvl */
	int n_windows;           /* CGF windows (512 samples @48k) completed in this block */
	const float* ppm;        /* [n_windows] tag.ppm */
	const int* group_window; /* reserved (NULL): window of group g is (5*(first_group+g)+4 - first_sample48)/512 */
	long long first_sample48;/* stream index (48 kHz) of the first sample of this block */
	const uint32_t* fm_bits; /* AISGPU_MODEL_CHALLENGER and AISGPU_MODEL_BASE (else NULL; with MODEL_BASE n_groups is 0 and only this is valid): bit n of word n/32 set <=> the filtered FM discriminator
	                          * sample first_sample48 + n is > 0 (what Deinterleave S_af hands to DEC_af[n % 5], Model.cpp:638-639) */
	const float* c48;        /* AISGPU_MODEL_V2 without AISGPU_FLAG_GPU_DECODE only (else NULL): the channel's 48 kHz front-end output of this block (FCIC5_a/b.out, Model.cpp:345-346),
	                          * 512 * n_windows complex samples, interleaved re/im; n_groups is 0 */
	/* AISGPU_MODEL_V2 without AISGPU_FLAG_GPU_DECODE only (else NULL, and fm_bits too): what V2::Engine computes from the channel alone, for every 512-sample engine block of
	 * this block at once (DSP/Decoder/V2/V2Engine.cpp).  Engine block i (i = 0 .. n_windows-1) is the one the engine decodes
	 * when block-relative samples [512 i, 512 i + 512) arrive as its look-ahead, i.e. samples [512 i - 512, 512 i):
	 *   v2_f[2 i], v2_prom[2 i]          FreqOffset::Estimate (:56-131) of the window at offset 0 of that engine block (f, prominence)
	 *   v2_f[2 i + 1], v2_prom[2 i + 1]  ... at offset 256 (Engine::CGF's "mid" window, :312-314)
	 *   v2_energy[i], v2_energy[i + 1]   midWins' head / tail sums (:281-291)
	 *   fm_bits (above)                  bit n: FilterFL37(FMDemod(x))[n] > 0 for block-relative sample n (:244-273, :175-188)
	 * The windows an engine asks for once it has learned a slot phase (:300-309) are not among them: it computes those itself. */
	const float* v2_f;
	const float* v2_prom;
	const float* v2_energy;  /* n_windows + 1 values */
} aisgpu_out;

void aisgpu_default_cfg(aisgpu_cfg* cfg);
int aisgpu_create(const aisgpu_cfg* cfg, aisgpu_t** out);
void aisgpu_destroy(aisgpu_t* h);

/* Copy one receiver's block from host memory (borrowed for the call only, like the reference's
 * Receive(const T*, int, TAG&), Library/Stream.h:36-45) into the staging buffer.  n_iq must equal
 * block_len (AISGPU_ERR_ARG otherwise): the context works on blocks of ONE size, because a block is one Receive() call of the
 * reference's chain and call boundaries are part of the arithmetic (Rotate renormalises once per call, DSP/DSP.cpp:315).  A caller
 * whose device hands over other sizes -- the reference's RAWFile sends one OR two FIFO blocks per call, Device/FileRAW.cpp:120-136
 * -- cuts them into block_len pieces and calls submit / run once per piece; the reference-side binding does exactly that
 * (GpuChain::Receive, integration/reference/Source/DSP/GPU/ModelGPU.cpp, tested behind the reference's real file reader).
 * Data is CU8 / CS8 pairs, CS16 pairs or CFLOAT32 per cfg.input_format.  Thread safe for different rx (receiver threads copy
 * their rows concurrently); the staging buffers are double buffered, so the rows of block f+1 may be submitted while
 * block f is still running. */
int aisgpu_submit(aisgpu_t* h, int rx, const void* iq, int n_iq);

/* Zero-copy alternative: the whole batch is already resident in device memory as
 * [n_receivers][rx_stride_samples] samples of the configured format (rx_stride_samples >= block_len). */
int aisgpu_submit_device(aisgpu_t* h, const void* iq_dev, long long rx_stride_samples);

/* Enqueue the whole chain for the submitted block on the context's stream (asynchronous). */
int aisgpu_run(aisgpu_t* h);
/* Enqueue the device->host copy of the outputs of the last aisgpu_run() and wait for them. */
int aisgpu_sync_outputs(aisgpu_t* h);
/* Wait for the stream without copying outputs (throughput runs). */
int aisgpu_sync(aisgpu_t* h);
int aisgpu_fetch(aisgpu_t* h, int rx, int ch, aisgpu_out* out); /* == aisgpu_fetch_sub(h, 0, ...) */
/* Sample rates between two 2^k buckets go through the reference's fractional resampler (DSP/DSP.cpp:192-212), which
 * hands fixed-size blocks downstream whenever one is full: one input block then completes 1 to 4 downstream blocks
 * (each is one Receive() call of everything behind the resampler in the reference; 1 or 2 at the dual-channel rates, up to 4 in
 * channel mode X below 24 kSPS).  aisgpu_out_count() tells how many the last aisgpu_run() completed (always 1 for the 2^k
 * rates); fetch them in order with aisgpu_fetch_sub(). */
int aisgpu_out_count(aisgpu_t* h);

/* AISGPU_FLAG_GPU_DECODE: the frames the ten AIS::Decoder objects of every receiver completed with a good CRC since the
 * previous aisgpu_sync_outputs() -- i.e. during the last aisgpu_run() when every run is followed by one (replaces feeding aisgpu_out's decisions to AIS::Decoder::Receive, Marine/AIS.h:82-181).  What is left
 * for the caller is AIS::Decoder::processData's tail (Marine/AIS.cpp:66-96): tag.level = level_sum / position (and its dB
 * conversion), Message::validate, buildNMEA.  Sorted the way the reference emits: by receiver, then downstream block,
 * channel A before channel B, group, phase.  Valid after aisgpu_sync_outputs() until the next aisgpu_sync_outputs() (the array is
 * host memory of the context that only that call rewrites: a pipelined caller reads it while the next aisgpu_run() is in flight). */
typedef struct aisgpu_frame {
	int rx, ch, phase, sub;   /* receiver, channel 0/1, decoder DEC_x[phase] (5..9: ModelChallenger's FM decoders DEC_xf[phase - 5]; ModelEngineV2: 0..4 the
	                           * decoders behind the phase trackers, 5 the FM decoder), downstream block of this run */
	int group;                /* group (symbol index of the phase chain) inside that block whose bit completed the closing flag (ModelBase: the 48 kHz sample;
	                           * ModelEngineV2: the bits of tag.ppm at that moment, a float -- the engine switches frequencies inside its blocks) */
	int position;             /* decoder bit position at that moment; the frame incl. its FCS is position - 7 bits long */
	float level_sum;          /* sum of tag.sample_lvl over the frame's bits */
	long long start_idx, end_idx; /* tag.sample_idx at the start flag / at the last bit */
	unsigned char data[144];  /* bits as received (Message::setBit order: bit i -> byte i/8, bit i%8) */
} aisgpu_frame;
int aisgpu_frames(aisgpu_t* h, const aisgpu_frame** frames, int* count);

/* Statistics of the chunk-parallel PhaseSearchEMA (reference DSP/Demod.cpp:39-101): the number of workgroups (four chains each) whose
 * speculative warm-up did not reproduce the sequential EMA bit for bit and that therefore went through the exact sequential kernel,
 * summed over the blocks completed so far (call it behind aisgpu_sync_outputs()).  Results are bit-exact either way; the count says
 * how often the slow path ran (an extreme level step of a receiver; option "ps_warm" sets the warm-up length, default 256 symbols). */
int aisgpu_ps_fallbacks(aisgpu_t* h, long long* count);

/* AISGPU_FLAG_GPU_DECODE, event-driven decoder kernels (ModelDefault / ModelStandard / ModelChallenger): the number of blocks that went
 * through the sequential decoder kernel instead, because some decoder had more candidate frame starts in the block than the
 * kernels' lists hold (128 frames / 1024 candidates per decoder and block: a carrier that repeats preamble + start flag every
 * few dozen symbols).  The frames are the same either way; the sequential kernel is ~10x slower.
 * ModelBase (chunk-parallel sampler + decoder kernels): the number of (channel, block) pairs that went through the sequential kernel
 * because a chunk or a boundary task completed more than four frames. */
/* (ModelEngineV2 with AISGPU_FLAG_GPU_DECODE: the number of FreqOffset::Estimate() calls the engine kernel made at a learned slot phase --
 * windows at an arbitrary offset, which the assist kernels cannot compute ahead: the kernel's slow path, exact like the rest.) */
int aisgpu_decoder_fallbacks(aisgpu_t* h, long long* count);
int aisgpu_fetch_sub(aisgpu_t* h, int sub, int rx, int ch, aisgpu_out* out);

/* Float taps of the last block (tests only; needs AISGPU_FLAG_TAPS):
 *   which 0/1: 48 kHz front-end output A/B (== FCIC5_a/b.out), 2/3: CGF output, 4/5: FIR-17 output.
 * Copies up to cap complex samples (interleaved re,im) to dst; returns the number available or <0.
 *   which 6/7: Demod::FM output A/B (FM_a/b.out, DSP/Demod.cpp:27-37), 8/9: Filter(Receiver, 37 taps) output A/B (FR_a/b.out,
 *   DSP/Model.cpp:431-432,638-639) of the FM receivers (AISGPU_MODEL_CHALLENGER / _BASE / _STANDARD): REAL samples, one float
 *   per 48 kHz sample; cap and the return value count floats. */
long long aisgpu_tap(aisgpu_t* h, int which, int rx, float* dst, long long cap);

/* raw HIP stream / timing hooks for the benchmark */
void* aisgpu_stream(aisgpu_t* h);
/* average device time (ms) of the front-end kernel over the launches since the last reset,
 * measured with HIP events on the context's stream; *launches receives the count */
float aisgpu_frontend_ms(aisgpu_t* h, int* launches);
void aisgpu_timing(aisgpu_t* h, int enable);

/* Test hooks: process-wide options read by aisgpu_create() (value NULL or "" removes one).  None of them changes a result; they
 * select an alternative -- equally exact -- code path so that the tests can reach it.  The release library reads no environment
 * variable (a -DAISGPU_EXPERIMENTS build also accepts AISGPU_<KEY>).  Keys:
 *   "serial"        1: like AISGPU_FLAG_SERIAL
 *   "ps_warm"       warm-up length of the speculative PhaseSearchEMA chunks in symbols (small values force the exact fallback)
 *   "ps_sequential" 1: the sequential PhaseSearch row kernels only
 *   "k7"            "seq": symbol-by-symbol device decoders; "alt": event-driven and sequential kernels alternate block by block
 *   "k7b_fcap"      1 .. 4: frames a list of ModelBase's chunk-parallel decoder kernels takes (small values force the exact fallback, k7_base)
 *   "fused"         0: the materialised back end (phasor / derotated-sample arrays in HBM: what AISGPU_FLAG_TAPS uses)
 *   "fft_in_k1"     0: the spectral analysis as FFT + search kernels instead of inside the front-end waves
 *   "k46"           1: derotation / FIR and PhaseSearchEMA in one workgroup that keeps the FIR outputs in LDS (k46_window_search, round 5:
 *                   exact, 0.24 GB less traffic per step, measured 5 % slower) instead of k6_window_fir + k4_phase_chunks
 *   "k1u_spw"       2 / 4 / 8: the resampler front end (k1u_resample_frontend) walks that many consecutive spans per workgroup whatever the
 *                   batch size (by default only batches of ~200 receivers and more get walks longer than one span).  On the ladders whose
 *                   front end is a one-wave kernel since round 6 (the decimate-by-3 tail k1k_wave -- also behind Upsample and, without the filter, for dual-channel 96 kSPS --, channel mode X at 48 / 96 / 192 kSPS k1x_wave):
 *                   4 / 8 = spans of that many tiles whatever the batch size, 2 = the workgroup form of rounds 2-5 + the FFT / search kernel
 *   "us_k1"         0: the tail of the resampled ladders (buckets from 384k up) as k1u_resample_frontend + the FFT / search kernel, the form
 *                   of rounds 3-5, instead of one-wave workgroups of the front-end kernel (k1_dpp<2, 5, false>, round 6)
 *   "v2_roles"      0 / 1 / 2: ModelEngineV2 on the device as kv2_engine (one wave per channel, round 5) / kv2_engine_roles (three waves per
 *                   channel, round 6) / the same compiled for three waves per SIMD, whatever the batch size (default: 1 up to 512 channels, else 2)
 * Returns AISGPU_ERR_ARG for an unknown key. */
int aisgpu_set_option(const char* key, const char* value);

const char* aisgpu_strerror(int code);
const char* aisgpu_last_error(aisgpu_t* h);
int aisgpu_device_count(void);

/* Device self-tests of arithmetic identities the kernels rely on (no reference counterpart; used by tests/).
 * which = 0: n pairs of floats (x, y) in `in`; returns the number of pairs on which the FFT-bin magnitude routine
 * differs from the glibc-equivalent hypotf restatement, or a negative AISGPU_ERR_* code. */
long long aisgpu_selftest(int device_id, int which, const void* in, long long n);

#ifdef __cplusplus
}
#endif
#endif /* AISGPU_H */
