/*
 * oracle/ais_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, sequential, float32 restatement of the reference hot path
 * (AIS::ModelDefault / ModelChallenger of jvde-github/AIS-catcher v0.70): the checker the HIP
 * path is compared against.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this; the product (ais-catcher_amd/) never does.
 *
 * Parity status: PINNED -- every stage is checked bit-for-bit against the compiled reference
 * itself (oracle/_ref/libaisref_strict.so, built from /root/reference by oracle/Makefile) in
 * tests/test_oracle_vs_ref.py, and against the committed golden fixtures in tests/golden/
 * (generated from that same compiled reference by tests/golden/make_golden.py).  The reference
 * ships no DSP golden vectors of its own.
 */
#ifndef AIS_ORACLE_H
#define AIS_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ao_chain ao_chain;

/* model: 11 = ModelEngineV2 (V2::Engine behind the front end, oracle/ais_oracle_v2.inc), 0 = ModelStandard (FM receiver, five decoders on the deinterleaved discriminator), 1 = ModelBase (FM receiver, SimplePLL with decoder feedback), 2 = ModelDefault, 4 = ModelChallenger.  fmt: 0 = CU8, 1 = CF32, 2 = CS8, 3 = CS16.
 * flags: bit 0 record taps, bit 1 `-go DSK on` (decimate-by-3 ladders for 576k/1152k/2304k), bit 2 `-go PS_EMA off`
 * (PhaseSearch with a boxcar history instead of PhaseSearchEMA), bit 3 `-go FP_DS on` (fixed-point ladder Downsample16_CU8 at
 * 1536 kSPS, CU8 input), bit 4 channel mode X, bit 5 `-go MA on`, bit 8 `-go AFC_WIDE off`, bit 9 `-go DROOP off`. */
ao_chain* ao_create(int model, int sample_rate, int fmt, int flags);
void ao_destroy(ao_chain*);
/* one call == one reference Receive() block (call boundaries are part of the numerical contract:
 * Rotate renormalises once per call, Source/DSP/DSP.cpp:315) */
int ao_feed(ao_chain*, const void* data, int nbytes);
int ao_msg_count(ao_chain*);
int ao_nmea(ao_chain*, char* dst, int cap);
int ao_msg_meta(ao_chain*, float* level, float* ppm, int cap);
/* taps: 0/1 = 48 kHz front-end output A/B, 2/3 = CGF out, 4/5 = FIR-17 out (complex, interleaved) */
long long ao_tap(ao_chain*, int which, float* dst, long long cap);
long long ao_tap_ppm(ao_chain*, int which, float* dst, long long cap);
/* switch the recording of taps / decisions on or off from the next ao_feed() on (the chain must have been created with taps) */
void ao_set_taps(ao_chain*, int on);
/* real taps of the FM receivers: 6/7 = Demod::FM output A/B, 8/9 = Filter(Receiver, 37 taps) output A/B, one float per 48 kHz sample */
long long ao_tapf(ao_chain*, int which, float* dst, long long cap);
long long ao_bits(ao_chain*, int ch, int j, int fm, float* bits, float* lvl, long long* idx, long long cap);
void ao_reset_seq(void);

/* stand-alone stage functions (used by unit tests and to check individual HIP kernels) */
void ao_cic5_decimate(const float* x, int n, float* state10, float* y);          /* n even, y: n/2 */
void ao_cic5_filter(const float* x, int n, float* state10, float* y);
void ao_fdc(const float* x, int n, float alpha, float* state4, float* y);
void ao_rotate(const float* x, int n, float* rot2, const float* mult2, float* up, float* down);
void ao_rotate_mult(float* mult2);
void ao_fir_complex(const float* x, int n, const float* taps, int nt, float* hist, float* y);
float ao_hypotf(float a, float b);

#ifdef __cplusplus
}
#endif
#endif
