/*
 * oracle/ais_oracle.c -- TEST INFRASTRUCTURE ONLY (see ais_oracle.h).
 *
 * Sequential float32 restatement of the reference's ModelDefault / ModelChallenger hot path.
 * Compile with: gcc -std=c11 -O2 -fno-fast-math -ffp-contract=off   (no FMA contraction, no
 * re-association: every expression below is evaluated exactly as written, in IEEE binary32).
 * All file:line citations are relative to /root/reference/Source/.
 */
#include "ais_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float re, im; } cf;

static const float PI_F = 3.14159265358979323846f; /* Library/Common.h:318 (a float constant) */

static inline cf cadd(cf a, cf b) { cf r = { a.re + b.re, a.im + b.im }; return r; }
static inline cf csub(cf a, cf b) { cf r = { a.re - b.re, a.im - b.im }; return r; }
static inline cf cscale(float t, cf a) { cf r = { t * a.re, t * a.im }; return r; }
/* std::complex<float> operator* as g++ -fno-fast-math evaluates it for finite operands */
static inline cf cmul(cf a, cf b) {
	cf r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re };
	return r;
}
/* std::abs(std::complex<float>) == cabsf == hypotf (glibc) */
float ao_hypotf(float a, float b) { return hypotf(a, b); }

/* ---------------------------------------------------------------- growable buffers */
typedef struct { float* p; long long n, cap; } fvec;
static void fv_push(fvec* v, const float* src, long long n) {
	if (v->n + n > v->cap) {
		long long c = v->cap ? v->cap : 4096;
		while (c < v->n + n) c *= 2;
		v->p = (float*)realloc(v->p, sizeof(float) * (size_t)c);
		v->cap = c;
	}
	memcpy(v->p + v->n, src, sizeof(float) * (size_t)n);
	v->n += n;
}
typedef struct { long long* p; long long n, cap; } lvec;
static void lv_push1(lvec* v, long long x) {
	if (v->n + 1 > v->cap) {
		v->cap = v->cap ? v->cap * 2 : 4096;
		v->p = (long long*)realloc(v->p, sizeof(long long) * (size_t)v->cap);
	}
	v->p[v->n++] = x;
}

/* ---------------------------------------------------------------- a2/a5: CIC5 (DSP/DSP.cpp:85-157)
 * Five cascaded 2-tap sums; s_k[n] = s_{k-1}[n] + s_{k-1}[n-1]; output s_4 * 2^-5.
 * state: h[0..4] = s_{k-1}[n-1] for the next even sample (complex, 10 floats).              */
typedef struct { cf h[5]; } cic5_t;

static void cic5_run(cic5_t* s, const cf* x, int n, cf* y, int decimate) {
	cf r[5];
	for (int i = 0, j = 0; i < n; i += 2) {
		cf z = x[i];
		for (int k = 0; k < 5; k++) { r[k] = z; z = cadd(z, s->h[k]); }
		cf o = { z.re * 0.03125f, z.im * 0.03125f };
		if (decimate) y[j++] = o; else y[i] = o;
		z = x[i + 1];
		for (int k = 0; k < 5; k++) { s->h[k] = z; z = cadd(z, r[k]); }
		if (!decimate) { cf o2 = { z.re * 0.03125f, z.im * 0.03125f }; y[i + 1] = o2; }
	}
}
void ao_cic5_decimate(const float* x, int n, float* st, float* y) { cic5_run((cic5_t*)st, (const cf*)x, n, (cf*)y, 1); }
void ao_cic5_filter(const float* x, int n, float* st, float* y) { cic5_run((cic5_t*)st, (const cf*)x, n, (cf*)y, 0); }

/* ---------------------------------------------------------------- a3: FDC (DSP/DSP.cpp:283-293, DSP.h:293-297) */
typedef struct { cf h1, h2; } fdc_t;
static void fdc_run(fdc_t* s, float alpha, const cf* x, int n, cf* y) {
	float beta = 1 - 2 * alpha;
	for (int i = 0; i < n; i++) {
		cf a = cscale(alpha, cadd(s->h1, x[i]));
		cf b = { s->h2.re * beta, s->h2.im * beta };
		y[i] = cadd(a, b);
		s->h1 = s->h2;
		s->h2 = x[i];
	}
}
void ao_fdc(const float* x, int n, float alpha, float* st, float* y) { fdc_run((fdc_t*)st, alpha, (const cf*)x, n, (cf*)y); }

/* ---------------------------------------------------------------- a4: Rotate (DSP/DSP.cpp:296-316, Model.cpp:31) */
void ao_rotate_mult(float* m) {
	float angle = (float)((double)PI_F * 25000.0 / 48000.0); /* float*double literals -> double, then cast */
	m[0] = cosf(angle); /* std::polar(1.0f, angle) */
	m[1] = sinf(angle);
}
static void rotate_run(cf* rot, cf mult, const cf* x, int n, cf* up, cf* down) {
	cf r = *rot;
	for (int i = 0; i < n; i++) {
		float RR = x[i].re * r.re, II = x[i].im * r.im;
		float RI = x[i].re * r.im, IR = x[i].im * r.re;
		up[i].re = RR - II;   up[i].im = IR + RI;
		down[i].re = RR + II; down[i].im = IR - RI;
		r = cmul(r, mult);
	}
	float a = hypotf(r.re, r.im); /* once per call */
	r.re /= a; r.im /= a;
	*rot = r;
}
void ao_rotate(const float* x, int n, float* rot, const float* mult, float* up, float* down) {
	cf m = { mult[0], mult[1] };
	rotate_run((cf*)rot, m, (const cf*)x, n, (cf*)up, (cf*)down);
}

/* ---------------------------------------------------------------- a16: Upsample (DSP/DSP.cpp:192-212, DSP.h:172-176) */
typedef struct { float alpha, increment; cf a; cf* out; int idx_out, cap; } ups_t;

/* ---------------------------------------------------------------- a7/a12: FIR (DSP/DSP.cpp:215-280, DSP.h:224-230) */
static const float TAPS_COHERENT[17] = { /* DSP/Filters.h:35-41 */
	2.06995719e-06f, 3.18610148e-05f, 3.40605309e-04f, 2.52892989e-03f, 1.30411453e-02f, 4.67076746e-02f,
	1.16186141e-01f, 2.00730781e-01f, 2.40861391e-01f, 2.00730781e-01f, 1.16186141e-01f, 4.67076746e-02f,
	1.30411453e-02f, 2.52892989e-03f, 3.40605309e-04f, 3.18610148e-05f, 2.06995719e-06f };
static const float TAPS_RECEIVER[37] = { /* DSP/Filters.h:24-33 */
	0.00119025f, -0.00148464f, -0.00282428f, -0.00200561f, -0.00068852f, 0.00343044f, 0.00902093f, 0.01367867f,
	0.01147965f, 0.0027259f, -0.01766614f, -0.04244429f, -0.0577468f, -0.05245161f, -0.01072754f, 0.0732564f,
	0.17643278f, 0.25582214f, 0.28200453f, 0.25582214f, 0.17643278f, 0.0732564f, -0.01072754f, -0.05245161f,
	-0.0577468f, -0.04244429f, -0.01766614f, 0.0027259f, 0.01147965f, 0.01367867f, 0.00902093f, 0.00343044f,
	-0.00068852f, -0.00200561f, -0.00282428f, -0.00148464f, 0.00119025f };

/* y[j] = sum_{i=0}^{nt-1} taps[i]*x[j-(nt-1)+i], accumulated left to right from 0 (hist = last nt-1 inputs) */
static void fir_c_run(cf* hist, const float* taps, int nt, const cf* x, int n, cf* y) {
	cf w[64];
	for (int j = 0; j < n; j++) {
		for (int i = 0; i < nt - 1; i++) w[i] = hist[i];
		w[nt - 1] = x[j];
		cf acc = { 0.0f, 0.0f };
		for (int i = 0; i < nt; i++) acc = cadd(acc, cscale(taps[i], w[i]));
		y[j] = acc;
		for (int i = 0; i < nt - 1; i++) hist[i] = w[i + 1];
	}
}
void ao_fir_complex(const float* x, int n, const float* taps, int nt, float* hist, float* y) {
	fir_c_run((cf*)hist, taps, nt, (const cf*)x, n, (cf*)y);
}
static float fir_r_step(float* hist, const float* taps, int nt, float x) {
	float w[64];
	for (int i = 0; i < nt - 1; i++) w[i] = hist[i];
	w[nt - 1] = x;
	float acc = 0.0f;
	for (int i = 0; i < nt; i++) acc += taps[i] * w[i];
	for (int i = 0; i < nt - 1; i++) hist[i] = w[i + 1];
	return acc;
}

/* ---------------------------------------------------------------- a6: CGF (DSP/DSP.cpp:417-489, DSP/FFT.h:36-131) */
#define CGF_N 512
#define CGF_LOGN 9
static cf g_omega[CGF_N];
static int g_rev[CGF_N];
static int g_tables_ready = 0;

static int bitrev(int x, int logn) { /* FFT.h:36-66: plain bit reversal of the low logn bits */
	int y = 0;
	for (int i = 0; i < logn; i++) { y = (y << 1) | (x & 1); x >>= 1; }
	return y;
}
static void tables_init(void) {
	if (g_tables_ready) return;
	for (int s = 0; s < CGF_N; s++) {
		/* FFT.h:83: std::polar(T(1), T(-2.0 * PI) * T(s) / T(N)) */
		float th = ((float)(-2.0 * (double)PI_F) * (float)s) / (float)CGF_N;
		g_omega[s].re = cosf(th);
		g_omega[s].im = sinf(th);
	}
	for (int i = 0; i < CGF_N; i++) g_rev[i] = bitrev(i, CGF_LOGN);
	g_tables_ready = 1;
}
static void fft512(cf* x) { /* FFT.h:94-129: in-place radix-2 DIT on bit-reversed input */
	int m = 2, m2 = 1, r = CGF_N;
	for (int s = 0; s < CGF_LOGN; s++) {
		int w = 0;
		r >>= 1;
		for (int j = 0; j < m2; j++) {
			cf o = g_omega[w];
			for (int k = 0; k < CGF_N; k += m) {
				cf t = cmul(o, x[k + j + m2]);
				x[k + j + m2] = csub(x[k + j], t);
				x[k + j] = cadd(x[k + j], t);
			}
			w += r;
		}
		m2 = m;
		m <<= 1;
	}
}
typedef struct {
	cf out[CGF_N], fft[CGF_N];
	float cumsum[CGF_N];
	cf rot;
	int count, window, wide;
} cgf_t;

static float cgf_correct(cgf_t* c) { /* DSP.cpp:417-467 */
	const int N = CGF_N;
	float max_val = 0.0f, fz = -1;
	int delta = (int)(9600.0 / 48000.0 * N);
	int wi = 0;
	fft512(c->fft);
	if (c->wide) {
		int M = (int)(12500.0 / 48000.0 * N);
		int ofs = (M - delta) / 2;
		float wm = -1;
		c->cumsum[0] = 0;
		for (int i = 1; i < N; i++) {
			cf v = c->fft[(i + N / 2) % N];
			c->cumsum[i] = c->cumsum[i - 1] + hypotf(v.re, v.im);
		}
		for (int i = 0; i < N - M; i++) {
			cf p = c->fft[(i + ofs + N / 2) % N], q = c->fft[(i + ofs + delta + N / 2) % N];
			float v = c->cumsum[i + M] - c->cumsum[i] + 0.6f * (hypotf(p.re, p.im) + hypotf(q.re, q.im));
			if (v > wm) { wm = v; wi = i; }
		}
		wi = (wi + M / 2 - N / 2);
	}
	for (int i = wi + c->window; i < wi + N - c->window - delta; i++) {
		cf p = c->fft[(i + N / 2) % N], q = c->fft[(i + delta + N / 2) % N];
		float h = hypotf(p.re, p.im) + hypotf(q.re, q.im);
		if (h > max_val) { max_val = h; fz = (N / 2 - (i + delta / 2.0f)); }
	}
	float f = fz / 2.0f / N;
	float ang = (float)(f * 2 * PI_F);
	cf step = { cosf(ang), sinf(ang) };
	for (int i = 0; i < N; i++) {
		c->rot = cmul(c->rot, step);
		c->out[i] = cmul(c->out[i], c->rot);
	}
	float a = hypotf(c->rot.re, c->rot.im);
	c->rot.re /= a; c->rot.im /= a;
	return f * 48000.0f / 162.0f;
}

/* ---------------------------------------------------------------- a9: PhaseSearchEMA (DSP/Demod.cpp:39-101, Demod.h:29-31,68-86) */
static const cf PS_PHASE[8] = {
	{ 9.9518472640441780e-01f, 9.8017143048367339e-02f }, { 9.5694033335306883e-01f, 2.9028468509743588e-01f },
	{ 8.8192125790916542e-01f, 4.7139674887287397e-01f }, { 7.7301044123076901e-01f, 6.3439329894649099e-01f },
	{ 6.3439326515712957e-01f, 7.7301046896098113e-01f }, { 4.7139671032286945e-01f, 8.8192127851457169e-01f },
	{ 2.9028464326824349e-01f, 9.5694034604181499e-01f }, { 9.8017099547459546e-02f, 9.9518473068888236e-01f } };
typedef struct { float ma[16]; uint8_t bits[16]; int max_idx, rot; } psema_t;

static float psema_step(psema_t* p, cf x, int nDelay) {
	const float weight = 0.85f;
	float re = 0, im = 0;
	switch (p->rot) {
	case 0: re = x.re; im = x.im; break;
	case 1: im = x.re; re = -x.im; break;
	case 2: re = -x.re; im = -x.im; break;
	case 3: im = -x.re; re = x.im; break;
	}
	p->rot = (p->rot + 1) & 3;
	for (int j = 0; j < 8; j++) {
		float t, a = re * PS_PHASE[j].re, b = im * PS_PHASE[j].im;
		t = a + b;
		p->bits[j] = (uint8_t)((p->bits[j] << 1) | (t > 0));
		p->ma[j] = weight * p->ma[j] + (1 - weight) * fabsf(t);
		t = a - b;
		p->bits[15 - j] = (uint8_t)((p->bits[15 - j] << 1) | (t > 0));
		p->ma[15 - j] = weight * p->ma[15 - j] + (1 - weight) * fabsf(t);
	}
	int idx = (p->max_idx - 1 + 16) & 15; /* nSearch = 1 */
	float max_val = p->ma[idx];
	p->max_idx = idx;
	for (int q = 0; q < 2; q++) {
		idx = (idx + 1) & 15;
		if (p->ma[idx] > max_val) { max_val = p->ma[idx]; p->max_idx = idx; }
	}
	int b2 = (p->bits[p->max_idx] >> (nDelay + 1)) & 1;
	int b1 = (p->bits[p->max_idx] >> nDelay) & 1;
	return (b1 ^ b2) ? 1.0f : -1.0f;
}

/* a9': Demod::PhaseSearch (Demod.cpp:103-170), the boxcar variant used with `-go PS_EMA off`: |t| of the last nHistory
 * symbols in a ring, summed in SLOT order (not time order), first maximum over prev-2 .. prev+2 starting from max_val = 0 */
typedef struct { float memory[16][14]; uint8_t bits[16]; int max_idx, rot, last; } psbox_t;

static float psbox_step(psbox_t* p, cf x, int nHistory, int nDelay) {
	float re = 0, im = 0;
	switch (p->rot) {
	case 0: re = x.re; im = x.im; break;
	case 1: im = x.re; re = -x.im; break;
	case 2: re = -x.re; im = -x.im; break;
	case 3: im = -x.re; re = x.im; break;
	}
	p->rot = (p->rot + 1) & 3;
	for (int j = 0; j < 8; j++) {
		float t, a = re * PS_PHASE[j].re, b = im * PS_PHASE[j].im;
		t = a + b;
		p->bits[j] = (uint8_t)((p->bits[j] << 1) | (t > 0));
		p->memory[j][p->last] = fabsf(t);
		t = a - b;
		p->bits[15 - j] = (uint8_t)((p->bits[15 - j] << 1) | (t > 0));
		p->memory[15 - j][p->last] = fabsf(t);
	}
	p->last = (p->last + 1) % nHistory;
	float max_val = 0;
	int prev_max = p->max_idx;
	for (int q = 16 + prev_max - 2; q <= 16 + prev_max + 2; q++) { /* nSearch = 2 */
		int j = q % 16;
		float avg = p->memory[j][0];
		for (int l = 1; l < nHistory; l++) avg += p->memory[j][l];
		if (avg > max_val) { max_val = avg; p->max_idx = j; }
	}
	int b2 = (p->bits[p->max_idx] >> (nDelay + 1)) & 1;
	int b1 = (p->bits[p->max_idx] >> nDelay) & 1;
	return (b1 ^ b2) ? 1.0f : -1.0f;
}

/* ---------------------------------------------------------------- a10: AIS::Decoder + NMEA (Marine/AIS.h:82-181, AIS.cpp:33-142,
 *                                                                   Marine/Message.h:36-41,171-183,264-281, Message.cpp:398-413,569-686) */
#define MAX_AIS_LENGTH 1064
#define MAX_AIS_FRAME_LENGTH (MAX_AIS_LENGTH + 16 + 7)
#define MAX_AIS_FRAME_BYTES ((MAX_AIS_FRAME_LENGTH + 7) / 8)
enum { ST_TRAINING, ST_STARTFLAG, ST_STOPFLAG, ST_DATAFCS, ST_FOUNDMESSAGE };

typedef struct ao_dec {
	int state, lastBit, prev, position, one_seq_count;
	float level;
	long long start_idx, end_idx;
	uint8_t data[MAX_AIS_FRAME_BYTES + 4];
	int length;
	char channel;
	struct ao_dec* sib[16];
	int nsib;
	int* fast_pll; /* ModelBase: DecoderMessage -> SimplePLL::Signal (Model.cpp:434-435, DSP.cpp:46-57) */
} ao_dec;

static int g_seq_id = 0; /* Message.cpp:28-39: process-global sequence counter */
void ao_reset_seq(void) { g_seq_id = 0; }

typedef struct {
	fvec level, ppm;
	char* text; long long tn, tcap;
	int count;
} msgsink;

static void sink_text(msgsink* s, const char* line, int n) {
	if (s->tn + n + 2 > s->tcap) {
		s->tcap = s->tcap ? s->tcap * 2 : 65536;
		while (s->tcap < s->tn + n + 2) s->tcap *= 2;
		s->text = (char*)realloc(s->text, (size_t)s->tcap);
	}
	memcpy(s->text + s->tn, line, (size_t)n);
	s->tn += n;
	s->text[s->tn++] = '\n';
	s->text[s->tn] = 0;
}

static void dec_next(ao_dec* d, int s, int pos) { /* AIS.cpp:33-53 */
	d->state = s; d->position = pos; d->one_seq_count = 0;
	if (d->fast_pll) { /* AIS.cpp:39-46: StartTraining / StopTraining */
		if (s == ST_TRAINING) *d->fast_pll = 1;
		else if (s == ST_STARTFLAG) *d->fast_pll = 0;
	}
	if (s == ST_FOUNDMESSAGE) /* Reset broadcast to the connected sibling decoders: AIS.cpp:47-49,98-108 */
		for (int i = 0; i < d->nsib; i++) { ao_dec* o = d->sib[i]; o->state = ST_TRAINING; o->position = 0; o->one_seq_count = 0; }
}
static void msg_setbit(ao_dec* d, int i, int b) {
	if (i >= MAX_AIS_FRAME_LENGTH || i < 0) return;
	if (b) d->data[i >> 3] |= (uint8_t)(1 << (i & 7)); else d->data[i >> 3] &= (uint8_t)~(1 << (i & 7));
}
static int msg_getbit(const ao_dec* d, int i) {
	if (i >= MAX_AIS_FRAME_LENGTH || i < 0) return 0;
	return (d->data[i >> 3] >> (i & 7)) & 1;
}
static unsigned msg_type(const ao_dec* d) { return d->data[0] >> 2; }
static unsigned msg_mmsi(const ao_dec* d) { return ((unsigned)d->data[1] << 22) | (d->data[2] << 14) | (d->data[3] << 6) | (d->data[4] >> 2); }

static int dec_cannot_be_valid(const ao_dec* d, int len) { /* AIS.cpp:111-142 */
	const int END = 24;
	if (len < 6 + END) return 0;
	int t = (int)msg_type(d);
	switch (len) {
	case 6 + 24: return t > 28 || t == 0;
	case 8 + 30 + 24: return msg_mmsi(d) > 999999999;
	case 72 + 24: return t == 10;
	case 144 + 24: return t == 16;
	case 160 + 24: return t == 15 || t == 20 || t == 23;
	case 168 + 24: return t == 1 || t == 2 || t == 3 || t == 4 || t == 7 || t == 9 || t == 11 || t == 18 || t == 22 || t == 24 || t == 25 || t == 27 || t == 28;
	case 312 + 24: return t == 19;
	case 361 + 24: return t == 21;
	case 424 + 24: return t == 5;
	}
	(void)END;
	return 0;
}
static char msg_letter(const ao_dec* d, int pos) { /* Message.cpp:642-662 */
	static const char sixbit[65] = "0123456789:;<=>?@ABCDEFGHIJKLMNOPQRSTUVW`abcdefghijklmnopqrstuvw";
	int start = pos * 6, end = start + 6;
	if (end > MAX_AIS_LENGTH || start < 0) return 0;
	int x = start >> 3, y = start & 7;
	unsigned w = ((unsigned)d->data[x] << 8) | d->data[x + 1];
	int l = (w >> (16 - 6 - y)) & 0x3F;
	int overrun = end - d->length;
	if (overrun > 0) l &= 0x3F << overrun;
	return sixbit[l];
}
static void msg_build_nmea(ao_dec* d, msgsink* sink) { /* Message.cpp:569-631 (own_mmsi = -1 -> 'M') */
	int nletters = (d->length + 5) / 6;
	int nsent = (nletters == 0) ? 1 : (nletters + 55) / 56;
	char seq = 0;
	if (nsent > 1) { seq = (char)(g_seq_id + '0'); g_seq_id = (g_seq_id + 1) % 10; }
	for (int s = 0, l = 0; s < nsent; s++) {
		char p[160];
		int i = 0;
		memcpy(p, "!AIVDM,X,X,", 11);
		p[7] = (char)(nsent + '0');
		p[9] = (char)(s + 1 + '0');
		i = 11;
		if (seq) p[i++] = seq;
		p[i++] = ',';
		if (d->channel != '?') p[i++] = d->channel;
		p[i++] = ',';
		int letters = nletters - l < 56 ? nletters - l : 56;
		for (int k = 0; k < letters; k++) p[i++] = msg_letter(d, l + k);
		l += letters;
		p[i++] = ',';
		p[i++] = (char)(((s == nsent - 1) ? nletters * 6 - d->length : 0) + '0');
		int c = 0;
		for (int k = 1; k < i; k++) c ^= (unsigned char)p[k];
		p[i++] = '*';
		p[i++] = "0123456789ABCDEF"[(c >> 4) & 0xF];
		p[i++] = "0123456789ABCDEF"[c & 0xF];
		sink_text(sink, p, i);
	}
}
static int msg_validate(const ao_dec* d) { /* Message.cpp:398-413 */
	static const int ml[28] = { 149, 149, 149, 168, 418, 88, 72, 56, 168, 70, 168, 72, 40, 40, 88, 92, 80, 168, 312, 70, 271, 145, 154, 160, 72, 60, 96, 168 };
	if (d->length == 0) return 1;
	if (d->length > MAX_AIS_LENGTH) return 0;
	unsigned t = msg_type(d);
	if (t < 1 || t > 28) return 0;
	if (d->length < ml[t - 1]) return 0;
	return 1;
}
typedef struct { float sample_lvl, level, ppm; long long sample_idx; unsigned mode; } tag_t;

static int dec_process(ao_dec* d, int len, tag_t* tag, msgsink* sink) { /* AIS.cpp:55-96 */
	if (len < 16) return 0;
	uint16_t crc = 0xFFFF;
	for (int i = 0; i < len; i++) crc = (((uint16_t)msg_getbit(d, i) ^ crc) & 1) ? (uint16_t)((crc >> 1) ^ 0x8408) : (uint16_t)(crc >> 1);
	if (crc != (uint16_t)~0x0F47) return 0;
	int nBits = len - 16;
	if ((tag->mode & 1) && tag->level != 0.0) tag->level = (float)(10.0f * log10(tag->level));
	if (nBits >= 0 && nBits <= MAX_AIS_LENGTH) d->length = nBits;
	if (msg_validate(d)) {
		msg_build_nmea(d, sink);
		fv_push(&sink->level, &tag->level, 1);
		fv_push(&sink->ppm, &tag->ppm, 1);
		sink->count++;
	}
	return 1;
}
static int dec_run(ao_dec* d, float sample, tag_t* tag, msgsink* sink) { /* AIS.h:91-181; returns 1 on the bit that completes a frame with a good CRC */
	int found = 0;
	int dd = sample > 0;
	int Bit = !(dd ^ d->prev);
	d->prev = dd;
	switch (d->state) {
	case ST_TRAINING:
		if (Bit != d->lastBit) d->position++;
		else {
			if (d->position > 4) { d->start_idx = tag->sample_idx; dec_next(d, ST_STARTFLAG, Bit ? 3 : 1); }
			else dec_next(d, ST_TRAINING, 0);
		}
		break;
	case ST_STARTFLAG:
		if (d->position == 7) {
			if (Bit == 0) { dec_next(d, ST_DATAFCS, 0); d->level = 0.0f; d->length = 0; memset(d->data, 0, sizeof(d->data)); }
			else dec_next(d, ST_TRAINING, 0);
		} else {
			if (Bit == 1) d->position++; else dec_next(d, ST_TRAINING, 0);
		}
		break;
	case ST_DATAFCS:
		msg_setbit(d, d->position++, Bit);
		if (tag->mode & 1) d->level += tag->sample_lvl;
		if (Bit == 1) {
			if (d->one_seq_count == 5) {
				if (tag->mode & 1) tag->level = d->level / d->position;
				d->end_idx = tag->sample_idx;
				found = dec_process(d, d->position - 7, tag, sink);
				if (found) dec_next(d, ST_FOUNDMESSAGE, 0);
				dec_next(d, ST_TRAINING, 0);
			} else d->one_seq_count++;
		} else {
			if (d->one_seq_count == 5) d->position--;
			d->one_seq_count = 0;
		}
		if (d->position == MAX_AIS_FRAME_LENGTH || dec_cannot_be_valid(d, d->position)) dec_next(d, ST_TRAINING, 0);
		break;
	default: break;
	}
	d->lastBit = Bit;
	return found;
}

#include "ais_oracle_v2.inc"

/* ---------------------------------------------------------------- per-channel back end */
typedef struct { fvec bits, lvl; lvec idx; } bitrec;
typedef struct {
	cic5_t ds2, fcic;
	cgf_t cgf;
	cf fir_hist[16];
	/* ScatterPLL (DSP.h:76-117) */
	cf sample[5]; int lastSymbol; float level; long long sample_idx;
	psema_t ps[5];
	psbox_t pb[5];
	ao_dec dec[5];
	/* Challenger FM branch: Demod::FM (Demod.cpp:27-37), Filter (Receiver taps), Deinterleave(5) (DSP.h:51-74) */
	cf fm_prev; float fr_hist[36]; int fm_last; long long fm_idx;
	ao_dec decf[5];
	/* ModelBase: FM -> Filter(Receiver) -> SimplePLL -> one Decoder (Model.cpp:419-438) */
	float pll; int pll_prev, pll_fast; ao_dec decb;
	v2_t* v2; /* ModelEngineV2 */
	fvec tap48, tapcgf, tapfir, ppm_cgf, ppm_fir;
	fvec tapfm, tapfr; /* real taps: Demod::FM output, Filter(Receiver) output, every 48 kHz sample in order */
	bitrec br[5], brf[5];
} chan_t;

struct ao_chain {
	int model, fmt, rate, taps;
	int npre, npost, has_us, has_fdc, has_dsk, ps_ema, mode_x;
	int fixed; uint32_t fix_h[4][5]; /* `-go FP_DS on` at 1536 kSPS: Downsample16_CU8, h0..h4 of its four DS_UINT16 stages */
	/* DownsampleKFilter (DSP.cpp:160-189, DSP.h:181-211): BlackmanHarris_28_3, K = 3, output blocks of 8192 */
	cf* dsk_buf; long long dsk_cap; cf dsk_out[8192]; int dsk_in, dsk_idx_out;
	int ma; cf ma_D; int ma_df, ma_idx_out, ma_idx_in; /* `-go MA on`: DownsampleMovingAverage (DSP.cpp:60-82); its 8192-sample output block is dsk_out */
	float fdc_alpha;
	cic5_t pre[8], post[2];
	ups_t us;
	fdc_t fdc;
	cf rot, mult;
	chan_t ch[2];
	tag_t tag;
	msgsink sink;
	cf *b0, *b1, *up, *down;
	long long bcap;
};

/* FIR-17 -> ScatterPLL -> 5x PhaseSearchEMA -> 5x Decoder, for n samples (one CGF window or one throttled sample) */
static void coherent_branch(ao_chain* c, chan_t* ch, const cf* x, int n) {
	cf y[CGF_N];
	fir_c_run(ch->fir_hist, TAPS_COHERENT, 17, x, n, y);
	if (c->taps) { fv_push(&ch->tapfir, (const float*)y, 2LL * n); fv_push(&ch->ppm_fir, &c->tag.ppm, 1); }
	for (int i = 0; i < n; i++) { /* ScatterPLL::Receive, DSP.h:95-117 */
		ch->sample[ch->lastSymbol] = y[i];
		if (c->tag.mode & 1) ch->level += y[i].re * y[i].re + y[i].im * y[i].im; /* std::norm */
		if (++ch->lastSymbol == 5) {
			if (c->tag.mode & 1) c->tag.sample_lvl = ch->level / 5;
			for (int j = 0; j < 5; j++) {
				c->tag.sample_idx = ch->sample_idx++;
				float b = c->ps_ema ? psema_step(&ch->ps[j], ch->sample[j], 3) : psbox_step(&ch->pb[j], ch->sample[j], 12, 3); /* Model.h:218-219 */
				dec_run(&ch->dec[j], b, &c->tag, &c->sink);
				if (c->taps) { fv_push(&ch->br[j].bits, &b, 1); fv_push(&ch->br[j].lvl, &c->tag.sample_lvl, 1); lv_push1(&ch->br[j].idx, c->tag.sample_idx); }
			}
			ch->level = 0.0f;
			ch->lastSymbol = 0;
		}
	}
}
static void fm_branch(ao_chain* c, chan_t* ch, cf x) { /* Model.cpp:638-639 */
	cf p = { x.re * ch->fm_prev.re - x.im * (-ch->fm_prev.im), x.re * (-ch->fm_prev.im) + x.im * ch->fm_prev.re };
	float v = atan2f(p.im, p.re) / PI_F;
	ch->fm_prev = x;
	float f = fir_r_step(ch->fr_hist, TAPS_RECEIVER, 37, v);
	if (c->taps) { fv_push(&ch->tapfm, &v, 1); fv_push(&ch->tapfr, &f, 1); }
	c->tag.sample_idx = ch->fm_idx++;
	int j = ch->fm_last;
	dec_run(&ch->decf[j], f, &c->tag, &c->sink);
	if (c->taps) { fv_push(&ch->brf[j].bits, &f, 1); fv_push(&ch->brf[j].lvl, &c->tag.sample_lvl, 1); lv_push1(&ch->brf[j].idx, c->tag.sample_idx); }
	ch->fm_last = (ch->fm_last + 1) % 5;
}

static void base_branch(ao_chain* c, chan_t* ch, cf x) { /* Model.cpp:431-432 */
	cf p = { x.re * ch->fm_prev.re - x.im * (-ch->fm_prev.im), x.re * (-ch->fm_prev.im) + x.im * ch->fm_prev.re };
	float v = atan2f(p.im, p.re) / PI_F; /* Demod::FM, Demod.cpp:27-37 */
	ch->fm_prev = x;
	float f = fir_r_step(ch->fr_hist, TAPS_RECEIVER, 37, v);
	if (c->taps) { fv_push(&ch->tapfm, &v, 1); fv_push(&ch->tapfr, &f, 1); }
	if (c->taps) { fv_push(&ch->brf[0].bits, &f, 1); fv_push(&ch->brf[0].lvl, &c->tag.sample_lvl, 1); lv_push1(&ch->brf[0].idx, c->tag.sample_idx); }
	/* SimplePLL::Receive, DSP.cpp:28-44 */
	int bit = f > 0;
	if (bit != ch->pll_prev) ch->pll += (0.5f - ch->pll) * (ch->pll_fast ? 0.6f : 0.05f);
	ch->pll += 0.2f;
	if (ch->pll >= 1.0f) {
		dec_run(&ch->decb, f, &c->tag, &c->sink);
		if (c->taps) { fv_push(&ch->br[0].bits, &f, 1); fv_push(&ch->br[0].lvl, &c->tag.sample_lvl, 1); lv_push1(&ch->br[0].idx, c->tag.sample_idx); }
		ch->pll -= (int)ch->pll;
	}
	ch->pll_prev = bit;
}

static void channel_48k(ao_chain* c, chan_t* ch, const cf* x48, int n);

static void channel_receive(ao_chain* c, chan_t* ch, const cf* x96, int n96) {
	/* DS2_a -> FCIC5_a (Model.cpp:341-346) */
	int n = n96 / 2;
	cf* t = (cf*)malloc(sizeof(cf) * (size_t)(n + 2));
	cic5_run(&ch->ds2, x96, n96, t, 1);
	channel_48k(c, ch, t, n);
	free(t);
}

/* FCIC5 and everything behind it; mode X feeds it directly (Model.cpp:35-107: ... >> FCIC5_a) */
static void channel_48k(ao_chain* c, chan_t* ch, const cf* x48, int n) {
	cf* t = (cf*)malloc(sizeof(cf) * (size_t)(n + 2));
	cf* y = t;
	cic5_run(&ch->fcic, x48, n, y, 0);
	if (c->taps) fv_push(&ch->tap48, (const float*)y, 2LL * n);
	if (c->model == 1) { /* ModelBase: no CGF, the channel goes straight into the FM receiver */
		for (int i = 0; i < n; i++) base_branch(c, ch, y[i]);
		free(t);
		return;
	}
	if (c->model == 11) { /* ModelEngineV2 (Model.cpp:440-463): the channel goes into V2::Engine */
		v2_receive(ch->v2, y, n, &c->tag, &c->sink);
		free(t);
		return;
	}
	if (c->model == 0) { /* ModelStandard (Model.cpp:484-518): FM -> Filter(Receiver) -> Deinterleave(5) -> five decoders */
		for (int i = 0; i < n; i++) fm_branch(c, ch, y[i]);
		free(t);
		return;
	}
	/* SquareFreqOffsetCorrection::Receive, DSP.cpp:475-489 */
	for (int i = 0; i < n; i++) {
		cgf_t* g = &ch->cgf;
		g->fft[g_rev[g->count]] = cmul(y[i], y[i]);
		g->out[g->count] = y[i];
		if (++g->count == CGF_N) {
			c->tag.ppm = cgf_correct(g);
			g->count = 0;
			if (c->taps) { fv_push(&ch->tapcgf, (const float*)g->out, 2LL * CGF_N); fv_push(&ch->ppm_cgf, &c->tag.ppm, 1); }
			if (c->model == 4) {
				/* throttle: Deinterleave n=1, one sample per Send (Model.cpp:630-639): FC branch first, then FM branch */
				for (int k = 0; k < CGF_N; k++) {
					coherent_branch(c, ch, &g->out[k], 1);
					fm_branch(c, ch, g->out[k]);
				}
			} else coherent_branch(c, ch, g->out, CGF_N);
		}
	}
	free(t);
}

static void frontend_96k(ao_chain* c, const cf* x, int n) { /* FDC -> ROT -> up/down (Model.cpp:222-229,341-342) */
	if (n <= 0) return;
	cf* f = (cf*)malloc(sizeof(cf) * (size_t)n * 3);
	cf* up = f + n; cf* down = up + n;
	const cf* in = x;
	if (c->has_fdc) { fdc_run(&c->fdc, c->fdc_alpha, x, n, f); in = f; }
	rotate_run(&c->rot, c->mult, in, n, up, down);
	/* Rotate sends the whole block to channel A first, then channel B (DSP.cpp:312-313); the per-call
	 * renormalisation inside rotate_run happens after both in the reference but touches only `rot`. */
	channel_receive(c, &c->ch[0], up, n);
	channel_receive(c, &c->ch[1], down, n);
	free(f);
}

static void post_us(ao_chain* c, const cf* x, int n) { /* DS2_2 -> DS2_1 (or fewer) -> 96k */
	const cf* in = x;
	int m = n;
	cf* bufs[2] = { NULL, NULL };
	for (int s = 0; s < c->npost; s++) {
		cf* o = (cf*)malloc(sizeof(cf) * (size_t)(m / 2 + 2));
		cic5_run(&c->post[s], in, m, o, 1);
		if (bufs[1]) free(bufs[1]);
		bufs[1] = o; in = o; m /= 2;
	}
	if (c->mode_x) { /* Model.cpp:35-107: [US] >> [DS2_2] >> [DS2_1] >> [FDC] >> FCIC5_a, one channel */
		if (c->has_fdc && m > 0) {
			cf* f = (cf*)malloc(sizeof(cf) * (size_t)m);
			fdc_run(&c->fdc, c->fdc_alpha, in, m, f);
			channel_48k(c, &c->ch[0], f, m);
			free(f);
		} else if (m > 0) channel_48k(c, &c->ch[0], in, m);
	} else frontend_96k(c, in, m);
	if (bufs[1]) free(bufs[1]);
}

static const float TAPS_BH_28_3[26] = { /* DSP/Filters.h:45-53 */
	6.32542387e-05f, -2.90015252e-04f, -1.54206250e-03f, -1.64972455e-03f, 3.12793899e-03f, 1.09494413e-02f, 9.04975801e-03f,
	-1.43685846e-02f, -4.45615933e-02f, -3.44883647e-02f, 5.53474269e-02f, 2.01827915e-01f, 3.16534610e-01f, 3.16534610e-01f,
	2.01827915e-01f, 5.53474269e-02f, -3.44883647e-02f, -4.45615933e-02f, -1.43685846e-02f, 9.04975801e-03f, 1.09494413e-02f,
	3.12793899e-03f, -1.64972455e-03f, -1.54206250e-03f, -2.90015252e-04f, 6.32542387e-05f };

static void dsk_run(ao_chain* c, const cf* data, int len) { /* DSP.cpp:160-189 */
	const int nt = 26, K = 3;
	if (len < nt - 1) return;
	if (c->dsk_cap < len + nt) {
		cf* nb = (cf*)calloc((size_t)len + nt, sizeof(cf));
		if (c->dsk_buf) { memcpy(nb, c->dsk_buf, sizeof(cf) * (nt - 1)); free(c->dsk_buf); }
		c->dsk_buf = nb; c->dsk_cap = len + nt;
	}
	cf* buffer = c->dsk_buf;
	for (int i = 0, j = nt - 1; i < len; i++, j++) buffer[j] = data[i];
	while (c->dsk_in < len) {
		cf x = { 0.0f, 0.0f };
		const cf* d = &buffer[c->dsk_in];
		for (int i = 0; i < nt; i++) { x.re += TAPS_BH_28_3[i] * d[i].re; x.im += TAPS_BH_28_3[i] * d[i].im; } /* DSP.h:195-201 */
		c->dsk_out[c->dsk_idx_out] = x;
		if (++c->dsk_idx_out == 8192) { frontend_96k(c, c->dsk_out, 8192); c->dsk_idx_out = 0; }
		c->dsk_in += K;
	}
	c->dsk_in -= len;
	for (int j = 0, i = len - nt + 1; j < nt - 1; i++, j++) buffer[j] = data[i];
}

static void upsample_run(ao_chain* c, const cf* x, int len) { /* DSP.cpp:192-212 */
	ups_t* u = &c->us;
	if (u->cap < len) { u->out = (cf*)realloc(u->out, sizeof(cf) * (size_t)len); u->cap = len; }
	for (int i = 0; i < len; i++) {
		cf b = x[i];
		do {
			float w0 = 1 - u->alpha;
			cf o = { w0 * u->a.re + u->alpha * b.re, w0 * u->a.im + u->alpha * b.im };
			u->out[u->idx_out++] = o;
			u->alpha += u->increment;
			if (u->idx_out == len || u->idx_out == u->cap) { /* US >> DS2_2 .., or US >> DSK on the decimate-by-3 ladders (Model.cpp:213-219 etc.) */
				if (c->has_dsk) dsk_run(c, u->out, u->idx_out); else post_us(c, u->out, u->idx_out);
				u->idx_out = 0;
			}
		} while (u->alpha < 1.0f);
		u->alpha -= 1.0f;
		u->a = b;
	}
}

/* One DS_UINT16 stage (DSP.cpp:499-522, macros :85-90): I in bits 0..15, Q in bits 16..31 of one word, five cascaded two-tap
 * sums, output at the first sample of every pair, (z >> shift) & mask */
static int ds_uint16_run(uint32_t* h, const uint32_t* in, uint32_t* out, int len, int shift) {
	uint32_t mask = 0xFFFFu >> shift;
	mask |= mask << 16;
	len >>= 1;
	for (int i = 0; i < len; i++) {
		uint32_t r[5], z = in[2 * i];
		for (int k = 0; k < 5; k++) { r[k] = z; z += h[k]; }   /* MA1(0..4) */
		out[i] = (z >> shift) & mask;
		z = in[2 * i + 1];
		for (int k = 0; k < 5; k++) { h[k] = z; z += r[k]; }   /* MA2(0..4) */
	}
	return len;
}
/* Downsample16_CU8::Receive (DSP.cpp:639-651): shifts 3, 4, 5, then the stage that ends in int16 / 32768.0f (:587-607) */
static int ds16_cu8(ao_chain* c, const uint8_t* u, int n, cf* out) {
	uint32_t* buf = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n + 2));
	for (int i = 0; i < n; i++) buf[i] = (uint32_t)u[2 * i] | ((uint32_t)u[2 * i + 1] << 16);
	int len = ds_uint16_run(c->fix_h[0], buf, buf, n, 3);
	len = ds_uint16_run(c->fix_h[1], buf, buf, len, 4);
	len = ds_uint16_run(c->fix_h[2], buf, buf, len, 5);
	len = ds_uint16_run(c->fix_h[3], buf, buf, len, 0);
	for (int i = 0; i < len; i++) {
		uint32_t z = buf[i] ^ 0x80008000u; /* uint to int in parallel by flipping the sign bits */
		out[i].re = ((int16_t)(z & 0xFFFFu)) / 32768.0f;
		out[i].im = ((int16_t)(z >> 16)) / 32768.0f;
	}
	free(buf);
	return len;
}

static void ma_run(ao_chain* c, const cf* data, int len) { /* DSP.cpp:60-82: integrate and dump to 96 kHz, blocks of 8192 */
	for (int i = 0; i < len; i++) {
		c->ma_D.re += data[i].re; c->ma_D.im += data[i].im; /* D += data[i] */
		c->ma_df++;
		c->ma_idx_out += 96000; /* out_rate (Model.cpp:124) */
		if (c->ma_idx_out >= c->rate) {
			c->ma_idx_out %= c->rate;
			const float df = (float)c->ma_df; /* D / (FLOAT32)df: complex / real = two divisions */
			c->dsk_out[c->ma_idx_in].re = c->ma_D.re / df; c->dsk_out[c->ma_idx_in].im = c->ma_D.im / df;
			c->ma_D.re = 0.0f; c->ma_D.im = 0.0f; c->ma_df = 0;
			if (++c->ma_idx_in == 8192) { frontend_96k(c, c->dsk_out, 8192); c->ma_idx_in = 0; }
		}
	}
}

int ao_feed(ao_chain* c, const void* data, int nbytes) {
	int n = c->fmt == 1 ? nbytes / 8 : c->fmt == 3 ? nbytes / 4 : nbytes / 2;
	if (n > c->bcap) {
		c->b0 = (cf*)realloc(c->b0, sizeof(cf) * (size_t)n);
		c->b1 = (cf*)realloc(c->b1, sizeof(cf) * (size_t)n);
		c->bcap = n;
	}
	const cf* in;
	if (c->fixed) { /* Model.cpp:231-237: convert.outCU8 >> DS16_CU8 >> [FDC] >> ROT; any other format produces nothing */
		if (c->fmt != 0) return 0;
		int m16 = ds16_cu8(c, (const uint8_t*)data, n, c->b0);
		frontend_96k(c, c->b0, m16);
		return 0;
	}
	if (c->fmt == 0) { /* Utilities/Convert.cpp:255-264 */
		const uint8_t* u = (const uint8_t*)data;
		for (int i = 0; i < n; i++) { c->b0[i].re = ((int)u[2 * i] - 128) / 128.0f; c->b0[i].im = ((int)u[2 * i + 1] - 128) / 128.0f; }
		in = c->b0;
	} else if (c->fmt == 2) { /* CS8, Utilities/Convert.cpp:266-275 */
		const int8_t* u = (const int8_t*)data;
		for (int i = 0; i < n; i++) { c->b0[i].re = u[2 * i] / 128.0f; c->b0[i].im = u[2 * i + 1] / 128.0f; }
		in = c->b0;
	} else if (c->fmt == 3) { /* CS16, Utilities/Convert.cpp:277-286 */
		const int16_t* u = (const int16_t*)data;
		for (int i = 0; i < n; i++) { c->b0[i].re = u[2 * i] / 32768.0f; c->b0[i].im = u[2 * i + 1] / 32768.0f; }
		in = c->b0;
	} else in = (const cf*)data;
	if (c->ma) { ma_run(c, in, n); return 0; } /* Model.cpp:122-126: physical >> convert >> DS_MA >> ROT */
	int m = n;
	cf* o = c->b1;
	for (int s = 0; s < c->npre; s++) {
		cic5_run(&c->pre[s], in, m, o, 1);
		in = o; m /= 2;
		o = (o == c->b1) ? c->b0 : c->b1;
	}
	if (c->has_us) upsample_run(c, in, m);
	else if (c->has_dsk) dsk_run(c, in, m);
	else post_us(c, in, m);
	return 0;
}

ao_chain* ao_create(int model, int sample_rate, int fmt, int flags) {
	/* flags: bit 0 record taps, bit 1 `-go DSK on`, bit 2 `-go PS_EMA off`, bit 3 `-go FP_DS on`, bit 5 `-go MA on` (bit 4 below),
	 * bit 8 `-go AFC_WIDE off` (Model.cpp:536-540,586-588), bit 9 `-go DROOP off` (Model.cpp:384-386: every ladder without FDC) */
	static const unsigned buckets_nodsk[] = { 96000, 192000, 288000, 384000, 768000, 1536000, 3072000, 6144000, 12288000 }; /* Model.cpp:129-130 */
	static const unsigned buckets_dsk[] = { 96000, 192000, 288000, 384000, 576000, 768000, 1152000, 1536000, 2304000, 3072000, 6144000, 12288000 };
	/* ... bit 4 channel mode X (`-c X`: one channel, 12k .. 192k, Model.cpp:35-107) */
	static const unsigned buckets_x[] = { 48000, 96000, 192000 };
	const int taps = flags & 1, dsk = (flags >> 1) & 1, mode_x = (flags >> 4) & 1;
	const unsigned* buckets = mode_x ? buckets_x : dsk ? buckets_dsk : buckets_nodsk;
	const int nb = mode_x ? 3 : dsk ? 12 : 9;
	tables_init();
	int bi = -1;
	for (int i = 0; i < nb; i++) if (buckets[i] >= (unsigned)sample_rate) { bi = i; break; }
	if (bi < 0 || sample_rate < (mode_x ? 12000 : 96000)) return NULL;
	const unsigned bucket = buckets[bi];
	int k = 0, is3 = 0; /* bucket = 96000 * 2^k or 288000 * 2^k (mode X: 48000 * 2^k) */
	if (mode_x) while ((48000u << k) != bucket) k++;
	else if (bucket % 288000 == 0) { is3 = 1; while ((288000u << k) != bucket) k++; }
	else while ((96000u << k) != bucket) k++;
	static const float alphas[] = { 0.0f, -0.8f, -1.1f, -1.2f, -1.2f, -1.5f, -2.0f, -2.0f }; /* Model.cpp:157-338 */
	ao_chain* c = (ao_chain*)calloc(1, sizeof(ao_chain));
	c->model = model; c->fmt = fmt; c->rate = sample_rate; c->taps = taps;
	c->ps_ema = !((flags >> 2) & 1);
	c->fixed = ((flags >> 3) & 1) && sample_rate == 1536000; /* Model.cpp:224: only the 1536k case looks at fixedpointDS */
	c->has_dsk = is3;
	c->has_us = bucket != (unsigned)sample_rate;
	c->mode_x = mode_x;
	c->has_fdc = !is3 && k > 0 && !((flags >> 9) & 1); /* the decimate-by-3 ladders have no droop compensation (Model.cpp:207-219 etc.); `DROOP off`: nobody has */
	c->fdc_alpha = is3 ? 0.0f : alphas[k];
	if (mode_x) c->fdc_alpha = k == 2 ? -1.1f : -0.8f; /* Model.cpp:64,76 */
	if (mode_x) { c->npre = 0; c->npost = k; } /* convert >> [US] >> DS2_2 >> DS2_1 */
	else if (is3) { c->npre = k; c->npost = 0; } /* convert >> DS2_k .. DS2_1 >> [US] >> DSK */
	else if (c->has_us) { c->npost = k >= 2 ? 2 : k; c->npre = k - c->npost; }
	else { c->npre = k; c->npost = 0; }
	if (((flags >> 5) & 1) && !mode_x) { /* Model.cpp:111-126: the MA branch comes before the ladders (and before FP_DS is looked at) */
		c->ma = 1; c->fixed = 0; c->has_dsk = 0; c->has_us = 0; c->has_fdc = 0; c->npre = 0; c->npost = 0;
	}
	c->us.increment = (float)sample_rate / (float)bucket;
	c->rot.re = 1.0f; c->rot.im = 0.0f;
	ao_rotate_mult((float*)&c->mult);
	c->tag.mode = 3; /* Common.h:242 */
	for (int q = 0; q < 2; q++) {
		chan_t* ch = &c->ch[q];
		ch->cgf.rot.re = 1.0f;
		ch->cgf.window = 187; ch->cgf.wide = !((flags >> 8) & 1); /* Model.cpp:533-540 */
		ch->decb.channel = mode_x ? 'X' : "AB"[q]; ch->decb.fast_pll = &ch->pll_fast; /* Model.cpp:434-435 */
		ch->pll_fast = 1; /* DSP.h:40 */
		if (model == 11) { ch->v2 = (v2_t*)malloc(sizeof(v2_t)); v2_init(ch->v2, "AB"[q]); }
		for (int j = 0; j < 5; j++) {
			ch->dec[j].channel = mode_x ? 'X' : "AB"[q]; ch->decf[j].channel = mode_x ? 'X' : "AB"[q];
			/* Reset mesh: Model.cpp:566-573 (Default), :658-674 (Challenger) */
			for (int i = 0; i < 5; i++) {
				if (model == 4) {
					ch->decf[j].sib[ch->decf[j].nsib++] = &ch->dec[i];
					ch->dec[j].sib[ch->dec[j].nsib++] = &ch->decf[i];
				}
				if (i != j) {
					ch->dec[j].sib[ch->dec[j].nsib++] = &ch->dec[i];
					if (model == 4 || model == 0) ch->decf[j].sib[ch->decf[j].nsib++] = &ch->decf[i]; /* Standard: Model.cpp:505-514 */
				}
			}
		}
	}
	return c;
}

static void fv_free(fvec* v) { free(v->p); }
void ao_destroy(ao_chain* c) {
	if (!c) return;
	free(c->dsk_buf);
	for (int q = 0; q < 2; q++) {
		chan_t* ch = &c->ch[q];
		free(ch->v2);
		fv_free(&ch->tap48); fv_free(&ch->tapcgf); fv_free(&ch->tapfir); fv_free(&ch->ppm_cgf); fv_free(&ch->ppm_fir); fv_free(&ch->tapfm); fv_free(&ch->tapfr);
		for (int j = 0; j < 5; j++) {
			fv_free(&ch->br[j].bits); fv_free(&ch->br[j].lvl); free(ch->br[j].idx.p);
			fv_free(&ch->brf[j].bits); fv_free(&ch->brf[j].lvl); free(ch->brf[j].idx.p);
		}
	}
	free(c->b0); free(c->b1); free(c->us.out); free(c->sink.text); fv_free(&c->sink.level); fv_free(&c->sink.ppm);
	free(c);
}
int ao_msg_count(ao_chain* c) { return c->sink.count; }
int ao_nmea(ao_chain* c, char* dst, int cap) {
	int n = (int)c->sink.tn;
	if (dst && cap > 0) { int k = n < cap - 1 ? n : cap - 1; if (k) memcpy(dst, c->sink.text, (size_t)k); dst[k] = 0; }
	return n + 1;
}
int ao_msg_meta(ao_chain* c, float* level, float* ppm, int cap) {
	int n = (int)c->sink.level.n;
	for (int i = 0; i < n && i < cap; i++) { level[i] = c->sink.level.p[i]; ppm[i] = c->sink.ppm.p[i]; }
	return n;
}
static long long copy_out(const fvec* v, float* dst, long long cap, int per) {
	long long n = v->n / per;
	if (dst) memcpy(dst, v->p, sizeof(float) * (size_t)((n < cap ? n : cap) * per));
	return n;
}
long long ao_tap(ao_chain* c, int which, float* dst, long long cap) {
	chan_t* ch = &c->ch[which & 1];
	const fvec* v = which < 2 ? &ch->tap48 : which < 4 ? &ch->tapcgf : &ch->tapfir;
	return copy_out(v, dst, cap, 2);
}
/* recording of taps / decisions on or off from now on (created with taps: a long stream can be fed with only its end recorded) */
void ao_set_taps(ao_chain* c, int on) { c->taps = on ? 1 : 0; }
/* real-valued taps of the FM receivers (ModelChallenger FM branch, ModelBase, ModelStandard): 6/7 = Demod::FM output A/B
 * (Demod.cpp:27-37), 8/9 = Filter(Receiver) output A/B (DSP.cpp:249-280), one float per 48 kHz sample */
long long ao_tapf(ao_chain* c, int which, float* dst, long long cap) {
	if (which < 6 || which > 9) return 0;
	chan_t* ch = &c->ch[which & 1];
	return copy_out(which < 8 ? &ch->tapfm : &ch->tapfr, dst, cap, 1);
}
long long ao_tap_ppm(ao_chain* c, int which, float* dst, long long cap) {
	chan_t* ch = &c->ch[which & 1];
	if (which < 2) return 0;
	return copy_out(which < 4 ? &ch->ppm_cgf : &ch->ppm_fir, dst, cap, 1);
}
long long ao_bits(ao_chain* c, int chn, int j, int fm, float* bits, float* lvl, long long* idx, long long cap) {
	bitrec* r = fm ? &c->ch[chn].brf[j] : &c->ch[chn].br[j];
	long long n = r->bits.n, k = n < cap ? n : cap;
	if (bits) memcpy(bits, r->bits.p, sizeof(float) * (size_t)k);
	if (lvl) memcpy(lvl, r->lvl.p, sizeof(float) * (size_t)k);
	if (idx) memcpy(idx, r->idx.p, sizeof(long long) * (size_t)k);
	return n;
}
