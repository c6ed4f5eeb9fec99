// ais-catcher_amd/csrc/dec_core.h -- the frame decoder's arithmetic (AIS::Decoder, reference Marine/AIS.h:82-181, Marine/AIS.cpp:33-142)
// as code that compiles for the device (kernels.hip) AND for the host: tests/dec_core_fuzz.cpp runs the word-parallel frame
// evaluator below against the symbol-by-symbol step on millions of random streams without a GPU.  Nothing here is shipped as a
// CPU path: the host build exists for that test only.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define DEC_HD __device__ __forceinline__
#else
#define DEC_HD static inline
#endif

constexpr int DEC_LANES = 64; // a decoder's frame buffer is one column of a [word][64 lanes] tile: word w at data[64 * w]

struct DecReg { int state, lastBit, prev, position, osc; float level; long long start_idx; uint32_t crc, cw, tail; int cwi, abort_pos; };
enum { DST_TRAINING = 0, DST_STARTFLAG = 1, DST_DATAFCS = 3 };
constexpr int DEC_MAX_FRAME = 1064 + 16 + 7;

// Decoder::cannotBeValid (Marine/AIS.cpp:111-142) looks at the message type at frame positions 30, 96, 168, 184, 192, 336, 385,
// 448 (and at the MMSI at 62).  The type is final long before position 30, so the one position at which a frame of this type
// gets aborted is looked up once, at position 30: 0 = never, 30 = now (type 0 or > 28).
DEC_HD int dec_abort_position(int t) {
	constexpr uint32_t at192 = (1u << 1) | (1u << 2) | (1u << 3) | (1u << 4) | (1u << 7) | (1u << 9) | (1u << 11) | (1u << 18) | (1u << 22) |
	                           (1u << 24) | (1u << 25) | (1u << 27) | (1u << 28);
	if (t > 28 || t == 0) return 30;
	if ((at192 >> t) & 1u) return 192;
	if (t == 15 || t == 20 || t == 23) return 184;
	return t == 10 ? 96 : t == 16 ? 168 : t == 19 ? 336 : t == 21 ? 385 : t == 5 ? 448 : 0;
}

// one symbol; data = this lane's column of the LDS frame buffer (word w at data[64 * w]); returns true when a frame with a
// good CRC has just been completed (r.position / r.level still hold the frame's values, the caller finishes the transition).
// A wave's decoders are in all states at once and the wave is alone on its SIMD, so what counts is the number of
// instructions per symbol; everything is therefore evaluated with selects for all lanes, and only two rare events branch
// (the type / MMSI look-ups at positions 30 and 62):
//  * the 32-bit word of the frame that is being filled lives in a register (r.cw) and is written to LDS when the position
//    moves on to the next word;
//  * the CRC-16/X.25 register (AIS.cpp:55-64) runs SEVEN BITS BEHIND the stored bits (r.tail holds those seven): the
//    residue check covers the first position-7 bits, so when the closing flag is complete the register already is the
//    answer -- no loop over the frame, no undoing.  A de-stuffed bit advances nothing.
// DATA_ONLY: the caller guarantees r.state == DST_DATAFCS (k7e_sim: a run leaves TRAINING / STARTFLAG after a few symbols and ends
// when it leaves DATAFCS) -- the TRAINING / STARTFLAG half of the step folds away
template <bool DATA_ONLY = false>
DEC_HD bool dec_step(DecReg& r, int dd, float slvl, long long sidx, uint32_t* data) {
	const int Bit = dd == r.prev; // NRZI: !(d ^ prev)
	r.prev = dd;
	const int st = DATA_ONLY ? (int)DST_DATAFCS : r.state, pos = r.position, osc = r.osc;
	const bool isD = DATA_ONLY || st == DST_DATAFCS, isT = !DATA_ONLY && st == DST_TRAINING;
	// ---- TRAINING: count alternations; two equal bits after more than four of them are the start of a flag
	const bool alt = Bit != r.lastBit;
	const bool to_flag = isT && !alt && pos > 4;
	// ---- STARTFLAG: ones up to position 7, then a zero opens the frame
	const bool open = st == DST_STARTFLAG && pos == 7 && Bit == 0;
	const bool more = st == DST_STARTFLAG && pos != 7 && Bit == 1;
	const int tf_state = isT ? (to_flag ? DST_STARTFLAG : DST_TRAINING) : (open ? DST_DATAFCS : (more ? DST_STARTFLAG : DST_TRAINING));
	const int tf_pos = isT ? (alt ? pos + 1 : (to_flag ? (Bit ? 3 : 1) : 0)) : (more ? pos + 1 : 0);
	const int tf_osc = (isT ? alt : more) ? osc : 0; // every NextState() call clears one_seq_count (AIS.cpp:33-37)
	// ---- DATAFCS
	const bool stuffed = Bit == 0 && osc == 5; // bit de-stuffing: the position does not advance, the next bit overwrites this one
	const bool close = Bit == 1 && osc == 5;   // six ones: closing flag (or abort)
	const bool adv = isD && !stuffed;
	const int wi = pos >> 5;
	const bool next_word = isD && wi != r.cwi;
	if (next_word) data[64 * r.cwi] = r.cw;
	uint32_t cw = next_word ? 0u : r.cw;
	const uint32_t m = 1u << (pos & 31);
	if (isD && pos < DEC_MAX_FRAME) cw = Bit ? (cw | m) : (cw & ~m);
	const uint32_t outb = (r.tail >> 6) & 1u; // the bit that leaves the seven-bit window enters the CRC
	const uint32_t crc_n = ((outb ^ r.crc) & 1u) ? ((r.crc >> 1) ^ 0x8408u) : (r.crc >> 1);
	const uint32_t crc = (adv && pos >= 7) ? crc_n : r.crc;
	const uint32_t tail = adv ? (((r.tail << 1) | (uint32_t)Bit) & 127u) : r.tail;
	const int np = stuffed ? pos : pos + 1;
	const bool found = isD && close && np - 7 >= 16 && crc == (uint32_t)(uint16_t)~0x0F47;
	bool abort_frame = np == DEC_MAX_FRAME || (r.abort_pos != 0 && np == r.abort_pos);
	int abort_pos = r.abort_pos;
	if (isD && !close && (np == 30 || np == 62)) { // once per frame each
		if (np == 30) { // type = first byte >> 2; bits 0..29 are all in the first word, which is still in the register
			abort_pos = dec_abort_position((int)((cw & 255u) >> 2));
			abort_frame = abort_frame || abort_pos == 30;
		} else { // MMSI = bits 8..37: first word is in LDS by now, the second one in the register
			const uint32_t w0 = data[0];
			abort_frame = abort_frame || (((w0 >> 8) & 255u) << 22 | ((w0 >> 16) & 255u) << 14 | (w0 >> 24) << 6 | (cw & 255u) >> 2) > 999999999u;
		}
	}
	const bool leave = (close && !found) || (!close && abort_frame);
	const int d_state = leave ? DST_TRAINING : DST_DATAFCS;
	const int d_pos = leave ? 0 : np; // (when found, position still is the frame's: the caller needs it)
	const int d_osc = (close || leave) ? 0 : (Bit ? osc + 1 : 0);
	// ---- commit
	r.state = isD ? d_state : tf_state;
	r.position = isD ? d_pos : tf_pos;
	r.osc = isD ? d_osc : tf_osc;
	r.level = isD ? r.level + slvl : (open ? 0.0f : r.level); // tag.mode & 1 (Common.h:242)
	if (to_flag) r.start_idx = sidx;
	r.crc = open ? 0xFFFFu : crc;
	r.tail = open ? 0u : tail;
	r.cw = open ? 0u : cw; // (msg.clear(): bits at and beyond `position` are never read)
	r.cwi = open ? 0 : (isD ? wi : r.cwi);
	r.abort_pos = open ? 0 : abort_pos;
	r.lastBit = Bit;
	if (found) data[64 * r.cwi] = r.cw; // the record is copied out of LDS
	return found;
}

// dec_step() for a decoder that is NOT inside a frame (TRAINING or STARTFLAG; Marine/AIS.h:91-181 with state != DATAFCS), as arithmetic
// on 0 / 1 integers -- no branch, no memory: the step may open a frame (STARTFLAG -> DATAFCS), it can never complete one.  `on` = 0
// leaves the decoder as it is.  Field by field what dec_step computes for such a decoder (tests/dec_core_fuzz.cpp `idle` checks it against dec_step on random states; the engine's parity tests).
DEC_HD void dec_step_idle(DecReg& r, int dd, long long sidx, int on) {
	const int Bit = dd == r.prev;
	const int pos = r.position, st = r.state;
	const int isT = st == DST_TRAINING, isS = st == DST_STARTFLAG;
	const int alt = Bit != r.lastBit;
	const int to_flag = isT & (alt ^ 1) & (pos > 4);
	const int at7 = pos == 7;
	const int open = isS & at7 & (Bit ^ 1);
	const int more = isS & (at7 ^ 1) & Bit;
	const int n_state = to_flag | more ? (int)DST_STARTFLAG : (open ? (int)DST_DATAFCS : (int)DST_TRAINING);
	const int grow = (isT & alt) | more;                       // position + 1
	const int n_pos = grow ? pos + 1 : (to_flag ? 1 + 2 * Bit : 0);
	const int n_osc = grow ? r.osc : 0;                        // every NextState() call clears one_seq_count (AIS.cpp:33-37)
	r.prev = on ? dd : r.prev;
	r.state = on ? n_state : st;
	r.position = on ? n_pos : pos;
	r.osc = on ? n_osc : r.osc;
	const int op = open & on;
	r.level = op ? 0.0f : r.level;
	r.start_idx = (to_flag & on) ? sidx : r.start_idx;
	r.crc = op ? 0xFFFFu : r.crc;
	r.tail = op ? 0u : r.tail;
	r.cw = op ? 0u : r.cw;
	r.cwi = op ? 0 : r.cwi;
	r.abort_pos = op ? 0 : r.abort_pos;
	r.lastBit = on ? Bit : r.lastBit;
}

// ------------------------------------------------------------------------------------------
// A frame in DATAFCS, 32 symbols at a time.
//
// Inside a frame the step above is a function of very little: the position advances with every symbol that is not a stuffed zero
// (a zero behind exactly five ones), the frame closes at the first run of six ones, and it is abandoned where the position
// reaches 30 (message type impossible), 62 (MMSI impossible), the type's own limit, or the maximum length.  The evaluator works
// on words of NRZI bits: run detection, the stuffed-bit mask and the close position are shifts and ANDs, the de-stuffed bits are
// appended to the frame buffer word-wise, the few thresholds are looked up with a rank-select, and the CRC register -- which the
// step keeps seven bits behind the stored bits -- is computed from the frame buffer (byte-wise, table in `tab`) only when a
// closing flag needs it or the block ends.  The level sum (tag.sample_lvl of every DATAFCS symbol, added in symbol order like
// the step does) is likewise only formed for a completed message or a frame that continues in the next block.
// Everything it leaves behind -- (end, flags) and, for flags 1 / 2, the DecReg and the frame buffer -- is what stepping
// dec_step() symbol by symbol would have left (tests/dec_core_fuzz.cpp checks exactly that).
// ------------------------------------------------------------------------------------------
DEC_HD void dec_crc_table_entry(int i, uint16_t* tab) { // tab[i]: eight steps of the bit-serial register (AIS.cpp:55-64) from i
	uint32_t c = (uint32_t)i;
	for (int k = 0; k < 8; k++) c = (c & 1u) ? ((c >> 1) ^ 0x8408u) : (c >> 1);
	tab[i] = (uint16_t)c;
}

// CRC register after the frame's first `count` bits (bit p = bit p & 31 of word p >> 5)
template <int STRIDE = DEC_LANES> // (columns of the frame-buffer tile: kv2_engine_roles keeps eight)
DEC_HD uint32_t dec_crc_bits(const uint32_t* data, int count, const uint16_t* tab) {
	uint32_t crc = 0xFFFFu;
	const int nbytes = count >> 3;
	for (int k = 0; k < nbytes; k += 4) {
		const uint32_t w = data[STRIDE * (k >> 2)];
		const int m = nbytes - k < 4 ? nbytes - k : 4;
		for (int b = 0; b < m; b++) crc = (crc >> 8) ^ tab[(crc ^ (w >> (8 * b))) & 255u];
	}
	for (int p = nbytes * 8; p < count; p++) {
		const uint32_t bit = (data[STRIDE * (p >> 5)] >> (p & 31)) & 1u;
		crc = ((bit ^ crc) & 1u) ? ((crc >> 1) ^ 0x8408u) : (crc >> 1);
	}
	return crc;
}

// ------------------------------------------------------------------------------------------
// The lean pair (round 6, kv2_engine's two-wave form): the same decoder with fewer instructions per symbol.  A decoder that uses them
// uses them for every symbol: r.crc and r.tail are NOT maintained (the CRC of a frame is formed from its buffer when a closing flag asks
// for it, byte-wise through `tab`), and outside DATAFCS the frame registers (osc, cw, cwi, level, abort_pos) are DON'T-CARES: dec_lean_idle
// never reads them and clears them where it opens a frame (kv2_engine_roles' tracker wave relies on that: it commits the in-frame
// arithmetic for every lane).  Whoever stores a decoder must not use cwi as an index unless the decoder is inside a frame.
//  * dec_lean_idle: a decoder in TRAINING / STARTFLAG; returns 1 when the step opened a frame (the frame registers are cleared).
//  * dec_lean_data: a decoder in DATAFCS; returns true when a frame with a good CRC has just been completed (as dec_step: position and
//    level still the frame's).  Everything that is rare -- closing flag, the look-ups at positions 30 / 62, the type's own limit, the
//    maximum length -- sits behind one branch.
// tests/dec_core_fuzz.cpp `lean` steps both forms side by side over random streams.
// ------------------------------------------------------------------------------------------
DEC_HD int dec_lean_idle(DecReg& r, int dd, long long sidx) {
	const int Bit = dd == r.prev;
	const int pos = r.position;
	const bool isT = r.state == DST_TRAINING;
	const bool alt = Bit != r.lastBit;
	const bool at7 = pos == 7;
	const bool to_flag = isT && !alt && pos > 4;
	const bool open = !isT && at7 && !Bit;
	const bool more = !isT && !at7 && Bit;
	const bool grow = (isT && alt) || more;
	r.prev = dd;
	r.lastBit = Bit;
	r.state = (to_flag || more) ? (int)DST_STARTFLAG : (open ? (int)DST_DATAFCS : (int)DST_TRAINING);
	r.position = grow ? pos + 1 : (to_flag ? 1 + 2 * Bit : 0);
	r.start_idx = to_flag ? sidx : r.start_idx;
	if (open) { r.level = 0.0f; r.cw = 0u; r.cwi = 0; r.abort_pos = 0; r.osc = 0; }
	return open ? 1 : 0;
}

template <int STRIDE = DEC_LANES>
DEC_HD bool dec_lean_data(DecReg& r, int dd, float slvl, uint32_t* data, const uint16_t* tab) {
	const int Bit = dd == r.prev;
	r.prev = dd;
	r.lastBit = Bit;
	const int pos = r.position, osc = r.osc;
	const bool six = osc == 5;
	const bool close = six && Bit;
	const int np = (six && !Bit) ? pos : pos + 1; // a stuffed zero does not advance: the next bit overwrites it
	const int wi = pos >> 5;
	if (wi != r.cwi) { data[STRIDE * r.cwi] = r.cw; r.cw = 0u; r.cwi = wi; }
	const uint32_t sh = (uint32_t)pos & 31u;
	if (pos < DEC_MAX_FRAME) r.cw = (r.cw & ~(1u << sh)) | ((uint32_t)Bit << sh);
	r.level = r.level + slvl;
	r.position = np;
	r.osc = Bit ? osc + 1 : 0;
	if (!(close || np == 30 || np == 62 || np == DEC_MAX_FRAME || np == r.abort_pos)) return false;
	// ---- the rare rest of the step
	if (close) {
		data[STRIDE * r.cwi] = r.cw;
		r.osc = 0;
		if (np - 7 >= 16 && dec_crc_bits<STRIDE>(data, np - 7, tab) == (uint32_t)(uint16_t)~0x0F47) return true; // (state, position, level: the frame's)
		r.state = DST_TRAINING; r.position = 0;
		return false;
	}
	bool abort_frame = np == DEC_MAX_FRAME || (r.abort_pos != 0 && np == r.abort_pos);
	if (np == 30) { // type = first byte >> 2; bits 0..29 are all in the first word, which is still in the register
		r.abort_pos = dec_abort_position((int)((r.cw & 255u) >> 2));
		abort_frame = abort_frame || r.abort_pos == 30;
	} else if (np == 62) { // MMSI = bits 8..37: first word is in the buffer by now, the second one in the register
		const uint32_t w0 = data[0];
		abort_frame = abort_frame || (((w0 >> 8) & 255u) << 22 | ((w0 >> 16) & 255u) << 14 | (w0 >> 24) << 6 | (r.cw & 255u) >> 2) > 999999999u;
	}
	if (abort_frame) { r.state = DST_TRAINING; r.position = 0; r.osc = 0; }
	return false;
}

DEC_HD int dec_select_bit(uint32_t m, int k) { // index of the k-th (1-based) set bit of m (k <= popcount)
	for (int i = 1; i < k; i++) m &= m - 1;
	return __builtin_ctz(m);
}

// r: a decoder in DATAFCS in front of symbol g (r.prev = decision g-1); brow: the packed decisions of the block (nw words, n
// symbols); lrow: the levels.  Returns flags -- 0: back in TRAINING at symbol `end`; 1: message completed at `end`;
// 2: the block ended (`end` = n) -- and for 1 / 2 the state and the frame buffer.
// lvl_shift / lvl_first: ModelChallenger's FM0..FM3 see tag.sample_lvl as the PREVIOUS group's ScatterPLL left it (symbol i adds
// lrow[i - 1]; symbol 0 what the block before -- or the other channel -- left: lvl_first).
// LAZY: r.crc and r.tail are NOT brought up to date for flags 2 (a caller that feeds the frame word by word would pay a pass over
// the whole frame buffer per word); dec_fix_crc_tail() forms them when the symbol-by-symbol step is about to need them.
DEC_HD void dec_fix_crc_tail(DecReg& r, uint32_t* data, const uint16_t* tab) {
	data[DEC_LANES * r.cwi] = r.cw; // (the word that is being filled lives in the register)
	const int pos = r.position;
	r.crc = dec_crc_bits(data, pos >= 7 ? pos - 7 : 0, tab);
	uint32_t tail = 0;
	for (int k = 0; k < 7; k++) {
		const int p = pos - 1 - k;
		if (p >= 0) tail |= ((data[DEC_LANES * (p >> 5)] >> (p & 31)) & 1u) << k;
	}
	r.tail = tail;
}
template <bool LAZY = false>
DEC_HD int dec_run_frame(DecReg& r, uint32_t* data, const uint32_t* brow, const float* lrow, int g, int n, const uint16_t* tab, int& end,
                         int lvl_shift = 0, float lvl_first = 0.0f) {
	const int g_first = g;
	int pos = r.position, osc = r.osc, abort_pos = r.abort_pos;
	uint32_t dprev = (uint32_t)r.prev;
	// the word that is being filled: the step flushes it when the position has moved on to the next word
	int wi = pos >> 5;
	uint32_t cw = r.cw;
	if (wi != r.cwi) { data[DEC_LANES * r.cwi] = cw; cw = 0u; }
	const int nw = (n + 31) >> 5;
	int flags = 2, last_stuffed = 0;
	uint32_t lastB = (uint32_t)r.lastBit;
	end = n;
	// a window is 32 symbols from g on: two adjacent words of the row, and the window after it needs the next one -- requested two
	// windows ahead, a wave that is alone on its SIMD has nothing else to cover a memory round trip with
	const int sh = g & 31;
	const auto word_at = [&](int i) { return brow[i < nw ? i : nw - 1]; };
	uint32_t wa = g < n ? word_at(g >> 5) : 0u, wb = g < n ? word_at((g >> 5) + 1) : 0u, wc = g < n ? word_at((g >> 5) + 2) : 0u;
	while (g < n) {
		const int nv = n - g < 32 ? n - g : 32;
		const uint32_t wd = word_at((g >> 5) + 3);
		const uint32_t D = sh ? ((wa >> sh) | (wb << (32 - sh))) : wa;
		wa = wb; wb = wc; wc = wd;
		const uint32_t B = ~(D ^ ((D << 1) | dprev)); // NRZI
		const uint32_t valid = nv < 32 ? ((1u << nv) - 1u) : 0xFFFFFFFFu;
		// ones in front of the window: the `osc` counted so far
		const unsigned long long X = ((unsigned long long)B << 5) | (unsigned long long)(((1u << osc) - 1u) << (5 - osc));
		const unsigned long long R5 = X & (X >> 1) & (X >> 2) & (X >> 3) & (X >> 4); // bit i: the five symbols before symbol i are ones
		const uint32_t five = (uint32_t)R5;
		const uint32_t C = five & B & valid; // sixth one: closing flag
		const int e = C ? __builtin_ctz(C) : 32;
		const int lim = C ? e + 1 : nv; // symbols of this window that are stepped at all
		const uint32_t lm = lim < 32 ? ((1u << lim) - 1u) : 0xFFFFFFFFu;
		const uint32_t S = five & ~B & lm;  // stuffed zeros
		const uint32_t K = lm & ~S;         // symbols that advance the position
		const int a = __builtin_popcount(K);
		// de-stuff (from the top, so that the lower indices stay valid) and append at `pos`
		uint32_t v = B & lm;
		for (uint32_t sm = S; sm;) {
			const int s = 31 - __builtin_clz(sm);
			sm &= ~(1u << s);
			const uint32_t low = v & ((1u << s) - 1u);
			v = low | (s < 31 ? ((v >> (s + 1)) << s) : 0u);
		}
		{
			const int o = pos & 31;
			const unsigned long long t = (unsigned long long)v << o;
			cw |= (uint32_t)t;
			if (o + a >= 32) { data[DEC_LANES * wi] = cw; cw = (uint32_t)(t >> 32); wi++; }
		}
		// thresholds the position passes in this window, in order; the closing symbol itself is exempt (AIS.cpp:`!close && abort`)
		int abort_at = -1;
		const int np1 = pos + a; // position behind the window
		if (pos < 30 && np1 >= 30) {
			const int sy = dec_select_bit(K, 30 - pos);
			if (!(C && sy == e)) {
				const uint32_t w0 = wi == 0 ? cw : data[0];
				abort_pos = dec_abort_position((int)((w0 & 255u) >> 2));
				if (abort_pos == 30) abort_at = sy;
			}
		}
		if (abort_at < 0 && pos < 62 && np1 >= 62) {
			const int sy = dec_select_bit(K, 62 - pos);
			if (!(C && sy == e)) {
				const uint32_t w0 = data[0], w1 = wi == 1 ? cw : data[DEC_LANES];
				if ((((w0 >> 8) & 255u) << 22 | ((w0 >> 16) & 255u) << 14 | (w0 >> 24) << 6 | (w1 & 255u) >> 2) > 999999999u) abort_at = sy;
			}
		}
		if (abort_at < 0 && abort_pos > 30 && pos < abort_pos && np1 >= abort_pos) {
			const int sy = dec_select_bit(K, abort_pos - pos);
			if (!(C && sy == e)) abort_at = sy;
		}
		if (abort_at < 0 && pos < DEC_MAX_FRAME && np1 >= DEC_MAX_FRAME) {
			const int sy = dec_select_bit(K, DEC_MAX_FRAME - pos);
			if (!(C && sy == e)) abort_at = sy;
		}
		if (abort_at >= 0) { end = g + abort_at; return 0; }
		pos = np1;
		dprev = (D >> (nv - 1)) & 1u; // (a closed window ends the run: dprev / lastB are then set below)
		if (C) {
			end = g + e;
			data[DEC_LANES * wi] = cw;
			if (pos - 7 < 16 || dec_crc_bits(data, pos - 7, tab) != (uint32_t)(uint16_t)~0x0F47) return 0;
			flags = 1;
			dprev = (D >> e) & 1u; lastB = 1u; osc = 0; last_stuffed = 0;
			break;
		}
		lastB = (B >> (nv - 1)) & 1u;
		last_stuffed = (int)((S >> (nv - 1)) & 1u);
		{ // ones at the end of the window
			const uint32_t top = ~(B << (32 - nv)); // (nv >= 1)
			const int t = top ? __builtin_clz(top) : 32;
			osc = t >= nv ? osc + nv : t;
		}
		g += nv;
	}
	if (flags == 2 && g == g_first) return 2; // (nothing stepped: the state is the one that came in)
	// ---- the state the step would have left
	// (cwi, cw): the word of the position at which the last symbol was stored -- `pos` itself for a stuffed zero, pos - 1 otherwise.
	// Where the position has just crossed into a word that no symbol has touched, the step still holds the full one.
	const int last_pos = last_stuffed ? pos : pos - 1;
	const int cwi = last_pos >> 5;
	if (cwi != wi) cw = data[DEC_LANES * cwi];
	else data[DEC_LANES * wi] = cw;
	const int last_sym = flags == 1 ? end : n - 1;
	float level = r.level;
	if (lrow) { // (in symbol order, like the step; eight loads in flight.  No level row: tag.sample_lvl is never set in that engine, the sum stays 0)
		int i = g_first;
		if (lvl_shift) {
			if (i == 0 && i <= last_sym) { level = level + lvl_first; i = 1; }
			lrow -= 1;
		}
		for (; i + 8 <= last_sym + 1; i += 8) {
			float l[8];
			for (int e = 0; e < 8; e++) l[e] = lrow[i + e];
			for (int e = 0; e < 8; e++) level = level + l[e];
		}
		for (; i <= last_sym; i++) level = level + lrow[i];
	}
	r.level = level;
	r.state = DST_DATAFCS; r.position = pos; r.osc = osc; r.prev = (int)dprev; r.lastBit = (int)lastB;
	r.cw = cw; r.cwi = cwi; r.abort_pos = abort_pos;
	if (LAZY && flags == 2) return flags;
	r.crc = dec_crc_bits(data, pos >= 7 ? pos - 7 : 0, tab);
	uint32_t tail = 0;
	for (int k = 0; k < 7; k++) {
		const int p = pos - 1 - k;
		if (p >= 0) tail |= ((data[DEC_LANES * (p >> 5)] >> (p & 31)) & 1u) << k;
	}
	r.tail = tail;
	return flags;
}

// ------------------------------------------------------------------------------------------
// k7e_scan: where could a frame start?  A decoder in TRAINING leaves it at the first symbol with alt == 0 after more than four
// alternations (Marine/AIS.h:109-119) -- a pattern of seven decisions.  Every such candidate is classified from the bits behind
// it: the start flag fails (K7E_FAIL, at symbol c + off) or a frame opens / the block ends first (K7E_RUN).
//   event word: c | kind << 13 | off << 15 | slot << 19
// A FAIL candidate only matters through the candidates of the same decoder that it blocks (those up to five symbols behind its
// failing symbol) and through the alternation count at the end of the block; alone in the noise -- the usual case, one every
// ~64 symbols -- it has no effect whatever the siblings do, and is not listed.
// The row is scanned in segments of words, one lane each (dec_scan_words); a segment's last FAIL stays pending until the next
// candidate behind the segment is known.  dec_scan_row is the same scan as one loop (the first implementation; the host test
// compares the two, the device runs the segments).
// ------------------------------------------------------------------------------------------
enum { K7E_FAIL = 0, K7E_RUN = 1 };
constexpr int DEC_SCAN_MAXW = 16; // words per segment at most (16 segments: rows of up to 8192 symbols)
constexpr int DEC_SCAN_INF = 1 << 30;
struct ScanSeg { int nev, nrun, first_c; uint32_t pend; int pend_until; };

// one candidate: (kind, off) from the NRZI bits behind it; BB: bits of this word (low half) and of the next one
DEC_HD uint32_t dec_scan_classify(unsigned long long BB, int i, int c, int n) {
	// STARTFLAG (AIS.h:121-141): entered with position 3 (Bit == 1) or 1 (Bit == 0); ones up to position 7, then a zero
	const int need = ((BB >> i) & 1ull) ? 4 : 6;
	const unsigned long long seq = BB >> (i + 1);
	int t = __builtin_ctzll(~seq); // ones that follow the candidate (need <= 6 of them are looked at)
	t = t < 8 ? t : 8;
	const int avail = n - (c + 1); // symbols of this block behind the candidate
	int kind, off = 0;
	if (t < need) { // a zero where a one was needed, at c + 1 + t
		if (t < avail) { kind = K7E_FAIL; off = 1 + t; } else kind = K7E_RUN; // (not decided inside this block)
	} else if (need < avail) { // the symbol at position 7 exists: it must be a zero
		if (t == need) kind = K7E_RUN; else { kind = K7E_FAIL; off = 1 + need; }
	} else kind = K7E_RUN;
	return (uint32_t)c | ((uint32_t)kind << 13) | ((uint32_t)off << 15);
}

// W[i] = word w_begin + i of the row for i <= cnt (0 behind the row's last word); prevD / prevB / prevA: the word in front of the
// segment (decisions, NRZI bits, alternations -- of the latter two only the top bits matter).  emit(e32) for every event that is
// listed for sure, in order, RUN events numbered from 0 in bits 19..
template <class Emit>
DEC_HD void dec_scan_words(const uint32_t (&W)[DEC_SCAN_MAXW + 1], int cnt, int w_begin, int n, uint32_t prevD, uint32_t prevB, uint32_t prevA,
                           ScanSeg& sg, Emit emit) {
	sg.nev = 0; sg.nrun = 0; sg.first_c = DEC_SCAN_INF; sg.pend = 0u; sg.pend_until = -1;
#if defined(__HIPCC__)
#pragma unroll
#endif
	for (int k = 0; k < DEC_SCAN_MAXW; k++) {
		if (k < cnt) {
			const int w = w_begin + k;
			const uint32_t D = W[k], Dn = W[k + 1];
			const uint32_t B = ~(D ^ ((D << 1) | (prevD >> 31)));
			const uint32_t A = B ^ ((B << 1) | (prevB >> 31));
			const int nv = n - 32 * w < 32 ? n - 32 * w : 32; // valid symbols in this word
			const uint32_t valid = nv < 32 ? ((1u << nv) - 1u) : 0xFFFFFFFFu;
			const unsigned long long X = ((unsigned long long)A << 32) | prevA;
			const unsigned long long R = X & (X >> 1) & (X >> 2) & (X >> 3) & (X >> 4);
			uint32_t cand = ~A & (uint32_t)(R >> 27) & valid; // alt == 0 with the five symbols before it all alternating
			if (cand) {
				const uint32_t Bn = ~(Dn ^ ((Dn << 1) | (D >> 31)));
				const unsigned long long BB = ((unsigned long long)Bn << 32) | B;
				while (cand) {
					const int i = __builtin_ctz(cand);
					cand &= cand - 1;
					const int c = 32 * w + i;
					uint32_t e32 = dec_scan_classify(BB, i, c, n);
					if (sg.first_c == DEC_SCAN_INF) sg.first_c = c;
					if (sg.pend_until >= 0 && c < sg.pend_until) { emit(sg.pend); sg.nev++; }
					sg.pend_until = -1;
					if (((e32 >> 13) & 3u) == K7E_FAIL) { sg.pend = e32; sg.pend_until = c + (int)((e32 >> 15) & 15u) + 6; }
					else { e32 |= (uint32_t)sg.nrun << 19; sg.nrun++; emit(e32); sg.nev++; }
				}
			}
			prevD = D; prevB = B; prevA = A;
		}
	}
}

// carries of a segment that starts at word w_begin > 0, from the word in front of it (their low bits are not exact, and never used)
DEC_HD void dec_scan_carry(uint32_t Dp, uint32_t& prevD, uint32_t& prevB, uint32_t& prevA) {
	const uint32_t Bp = ~(Dp ^ (Dp << 1));
	prevD = Dp; prevB = Bp; prevA = Bp ^ (Bp << 1);
}
