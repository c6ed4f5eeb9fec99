// tests/dec_core_fuzz.cpp -- the word-parallel frame evaluator of ais-catcher_amd/csrc/dec_core.h (what k7e_sim runs on the
// device) against the symbol-by-symbol step of the same header (whose semantics the GPU parity tests pin to the reference's
// AIS::Decoder), on random streams cut into random blocks.  Host build of the device header, test infrastructure only.
//   g++ -O2 -std=c++17 -I ais-catcher_amd/csrc tests/dec_core_fuzz.cpp -o /tmp/dec_core_fuzz && /tmp/dec_core_fuzz [trials] [seed]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "dec_core.h"

constexpr int WORDS = 36;
constexpr int LANE = 7;

struct Dec {
	DecReg r;
	uint32_t tile[WORDS * DEC_LANES];
	uint32_t* data() { return tile + LANE; }
};

static uint16_t g_tab[256];
static bool g_lazy = false; // third argument "lazy": the evaluator as K7b's boundary tasks call it (CRC register / tail formed on demand)

static void fresh(Dec& d, int prev, int lastBit) {
	memset(&d, 0, sizeof d);
	d.r.state = DST_TRAINING; d.r.position = 5; d.r.osc = 0; d.r.level = 0.0f; d.r.start_idx = 0;
	d.r.prev = prev; d.r.lastBit = lastBit;
	d.r.crc = 0xFFFFu; d.r.cw = 0u; d.r.cwi = 0; d.r.tail = 0u; d.r.abort_pos = 0;
}

struct Block { std::vector<uint32_t> bits; std::vector<float> lvl; int n; long long first_group; int shift; float first; };

// the run as k7e_sim used to make it: one dec_step per symbol
static int run_steps(Dec& d, const Block& b, int c, int& end) {
	int g = c, flags = 2;
	end = b.n;
	while (g < b.n) {
		const int dbit = (int)((b.bits[g >> 5] >> (g & 31)) & 1u);
		const float slvl = b.shift ? (g == 0 ? b.first : b.lvl[g - 1]) : b.lvl[g]; // (ModelChallenger's FM0..FM3: the previous group's level)
		const bool found = dec_step<false>(d.r, dbit, slvl, 5 * (b.first_group + g) + 2, d.data());
		if (found) { end = g; flags = 1; break; }
		if (d.r.state == DST_TRAINING) { end = g; flags = 0; break; }
		g++;
	}
	if (flags == 2) d.data()[DEC_LANES * d.r.cwi] = d.r.cw;
	return flags;
}

// the run as k7e_sim makes it now: steps up to the frame's first symbol, then the evaluator
static int run_words(Dec& d, const Block& b, int c, int& end) {
	int g = c;
	end = b.n;
	while (g < b.n && d.r.state != DST_DATAFCS) {
		const int dbit = (int)((b.bits[g >> 5] >> (g & 31)) & 1u);
		dec_step<false>(d.r, dbit, 0.0f, 5 * (b.first_group + g) + 2, d.data());
		if (d.r.state == DST_TRAINING) { end = g; return 0; }
		g++;
	}
	if (d.r.state != DST_DATAFCS) { d.data()[DEC_LANES * d.r.cwi] = d.r.cw; return 2; }
	const int flags = g_lazy ? dec_run_frame<true>(d.r, d.data(), b.bits.data(), b.lvl.data(), g, b.n, g_tab, end, b.shift, b.first)
	                         : dec_run_frame<false>(d.r, d.data(), b.bits.data(), b.lvl.data(), g, b.n, g_tab, end, b.shift, b.first);
	if (flags == 2) d.data()[DEC_LANES * d.r.cwi] = d.r.cw;
	return flags;
}

static bool same_state(Dec& a, Dec& b, int flags, const char** what) {
#define CMP(f) if (a.r.f != b.r.f) { *what = #f; return false; }
	CMP(position) CMP(start_idx)
	if (memcmp(&a.r.level, &b.r.level, 4)) { *what = "level"; return false; }
	const int nwords = flags == 1 ? (a.r.position + 31) / 32 : a.r.cwi + 1;
	if (flags == 2) { CMP(state) CMP(lastBit) CMP(prev) CMP(osc) CMP(crc) CMP(cw) CMP(tail) CMP(cwi) CMP(abort_pos) }
	for (int w = 0; w < nwords; w++)
		if (a.data()[DEC_LANES * w] != b.data()[DEC_LANES * w]) { *what = "data"; return false; }
	return true;
#undef CMP
}

static uint16_t crc16(const std::vector<int>& bits) {
	uint32_t c = 0xFFFFu;
	for (int b : bits) c = (((uint32_t)b ^ c) & 1u) ? ((c >> 1) ^ 0x8408u) : (c >> 1);
	return (uint16_t)~c;
}

int main(int argc, char** argv) {
	const long trials = argc > 1 ? atol(argv[1]) : 200000;
	const unsigned seed = argc > 2 ? (unsigned)atol(argv[2]) : 1u;
	g_lazy = argc > 3 && !strcmp(argv[3], "lazy");
	for (int i = 0; i < 256; i++) dec_crc_table_entry(i, g_tab);
	std::mt19937 rng(seed);
	const auto rnd = [&](int lo, int hi) { return lo + (int)(rng() % (unsigned)(hi - lo + 1)); };
	if (argc > 3 && !strcmp(argv[3], "lean")) {
		// dec_lean_idle / dec_lean_data (kv2_engine's two-wave form) against dec_step, symbol by symbol over long streams of noise, preambles,
		// flags and stuffed payloads with good and broken CRCs; after a completed message both decoders are reset like V2's resetDecoders().
		// Every field the lean pair maintains must agree after every symbol, the frame buffers wherever a message completes.
		long steps = 0, n_msg = 0, n_open = 0, n_abort_ = 0;
		for (long t = 0; t < trials; t++) {
			std::vector<int> nrzi;
			const int segs = rnd(1, 6);
			for (int sg = 0; sg < segs; sg++) {
				if (rnd(0, 2) == 0) {
					const int pone = rnd(0, 3) == 0 ? 80 : rnd(30, 60);
					for (int i = rnd(1, 400); i > 0; i--) nrzi.push_back(rnd(0, 99) < pone);
					continue;
				}
				for (int i = 0, pre = rnd(4, 30); i < pre; i++) nrzi.push_back(i & 1);
				const int flag[8] = { 0, 1, 1, 1, 1, 1, 1, 0 };
				for (int b : flag) nrzi.push_back(b);
				int L;
				switch (rnd(0, 6)) {
				case 6: L = rnd(0, 3); break;
				case 0: L = 168; break;
				case 1: L = 8 * rnd(1, 132); break;
				case 2: L = rnd(1, 1100); break;
				case 3: L = 424; break;
				case 4: L = rnd(1040, 1090); break;
				default: L = 8 * rnd(2, 60); break;
				}
				std::vector<int> pay(L);
				const int pone = rnd(0, 3) == 0 ? 80 : 50;
				for (int& b : pay) b = rnd(0, 99) < pone;
				if (rnd(0, 3) && L >= 8) { const int type = rnd(1, 27); for (int k = 0; k < 6; k++) pay[2 + k] = (type >> k) & 1; }
				if (rnd(0, 2) && L >= 40) for (int k = 34; k < 40; k++) pay[k] = 0;
				const uint16_t fcs = crc16(pay);
				for (int k = 0; k < 16; k++) pay.push_back((fcs >> k) & 1);
				if (rnd(0, 4) == 0) pay[rnd(0, (int)pay.size() - 1)] ^= 1;
				if (rnd(0, 30) == 0) pay.pop_back();
				int ones = 0;
				for (int b : pay) { nrzi.push_back(b); ones = b ? ones + 1 : 0; if (ones == 5) { nrzi.push_back(0); ones = 0; } }
				if (rnd(0, 9)) for (int b : flag) nrzi.push_back(b);
			}
			Dec a, b;
			fresh(a, rnd(0, 1), rnd(0, 1));
			a.r.position = rnd(0, 6);
			b = a;
			int p = a.r.prev;
			for (size_t i = 0; i < nrzi.size(); i++) {
				const int dd = nrzi[i] ? p : !p; p = dd;
				const float slvl = (float)rnd(0, 1 << 20) / 4096.0f;
				const long long sidx = 7 * (long long)i + 3;
				const int st_before = a.r.state;
				const bool fa = dec_step<false>(a.r, dd, slvl, sidx, a.data());
				bool fb = false;
				if (b.r.state == DST_DATAFCS) fb = dec_lean_data(b.r, dd, slvl, b.data(), g_tab);
				else n_open += dec_lean_idle(b.r, dd, sidx);
				steps++;
				const DecReg &x = a.r, &y = b.r;
				bool bad = fa != fb || x.state != y.state || x.lastBit != y.lastBit || x.prev != y.prev || x.position != y.position || x.start_idx != y.start_idx;
				if (x.state == DST_DATAFCS) bad = bad || memcmp(&x.level, &y.level, 4) || x.cw != y.cw || x.cwi != y.cwi || x.abort_pos != y.abort_pos || x.osc != y.osc;
				else if (!fa) { // outside a frame the lean pair's frame registers are don't-cares (kv2_engine_roles leaves arithmetic garbage there): scribble
					b.r.osc = rnd(0, 7); b.r.cw = (uint32_t)rng(); b.r.cwi = rnd(0, 1 << 20); b.r.level = (float)rnd(-1000, 1000); b.r.abort_pos = rnd(0, 2000);
				}
				if (fa && !bad) for (int w = 0; w < (x.position + 31) / 32; w++) bad = bad || a.data()[DEC_LANES * w] != b.data()[DEC_LANES * w];
				if (bad) { printf("MISMATCH lean trial %ld symbol %zu (state %d -> %d / %d, position %d / %d, found %d / %d)\n", t, i, st_before, x.state, y.state, x.position, y.position, (int)fa, (int)fb); return 1; }
				n_abort_ += st_before == DST_DATAFCS && !fa && x.state == DST_TRAINING;
				if (fa) { n_msg++; a.r.state = b.r.state = DST_TRAINING; a.r.position = b.r.position = 0; a.r.osc = 0; b.r.osc = rnd(0, 7); }
			}
		}
		printf("lean steps %ld, frames opened %ld, messages %ld, frames abandoned %ld: all equal\n", steps, n_open, n_msg, n_abort_);
		return 0;
	}
	if (argc > 3 && !strcmp(argv[3], "idle")) {
		// dec_step_idle (kv2_engine's branch-free step for a decoder that is not inside a frame) against dec_step, field by field: random
		// TRAINING / STARTFLAG states incl. the ones where a flag starts, continues, opens a frame or falls back, random stale frame
		// registers (a step that opens a frame must clear them exactly like dec_step), and `on` = 0 (nothing may change)
		long opened = 0, flagged = 0;
		for (long t = 0; t < trials; t++) {
			Dec a;
			a.r.state = rnd(0, 1) ? (int)DST_TRAINING : (int)DST_STARTFLAG;
			a.r.lastBit = rnd(0, 1); a.r.prev = rnd(0, 1);
			a.r.position = a.r.state == DST_STARTFLAG ? rnd(0, 9) : rnd(0, 12);
			a.r.osc = rnd(0, 7); a.r.level = (float)rnd(0, 1000) * 0.125f; a.r.start_idx = rnd(0, 1 << 30);
			a.r.crc = (uint32_t)rnd(0, 0xFFFF); a.r.cw = (uint32_t)rng(); a.r.tail = (uint32_t)rnd(0, 127); a.r.cwi = rnd(0, 35); a.r.abort_pos = rnd(0, 448);
			Dec b = a;
			const int dd = rnd(0, 1), on = rnd(0, 7) != 0;
			const long long sidx = rnd(0, 1 << 30);
			const float slvl = (float)rnd(0, 100);
			Dec ref = a;
			bool found = false;
			if (on) found = dec_step<false>(ref.r, dd, slvl, sidx, ref.data());
			dec_step_idle(b.r, dd, sidx, on);
			const DecReg &x = ref.r, &y = b.r;
			if (found || x.state != y.state || x.lastBit != y.lastBit || x.prev != y.prev || x.position != y.position || x.osc != y.osc || x.level != y.level ||
			    x.start_idx != y.start_idx || x.crc != y.crc || x.cw != y.cw || x.tail != y.tail || x.cwi != y.cwi || x.abort_pos != y.abort_pos) {
				printf("MISMATCH at trial %ld (state %d pos %d dd %d on %d)\n", t, a.r.state, a.r.position, dd, on);
				return 1;
			}
			opened += on && y.state == DST_DATAFCS;
			flagged += on && a.r.state == DST_TRAINING && y.state == DST_STARTFLAG;
		}
		printf("idle steps %ld, frames opened %ld, flags started %ld: all equal\n", trials, opened, flagged);
		return 0;
	}
	long n_found = 0, n_cont = 0, n_abort = 0, n_runs = 0;
	for (long t = 0; t < trials; t++) {
		// ---- a stream of NRZI bits
		std::vector<int> nrzi;
		const int mode = rnd(0, 5);
		int c_stream = -1;        // symbol at which the run is started (TRAINING, five alternations counted)
		bool start_in_frame = false;
		if (mode <= 1) { // a frame in the open: preamble, flag, payload (+ CRC), flag
			for (int i = rnd(0, 40); i > 0; i--) nrzi.push_back(rnd(0, 1));
			const int pre = rnd(6, 30);
			for (int i = 0; i < pre; i++) nrzi.push_back(i & 1);
			// flag 01111110: the run starts where two equal bits follow the alternations
			const int last = nrzi.back();
			std::vector<int> flag = { 0, 1, 1, 1, 1, 1, 1, 0 };
			if (last == 0) { /* ...0 then flag's 0: equal bits at the flag's first symbol */ }
			const int f0 = (int)nrzi.size();
			for (int b : flag) nrzi.push_back(b);
			c_stream = last == 0 ? f0 : f0 + 2;
			int L;
			switch (rnd(0, 6)) {
			case 6: L = rnd(0, 3); break;
			case 0: L = 168; break;
			case 1: L = 8 * rnd(1, 132); break;
			case 2: L = rnd(1, 1100); break;
			case 3: L = 424; break;
			case 4: L = rnd(1040, 1090); break;
			default: L = 8 * rnd(2, 60); break;
			}
			std::vector<int> pay(L);
			const int pone = rnd(0, 3) == 0 ? 80 : 50;
			for (int& b : pay) b = rnd(0, 99) < pone;
			if (rnd(0, 3) && L >= 8) { // a plausible type in the first byte (bits 2..7) more often than chance would have it
				const int type = rnd(1, 27);
				for (int k = 0; k < 6; k++) pay[2 + k] = (type >> k) & 1;
			}
			if (rnd(0, 2) && L >= 40) for (int k = 34; k < 40; k++) pay[k] = 0; // MMSI below 2^24 << 6: valid
			const uint16_t fcs = crc16(pay);
			for (int k = 0; k < 16; k++) pay.push_back((fcs >> k) & 1);
			if (rnd(0, 4) == 0) pay[rnd(0, (int)pay.size() - 1)] ^= 1; // broken
			if (rnd(0, 30) == 0) pay.pop_back(); // one bit short
			int ones = 0;
			for (int b : pay) {
				nrzi.push_back(b);
				ones = b ? ones + 1 : 0;
				if (ones == 5) { nrzi.push_back(0); ones = 0; }
			}
			if (rnd(0, 9)) for (int b : flag) nrzi.push_back(b);
			for (int i = rnd(0, 300); i > 0; i--) nrzi.push_back(rnd(0, 1));
		} else { // noise of some density, entered inside a frame or at a random candidate
			const int pone = mode == 2 ? 30 : mode == 3 ? 45 : mode == 4 ? 60 : 80;
			const int len = rnd(1, 2600);
			for (int i = 0; i < len; i++) nrzi.push_back(rnd(0, 99) < pone);
			start_in_frame = rnd(0, 3) != 0;
			c_stream = start_in_frame ? 0 : rnd(0, len - 1);
		}
		const int total = (int)nrzi.size();
		std::vector<int> dd(total);
		const int prev0 = rnd(0, 1);
		{ int p = prev0; for (int i = 0; i < total; i++) { dd[i] = nrzi[i] ? p : !p; p = dd[i]; } }
		std::vector<float> lvl(total);
		for (float& l : lvl) l = (float)rnd(0, 1 << 20) / 4096.0f + (rnd(0, 7) == 0 ? 1e-3f : 0.0f);

		// ---- initial decoders
		Dec a, b;
		const int cprev = c_stream > 0 ? dd[c_stream - 1] : prev0;
		const int clast = c_stream > 0 ? nrzi[c_stream - 1] : rnd(0, 1);
		fresh(a, cprev, clast);
		if (start_in_frame) { a.r.state = DST_DATAFCS; a.r.position = 0; a.r.start_idx = 12345; }
		b = a;

		// ---- blocks
		int at = c_stream; // next stream symbol
		bool cont = false;
		long long first_group = rnd(0, 1 << 20);
		n_runs++;
		int nblocks = 0;
		const int shifted = rnd(0, 2) == 0;
		while (at < total) {
			// the block starts at stream symbol s0 <= at (the run's first block: the candidate is somewhere inside it)
			const int lead = cont ? 0 : rnd(0, at < 70 ? at : 70);
			const int s0 = at - lead;
			const int bl = rnd(0, 5) == 0 ? rnd(1, 40) : rnd(1, 700);
			const int n = lead + bl < total - s0 ? lead + bl : total - s0;
			Block blk;
			blk.n = n; blk.first_group = first_group;
			blk.shift = shifted; blk.first = (float)rnd(0, 1 << 16) / 256.0f;
			blk.bits.assign((n + 31) / 32 + 1, 0u);
			blk.lvl.assign(lvl.begin() + s0, lvl.begin() + s0 + n);
			for (int i = 0; i < n; i++) blk.bits[i >> 5] |= (uint32_t)dd[s0 + i] << (i & 31);
			if (rnd(0, 1)) blk.bits[(n + 31) / 32 - 1] |= n & 31 ? (0xFFFFFFFFu << (n & 31)) : 0u; // stale bits behind the block's end
			int ea, eb;
			const int fa = run_steps(a, blk, lead, ea);
			const int fb = run_words(b, blk, lead, eb);
			const char* what = "";
			Dec bc = b; // (lazy: the stale registers stay in b -- the next block's call must not depend on them)
			if (g_lazy && fb == 2) dec_fix_crc_tail(bc.r, bc.data(), g_tab);
			if (fa != fb || ea != eb || (fa != 0 && !same_state(a, bc, fa, &what))) {
				printf("MISMATCH trial %ld seed %u block %d (mode %d, lead %d, n %d): flags %d / %d, end %d / %d, %s (position %d / %d)\n", t, seed,
				       nblocks, mode, lead, n, fa, fb, ea, eb, what, a.r.position, b.r.position);
				return 1;
			}
			if (fa == 1) { n_found++; break; }
			if (fa == 0) { n_abort++; break; }
			n_cont++;
			cont = true;
			at = s0 + n;
			first_group += n;
			nblocks++;
		}
	}
	printf("dec_core_fuzz: %ld runs, %ld messages, %ld ended otherwise, %ld block crossings: all equal\n", n_runs, n_found, n_abort, n_cont);
	return 0;
}
