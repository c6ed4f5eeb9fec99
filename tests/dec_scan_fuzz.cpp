// tests/dec_scan_fuzz.cpp -- the segmented candidate scan of the device frame decoders (ais-catcher_amd/csrc/dec_core.h:
// dec_scan_words, one lane per segment of a decoder's row in k7e_scan) against the same scan as one loop over the row (the first
// implementation of k7e_scan, pinned by the GPU parity tests), on random rows.  Host build of the device header, test only.
//   g++ -O2 -std=c++17 -I ais-catcher_amd/csrc tests/dec_scan_fuzz.cpp -o /tmp/dec_scan_fuzz && /tmp/dec_scan_fuzz [trials] [seed]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "dec_core.h"

constexpr int EVCAP = 1024, OPENCAP = 128, SEGS = 16;
constexpr uint32_t CONT = 0xFFFFu;

struct Result { std::vector<uint32_t> ev; std::vector<uint16_t> oc; int overflow = 0; };
struct State { int state, prev, lastBit, position; };

static int training_pos(const State& st) { return st.state == DST_TRAINING ? (st.position < 5 ? st.position : 5) : 0; }

// the scan as one loop (k7e_scan as it was)
static Result scan_row(const std::vector<uint32_t>& brow, int n, const State& st) {
	Result r;
	const int nw = (n + 31) >> 5;
	int nrun = 0;
	if (st.state != DST_TRAINING) { r.ev.push_back(0u | (K7E_RUN << 13) | (0u << 19)); r.oc.push_back((uint16_t)CONT); nrun++; }
	uint32_t prevD = st.prev ? 0x80000000u : 0u, prevB = st.lastBit ? 0x80000000u : 0u;
	const int p5 = training_pos(st);
	uint32_t prevA = p5 ? (0xFFFFFFFFu << (32 - p5)) : 0u;
	uint32_t pend = 0; int pend_until = -1;
	const auto flush_pend = [&](int next_c) {
		if (pend_until >= 0 && next_c < pend_until) { if ((int)r.ev.size() < EVCAP) r.ev.push_back(pend); else r.overflow |= 2; }
		pend_until = -1;
	};
	for (int w = 0; w < nw; w++) {
		const uint32_t D = brow[w], Dn = w + 1 < nw ? brow[w + 1] : 0u;
		const uint32_t B = ~(D ^ ((D << 1) | (prevD >> 31)));
		const uint32_t A = B ^ ((B << 1) | (prevB >> 31));
		const int nv = n - 32 * w < 32 ? n - 32 * w : 32;
		const uint32_t valid = nv < 32 ? ((1u << nv) - 1u) : 0xFFFFFFFFu;
		const unsigned long long X = ((unsigned long long)A << 32) | prevA;
		const unsigned long long R = X & (X >> 1) & (X >> 2) & (X >> 3) & (X >> 4);
		uint32_t cand = ~A & (uint32_t)(R >> 27) & valid;
		if (cand) {
			const uint32_t Bn = ~(Dn ^ ((Dn << 1) | (D >> 31)));
			const unsigned long long BB = ((unsigned long long)Bn << 32) | B;
			while (cand) {
				const int i = __builtin_ctz(cand);
				cand &= cand - 1;
				const int c = 32 * w + i;
				const int need = ((BB >> i) & 1ull) ? 4 : 6;
				const unsigned long long seq = BB >> (i + 1);
				int t = __builtin_ctzll(~seq);
				t = t < 8 ? t : 8;
				const int avail = n - (c + 1);
				int kind, off = 0;
				if (t < need) {
					if (t < avail) { kind = K7E_FAIL; off = 1 + t; } else kind = K7E_RUN;
				} else if (need < avail) {
					if (t == need) kind = K7E_RUN; else { kind = K7E_FAIL; off = 1 + need; }
				} else kind = K7E_RUN;
				int slot = 0;
				if (kind == K7E_RUN) {
					if (nrun < OPENCAP) { slot = nrun; r.oc.push_back((uint16_t)c); nrun++; } else { r.overflow |= 1; kind = K7E_FAIL; off = 1; }
				}
				const uint32_t e32 = (uint32_t)c | ((uint32_t)kind << 13) | ((uint32_t)off << 15) | ((uint32_t)slot << 19);
				flush_pend(c);
				if (kind == K7E_FAIL) { pend = e32; pend_until = c + off + 6; }
				else if ((int)r.ev.size() < EVCAP) r.ev.push_back(e32);
				else r.overflow |= 2;
			}
		}
		prevD = D; prevB = B; prevA = A;
	}
	flush_pend(pend_until > n ? n : 1 << 30);
	return r;
}

// the scan in segments, combined the way k7e_scan combines its lanes
static Result scan_segments(const std::vector<uint32_t>& brow, int n, const State& st) {
	Result r;
	const int nw = (n + 31) >> 5, wps = (nw + SEGS - 1) / SEGS;
	ScanSeg sg[SEGS];
	std::vector<uint32_t> list[SEGS];
	for (int l = 0; l < SEGS; l++) {
		const int w_begin = l * wps, cnt = w_begin >= nw ? 0 : (nw - w_begin < wps ? nw - w_begin : wps);
		uint32_t W[DEC_SCAN_MAXW + 1] = {};
		for (int k = 0; k <= cnt && k <= DEC_SCAN_MAXW; k++) W[k] = w_begin + k < nw ? brow[w_begin + k] : 0u;
		uint32_t prevD, prevB, prevA;
		if (w_begin == 0) {
			prevD = st.prev ? 0x80000000u : 0u; prevB = st.lastBit ? 0x80000000u : 0u;
			const int p5 = training_pos(st);
			prevA = p5 ? (0xFFFFFFFFu << (32 - p5)) : 0u;
		} else dec_scan_carry(brow[w_begin - 1 < nw ? w_begin - 1 : nw - 1], prevD, prevB, prevA);
		dec_scan_words(W, cnt, w_begin, n, prevD, prevB, prevA, sg[l], [&](uint32_t e) { list[l].push_back(e); });
	}
	const bool cont = st.state != DST_TRAINING;
	int ev_off = cont ? 1 : 0, run_off = cont ? 1 : 0;
	if (cont) { r.ev.push_back(0u | (K7E_RUN << 13)); r.oc.push_back((uint16_t)CONT); }
	r.ev.resize(EVCAP + 8, 0xDEADBEEFu); r.oc.resize(OPENCAP + 8, 0xDEAD);
	for (int l = 0; l < SEGS; l++) {
		int nf = DEC_SCAN_INF;
		for (int k = l + 1; k < SEGS; k++) if (sg[k].first_c < nf) nf = sg[k].first_c;
		const bool trailing = sg[l].pend_until >= 0 && (nf != DEC_SCAN_INF ? nf < sg[l].pend_until : sg[l].pend_until > n);
		for (int i = 0; i < sg[l].nev; i++) {
			uint32_t e = list[l][i];
			if (((e >> 13) & 3u) == K7E_RUN) {
				const int slot = (int)(e >> 19) + run_off;
				if (slot >= OPENCAP) { r.overflow |= 1; e = (e & 0x1FFFu) | (K7E_FAIL << 13) | (1u << 15); }
				else { r.oc[slot] = (uint16_t)(e & 0x1FFFu); e = (e & 0x7FFFFu) | ((uint32_t)slot << 19); }
			}
			if (ev_off + i < EVCAP) r.ev[ev_off + i] = e; else r.overflow |= 2;
		}
		if (trailing) { if (ev_off + sg[l].nev < EVCAP) r.ev[ev_off + sg[l].nev] = sg[l].pend; else r.overflow |= 2; }
		ev_off += sg[l].nev + (trailing ? 1 : 0);
		run_off += sg[l].nrun;
	}
	r.ev.resize(ev_off < EVCAP ? ev_off : EVCAP); r.oc.resize(run_off < OPENCAP ? run_off : OPENCAP);
	return r;
}

int main(int argc, char** argv) {
	const long trials = argc > 1 ? atol(argv[1]) : 100000;
	const unsigned seed = argc > 2 ? (unsigned)atol(argv[2]) : 1u;
	std::mt19937 rng(seed);
	const auto rnd = [&](int lo, int hi) { return lo + (int)(rng() % (unsigned)(hi - lo + 1)); };
	long n_ev = 0, n_run = 0, n_fail = 0, n_over = 0;
	for (long t = 0; t < trials; t++) {
		const int n = rnd(0, 9) == 0 ? rnd(1, 100) : rnd(0, 3) == 0 ? rnd(1, 8191) : rnd(4000, 5200);
		const int nw = (n + 31) >> 5;
		// decisions: NRZI bits with a chosen share of alternations (preambles), ones (flags) and noise
		const int p_alt = rnd(0, 3) == 0 ? rnd(60, 97) : 50, p_one = rnd(20, 80);
		std::vector<uint32_t> brow(nw + 1, 0u);
		int dd = rnd(0, 1), bit = rnd(0, 1);
		const int p_flag = rnd(0, 4) == 0 ? rnd(1, 12) : rnd(0, 2) ? 1 : 0; // per cent of symbols at which a preamble + start flag is inserted
		std::vector<int> script;
		for (int g = 0; g < n; g++) {
			if (script.empty() && rnd(0, 99) < p_flag) { // alternations, then 0 1 1 1 1 1 1 0 (sometimes damaged)
				for (int k = rnd(4, 9); k > 0; k--) script.push_back(k & 1);
				const int flag[8] = { 0, 1, 1, 1, 1, 1, 1, 0 };
				for (int k = 7; k >= 0; k--) script.insert(script.begin(), 0); // placeholder, filled below (script is consumed from the back)
				for (int k = 0; k < 8; k++) script[7 - k] = flag[k];
				if (rnd(0, 5) == 0) script[rnd(0, 7)] ^= 1;
			}
			if (!script.empty()) { bit = script.back(); script.pop_back(); }
			else {
				const int mode = rnd(0, 99);
				if (mode < p_alt) bit = !bit; else bit = rnd(0, 99) < p_one;
			}
			dd = bit ? dd : !dd;
			brow[g >> 5] |= (uint32_t)dd << (g & 31);
		}
		if (rnd(0, 1) && (n & 31)) brow[nw - 1] |= 0xFFFFFFFFu << (n & 31); // stale bits behind the row's end
		State st;
		st.state = rnd(0, 3) == 0 ? (rnd(0, 1) ? DST_STARTFLAG : DST_DATAFCS) : DST_TRAINING;
		st.prev = rnd(0, 1); st.lastBit = rnd(0, 1); st.position = rnd(0, 9);
		const Result a = scan_row(brow, n, st), b = scan_segments(brow, n, st);
		if (a.overflow || b.overflow) { // (the device raises an error for such a block; only the flag has to agree)
			n_over++;
			if (!a.overflow != !b.overflow) { printf("MISMATCH trial %ld: overflow %d / %d\n", t, a.overflow, b.overflow); return 1; }
			continue;
		}
		if (a.ev != b.ev || a.oc != b.oc) {
			printf("MISMATCH trial %ld seed %u (n %d): %zu / %zu events, %zu / %zu runs\n", t, seed, n, a.ev.size(), b.ev.size(), a.oc.size(), b.oc.size());
			for (size_t i = 0; i < a.ev.size() && i < b.ev.size(); i++)
				if (a.ev[i] != b.ev[i]) { printf("  first difference at event %zu: %08x / %08x\n", i, a.ev[i], b.ev[i]); break; }
			return 1;
		}
		n_ev += (long)a.ev.size(); n_run += (long)a.oc.size();
		for (uint32_t e : a.ev) n_fail += ((e >> 13) & 3u) == K7E_FAIL;
	}
	printf("dec_scan_fuzz: %ld rows, %ld events (%ld listed failures), %ld runs, %ld rows over capacity: all equal\n", trials, n_ev, n_fail, n_run, n_over);
	return 0;
}
