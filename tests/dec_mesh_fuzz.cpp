// tests/dec_mesh_fuzz.cpp -- the event-driven frame decoders as an algorithm (candidate scan, one run per possible frame, the walk
// that decides in the reference's time order which runs really happened and what the Reset of a sibling cuts short: k7e_scan /
// k7e_sim / k7e_resolve of ais-catcher_amd/csrc/kernels.hip) against the decoders stepped symbol by symbol in the reference's
// order with their Reset mesh (Marine/AIS.cpp:33-49, DSP/Model.cpp:566-573, :658-674), on random multi-block streams: meshes of
// five (ModelDefault / ModelStandard) and of ten with the previous-group level of FM0..FM3 (ModelChallenger).
// The scan and the frame evaluator are the device's own code (dec_core.h); the run set-up and the walk are restated here from
// the kernels line by line -- this test checks the method (and would have to be changed with it), the GPU tests check the kernels.
//   g++ -O2 -std=c++17 -I ais-catcher_amd/csrc tests/dec_mesh_fuzz.cpp -o /tmp/dec_mesh_fuzz && /tmp/dec_mesh_fuzz [trials] [seed]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "dec_core.h"

constexpr int WORDS = 36, LANE = 5, EVCAP = 1024, OPENCAP = 128;
constexpr uint32_t CONT = 0xFFFFu;
static uint16_t g_tab[256];

struct Dec {
	DecReg r;
	uint32_t tile[WORDS * DEC_LANES];
	uint32_t* data() { return tile + LANE; }
	const uint32_t* data() const { return tile + LANE; }
};
static void reset_fresh(Dec& d) {
	memset(&d, 0, sizeof d);
	d.r.state = DST_TRAINING; d.r.crc = 0xFFFFu;
}

struct Frame { int o, e, position; uint32_t level; long long start_idx; std::vector<uint32_t> data; };
static bool operator==(const Frame& a, const Frame& b) {
	return a.o == b.o && a.e == b.e && a.position == b.position && a.level == b.level && a.start_idx == b.start_idx && a.data == b.data;
}

struct Row { std::vector<uint32_t> bits; int j; int lvl_shift; float lvl_first; }; // one decoder's decisions of the block
struct Block { int n; long long first_group; std::vector<Row> rows; std::vector<float> lvl; };

static int bit_of(const Row& r, int g) { return (int)((r.bits[g >> 5] >> (g & 31)) & 1u); }
static float level_of(const Block& b, const Row& r, int g) { return r.lvl_shift ? (g == 0 ? r.lvl_first : b.lvl[g - 1]) : b.lvl[g]; }
static Frame frame_of(int o, int e, const Dec& d) {
	Frame f; f.o = o; f.e = e; f.position = d.r.position; memcpy(&f.level, &d.r.level, 4); f.start_idx = d.r.start_idx;
	for (int w = 0; w < (d.r.position + 31) / 32; w++) f.data.push_back(d.data()[DEC_LANES * w]);
	if (!f.data.empty() && (d.r.position & 31)) f.data.back() &= (1u << (d.r.position & 31)) - 1u;
	return f;
}

// ---- the reference's way: every group, every decoder in order; a completed message resets all the others
static void mesh_steps(std::vector<Dec>& dec, const Block& b, std::vector<Frame>& out) {
	const int M = (int)dec.size();
	for (int g = 0; g < b.n; g++)
		for (int o = 0; o < M; o++) {
			const Row& r = b.rows[o];
			const bool found = dec_step<false>(dec[o].r, bit_of(r, g), level_of(b, r, g), 5 * (b.first_group + g) + r.j, dec[o].data());
			if (found) {
				out.push_back(frame_of(o, g, dec[o]));
				for (int k = 0; k < M; k++) { dec[k].r.state = DST_TRAINING; dec[k].r.position = 0; dec[k].r.osc = 0; } // NextState(TRAINING, 0) for all ten / five
			}
		}
}

// ---- the event-driven way
struct Slot { int end, flags; Dec s; };
struct Lists { std::vector<uint32_t> ev; std::vector<uint16_t> oc; bool overflow = false; };
static int training_pos(const DecReg& r) { return r.state == DST_TRAINING ? (r.position < 5 ? r.position : 5) : 0; }

static Lists scan(const Row& row, int n, const DecReg& st) { // k7e_scan (segments of the device header, combined as the kernel does)
	Lists L;
	const int nw = (n + 31) >> 5, wps = (nw + 15) >> 4;
	ScanSeg sg[16];
	std::vector<uint32_t> list[16];
	for (int l = 0; l < 16; l++) {
		const int w_begin = l * wps, cnt = w_begin >= nw ? 0 : (nw - w_begin < wps ? nw - w_begin : wps);
		uint32_t W[DEC_SCAN_MAXW + 1] = {};
		for (int k = 0; k <= cnt && k <= DEC_SCAN_MAXW; k++) W[k] = w_begin + k < nw ? row.bits[w_begin + k] : 0u;
		uint32_t prevD, prevB, prevA;
		if (w_begin == 0) {
			prevD = st.prev ? 0x80000000u : 0u; prevB = st.lastBit ? 0x80000000u : 0u;
			const int p5 = training_pos(st);
			prevA = p5 ? (0xFFFFFFFFu << (32 - p5)) : 0u;
		} else dec_scan_carry(cnt > 0 ? row.bits[w_begin - 1] : 0u, prevD, prevB, prevA);
		dec_scan_words(W, cnt, w_begin, n, prevD, prevB, prevA, sg[l], [&](uint32_t e) { list[l].push_back(e); });
	}
	const bool cont = st.state != DST_TRAINING;
	if (cont) { L.ev.push_back(0u | (K7E_RUN << 13)); L.oc.push_back((uint16_t)CONT); }
	int run_off = cont ? 1 : 0;
	for (int l = 0; l < 16; l++) {
		int nf = DEC_SCAN_INF;
		for (int k = l + 1; k < 16; k++) if (sg[k].first_c < nf) nf = sg[k].first_c;
		const bool trailing = sg[l].pend_until >= 0 && (nf != DEC_SCAN_INF ? nf < sg[l].pend_until : sg[l].pend_until > n);
		for (int i = 0; i < sg[l].nev; i++) {
			uint32_t e = list[l][i];
			if (((e >> 13) & 3u) == K7E_RUN) {
				const int slot = (int)(e >> 19) + run_off;
				if (slot >= OPENCAP) { L.overflow = true; continue; }
				if ((int)L.oc.size() <= slot) L.oc.resize(slot + 1);
				L.oc[slot] = (uint16_t)(e & 0x1FFFu); e = (e & 0x7FFFFu) | ((uint32_t)slot << 19);
			}
			L.ev.push_back(e);
		}
		if (trailing) L.ev.push_back(sg[l].pend);
		run_off += sg[l].nrun;
	}
	if ((int)L.ev.size() > EVCAP) L.overflow = true;
	return L;
}

static Slot sim(const Dec& st, const Block& b, const Row& row, uint32_t c0) { // one run of k7e_sim
	Slot s;
	Dec& d = s.s;
	const bool cont = c0 == CONT;
	const int c = cont ? 0 : (int)c0, n = b.n;
	const auto dd_at = [&](int g) -> int { return g < 0 ? st.r.prev : bit_of(row, g); };
	if (cont) d = st;
	else {
		reset_fresh(d);
		d.r.position = 5;
		d.r.prev = dd_at(c - 1);
		d.r.lastBit = c - 1 < 0 ? st.r.lastBit : (dd_at(c - 1) == (c - 2 < 0 ? st.r.prev : dd_at(c - 2)));
	}
	int g = c;
	s.end = n; s.flags = 2;
	bool running = true;
	while (running && d.r.state != DST_DATAFCS) {
		if (g >= n) { running = false; break; }
		dec_step<false>(d.r, bit_of(row, g), 0.0f, 5 * (b.first_group + g) + row.j, d.data());
		if (d.r.state == DST_TRAINING) { s.end = g; s.flags = 0; running = false; }
		g++;
	}
	if (running) s.flags = dec_run_frame(d.r, d.data(), row.bits.data(), b.lvl.data(), g, n, g_tab, s.end, row.lvl_shift, row.lvl_first);
	if (s.flags == 2) d.data()[DEC_LANES * d.r.cwi] = d.r.cw;
	return s;
}

// k7e_resolve: the walk over the decoders' events in (group, place in the order) order
static bool event_driven(std::vector<Dec>& dec, const Block& b, std::vector<Frame>& out) {
	const int M = (int)dec.size(), n = b.n;
	constexpr int INF = 1 << 26;
	std::vector<Lists> L(M);
	std::vector<std::vector<Slot>> slots(M);
	for (int o = 0; o < M; o++) {
		L[o] = scan(b.rows[o], n, dec[o].r);
		if (L[o].overflow) return false;
		for (uint16_t c0 : L[o].oc) slots[o].push_back(sim(dec[o], b, b.rows[o], c0));
	}
	struct W { int ptr = 0, free_at = 0, end_ = INF, slot_ = 0; bool busy = false, fnd = false; };
	std::vector<W> w(M);
	for (int o = 0; o < M; o++) w[o].free_at = dec[o].r.state == DST_TRAINING ? 5 - training_pos(dec[o].r) : 0;
	for (;;) {
		int mk = INF * 16, owner = -1;
		for (int o = 0; o < M; o++) {
			const int t = w[o].busy ? w[o].end_ : (w[o].ptr < (int)L[o].ev.size() ? (int)(L[o].ev[w[o].ptr] & 0x1FFFu) : INF);
			const int key = t < n ? t * 16 + o : INF * 16;
			if (key < mk) { mk = key; owner = o; }
		}
		if (owner < 0) break;
		W& me = w[owner];
		int bcast = -1;
		if (!me.busy) {
			const uint32_t e = L[owner].ev[me.ptr++];
			const int c = (int)(e & 0x1FFFu), kind = (int)((e >> 13) & 3u), off = (int)((e >> 15) & 15u), sl = (int)(e >> 19);
			if (c >= me.free_at) {
				me.busy = true; me.slot_ = sl;
				if (kind == K7E_FAIL) { me.end_ = c + off; me.fnd = false; }
				else {
					const Slot& s = slots[owner][sl];
					me.fnd = (s.flags & 1) != 0;
					me.end_ = (s.flags & 2) ? INF : s.end;
				}
			}
		} else {
			const int e = me.end_;
			me.busy = false;
			me.free_at = e + 6;
			if (me.fnd) { out.push_back(frame_of(owner, e, slots[owner][me.slot_].s)); bcast = (e << 4) | owner; }
		}
		if (bcast >= 0) {
			const int e = bcast >> 4, jw = bcast & 15;
			for (int o = 0; o < M; o++) {
				if (o == owner) continue;
				const int fa = e + (o > jw ? 5 : 6);
				if (w[o].busy) { w[o].busy = false; w[o].free_at = fa; }
				else if (fa > w[o].free_at) w[o].free_at = fa;
			}
		}
	}
	// state for the next block
	for (int o = 0; o < M; o++) {
		const DecReg r0 = dec[o].r;
		if (w[o].busy) { dec[o] = slots[o][w[o].slot_].s; continue; }
		const Row& row = b.rows[o];
		const auto dd_at = [&](int g) -> int { return g < 0 ? r0.prev : bit_of(row, g); };
		const auto bit_at = [&](int g) -> int { return g < 0 ? r0.lastBit : (dd_at(g) == dd_at(g - 1)); };
		const int pos0 = training_pos(r0);
		int run = 0;
		for (int g = n - 1; run < 5; g--) {
			if (g < 0) { run += pos0 < 5 - run ? pos0 : 5 - run; break; }
			if (bit_at(g) == bit_at(g - 1)) break;
			run++;
		}
		int lim = n + 5 - w[o].free_at;
		lim = lim < 0 ? 0 : lim;
		reset_fresh(dec[o]);
		dec[o].r.position = run < lim ? run : lim;
		dec[o].r.lastBit = n > 0 ? bit_at(n - 1) : r0.lastBit; dec[o].r.prev = n > 0 ? dd_at(n - 1) : r0.prev;
	}
	return true;
}

static bool same_state(const Dec& a, const Dec& b, const char** what) {
#define CMP(f) if (a.r.f != b.r.f) { *what = #f; return false; }
	CMP(state) CMP(lastBit) CMP(prev)
	if (a.r.state == DST_TRAINING) {
		if ((a.r.position < 5 ? a.r.position : 5) != (b.r.position < 5 ? b.r.position : 5)) { *what = "training position"; return false; }
		return true;
	}
	CMP(position) CMP(osc) CMP(start_idx)
	if (a.r.state == DST_STARTFLAG) return true; // (everything else is set when the frame opens)
	CMP(crc) CMP(cw) CMP(tail) CMP(cwi) CMP(abort_pos)
	if (memcmp(&a.r.level, &b.r.level, 4)) { *what = "level"; return false; }
	for (int w = 0; w < a.r.cwi; w++) if (a.data()[DEC_LANES * w] != b.data()[DEC_LANES * w]) { *what = "data"; return false; }
	return true;
#undef CMP
}

static uint16_t crc16(const std::vector<int>& bits) {
	uint32_t c = 0xFFFFu;
	for (int b : bits) c = (((uint32_t)b ^ c) & 1u) ? ((c >> 1) ^ 0x8408u) : (c >> 1);
	return (uint16_t)~c;
}

int main(int argc, char** argv) {
	const long trials = argc > 1 ? atol(argv[1]) : 20000;
	const unsigned seed = argc > 2 ? (unsigned)atol(argv[2]) : 1u;
	for (int i = 0; i < 256; i++) dec_crc_table_entry(i, g_tab);
	std::mt19937 rng(seed);
	const auto rnd = [&](int lo, int hi) { return lo + (int)(rng() % (unsigned)(hi - lo + 1)); };
	long n_frames = 0, n_blocks = 0, n_over = 0, n_cut = 0;
	for (long t = 0; t < trials; t++) {
		const int M = rnd(0, 1) ? 5 : 10;
		// ---- one transmitted stream of NRZI bits (noise, preambles, frames good and bad, back to back or far apart) ...
		std::vector<int> nrzi;
		const int total_target = rnd(200, 6000);
		while ((int)nrzi.size() < total_target) {
			const int what = rnd(0, 9);
			if (what < 4) for (int i = rnd(1, 120); i > 0; i--) nrzi.push_back(rnd(0, 1));
			else {
				for (int i = 0, pre = rnd(3, 26); i < pre; i++) nrzi.push_back(i & 1);
				const int flag[8] = { 0, 1, 1, 1, 1, 1, 1, 0 };
				for (int b : flag) nrzi.push_back(b);
				const int Lp = rnd(0, 4) == 0 ? rnd(1, 400) : 8 * rnd(2, 53);
				std::vector<int> pay(Lp);
				for (int& b : pay) b = rnd(0, 1);
				if (rnd(0, 4) && Lp >= 8) { const int type = rnd(1, 27); for (int k = 0; k < 6; k++) pay[2 + k] = (type >> k) & 1; }
				if (rnd(0, 2) && Lp >= 40) for (int k = 34; k < 40; k++) pay[k] = 0;
				const uint16_t fcs = crc16(pay);
				for (int k = 0; k < 16; k++) pay.push_back((fcs >> k) & 1);
				if (rnd(0, 5) == 0) pay[rnd(0, (int)pay.size() - 1)] ^= 1;
				int ones = 0;
				for (int b : pay) { nrzi.push_back(b); ones = b ? ones + 1 : 0; if (ones == 5) { nrzi.push_back(0); ones = 0; } }
				if (rnd(0, 9)) for (int b : flag) nrzi.push_back(b);
			}
		}
		const int total = (int)nrzi.size();
		// ---- ... as every decoder of the mesh sees it: its own decision errors, sometimes a symbol early or late
		std::vector<std::vector<int>> dd(M, std::vector<int>(total));
		const int err = rnd(0, 3) == 0 ? 0 : rnd(1, 40); // decision errors per 1000 symbols
		for (int o = 0; o < M; o++) {
			const int shift = rnd(0, 5) == 0 ? rnd(-1, 1) : 0;
			int p = rnd(0, 1);
			for (int i = 0; i < total; i++) {
				const int k = std::min(std::max(i + shift, 0), total - 1);
				int bit = nrzi[k];
				if (rnd(0, 999) < err) bit ^= 1;
				dd[o][i] = bit ? p : !p;
				p = dd[o][i];
			}
		}
		std::vector<float> lvl(total);
		for (float& l : lvl) l = (float)rnd(0, 1 << 18) / 1024.0f;

		std::vector<Dec> a(M), bb(M);
		for (int o = 0; o < M; o++) { reset_fresh(a[o]); a[o].r.prev = rnd(0, 1); a[o].r.lastBit = rnd(0, 1); a[o].r.position = rnd(0, 7); bb[o] = a[o]; }
		int at = 0;
		long long first_group = rnd(0, 1 << 20);
		float last_lvl = 0.0f;
		while (at < total) {
			const int n = std::min(total - at, rnd(0, 4) == 0 ? rnd(1, 60) : rnd(200, 3000));
			Block blk;
			blk.n = n; blk.first_group = first_group;
			blk.lvl.assign(lvl.begin() + at, lvl.begin() + at + n);
			blk.rows.resize(M);
			for (int o = 0; o < M; o++) {
				Row& r = blk.rows[o];
				r.bits.assign((n + 31) / 32 + 1, 0u);
				for (int i = 0; i < n; i++) r.bits[i >> 5] |= (uint32_t)dd[o][at + i] << (i & 31);
				if (M == 10) { r.j = o < 4 ? o : o == 9 ? 4 : o - 4; r.lvl_shift = o < 4; r.lvl_first = rnd(0, 1) ? last_lvl : (float)rnd(0, 1 << 12); }
				else { r.j = o; r.lvl_shift = 0; r.lvl_first = 0.0f; }
			}
			std::vector<Frame> fa, fb;
			mesh_steps(a, blk, fa);
			if (!event_driven(bb, blk, fb)) { n_over++; break; } // (more frame starts than the lists hold: the device falls back, nothing to compare)
			n_blocks++;
			std::stable_sort(fb.begin(), fb.end(), [](const Frame& x, const Frame& y) { return x.e * 16 + x.o < y.e * 16 + y.o; });
			if (!(fa == fb)) {
				printf("MISMATCH trial %ld seed %u (mesh of %d, block at %d, n %d): %zu / %zu messages\n", t, seed, M, at, n, fa.size(), fb.size());
				for (size_t i = 0; i < fa.size() || i < fb.size(); i++) {
					if (i < fa.size()) printf("  steps : decoder %d group %d position %d\n", fa[i].o, fa[i].e, fa[i].position);
					if (i < fb.size()) printf("  events: decoder %d group %d position %d\n", fb[i].o, fb[i].e, fb[i].position);
				}
				return 1;
			}
			for (int o = 0; o < M; o++) {
				const char* what = "";
				if (!same_state(a[o], bb[o], &what)) {
					printf("MISMATCH trial %ld seed %u (mesh of %d, block at %d, n %d): state of decoder %d differs in %s (state %d / %d, position %d / %d)\n", t,
					       seed, M, at, n, o, what, a[o].r.state, bb[o].r.state, a[o].r.position, bb[o].r.position);
					return 1;
				}
				if (a[o].r.state != DST_TRAINING) n_cut++;
			}
			n_frames += (long)fa.size();
			last_lvl = blk.lvl[n - 1];
			at += n; first_group += n;
		}
	}
	printf("dec_mesh_fuzz: %ld streams, %ld blocks, %ld messages, %ld decoder states carried inside a frame, %ld streams over capacity: all equal\n", trials,
	       n_blocks, n_frames, n_cut, n_over);
	return 0;
}
