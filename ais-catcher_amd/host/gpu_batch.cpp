// ais-catcher_amd/host/gpu_batch.cpp -- see gpu_batch.h
#include "gpu_batch.h"

#include <algorithm>
#include <chrono>
#include <stdexcept>
#include <string>

namespace aisamd {

GpuBatch::GpuBatch(const aisgpu_cfg& c) : cfg(c) {
	int rc = aisgpu_create(&cfg, &ctx);
	if (rc != AISGPU_OK) {
		std::string msg = std::string("GpuBatch: ") + aisgpu_strerror(rc);
		if (ctx) { msg += std::string(": ") + aisgpu_last_error(ctx); aisgpu_destroy(ctx); ctx = nullptr; }
		throw std::runtime_error(msg);
	}
	active = cfg.n_receivers;
	present.assign(cfg.n_receivers, 0);
	gone.assign(cfg.n_receivers, 0);
	frame_snap = std::make_shared<const std::vector<aisgpu_frame>>();
}

GpuBatch::~GpuBatch() { aisgpu_destroy(ctx); }

void GpuBatch::launch() {
	// run the whole batch for this block, copy the outputs back, release everyone (rows of receivers that are gone keep
	// whatever their staging rows held: receivers are closed systems, nobody reads those outputs)
	int st = AISGPU_OK;
	if (pipelined) { // the previous block's outputs first (the device has had the receivers' whole replay time to finish it), then start this one
		if (generation > 0) st = aisgpu_sync_outputs(ctx);
		if (st == AISGPU_OK && fed > 0) st = aisgpu_run(ctx);
	} else {
		st = aisgpu_run(ctx);
		if (st == AISGPU_OK) st = aisgpu_sync_outputs(ctx);
	}
	if (cfg.flags & AISGPU_FLAG_GPU_DECODE) { // this generation's frames, for as long as any receiver still reads them
		const aisgpu_frame* fr = nullptr;
		int nf = 0;
		if (st == AISGPU_OK && aisgpu_out_count(ctx) > 0) st = aisgpu_frames(ctx, &fr, &nf);
		frame_snap = std::make_shared<const std::vector<aisgpu_frame>>(fr, fr + (st == AISGPU_OK ? nf : 0));
	}
	gen_status[generation & 1] = st;
	arrived = 0;
	fed = 0;
	std::fill(present.begin(), present.end(), 0);
	generation++;
	cv.notify_all();
}

int GpuBatch::submitAndWait(int rx, const void* iq, int n_iq) {
	if (rx < 0 || rx >= cfg.n_receivers) return AISGPU_ERR_ARG;
	{
		std::lock_guard<std::mutex> lock(mtx);
		if (gone[rx]) return AISGPU_ERR_STATE;
	}
	// the copy into the pinned staging row happens OUTSIDE the batch lock: the receivers' threads copy their rows concurrently
	// (aisgpu_submit is thread safe for different rx)
	const int rc = iq ? aisgpu_submit(ctx, rx, iq, n_iq) : AISGPU_OK; // iq == nullptr: drain request of a pipelined batch
	std::unique_lock<std::mutex> lock(mtx);
	if (gone[rx]) return AISGPU_ERR_STATE; // (evicted meanwhile)
	if (rc != AISGPU_OK) { // this receiver's problem only (wrong block length ...): it leaves, the others go on
		gone[rx] = 1;
		active--;
		if (arrived > 0 && arrived == active) launch();
		return rc;
	}
	const long long my_gen = generation;
	present[rx] = 1;
	if (iq) fed++;
	if (++arrived == active) {
		launch(); // last receiver of this block
	} else {
		const auto done = [&] { return generation != my_gen; };
		if (timeout_ms <= 0) cv.wait(lock, done);
		else if (!cv.wait_for(lock, std::chrono::milliseconds(timeout_ms), done)) {
			// some receiver stopped delivering: evict everyone who has not handed in this block and run with the rest
			for (int r = 0; r < cfg.n_receivers; r++)
				if (!gone[r] && !present[r]) { gone[r] = 1; active--; }
			if (generation == my_gen) launch();
		}
	}
	return gen_status[my_gen & 1];
}

void GpuBatch::leave(int rx) {
	std::unique_lock<std::mutex> lock(mtx);
	if (rx < 0 || rx >= cfg.n_receivers || gone[rx]) return;
	gone[rx] = 1;
	active--;
	if (present[rx]) { present[rx] = 0; arrived--; }
	if (arrived > 0 && arrived == active) launch();
}

} // namespace aisamd
