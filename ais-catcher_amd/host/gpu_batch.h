// ais-catcher_amd/host/gpu_batch.h -- GpuBatch: one aisgpu_t (include/aisgpu.h) shared by R receiver threads.
//
// The reference runs one thread per receiver (Device/FileRAW.cpp:205-206), each calling its model's Receive() with its own
// block; the GPU wants the blocks of all receivers in ONE launch.  GpuBatch is the meeting point: every receiver thread copies its
// block into the context's pinned staging row (concurrently), the last one to arrive launches the chain, and every thread then
// replays ITS receiver's outputs into its own decoders.  Nothing here knows the block API: the same class serves the repository's
// mirror of it (gpu_model.h) and the binding compiled against the reference's real headers
// (integration/reference/Source/DSP/GPU/ModelGPU.cpp).  Depends on the C ABI and the standard library only.
#pragma once
#include <condition_variable>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/aisgpu.h"

namespace aisamd {

class GpuBatch {
	aisgpu_t* ctx = nullptr;
	aisgpu_cfg cfg;
	std::mutex mtx;
	std::condition_variable cv;
	int arrived = 0;             // receivers that have handed in their block of the current generation
	int active = 0;              // receivers still taking part (n_receivers minus those that left or were evicted)
	std::vector<char> present;   // [rx] handed in its block of the current generation
	std::vector<char> gone;      // [rx] left (end of its stream, failure) or evicted (stopped delivering): never waited for again
	long long generation = 0;
	int gen_status[2] = { AISGPU_OK, AISGPU_OK }; // status of generation g in slot g & 1 (a failed run does not poison later ones)
	int timeout_ms = 10000;      // a receiver that has not delivered this long after the first one of a generation is evicted
	bool pipelined = false;      // launch = collect the PREVIOUS block's outputs, then start this one: the host consumes block f-1 while the device runs f
	int fed = 0;                 // receivers that handed in data (not a drain request) in the current generation
	// AISGPU_FLAG_GPU_DECODE: the frames of the generation whose outputs are being served, copied out of the context at launch().  A
	// receiver reads them through its own reference: one that the timeout evicted while it was still delivering keeps a valid (old)
	// list instead of the context's array, which the next aisgpu_sync_outputs() clears and refills under it.
	std::shared_ptr<const std::vector<aisgpu_frame>> frame_snap;
	void launch();               // run the batch for the current generation and release the waiting threads (mtx held)

public:
	explicit GpuBatch(const aisgpu_cfg& c);
	~GpuBatch();
	GpuBatch(const GpuBatch&) = delete;
	const aisgpu_cfg& config() const { return cfg; }
	// Copies the receiver's block in; returns once the whole batch has been processed for this block.  AISGPU_ERR_STATE for a
	// receiver that was evicted (it did not deliver within the timeout while the others waited -- the reference marks such a
	// device lost, Device/Device.h:60-61) or has left.  Errors of one receiver's submit concern only that receiver.
	int submitAndWait(int rx, const void* iq, int n_iq);
	// The receiver will not deliver any more (end of its input, or it failed): the others stop waiting for it.
	void leave(int rx);
	void setTimeout(int ms) { std::lock_guard<std::mutex> l(mtx); timeout_ms = ms; } // <= 0: wait for ever
	// Pipelined hand-off: submitAndWait() of block f returns as soon as block f has been STARTED, with the outputs of block f-1 ready
	// (aisgpu_fetch serves them until the next launch); the receivers' replay / decoding of f-1 and the copying-in of f+1 then overlap
	// the device's work on f.  Messages come out one block later; after the last block every receiver calls submitAndWait(rx, nullptr, 0)
	// once to collect the last outputs.  Set before the first block.
	void setPipelined(bool b) { std::lock_guard<std::mutex> l(mtx); pipelined = b; }
	bool isPipelined() const { return pipelined; }
	int activeReceivers() { std::lock_guard<std::mutex> l(mtx); return active; }
	int outCount() { return aisgpu_out_count(ctx); }
	int fetch(int sub, int rx, int ch, aisgpu_out* out) { return aisgpu_fetch_sub(ctx, sub, rx, ch, out); }
	std::shared_ptr<const std::vector<aisgpu_frame>> frames() { std::lock_guard<std::mutex> l(mtx); return frame_snap; } // never null
	const char* lastError() { return aisgpu_last_error(ctx); }
};

} // namespace aisamd
