/*
 * Source/DSP/GPU/ModelGPU.cpp -- see ModelGPU.h.  Reference-side binding of libaisgpu.so; test infrastructure of the
 * MI355X repository (compiled by oracle/Makefile against the unmodified reference), not part of its product libraries.
 */
#include <algorithm>
#include <cstring>
#include <stdexcept>

#include "ModelGPU.h"

namespace AIS
{
	// ---- GpuPool

	GpuPool &GpuPool::instance()
	{
		static GpuPool pool;
		return pool;
	}

	static bool same_group(const aisgpu_cfg &a, const aisgpu_cfg &b)
	{
		return a.sample_rate == b.sample_rate && a.model == b.model && a.afc_wide == b.afc_wide && a.droop == b.droop && a.flags == b.flags &&
			   a.device_id == b.device_id;
	}

	GpuPool::Group *GpuPool::reserve(const aisgpu_cfg &c, int &rx)
	{
		std::lock_guard<std::mutex> l(mtx);
		aisgpu_cfg want = c;
		if (gpu_decode) want.flags |= AISGPU_FLAG_GPU_DECODE;
		for (Group *g : groups)
			if (!g->batch && same_group(g->cfg, want)) // (a group whose context exists is closed: later receivers form the next one)
			{
				rx = g->reserved++;
				g->users++;
				return g;
			}
		Group *g = new Group();
		g->cfg = want;
		rx = g->reserved++;
		g->users++;
		groups.push_back(g);
		return g;
	}

	aisamd::GpuBatch *GpuPool::open(Group *g, int block_len, int input_format)
	{
		std::lock_guard<std::mutex> l(mtx);
		if (!g->batch)
		{
			g->cfg.n_receivers = g->reserved;
			g->cfg.block_len = block_len;
			g->cfg.input_format = input_format;
			// Model.cpp:224-237: the fixed-point ladder only exists at 1536 kSPS and is fed from ConvertRAW::outCU8
			if ((g->cfg.flags & AISGPU_FLAG_FP_DS) && (g->cfg.sample_rate != 1536000 || input_format != AISGPU_FMT_CU8)) g->cfg.flags &= ~AISGPU_FLAG_FP_DS;
			g->batch = new aisamd::GpuBatch(g->cfg); // throws std::runtime_error (unsupported rate / block length, no GPU) like the reference's models at set-up
			g->batch->setTimeout(timeout_ms);
			g->batch->setPipelined(pipelined);
		}
		else if (g->cfg.block_len != block_len || g->cfg.input_format != input_format)
			throw std::runtime_error("GPU model: the receivers of one batch must deliver blocks of one size and format");
		return g->batch;
	}

	void GpuPool::release(Group *g, int rx)
	{
		// leave() may run a whole block (aisgpu_run + aisgpu_sync_outputs): not under the pool's lock, which every other receiver's
		// reserve() / open() / release() takes.  The batch pointer cannot go away meanwhile: this receiver still counts in users.
		aisamd::GpuBatch *b;
		{
			std::lock_guard<std::mutex> l(mtx);
			b = g->batch;
		}
		if (b) b->leave(rx);
		std::lock_guard<std::mutex> l(mtx);
		if (--g->users == 0)
		{
			delete g->batch;
			for (size_t i = 0; i < groups.size(); i++)
				if (groups[i] == g)
				{
					groups.erase(groups.begin() + i);
					break;
				}
			delete g;
		}
	}

	// ---- the tail of AIS::Decoder::Run for a frame the GPU decoders completed (see ModelGPU.h)

	void GpuEmitFrame(Decoder &d, const aisgpu_frame &f, TAG &tag)
	{
		d.msg.clear();
		for (int i = 0; i < f.position; i++)
			d.msg.setBit(i, (f.data[i >> 3] >> (i & 7)) & 1);
		if (tag.mode & 1) tag.level = f.level_sum / f.position; // Marine/AIS.h:152-153
		d.start_idx = f.start_idx;
		d.end_idx = f.end_idx;
		d.processData(f.position - 7, tag); // CRC (again), dB, validate, buildNMEA, Send (Marine/AIS.cpp:66-96)
	}

	// ---- GpuChain

	GpuChain::~GpuChain()
	{
		if (group) GpuPool::instance().release(group, rx);
	}

	void GpuChain::replay(Connection<FLOAT32> *out, const aisgpu_out &o, TAG &tag, int n0, int n1)
	{
		for (int g = 0; g < o.n_groups; g++)
		{
			const long long rel = 5 * (o.first_group + g) + 4 - o.first_sample48; // the sample that completes the group
			if (rel < n0 || rel >= n1) continue;
			const int w = (int)(rel / 512);
			if (o.ppm && w >= 0 && w < o.n_windows) tag.ppm = o.ppm[w];
			if (tag.mode & 1) tag.sample_lvl = o.lvl[g];
			for (int j = 0; j < N_SAMPLES_PER_SYMBOL; j++)
			{
				tag.sample_idx = 5 * (o.first_group + g) + j;
				FLOAT32 b = ((o.bits[j][g >> 5] >> (g & 31)) & 1u) ? 1.0f : -1.0f;
				out[j].Send((const FLOAT32 *)&b, 1, tag);
			}
		}
	}

	// the reference throttles the CGF output one sample at a time into both branches (Deinterleave n = 1, Model.cpp:630-639):
	// per 48 kHz sample N first the coherent branch (its five decoders fire when N completes a group), then FM decoder N % 5
	void GpuChain::replayChallenger(Connection<FLOAT32> *coh, Connection<FLOAT32> *fm, const aisgpu_out &o, TAG &tag, int n0, int n1)
	{
		const int L = o.n_windows * 512;
		for (int n = n0; n < L && n < n1; n++)
		{
			const long long N = o.first_sample48 + n;
			if (o.ppm) tag.ppm = o.ppm[n >> 9];
			if (N % 5 == 4)
			{
				const long long g = N / 5;
				const int gi = (int)(g - o.first_group);
				if (tag.mode & 1) tag.sample_lvl = o.lvl[gi];
				for (int j = 0; j < N_SAMPLES_PER_SYMBOL; j++)
				{
					tag.sample_idx = 5 * g + j;
					FLOAT32 b = ((o.bits[j][gi >> 5] >> (gi & 31)) & 1u) ? 1.0f : -1.0f;
					coh[j].Send((const FLOAT32 *)&b, 1, tag);
				}
			}
			tag.sample_idx = N;
			FLOAT32 f = ((o.fm_bits[n >> 5] >> (n & 31)) & 1u) ? 1.0f : -1.0f;
			fm[(int)(N % 5)].Send((const FLOAT32 *)&f, 1, tag);
		}
	}

	// ModelBase / ModelStandard: every 48 kHz sample of the block, in order (Demod::FM, Filter and SimplePLL leave the tag alone)
	void GpuChain::replayFM(Connection<FLOAT32> &fm, const aisgpu_out &o, TAG &tag, int n0, int n1)
	{
		const int L = o.n_windows * 512;
		for (int n = n0; n < L && n < n1; n++)
		{
			FLOAT32 f = ((o.fm_bits[n >> 5] >> (n & 31)) & 1u) ? 1.0f : -1.0f;
			fm.Send((const FLOAT32 *)&f, 1, tag);
		}
	}

	// One block of the context's size through the batch: copies it (it is only borrowed, Device/FileRAW.cpp:131-136), meets the other
	// receivers of the batch, returns when the whole batch has been through the GPU, then replays this receiver's outputs.
	bool GpuChain::runBlock(const void *iq, TAG &tag)
	{
		const int rc = batch->submitAndWait(rx, iq, block_bytes / sample_bytes);
		if (rc != AISGPU_OK)
		{
			failed = true; // the reference logs and stops (Device/FileRAW.cpp:111-115)
			Error() << "GPU model: " << aisgpu_strerror(rc) << ": " << batch->lastError();
			StopRequest();
			return false;
		}
		deliver(tag);
		return !failed;
	}

	// A device does not promise one block size per call.  RAWFile hands over `nblocks * fifo.BlockSize()` bytes with nblocks = 1 or 2
	// depending on how far its reader thread got (Device/FileRAW.cpp:120-136, Library/FIFO.h:99-109); the SDR devices send whatever
	// their FIFO holds.  The GPU context works on blocks of ONE size, so the chain re-blocks: whole blocks go straight from the
	// caller's buffer, a remainder waits in `carry` for the next call.  What comes out equals the reference's own chain driven with
	// calls of that block size (the one thing a call boundary changes in the arithmetic is where Rotate renormalises its phasor,
	// DSP/DSP.cpp:315 -- in the reference itself that depends on the reader thread's timing).
	void GpuChain::Receive(const RAW *raw, int len, TAG &tag)
	{
		if (failed || len != 1 || !group || raw->size <= 0) return;
		int fmt = 0, bytes = 0;
		switch (raw->format)
		{
		case Format::CU8: fmt = AISGPU_FMT_CU8; bytes = 2; break;
		case Format::CS8: fmt = AISGPU_FMT_CS8; bytes = 2; break;
		case Format::CS16: fmt = AISGPU_FMT_CS16; bytes = 4; break;
		case Format::CF32: fmt = AISGPU_FMT_CF32; bytes = 8; break;
		default:
			throw std::runtime_error("GPU model: input format not supported (CU8, CS8, CS16, CF32)");
		}
		if (!batch)
		{
			// The block of the group's context: what `GpuPool::setBlockBytes` asked for; else the file reader's FIFO block when the
			// device is one and this call is a whole number of them (Device/FileRAW.h:43: 24 * 16 * 16384 bytes, whether this first
			// call carries one or two); else the size of this first call (the SDR devices' callbacks, Device/RTLSDR.h:57).  The
			// first block of the group creates the context (throws like the reference's models do at set-up, Model.cpp:109-110).
			static const int RAWFILE_FIFO_BLOCK = 24 * 16 * 16384;
			block_bytes = GpuPool::instance().blockBytes();
			if (block_bytes <= 0) block_bytes = (device_type == Type::RAWFILE && raw->size % RAWFILE_FIFO_BLOCK == 0) ? RAWFILE_FIFO_BLOCK : raw->size;
			if (block_bytes % bytes) throw std::runtime_error("GPU model: block size is not a whole number of samples");
			sample_bytes = bytes;
			format = fmt;
			batch = GpuPool::instance().open(group, block_bytes / bytes, fmt);
			carry.reserve(block_bytes);
		}
		else if (fmt != format)
			throw std::runtime_error("GPU model: the input format changed between blocks");

		const char *p = (const char *)raw->data;
		size_t left = (size_t)raw->size;
		if (!carry.empty())
		{ // complete the block a previous call began
			const size_t take = std::min(left, (size_t)block_bytes - carry.size());
			carry.insert(carry.end(), p, p + take);
			p += take;
			left -= take;
			if (carry.size() < (size_t)block_bytes) return;
			const bool ok = runBlock(carry.data(), tag);
			carry.clear();
			if (!ok) return;
		}
		for (; left >= (size_t)block_bytes; p += block_bytes, left -= block_bytes)
			if (!runBlock(p, tag)) return;
		carry.insert(carry.end(), p, p + left);
	}

	void GpuChain::Flush(TAG &tag)
	{
		if (failed || !batch || !batch->isPipelined()) return;
		if (batch->submitAndWait(rx, nullptr, 0) == AISGPU_OK) deliver(tag);
	}

	void GpuChain::deliver(TAG &tag)
	{
		const aisgpu_cfg &cfg = batch->config();
		const int nsub = batch->outCount(); // downstream blocks this input block completed (1, or 1..2 behind the resampler; 0: the first call of a pipelined batch)
		if (nsub == 0) return;
		if (cfg.flags & AISGPU_FLAG_GPU_DECODE)
		{ // the decoders' state machines ran on the device: only completed frames come back, each to the tail of ITS decoder
			const auto frames = batch->frames(); // (this generation's list, valid for as long as this reference lives)
			for (size_t i = 0; i < frames->size(); i++)
			{
				const aisgpu_frame &f = (*frames)[i];
				if (f.rx != rx || f.sub >= nsub) continue;
				aisgpu_out o;
				if (batch->fetch(f.sub, rx, f.ch, &o) != AISGPU_OK) { failed = true; return; }
				if (o.n_groups > 0)
				{ // what the tag held when the frame closed (the FM receivers never touch tag.ppm / tag.sample_lvl)
					const long long n_last = f.phase >= 5 ? f.end_idx : 5 * (o.first_group + f.group) + 4;
					const int w = (int)((n_last - o.first_sample48) / 512);
					if (o.ppm && w >= 0 && w < o.n_windows) tag.ppm = o.ppm[w];
					if (tag.mode & 1) tag.sample_lvl = o.lvl[f.group];
				}
				tag.sample_idx = f.end_idx;
				if (cfg.model == AISGPU_MODEL_V2) memcpy(&tag.ppm, &f.group, sizeof(float)); // (the engine switches frequencies inside its blocks: the frame carries the bits of tag.ppm, aisgpu.h)
				Decoder *d = (f.phase >= 0 && f.phase < 2 * N_SAMPLES_PER_SYMBOL) ? dec[f.ch & 1][f.phase] : nullptr;
				if (d) GpuEmitFrame(*d, f, tag);
			}
			return;
		}
		// on the decimate-by-3 ladders Rotate works on DownsampleKFilter's 8192-sample blocks (DSP/DSP.h:193): A then B every
		// 4096 samples at 48 kHz; everywhere else channel A's whole block comes first
		static const int b2[8] = {96000, 192000, 384000, 768000, 1536000, 3072000, 6144000, 12288000};
		static const int b3[4] = {288000, 576000, 1152000, 2304000};
		int best = 0;
		bool by3 = false;
		for (int i = 0; i < 8 && !best; i++) if (b2[i] >= cfg.sample_rate) best = b2[i];
		for (int i = 0; i < ((cfg.flags & AISGPU_FLAG_DSK) ? 4 : 1); i++)
			if (b3[i] >= cfg.sample_rate && (!best || b3[i] < best)) { by3 = true; break; }
		if (cfg.flags & AISGPU_FLAG_MA_DS) by3 = true; // DownsampleMovingAverage hands on blocks of 8192 samples too (DSP/DSP.h:128)

		const int nch = (cfg.flags & AISGPU_FLAG_MODE_X) ? 1 : 2; // (channel mode X: FCIC5_b never sends anything in the reference either, Model.cpp:101-104)
		for (int s = 0; s < nsub; s++)
		{
			aisgpu_out o[2];
			for (int ch = 0; ch < nch; ch++)
				if (batch->fetch(s, rx, ch, &o[ch]) != AISGPU_OK) { failed = true; return; }
			const int L = o[0].n_windows * 512, step = by3 ? 4096 : L;
			for (int n0 = 0; n0 < L; n0 += step)
				for (int ch = 0; ch < nch; ch++)
				{
					const int n1 = n0 + step < L ? n0 + step : L;
					if (o[ch].c48) (ch == 0 ? outC48a : outC48b).Send((const CFLOAT32 *)o[ch].c48 + n0, n1 - n0, tag); // ModelEngineV2, engine on the host: the channel itself
					else if (o[ch].fm_bits && o[ch].n_groups == 0) replayFM(ch == 0 ? outFMa : outFMb, o[ch], tag, n0, n1);
					else if (o[ch].fm_bits) replayChallenger(ch == 0 ? outA : outB, ch == 0 ? outAf : outBf, o[ch], tag, n0, n1);
					else replay(ch == 0 ? outA : outB, o[ch], tag, n0, n1);
				}
		}
	}

	// ---- ModelDefaultGPU

	void ModelDefaultGPU::buildFrontend(int sample_rate, bool timerOn, Device::Device *dev, int model)
	{
		device = dev;
		chain.device_type = dev ? dev->getDriver() : Type::NONE;
		if (mode == Mode::X)
		{ // one channel, already centred (Model.cpp:35-107): the single-channel front end, ModelDefault's chain behind it
			if (model != AISGPU_MODEL_DEFAULT) throw std::runtime_error("GPU model: channel mode X with the ModelDefault engine only");
			if (sample_rate < 12000 || sample_rate > 192000)
				throw std::runtime_error("Model: sample rate must be between 12k and 192k (inclusive)."); // Model.cpp:38-39
		}
		else
		{
			if (mode != Mode::AB && mode != Mode::CD) throw std::runtime_error("GPU model: channel modes AB / CD / X only");
			if (sample_rate < 96000 || sample_rate > 12288000)
				throw std::runtime_error("Model: sample rate must be between 96K and 12288K (inclusive)."); // Model.cpp:109-110
		}

		Connection<RAW> &physical = timerOn ? (*device >> timer).out : device->out; // Model.cpp:33
		physical >> chain;

		aisgpu_cfg c;
		aisgpu_default_cfg(&c);
		c.sample_rate = sample_rate;
		c.model = model;
		c.afc_wide = CGF_wide;
		c.droop = droop_compensation;
		c.flags = (PS_EMA ? 0 : AISGPU_FLAG_PS_BOXCAR) | (fixedpointDS ? AISGPU_FLAG_FP_DS : 0) | (allowDSK ? AISGPU_FLAG_DSK : 0) | (MA_DS ? AISGPU_FLAG_MA_DS : 0);
		if (mode == Mode::X) c.flags = (c.flags & AISGPU_FLAG_PS_BOXCAR) | AISGPU_FLAG_MODE_X; // (the single-channel ladders know neither FP_DS nor DSK nor MA, Model.cpp:35-107)
		chain.join(c); // this receiver's row in the batch of all receivers built with the same configuration
	}

	void ModelDefaultGPU::buildModel(char CH1, char CH2, int sample_rate, bool timerOn, Device::Device *dev)
	{
		buildFrontend(sample_rate, timerOn, dev, AISGPU_MODEL_DEFAULT);

		for (int i = 0; i < N_SAMPLES_PER_SYMBOL; i++) // the decoder wiring of ModelDefault::buildModel (Model.cpp:545-574)
		{
			DEC_a[i].setOrigin(CH1, station, own_mmsi);
			DEC_b[i].setOrigin(CH2, station, own_mmsi);

			chain.outA[i] >> DEC_a[i] >> output;
			chain.outB[i] >> DEC_b[i] >> output;
			chain.dec[0][i] = &DEC_a[i];
			chain.dec[1][i] = &DEC_b[i];

			for (int j = 0; j < N_SAMPLES_PER_SYMBOL; j++)
				if (i != j)
				{
					DEC_a[i].DecoderMessage.Connect(DEC_a[j]);
					DEC_b[i].DecoderMessage.Connect(DEC_b[j]);
				}
		}
	}

	Setting &ModelDefaultGPU::SetKey(AIS::Keys key, const std::string &arg)
	{
		switch (key)
		{
		case AIS::KEY_SETTING_PS_EMA: // Model.cpp:579-594
			PS_EMA = Util::Parse::Switch(arg);
			break;
		case AIS::KEY_SETTING_AFC_WIDE:
			CGF_wide = Util::Parse::Switch(arg);
			break;
		case AIS::KEY_SETTING_FP_DS: // Model.cpp:358-402
			fixedpointDS = Util::Parse::Switch(arg);
			MA_DS = false;
			break;
		case AIS::KEY_SETTING_MA: // (sample rates that are a multiple of 96 kHz, blocks that are whole output blocks: aisgpu.h)
			MA_DS = Util::Parse::Switch(arg);
			break;
		case AIS::KEY_SETTING_DSK:
			allowDSK = Util::Parse::Switch(arg);
			break;
		case AIS::KEY_SETTING_DROOP:
			droop_compensation = Util::Parse::Switch(arg);
			break;
		case AIS::KEY_SETTING_SOXR:
		case AIS::KEY_SETTING_SRC:
			if (Util::Parse::Switch(arg)) throw std::runtime_error(getName() + ": the soxr / samplerate downsamplers are CPU-only (external libraries)");
			MA_DS = false;
			break;
		default:
			Model::SetKey(key, arg);
			break;
		}
		return *this;
	}

	std::string ModelDefaultGPU::Get()
	{
		return "ps_ema " + Util::Convert::toString(PS_EMA) + " afc_wide " + Util::Convert::toString(CGF_wide) + " droop " + Util::Convert::toString(droop_compensation) +
			   " fp_ds " + Util::Convert::toString(fixedpointDS) + " dsk " + Util::Convert::toString(allowDSK) + (MA_DS ? " MA ON " : " ") + Model::Get();
	}

	// ---- ModelChallengerGPU

	void ModelChallengerGPU::buildModel(char CH1, char CH2, int sample_rate, bool timerOn, Device::Device *dev)
	{
		buildFrontend(sample_rate, timerOn, dev, AISGPU_MODEL_CHALLENGER);

		for (int i = 0; i < N_SAMPLES_PER_SYMBOL; i++) // Model.cpp:641-674
		{
			DEC_a[i].setOrigin(CH1, station, own_mmsi);
			DEC_b[i].setOrigin(CH2, station, own_mmsi);
			DEC_af[i].setOrigin(CH1, station, own_mmsi);
			DEC_bf[i].setOrigin(CH2, station, own_mmsi);

			chain.outA[i] >> DEC_a[i] >> output;
			chain.outB[i] >> DEC_b[i] >> output;
			chain.outAf[i] >> DEC_af[i] >> output;
			chain.outBf[i] >> DEC_bf[i] >> output;
			chain.dec[0][i] = &DEC_a[i];
			chain.dec[1][i] = &DEC_b[i];
			chain.dec[0][N_SAMPLES_PER_SYMBOL + i] = &DEC_af[i]; // (frames of the device decoders: phase 5..9 = the FM decoders)
			chain.dec[1][N_SAMPLES_PER_SYMBOL + i] = &DEC_bf[i];

			for (int j = 0; j < N_SAMPLES_PER_SYMBOL; j++)
			{
				DEC_af[i].DecoderMessage.Connect(DEC_a[j]);
				DEC_a[i].DecoderMessage.Connect(DEC_af[j]);
				DEC_bf[i].DecoderMessage.Connect(DEC_b[j]);
				DEC_b[i].DecoderMessage.Connect(DEC_bf[j]);
				if (i != j)
				{
					DEC_a[i].DecoderMessage.Connect(DEC_a[j]);
					DEC_b[i].DecoderMessage.Connect(DEC_b[j]);
					DEC_af[i].DecoderMessage.Connect(DEC_af[j]);
					DEC_bf[i].DecoderMessage.Connect(DEC_bf[j]);
				}
			}
		}
	}

	// ---- ModelEngineV2GPU (Model.cpp:440-482)

	void ModelEngineV2GPU::buildModel(char CH1, char CH2, int sample_rate, bool timerOn, Device::Device *dev)
	{
		if (GpuPool::instance().gpuDecode() && (dd_train != 0.75f || dd_weight != 0.86f))
			throw std::runtime_error(getName() + ": the engine on the device has the default DD_TRAIN / DD_WEIGHT only");
		buildFrontend(sample_rate, timerOn, dev, AISGPU_MODEL_V2);

		V2_a.setOrigin(CH1, station, own_mmsi);
		V2_b.setOrigin(CH2, station, own_mmsi);
		V2_a.setWeights(dd_train, dd_weight);
		V2_b.setWeights(dd_train, dd_weight);

		chain.outC48a >> V2_a; // *C_a >> V2_a (Model.cpp:452-453)
		chain.outC48b >> V2_b;

		for (int i = 0; i < V2::Engine::N_DECODERS; i++)
		{
			V2_a.getDecoder(i) >> output;
			V2_b.getDecoder(i) >> output;
			chain.dec[0][i] = &V2_a.getDecoder(i);
			chain.dec[1][i] = &V2_b.getDecoder(i);
		}
	}

	Setting &ModelEngineV2GPU::SetKey(AIS::Keys key, const std::string &arg)
	{
		switch (key)
		{
		case AIS::KEY_SETTING_DD_TRAIN: // Model.cpp:465-482
			dd_train = Util::Parse::Float(arg, 0.0, 1.0);
			break;
		case AIS::KEY_SETTING_DD_WEIGHT:
			dd_weight = Util::Parse::Float(arg, 0.0, 1.0);
			break;
		case AIS::KEY_SETTING_PS_EMA: // (keys of ModelDefault, not of ModelFrontend: the reference's engine does not know them either)
		case AIS::KEY_SETTING_AFC_WIDE:
			Model::SetKey(key, arg);
			break;
		default:
			ModelDefaultGPU::SetKey(key, arg);
			break;
		}
		return *this;
	}

	std::string ModelEngineV2GPU::Get()
	{
		return "dd_train " + Util::Convert::toString(dd_train) + " dd_weight " + Util::Convert::toString(dd_weight) + " " + ModelDefaultGPU::Get();
	}

	// ---- ModelStandardGPU (Model.cpp:484-518)

	void ModelStandardGPU::buildModel(char CH1, char CH2, int sample_rate, bool timerOn, Device::Device *dev)
	{
		buildFrontend(sample_rate, timerOn, dev, AISGPU_MODEL_STANDARD);

		S_a.setConnections(N_SAMPLES_PER_SYMBOL);
		S_b.setConnections(N_SAMPLES_PER_SYMBOL);
		chain.outFMa >> S_a;
		chain.outFMb >> S_b;

		for (int i = 0; i < N_SAMPLES_PER_SYMBOL; i++)
		{
			DEC_a[i].setOrigin(CH1, station, own_mmsi);
			DEC_b[i].setOrigin(CH2, station, own_mmsi);

			S_a.out[i] >> DEC_a[i] >> output;
			S_b.out[i] >> DEC_b[i] >> output;
			chain.dec[0][i] = &DEC_a[i];
			chain.dec[1][i] = &DEC_b[i];

			for (int j = 0; j < N_SAMPLES_PER_SYMBOL; j++)
				if (i != j)
				{
					DEC_a[i].DecoderMessage.Connect(DEC_a[j]);
					DEC_b[i].DecoderMessage.Connect(DEC_b[j]);
				}
		}
	}

	// ---- ModelBaseGPU (Model.cpp:419-438): the reference's own SimplePLL with the decoder's training signals fed back into it

	void ModelBaseGPU::buildModel(char CH1, char CH2, int sample_rate, bool timerOn, Device::Device *dev)
	{
		buildFrontend(sample_rate, timerOn, dev, AISGPU_MODEL_BASE);

		DEC_a[0].setOrigin(CH1, station, own_mmsi);
		DEC_b[0].setOrigin(CH2, station, own_mmsi);

		chain.outFMa >> sampler_a >> DEC_a[0] >> output;
		chain.outFMb >> sampler_b >> DEC_b[0] >> output;
		chain.dec[0][0] = &DEC_a[0];
		chain.dec[1][0] = &DEC_b[0];

		DEC_a[0].DecoderMessage.Connect(sampler_a);
		DEC_b[0].DecoderMessage.Connect(sampler_b);
	}
}
