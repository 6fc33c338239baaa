// receiver.hip -- the receiver bank: Receiver::processIQData's DSP (application/receiver.cpp:826-987)
// for C tuned channels, composed from the device cores in the reference's step order.
#include <cmath>
#include "receiver.h"

namespace pg {

static long long lcm_ll(long long a, long long b)
{
    long long x = a, y = b;
    while (y) { long long t = x % y; x = y; y = t; }
    return a / x * b;
}

int Receiver::create(const pebblegpu_config *cfg)
{
    device = cfg->device;
    fs = cfg->sample_rate;
    nf = cfg->frames_per_buffer ? cfg->frames_per_buffer : 2048;  // settings.cpp:57
    C = cfg->n_channels;
    shared_input = cfg->shared_input != 0;
    S = shared_input ? 1 : C;
    wfm = cfg->wfm != 0;
    bins = cfg->spectrum_bins;
    ff_n = cfg->fastfir_fft ? cfg->fastfir_fft : 2048;      // fastfir.cpp:65
    ff_taps = cfg->fastfir_taps ? cfg->fastfir_taps : 1025;  // fastfir.cpp:66
    max_sf = cfg->max_superframes ? cfg->max_superframes : 1;
    if (C == 0 || fs <= 0 || fs > 4.0e9 || fs != std::floor(fs)) return fail(PEBBLEGPU_E_INVALID, "bad channel count or sample rate");
    if (nf < 256 || nf > 65535) return fail(PEBBLEGPU_E_INVALID, "frames_per_buffer must be 256..65535 (quint16, device_interfaces.h:32)");
    PG_HIP(hipSetDevice(device));
    PG_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    PG_HIP(hipStreamCreateWithFlags(&chain_stream_, hipStreamNonBlocking));
    { const char *e = getenv("PEBBLEGPU_PIPELINE"); pipeline_ = e && e[0] == '1'; }
    { const char *e = getenv("PEBBLEGPU_FUSE_DEC"); fuse_dec_ = e && e[0] == '1'; }
    for (auto &row : tm.ev)
        for (auto &e : row) PG_HIP(hipEventCreate(&e));

    chain = design::build_chain((uint32_t)fs, wfm ? 200000u : 30000u, 0);  // receiver.cpp:195,213
    if (chain.stages.empty() || chain.stages.size() > (size_t)kMaxStages)
        return fail(PEBBLEGPU_E_UNSUPPORTED, "sample rate %.0f yields no decimation chain; not built", fs);
    demod_rate_int = (uint32_t)(int)chain.rate;  // int members, receiver.h:165-166
    if (!wfm && (ff_taps < 2 || ff_taps > ff_n)) return fail(PEBBLEGPU_E_INVALID, "FastFIR taps must be in [2, fft size]");
    const long long L = wfm ? (long long)nf : (long long)ff_n - ((long long)ff_taps - 1);
    superframe = (uint64_t)chain.total * (uint64_t)lcm_ll(nf, L);
    const long long max_n = (long long)max_sf * (long long)superframe;
    const long long nd_max = max_n / chain.total;

    ctl_.assign(C, ChanCtl());
    for (auto &c : ctl_) c.mode = wfm ? PEBBLEGPU_DM_FMM : PEBBLEGPU_DM_AM;  // Demod ctor default dmAM, demod.cpp:56
    if (int rc = osc_.init(C, fs)) return rc;
    osc_.allow_inline = true;
    osc_.device_advance = true;  // (banks of more than kOscInline channels: no per-call copy of the oscillators' phases)
    // "Restore gain lost in decimation" 10^(2*stages/20) only on the narrow branch (receiver.cpp:935-938 vs :854-901)
    const float gain = wfm ? 1.f : (float)std::pow(10.0, (double)(chain.dec_by2 * 2) / 20.0);
    if (int rc = dec_.init(C, chain, max_n, wfm ? 0 : (int)ff_taps - 1, gain)) return rc;
    if (int rc = audio.alloc((int)C, 0, nd_max)) return rc;
    if (!wfm) {
        // Two-stage calls (the band-pass and everything behind it on the chain's stream, beside the NEXT call's decimator): needs the
        // decimator's output twice (PEBBLEGPU_BANK_PIPELINE=0 when the receiver is created keeps every call on one stream)
        { const char *e = getenv("PEBBLEGPU_BANK_PIPELINE"); bank_pipe_ok_ = !(e && e[0] == '0') && chain.stages.size() > 1 && !bins; }
        if (bank_pipe_ok_) {
            if (int rc = dec_.enable_double_out()) return rc;
            // The second stage runs in what the decimator leaves idle: its stream has the lower priority, so that when a call's decimator
            // and the previous call's band-pass become ready together (both wait for the same launch) the decimator's workgroups are placed
            // first -- the other way round the band-pass filled the CUs and the decimator, one 230-register wave per SIMD, took 112 us
            // instead of 65 waiting for room
            int lo = 0, hi = 0;
            PG_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
            (void)hipStreamDestroy(chain_stream_);
            chain_stream_ = nullptr;
            PG_HIP(hipStreamCreateWithPriority(&chain_stream_, hipStreamNonBlocking, lo));
        }
        if (int rc = ff_.init(C, ff_n, ff_taps)) return rc;
        if (int rc = am_.init(C, (double)demod_rate_int, nd_max)) return rc;  // Demod_AM(m_inputSampleRate), demod.cpp:62
        // Demod_SAM / Demod_NFM objects also exist in every Receiver (demod.cpp:63-64); their buffers are allocated on first use
        pll_cap_ = nd_max;
        if (int rc = agc_.init(C, (double)demod_rate_int)) return rc;  // AGC(m_demodSampleRate, m_demodFrames), receiver.cpp:264
        if (int rc = anf_.init(C)) return rc;
    } else {
        if (int rc = wfmc_.init(C, (double)demod_rate_int, nd_max)) return rc;  // Demod_WFM(m_inputWfmSampleRate), demod.cpp:65
        wfmc_.stereo_block = (int)nf;  // the reference demodulates one accumulated frame per call (receiver.cpp:896)
    }
    if (int rc = cond_.init(S, nf, fs, max_n)) return rc;
    audio_rate = cfg->audio_rate;
    if (audio_rate) {
        // resampRate = (m_demodSampleRate*1.0) / (m_audioOutRate*1.0), the int members (receiver.cpp:901,994)
        const double rr = ((double)demod_rate_int * 1.0) / ((double)audio_rate * 1.0);
        if (rr == 1.0) audio_rate = 0;  // copyCPX branch (receiver.cpp:1002-1003)
        else {
            if (int rc = resamp_.init(C, nf, rr, (uint32_t)(nd_max / nf))) return rc;
            rs_pitch = resamp_.max_out(nd_max);
            PG_HIP(hipMalloc((void **)&d_audio_rs, sizeof(float2) * (size_t)rs_pitch * C));
        }
    }
    if (bins) {
        if (int rc = spec_.init(S, nf, bins)) return rc;
        bins = spec_.bins;
        PG_HIP(hipMalloc((void **)&d_spec, sizeof(float) * (size_t)(max_n / nf) * bins * S));
        if (fuse_dec_ && spec_.dec_ready() && nf == 2048) { if (int rc = dec_.set_fuse_window(spec_.d_window, spec_.h_window)) return rc; }  // (opt-in) the decimator may run inside the transform's kernel
    }
    zoom_bins = cfg->hires_bins;
    if (zoom_bins) {  // m_fftHiRes->fftParams(m_numHiResSpectrumBins, maxDb, m_hiResSampleRate, numSamples, BLACKMANHARRIS), signalspectrum.cpp:59
        if (nd_max % nf != 0) return fail(PEBBLEGPU_E_UNSUPPORTED, "the zoomed spectrum needs whole frames at the demodulator rate");
        if (int rc = zoom_.init(C, nf, zoom_bins)) return rc;
        zoom_bins = zoom_.bins;
        PG_HIP(hipMalloc((void **)&d_zoom, sizeof(float) * (size_t)(nd_max / nf) * zoom_bins * C));
    }
    return 0;
}

Receiver::~Receiver()
{
    (void)hipSetDevice(device);
    if (chain_stream_) (void)hipStreamSynchronize(chain_stream_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    osc_.release(); dec_.release(); ff_.release(); am_.release(); nfm_.release(); sam_.release(); wfmc_.release(); spec_.release(); zoom_.release();
    if (copy_stream_) { (void)hipStreamSynchronize(copy_stream_); (void)hipStreamDestroy(copy_stream_); }
    for (hipEvent_t e : sync_ev_) if (e) (void)hipEventDestroy(e);
    for (IngestSlot &g : ingest_) {
        if (g.h) (void)hipHostFree(g.h);
        if (g.d) (void)hipFree(g.d);
        for (hipEvent_t e : {g.uploaded, g.done_main, g.done_chain}) if (e) (void)hipEventDestroy(e);
    }
    if (d_zoom) (void)hipFree(d_zoom);
    agc_.release(); resamp_.release(); cond_.release(); anf_.release();
    if (d_audio_rs) (void)hipFree(d_audio_rs);
    if (h_gate_) (void)hipHostFree(h_gate_);
    if (d_squelch) (void)hipFree(d_squelch);
    if (d_gate) (void)hipFree(d_gate);
    if (d_raw_stage_) (void)hipFree(d_raw_stage_);
    if (d_smeter) (void)hipFree(d_smeter);
    if (d_sm_bins) (void)hipFree(d_sm_bins);
    audio.release();
    if (d_spec) (void)hipFree(d_spec);
    if (d_stage_in_) (void)hipFree(d_stage_in_);
    for (auto &row : tm.ev)
        for (auto &e : row) if (e) (void)hipEventDestroy(e);
    if (chain_stream_) (void)hipStreamDestroy(chain_stream_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

int Receiver::set_mixer(uint32_t ch, double f)
{
    if (ch >= C) return fail(PEBBLEGPU_E_INVALID, "channel %u out of range", ch);
    std::lock_guard<std::mutex> g(mu_);
    touched_ = true;  // the next call joins its two pipelines before the change is applied
    osc_.retune(ch, f);
    sm_dirty_ = true;
    return 0;
}

int Receiver::set_mode(uint32_t ch, int mode)
{
    if (ch >= C) return fail(PEBBLEGPU_E_INVALID, "channel %u out of range", ch);
    if (wfm) {
        if (mode != PEBBLEGPU_DM_FMM && mode != PEBBLEGPU_DM_FMS) return fail(PEBBLEGPU_E_UNSUPPORTED, "a WFM bank demodulates FMM and FMS only");
        std::lock_guard<std::mutex> g(mu_);
        touched_ = true;  // the next call joins its two pipelines before the change is applied
        if (int rc = wfmc_.set_stereo(ch, mode == PEBBLEGPU_DM_FMS)) return rc;
        ctl_[ch].mode = mode;
        return 0;
    } else if (mode == PEBBLEGPU_DM_FMM || mode == PEBBLEGPU_DM_FMS || mode < 0 || mode > PEBBLEGPU_DM_NONE) {
        return fail(PEBBLEGPU_E_UNSUPPORTED, "demod mode %d is not available in a narrow bank (FMM and FMS need a wfm bank)", mode);
    }
    std::lock_guard<std::mutex> g(mu_);
    touched_ = true;  // the next call joins its two pipelines before the change is applied
    if (ctl_[ch].mode != mode) am_list_dirty_ = true;
    ctl_[ch].mode = mode;
    if (!wfm) {  // "Tune only mode": the reference returns before NoiseFilter and AGC (receiver.cpp:968-971): their states stay frozen
        agc_.set_muted(ch, mode == PEBBLEGPU_DM_NONE);
        anf_.set_muted(ch, mode == PEBBLEGPU_DM_NONE);
    }
    return 0;
}

int Receiver::enable_smeter(bool on)
{
    std::lock_guard<std::mutex> g(mu_);
    if (on && !bins) return fail(PEBBLEGPU_E_INVALID, "signal strength is measured on the spectrum: create the bank with spectrum_bins");
    PG_HIP(hipSetDevice(device));
    if (on && !d_smeter) {
        smeter_pitch = (long long)max_sf * (long long)(superframe / nf);
        PG_HIP(hipMalloc((void **)&d_smeter, sizeof(float4) * (size_t)smeter_pitch * C));
        PG_HIP(hipMalloc((void **)&d_sm_bins, sizeof(SmBins) * C));
        sm_dirty_ = true;
    }
    smeter_on = on;
    return 0;
}

int Receiver::rds_groups(uint32_t ch, RdsGroup *g, unsigned char *changed, uint32_t cap, uint32_t *n)
{
    if (n) *n = 0;
    if (ch >= C) return fail(PEBBLEGPU_E_INVALID, "channel %u out of range", ch);
    if (!wfm) return fail(PEBBLEGPU_E_UNSUPPORTED, "RDS groups come from the dmFMS channels of a WFM bank");
    if (int rc = sync()) return rc;
    return wfmc_.rds.groups(nullptr, ch, g, changed, cap, n);
}
int Receiver::stereo_lock(uint32_t ch, int *lock, int *changed)
{
    if (ch >= C) return fail(PEBBLEGPU_E_INVALID, "channel %u out of range", ch);
    if (!wfm) return fail(PEBBLEGPU_E_UNSUPPORTED, "the stereo lock belongs to the dmFMS channels of a WFM bank");
    if (int rc = sync()) return rc;
    return wfmc_.stereo_lock(nullptr, ch, lock, changed);
}
int Receiver::set_squelch(uint32_t ch, double squelch_db)
{
    if (ch >= C) return fail(PEBBLEGPU_E_INVALID, "channel %u of %u", ch, C);
    if (C != 1 || max_sf != 1) {
        // a bank, or calls of several super-frames: per-channel thresholds, the decision per (channel, super-frame) on the device
        if (wfm) {
            if (squelch_db <= -120.0) return 0;  // "never closes": nothing to set up
            return fail(PEBBLEGPU_E_UNSUPPORTED, "the per-channel gate of a bank is built for the narrow branch (a WFM receiver gates as one channel, one super-frame per call)");
        }
        if (squelch_db > -120.0) {
            if (int rc = enable_smeter(true)) return rc;
        }
        std::lock_guard<std::mutex> g(mu_);
        touched_ = true;  // the next call joins its two pipelines before the change is applied
        PG_HIP(hipSetDevice(device));
        if (squelch_.empty()) squelch_.assign(C, -120.f);
        if (!d_squelch) {
            PG_HIP(hipMalloc((void **)&d_squelch, sizeof(float) * C));
            PG_HIP(hipMalloc((void **)&d_gate, (size_t)C * max_sf));
        }
        squelch_[ch] = (float)squelch_db;
        bank_gate_ = false;
        for (float v : squelch_) bank_gate_ = bank_gate_ || v > -120.f;
        squelch_dirty_ = true;
        return 0;
    }
    if (squelch_db > -120.0) {
        if (int rc = enable_smeter(true)) return rc;
    }
    std::lock_guard<std::mutex> g(mu_);
    touched_ = true;  // the next call joins its two pipelines before the change is applied
    if (squelch_db > -120.0 && !h_gate_) {
        PG_HIP(hipSetDevice(device));
        PG_HIP(hipHostMalloc((void **)&h_gate_, sizeof(float4)));
    }
    squelch_db_ = squelch_db;
    return 0;
}

int Receiver::set_conditioners(uint32_t stream, int flags, double iq_gain, double iq_phase)
{
    std::lock_guard<std::mutex> g(mu_);
    touched_ = true;  // the next call joins its two pipelines before the change is applied
    PG_HIP(hipSetDevice(device));
    return cond_.set(stream, flags, iq_gain, iq_phase);
}

int Receiver::set_noise_filter(uint32_t ch, bool on)
{
    if (wfm) return fail(PEBBLEGPU_E_UNSUPPORTED, "the WFM branch has no noise filter step (receiver.cpp:854-901)");
    std::lock_guard<std::mutex> g(mu_);
    touched_ = true;  // the next call joins its two pipelines before the change is applied
    PG_HIP(hipSetDevice(device));
    return anf_.set(ch, on);
}

int Receiver::set_agc(uint32_t ch, int mode, int threshold)
{
    if (ch >= C) return fail(PEBBLEGPU_E_INVALID, "channel %u out of range", ch);
    if (wfm) return fail(PEBBLEGPU_E_UNSUPPORTED, "the WFM branch has no AGC (receiver.cpp:854-901)");
    std::lock_guard<std::mutex> g(mu_);
    touched_ = true;  // the next call joins its two pipelines before the change is applied
    return agc_.set_mode(ch, mode, threshold);
}

int Receiver::set_bandpass(uint32_t ch, double lo, double hi)
{
    if (ch >= C) return fail(PEBBLEGPU_E_INVALID, "channel %u out of range", ch);
    if (wfm) return fail(PEBBLEGPU_E_UNSUPPORTED, "the WFM branch has no band-pass (receiver.cpp:854-901)");
    std::lock_guard<std::mutex> g(mu_);
    touched_ = true;  // the next call joins its two pipelines before the change is applied
    ChanCtl &c = ctl_[ch];
    const double flo = (double)(float)lo, fhi = (double)(float)hi;  // setBandPass(float, float), bandpassfilter.cpp:38
    if (c.mode == PEBBLEGPU_DM_AM) {  // Demod::setBandwidth only acts in AM (demod.cpp:230-239)
        c.am_bw = hi - lo;
        c.am_dirty = true;
    }
    if (c.bp_valid && flo == c.lo && fhi == c.hi) return 0;  // "return if no changes", fastfir.cpp:195-199
    c.lo = flo;  // stored before the sanity check, fastfir.cpp:200-203
    c.hi = fhi;
    sm_dirty_ = true;
    c.bp_valid = true;
    const double rate = (double)demod_rate_int;
    if (flo >= fhi || flo >= rate / 2.0 || flo <= -rate / 2.0 || fhi >= rate / 2.0 || fhi <= -rate / 2.0)
        return fail(PEBBLEGPU_E_FILTER_PARAM, "Filter Parameter error: lo %.1f hi %.1f rate %.1f", flo, fhi, rate);
    c.bp_dirty = true;
    return 0;
}

int Receiver::apply_controls(hipStream_t osc_stream)
{
    if (int rc = osc_.upload(osc_stream)) return rc;
    if (smeter_on && sm_dirty_) {
        // bin indices of fdEstimate (signalstrength.cpp:313-337): integer bin width, truncating conversions, qBound
        std::vector<SmBins> hb(C);
        const int nb = (int)bins;
        const double bin_width = (double)((uint32_t)fs / (uint32_t)nb);
        auto qb = [nb](int v) { return v < 0 ? 0 : (v > nb ? nb : v); };
        for (uint32_t ch = 0; ch < C; ch++) {
            const float lo = wfm ? -100000.f : (float)ctl_[ch].lo, hi = wfm ? 100000.f : (float)ctl_[ch].hi;  // receiver.cpp:891-892,959-960
            const int mixer_bin = qb((int)(nb / 2 + (osc_.ctl[ch].freq / bin_width)));
            SmBins &b = hb[ch];
            b.lo = qb((int)(mixer_bin + (lo / bin_width)));
            b.hi = qb((int)(mixer_bin + (hi / bin_width)));
            b.bp_bins = b.hi - b.lo;
            b.nlo = qb(b.lo - b.bp_bins);
            b.nhi = qb(b.hi + b.bp_bins);
            b.stream = shared_input ? 0 : (int)ch;
        }
        PG_HIP(hipMemcpyAsync(d_sm_bins, hb.data(), sizeof(SmBins) * C, hipMemcpyHostToDevice, stream_));
        PG_HIP(hipStreamSynchronize(stream_));
        sm_dirty_ = false;
    }
    if (squelch_dirty_) {
        PG_HIP(hipMemcpyAsync(d_squelch, squelch_.data(), sizeof(float) * C, hipMemcpyHostToDevice, stream_));
        PG_HIP(hipStreamSynchronize(stream_));
        squelch_dirty_ = false;
    }
    if (wfm) return 0;
    for (uint32_t ch = 0; ch < C; ch++) {
        ChanCtl &c = ctl_[ch];
        if (c.bp_dirty) {
            bool ok = false;
            if (int rc = ff_.design(stream_, ch, c.lo, c.hi, 0.0, (double)demod_rate_int, &ok)) return rc;
            c.bp_dirty = false;
        }
        if (c.am_dirty && c.mode == PEBBLEGPU_DM_AM) {
            if (int rc = am_.set_bandwidth(stream_, ch, c.am_bw)) return rc;
            c.am_dirty = false;
        }
    }
    if (int rc = agc_.apply(stream_)) return rc;
    if (int rc = anf_.apply(stream_)) return rc;
    if (am_list_dirty_) {
        std::vector<int> l, ls, ln;
        for (uint32_t ch = 0; ch < C; ch++) {
            if (ctl_[ch].mode == PEBBLEGPU_DM_AM) l.push_back((int)ch);
            else if (ctl_[ch].mode == PEBBLEGPU_DM_SAM) ls.push_back((int)ch);
            else if (ctl_[ch].mode == PEBBLEGPU_DM_FMN) ln.push_back((int)ch);
        }
        if (int rc = am_.set_list(stream_, l)) return rc;
        if (!ls.empty() && sam_.C == 0) { if (int rc = sam_.init(C, (double)demod_rate_int, pll_cap_, 1)) return rc; }
        if (!ln.empty() && nfm_.C == 0) { if (int rc = nfm_.init(C, (double)demod_rate_int, pll_cap_, 0)) return rc; }
        if (sam_.C) { if (int rc = sam_.set_list(stream_, ls)) return rc; }
        if (nfm_.C) { if (int rc = nfm_.set_list(stream_, ln)) return rc; }
        am_list_dirty_ = false;
    }
    return 0;
}

int Receiver::process(const float2 *d_iq, uint64_t n, bool with_spectrum, bool with_chain, const RawSrc *raw)
{
    std::lock_guard<std::mutex> g(mu_);
    PG_HIP(hipSetDevice(device));
    if (failed_) return fail(PEBBLEGPU_E_HIP, "an earlier call on this receiver failed half-way (its filter histories no longer match its oscillators): destroy it");
    if ((!d_iq && !raw) || n == 0) return fail(PEBBLEGPU_E_INVALID, "null input or zero samples");
    if (with_chain && (n % superframe != 0 || n / superframe > max_sf))
        return fail(PEBBLEGPU_E_SIZE, "n_samples %llu is not 1..%u super-frames of %llu", (unsigned long long)n, max_sf,
                    (unsigned long long)superframe);
    if (with_spectrum && (!bins || n % nf != 0 || n > (uint64_t)max_sf * superframe))
        return fail(PEBBLEGPU_E_SIZE, "spectrum needs whole frames of %u samples within capacity", nf);
    // caller mistakes around the squelch gate are refused HERE, before anything is queued: they leave the handle usable (a failure
    // behind this point has kernels in flight and histories half advanced, and closes the handle)
    if (with_chain && squelch_db_ > -120.0 && !with_spectrum && !last_spec_frames)
        return fail(PEBBLEGPU_E_INVALID, "the squelch gate needs a spectrum: none has been computed yet");
    if (with_chain && !wfm && bank_gate_ && !with_spectrum && !(squelch_db_ > -120.0) && !(C == 1 && ctl_[0].mode == PEBBLEGPU_DM_NONE))
        return fail(PEBBLEGPU_E_INVALID, "the squelch gate of a bank reads the spectra of the same call: create the bank with spectrum_bins");
    // side by side: the chain goes to its own stream while the display transform keeps the arithmetic units busy (only when
    // the chain's first kernel needs no LDS -- the transform's workgroups leave none -- and nothing downstream reads the
    // spectrum or a conditioned copy of the input)
    const bool side = with_spectrum && with_chain && !profile_detail && squelch_db_ <= -120.0 && !bank_gate_ && dec_.front_is_lds_free() && !cond_.any && !cond_.dirty;
    // Pipelined calls: the display transforms of successive calls follow one another on the main stream and the chains on the
    // chain's stream -- neither waits for the other's previous call (they share nothing: the transform carries its previous
    // amplitudes, the chain its histories and oscillators), so a call's short, LDS-hungry tail kernels run beside the NEXT
    // call's transform instead of on an idle GPU.  Results are complete after sync() (the contract of include/pebblegpu.h).
    // Anything else -- a control change to apply, a call of another shape -- first orders the two queues behind each other.
    const bool was_touched = touched_;
    // Two-stage calls of a receiver without a display transform: mixer + decimator (and the refresh of their histories) on the main
    // stream, band-pass, noise filter, AGC, demodulators and resampler on the chain's stream behind an event -- the decimator of the next
    // call does not wait for them (it writes the other output buffer; it does wait for the band-pass of the call before the last, which
    // read that buffer).  The decimator of a bank leaves the vector units idle two thirds of the time (one wave per SIMD, bound by
    // its own instruction stream): the band-pass of the previous call fits beside it.  Results are complete after sync().
    const bool bank_pipe = bank_pipe_ok_ && with_chain && !with_spectrum && !profile_detail && squelch_db_ <= -120.0 && !bank_gate_ && !zoom_bins &&
                           !cond_.any && !cond_.dirty && dec_.double_out();
    const bool plain = (side && pipeline_ && !touched_) || (bank_pipe && !touched_);
    auto join = [&]() -> int {
        if (chain_end_) PG_HIP(hipStreamWaitEvent(stream_, chain_end_, 0));
        if (spec_end_) PG_HIP(hipStreamWaitEvent(chain_stream_, spec_end_, 0));
        chain_end_ = spec_end_ = nullptr;
        return 0;
    };
    if (!plain) { if (int rc = join()) return rc; }
    if (int rc = apply_controls(plain ? chain_stream_ : stream_)) return rc;
    touched_ = false;
    if (int rc = cond_.apply(stream_)) return rc;
    bool staged = false;  // a conversion pass was queued in front of the call
    if (raw) {
        // Raw device-format input: when the call's first kernels convert in their own loads (the 8192-bin display transform
        // and the one-channel first stage beside it) there is no float2 copy of the stream at all; otherwise normalizeIQ runs
        // as its own pass into a staging buffer and the call goes on from there.
        dec_.want_lds_free = side;
        const bool fused = side && S == 1 && spec_.raw_ready() && dec_.raw_ready(osc_);
        staged = !fused;
        if (!fused) {
            if (plain && !bank_pipe) { if (int rc = join()) return rc; }  // the staging buffer is shared by successive calls (two-stage calls: only their first stage touches it)
            if (!d_raw_stage_) PG_HIP(hipMalloc((void **)&d_raw_stage_, sizeof(float2) * (size_t)S * max_sf * superframe));
            // streams are stream-major in both layouts, so one pass over S * n pairs converts them all
            if (int rc = run_normalize_iq(raw->fmt, raw->order, 1.0, raw->base, (long long)(S * n), d_raw_stage_, stream_, false, &raw->scale)) return rc;
            d_iq = d_raw_stage_;
            raw = nullptr;
        }
    }
    long long in_pitch = (long long)n;
    // DCRemoval, IQBalance, NoiseBlanker 1/2 on the raw streams, ahead of the spectrum and the mixer (receiver.cpp:814-823)
    if (int rc = cond_.run(stream_, d_iq, in_pitch, (long long)n, &d_iq, &in_pitch)) return rc;
    hipEvent_t *ev = tm.slot();
    const int slot = (int)(tm.calls % Timers::kRing);
    tm.calls++;
    // Every event record is a packet of its own in the queue (~6 us of idle GPU between two kernels).  A side-by-side call therefore
    // records no end event: it ends where the next call's start event is recorded (same queue, nothing in between), or where
    // sync() / a timing query closes it (close_timing).
    hipEvent_t start = ev[0];
    if (bank_pipe && plain && d_end_prev_) start = d_end_prev_;  // (back to back, a two-stage call is timed from where the previous one ended: one queue packet less)
    else PG_HIP(hipEventRecord(ev[0], stream_));
    tm.start_ev[slot] = start;
    if (tm.open_slot >= 0) {
        tm.end_ev[tm.open_slot] = start;
        tm.open_slot = -1;
    }
    tm.end_ev[slot] = ev[6];
    hipStream_t cs = side ? chain_stream_ : stream_;
    if (side) {
        PG_HIP(hipStreamWaitEvent(chain_stream_, start, 0));  // fork: the input is ready where the call starts
    }
    // the output buffer this call writes was read three (two) calls ago (a wait is a queue packet: none when the host can see that it is over)
    // three output buffers in rotation for short calls, two for long ones (chunks of 128 outputs or more: 0.917 / 0.923 ms per configs[2]
    // call of 128 super-frames with two against 0.950-0.989 with three, the same at 32, 0.0775 against 0.0658 at 8)
    const bool rot3 = bank_pipe && dec_.fin3.base && !dec_.long_call((long long)n);
    dec_.rotate3 = rot3;
    {
        hipEvent_t last_reader = nullptr;  // where the second stage that last read the buffer this call writes (dec_.fin2) ended
        for (const auto &pr : out_reader_) if (pr.first == (const void *)dec_.fin2.base) last_reader = pr.second;
        if (bank_pipe && last_reader && hipEventQuery(last_reader) != hipSuccess) {
            // the host is more than two calls ahead of the device: it waits here (PEBBLEGPU_BANK_PIPE_HOSTWAIT=0: a wait in the queue instead,
            // one more packet between this decimator and the last)
            static const bool host_wait = [] { const char *e = getenv("PEBBLEGPU_BANK_PIPE_HOSTWAIT"); return !(e && e[0] == '0'); }();
            if (host_wait) PG_HIP(hipEventSynchronize(last_reader));
            else PG_HIP(hipStreamWaitEvent(stream_, last_reader, 0));
        }
    }
    // From here on a failing step leaves kernels queued (on the chain stream too) and histories half advanced: whatever the
    // exit, join the two streams so later work is ordered behind what was queued, and refuse further calls on the handle.
    struct Guard {
        Receiver *r; hipEvent_t *ev; hipStream_t cs; bool side, armed;
        ~Guard()
        {
            if (!armed) return;
            r->failed_ = true;
            if (side && hipEventRecord(ev[6], cs) == hipSuccess) r->chain_end_ = ev[6];
        }
    } guard{this, ev, bank_pipe ? chain_stream_ : cs, side || bank_pipe, true};
    // One channel through hb11 x 8, hb15, hb23, hb47 beside the 8192-bin transform: the transform's workgroups can compute the decimator
    // from the frames they hold (k_spectrum_t128<.., DEC>): the stream crosses HBM once, nothing is written at the intermediate rates.
    // Opt-in (PEBBLEGPU_FUSE_DEC=1 when the receiver is created): measured slower -- the stages sit in the kernel's barrier intervals,
    // 0.297 ms against 0.247 beside the stand-alone first stage, the call 0.330 against 0.293
    dec_.want_lds_free = side;
    const bool fuse_dec = fuse_dec_ && side && with_chain && !pipeline_ && S == 1 && spec_.dec_ready() && nf == 2048 && dec_.spectrum_can_run(osc_) && (!raw || !staged);
    DecFuse df;
    if (fuse_dec) { if (int rc = dec_.fill_dec_fuse(stream_, &df, osc_, (long long)n)) return rc; }
    if (with_spectrum) {  // SignalSpectrum::unprocessed on the raw frame, receiver.cpp:826
        // (a call whose chain follows on the same stream, or that has none, leaves the GPU to the transform: its all-registers variant)
        if (int rc = spec_.run(stream_, d_iq, in_pitch, (long long)(n / nf), d_spec, raw, fuse_dec ? &df : nullptr, !side)) return rc;
        last_spec_frames = n / nf;
        if (smeter_on) {
            const long long F = (long long)(n / nf);
            if (int rc = run_signal_strength(stream_, d_spec, F * (long long)bins, (int)bins, F, d_sm_bins, d_smeter, smeter_pitch, C)) return rc;
        }
    }
    // (every record is a ~5 us bubble in the stream: a call with no display transform does without the one behind it)
    const bool mid = with_spectrum || profile_detail || side;
    if (mid) PG_HIP(hipEventRecord(ev[1], stream_));
    tm.detailed[(tm.calls - 1) % Timers::kRing] = profile_detail;
    tm.has_mid[(tm.calls - 1) % Timers::kRing] = mid;
    if (!with_chain) {
        if (profile_detail) for (int i = 2; i <= 5; i++) PG_HIP(hipEventRecord(ev[i], stream_));
        PG_HIP(hipEventRecord(ev[6], stream_));
        guard.armed = false;
        return 0;
    }
    // Mixer::processBlock + Decimator::process, receiver.cpp:867-868 / :910-911
    dec_.want_lds_free = side;
    // an event record costs the stream a ~5 us bubble: per-kernel events only when asked for (set_profiling)
    OscAdvance oa_pre;
    memset(&oa_pre, 0, sizeof(oa_pre));
    bool have_oa = false;
    if (fuse_dec) {
        if (int rc = dec_.run_beside_spectrum(cs, d_iq, in_pitch, shared_input, (long long)n, osc_, raw)) return rc;
        PG_HIP(hipStreamWaitEvent(cs, ev[1], 0));  // everything behind the decimator reads what the transform's kernel wrote
    } else {
        // (the oscillators' advance is offered to the decimator: the bank kernel carries it in its own launch, DecimCore::osc_advanced)
        if (!wfm) {
            if (int rc = osc_.advance_job(cs, n, &oa_pre)) return rc;
            have_oa = true;
        }
        static const bool ext_ev = [] { const char *e = getenv("PEBBLEGPU_BANK_PIPE_EXTEV"); return e && e[0] == '1'; }();  // opt-in: measured 0.0695 / 0.0753 ms (configs[2] / configs[3] shard) against 0.0663 / 0.0774 without
        dec_.done_event = nullptr;
        if (bank_pipe && ext_ev) {  // the hand-over event of a two-stage call: completed by the bank kernel's own dispatch when that ends the first stage
            if (!sync_ev_[0]) for (hipEvent_t &e : sync_ev_) PG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            dec_.done_event = sync_ev_[tm.calls % 4];
        }
        if (int rc = dec_.run(cs, d_iq, in_pitch, shared_input, (long long)n, osc_, profile_detail ? ev[2] : nullptr, raw, have_oa ? &oa_pre : nullptr)) return rc;
        dec_.done_event = nullptr;
        if (have_oa && dec_.osc_advanced) memset(&oa_pre, 0, sizeof(oa_pre));
    }
    if (profile_detail) PG_HIP(hipEventRecord(ev[3], cs));
    if (bank_pipe) {
        // the decimator's own histories and the oscillators' phases stay on its stream; the rest of the call moves over
        std::vector<TailJob> jobs;
        dec_.tail_jobs_dec(jobs);
        const bool nothing_behind = jobs.empty() && oa_pre.osc == nullptr;
        if (int rc = run_save_tails(stream_, jobs, C, &oa_pre)) return rc;  // (no launch at all behind the bank kernel: nothing left to do)
        // (an event without timing for the hand-over: PEBBLEGPU_BANK_PIPE_TIMED_EV=1 records the call's timing event instead -- A/B)
        static const bool timed_ev = [] { const char *e = getenv("PEBBLEGPU_BANK_PIPE_TIMED_EV"); return e && e[0] == '1'; }();
        if (!sync_ev_[0]) for (hipEvent_t &e : sync_ev_) PG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        pipe_ev_ = timed_ev ? ev[1] : sync_ev_[tm.calls % 4];
        if (!(dec_.done_recorded && nothing_behind && !timed_ev)) PG_HIP(hipEventRecord(pipe_ev_, stream_));
        cs = chain_stream_;
        PG_HIP(hipStreamWaitEvent(cs, pipe_ev_, 0));
        // (with two output buffers the next call's decimator becomes ready with the same event as this band-pass: a short nap lets its
        // one-wave-per-SIMD workgroups be placed before the band-pass fills the CUs -- placed behind them it ran 112 us instead of 65.
        // With three the next decimator is already running when this point is reached: no nap)
        static const int nap_env = [] { const char *e = getenv("PEBBLEGPU_BANK_PIPE_NAP_US"); return e ? (int)(100.0 * atof(e)) : -1; }();
        if (int rc = run_nap(cs, nap_env >= 0 ? (unsigned)nap_env : (rot3 ? 0u : 800u))) return rc;
    }
    const long long nd = dec_.out_len();
    if (zoom_bins) {  // SignalSpectrum::zoomed(m_sampleBuf, numStepSamples), receiver.cpp:884 / :942 (the update timer forced open)
        if (int rc = zoom_.run(cs, dec_.out().data(), dec_.out().pitch, nd / nf, d_zoom)) return rc;
        last_zoom_frames = (uint64_t)(nd / nf);
    }
    if (!wfm) {
        if (int rc = ff_.run(cs, dec_.out(), nd, audio.data(), audio.pitch)) return rc;  // receiver.cpp:950
        if (profile_detail) PG_HIP(hipEventRecord(ev[4], cs));
    }
    // Squelch, receiver.cpp:893-897 / :962-965: below the threshold the reference returns here -- nothing behind the gate
    // runs or changes state, and no audio leaves the call.
    bool gate_closed = false, tails_carried = false;
    if (squelch_db_ > -120.0) {
        if (!last_spec_frames) return fail(PEBBLEGPU_E_INVALID, "the squelch gate needs a spectrum: none has been computed yet");
        PG_HIP(hipMemcpyAsync(h_gate_, d_smeter + (last_spec_frames - 1), sizeof(float4), hipMemcpyDeviceToHost, cs));
        PG_HIP(hipStreamSynchronize(cs));
        gate_closed = (double)h_gate_->y < squelch_db_;  // m_avgDb < m_squelchDb
    }
    // dmNONE, "Tune only mode, no demod or output" (receiver.cpp:968-971): for the reference's own shape (one channel) the call
    // ends here like a closed gate -- nothing behind the band-pass runs or changes state, no audio leaves; in a bank the
    // tune-only channels sit out the noise filter, AGC and demodulators (muted lists) and their audio rows are cleared
    bool tune_only = false;
    if (!wfm && !gate_closed) {
        if (C == 1) tune_only = ctl_[0].mode == PEBBLEGPU_DM_NONE;
    }
    if (gate_closed || tune_only) {
        if (gate_closed) squelched_calls++;
        last_audio_n = 0;
        gate_closed = true;
        if (profile_detail) { if (wfm) PG_HIP(hipEventRecord(ev[4], cs)); }
    } else if (!wfm && bank_gate_) {
        // Per-channel squelch of a bank: the decision is made on the device from the S-meter of each super-frame's last raw frame
        // (no read-back, no stream synchronisation); everything behind the band-pass then runs one super-frame at a time and
        // leaves a closed channel alone -- no output, no state change: the reference's early return (receiver.cpp:962-965) per
        // channel.  A closed (channel, super-frame) reads as silence in the bank's audio rows.
        if (!with_spectrum) return fail(PEBBLEGPU_E_INVALID, "the squelch gate of a bank reads the spectra of the same call: create the bank with spectrum_bins");
        const int k = (int)(n / superframe);
        const long long spf = nd / k;
        if (int rc = run_gate_eval(cs, d_smeter, smeter_pitch, (int)(superframe / nf), k, d_squelch, d_gate, (int)max_sf, C)) return rc;
        for (int j = 0; j < k; j++) {
            const Gate gate{d_gate, (int)max_sf, j};
            float2 *seg = audio.data() + (long long)j * spf;
            if (int rc = anf_.run(cs, seg, audio.pitch, spf, gate)) return rc;
            if (int rc = agc_.run(cs, seg, audio.pitch, spf, gate)) return rc;
            am_.defer_tail = false;
            if (int rc = am_.run(cs, seg, audio.pitch, seg, audio.pitch, spf, gate)) return rc;
            if (sam_.C) { if (int rc = sam_.run(cs, seg, audio.pitch, seg, audio.pitch, spf, gate)) return rc; }
            if (nfm_.C) { if (int rc = nfm_.run(cs, seg, audio.pitch, seg, audio.pitch, spf, gate)) return rc; }
        }
        if (int rc = run_gate_zero(cs, audio.data(), audio.pitch, spf, d_gate, (int)max_sf, C, k)) return rc;
        for (uint32_t ch = 0; ch < C; ch++)
            if (ctl_[ch].mode == PEBBLEGPU_DM_NONE) PG_HIP(hipMemsetAsync(audio.data((int)ch), 0, sizeof(float2) * (size_t)nd, cs));
    } else if (!wfm) {
        if (int rc = anf_.run(cs, audio.data(), audio.pitch, nd)) return rc;  // NoiseFilter::ProcessBlock, receiver.cpp:974
        if (int rc = agc_.run(cs, audio.data(), audio.pitch, nd)) return rc;  // AGC::processBlock, receiver.cpp:983
        // Demod::processBlock, receiver.cpp:987: AM channels are demodulated in place; every other narrow mode returns its input
        am_.defer_tail = bank_pipe;  // (a two-stage call's tail launch carries the AM demodulator's history refresh: one launch fewer)
        if (int rc = am_.run(cs, audio.data(), audio.pitch, audio.data(), audio.pitch, nd)) return rc;
        if (sam_.C) { if (int rc = sam_.run(cs, audio.data(), audio.pitch, audio.data(), audio.pitch, nd)) return rc; }
        if (nfm_.C) { if (int rc = nfm_.run(cs, audio.data(), audio.pitch, audio.data(), audio.pitch, nd)) return rc; }
        for (uint32_t ch = 0; ch < C; ch++)  // clearCPX(m_audioBuf, ...) of the bank's tune-only channels
            if (ctl_[ch].mode == PEBBLEGPU_DM_NONE) PG_HIP(hipMemsetAsync(audio.data((int)ch), 0, sizeof(float2) * (size_t)nd, cs));
    } else {
        if (profile_detail) PG_HIP(hipEventRecord(ev[4], cs));
        // (the call's tail refresh rides on the demodulator's launch: it is the last kernel of the call, run on an idle GPU)
        std::vector<TailJob> jobs;
        dec_.tail_jobs(jobs);
        OscAdvance oa;
        if (int rc = osc_.advance_job(cs, n, &oa)) return rc;
        if (int rc = wfmc_.run(cs, dec_.out().data(), dec_.out().pitch, audio.data(), audio.pitch, nd, &jobs, &oa, &tails_carried)) return rc;  // receiver.cpp:896
    }
    if (!gate_closed) {
        last_audio_n = (uint64_t)nd;
        if (audio_rate) {  // CFractResampler::Resample into the audio buffer, receiver.cpp:1000-1001
            long long n_rs = 0;
            if (int rc = resamp_.run(cs, audio.data(), audio.pitch, nd, d_audio_rs, rs_pitch, &n_rs)) return rc;
            last_audio_n = (uint64_t)n_rs;
        }
    }
    if (profile_detail) PG_HIP(hipEventRecord(ev[5], cs));
    if (bank_pipe) {
        std::vector<TailJob> jobs;
        dec_.tail_job_out(jobs);
        am_.tail_jobs(jobs);
        if (int rc = run_save_tails(cs, jobs, C, nullptr)) return rc;
    } else if (!tails_carried) {  // one launch refreshes every history head-room for the next call
        std::vector<TailJob> jobs;
        dec_.tail_jobs(jobs);
        if (wfm && !gate_closed) wfmc_.tail_jobs(jobs);  // a gated super-frame never reached the demodulator: its history stays
        OscAdvance oa = oa_pre;
        if (!have_oa) { if (int rc = osc_.advance_job(cs, n, &oa)) return rc; }
        if (int rc = run_save_tails(cs, jobs, C, &oa)) return rc;
    }
    if (!bank_pipe) d_end_prev_ = nullptr;
    if (bank_pipe) {
        PG_HIP(hipEventRecord(ev[6], cs));
        chain_end_ = ev[6];   // for whoever needs both stages over: sync(), a call after a setter, a call of another shape
        spec_end_ = pipe_ev_;
        f_end_[2] = f_end_[1];
        f_end_[1] = f_end_[0];
        f_end_[0] = ev[6];
        {   // this call's second stage reads the buffer the decimator has just written
            bool found = false;
            for (auto &pr : out_reader_) if (pr.first == (const void *)dec_.fin.base) { pr.second = ev[6]; found = true; }
            if (!found) out_reader_.push_back({(const void *)dec_.fin.base, ev[6]});
        }
        d_end_prev_ = ev[6];
    } else if (side && pipeline_) {
        // the call's two pipelines end separately: whoever needs both waits for both (sync(), the next call that is not plain)
        PG_HIP(hipEventRecord(ev[6], cs));
        chain_end_ = ev[6];
        spec_end_ = ev[1];
    } else if (side) {
        // join: the call has ended once both pipelines have, and it ends on the chain's stream.  That stream is the main stream
        // of the next call (the two swap roles): its first kernel then follows this call's last in queue order, where a wait
        // on an event from the other queue cost ~25 us of idle GPU per call
        if (!fuse_dec) PG_HIP(hipStreamWaitEvent(cs, ev[1], 0));
        static const bool end_records = [] { const char *e = getenv("PEBBLEGPU_EVENTS"); return e && e[0] == 'f'; }();  // =full: an end record per call (A/B)
        if (end_records) PG_HIP(hipEventRecord(ev[6], cs));
        else tm.open_slot = slot;  // (closed by the next call's start record, by sync() or by a timing query)
        std::swap(stream_, chain_stream_);
    } else {
        PG_HIP(hipEventRecord(ev[6], stream_));
    }
    osc_.advance(n);
    guard.armed = false;
    return 0;
}

int Receiver::process_raw(int fmt, int order, double gain, const void *d_raw, uint64_t n)
{
    if (!d_raw || n == 0) return fail(PEBBLEGPU_E_INVALID, "null input or zero samples");
    if (fmt < 0 || fmt > 4 || order < 0 || order > 3) return fail(PEBBLEGPU_E_INVALID, "unknown sample format %d / IQ order %d", fmt, order);
    if (n > (uint64_t)max_sf * superframe) return fail(PEBBLEGPU_E_SIZE, "%llu samples exceed this object's capacity", (unsigned long long)n);
    double scale = gain;
    if (fmt == 0 || fmt == 1) scale *= 1 / 128.0;        // deviceinterfacebase.cpp:651,689
    else if (fmt == 2) scale *= 1 / 32768.0;             // :729
    else if (fmt == 4) scale *= 1 / 32767.0;             // wavfile.cpp:299-300
    const RawSrc raw{d_raw, fmt, order, (float)scale, 0};
    return process(nullptr, n, bins != 0, true, &raw);
}

// ---- host ingest: pinned double buffer (the library owns the pinned host buffers; the producer side of
// plugins/HackRFDevice/hackrfdevice.cpp:533-566 writes into them instead of into its own ring) ----
int Receiver::ingest_acquire(uint32_t slot, uint64_t bytes, void **host_ptr)
{
    if (slot > 1 || !host_ptr || bytes == 0) return fail(PEBBLEGPU_E_INVALID, "ingest slot is 0 or 1, bytes > 0");
    PG_HIP(hipSetDevice(device));
    IngestSlot &g = ingest_[slot];
    if (g.in_flight) {  // the call that read this slot's device copy (and the upload before it) must be over before the host refills it
        PG_HIP(hipEventSynchronize(g.done_main));
        PG_HIP(hipEventSynchronize(g.done_chain));
        g.in_flight = false;
    }
    if (!copy_stream_) PG_HIP(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
    if (!g.uploaded) {
        PG_HIP(hipEventCreateWithFlags(&g.uploaded, hipEventDisableTiming));
        PG_HIP(hipEventCreateWithFlags(&g.done_main, hipEventDisableTiming));
        PG_HIP(hipEventCreateWithFlags(&g.done_chain, hipEventDisableTiming));
    }
    if (g.cap < bytes) {
        PG_HIP(hipStreamSynchronize(copy_stream_));
        if (g.h) (void)hipHostFree(g.h);
        if (g.d) (void)hipFree(g.d);
        g.h = g.d = nullptr;
        g.cap = 0;
        PG_HIP(hipHostMalloc(&g.h, bytes));
        PG_HIP(hipMalloc(&g.d, bytes));
        g.cap = bytes;
    }
    g.submitted = 0;
    *host_ptr = g.h;
    return 0;
}
int Receiver::ingest_submit(uint32_t slot, uint64_t bytes)
{
    if (slot > 1) return fail(PEBBLEGPU_E_INVALID, "ingest slot is 0 or 1");
    IngestSlot &g = ingest_[slot];
    if (!g.h || bytes == 0 || bytes > g.cap) return fail(PEBBLEGPU_E_SIZE, "%llu bytes do not fit the slot acquired (%zu)", (unsigned long long)bytes, g.cap);
    if (g.in_flight) return fail(PEBBLEGPU_E_INVALID, "the slot's previous call is still in flight: acquire it again first");
    PG_HIP(hipSetDevice(device));
    PG_HIP(hipMemcpyAsync(g.d, g.h, bytes, hipMemcpyHostToDevice, copy_stream_));
    PG_HIP(hipEventRecord(g.uploaded, copy_stream_));
    g.submitted = bytes;
    return 0;
}
int Receiver::process_ingested(uint32_t slot, int fmt, int order, double gain, uint64_t n)
{
    if (slot > 1) return fail(PEBBLEGPU_E_INVALID, "ingest slot is 0 or 1");
    IngestSlot &g = ingest_[slot];
    if (fmt < 0 || fmt > 4) return fail(PEBBLEGPU_E_INVALID, "unknown sample format %d", fmt);
    static const size_t kPair[5] = {2, 2, 4, 8, 4};  // bytes per IQ pair: CPX8, CPXU8, CPX16, CPXFLOAT, WAV PCM16
    if (!g.submitted || (uint64_t)S * n * kPair[fmt] > g.submitted) return fail(PEBBLEGPU_E_SIZE, "the slot holds %zu submitted bytes; %llu samples of this format need more", g.submitted, (unsigned long long)n);
    PG_HIP(hipSetDevice(device));
    // both of the call's streams read the raw samples (the display transform and the chain's first stage convert in their own loads)
    PG_HIP(hipStreamWaitEvent(stream_, g.uploaded, 0));
    PG_HIP(hipStreamWaitEvent(chain_stream_, g.uploaded, 0));
    if (int rc = process_raw(fmt, order, gain, g.d, n)) return rc;
    PG_HIP(hipEventRecord(g.done_main, stream_));
    PG_HIP(hipEventRecord(g.done_chain, chain_stream_));
    g.in_flight = true;
    return 0;
}

const char *Receiver::kernel_name(int which) const
{
    switch (which) {
    case 1: return !bins ? "" : spec_.big ? "k_big256_cols + k_big256_rows" : spec_.per_q ? "k_spectrum_q128" : bins == 8192 ? (spec_.use_w64 ? "k_spectrum_w64" : spec_.last_fullc ? "k_spectrum_t128 (twiddles held)" : "k_spectrum_t128") : bins == 4096 ? "k_spectrum<2>" : "k_spectrum_1to1";
    case 2: return dec_.front_name;
    case 3: return dec_.rest_name;
    case 4: return wfm ? "" : ff_n == 2048 ? "k_fastfir_t128" : "k_fastfir";
    case 5: return wfm ? (wfmc_.fused ? "k_wfm_fir" : "k_iir_scan + k_discrim + k_fir_dec") : "k_anf/k_agc/k_iir_scan/k_pll_demod + k_fir_dec (listed channels only)";
    default: return "";
    }
}

int Receiver::close_timing()
{
    if (tm.open_slot >= 0) {  // the last side-by-side call recorded no end event: it ended on what is now the main stream
        PG_HIP(hipEventRecord(tm.ev[tm.open_slot][6], stream_));
        tm.end_ev[tm.open_slot] = tm.ev[tm.open_slot][6];
        tm.open_slot = -1;
    }
    return 0;
}

int Receiver::sync()
{
    PG_HIP(hipSetDevice(device));
    if (int rc = close_timing()) return rc;
    PG_HIP(hipStreamSynchronize(stream_));
    PG_HIP(hipStreamSynchronize(chain_stream_));
    return 0;
}

// CB_ProcessIQData shape: one frame in; audio appears once a whole super-frame has been collected, exactly where
// the reference stops returning early (receiver.cpp:922-931).
int Receiver::process_iq(const double *iq, uint16_t n, double *audio_out, uint32_t *n_audio, double *spectrum_db)
{
    if (!iq || !n_audio) return fail(PEBBLEGPU_E_INVALID, "null argument");
    if (n != nf) return fail(PEBBLEGPU_E_SIZE, "process_iq takes frames of %u samples", nf);
    if (S != 1) return fail(PEBBLEGPU_E_UNSUPPORTED, "process_iq feeds one stream; this bank has %u", S);
    if (cond_.any || cond_.dirty)
        return fail(PEBBLEGPU_E_UNSUPPORTED, "the input conditioners run on the batched device path (pebblegpu_receiver_process) only");
    PG_HIP(hipSetDevice(device));
    if (!d_stage_in_) PG_HIP(hipMalloc((void **)&d_stage_in_, sizeof(float2) * superframe));
    h_frame_.resize((size_t)nf * 2);
    for (size_t i = 0; i < (size_t)nf * 2; i++) h_frame_[i] = (float)iq[i];
    float2 *dst = d_stage_in_ + acc_frames_ * nf;
    PG_HIP(hipMemcpy(dst, h_frame_.data(), sizeof(float2) * nf, hipMemcpyHostToDevice));
    *n_audio = 0;
    if ((spectrum_db || squelch_db_ > -120.0) && bins) {  // the gate reads the latest frame's spectrum, wanted by the host or not
        if (int rc = process(dst, nf, true, false)) return rc;
    }
    if (spectrum_db && bins) {
        if (int rc = sync()) return rc;
        h_out_.resize(bins);
        PG_HIP(hipMemcpy(h_out_.data(), d_spec, sizeof(float) * bins, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < bins; i++) spectrum_db[i] = (double)h_out_[i];
    }
    acc_frames_++;
    if (acc_frames_ * nf >= superframe) {
        acc_frames_ = 0;
        if (int rc = process(d_stage_in_, superframe, false, true)) return rc;
        if (int rc = sync()) return rc;
        const size_t na = (size_t)last_audio_n;
        h_out_.resize(na * 2);
        PG_HIP(hipMemcpy(h_out_.data(), audio_ptr(), sizeof(float2) * na, hipMemcpyDeviceToHost));
        if (audio_out)
            for (size_t i = 0; i < na * 2; i++) audio_out[i] = (double)h_out_[i];
        *n_audio = (uint32_t)na;
    }
    return 0;
}

}  // namespace pg
