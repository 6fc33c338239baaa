/*
 * pebblegpu.h -- C ABI of libpebblegpu: PebbleSDR's per-frame IQ receive chain on MI355X (gfx950).
 *
 * This is the drop-in boundary for ONE hot path of the reference: what Receiver::processIQData
 * (application/receiver.cpp:758-1009) runs between the device plugin's callback and the audio
 * resampler -- Mixer, Decimator, CFastFIR band-pass, Demod (AM / WFM mono / SSB-CW-DIG pass-through)
 * and the SignalSpectrum FFT.  Plain pointers and sizes only; no C++/Qt/torch types cross it.
 * Every entry point names the reference interface it replaces (paths relative to the reference).
 *
 * Conventions
 *   - host complex buffers are interleaved (re, im) doubles == CPX = std::complex<double>
 *     (pebblelib/cpx.h:96); device complex buffers are interleaved (re, im) floats ("float2").
 *   - all functions return PEBBLEGPU_OK (0) or a negative pebblegpu_status; nothing throws across
 *     the ABI; pebblegpu_last_error() returns text for the calling thread's last failure.
 *   - process_* calls are single-caller per handle (the reference calls processIQData from one
 *     consumer thread, pebblelib/producerconsumer.cpp:101-109); setters may be called from another
 *     thread and take effect at the next process call (= frame boundary).
 *   - ASYNCHRONY: pebblegpu_receiver_process / _process_raw and pebblegpu_streambank_process only QUEUE their kernels on
 *     streams private to the handle and return.  The device buffers behind pebblegpu_receiver_audio / _spectrum /
 *     _signal_strength and pebblegpu_streambank_filtered / _spectrum hold the call's results, and the call's INPUT buffer
 *     may be overwritten, only after pebblegpu_receiver_synchronize / pebblegpu_streambank_synchronize (or any of
 *     pebblegpu_memcpy_h2d / _d2h / pebblegpu_device_synchronize, which wait for all work queued on the device first).
 *     A host that touches those buffers with its own HIP calls must synchronise itself.  Calls on one handle execute in
 *     the order they were made.  A receiver without a display transform runs a call as two overlapping stages on two streams
 *     (two-stage calls) and bounds how far the host may run ahead: such a call returns once the call three
 *     before it has completed.  The host-buffer entry points (pebblegpu_process_iq and every stand-alone step) return
 *     with their results complete.
 *   - the library owns every device buffer it returns; host pointers returned by *_result() stay
 *     valid until the next process call on the same handle (ProcessStep ownership rule,
 *     application/processstep.cpp:12-20).  Inputs are never modified (receiver.cpp:747-755).
 */
#ifndef PEBBLEGPU_H
#define PEBBLEGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PEBBLEGPU_ABI_VERSION 1

typedef enum {
    PEBBLEGPU_OK = 0,
    PEBBLEGPU_E_INVALID = -1,      /* bad argument / bad handle */
    PEBBLEGPU_E_NO_DEVICE = -2,    /* no HIP device: the library never falls back to the CPU */
    PEBBLEGPU_E_HIP = -3,          /* HIP runtime error, see pebblegpu_last_error() */
    PEBBLEGPU_E_FILTER_PARAM = -4, /* CFastFIR "Filter Parameter error": previous taps stay active
                                      (pebblelib/fastfir.cpp:208-216) */
    PEBBLEGPU_E_SIZE = -5,         /* sample count not a whole number of super-frames / too large */
    PEBBLEGPU_E_UNSUPPORTED = -6   /* mode or size outside what this build implements */
} pebblegpu_status;

/* DeviceInterface::DemodMode numeric values, pebblelib/device_interfaces.h:124-138 (a Qt shim maps 1:1) */
typedef enum {
    PEBBLEGPU_DM_AM = 0, PEBBLEGPU_DM_SAM, PEBBLEGPU_DM_FMN, PEBBLEGPU_DM_FMM, PEBBLEGPU_DM_FMS,
    PEBBLEGPU_DM_DSB, PEBBLEGPU_DM_LSB, PEBBLEGPU_DM_USB, PEBBLEGPU_DM_CWL, PEBBLEGPU_DM_CWU,
    PEBBLEGPU_DM_DIGL, PEBBLEGPU_DM_DIGU, PEBBLEGPU_DM_NONE
} pebblegpu_demod_mode;

const char *pebblegpu_last_error(void);
int pebblegpu_abi_version(void);
/* number of HIP devices visible (0 => every create() fails with PEBBLEGPU_E_NO_DEVICE) */
int pebblegpu_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Device memory + stream plumbing (so a host needs no other GPU library to feed the chain)
 * ---------------------------------------------------------------------------------------------- */
int pebblegpu_malloc(int device, size_t bytes, void **dptr);
int pebblegpu_free(int device, void *dptr);
/* both copies first wait for all work queued on the device (the library's private streams included), then copy and block */
int pebblegpu_memcpy_h2d(int device, void *dst, const void *src, size_t bytes);
int pebblegpu_memcpy_d2h(int device, void *dst, const void *src, size_t bytes);
int pebblegpu_memset(int device, void *dst, int value, size_t bytes);
int pebblegpu_device_synchronize(int device);
/* Streaming-copy probe (read + write GB/s of a plain device copy with 16- or 8-byte lanes): the measured HBM ceiling
 * the bench quotes next to the 8 TB/s datasheet peak. */
int pebblegpu_probe_copy_gbps(int device, int lane_bytes, size_t bytes, int iters, float *gbps);
/* Ingest on the device: DeviceInterfaceBase::normalizeIQ (pebblelib/deviceinterfacebase.cpp:648-838) and
 * WavFile::ReadSamples' PCM16 scaling (wavfile.cpp:299-300).  d_src holds n_samples raw IQ pairs, d_dst receives float2.
 * gain = m_userIQGain * m_normalizeIQGain; iq_order = DeviceInterface::IQOrder (device_interfaces.h:140-145). */
typedef enum {
    PEBBLEGPU_IQ_S8 = 0,    /* CPX8  (HackRF):  v / 128 */
    PEBBLEGPU_IQ_U8 = 1,    /* CPXU8 (RTL2832): (v - 128) / 128 */
    PEBBLEGPU_IQ_S16 = 2,   /* CPX16: v / 32768 */
    PEBBLEGPU_IQ_F32 = 3,   /* CPXFLOAT */
    PEBBLEGPU_IQ_WAV16 = 4  /* 16-bit PCM stereo WAV: v / 32767 */
} pebblegpu_iq_format;
typedef enum { PEBBLEGPU_IQO_IQ = 0, PEBBLEGPU_IQO_QI, PEBBLEGPU_IQO_IONLY, PEBBLEGPU_IQO_QONLY } pebblegpu_iq_order;
int pebblegpu_normalize_iq(int device, int format, int iq_order, double gain, const void *d_src, uint64_t n_samples, void *d_dst);

/* ------------------------------------------------------------------------------------------------
 * Receiver bank: C tuned channels over one shared wideband stream, or C independent streams.
 * Replaces Receiver::turnPowerOn's step construction (receiver.cpp:154-264) and
 * Receiver::processIQData's DSP (receiver.cpp:826-987).
 * ---------------------------------------------------------------------------------------------- */
typedef struct pebblegpu_receiver pebblegpu_receiver;

typedef struct {
    uint32_t struct_size;        /* = sizeof(pebblegpu_config) */
    int32_t device;              /* HIP device ordinal */
    double sample_rate;          /* Fs of the device stream, Hz (Key_SampleRate, receiver.cpp:149) */
    uint32_t frames_per_buffer;  /* reference N: settings.cpp:57 default 2048; spectrum window length */
    uint32_t n_channels;         /* C */
    uint32_t shared_input;       /* 1: all channels read stream 0; 0: channel c reads stream c */
    uint32_t wfm;                /* 0: narrow branch (protect 30 kHz + FastFIR, receiver.cpp:903-993)
                                    1: WFM branch (protect 200 kHz, no band-pass, receiver.cpp:854-901) */
    uint32_t spectrum_bins;      /* 0: no spectrum; else FFT size (settings.cpp:59 default 4096) */
    uint32_t fastfir_fft;        /* 0 -> 2048 (fastfir.cpp:65) */
    uint32_t fastfir_taps;       /* 0 -> 1025 (fastfir.cpp:66) */
    uint32_t max_superframes;    /* capacity of one process call, in super-frames (>=1) */
    uint32_t audio_rate;         /* 0: audio stays at the demod rate (resampRate == 1, receiver.cpp:1002-1003); else
                                    Key_AudioOutputSampleRate (receiver.cpp:203, default 11025): the audio buffer is
                                    CFractResampler::Resample(n, demodRate / audio_rate, ...) of the demodulated frames
                                    (receiver.cpp:994-1001, pebblelib/fractresampler.cpp:149-195) */
    uint32_t hires_bins;         /* 0: no zoomed spectrum (m_useHiRes off); else the bin count of SignalSpectrum::zoomed
                                    (application/signalspectrum.cpp:89-113; settings.cpp:61 default 2048): fftSpectrum, BlackmanHarris
                                    over frames_per_buffer samples, of every DECIMATED frame of every channel -- m_sampleBuf at the
                                    demodulator rate, after the gain restore on the narrow branch (receiver.cpp:884, 942) */
    uint32_t reserved[3];
} pebblegpu_config;

int pebblegpu_receiver_create(const pebblegpu_config *cfg, pebblegpu_receiver **out);
int pebblegpu_receiver_destroy(pebblegpu_receiver *rx);

/* What buildDecimationChain (pebblelib/decimator.cpp:64-149) produced for this bank. */
typedef struct {
    double demod_rate;           /* achieved rate (float in the reference) */
    uint32_t demod_rate_int;     /* the int the Receiver stores and designs filters with (receiver.h:165-166) */
    uint32_t dec_by2_stages;     /* Decimator::decBy2Stages() */
    uint32_t total_decimation;   /* D */
    uint32_t chain_len;          /* merged stages */
    uint32_t stage_taps[16];     /* 0 => CIC3 */
    uint32_t stage_stride[16];
    uint64_t superframe;         /* input samples per super-frame = D * frames_per_buffer */
    uint32_t n_streams;
    uint32_t spectrum_bins;      /* after the reference's [2048, 65535] clamp (fft.cpp:72-79) */
} pebblegpu_info;
int pebblegpu_receiver_info(const pebblegpu_receiver *rx, pebblegpu_info *info);

/* Receiver::mixerChanged -> Mixer::setFrequency (receiver.cpp:709-716, mixer.cpp:25-40): negated
 * frequency, oscillator phase AND amplitude reset to (1,0). */
int pebblegpu_set_mixer_freq(pebblegpu_receiver *rx, uint32_t channel, double freq_hz);
/* Receiver::filterChanged (receiver.cpp:658-664): BandPassFilter::setBandPass -> CFastFIR::SetupParameters
 * (lo, hi, offset 0, demod rate) and, for AM channels, Demod_AM::setBandwidth(hi - lo). */
int pebblegpu_set_bandpass(pebblegpu_receiver *rx, uint32_t channel, double lo_hz, double hi_hz);
/* Receiver::demodModeChanged -> Demod::setDemodMode (receiver.cpp:640-655).  Narrow banks accept AM, SAM, FMN and
 * every pass-through mode (DSB/LSB/USB/CWL/CWU/DIGL/DIGU/NONE); WFM banks accept FMM (mono) and FMS.
 * FMS is Demod_WFM::processDataStereo (demod_wfm.cpp:255-365): the discriminator output WITHOUT processDataMono's 75 kHz
 * pre-filter, low-passed, de-emphasised and notched; while the pilot PLL (processPilotPll, :390-430) reports lock at the end of a
 * block of frames_per_buffer samples that block is demultiplexed (left - right = 2 raw sin(2 phase)), otherwise it carries the same
 * signal in both channels.  The PLL's phase detector (:792-821) is discontinuous at the loop's operating point and the lock is lost
 * within the first blocks on every input it has been tried on (pinned on the oracle's line-by-line restatement with clean, noisy,
 * weak and absent pilots at the demodulator rates the receiver runs, tests/test_oracle_pins.py -- other inputs: parity unpinned);
 * the library runs the loop, serially per channel, until the first block that ends without lock and treats the stream as mono from
 * there (the lock average would need seconds of a quiet detector to come back).  The RDS branch of the same function (:296-357:
 * m_RdsDownConvert, the 2400 Hz low-pass, processRdsPll, the biphase matched filter, the bit-rate resonator and slicer,
 * processNewRdsBit's block synchroniser with its burst corrector) runs for every dmFMS channel, in double on the device; its groups are
 * read through pebblegpu_receiver_rds_groups.  Calls and frames of a dmFMS bank must be multiples of the RDS down-converter's
 * decimation (8 up to 312.5 kHz of demodulator rate, 16 from 390.625 kHz on) and at least as long as its widest stage.  Not reproduced: what the GUI makes of a group (rdsdecode.cpp). */
int pebblegpu_set_demod_mode(pebblegpu_receiver *rx, uint32_t channel, int mode);
/* tRDS_GROUPS (application/demod/rbdsconstants.h) */
typedef struct pebblegpu_rds_group { uint16_t block_a, block_b, block_c, block_d; } pebblegpu_rds_group;
/* int Demod_WFM::getNextRdsGroupData(tRDS_GROUPS *), demod_wfm.h:39, as its one caller uses it (Demod::fmStereo, demod.cpp:196-226:
 * ONE call behind every processDataStereo, i.e. per frame of frames_per_buffer demodulator samples): the groups that caller would
 * have taken from m_RdsGroupQueue (RDS_Q_SIZE 100, cleared and stuffed with a zero group after BLOCK_ERROR_LIMIT bad blocks) over the
 * frames processed since the last call of this function, oldest first, and for each the function's return value -- changed[i] != 0:
 * the group differs from the one delivered before it (only those reach the reference's text decoder, and only when block_a != 0).
 * Waits for the receiver's queued work.  *n: entries written (<= cap; the rest stays for the next call). */
int pebblegpu_receiver_rds_groups(pebblegpu_receiver *rx, uint32_t channel, pebblegpu_rds_group *groups, uint8_t *changed, uint32_t cap,
                                  uint32_t *n);
/* int Demod_WFM::getStereoLock(int *pPilotLock), demod_wfm.h:40 (demod_wfm.cpp:436-447): *pilot_lock = m_PilotLocked after the last
 * frame of the channel (0 before its first dmFMS frame), *changed != 0 when that differs from what the previous call of this function
 * reported (the first call reports a change: m_LastPilotLocked starts as the opposite).  Waits for the receiver's queued work. */
int pebblegpu_receiver_stereo_lock(pebblegpu_receiver *rx, uint32_t channel, int *pilot_lock, int *changed);
/* AGC::setAgcMode(mode, threshold) (application/agc.cpp:53-82; Receiver::agcModeChanged/agcThresholdChanged).
 * agc_mode: the reference's AgcMode values.  With PEBBLEGPU_AGC_OFF the threshold is a manual gain slider in dB
 * (amplitude 10^((threshold/5)/20), integer division as written, agc.cpp:239-246; the constructor's OFF/1 is unit
 * gain); otherwise it is the knee (0..120, negated inside).  Narrow banks only: the WFM branch has no AGC step. */
typedef enum pebblegpu_agc_mode {
    PEBBLEGPU_AGC_OFF = 0, PEBBLEGPU_AGC_FAST = 1, PEBBLEGPU_AGC_MED = 2, PEBBLEGPU_AGC_SLOW = 3, PEBBLEGPU_AGC_LONG = 4
} pebblegpu_agc_mode;
int pebblegpu_set_agc(pebblegpu_receiver *rx, uint32_t channel, int agc_mode, int threshold);

/* Batched device path.  d_iq: n_streams x n_samples float2 (stream-major, [stream][time]); n_samples must be
 * k * superframe (k <= max_superframes).  Outputs (library-owned device buffers, valid until the next call):
 *   audio    [channel][k * frames_per_buffer] float2  (re = left, im = right, receiver.cpp:1029); with audio_rate
 *            set, the resampled audio instead: the count pebblegpu_receiver_audio reports (same for all channels)
 *   spectrum [stream][n_samples / frames_per_buffer][bins] float, dB amplitude, -f..+f (fft.cpp:395) */
/* Frames shorter than a stage's tap count: the library streams with exact history for any n_samples that is a whole number of
 * super-frames.  The reference, fed 2048-sample frames at >= 20 Msps, degrades such stages to unfiltered sample dropping and
 * refills their history from indeterminate memory (pebblelib/decimator.cpp:602-625); that fallback is NOT reproduced (DESIGN.md
 * section 4, tests/test_parity_gpu.py::test_decimator_short_frames_stream_exactly).  Queues and returns: see ASYNCHRONY above. */
int pebblegpu_receiver_process(pebblegpu_receiver *rx, const void *d_iq, uint64_t n_samples);
/* The same call fed with the device's own sample format (what ProducerConsumer hands normalizeIQ,
 * deviceinterfacebase.cpp:648-838): d_raw holds n_streams x n_samples raw IQ pairs, stream-major, in `format`
 * (pebblegpu_iq_format) and `iq_order`; they are scaled by the format's constant and `gain` into a library-owned float2
 * buffer on the library's stream (no host synchronisation) and processed as above.  Moves 2-4 bytes per sample over
 * PCIe/HBM on the way in instead of 8. */
int pebblegpu_receiver_process_raw(pebblegpu_receiver *rx, int format, int iq_order, double gain, const void *d_raw, uint64_t n_samples);
/* Host ingest through the library's own pinned buffers (the device plugins' producer side, e.g. the HackRF callback that fills
 * the producer/consumer ring, plugins/HackRFDevice/hackrfdevice.cpp:533-566, writes its raw samples straight into one): two slots,
 * so that the upload of one batch crosses PCIe on a copy stream while the call on the other batch computes.
 *   acquire  -- the slot's pinned host buffer of at least `bytes` (grown on demand); blocks until the last call that read the slot
 *               is over, then the host may fill it.  The pointer stays valid until the next acquire of that slot with a larger size.
 *   submit   -- queues the upload of the first `bytes` of the slot (returns at once; the host must not touch the slot until it
 *               has acquired it again);
 *   process_ingested -- pebblegpu_receiver_process_raw on the uploaded samples, ordered behind the upload on the device (no host
 *               synchronisation).  n_streams x n_samples pairs of `format` must have been submitted.
 * Steady state of a producer: acquire(s), fill, submit(s), process_ingested(s), s ^= 1 -- with the audio of the previous call read
 * in between.  The PCIe link bounds this path (2 bytes per sample for HackRF/RTL pairs); DESIGN.md section 5 has the measured rates. */
int pebblegpu_receiver_ingest_acquire(pebblegpu_receiver *rx, uint32_t slot, uint64_t bytes, void **host_ptr);
int pebblegpu_receiver_ingest_submit(pebblegpu_receiver *rx, uint32_t slot, uint64_t bytes);
int pebblegpu_receiver_process_ingested(pebblegpu_receiver *rx, uint32_t slot, int format, int iq_order, double gain, uint64_t n_samples);
/* returns channel 0's row; channel c starts *pitch_samples float2 further per channel */
const void *pebblegpu_receiver_audio(const pebblegpu_receiver *rx, uint64_t *samples_per_channel, uint64_t *pitch_samples);
const void *pebblegpu_receiver_spectrum(const pebblegpu_receiver *rx, uint64_t *frames_per_stream);
/* the zoomed (hi-res) spectra of the last call: [channel][frames_per_channel][bins] float dB, -f..+f at the demodulator rate */
const void *pebblegpu_receiver_zoom_spectrum(const pebblegpu_receiver *rx, uint64_t *frames_per_channel, uint32_t *bins);
/* Time of the last process call's kernels in ms, from HIP events on the library's stream.  which: 0 whole
 * call; 1 spectrum kernel; 2 mixer+first-decimator kernel; 3 remaining decimator stages; 4 FastFIR;
 * 5 demod.  A call whose chain runs beside its display transform records no end event of its own: which = 0
 * of such a call runs to the start event of the NEXT call on the handle (recorded behind that call's control
 * updates and, for raw input that is staged, its conversion pass), or to the pebblegpu_receiver_synchronize /
 * timing query that closes it -- back to back the sum over calls is the wall time, a single call's figure
 * includes whatever the host let pass before it queued the next one.  PEBBLEGPU_EVENTS=full (read when the
 * receiver is created) gives every call an end event of its own.
 * A receiver WITHOUT a display transform (spectrum_bins = 0) runs a call in two stages on two streams -- mixer + decimator, then
 * band-pass .. resampler -- the second beside the NEXT call's first when calls are queued back to back: which = 0 of such a call
 * runs from the end of the previous call's first stage to the end of its own second stage (its latency; back to back the calls
 * overlap, so the figures sum to more than the wall time).  PEBBLEGPU_BANK_PIPELINE=0 (read when the receiver is created) keeps
 * every call on one stream; set_profiling(rx, 1) does the same for as long as it is on. */
int pebblegpu_receiver_last_ms(const pebblegpu_receiver *rx, int which, float *ms);
/* name(s) of the kernel(s) behind group `which` (1..5) as the last process call ran them ("" when the group is empty): the
 * bench labels its per-kernel roofline lines with these */
const char *pebblegpu_receiver_kernel_name(const pebblegpu_receiver *rx, int which);
/* which 0 and 1 are always available (two event records per call, three with a display transform; 1 reads 0 without one).  The per-kernel splits 2..5 need four more
 * records, each a ~5 us bubble in the stream, so they are recorded only after set_profiling(rx, 1). */
int pebblegpu_receiver_set_profiling(pebblegpu_receiver *rx, int per_kernel);
/* the same, averaged over the last `last_k` process calls (the library keeps events for 64): lets a caller queue calls
 * back to back without a host sync per call and read the kernel times afterwards */
int pebblegpu_receiver_mean_ms(const pebblegpu_receiver *rx, int which, uint32_t last_k, float *ms);
/* Input conditioners, applied in the reference's order to a copy of the stream before the spectrum and the mixer
 * (receiver.cpp:814-823); all default-off.  flags: PEBBLEGPU_COND_* or'ed.  DC: DCRemoval (CIir high-pass 10 Hz, Q
 * 0.7071, dcremoval.cpp:3-19); IQBALANCE: IQBalance::ProcessBlock with setGainFactor/setPhaseFactor values
 * (iqbalance.cpp:65-86); NB1/NB2: NoiseBlanker::ProcessBlock/ProcessBlock2 (noiseblanker.cpp:45-97; switching one on
 * resets its averages as setNbEnabled does).  These are serial-in-time algorithms: the library runs one lane per stream
 * (per frame for IQBALANCE), so they parallelise over banks of streams, not within one.  Batched device path only. */
enum { PEBBLEGPU_COND_DC = 1, PEBBLEGPU_COND_IQBALANCE = 2, PEBBLEGPU_COND_NB1 = 4, PEBBLEGPU_COND_NB2 = 8 };
int pebblegpu_set_conditioners(pebblegpu_receiver *rx, uint32_t stream, int flags, double iq_gain, double iq_phase);
/* NoiseFilter (ANF, 45-tap leaky LMS on a 64-sample delay, noisefilter.cpp:31-88) on a narrow channel, between the
 * band-pass and the AGC (receiver.cpp:974) */
int pebblegpu_set_noise_filter(pebblegpu_receiver *rx, uint32_t channel, int on);
/* S-meter: SignalStrength::fdEstimate (application/signalstrength.cpp:287-380; receiver.cpp:891-892, 959-960) on every
 * frame's unprocessed spectrum, per channel: float4 (peakDb, avgDb, snrDb, floorDb) at [channel * pitch + frame].  The
 * band window is the channel's band-pass (+-100 kHz in a WFM bank) around its mixer frequency.  avgDb is the value the
 * reference's squelch compares with m_squelchDb (receiver.cpp:893-897, 962-965); see pebblegpu_set_squelch.
 * Needs spectrum_bins != 0.  The reference's 10-per-second update timer is forced open: every frame is measured. */
int pebblegpu_receiver_enable_signal_strength(pebblegpu_receiver *rx, int on);
const void *pebblegpu_receiver_signal_strength(const pebblegpu_receiver *rx, uint64_t *frames, uint64_t *pitch_frames);
/* Squelch (Receiver::squelchChanged, receiver.cpp:704-707; the gate at :893-897 for WFM and :962-965 otherwise).  Once a
 * super-frame has been mixed, decimated and (narrow chains) band-passed, avgDb of the unprocessed spectrum of its last raw frame
 * is compared with squelch_db: below it the channel's processing ends there -- noise filter, AGC, demodulator and resampler
 * are not run and keep their state -- exactly the reference's early return.  -120 (DB::minDb, the reference's default) never
 * closes the gate.  Turns the S-meter on; needs spectrum_bins != 0.
 *   - the reference's own shape, one channel called one super-frame at a time (narrow or WFM): the call reports ZERO audio
 *     samples (pebblegpu_receiver_audio's n, process_iq's n_audio); the decision costs one 16-byte read-back and a stream
 *     synchronisation per call, made only while a threshold above -120 is set;
 *   - a narrow bank, or calls of several super-frames: one threshold per channel, the decision per (channel, super-frame) made on
 *     the device from the same call's spectra (no read-back, no synchronisation); a closed (channel, super-frame) reads as
 *     silence in the bank's audio rows (the count is common to all channels), and with audio_rate set the resampler sees that
 *     silence (a single Receiver's resampler would have slept).  A WFM receiver with more than one channel, or one created
 *     for several super-frames per call: a threshold above -120 is PEBBLEGPU_E_UNSUPPORTED (-120 and below, "never closes", is
 *     accepted and does nothing). */
int pebblegpu_set_squelch(pebblegpu_receiver *rx, uint32_t channel, double squelch_db);
/* waits until every process call made on this handle has finished: its outputs are then valid and its input may be reused */
int pebblegpu_receiver_synchronize(pebblegpu_receiver *rx);

/* Host single-frame path with the reference's callback shape:
 *   CB_ProcessIQData  = std::function<void(CPX*, quint16)>  (pebblelib/device_interfaces.h:32)
 *   CB_ProcessAudioData same shape (device_interfaces.h:38).
 * One frame of n == frames_per_buffer samples for channel/stream 0 in; frames accumulate until a whole
 * super-frame is present (the reference returns early until m_sampleBuf is full, receiver.cpp:922-931);
 * then *n_audio = frames_per_buffer (more for FastFIR variants) and audio holds left/right doubles.  Otherwise *n_audio = 0.
 * spectrum_db (may be NULL) receives this frame's dB spectrum (bins doubles). */
int pebblegpu_process_iq(pebblegpu_receiver *rx, const double *iq, uint16_t n, double *audio,
                         uint32_t *n_audio, double *spectrum_db);

/* ------------------------------------------------------------------------------------------------
 * Stream bank: S independent full-rate IQ streams, each through the overlap-save band-pass
 * (CFastFIR::ProcessData, pebblelib/fastfir.cpp:281-334, one filter per stream) and the display
 * transform (FFT::fftSpectrum, pebblelib/fft.cpp:317-374) at the stream rate -- the two transforms of
 * Receiver::processIQData with no tuner/decimator in front (BASELINE.json configs[4]: 1024 streams over 8 GPUs,
 * 128 per GPU, 65536-point spectrum, 2048/1025 band-pass).  Device buffers, one process per GPU, streams shard
 * across ranks with no exchange.  frame/spectrum_bins: 2048-sample frames with 2048/4096/8192 bins
 * (the reference's setup), or 65536/65536, which is past the reference's own m_maxFFTSize clamp
 * (fft.h:21) and uses the same formulas with the clamp lifted.
 * ---------------------------------------------------------------------------------------------- */
typedef struct pebblegpu_streambank pebblegpu_streambank;
typedef struct pebblegpu_streambank_config {
    uint32_t struct_size;
    int32_t device;
    double sample_rate;      /* stream rate, used by the band-pass design (fastfir.cpp:186-261) */
    uint32_t n_streams;
    uint32_t frame;          /* samples per spectrum frame */
    uint32_t spectrum_bins;
    uint32_t fastfir_fft;    /* 0 -> 2048 */
    uint32_t fastfir_taps;   /* 0 -> 1025 */
    uint32_t max_frames;     /* capacity per call, frames per stream */
    uint32_t reserved[5];
} pebblegpu_streambank_config;
int pebblegpu_streambank_create(const pebblegpu_streambank_config *cfg, pebblegpu_streambank **out);
int pebblegpu_streambank_destroy(pebblegpu_streambank *sb);
/* CFastFIR::SetupParameters(lo, hi, 0, sample_rate) for one stream; E_FILTER_PARAM on the reference's
 * "Filter Parameter error" (fastfir.cpp:201-208), the previous filter stays in place */
int pebblegpu_streambank_set_bandpass(pebblegpu_streambank *sb, uint32_t stream, double lo, double hi);
/* d_iq: [stream][n_samples] float2, n_samples a multiple of frame.  what: bit 0 band-pass, bit 1 spectrum */
int pebblegpu_streambank_process(pebblegpu_streambank *sb, const void *d_iq, uint64_t n_samples, uint32_t what);
/* filtered [stream][n_samples] float2 (row pitch returned); spectrum [stream][frames][bins] float dB */
const void *pebblegpu_streambank_filtered(const pebblegpu_streambank *sb, uint64_t *samples_per_stream, uint64_t *pitch_samples);
const void *pebblegpu_streambank_spectrum(const pebblegpu_streambank *sb, uint64_t *frames_per_stream, uint32_t *bins);
/* which: 0 whole call, 1 band-pass kernel, 2 spectrum kernels */
int pebblegpu_streambank_last_ms(const pebblegpu_streambank *sb, int which, float *ms);
int pebblegpu_streambank_synchronize(pebblegpu_streambank *sb);

/* ------------------------------------------------------------------------------------------------
 * Stand-alone process steps with the reference's per-class call shapes, host buffers in and out.
 * These back the C++ adapter classes in include/pebblegpu_steps.hpp.
 * ---------------------------------------------------------------------------------------------- */
typedef struct pebblegpu_mixer pebblegpu_mixer;
/* Mixer::Mixer(sampleRate, bufferSize), mixer.cpp:5-17 */
int pebblegpu_mixer_create(int device, uint32_t sample_rate, uint32_t buffer_size, pebblegpu_mixer **out);
int pebblegpu_mixer_destroy(pebblegpu_mixer *m);
int pebblegpu_mixer_set_frequency(pebblegpu_mixer *m, double f);                 /* mixer.cpp:25-40 */
/* CPX *Mixer::processBlock(CPX *in), pebblelib/mixer.h:15: *out = library buffer, or = in when f == 0 */
int pebblegpu_mixer_process(pebblegpu_mixer *m, const double *in, const double **out);

typedef struct pebblegpu_decimator pebblegpu_decimator;
/* Decimator::Decimator + buildDecimationChain, decimator.cpp:6-46, 64-149 */
int pebblegpu_decimator_create(int device, uint32_t sample_rate, uint32_t buffer_size, pebblegpu_decimator **out);
int pebblegpu_decimator_destroy(pebblegpu_decimator *d);
int pebblegpu_decimator_build_chain(pebblegpu_decimator *d, uint32_t sample_rate_in, uint32_t protect_bw,
                                    uint32_t sample_rate_out, float *achieved_rate);
int pebblegpu_decimator_dec_by2_stages(const pebblegpu_decimator *d, uint32_t *stages);
/* quint32 Decimator::process(CPX *in, CPX *out, quint32 n), pebblelib/decimator.h:238.  n must be a multiple of
 * the total decimation.  Streams with exact history for any such n: a frame shorter than a stage's tap count is NOT
 * degraded to sample dropping (the reference's fallback, decimator.cpp:602-625, reads indeterminate memory) --
 * see DESIGN.md section 4. */
int pebblegpu_decimator_process(pebblegpu_decimator *d, const double *in, double *out, uint32_t n, uint32_t *n_out);

typedef struct pebblegpu_downconvert pebblegpu_downconvert;
/* CDownConvert (pebblelib/downconvert.h:25-50, downconvert.cpp): the alternate mixer + decimator -- a quadrature oscillator
 * (the same recurrence as Mixer, but SetFrequency keeps its phasor and there is no "frequency 0 returns the input" exit) and a
 * cascade of decimate-by-2 stages picked per octave: CIC3, a fixed 11-tap halfband, 15..51-tap halfbands whose DecBy2 counts
 * tap 0 twice (downconvert.cpp:368-376: reproduced).  max_in_length: the largest InLength a ProcessData call will pass. */
int pebblegpu_downconvert_create(int device, uint32_t max_in_length, pebblegpu_downconvert **out);
int pebblegpu_downconvert_destroy(pebblegpu_downconvert *d);
/* TYPEREAL SetDataRate(InRate, MaxBW) (simple = 0, downconvert.cpp:139-206) / SetDataRateSimple (simple = 1, :213-237): builds
 * the stage list when either argument changed, returns the output rate.  As in the reference the call ends with
 * SetFrequency(m_NcoFreq) on the STORED frequency, which mirrors an earlier tuning (call it first, as receiver.cpp:198 does).
 * More than nine stages (the reference's pointer array holds ten entries including the terminating NULL): E_UNSUPPORTED. */
int pebblegpu_downconvert_set_data_rate(pebblegpu_downconvert *d, double in_rate, double max_bw, int simple, double *out_rate);
int pebblegpu_downconvert_set_frequency(pebblegpu_downconvert *d, double nco_freq);   /* SetFrequency, downconvert.cpp:100-112 */
int pebblegpu_downconvert_set_cw_offset(pebblegpu_downconvert *d, double offset);     /* SetCwOffset, downconvert.h:34 */
/* the stage list: taps[j] = tap count of stage j, 0 for the CIC3 */
int pebblegpu_downconvert_stages(const pebblegpu_downconvert *d, uint32_t *n_stages, uint32_t *taps, uint32_t taps_cap);
/* int ProcessData(int InLength, TYPECPX *pInData, TYPECPX *pOutData), downconvert.cpp:250-335: InLength a multiple of 2^stages;
 * returns the output count in *n_out.  The input is NOT modified (the reference mixes it in place).  Streams with exact history for
 * any such InLength: a call that leaves a stage fewer samples than it has taps is not skipped as the reference's "safety net" does
 * (:361-362, which returns stale samples). */
int pebblegpu_downconvert_process(pebblegpu_downconvert *d, uint32_t in_length, const double *in, double *out, uint32_t *n_out);
/* the same on device buffers (float2 in; *d_out: library-owned float2 row, valid until the next call); queues and returns */
int pebblegpu_downconvert_process_device(pebblegpu_downconvert *d, const void *d_iq, uint32_t in_length, const void **d_out, uint32_t *n_out);
int pebblegpu_downconvert_synchronize(pebblegpu_downconvert *d);

typedef struct pebblegpu_fastfir pebblegpu_fastfir;
/* CFastFIR::CFastFIR, fastfir.cpp:77-145 (fft/fir sizes are #defines there; 0,0 -> 2048,1025) */
int pebblegpu_fastfir_create(int device, uint32_t fft_size, uint32_t fir_size, pebblegpu_fastfir **out);
int pebblegpu_fastfir_destroy(pebblegpu_fastfir *f);
/* void CFastFIR::SetupParameters(FLoCut, FHiCut, Offset, SampleRate), pebblelib/fastfir.h:57 */
int pebblegpu_fastfir_setup(pebblegpu_fastfir *f, double lo, double hi, double offset, double sample_rate);
/* int CFastFIR::ProcessData(int InLength, CPX *in, CPX *out), pebblelib/fastfir.h:59: *n_out samples written */
int pebblegpu_fastfir_process(pebblegpu_fastfir *f, int n, const double *in, double *out, int *n_out);

typedef struct pebblegpu_demod pebblegpu_demod;
/* Demod::Demod(sampleRate, wfmSampleRate, bufferSize), application/demod.cpp:49-69 */
int pebblegpu_demod_create(int device, uint32_t sample_rate, uint32_t wfm_sample_rate, uint32_t buffer_size,
                           pebblegpu_demod **out);
int pebblegpu_demod_destroy(pebblegpu_demod *d);
int pebblegpu_demod_set_mode(pebblegpu_demod *d, int mode);          /* demod.cpp:241-257 */
int pebblegpu_demod_set_bandwidth(pebblegpu_demod *d, double bw);    /* demod.cpp:230-239 */
/* CPX *Demod::processBlock(CPX *in, int n), application/demod.h:33: *out = library buffer, or = in for the
 * pass-through modes (demod.cpp:127-138) */
int pebblegpu_demod_process(pebblegpu_demod *d, const double *in, int n, const double **out);
/* dmFMS: getNextRdsGroupData as above, a processBlock call being one frame; and m_RdsData (the matched filter's output the bit
 * slicer reads, demod_wfm.cpp:309) of the last processBlock call: *n its length, at most cap values copied */
int pebblegpu_demod_rds_groups(pebblegpu_demod *d, pebblegpu_rds_group *groups, uint8_t *changed, uint32_t cap, uint32_t *n);
int pebblegpu_demod_rds_signal(pebblegpu_demod *d, double *data, uint32_t cap, uint32_t *n);
int pebblegpu_demod_stereo_lock(pebblegpu_demod *d, int *pilot_lock, int *changed);  /* getStereoLock, as above */

typedef struct pebblegpu_spectrum pebblegpu_spectrum;
/* FFT::factory + fftParams(fftSize, 0, sampleRate, samplesPerBuffer, BLACKMANHARRIS), fft.cpp:45-118 */
int pebblegpu_spectrum_create(int device, uint32_t fft_size, double sample_rate, uint32_t samples_per_buffer,
                              pebblegpu_spectrum **out);
int pebblegpu_spectrum_destroy(pebblegpu_spectrum *s);
int pebblegpu_spectrum_bins(const pebblegpu_spectrum *s, uint32_t *bins);
/* bool FFT::fftSpectrum(CPX *in, double *out, int numSamples), pebblelib/fft.h:38; *overload = return value */
int pebblegpu_spectrum_process(pebblegpu_spectrum *s, const double *in, int n, double *out_db, int *overload);

#ifdef __cplusplus
}
#endif
#endif /* PEBBLEGPU_H */
