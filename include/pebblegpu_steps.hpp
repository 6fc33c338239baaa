// pebblegpu_steps.hpp -- header-only C++ adapters over the C ABI (pebblegpu.h) with the reference's own
// call shapes, so application/receiver.cpp could swap these classes in for pebblelib's, plus a Qt-free
// stand-in for the FileSDRDevice plugin that feeds any CB_ProcessIQData-shaped callback from an IQ .wav.
//
//   reference interface                                                 adapter here
//   ------------------------------------------------------------------  -----------------------------------
//   using CPX = std::complex<double>            pebblelib/cpx.h:96       pebblegpu::CPX
//   CPX *ProcessStep::process(CPX*, quint32)    application/processstep.h:24   ProcessStep::process
//   CPX *Mixer::processBlock(CPX*) / setFrequency   pebblelib/mixer.h:15-16    Mixer
//   float Decimator::buildDecimationChain / quint32 process / decBy2Stages
//                                               pebblelib/decimator.h:236-239  Decimator
//   void CFastFIR::SetupParameters / int ProcessData    pebblelib/fastfir.h:57-59  CFastFIR
//   BandPassFilter::setBandPass / process       application/bandpassfilter.h    BandPassFilter
//   CPX *Demod::processBlock(CPX*, int) / setDemodMode / setBandwidth
//                                               application/demod.h:33-40       Demod
//   FFT::fftParams / bool fftSpectrum(CPX*, double*, int)   pebblelib/fft.h:30-38   FFT
//   void Receiver::processIQData(CPX*, quint16) application/receiver.cpp:758    Receiver::processIQData
//   CB_ProcessIQData / CB_ProcessAudioData      pebblelib/device_interfaces.h:32,38   same std::function shapes
//   FileSDRDevice (initialize / Cmd_Start pump) plugins/FileSDRDevice/filesdrdevice.cpp:24-33,226-289   FileSdrFeeder
//
// Error behaviour follows the reference: no exceptions across step calls; a failing call logs to stderr (the
// reference uses qDebug) and returns the input pointer / zero count; lastStatus() exposes the C status code.
#ifndef PEBBLEGPU_STEPS_HPP
#define PEBBLEGPU_STEPS_HPP
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include "pebblegpu.h"

namespace pebblegpu {

using CPX = std::complex<double>;
static_assert(sizeof(CPX) == 2 * sizeof(double), "CPX must be interleaved doubles");

typedef std::function<void(CPX *, uint16_t)> CB_ProcessIQData;     // device_interfaces.h:32
typedef std::function<void(CPX *, uint16_t)> CB_ProcessAudioData;  // device_interfaces.h:38

enum DemodMode { dmAM = 0, dmSAM, dmFMN, dmFMM, dmFMS, dmDSB, dmLSB, dmUSB, dmCWL, dmCWU, dmDIGL, dmDIGU, dmNONE };  // :124-138

inline int report(const char *what, int rc)
{
    if (rc != 0) std::fprintf(stderr, "pebblegpu: %s failed (%d): %s\n", what, rc, pebblegpu_last_error());
    return rc;
}

// application/processstep.{h,cpp}: owns nothing here (the library owns the buffers); keeps the enable flag contract
class ProcessStep {
public:
    ProcessStep(uint32_t sampleRate_, uint32_t bufferSize_) : sampleRate(sampleRate_), numSamples(bufferSize_), bufferSize(bufferSize_) {}
    virtual ~ProcessStep() {}
    virtual CPX *process(CPX *in, uint32_t) { return in; }
    uint32_t getSampleRate() const { return sampleRate; }
    uint32_t getBufferSize() const { return bufferSize; }
    void enableStep(bool e) { enabled = e; }
    bool isEnabled() const { return enabled; }
    int lastStatus() const { return status; }

protected:
    uint32_t sampleRate, numSamples, bufferSize;
    bool enabled = false;
    int status = 0;
};

class Mixer {
public:
    Mixer(uint32_t sampleRate, uint32_t bufferSize, int device = 0) { status = report("mixer_create", pebblegpu_mixer_create(device, sampleRate, bufferSize, &h)); }
    ~Mixer() { pebblegpu_mixer_destroy(h); }
    Mixer(const Mixer &) = delete;
    Mixer &operator=(const Mixer &) = delete;
    void setFrequency(double f) { if (h) status = report("mixer_set_frequency", pebblegpu_mixer_set_frequency(h, f)); }
    CPX *processBlock(CPX *in)
    {
        const double *out = nullptr;
        if (!h || (status = report("mixer_process", pebblegpu_mixer_process(h, reinterpret_cast<const double *>(in), &out))) != 0) return in;
        return reinterpret_cast<CPX *>(const_cast<double *>(out));
    }
    int lastStatus() const { return status; }

private:
    pebblegpu_mixer *h = nullptr;
    int status = 0;
};

class Decimator {
public:
    Decimator(uint32_t sampleRate, uint32_t bufferSize, int device = 0) { status = report("decimator_create", pebblegpu_decimator_create(device, sampleRate, bufferSize, &h)); }
    ~Decimator() { pebblegpu_decimator_destroy(h); }
    Decimator(const Decimator &) = delete;
    Decimator &operator=(const Decimator &) = delete;
    float buildDecimationChain(uint32_t sampleRateIn, uint32_t protectBw, uint32_t sampleRateOut = 0)
    {
        float r = (float)sampleRateIn;
        if (h) status = report("decimator_build_chain", pebblegpu_decimator_build_chain(h, sampleRateIn, protectBw, sampleRateOut, &r));
        return r;
    }
    uint32_t process(CPX *in, CPX *out, uint32_t numSamples)
    {
        uint32_t n = 0;
        if (!h) return 0;
        status = report("decimator_process", pebblegpu_decimator_process(h, reinterpret_cast<const double *>(in), reinterpret_cast<double *>(out), numSamples, &n));
        return status ? 0 : n;
    }
    uint32_t decBy2Stages()
    {
        uint32_t s = 0;
        if (h) pebblegpu_decimator_dec_by2_stages(h, &s);
        return s;
    }
    int lastStatus() const { return status; }

private:
    pebblegpu_decimator *h = nullptr;
    int status = 0;
};

// CDownConvert (pebblelib/downconvert.h:25-50): same member names and argument meaning; TYPECPX = CPX
class CDownConvert {
public:
    explicit CDownConvert(uint32_t maxInLength = 65536, int device = 0) { status = report("downconvert_create", pebblegpu_downconvert_create(device, maxInLength, &h)); }
    ~CDownConvert() { pebblegpu_downconvert_destroy(h); }
    CDownConvert(const CDownConvert &) = delete;
    CDownConvert &operator=(const CDownConvert &) = delete;
    void SetFrequency(double NcoFreq) { if (h) status = report("downconvert_set_frequency", pebblegpu_downconvert_set_frequency(h, NcoFreq)); }
    void SetCwOffset(double offset) { if (h) status = report("downconvert_set_cw_offset", pebblegpu_downconvert_set_cw_offset(h, offset)); }
    double SetDataRate(double InRate, double MaxBW)
    {
        double r = InRate;
        if (h) status = report("downconvert_set_data_rate", pebblegpu_downconvert_set_data_rate(h, InRate, MaxBW, 0, &r));
        return r;
    }
    double SetDataRateSimple(double InRate, double MaxBW)
    {
        double r = InRate;
        if (h) status = report("downconvert_set_data_rate", pebblegpu_downconvert_set_data_rate(h, InRate, MaxBW, 1, &r));
        return r;
    }
    // returns the number of samples written to pOutData; pInData is left as it was (the reference mixes it in place)
    int ProcessData(int InLength, CPX *pInData, CPX *pOutData)
    {
        uint32_t n = 0;
        if (!h || InLength <= 0) return 0;
        status = report("downconvert_process", pebblegpu_downconvert_process(h, (uint32_t)InLength, reinterpret_cast<const double *>(pInData), reinterpret_cast<double *>(pOutData), &n));
        return status ? 0 : (int)n;
    }
    int lastStatus() const { return status; }

private:
    pebblegpu_downconvert *h = nullptr;
    int status = 0;
};

class CFastFIR {
public:
    explicit CFastFIR(uint32_t fftSize = 0, uint32_t firSize = 0, int device = 0) { status = report("fastfir_create", pebblegpu_fastfir_create(device, fftSize, firSize, &h)); }
    ~CFastFIR() { pebblegpu_fastfir_destroy(h); }
    CFastFIR(const CFastFIR &) = delete;
    CFastFIR &operator=(const CFastFIR &) = delete;
    void SetupParameters(double FLoCut, double FHiCut, double Offset, double SampleRate)
    {
        if (!h) return;
        status = pebblegpu_fastfir_setup(h, FLoCut, FHiCut, Offset, SampleRate);
        if (status == PEBBLEGPU_E_FILTER_PARAM) std::fprintf(stderr, "Filter Parameter error\n");  // fastfir.cpp:214
        else report("fastfir_setup", status);
    }
    int ProcessData(int InLength, CPX *InBuf, CPX *OutBuf)
    {
        int n = 0;
        if (!h) return 0;
        status = report("fastfir_process", pebblegpu_fastfir_process(h, InLength, reinterpret_cast<const double *>(InBuf), reinterpret_cast<double *>(OutBuf), &n));
        return status ? 0 : n;
    }
    int lastStatus() const { return status; }

private:
    pebblegpu_fastfir *h = nullptr;
    int status = 0;
};

// application/bandpassfilter.{h,cpp} with m_useFastFIR = true
class BandPassFilter : public ProcessStep {
public:
    BandPassFilter(uint32_t sampleRate_, uint32_t bufferSize_, int device = 0) : ProcessStep(sampleRate_, bufferSize_), fir(0, 0, device), out(bufferSize_ + 2048) {}
    void setBandPass(float low, float high)
    {
        lowFreq_ = low;
        highFreq_ = high;
        fir.SetupParameters(low, high, 0, sampleRate);  // bandpassfilter.cpp:43
    }
    CPX *process(CPX *in, uint32_t n) override
    {
        fir.ProcessData((int)n, in, out.data());  // the count is ignored, as in bandpassfilter.cpp:53-56
        status = fir.lastStatus();
        return out.data();
    }
    float lowFreq() const { return lowFreq_; }
    float highFreq() const { return highFreq_; }

private:
    CFastFIR fir;
    std::vector<CPX> out;
    float lowFreq_ = 0, highFreq_ = 0;
};

class Demod : public ProcessStep {
public:
    Demod(uint32_t sampleRate_, uint32_t wfmSampleRate, uint32_t bufferSize_, int device = 0) : ProcessStep(sampleRate_, bufferSize_)
    {
        status = report("demod_create", pebblegpu_demod_create(device, sampleRate_, wfmSampleRate, bufferSize_, &h));
    }
    ~Demod() override { pebblegpu_demod_destroy(h); }
    Demod(const Demod &) = delete;
    Demod &operator=(const Demod &) = delete;
    void setDemodMode(DemodMode m, int /*sourceSampleRate*/ = 0, int /*audioSampleRate*/ = 0)
    {
        mode = m;
        if (h) status = report("demod_set_mode", pebblegpu_demod_set_mode(h, (int)m));
    }
    DemodMode demodMode() const { return mode; }
    void setBandwidth(double bw) { if (h) status = report("demod_set_bandwidth", pebblegpu_demod_set_bandwidth(h, bw)); }
    CPX *processBlock(CPX *in, int bufSize)
    {
        const double *out = nullptr;
        if (!h || (status = report("demod_process", pebblegpu_demod_process(h, reinterpret_cast<const double *>(in), bufSize, &out))) != 0) return in;
        return reinterpret_cast<CPX *>(const_cast<double *>(out));
    }
    // int Demod_WFM::getNextRdsGroupData(tRDS_GROUPS *), demod_wfm.h:39, as Demod::fmStereo calls it (demod.cpp:207-219: once behind
    // every processDataStereo): 0 while nothing is queued or the group repeats the one before it, else 1 with the group in *g -- the
    // value to hand to CRdsDecode::decodeRdsGroup when BlockA != 0.  (The library has popped one group per processBlock call already;
    // this walks that list.)
    int getStereoLock(int *pPilotLock)  // Demod_WFM::getStereoLock, demod_wfm.h:40
    {
        int lock = 0, changed = 0;
        if (!h || pebblegpu_demod_stereo_lock(h, &lock, &changed) != 0) return 0;
        if (pPilotLock) *pPilotLock = lock;
        return changed;
    }
    int getNextRdsGroupData(pebblegpu_rds_group *g)
    {
        uint32_t n = 0;
        uint8_t changed = 0;
        if (!h || !g || pebblegpu_demod_rds_groups(h, g, &changed, 1, &n) != 0 || n == 0) return 0;
        return changed ? 1 : 0;
    }

private:
    pebblegpu_demod *h = nullptr;
    DemodMode mode = dmAM;
};

// pebblelib/fft.h as SignalSpectrum uses it (factory + fftParams + fftSpectrum); window type is BLACKMANHARRIS
class FFT {
public:
    explicit FFT(int device_ = 0) : device(device_) {}
    ~FFT() { pebblegpu_spectrum_destroy(h); }
    FFT(const FFT &) = delete;
    FFT &operator=(const FFT &) = delete;
    void fftParams(uint32_t fftSize, double /*dBCompensation*/, double sampleRate, int samplesPerBuffer)
    {
        pebblegpu_spectrum_destroy(h);
        h = nullptr;
        status = report("spectrum_create", pebblegpu_spectrum_create(device, fftSize, sampleRate, (uint32_t)samplesPerBuffer, &h));
        bins = 0;
        if (h) pebblegpu_spectrum_bins(h, &bins);
    }
    int getFFTSize() const { return (int)bins; }
    bool fftSpectrum(CPX *in, double *out, int numSamples)
    {
        int ov = 0;
        if (!h) return false;  // "if (!m_fftParamsSet) return false;"
        status = report("spectrum_process", pebblegpu_spectrum_process(h, reinterpret_cast<const double *>(in), numSamples, out, &ov));
        return ov != 0;
    }
    int lastStatus() const { return status; }

private:
    pebblegpu_spectrum *h = nullptr;
    uint32_t bins = 0;
    int device, status = 0;
};

// The slice of application/receiver.cpp this library replaces: turnPowerOn's step construction and
// processIQData's DSP for one tuned channel, audio delivered through the CB_ProcessAudioData-shaped callback.
class Receiver {
public:
    // audioOutRate: Key_AudioOutputSampleRate (receiver.cpp:203); 0 keeps the audio at the demod rate
    Receiver(uint32_t sampleRate, uint16_t framesPerBuffer, bool wfm, uint32_t spectrumBins, CB_ProcessAudioData audioCb,
             uint32_t fastfirFft = 0, uint32_t fastfirTaps = 0, int device = 0, uint32_t audioOutRate = 0)
        : n(framesPerBuffer), cb(audioCb)
    {
        pebblegpu_config cfg;
        std::memset(&cfg, 0, sizeof(cfg));
        cfg.struct_size = sizeof(cfg);
        cfg.device = device;
        cfg.sample_rate = sampleRate;
        cfg.frames_per_buffer = framesPerBuffer;
        cfg.n_channels = 1;
        cfg.shared_input = 1;
        cfg.wfm = wfm ? 1 : 0;
        cfg.spectrum_bins = spectrumBins;
        cfg.fastfir_fft = fastfirFft;
        cfg.fastfir_taps = fastfirTaps;
        cfg.max_superframes = 1;
        cfg.audio_rate = audioOutRate;
        status = report("receiver_create", pebblegpu_receiver_create(&cfg, &h));
        pebblegpu_info info;
        if (h && pebblegpu_receiver_info(h, &info) == 0) {
            audio.resize((size_t)(info.superframe / info.total_decimation) + framesPerBuffer);
            spectrum.resize(info.spectrum_bins);
            demodRate = info.demod_rate_int;
        }
    }
    ~Receiver() { pebblegpu_receiver_destroy(h); }
    Receiver(const Receiver &) = delete;
    Receiver &operator=(const Receiver &) = delete;
    void mixerChanged(int f) { if (h) status = report("set_mixer_freq", pebblegpu_set_mixer_freq(h, 0, f)); }               // receiver.cpp:709
    void filterChanged(int lo, int hi) { if (h) status = report("set_bandpass", pebblegpu_set_bandpass(h, 0, lo, hi)); }    // receiver.cpp:658
    void demodModeChanged(DemodMode m) { if (h) status = report("set_demod_mode", pebblegpu_set_demod_mode(h, 0, (int)m)); } // receiver.cpp:640
    // agcModeChanged / agcThresholdChanged -> AGC::setAgcMode(mode, threshold) (agc.cpp:53-82)
    void agcModeChanged(int agcMode, int threshold) { if (h) status = report("set_agc", pebblegpu_set_agc(h, 0, agcMode, threshold)); }
    void squelchChanged(double s) { if (h) status = report("set_squelch", pebblegpu_set_squelch(h, 0, s)); }                // receiver.cpp:704
    // bound as the device plugin's CB_ProcessIQData, like receiver.cpp:135-138
    void processIQData(CPX *in, uint16_t numSamples)
    {
        if (!h) return;
        uint32_t na = 0;
        status = report("process_iq", pebblegpu_process_iq(h, reinterpret_cast<const double *>(in), numSamples, reinterpret_cast<double *>(audio.data()), &na,
                                                          spectrum.empty() ? nullptr : spectrum.data()));
        if (status == 0 && na > 0 && cb) {
            for (uint32_t off = 0; off < na; off += n) cb(audio.data() + off, (uint16_t)((na - off) < n ? (na - off) : n));  // processAudioData, receiver.cpp:1007
        }
    }
    const std::vector<double> &unprocessedSpectrum() const { return spectrum; }  // SignalSpectrum::getUnprocessed
    uint32_t demodSampleRate() const { return demodRate; }
    int lastStatus() const { return status; }

private:
    pebblegpu_receiver *h = nullptr;
    uint16_t n;
    CB_ProcessAudioData cb;
    std::vector<CPX> audio;
    std::vector<double> spectrum;
    uint32_t demodRate = 0;
    int status = 0;
};

// Qt-free stand-in for plugins/FileSDRDevice: reads a RIFF/WAVE IQ recording (16-bit PCM stereo, /32767 as
// wavfile.cpp:299-300, or float32 stereo) and pumps framesPerBuffer-sized CPX frames into the callback the host bound
// with initialize().  start() runs the whole file synchronously (the reference paces it in real time through
// ProducerConsumer; pacing is host threading, out of scope -- SURVEY.md 2.1).
class FileSdrFeeder {
public:
    bool initialize(CB_ProcessIQData callback, uint16_t framesPerBuffer_)
    {
        cb = callback;
        framesPerBuffer = framesPerBuffer_;
        return true;
    }
    bool connectDevice(const std::string &fileName)
    {
        f = std::fopen(fileName.c_str(), "rb");
        if (!f) return false;
        unsigned char hdr[12];
        if (std::fread(hdr, 1, 12, f) != 12 || std::memcmp(hdr, "RIFF", 4) || std::memcmp(hdr + 8, "WAVE", 4)) return closeFail();
        bool gotFmt = false;
        for (;;) {  // loop over sub-chunks until "data" (wavfile.cpp:66-140)
            unsigned char ck[8];
            if (std::fread(ck, 1, 8, f) != 8) return closeFail();
            const uint32_t size = ck[4] | (ck[5] << 8) | (ck[6] << 16) | ((uint32_t)ck[7] << 24);
            if (!std::memcmp(ck, "fmt ", 4)) {
                unsigned char fm[16];
                if (size < 16 || std::fread(fm, 1, 16, f) != 16) return closeFail();
                format = fm[0] | (fm[1] << 8);
                channels = fm[2] | (fm[3] << 8);
                sampleRate = fm[4] | (fm[5] << 8) | (fm[6] << 16) | ((uint32_t)fm[7] << 24);
                bits = fm[14] | (fm[15] << 8);
                std::fseek(f, (long)(size - 16 + (size & 1)), SEEK_CUR);
                gotFmt = true;
            } else if (!std::memcmp(ck, "data", 4)) {
                dataStart = std::ftell(f);
                dataBytes = size;
                break;
            } else {
                std::fseek(f, (long)(size + (size & 1)), SEEK_CUR);
            }
        }
        if (!gotFmt || channels != 2 || !((format == 1 && bits == 16) || (format == 3 && bits == 32))) return closeFail();
        return true;
    }
    uint32_t getSampleRate() const { return sampleRate; }
    void setIQGain(double g) { gain = g; }       // Key_IQGain, applied by normalizeIQ (deviceinterfacebase.cpp:532-)
    void setIQSwap(bool s) { swapIQ = s; }       // IQO_QI
    // Cmd_Start: deliver every whole frame of the file once; returns the number of frames delivered
    uint64_t start()
    {
        if (!f || !cb || !framesPerBuffer) return 0;
        std::fseek(f, dataStart, SEEK_SET);
        const size_t bps = (format == 1) ? 4 : 8;
        std::vector<unsigned char> raw(bps * framesPerBuffer);
        std::vector<CPX> frame(framesPerBuffer);
        uint64_t frames = 0, left = dataBytes;
        while (left >= raw.size() && std::fread(raw.data(), 1, raw.size(), f) == raw.size()) {
            left -= raw.size();
            for (uint32_t i = 0; i < framesPerBuffer; i++) {
                double l, r;
                if (format == 1) {
                    const int16_t a = (int16_t)(raw[4 * i] | (raw[4 * i + 1] << 8)), b = (int16_t)(raw[4 * i + 2] | (raw[4 * i + 3] << 8));
                    l = a / 32767.0;  // wavfile.cpp:299-300
                    r = b / 32767.0;
                } else {
                    float a, b;
                    std::memcpy(&a, &raw[8 * i], 4);
                    std::memcpy(&b, &raw[8 * i + 4], 4);
                    l = a;
                    r = b;
                }
                frame[i] = swapIQ ? CPX(r * gain, l * gain) : CPX(l * gain, r * gain);
            }
            cb(frame.data(), framesPerBuffer);  // consumerWorker -> processIQData(bufPtr, m_framesPerBuffer), filesdrdevice.cpp:280
            frames++;
        }
        return frames;
    }
    void disconnectDevice()
    {
        if (f) std::fclose(f);
        f = nullptr;
    }
    ~FileSdrFeeder() { disconnectDevice(); }

private:
    bool closeFail()
    {
        disconnectDevice();
        return false;
    }
    std::FILE *f = nullptr;
    CB_ProcessIQData cb;
    uint16_t framesPerBuffer = 0;
    uint32_t sampleRate = 0, dataBytes = 0;
    int format = 0, channels = 0, bits = 0;
    long dataStart = 0;
    double gain = 1.0;
    bool swapIQ = false;
};

}  // namespace pebblegpu
#endif
