// filesdr_chain.cpp -- BASELINE config 1 plumbing: a FileSDRDevice-style feeder pumps an IQ .wav through the
// CB_ProcessIQData callback into the receive chain, audio comes back through CB_ProcessAudioData.
//
//   filesdr_chain <in.wav> feed <out.bin>                               feeder only (no GPU): dump the frames it delivers
//   filesdr_chain <in.wav> am   <out.bin> <mixer_hz> <lo> <hi> [fft taps]   full chain on GPU 0, AM demod, dump audio
// Output: raw interleaved doubles (re, im).
//
// Build: g++ -std=c++14 -O2 -Iinclude examples/filesdr_chain.cpp -Lpebblesdr_amd -lpebblegpu -Wl,-rpath,$PWD/pebblesdr_amd
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "pebblegpu_steps.hpp"

using namespace pebblegpu;

int main(int argc, char **argv)
{
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <in.wav> feed|am <out.bin> [mixer lo hi [fft taps]]\n", argv[0]);
        return 2;
    }
    std::FILE *out = std::fopen(argv[3], "wb");
    if (!out) return 2;
    const uint16_t framesPerBuffer = 2048;  // settings.cpp:57
    FileSdrFeeder dev;
    if (!dev.connectDevice(argv[1])) {
        std::fprintf(stderr, "cannot open %s as a 2-channel PCM16/float32 WAV\n", argv[1]);
        return 1;
    }
    uint64_t delivered = 0;
    if (!std::strcmp(argv[2], "feed")) {
        dev.initialize([&](CPX *buf, uint16_t n) { std::fwrite(buf, sizeof(CPX), n, out); }, framesPerBuffer);
        delivered = dev.start();
    } else {
        if (argc < 7) return 2;
        const uint32_t fft = argc > 7 ? (uint32_t)std::atoi(argv[7]) : 0, taps = argc > 8 ? (uint32_t)std::atoi(argv[8]) : 0;
        Receiver rx(dev.getSampleRate(), framesPerBuffer, false, 4096,
                    [&](CPX *audio, uint16_t n) { std::fwrite(audio, sizeof(CPX), n, out); }, fft, taps);
        if (rx.lastStatus() != 0) return 1;
        rx.demodModeChanged(dmAM);
        rx.mixerChanged(std::atoi(argv[4]));
        rx.filterChanged(std::atoi(argv[5]), std::atoi(argv[6]));
        // what Receiver::turnPowerOn does: bind processIQData as the plugin's callback (receiver.cpp:135-138)
        dev.initialize(std::bind(&Receiver::processIQData, &rx, std::placeholders::_1, std::placeholders::_2), framesPerBuffer);
        delivered = dev.start();
        if (rx.lastStatus() != 0) return 1;
    }
    std::fclose(out);
    std::printf("%llu frames\n", (unsigned long long)delivered);
    return 0;
}
